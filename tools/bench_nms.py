#!/usr/bin/env python3
"""NMS line of bench.py alone (16 x 1000 boxes); run from the root of the tree to measure (cd _prev && python tools/bench_nms.py)."""
import json, os, sys
sys.path.insert(0, os.getcwd())
import torch
import bench
print(os.path.dirname(bench.__file__), json.dumps(bench.nms_bench(torch.device("cuda", 0), reps=200, cpu=False)))
