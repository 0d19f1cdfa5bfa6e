#!/usr/bin/env python3
"""Inference throughput (eval forward + decode, and + postprocess/NMS) of a YOLOX config on the GPU box."""
import os, sys, time
import torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pl_yolo_amd
from pl_yolo_amd.postprocess import postprocess
name, size, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
cfg = yaml.safe_load(open(os.path.join(ROOT, "configs/model/yolox/%s.yaml" % name)))
torch.manual_seed(96)
m = pl_yolo_amd.build_model(cfg, 80); m.compute_dtype = "bf16"; m = m.cuda().eval()
x = (torch.rand(B, 3, size, size) * 255).cuda(); lab = torch.zeros(B, 1, 5).cuda()
for what in ("forward+decode", "forward+decode+postprocess"):
    with torch.no_grad():
        for _ in range(3):
            out = m(x, lab)
            if "post" in what: postprocess(out, 0.01, 0.65)
        torch.cuda.synchronize(); t0 = time.perf_counter(); n = 20
        for _ in range(n):
            out = m(x, lab)
            if "post" in what: postprocess(out, 0.01, 0.65)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print("%s %dx%d B=%d %s: %.2f ms/batch = %.0f img/s" % (name, size, size, B, what, dt * 1e3, B / dt))
