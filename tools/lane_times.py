#!/usr/bin/env python3
"""Where each launch lane of the forward / backward plan ends (GPU box): tells whether the weight-gradient lane
or a head-level lane outlives the main lane.   python tools/lane_times.py [model] [batch] [size]"""
import os, sys
import torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, pl_yolo_amd

name = sys.argv[1] if len(sys.argv) > 1 else "yolox_s"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
S = int(sys.argv[3]) if len(sys.argv) > 3 else 640
dev = torch.device("cuda:0")
family = "yolov7" if name.startswith("yolov7") else "yolox"
cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "model", family, name + ".yaml")))
torch.manual_seed(96)
model = pl_yolo_amd.build_model(cfg, 80)
model.compute_dtype = "bf16"
model = model.to(dev).train()
imgs, labels = bench.synthetic(B, S, 80, 1234)
imgs, labels = imgs.to(dev), labels.to(dev)
for _ in range(5):
    out = model(imgs, labels); model.zero_grad(set_to_none=True); out["loss"].backward()
torch.cuda.synchronize()
r = model.runner()
s = [v for k, v in r.sessions.items() if k[4] == "train"][0]
st = torch.cuda.current_stream().cuda_stream
for rep in range(4):
    r._focus(s, imgs)
    f = s.fwd.lane_times(st) if s.fwd.lanes() > 1 else None
    b = s.bwd.lane_times(st) if s.bwd.lanes() > 1 else None
    if rep:
        fmt = lambda v: "single lane" if v is None else "  ".join("lane%d %.3f" % (i, x) for i, x in enumerate(v[:-1])) + "  join %.3f ms" % v[-1]
        print("fwd:", fmt(f)); print("bwd:", fmt(b))
