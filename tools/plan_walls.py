#!/usr/bin/env python3
"""Wall time of the forward and the backward plan replayed alone (GPU box), for lane / ablation variants given in the environment:
tells how much of a plan's wall time is launch gaps and cross-lane contention (compare with the sum of the stand-alone launch times of
bench.py --profile-out).   python tools/plan_walls.py [model] [batch] [size]"""
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
def wall(plan, n=20):
    for _ in range(3): plan.run(st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): plan.run(st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
r._focus(s, imgs)
print("env %s: fwd plan %.3f ms (%d launches, %d lanes), bwd plan %.3f ms (%d launches, %d lanes)"
      % ({k: v for k, v in os.environ.items() if k.startswith("PLYOLO_")}, wall(s.fwd), s.fwd.size(), s.fwd.lanes(), wall(s.bwd), s.bwd.size(), s.bwd.lanes()))
