#!/usr/bin/env python3
"""GPU time between the plans of one training step (GPU box): events around every plan replay of the normal
bench loop -> forward plan, glue (loss scalars / autograd), backward plan, step-to-step glue."""
import os, sys, time
import torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, pl_yolo_amd
from pl_yolo_amd import runner as R

cfg = yaml.safe_load(open(os.path.join(ROOT, "configs/model/yolox/yolox_s.yaml")))
torch.manual_seed(96)
model = pl_yolo_amd.build_model(cfg, 80); model.compute_dtype = "bf16"
model = model.to("cuda:0").train()
imgs, labels = bench.synthetic(32, 640, 80, 1234)
imgs, labels = imgs.cuda(), labels.cuda()
marks = []
orig = R.Runner._run_plan if hasattr(R, "Runner") else None
cls = [v for v in vars(R).values() if isinstance(v, type) and hasattr(v, "_run_plan")][0]
orig = cls._run_plan
def patched(self, plan):
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); t0 = time.perf_counter(); orig(self, plan); t1 = time.perf_counter(); b.record()
    marks.append((a, b, t1 - t0))
cls._run_plan = patched
def step():
    out = model(imgs, labels); model.zero_grad(set_to_none=True); out["loss"].backward()
for _ in range(5): step()
torch.cuda.synchronize(); marks.clear()
t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
ev = marks
f = sum(ev[i][0].elapsed_time(ev[i][1]) for i in range(0, 20, 2)) / 10
b = sum(ev[i][0].elapsed_time(ev[i][1]) for i in range(1, 20, 2)) / 10
g1 = sum(ev[i][1].elapsed_time(ev[i + 1][0]) for i in range(0, 20, 2)) / 10
g2 = sum(ev[i][1].elapsed_time(ev[i + 1][0]) for i in range(1, 19, 2)) / 9
cf = sum(ev[i][2] for i in range(0, 20, 2)) / 10 * 1e3; cb = sum(ev[i][2] for i in range(1, 20, 2)) / 10 * 1e3
print("step %.3f ms | GPU: fwd plan %.3f, fwd->bwd glue %.3f, bwd plan %.3f, bwd->next fwd glue %.3f | host issue: fwd %.3f ms, bwd %.3f ms" % (dt * 1e3, f, g1, b, g2, cf, cb))
