#!/usr/bin/env python3
"""Where the bf16 gradients of the toy YOLOv7 leave the fp32 ones (GPU box): per-parameter cosine in network order, on weights
warmed by N fp32 steps; optional environment variants (fusions off) to tell numerics from plumbing.  python tools/diag_v7warm.py [steps] [lr]"""
import os, sys
import numpy as np, torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pl_yolo_amd
from pl_yolo_amd.trainer import Trainer
DEV = "cuda:0"
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "model", "yolov7", "yolov7_test.yaml")))
torch.manual_seed(21)
warm = pl_yolo_amd.build_model(cfg, 3); warm.compute_dtype = "fp32"; warm = warm.to(DEV)
gen = torch.Generator().manual_seed(77)
def batch():
    x = torch.rand(4, 3, 256, 256, generator=gen) * 255
    lab = torch.zeros(4, 6, 5)
    for b, n in enumerate([3, 2, 0, 4]):
        lab[b, :n, 0] = torch.randint(0, 3, (n,), generator=gen).float()
        lab[b, :n, 1:3] = (0.15 + 0.7 * torch.rand(n, 2, generator=gen)) * 256
        lab[b, :n, 3:5] = 24.0 + torch.rand(n, 2, generator=gen) * 0.4 * 256
    return x.to(DEV), lab.to(DEV)
data = [batch() for _ in range(3)]
tr = Trainer(warm, learning_rate=lr, momentum=0.9, warmup=0.1, total_steps=max(400, steps), ema=False)
losses = [float(tr.train_step(*data[i % 3])["loss"].detach()) for i in range(steps)]
print("warm-up", losses[:3], "->", losses[-3:])
state = {k: v.detach().clone() for k, v in warm.state_dict().items()}
x, lab = batch()
def step(dt, env):
    old = {k: os.environ.get(k) for k in env}; os.environ.update(env)
    try:
        m = pl_yolo_amd.build_model(cfg, 3); m.load_state_dict(state); m.compute_dtype = dt; m = m.to(DEV).train()
        out = m(x, lab); out["loss"].backward(); torch.cuda.synchronize()
        return float(out["loss"].detach()), {n: p.grad.detach().double().cpu() for n, p in m.named_parameters() if p.grad is not None}
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
l32, g32 = step("fp32", {})
variants = [("bf16 default", {}), ("bf16 no fusions, one lane", {"PLYOLO_FUSE_PWBWD": "0", "PLYOLO_FUSE_BNRED": "0", "PLYOLO_FUSE_BNBWD": "0", "PLYOLO_LANES": "0", "PLYOLO_S2D": "0"})]
for name, env in variants:
    l, g = step("bf16", env)
    a = torch.cat([g[n].reshape(-1) for n in g32]); b = torch.cat([g32[n].reshape(-1) for n in g32])
    print("%s: loss %.5f (fp32 %.5f)  all-parameter cosine %.5f" % (name, l, l32, float((a * b).sum() / (a.norm() * b.norm()))))
    if not env:
        for n in g32:
            c = float((g[n].reshape(-1) * g32[n].reshape(-1)).sum() / (g[n].norm() * g32[n].norm() + 1e-30))
            print("   %-52s %8d  |g32| %.3e  cos %.4f" % (n, g32[n].numel(), float(g32[n].norm()), c))
l2, g2 = step("fp32", {})
a = torch.cat([g2[n].reshape(-1) for n in g32]); b = torch.cat([g32[n].reshape(-1) for n in g32])
print("fp32 twice: cosine %.8f" % float((a * b).sum() / (a.norm() * b.norm())))
