#!/usr/bin/env python3
"""Per-parameter gradient cosine, labels=None path, fixed upstream grads: HIP bf16 vs HIP fp32 (GPU box)."""
import os, sys
import numpy as np, torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pl_yolo_amd
name, B, S, nc = (sys.argv[1] if len(sys.argv) > 1 else "yolox_s"), 2, 320, 80
cfg = yaml.safe_load(open(os.path.join(ROOT, "configs/model/yolox/%s.yaml" % name)))
torch.manual_seed(96)
base = pl_yolo_amd.build_model(cfg, nc)
sd = {k: v.clone() for k, v in base.state_dict().items()}
g = torch.Generator().manual_seed(1234)
x = (torch.rand(B, 3, S, S, generator=g) * 255).cuda()
res = {}
rs = None
for dt in ("fp32", "bf16"):
    m = pl_yolo_amd.build_model(cfg, nc); m.load_state_dict(sd); m.compute_dtype = dt
    m = m.cuda().train()
    maps = m(x)
    if rs is None:
        gen = torch.Generator().manual_seed(4)
        rs = [torch.randn(mm.shape, generator=gen).cuda() for mm in maps]
    sum((mm * r).sum() for mm, r in zip(maps, rs)).backward(); torch.cuda.synchronize()
    res[dt] = {n: p.grad.double().cpu() for n, p in m.named_parameters() if p.grad is not None}
def cos(a, b):
    a = a.flatten(); b = b.flatten(); return float(a @ b / (a.norm() * b.norm() + 1e-30))
for n in res["fp32"]:
    a, b = res["bf16"][n], res["fp32"][n]
    print("%-46s cos %.4f  nrm16/nrm32 %.3f  |g32| %.3g" % (n, cos(a, b), float(a.norm() / (b.norm() + 1e-30)), float(b.norm())))
