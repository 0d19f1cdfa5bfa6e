#!/usr/bin/env python3
"""Layer-by-layer activation / activation-gradient comparison HIP bf16 vs HIP fp32 (GPU box)."""
import os, sys
import numpy as np, torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pl_yolo_amd
from pl_yolo_amd import graph as G
name, B, S, nc = (sys.argv[1] if len(sys.argv) > 1 else "yolox_s"), 2, 320, 80
cfg = yaml.safe_load(open(os.path.join(ROOT, "configs/model/yolox/%s.yaml" % name)))
torch.manual_seed(96)
base = pl_yolo_amd.build_model(cfg, nc)
sd = {k: v.clone() for k, v in base.state_dict().items()}
g = torch.Generator().manual_seed(1234)
x = (torch.rand(B, 3, S, S, generator=g) * 255).cuda()
sess = {}
rs = None
for dt in ("fp32", "bf16"):
    m = pl_yolo_amd.build_model(cfg, nc); m.load_state_dict(sd); m.compute_dtype = dt
    m = m.cuda().train()
    maps = m(x)
    if rs is None:
        gen = torch.Generator().manual_seed(4)
        rs = [torch.randn(mm.shape, generator=gen).cuda() for mm in maps]
    sum((mm * r).sum() for mm, r in zip(maps, rs)).backward(); torch.cuda.synchronize()
    r = m.runner()
    s = [v for k, v in r.sessions.items() if k[4] == "maps_grad"][0]
    sess[dt] = (m, s)
def view(gph, a, grad=False):
    st = a.storage
    t = (st.grad if grad else st.tensor)
    return t.view(st.rows, st.ld)[:, a.c_off:a.c_off + a.C].float()
def rr(a, b):
    return float((a - b).pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-30))
ops32 = [o for o in sess["fp32"][1].g.ops if isinstance(o, G.ConvUnitOp)]
ops16 = [o for o in sess["bf16"][1].g.ops if isinstance(o, G.ConvUnitOp)]
names = {}
for n, mod in sess["fp32"][0].named_modules():
    if hasattr(mod, "conv") and isinstance(getattr(mod, "conv"), torch.nn.Conv2d):
        names[id(mod.conv.weight)] = n
g32, g16 = sess["fp32"][1].g, sess["bf16"][1].g
print("%-34s %9s %9s %9s" % ("conv unit", "z rms", "out rms", "dout rms"))
for a, b in zip(ops32, ops16):
    nm = names.get(id(a.pc.sources[0][0]), "?")
    z32 = a.z.tensor.view(a.z.rows, a.z.ld).float(); z16 = b.z.tensor.view(b.z.rows, b.z.ld).float()
    o = rr(view(g16, b.out), view(g32, a.out))
    d = rr(view(g16, b.out, True), view(g32, a.out, True)) if a.out.storage.grad is not None else float("nan")
    print("%-34s %9.4f %9.4f %9.4f" % (nm, rr(z16, z32), o, d))
