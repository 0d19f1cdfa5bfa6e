#!/usr/bin/env python3
"""GPU box diagnostic: a full-width model warmed for N bf16 steps, then the raw head maps of one held-out batch in HIP fp32, HIP bf16 and
the CPU oracle (fp32) from the same state: per-level relative rms / largest value.   python tools/diag_wide_levels.py [yolox_l] [steps]"""
import os, sys
import numpy as np, torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pl_yolo_amd
from pl_yolo_amd.trainer import Trainer
from oracle import detector as odet
name = sys.argv[1] if len(sys.argv) > 1 else "yolox_l"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 41
fam = "yolov7" if name.startswith("yolov7") else "yolox"
cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "model", fam, name + ".yaml")))
nc, S, B = 80, 192, 4
dev = "cuda:0"
torch.manual_seed(96)
warm = pl_yolo_amd.build_model(cfg, nc)
warm.compute_dtype = os.environ.get("WARM_DTYPE", "bf16")
warm = warm.to(dev)
gen = torch.Generator().manual_seed(177)
def batch():
    x = torch.rand(B, 3, S, S, generator=gen) * 255
    lab = torch.zeros(B, 8, 5)
    for b, n in enumerate([3, 5, 1, 4]):
        lab[b, :n, 0] = torch.randint(0, nc, (n,), generator=gen).float()
        lab[b, :n, 1:3] = (0.15 + 0.7 * torch.rand(n, 2, generator=gen)) * S
        lab[b, :n, 3:5] = 16.0 + torch.rand(n, 2, generator=gen) * 0.4 * S
    return x.to(dev), lab.to(dev)
data = [batch() for _ in range(4)]
tr = Trainer(warm, learning_rate=0.01, momentum=0.9, warmup=1.0 / 4000, total_steps=4000, ema=False)
losses = [float(tr.train_step(*data[i % 4])["loss"].detach().sum()) for i in range(steps)]
print("warm-up", ["%.3f" % l for l in losses[:3]], "->", ["%.3f" % l for l in losses[-3:]])
state = {k: v.detach().clone() for k, v in warm.state_dict().items()}
x, lab = batch()
res = {}
for dt in ("fp32", "bf16"):
    m = pl_yolo_amd.build_model(cfg, nc)
    m.load_state_dict(state)
    m.compute_dtype = dt
    m = m.to(dev).train()
    with torch.no_grad():
        res[dt] = [t.float().cpu() for t in m(x, None)]
cpu_state = {k: v.detach().cpu().clone() for k, v in state.items()}
with torch.no_grad():
    res["oracle"] = [t.float() for t in odet.forward(cpu_state, cfg, nc, x.cpu(), None, training=True)]
for a, b in (("fp32", "oracle"), ("bf16", "oracle"), ("bf16", "fp32")):
    for i, (u, v) in enumerate(zip(res[a], res[b])):
        print("%-5s vs %-6s level %d: rel rms %.4g, max |%s| %.4g, max |diff| %.4g" % (a, b, i, float((u - v).pow(2).mean().sqrt() / v.pow(2).mean().sqrt()), b, float(v.abs().max()), float((u - v).abs().max())))
# BatchNorm running variances: degenerate channels?
rv = [(k, v) for k, v in state.items() if k.endswith("running_var")]
small = sorted(((float(v.min()), k) for k, v in rv))[:5]
print("smallest running_var:", small)
if os.environ.get("LAYERS"):
    # layer by layer: raw conv output z and activated output of every BaseConv unit, bf16 against fp32, same warm state (PLYOLO_PAIR=0 set by the caller)
    from pl_yolo_amd import graph as G
    sess = {}
    for dt in ("fp32", "bf16"):
        m = pl_yolo_amd.build_model(cfg, nc); m.load_state_dict(state); m.compute_dtype = dt
        m = m.to(dev).train()
        maps = m(x)
        sum(mm.float().sum() for mm in maps).backward(); torch.cuda.synchronize()
        s = [v for k, v in m.runner().sessions.items() if k[4] == "maps_grad"][0]
        sess[dt] = (m, s)
    def view(a):
        st = a.storage
        return st.tensor.view(st.rows, st.ld)[:, a.c_off:a.c_off + a.C].float()
    def rr(a, b):
        return float((a - b).pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-30))
    ops32 = [o for o in sess["fp32"][1].g.ops if isinstance(o, G.ConvUnitOp)]
    ops16 = [o for o in sess["bf16"][1].g.ops if isinstance(o, G.ConvUnitOp)]
    names = {}
    for n, mod in sess["fp32"][0].named_modules():
        if hasattr(mod, "conv") and isinstance(getattr(mod, "conv"), torch.nn.Conv2d):
            names[id(mod.conv.weight)] = n
    print("%-36s %5s %11s %9s %9s %10s %10s" % ("conv unit", "k", "M x Cout", "z rms", "out rms", "min var32", "min var16"))
    for a, b in zip(ops32, ops16):
        nm = names.get(id(a.pc.sources[0][0]), "?")
        z32 = a.z.tensor.view(a.z.rows, a.z.ld).float(); z16 = b.z.tensor.view(b.z.rows, b.z.ld).float()
        print("%-36s %5d %5dx%-5d %9.4f %9.4f %10.3g %10.3g" % (nm, a.k, z32.shape[0], a.Cout, rr(z16, z32), rr(view(b.out), view(a.out)),
                                                        float(z32.var(0).min()), float(z16.var(0).min())))
