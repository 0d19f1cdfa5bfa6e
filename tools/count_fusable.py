#!/usr/bin/env python3
"""Analysis (GPU box): for every conv unit, who writes the LAST contribution to its output gradient?  Units whose
last writer is a conv data-gradient covering exactly that tensor could take the BatchNorm-backward reduction in
that dgrad's store loop.  Prints the count and their share of the reduce time."""
import os, sys
import torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, pl_yolo_amd
from pl_yolo_amd import graph as G

log = []
orig = G.Graph.grad_mode
cur = {"op": None, "what": None}
def patched(self, a):
    log.append((id(a.storage), a.c_off, a.C, a.M, cur["op"], cur["what"]))
    return orig(self, a)
G.Graph.grad_mode = patched
for cls in (G.ConvUnitOp, G.ConvPairOp, G.HeadPredOp, G.UpsampleOp, G.SppPoolsOp, G.CopyOp):
    def wrap(c):
        ob = c.bwd
        def bwd(self):
            cur["op"], cur["what"] = self, c.__name__
            return ob(self)
        c.bwd = bwd
    wrap(cls)

cfg = yaml.safe_load(open(os.path.join(ROOT, "configs/model/yolox/yolox_s.yaml")))
torch.manual_seed(96)
model = pl_yolo_amd.build_model(cfg, 80); model.compute_dtype = "bf16"
model = model.to("cuda:0").train()
imgs, labels = bench.synthetic(32, 640, 80, 1234)
out = model(imgs.cuda(), labels.cuda()); out["loss"].backward(); torch.cuda.synchronize()
r = model.runner()
s = [v for k, v in r.sessions.items() if k[4] == "train"][0]
g = s.graph if hasattr(s, "graph") else s.g
units = [op for op in g.ops if isinstance(op, G.ConvUnitOp) and op.bn is not None]
pairs = [op for op in g.ops if isinstance(op, G.ConvPairOp)]
tot = fus = 0.0
nf = 0
for u in units:
    key = (id(u.out.storage), u.out.c_off, u.out.C)
    writes = [w for w in log if w[0] == key[0] and not (w[1] + w[2] <= key[1] or key[1] + key[2] <= w[1])]
    last = writes[-1] if writes else None
    by = u.out.M * u.Cout * 2 * 2
    tot += by
    ok = last is not None and last[5] in ("ConvUnitOp", "ConvPairOp", "HeadPredOp") and (last[1], last[2]) == (key[1], key[2])
    if ok:
        fus += by; nf += 1
    print("%-10s M=%7d C=%4d writers=%d last=%s%s" % ("unit", u.out.M, u.Cout, len(writes), last[5] if last else None, "  FUSABLE" if ok else ""))
print("single units: %d, fusable %d, byte share %.2f; pairs (not counted): %d" % (len(units), nf, fus / tot, len(pairs)))
