"""Fused (PLYOLO_FUSE_BNRED=1) against separate BatchNorm-backward reductions on the warm yolox_s fixture: per unit, in backward
order, the relative difference of the two slot sums -- a coverage bug shows as an O(1) difference at one unit, rounding as a slow drift."""
import os, sys
import numpy as np, torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pl_yolo_amd
from pl_yolo_amd import graph as G
from pl_yolo_amd._lib import STAT_SLOTS
from conftest import load_golden, warm_s_state
g = load_golden("network_yolox_s_warm")
cfg = yaml.safe_load(open(os.path.join(ROOT, "configs/model/yolox/yolox_s.yaml")))
def run(env):
    os.environ.update(env)
    m = pl_yolo_amd.build_model(cfg, int(g["num_classes"]))
    m.load_state_dict(warm_s_state(g)); m.compute_dtype = "bf16"; m = m.to("cuda").train()
    out = m(torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["labels"]).cuda())
    out["loss"].backward(); torch.cuda.synchronize()
    s = [s for k, s in m.runner().sessions.items() if k[4] == "train"][0]
    sums = []
    for op in s.g.ops:
        if hasattr(op, "slot_off") and isinstance(op, (G.ConvUnitOp, G.ConvPairOp)):
            Cc = op.Cout
            a = s.g.bstat_arena[op.slot_off: op.slot_off + STAT_SLOTS * 2 * Cc].view(STAT_SLOTS, 2, Cc).sum(0).cpu()
            sums.append((op.index, type(op).__name__, Cc, op.out.M if hasattr(op, "out") else 0, getattr(op, "red_done", False), a))
    return sums, {n: p.grad.double().cpu() for n, p in m.named_parameters() if p.grad is not None}
s0, g0 = run({"PLYOLO_FUSE_BNRED": "0"})
s1, g1 = run({"PLYOLO_FUSE_BNRED": "1"})
for (i, t, Cc, M, rd0, a), (_, _, _, _, rd1, b) in sorted(zip(s0, s1), key=lambda z: -z[0][0]):
    e = float((a - b).abs().max() / max(float(a.abs().max()), 1e-30))
    print("op %3d %-11s C=%4d M=%6d fused=%d  max |diff| / max |sum| = %.2e" % (i, t, Cc, M, rd1, e))
worst = sorted(((float((g0[n] - g1[n]).abs().max() / max(float(g0[n].abs().max()), 1e-12)), n) for n in g0), reverse=True)[:8]
for e, n in worst: print("%.2e  %s" % (e, n))
