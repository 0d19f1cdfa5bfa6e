#!/usr/bin/env python3
"""Per-parameter gradient comparison HIP(fp32) / HIP(bf16) vs the golden fixture (GPU box)."""
import os, sys
import numpy as np, torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pl_yolo_amd

g = dict(np.load(os.path.join(ROOT, "tests/golden/network_yolox_test.npz")))
cfg = yaml.safe_load(open(os.path.join(ROOT, "configs/model/yolox/yolox_test.yaml")))
x = torch.from_numpy(g["x"]).cuda(); labels = torch.from_numpy(g["labels"]).cuda()
res = {}
for dt in ("fp32", "bf16"):
    m = pl_yolo_amd.build_model(cfg, int(g["num_classes"]))
    m.load_state_dict({k[6:]: torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith("state/")})
    m.compute_dtype = dt
    m = m.cuda().train()
    out = m(x, labels); out["loss"].backward(); torch.cuda.synchronize()
    res[dt] = {n: p.grad.cpu().numpy() for n, p in m.named_parameters() if p.grad is not None}
    print(dt, "loss", float(out["loss"]), "ref", float(g["out/loss"]))
def cos(a, b):
    a = a.ravel().astype(np.float64); b = b.ravel().astype(np.float64)
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))
print("%-46s %10s %9s %9s %8s %8s" % ("param", "|ref|max", "rel32", "rel16", "cos16", "nrm16/ref"))
for n in res["fp32"]:
    ref = g["grad/" + n]
    a, b = res["fp32"][n], res["bf16"][n]
    s = max(np.abs(ref).max(), 1e-12)
    print("%-46s %10.3g %9.2g %9.2g %8.4f %8.3f" % (n, s, np.abs(a - ref).max() / s, np.abs(b - ref).max() / s, cos(b, ref), np.linalg.norm(b) / (np.linalg.norm(ref) + 1e-30)))
