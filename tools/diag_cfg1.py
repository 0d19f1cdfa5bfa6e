"""cfg1 (yolox_nano 416 B=4, random init): per-tensor gradient cosine of the HIP bf16 step against the reference fixture and against
the HIP fp32 step, in forward order -- where does the bf16 gradient of a random-initialised net decorrelate?"""
import os, sys
import numpy as np, torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pl_yolo_amd
g = dict(np.load(os.path.join(ROOT, "tests", "golden", "cfg1_nano416.npz")))
cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "model", "yolox", "yolox_nano.yaml")))
B, S = int(g["batch"]), int(g["size"])
gen = torch.Generator().manual_seed(int(g["seed_data"]))
imgs = torch.rand(B, 3, S, S, generator=gen) * 255
labels = torch.zeros(B, 100, 5)
labels[:, :30, 0] = torch.randint(0, 80, (B, 30), generator=gen).float()
labels[:, :30, 1:3] = (0.15 + 0.7 * torch.rand(B, 30, 2, generator=gen)) * S
labels[:, :30, 3:5] = 8 + torch.rand(B, 30, 2, generator=gen) * 0.3 * S
res = {}
for dt in ("fp32", "bf16"):
    torch.manual_seed(96)
    m = pl_yolo_amd.build_model(cfg, 80)
    m.compute_dtype = dt
    m = m.to("cuda").train()
    out = m(imgs.cuda(), labels.cuda())
    out["loss"].backward()
    torch.cuda.synchronize()
    res[dt] = ({n: p.grad.double().cpu() for n, p in m.named_parameters() if p.grad is not None}, float(out["loss"]), float(out["proportion"]) if "proportion" in out else None)
    print(dt, {k: float(v) for k, v in out.items()})
a, b = res["fp32"][0], res["bf16"][0]
cos = lambda u, v: float((u * v).sum() / max(float(u.norm() * v.norm()), 1e-30))
for i, n in enumerate(a):
    if i % 6 == 0 or ("grad/" + n) in g:
        extra = ""
        if ("grad/" + n) in g:
            r = torch.from_numpy(g["grad/" + n]).double()
            extra = " | vs fixture: fp32 %.5f bf16 %.5f" % (cos(a[n], r), cos(b[n], r))
        print("%-52s cos(bf16, fp32) %.4f  norm ratio %.3f%s" % (n, cos(a[n], b[n]), float(b[n].norm() / max(float(a[n].norm()), 1e-30)), extra))
allc = cos(torch.cat([v.flatten() for v in a.values()]), torch.cat([b[n].flatten() for n in a]))
print("all-parameter cosine bf16 vs fp32: %.5f" % allc)
