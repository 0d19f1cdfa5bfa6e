import os, sys, torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pl_yolo_amd, bench
import hiputil as hu
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = int(sys.argv[2]) if len(sys.argv) > 2 else 320
dt = sys.argv[3] if len(sys.argv) > 3 else "bf16"
cfg = yaml.safe_load(open(os.path.join(ROOT, "configs/model/yolox/yolox_s.yaml")))
torch.manual_seed(96)
model = pl_yolo_amd.build_model(cfg, 80); model.compute_dtype = dt
model = model.to("cuda:0")
sd0 = {k: v.clone() for k, v in model.state_dict().items()}
imgs, labels = bench.synthetic(B, S, 80, 1234)
imgs, labels = imgs.cuda(), labels.cuda()
perm = torch.randperm(B, generator=torch.Generator().manual_seed(3)).cuda()
def maps(x):
    model.load_state_dict(sd0); model.train()
    with torch.no_grad():
        return [m.float().clone() for m in model(x, None)]
m1 = maps(imgs); m2 = maps(imgs[perm].contiguous())
for a, b in zip(m1, m2):
    print("map", tuple(a.shape), "max |diff| permuted vs original %.3g (max %.3g)" % (float((a[perm] - b).abs().max()), float(a.abs().max())))
def step(x, l):
    model.load_state_dict(sd0); model.train(); model.zero_grad(set_to_none=True)
    out = model(x, l); out["loss"].backward(); torch.cuda.synchronize()
    return {k: float(v) for k, v in out.items()}, {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
l1, g1 = step(imgs, labels); l2, g2 = step(imgs[perm].contiguous(), labels[perm].contiguous())
print(l1); print(l2)
worst = sorted(((hu.cossim(g1[n], g2[n]), n) for n in g1))[:12]
for c, n in worst: print("  cos %.5f %s  |g| %.3g" % (c, n, float(g1[n].abs().max())))
a = torch.cat([g.flatten() for g in g1.values()]); b = torch.cat([g2[n].flatten() for n in g1])
print("all cos %.6f" % hu.cossim(a, b))
