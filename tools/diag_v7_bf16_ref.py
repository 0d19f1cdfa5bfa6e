#!/usr/bin/env python3
"""Build container only (imports /root/reference): how far do the REFERENCE's own bf16 gradients of the toy YOLOv7 lie from its fp32
gradients?  The reference model is warmed by N fp32 SGD steps on the CPU, then one step on a held-out batch in fp32 and under
torch.autocast(cpu, bfloat16) from the same state: all-parameter cosine and per-tensor cosines.  The yardstick for
tests/test_gpu_yolov7.py: test_v7_bf16_gradients_track_fp32_on_warm_weights.   python tools/diag_v7_bf16_ref.py [steps] [lr]"""
import os, sys
import torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")
from PL_Modules.build_detection import build_model
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "model", "yolov7", "yolov7_test.yaml")))
torch.manual_seed(21)
model = build_model(cfg, 3).train()
gen = torch.Generator().manual_seed(77)
def batch():
    x = torch.rand(4, 3, 256, 256, generator=gen) * 255
    lab = torch.zeros(4, 6, 5)
    for b, n in enumerate([3, 2, 0, 4]):
        lab[b, :n, 0] = torch.randint(0, 3, (n,), generator=gen).float()
        lab[b, :n, 1:3] = (0.15 + 0.7 * torch.rand(n, 2, generator=gen)) * 256
        lab[b, :n, 3:5] = 24.0 + torch.rand(n, 2, generator=gen) * 0.4 * 256
    return x, lab
data = [batch() for _ in range(3)]
opt = torch.optim.SGD(model.parameters(), lr=lr, momentum=0.9)
losses = []
for i in range(steps):
    out = model(*[t.clone() for t in data[i % 3]])
    opt.zero_grad(); out["loss"].backward(); opt.step()
    losses.append(float(out["loss"]))
print("warm-up", losses[:3], "->", losses[-3:])
state = {k: v.clone() for k, v in model.state_dict().items()}
x, lab = batch()
def step(bf16):
    m = build_model(cfg, 3).train(); m.load_state_dict(state)
    if bf16:
        with torch.autocast("cpu", dtype=torch.bfloat16):
            out = m(x.clone(), lab.clone())
    else:
        out = m(x.clone(), lab.clone())
    out["loss"].float().backward()
    return float(out["loss"]), {n: p.grad.double() for n, p in m.named_parameters() if p.grad is not None}
l32, g32 = step(False)
l16, g16 = step(True)
a = torch.cat([g16[n].reshape(-1) for n in g32]); b = torch.cat([g32[n].reshape(-1) for n in g32])
print("reference: loss fp32 %.5f, autocast bf16 %.5f; all-parameter gradient cosine %.5f" % (l32, l16, float((a * b).sum() / (a.norm() * b.norm()))))
for n in g32:
    if n.startswith("head.") or ".n5." in n or ".n3." in n or "stem" in n or "p5_p4.conv5" in n:
        c = float((g16[n].reshape(-1) * g32[n].reshape(-1)).sum() / (g16[n].norm() * g32[n].norm() + 1e-30))
        print("   %-48s %7d  cos %.4f" % (n, g32[n].numel(), c))
