#!/usr/bin/env python3
"""Build container only (imports /root/reference): the yardstick for tests/test_gpu_yolov7.py::test_v7_full_width_warm_bf16_...
yolov7.yaml at FULL width through the REFERENCE on the CPU: N fp32 SGD steps (192x192, batch 4), then from that state on a held-out
batch (a) the gradient of a fixed random linear functional of the raw head maps and (b) the training step's gradient, each in fp32
and under torch.autocast(cpu, bfloat16): all-parameter cosine + the worst tensors.   python tools/diag_v7_full_bf16_ref.py [steps] [lr] [yolov7|yolox_l|yolox_x]"""
import os, sys
import torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")
from PL_Modules.build_detection import build_model
torch.set_num_threads(8)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
name = sys.argv[3] if len(sys.argv) > 3 else "yolov7"
cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "model", "yolov7" if name.startswith("yolov7") else "yolox", name + ".yaml")))
nc, S, B = 80, 192, 4
torch.manual_seed(96)
model = build_model(cfg, nc).train()
gen = torch.Generator().manual_seed(177)
def batch():
    x = torch.rand(B, 3, S, S, generator=gen) * 255
    lab = torch.zeros(B, 8, 5)
    for b, n in enumerate([3, 5, 1, 4]):
        lab[b, :n, 0] = torch.randint(0, nc, (n,), generator=gen).float()
        lab[b, :n, 1:3] = (0.15 + 0.7 * torch.rand(n, 2, generator=gen)) * S
        lab[b, :n, 3:5] = 16.0 + torch.rand(n, 2, generator=gen) * 0.4 * S
    return x, lab
data = [batch() for _ in range(4)]
opt = torch.optim.SGD(model.parameters(), lr=lr, momentum=0.9)
losses = []
for i in range(steps):
    out = model(*[t.clone() for t in data[i % 4]])
    opt.zero_grad(); out["loss"].sum().backward(); opt.step()
    losses.append(float(out["loss"].sum()))
print("warm-up", losses[:3], "->", losses[-3:], flush=True)
state = {k: v.clone() for k, v in model.state_dict().items()}
x, lab = batch()
with torch.no_grad():
    _m = build_model(cfg, nc).train(); _m.load_state_dict(state)
    cot = [torch.randn(mp.shape, generator=gen) for mp in _m(x.clone(), None)]
def step(bf16, with_loss):
    m = build_model(cfg, nc).train(); m.load_state_dict(state)
    def run():
        if with_loss:
            return m(x.clone(), lab.clone())["loss"].sum()
        return sum((mp.float() * c).sum() for mp, c in zip(m(x.clone(), None), cot))
    if bf16:
        with torch.autocast("cpu", dtype=torch.bfloat16):
            v = run()
    else:
        v = run()
    v.float().backward()
    return float(v), {n: p.grad.double() for n, p in m.named_parameters() if p.grad is not None}
for with_loss in (False, True):
    v32, g32 = step(False, with_loss)
    v16, g16 = step(True, with_loss)
    a = torch.cat([g16[n].reshape(-1) for n in g32]); b = torch.cat([g32[n].reshape(-1) for n in g32])
    per = {n: float((g16[n].reshape(-1) * g32[n].reshape(-1)).sum() / (g16[n].norm() * g32[n].norm() + 1e-30)) for n in g32}
    order = sorted(per, key=per.get)
    print("reference, %s: value fp32 %.5f, autocast bf16 %.5f; all-parameter gradient cosine %.5f; worst tensors %s"
          % ("training step" if with_loss else "maps functional", v32, v16, float((a * b).sum() / (a.norm() * b.norm())),
             ", ".join("%s %.3f" % (n, per[n]) for n in order[:4])), flush=True)
