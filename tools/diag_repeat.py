#!/usr/bin/env python3
"""Run-to-run repeatability per parameter: two identical training steps, report which gradients differ.
    python tools/diag_repeat.py [model] [size] [batch]"""
import os
import sys

import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pl_yolo_amd  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "yolox_x"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 1280
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
fam = "yolov7" if name.startswith("yolov7") else "yolox"
cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "model", fam, name + ".yaml")))
torch.manual_seed(96)
model = pl_yolo_amd.build_model(cfg, 80)
model.compute_dtype = os.environ.get("PLYOLO_DTYPE", "bf16")
model = model.to("cuda:0").train()
sd0 = {k: v.clone() for k, v in model.state_dict().items()}
gen = torch.Generator().manual_seed(1234)
imgs = (torch.rand(B, 3, size, size, generator=gen) * 255).to("cuda:0")
labels = torch.zeros(B, 100, 5)
labels[:, :30, 0] = torch.randint(0, 80, (B, 30), generator=gen).float()
labels[:, :30, 1:3] = (0.15 + 0.7 * torch.rand(B, 30, 2, generator=gen)) * size
labels[:, :30, 3:5] = 8 + torch.rand(B, 30, 2, generator=gen) * 0.3 * size
labels = labels.to("cuda:0")


def step():
    model.load_state_dict(sd0)
    model.zero_grad(set_to_none=True)
    out = model(imgs, labels)
    out["loss"].backward()
    torch.cuda.synchronize()
    return float(out["loss"]), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}


runs = [step() for _ in range(3)]
print("losses", [r[0] for r in runs])
names = list(runs[0][1])
bad = 0
for n in names:
    a = runs[0][1][n]
    d = max(float((a - r[1][n]).abs().max()) for r in runs[1:])
    if d > 0:
        bad += 1
        d12 = float((runs[1][1][n] - runs[2][1][n]).abs().max())
        if bad <= int(os.environ.get("DIAG_MAX", "40")):
            print("%-60s max|g| %.4g  diff(run0, later) %.4g  diff(run1, run2) %.4g" % (n, float(a.abs().max()), d, d12))
print("%d of %d gradient tensors differ between runs" % (bad, len(names)))
