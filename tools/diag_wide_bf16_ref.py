#!/usr/bin/env python3
"""Build container only (imports /root/reference): yardstick for the bf16 bounds of tests/test_gpu_configs.py::test_wide_models_vs_reference_fixture.
The REFERENCE's own yolox_l / yolox_x / yolov7 (seed-96 initialisation, the fixture's batch) under torch.autocast(cpu, bfloat16) against
its fp32 run: raw head maps (relative rms), loss, prediction-bias gradient cosines.   python tools/diag_wide_bf16_ref.py"""
import os, sys
import numpy as np, torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")
from PL_Modules.build_detection import build_model
torch.set_num_threads(8)
for name in (sys.argv[1:] or ["yolox_l", "yolox_x", "yolov7"]):
    fam = "yolov7" if name.startswith("yolov7") else "yolox"
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "model", fam, name + ".yaml")))
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "wide_%s.npz" % name)))
    B, S, ngt, mgt, C = int(g["batch"]), int(g["size"]), int(g["num_gt"]), int(g["max_gt"]), 80
    gen = torch.Generator().manual_seed(int(g["seed_data"]))
    imgs = torch.rand(B, 3, S, S, generator=gen) * 255
    labels = torch.zeros(B, mgt, 5)
    labels[:, :ngt, 0] = torch.randint(0, C, (B, ngt), generator=gen).float()
    labels[:, :ngt, 1:3] = (0.15 + 0.7 * torch.rand(B, ngt, 2, generator=gen)) * S
    labels[:, :ngt, 3:5] = 8 + torch.rand(B, ngt, 2, generator=gen) * 0.3 * S
    res = {}
    for bf in (False, True):
        torch.manual_seed(96)
        m = build_model(cfg, C).train()
        sd0 = {k: v.clone() for k, v in m.state_dict().items()}
        def run(fn):
            if bf:
                with torch.autocast("cpu", dtype=torch.bfloat16):
                    return fn()
            return fn()
        with torch.no_grad():
            maps = [t.float() for t in run(lambda: m(imgs, None))]
        m.load_state_dict(sd0)
        out = run(lambda: m(imgs, labels))
        loss = out["loss"].float().sum()
        loss.backward()
        res[bf] = (maps, float(loss), {n: p.grad.double().clone() for n, p in m.named_parameters() if p.grad is not None})
    (m32, l32, g32), (m16, l16, g16) = res[False], res[True]
    rms = [float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()) for a, b in zip(m16, m32)]
    biases = [n for n in g32 if n.startswith("head") and n.endswith(".bias") and "norm" not in n]
    cos = {n: float((g16[n] * g32[n]).sum() / (g16[n].norm() * g32[n].norm() + 1e-300)) for n in biases}
    a = torch.cat([g16[n].reshape(-1) for n in g32]); b = torch.cat([g32[n].reshape(-1) for n in g32])
    print("%s reference autocast-bf16 vs fp32: map rel rms %s | loss %.5f vs %.5f (rel %.2e) | bias-gradient cosine min %.4f | all-parameter cosine %.4f"
          % (name, ", ".join("%.3g" % r for r in rms), l16, l32, abs(l16 - l32) / l32, min(cos.values()), float((a * b).sum() / (a.norm() * b.norm()))), flush=True)
