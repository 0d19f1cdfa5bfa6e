"""CPU: BASELINE.json configs[0] -- "YOLOX-nano 416x416 bs=4 random COCO-shaped tensors, CPU-only PyTorch reference
path" -- through the oracle, against the fixture the REFERENCE wrote for exactly this workload
(tools/gen_golden.py: gen_cfg1; seed-96 weights, the benchmark's synthetic batch).  The GPU side of the same
configuration is tests/test_gpu_configs.py::test_cfg1_nano416_b4_fp32_vs_reference_fixture."""
import os

import numpy as np
import torch
import yaml

from conftest import ROOT, load_golden
from oracle import net as onet, detector as odet


def test_oracle_cfg1_nano416_b4_vs_reference():
    g = load_golden("cfg1_nano416")
    with open(os.path.join(ROOT, "configs", "model", "yolox", "yolox_nano.yaml")) as f:
        cfg = yaml.safe_load(f)
    nc = int(g["num_classes"])
    torch.manual_seed(int(g["seed_weights"]))
    state = onet.build_state(cfg, nc)
    assert np.array_equal(state["backbone.stem.conv.conv.weight"].numpy(), g["stem_weight"])
    imgs, labels = odet.synthetic_batch(int(g["batch"]), int(g["size"]), nc, seed=int(g["seed_data"]))
    out, grads = odet.train_step_grads(state, cfg, nc, imgs, labels)
    for k in ("loss", "loss_iou", "loss_obj", "loss_cls"):
        got, want = float(out[k]), float(g["out/" + k])
        assert abs(got - want) <= 2e-6 * max(1.0, abs(want)), (k, got, want)
    for k in [k for k in g if k.startswith("grad/")]:
        ref = torch.from_numpy(g[k])
        assert float((grads[k[5:]] - ref).abs().max()) <= 1e-4 * max(float(ref.abs().max()), 1e-6), k   # fp32 summation order (thread count) moves the stem gradient by ~2e-5
    gsq = float(sum((v.double() ** 2).sum() for v in grads.values()))
    assert abs(gsq - float(g["grad_sq_sum"])) <= 1e-4 * float(g["grad_sq_sum"])
