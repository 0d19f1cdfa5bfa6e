"""CPU: the e-yolox oracle (oracle/net_e.py: depthwise 3x3 units, BatchNorm-free pointwise units, bicubic upsampling, the
four-tap CSP layer) against the fixture the reference wrote (tools/gen_golden.py: gen_network_e)."""
import os

import numpy as np
import torch
import yaml

from conftest import ROOT, load_golden
from oracle import detector as odet, net as onet


def test_eyolox_oracle_vs_reference():
    g = load_golden("network_eyolox_test")
    with open(os.path.join(ROOT, "configs", "model", "e-yolox", "e-yolox_test.yaml")) as f:
        cfg = yaml.safe_load(f)
    nc = int(g["num_classes"])
    state = {k[6:]: torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith("state/")}
    x, labels = torch.from_numpy(g["x"]), torch.from_numpy(g["labels"])
    with torch.no_grad():
        maps = odet.forward({k: v.clone() for k, v in state.items()}, cfg, nc, x, None, training=True)
    for i, m in enumerate(maps):
        np.testing.assert_allclose(m.numpy(), g["maps_train%d" % i], rtol=1e-4, atol=2e-5)
    st = {k: v.clone() for k, v in state.items()}
    out, grads = odet.train_step_grads(st, cfg, nc, x, labels)
    for k in ("loss", "loss_iou", "loss_obj", "loss_cls"):
        assert abs(float(out[k]) - float(g["out/" + k])) <= 2e-6 * max(1.0, abs(float(g["out/" + k]))), k
    gmax = max(float(np.abs(v).max()) for k, v in g.items() if k.startswith("grad/"))
    assert set(grads) == {k[5:] for k in g if k.startswith("grad/")}
    for k, v in g.items():
        if k.startswith("grad/"):
            err = float((grads[k[5:]] - torch.from_numpy(v)).abs().max())
            assert err <= 1e-4 * max(float(np.abs(v).max()), 1e-3 * gmax), (k, err)
    for k, v in g.items():
        if k.startswith("state_after/") and "running" in k:
            np.testing.assert_allclose(st[k[12:]].numpy(), v, rtol=1e-5, atol=1e-6, err_msg=k)
    with torch.no_grad():
        ev = odet.forward(st, cfg, nc, x, labels, training=False)
    np.testing.assert_allclose(ev.numpy(), g["eval_out"], rtol=1e-4, atol=1e-3)
