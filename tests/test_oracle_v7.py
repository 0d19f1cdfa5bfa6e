"""Pin oracle/net_v7.py (YOLOv7 family) against the reference-generated fixture."""
import os

import numpy as np
import torch
import yaml

from oracle import net, net_v7
from conftest import load_golden, ROOT


def _cfg():
    with open(os.path.join(ROOT, "configs", "model", "yolov7", "yolov7_test.yaml")) as f:
        return yaml.safe_load(f)


def test_v7_network_maps_grads_eval():
    g = load_golden("network_yolov7_test")
    cfg = _cfg()
    C = int(g["num_classes"])
    state = {k[6:]: torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith("state/")}
    names = net.param_names(state)
    for k in names:
        state[k].requires_grad_(True)
    x = torch.from_numpy(g["x"])
    maps = net_v7.yolov7_network(state, cfg, x, True)
    for i, m in enumerate(maps):
        np.testing.assert_allclose(m.detach().numpy(), g["maps_train%d" % i], rtol=1e-4, atol=1e-4)
    sum((m * torch.from_numpy(g["r%d" % i])).sum() for i, m in enumerate(maps)).backward()
    n = 0
    for k in names:
        ref = g["grad/" + k]
        got = state[k].grad.numpy()
        assert float(np.abs(got - ref).max()) <= 3e-4 * max(1e-3, float(np.abs(ref).max())), k
        n += 1
    assert n == 249
    for k, v in g.items():
        if k.startswith("state_after/") and "running" in k:
            np.testing.assert_allclose(state[k[12:]].detach().numpy(), v, rtol=1e-5, atol=1e-6, err_msg=k)
    # eval branch: the reference evaluated after TWO train-mode forwards (maps + loss)
    with torch.no_grad():
        net_v7.yolov7_network(state, cfg, x, True)
        ev = net_v7.eval_decode(net_v7.yolov7_network(state, cfg, x, False), cfg["loss"]["stride"], cfg["loss"]["anchors"], C)
    np.testing.assert_allclose(ev.numpy(), g["eval_out"], rtol=1e-3, atol=2e-3)
