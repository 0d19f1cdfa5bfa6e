"""Pin oracle/net_v7.py (YOLOv7 family) against the reference-generated fixture."""
import os

import numpy as np
import pytest
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


# ---- YOLOv7 training loss (row a24) ---------------------------------------------------------
V7LOSS_TIED = {"v7loss_case_B"}   # has duplicate candidate cells that are tied at a top-k boundary
V7LOSS_CASES = ["v7loss_case_A", "v7loss_case_B", "v7loss_case_C", "v7loss_case_D", "v7loss_case_E"]


def _v7loss_run(g):
    from oracle import yolov7_loss as ol
    maps = [torch.from_numpy(g["map%d" % i].copy()).requires_grad_(True) for i in range(3)]
    out = ol.yolov7_loss(maps, torch.from_numpy(g["labels"]), [int(s) for s in g["strides"]], g["anchors"].tolist(), int(g["num_classes"]))
    out["loss"].backward()
    return maps, out


@pytest.mark.parametrize("case", V7LOSS_CASES)
def test_v7_loss_oracle_vs_reference(case):
    """find_3_positive + per-image SimOTA + CIoU/obj/cls losses vs the reference's own run:
    matched index lists bit-exact (same order), loss and d loss/d maps to fp32 round-off."""
    g = load_golden(case)
    maps, out = _v7loss_run(g)
    for l, mt in enumerate(out["matched"]):
        mine = np.concatenate([np.stack([mt[k].numpy() for k in ("b", "a", "gj", "gi")], 1).astype(np.float64), mt["t"].numpy()[:, 1:]], 1)
        ref = np.concatenate([np.stack([g["m%d_%s" % (l, k)] for k in ("b", "a", "gj", "gi")], 1).astype(np.float64),
                              g["m%d_t" % l].reshape(-1, 6)[:, 1:]], 1)
        assert mine.shape == ref.shape, (case, l)
        if case in V7LOSS_TIED:
            # two GTs sharing a cell give bit-identical candidate columns; which copy torch.topk keeps
            # is std::nth_element's business, so only the matched (cell, GT) multiset is defined
            mine, ref = mine[np.lexsort(mine.T[::-1])], ref[np.lexsort(ref.T[::-1])]
        assert np.array_equal(mine, ref), (case, l)
        np.testing.assert_allclose(np.sort(mt["anch"].numpy(), 0), np.sort(g["m%d_anch" % l].reshape(-1, 2), 0), rtol=1e-6)
    assert abs(float(out["loss"]) - float(g["loss"][0])) <= 2e-6 * max(1.0, abs(float(g["loss"][0])))
    for i, m in enumerate(maps):
        ref = g["dmap%d" % i]
        assert np.abs(m.grad.numpy() - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max()) + 1e-9, (case, i)


def test_v7_network_training_loss_and_grads():
    """backbone+neck+head+loss end to end vs the reference's `out/loss` and parameter gradients."""
    from oracle import yolov7_loss as ol
    g = load_golden("network_yolov7_test")
    cfg = _cfg()
    state = {k[6:]: torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith("state/")}
    names = net.param_names(state)
    for k in names:
        state[k].requires_grad_(True)
    # the fixture took the loss on the model's SECOND train-mode forward: running statistics differ,
    # batch statistics (and therefore maps, loss, gradients) do not
    maps = net_v7.yolov7_network(state, cfg, torch.from_numpy(g["x"]), True)
    out = ol.yolov7_loss(maps, torch.from_numpy(g["labels"]), cfg["loss"]["stride"], cfg["loss"]["anchors"], int(g["num_classes"]))
    assert abs(float(out["loss"].detach()) - float(g["out/loss"][0])) <= 1e-5
    out["loss"].backward()
    for k in names:
        if "lossgrad/" + k not in g:
            continue
        ref = g["lossgrad/" + k]
        assert float(np.abs(state[k].grad.numpy() - ref).max()) <= 5e-4 * max(1e-3, float(np.abs(ref).max())), k


@pytest.mark.parametrize("tag", ["ne", "id"])
def test_repconv_oracle_vs_reference(tag):
    """RepConv train-time form (row a13) vs the reference class: output, input gradient, parameter gradients."""
    g = load_golden("repconv_blocks")
    state = {k[len(tag) + 7:]: torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith(tag + "/state/")}
    state = {"m." + k: v for k, v in state.items()}
    names = [k for k in state if k.endswith(".weight") or k.endswith(".bias")]
    for k in names:
        state[k].requires_grad_(True)
    x = torch.from_numpy(g[tag + "/x"]).requires_grad_(True)
    y = net_v7.repconv(state, "m", x, True)
    np.testing.assert_allclose(y.detach().numpy(), g[tag + "/y"], rtol=1e-5, atol=1e-5)
    (y * torch.from_numpy(g[tag + "/r"])).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), g[tag + "/dx"], rtol=1e-4, atol=1e-5)
    for k in names:
        np.testing.assert_allclose(state[k].grad.numpy(), g["%s/grad/%s" % (tag, k[2:])], rtol=2e-4, atol=2e-5, err_msg=k)
