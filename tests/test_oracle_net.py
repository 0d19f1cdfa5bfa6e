"""Pin oracle/net.py + oracle/detector.py + oracle/harness.py against the
reference-generated fixtures."""
import numpy as np
import pytest
import torch
import yaml
import os

from oracle import net, detector, harness
from conftest import load_golden, warm_s_state, ROOT


def _cfg(name="yolox_test"):
    with open(os.path.join(ROOT, "configs", "model", "yolox", name + ".yaml")) as f:
        return yaml.safe_load(f)


def _state(g, prefix="state/"):
    return {k[len(prefix):]: torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith(prefix)}


BLOCKS = {
    "conv3s2": lambda s, x: net.conv_unit(s, "", x, 2, True),
    "conv3s1": lambda s, x: net.conv_unit(s, "", x, 1, True),
    "conv1": lambda s, x: net.conv_unit(s, "", x, 1, True),
    "focus": lambda s, x: net.conv_unit(s, "conv", net.focus(x), 1, True),
    "bottleneck": lambda s, x: net.bottleneck(s, "", x, True, True, "bn", "silu"),
    "csp": lambda s, x: net.csp_layer(s, "", x, 2, True, True, "bn", "silu"),
    "csp_noshort": lambda s, x: net.csp_layer(s, "", x, 1, False, True, "bn", "silu"),
    "spp": lambda s, x: net.spp_bottleneck(s, "", x, True, "bn", "silu"),
}


@pytest.mark.parametrize("tag", list(BLOCKS))
def test_block(tag):
    g = load_golden("blocks")
    pre = tag + "/state/"
    # keys inside a bare block have no leading module name: map "conv.weight" -> ".conv.weight"
    state = {"." + k[len(pre):] if tag not in ("focus",) else k[len(pre):]: torch.from_numpy(v.copy())
             for k, v in g.items() if k.startswith(pre)}
    if tag == "focus":
        state = {k: v for k, v in state.items()}
    names = [k for k in state if k.endswith("weight") or k.endswith("bias")]
    for k in names:
        state[k].requires_grad_(True)
    x = torch.from_numpy(g[tag + "/x"]).requires_grad_(True)
    y = BLOCKS[tag](state, x)
    np.testing.assert_allclose(y.detach().numpy(), g[tag + "/y"], rtol=1e-5, atol=1e-5)
    (y * torch.from_numpy(g[tag + "/r"])).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), g[tag + "/dx"], rtol=1e-4, atol=1e-4)
    gp = tag + "/grad/"
    checked = 0
    for k, v in g.items():
        if k.startswith(gp):
            key = k[len(gp):]
            key = key if tag == "focus" else "." + key
            np.testing.assert_allclose(state[key].grad.numpy(), v, rtol=1e-4, atol=2e-4, err_msg=k)
            checked += 1
    assert checked > 0
    sp = tag + "/state_after/"
    for k, v in g.items():
        if k.startswith(sp):
            key = k[len(sp):]
            key = key if tag == "focus" else "." + key
            np.testing.assert_allclose(state[key].detach().numpy(), v, rtol=1e-5, atol=1e-6, err_msg=k)


def test_network_train_step():
    g = load_golden("network_yolox_test")
    cfg = _cfg()
    C = int(g["num_classes"])
    state = _state(g)
    x = torch.from_numpy(g["x"])
    labels = torch.from_numpy(g["labels"])
    # raw maps (labels=None)
    s0 = {k: v.clone() for k, v in state.items()}
    maps = detector.forward(s0, cfg, C, x, None, training=True)
    for i, m in enumerate(maps):
        np.testing.assert_allclose(m.detach().numpy(), g["maps_train%d" % i], rtol=1e-4, atol=1e-4)
    out, grads = detector.train_step_grads(state, cfg, C, x, labels)
    for k in ("loss", "loss_iou", "loss_obj", "loss_cls"):
        assert abs(float(out[k].detach()) - float(g["out/" + k])) < 1e-5 * max(1, abs(float(g["out/" + k]))), k
    assert abs(out["proportion"] - float(g["out/proportion"])) < 1e-6
    nograd = set(str(n) for n in g["nograd_names"])
    assert all(".bn." in n for n in nograd)  # the dead Bottleneck BNs
    n = 0
    for k, v in g.items():
        if k.startswith("grad/"):
            name = k[5:]
            ref = v
            got = grads[name].numpy()
            scale = max(1e-3, float(np.abs(ref).max()))
            assert float(np.abs(got - ref).max()) <= 2e-4 * scale, (name, float(np.abs(got - ref).max()), scale)
            n += 1
    assert n > 100
    for k, v in g.items():
        if k.startswith("state_after/"):
            np.testing.assert_allclose(state[k[12:]].detach().numpy(), v, rtol=1e-5, atol=1e-6, err_msg=k)


def test_network_train_step_and_eval_on_a_rectangular_image():
    """64 x 96 input (head maps 8x12 / 4x6 / 2x3): the reference's non-square grid quirk (yolox_loss.py:198-200) inside a whole training
    step -- tests/golden/network_yolox_rect.npz = the reference itself on the weights of network_yolox_test.npz (tools/gen_golden.py rect)."""
    g = load_golden("network_yolox_rect")
    cfg = _cfg()
    C = int(g["num_classes"])
    state = _state(load_golden("network_yolox_test"))
    x, labels = torch.from_numpy(g["x"]), torch.from_numpy(g["labels"])
    assert x.shape[2] != x.shape[3]
    maps = detector.forward({k: v.clone() for k, v in state.items()}, cfg, C, x, None, training=True)
    for i, m in enumerate(maps):
        assert m.shape[2] != m.shape[3]
        np.testing.assert_allclose(m.detach().numpy(), g["maps_train%d" % i], rtol=1e-4, atol=1e-4)
    out, grads = detector.train_step_grads(state, cfg, C, x, labels)
    for k in ("loss", "loss_iou", "loss_obj", "loss_cls"):
        assert abs(float(out[k].detach()) - float(g["out/" + k])) < 1e-5 * max(1, abs(float(g["out/" + k]))), k
    assert abs(out["proportion"] - float(g["out/proportion"])) < 1e-6
    n = 0
    for k, v in g.items():
        if k.startswith("grad/"):
            got = grads[k[5:]].numpy()
            scale = max(1e-3, float(np.abs(v).max()))
            assert float(np.abs(got - v).max()) <= 2e-4 * scale, (k, float(np.abs(got - v).max()), scale)
            n += 1
    assert n > 100
    for k, v in g.items():
        if k.startswith("state_after/"):
            np.testing.assert_allclose(state[k[12:]].detach().numpy(), v, rtol=1e-5, atol=1e-6, err_msg=k)
    with torch.no_grad():
        ev = detector.forward(state, cfg, C, x, labels, training=False)
    np.testing.assert_allclose(ev.numpy(), g["eval_out"], rtol=1e-4, atol=1e-3)


def test_network_eval():
    g = load_golden("network_yolox_test")
    cfg = _cfg()
    C = int(g["num_classes"])
    state = _state(g)
    for k, v in g.items():
        if k.startswith("state_after/"):
            state[k[12:]] = torch.from_numpy(v.copy())
    x = torch.from_numpy(g["x"])
    with torch.no_grad():
        maps = detector.forward(state, cfg, C, x, None, training=False)
        for i, m in enumerate(maps):
            np.testing.assert_allclose(m.numpy(), g["maps_eval%d" % i], rtol=1e-4, atol=1e-4)
        out = detector.forward(state, cfg, C, x, torch.from_numpy(g["labels"]), training=False)
    np.testing.assert_allclose(out.numpy(), g["eval_out"], rtol=1e-4, atol=1e-3)


def test_build_state_matches_reference_layout():
    g = load_golden("network_yolox_test")
    cfg = _cfg()
    torch.manual_seed(96)
    st = net.build_state(cfg, int(g["num_classes"]))
    ref = _state(g)
    assert set(st) == set(ref)
    for k in st:
        assert tuple(st[k].shape) == tuple(ref[k].shape), k
    # same RNG consumption order as the reference's module tree -> identical conv weights
    for k in st:
        if k.endswith("conv.weight") or "_preds" in k:
            np.testing.assert_array_equal(st[k].numpy(), ref[k].numpy(), err_msg=k)


def test_lr_schedule():
    g = load_golden("lr_schedule")
    for i in range(3):
        warm, T = float(g["sched%d_warm" % i]), int(g["sched%d_T" % i])
        got = np.array([harness.lr_factor(t, warm, T) for t in range(T + 1)])
        np.testing.assert_allclose(got, g["sched%d_factor" % i], rtol=1e-12, atol=0)


def test_harness_trajectory():
    """3 x (fwd, bwd, SGD-momentum step, EMA update, LR step) on the tiny model."""
    g = load_golden("network_yolox_test")
    h = load_golden("harness_trajectory")
    cfg = _cfg()
    C = int(g["num_classes"])
    state = _state(g)
    ema = {k: v.clone() for k, v in state.items()}
    x = torch.from_numpy(h["x"])
    labels = torch.from_numpy(h["labels"])
    names = net.param_names(state)
    bufs, updates = {}, 0
    for step in range(3):
        lr = 0.01 * harness.lr_factor(step, 0.1 * 20, 20)
        assert abs(lr - float(h["lrs"][step])) < 1e-12
        out, grads = detector.train_step_grads(state, cfg, C, x, labels)
        assert abs(float(out["loss"].detach()) - float(h["loss%d" % step])) < 2e-4 * abs(float(h["loss%d" % step]))
        harness.sgd_step({k: state[k] for k in names}, grads, bufs, lr, 0.9)
        updates = harness.ema_update(ema, state, updates)
    for k in state:
        ref = h["final/" + k]
        got = state[k].detach().numpy()
        assert float(np.abs(got - ref).max()) <= 1e-4 * max(1.0, float(np.abs(ref).max())), k
        refe = h["ema/" + k]
        gote = ema[k].detach().numpy()
        assert float(np.abs(gote - refe).max()) <= 1e-4 * max(1.0, float(np.abs(refe).max())), k


def test_warm_yolox_s_step_vs_reference():
    """The benchmarked network (yolox_s.yaml, 80 classes) on the reference's warm weights (50 SGD steps, tools/gen_golden.py:
    gen_network_warm_s), 160x160 batch 2: the oracle's losses, head maps and gradients against the reference's recorded step."""
    g = load_golden("network_yolox_s_warm")
    cfg, C = _cfg("yolox_s"), int(g["num_classes"])
    state = warm_s_state(g)
    x, labels = torch.from_numpy(g["x"]), torch.from_numpy(g["labels"])
    with torch.no_grad():
        maps = detector.forward({k: v.clone() for k, v in state.items()}, cfg, C, x, None, training=True)
    for i, m in enumerate(maps):
        ref = g["maps_train%d" % i]
        assert float(np.abs(m.numpy() - ref).max()) <= 2e-4 * max(1.0, float(np.abs(ref).max()))
    out, grads = detector.train_step_grads({k: v.clone() for k, v in state.items()}, cfg, C, x, labels)
    for k in ("loss", "loss_iou", "loss_obj", "loss_cls"):
        got, want = float(out[k].detach()), float(g["out/" + k])
        assert abs(got - want) <= 1e-4 * max(1.0, abs(want)), (k, got, want)
    gmax = max(float(v) for k, v in g.items() if k.startswith("gnorm/"))
    n_full = 0
    for k, v in g.items():
        if k.startswith("grad/"):
            got = grads[k[5:]].numpy()
            assert float(np.abs(got - v).max()) <= 2e-4 * max(float(np.abs(v).max()), 1e-6 * gmax), k
            n_full += 1
        elif k.startswith("gnorm/"):
            got = float(grads[k[6:]].double().norm())
            assert abs(got - float(v)) <= 1e-3 * max(float(v), 1e-6 * gmax), k
    assert n_full >= 200
