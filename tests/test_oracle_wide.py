"""CPU: BASELINE.json configs[2..4] at their FULL width -- yolov7.yaml, yolox_l.yaml, yolox_x.yaml unchanged -- through the oracle on the
small batch the REFERENCE ran (tools/gen_golden.py: gen_wide; seed-96 weights, 128x128, batch 2): head maps, loss scalars, the stored
gradients and the L2 norm of EVERY parameter's gradient.  The GPU side is tests/test_gpu_configs.py::test_wide_*."""
import os

import numpy as np
import pytest
import torch
import yaml

from conftest import ROOT, load_golden
from oracle import net as onet, detector as odet


def _cfg(name):
    fam = "yolov7" if name.startswith("yolov7") else "yolox"
    with open(os.path.join(ROOT, "configs", "model", fam, name + ".yaml")) as f:
        return yaml.safe_load(f)


def _batch(g):
    return odet.synthetic_batch(int(g["batch"]), int(g["size"]), int(g["num_classes"]), num_gt=int(g["num_gt"]), max_gt=int(g["max_gt"]), seed=int(g["seed_data"]))


def _check_grads(g, grads, tol_full, tol_norm):
    for k in [k for k in g if k.startswith("grad/")]:
        ref = torch.from_numpy(g[k])
        assert float((grads[k[5:]] - ref).abs().max()) <= tol_full * max(float(ref.abs().max()), 1e-6), k
    names, norms = [str(n) for n in g["grad_names"]], g["grad_norms"]
    scale = float(np.sqrt(float(g["grad_sq_sum"])))
    n = 0
    for k, want in zip(names, norms):
        if want < 0:
            assert k not in grads or grads[k] is None or float(grads[k].abs().max()) == 0.0, k   # (a dead Bottleneck.bn parameter)
            continue
        got = float(grads[k].double().norm())
        assert abs(got - want) <= tol_norm * max(want, 1e-4 * scale), (k, got, want)
        n += 1
    assert n == int((norms >= 0).sum())


@pytest.mark.parametrize("name", ["yolox_l", "yolox_x"])
def test_oracle_wide_yolox_vs_reference(name):
    torch.set_num_threads(4)
    g = load_golden("wide_" + name)
    cfg, nc = _cfg(name), int(g["num_classes"])
    torch.manual_seed(int(g["seed_weights"]))
    state = onet.build_state(cfg, nc)
    assert np.array_equal(state["backbone.stem.conv.conv.weight"].numpy(), g["first_weight"])
    imgs, labels = _batch(g)
    with torch.no_grad():
        maps = odet.forward({k: v.clone() for k, v in state.items()}, cfg, nc, imgs, None, training=True)
    for i, m in enumerate(maps):
        ref = torch.from_numpy(g["maps/%d" % i])
        assert float((m - ref).abs().max()) <= 2e-4 * float(ref.abs().max()), i
    out, grads = odet.train_step_grads(state, cfg, nc, imgs, labels)
    for k in ("loss", "loss_iou", "loss_obj", "loss_cls"):
        got, want = float(out[k]), float(g["out/" + k][0])
        assert abs(got - want) <= 1e-5 * max(1.0, abs(want)), (k, got, want)
    _check_grads(g, grads, 2e-4, 1e-3)


def test_oracle_wide_yolov7_vs_reference():
    torch.set_num_threads(4)
    from oracle import net_v7, yolov7_loss as ol
    import pl_yolo_amd
    g = load_golden("wide_yolov7")
    cfg, nc = _cfg("yolov7"), int(g["num_classes"])
    torch.manual_seed(int(g["seed_weights"]))
    model = pl_yolo_amd.build_model(cfg, nc)      # host-side module tree only: the reference's constructor order and initialisers
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    first = next(iter(model.named_parameters()))[0]
    assert np.array_equal(state[first].numpy(), g["first_weight"])
    imgs, labels = _batch(g)
    with torch.no_grad():
        maps = net_v7.yolov7_network({k: v.clone() for k, v in state.items()}, cfg, imgs, True)
    for i, m in enumerate(maps):
        ref = torch.from_numpy(g["maps/%d" % i])
        assert float((m - ref).abs().max()) <= 2e-4 * float(ref.abs().max()), i
    names = onet.param_names(state)
    for k in names:
        state[k].requires_grad_(True)
    maps = net_v7.yolov7_network(state, cfg, imgs, True)
    out = ol.yolov7_loss(maps, labels, cfg["loss"]["stride"], cfg["loss"]["anchors"], nc)
    out["loss"].sum().backward()
    got, want = float(out["loss"].sum()), float(g["out/loss"][0])
    assert abs(got - want) <= 1e-5 * max(1.0, abs(want)), (got, want)
    grads = {k: state[k].grad for k in names if state[k].grad is not None}
    _check_grads(g, grads, 3e-4, 1e-3)
