"""Pin oracle/yolox_loss.py against vectors produced by the imported reference
(tools/gen_golden.py -> tests/golden/loss_case_*.npz)."""
import numpy as np
import pytest
import torch

from oracle import yolox_loss as ol
from conftest import load_golden

CASES = ["A", "B", "C", "D", "E"]
L1_CASES = ["F", "G"]     # YOLOXLoss(use_l1=True) on the inputs of A and E


def _load(case):
    g = load_golden("loss_case_" + case)
    maps = [torch.from_numpy(g["map%d" % i]) for i in range(int(g["nmaps"]))]
    return g, maps, torch.from_numpy(g["labels"]), [int(s) for s in g["strides"]], int(g["num_classes"])


@pytest.mark.parametrize("case", CASES)
def test_assignment_bit_exact(case):
    g, maps, labels, strides, C = _load(case)
    out = ol.yolox_loss(maps, labels, strides, C, return_assign=True)
    a = out["_assign"]
    assert np.array_equal(a["fg"].numpy(), g["fg"])
    assert np.array_equal(a["matched_gt"].numpy(), g["matched_gt"])
    np.testing.assert_allclose(a["matched_iou"].numpy(), g["matched_iou"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("case", CASES)
def test_losses_and_grads(case):
    g, maps, labels, strides, C = _load(case)
    leafs = [m.clone().requires_grad_(True) for m in maps]
    out = ol.yolox_loss(leafs, labels, strides, C)
    for k in ("loss", "loss_iou", "loss_obj", "loss_cls"):
        assert abs(float(out[k].detach()) - float(g[k])) <= 1e-5 * max(1.0, abs(float(g[k]))), k
    assert out["loss_l1"] == 0.0
    assert abs(out["proportion"] - float(g["proportion"])) < 1e-7  # reference returns an fp32 tensor
    out["loss"].backward()
    for i, l in enumerate(leafs):
        np.testing.assert_allclose(l.grad.numpy(), g["grad%d" % i], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("case", L1_CASES)
def test_use_l1_losses_and_grads(case):
    """yolox_loss.py:128-135,157-160: same assignment, + L1 of the raw box outputs of the foreground anchors."""
    g, maps, labels, strides, C = _load(case)
    assert int(g["use_l1"]) == 1
    leafs = [m.clone().requires_grad_(True) for m in maps]
    out = ol.yolox_loss(leafs, labels, strides, C, return_assign=True, use_l1=True)
    assert np.array_equal(out["_assign"]["fg"].numpy(), g["fg"])
    assert np.array_equal(out["_assign"]["matched_gt"].numpy(), g["matched_gt"])
    for k in ("loss", "loss_iou", "loss_obj", "loss_cls", "loss_l1"):
        assert abs(float(out[k].detach()) - float(g[k])) <= 1e-5 * max(1.0, abs(float(g[k]))), k
    assert float(g["loss_l1"]) > 0.1
    out["loss"].backward()
    for i, l in enumerate(leafs):
        np.testing.assert_allclose(l.grad.numpy(), g["grad%d" % i], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("case", CASES)
def test_eval_decode(case):
    g, maps, labels, strides, C = _load(case)
    got = ol.eval_decode(maps, strides, C).numpy()
    np.testing.assert_allclose(got, g["eval_decode"], rtol=1e-6, atol=1e-6)


def test_fixtures_are_tie_free():
    # the reference's sort is unstable; fixtures are only valid pins if the
    # k-th boundary is not tied
    for case in CASES + L1_CASES:
        g = load_golden("loss_case_" + case)
        assert float(g["boundary_gap"]) > 1e-6, case


def test_nonsquare_grid_quirk():
    # yolox_loss.py:198-200: for h != w the grid is (a % h, a // h)
    grid = ol.make_grid(2, 3)[0]
    a = np.arange(6)
    assert np.array_equal(grid[:, 0].numpy(), (a % 2).astype(np.float32))
    assert np.array_equal(grid[:, 1].numpy(), (a // 2).astype(np.float32))
    sq = ol.make_grid(4, 4)[0]
    a = np.arange(16)
    assert np.array_equal(sq[:, 0].numpy(), (a % 4).astype(np.float32))
    assert np.array_equal(sq[:, 1].numpy(), (a // 4).astype(np.float32))


def test_take_all_branch_is_exercised():
    g, maps, labels, strides, C = _load("C")
    out = ol.yolox_loss(maps, labels, strides, C, return_assign=True)
    # 4 anchors, k >= N_c - 1 -> all four anchors selected in image 0
    assert int(out["_assign"]["fg"][0].sum()) == 4
