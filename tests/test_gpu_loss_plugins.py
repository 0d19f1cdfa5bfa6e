"""-m gpu: the loss plugins as CALLABLES -- the reference's own contract `loss(list of raw NCHW head maps, labels)`
(models/losses/yolox/yolox_loss.py:20-36, models/losses/yolov7/yolov7_loss.py:38-78,150-153), which is how the reference's
OneStageD.forward itself is written: `self.loss(self.head(self.neck(self.backbone(x))), labels)` (PL_Modules/build_detection.py:46-53).

  * every reference-generated loss fixture (YOLOX A-G, YOLOv7 A-E) through `plugin(maps, labels)` + `backward`: losses 1e-4,
    gradients of the maps 1e-5 / 2e-5 of the largest entry, eval decode 2e-6 -- the bars of the kernel-level tests; and the reference's
    side effects on its arguments (decoded boxes left in the caller's YOLOX maps, bit for bit against tests/golden/loss_side_effects.npz;
    the caller's YOLOv7 list re-pointed at the [B, na, h, w, ch] views);
  * `model.loss(model.head(model.neck(model.backbone(x))), labels)` == `model(x, labels)`: losses and every parameter gradient, train
    and eval, both plugin families, fp32 parity mode and bf16;
  * a backward whose session saw another forward in between recomputes (autograd semantics, no stale assignments); refusals."""
import os

import numpy as np
import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu

import pl_yolo_amd  # noqa: E402
from pl_yolo_amd._lib import PlyoloError  # noqa: E402
from pl_yolo_amd.losses import YOLOXLoss, YOLOv7Loss  # noqa: E402
from conftest import ROOT, load_golden  # noqa: E402
import hiputil as hu  # noqa: E402


def _cfg(name, family="yolox"):
    with open(os.path.join(ROOT, "configs", "model", family, name + ".yaml")) as f:
        return yaml.safe_load(f)


def _leafs(g, key, n):
    return [torch.from_numpy(g["%s%d" % (key, i)]).to(hu.DEV).requires_grad_(True) for i in range(n)]


def _nonleaf(leafs):
    """What a head hands the loss: non-leaf tensors (YOLOXLoss decodes their box channels in place, like the reference; a leaf that
    requires a gradient is refused with torch's own message)."""
    return [l * 1.0 for l in leafs]


@pytest.mark.parametrize("case", ["A", "B", "C", "D", "E", "F", "G"])
def test_yolox_loss_plugin_call_vs_reference_fixture(case):
    g = load_golden("loss_case_" + case)
    use_l1 = case in ("F", "G")
    plugin = YOLOXLoss(int(g["num_classes"]), [int(s) for s in g["strides"]], use_l1=use_l1).train()
    leafs = _leafs(g, "map", int(g["nmaps"]))
    maps = _nonleaf(leafs)
    before = [m.detach().clone() for m in maps]
    labels = torch.from_numpy(g["labels"]).to(hu.DEV)
    out = plugin(maps, labels)
    assert set(out) == {"loss", "loss_iou", "loss_obj", "loss_cls", "loss_l1", "proportion"}
    for k in ("loss", "loss_iou", "loss_obj", "loss_cls") + (("loss_l1",) if use_l1 else ()):
        got, want = float(out[k].detach()), float(g[k])
        assert abs(got - want) <= 1e-4 * max(1.0, abs(want)), (k, got, want)
    if not use_l1:
        assert out["loss_l1"] == 0.0 and not torch.is_tensor(out["loss_l1"])      # the python float of yolox_loss.py:159-160
    assert abs(float(out["proportion"]) - float(g["proportion"])) < 1e-5 and not out["proportion"].requires_grad
    out["loss"].backward()
    torch.cuda.synchronize()
    side = load_golden("loss_side_effects")
    for i, (l, m) in enumerate(zip(leafs, maps)):
        ref = g["grad%d" % i]
        assert float(np.abs(l.grad.cpu().numpy() - ref).max()) <= 1e-5 * max(1e-3, float(np.abs(ref).max())), i
        # the reference's decode writes THROUGH a view into channels 0..3 of the maps it is handed (yolox_loss.py:204-219): after the call
        # they hold (cx, cy, w, h) in pixels, the other channels are untouched
        assert torch.equal(m.detach()[:, 4:], before[i][:, 4:]) and not torch.equal(m.detach()[:, :4], before[i][:, :4])
        if case in ("A", "D"):     # the fixture is the reference run on the CPU: centres exact, extents to the ulp of the device's exp
            want = side["%s/boxes_after%d" % (case, i)]
            assert np.array_equal(m.detach()[:, :2].cpu().numpy(), want[:, :2]), i
            np.testing.assert_allclose(m.detach()[:, 2:4].cpu().numpy(), want[:, 2:4], rtol=1e-6, atol=0)
    raw = [b.clone() for b in before]
    ev = plugin.eval()(raw, labels)
    assert tuple(ev.shape) == tuple(g["eval_decode"].shape) and not ev.requires_grad
    np.testing.assert_allclose(ev.cpu().numpy(), g["eval_decode"], rtol=2e-6, atol=2e-5)
    for m, r in zip(maps, raw):                               # eval decodes the caller's boxes in place too
        assert torch.equal(m.detach(), r)
    with pytest.raises(PlyoloError, match="leaf Variable"):   # torch's refusal of the reference's in-place decode on such a tensor
        plugin.train()(leafs, labels)


def test_yolox_loss_plugin_weighted_terms_and_stale_session():
    """Upstream gradients of the single terms (gout of the loss vector), and a backward after ANOTHER forward of the same shapes: the
    node re-runs its own forward instead of back-propagating through the other call's assignments.  Second call: the batch of case A
    rolled by one image (same shapes, other assignments per slot; its gradients are the golden ones rolled the same way)."""
    ga = load_golden("loss_case_A")
    plugin = YOLOXLoss(int(ga["num_classes"]), [int(s) for s in ga["strides"]]).train()
    la = torch.from_numpy(ga["labels"]).to(hu.DEV)
    ma = _leafs(ga, "map", 3)
    mb = [torch.roll(m.detach(), 1, 0).clone().requires_grad_(True) for m in ma]
    lb = torch.roll(la, 1, 0).contiguous()
    oa = plugin(_nonleaf(ma), la)
    ob = plugin(_nonleaf(mb), lb)               # same buffers, other assignments
    # the loss normalises by the batch's foreground count and sums over images: rolling the batch changes nothing but the order
    assert abs(float(oa["loss"].detach()) - float(ob["loss"].detach())) <= 1e-5 * float(oa["loss"].detach())
    oa["loss"].backward()                       # stale: recomputed from the raw values set aside
    ob["loss"].backward()                       # stale again (a's recompute ran in between)
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(ma, mb)):
        ref = ga["grad%d" % i]
        tol = 1e-5 * max(1e-3, float(np.abs(ref).max()))
        assert float(np.abs(a.grad.cpu().numpy() - ref).max()) <= tol, i
        assert float(np.abs(b.grad.cpu().numpy() - np.roll(ref, 1, 0)).max()) <= tol, i
    # loss = 5 iou + obj + cls (yolox_loss.py:150-158): the three terms back-propagated through their own slots of the loss vector add up
    # to the gradient of `loss`
    m2 = _leafs(ga, "map", 3)
    o2 = plugin(_nonleaf(m2), la)
    (5.0 * o2["loss_iou"] + o2["loss_obj"] + o2["loss_cls"]).backward()
    torch.cuda.synchronize()
    for a, b in zip(ma, m2):
        assert float((a.grad - b.grad).abs().max()) <= 2e-6 * max(1e-3, float(a.grad.abs().max()))


@pytest.mark.parametrize("case", ["v7loss_case_A", "v7loss_case_B", "v7loss_case_C", "v7loss_case_D", "v7loss_case_E"])
def test_v7_loss_plugin_call_vs_reference_fixture(case):
    g = load_golden(case)
    nc = int(g["num_classes"])
    plugin = YOLOv7Loss(nc, g["strides"].tolist(), g["anchors"].tolist()).train()
    maps = _leafs(g, "map", 3)
    labels = torch.from_numpy(g["labels"]).to(hu.DEV)
    inputs = list(maps)
    out = plugin(inputs, labels)
    assert set(out) == {"loss"} and tuple(out["loss"].shape) == (1,)                  # yolov7_loss.py:150-153
    want = float(g["loss"][0])
    assert abs(float(out["loss"].detach()) - want) <= 1e-4 * max(1.0, abs(want))
    # the reference replaces the entries of the caller's LIST with its [B, na, h, w, ch] views (yolov7_loss.py:43-47)
    for m, v in zip(maps, inputs):
        B, _, h, w = m.shape
        assert tuple(v.shape) == (B, 3, h, w, 5 + nc)
        assert torch.equal(v.detach(), m.detach().view(B, 3, 5 + nc, h, w).permute(0, 1, 3, 4, 2))
    out["loss"].sum().backward()
    torch.cuda.synchronize()
    for i, m in enumerate(maps):
        ref = g["dmap%d" % i]
        err = float(np.abs(m.grad.cpu().numpy() - ref).max()) / max(1e-12, float(np.abs(ref).max()))
        assert err <= 2e-5, (i, err)


def _built(family, dtype, g):
    nc = int(g["num_classes"])
    sd = {k[6:]: torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith("state/")}
    m = pl_yolo_amd.build_model(dict(_cfg(family + "_test", family), compute_dtype=dtype), nc)
    m.load_state_dict(sd)
    for sub in (m.backbone, m.neck, m.head):
        sub.compute_dtype = dtype
    return m.to(hu.DEV)


@pytest.mark.parametrize("family", ["yolox", "yolov7"])
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_loss_of_head_of_neck_of_backbone_equals_the_detector(dtype, family):
    """The reference's OneStageD.forward, written out by the caller: model.loss(model.head(model.neck(model.backbone(x))), labels)."""
    g = load_golden("network_%s_test" % family)
    x = torch.from_numpy(g["x"]).to(hu.DEV)
    labels = torch.from_numpy(g["labels"]).to(hu.DEV)
    whole = _built(family, dtype, g).train()
    ow = whole(x, labels)
    ow["loss"].sum().backward()
    torch.cuda.synchronize()
    gw = {n: p.grad.clone() for n, p in whole.named_parameters() if p.grad is not None}
    parts = _built(family, dtype, g).train()
    op = parts.loss(parts.head(parts.neck(parts.backbone(x))), labels)
    assert set(op) == set(ow)
    # fp32: the same kernels on the same numbers (the maps cross the module boundaries as fp32 tensors) -- equal to rounding of the
    # atomically accumulated parity-mode weight gradients; bf16: the features are re-rounded at every boundary
    ltol = 1e-6 if dtype == "fp32" else 2e-2
    for k in ow:
        a, b = ow[k], op[k]
        if torch.is_tensor(a):
            assert tuple(a.shape) == tuple(b.shape)
            d = float((a.detach() - b.detach()).abs().max())
            print("%s %s: %s detector %.7f pieces %.7f" % (family, dtype, k, float(a.detach().sum()), float(b.detach().sum())))
            assert d <= ltol * max(1.0, float(a.detach().abs().max())), (k, d)
        else:
            assert a == b
    op["loss"].sum().backward()
    torch.cuda.synchronize()
    worst = 0.0
    for n, p in parts.named_parameters():
        if n in gw:
            assert p.grad is not None, n
            u, v = p.grad.double().reshape(-1), gw[n].double().reshape(-1)
            if float(v.norm()) == 0.0:
                assert float(u.norm()) == 0.0, n
                continue
            c = float((u * v).sum() / (u.norm() * v.norm()))
            worst = max(worst, 1.0 - c)
            assert c >= (1.0 - 1e-9 if dtype == "fp32" else 0.98), (n, c)
    print("%s %s: worst 1 - cosine of a parameter gradient, pieces against the detector: %.3g" % (family, dtype, worst))
    # eval: the decoded [B, A, 5+C] tensor of the detector
    whole.eval()
    parts.eval()
    with torch.no_grad():
        ew = whole(x, labels)
        ep = parts.loss(parts.head(parts.neck(parts.backbone(x))), labels)
    assert tuple(ew.shape) == tuple(ep.shape)
    e = hu.relerr(ep, ew)
    print("%s %s: eval decode, pieces against the detector: relerr %.3g" % (family, dtype, e))
    assert e <= (1e-6 if dtype == "fp32" else 3e-2)


def test_loss_plugin_refusals():
    plugin = YOLOXLoss(3, [8, 16, 32]).train()
    maps = [torch.zeros(1, 8, s, s) for s in (8, 4, 2)]
    with pytest.raises(PlyoloError, match="MI355X"):
        plugin(maps, torch.zeros(1, 2, 5))                                         # CPU tensors
    dm = [m.to(hu.DEV) for m in maps]
    with pytest.raises(PlyoloError, match="strides"):
        plugin(dm[:1], torch.zeros(1, 2, 5, device=hu.DEV))                        # one map for three strides
    with pytest.raises(PlyoloError, match="channels"):
        plugin([torch.zeros(1, 9, 8, 8, device=hu.DEV)] + dm[1:], torch.zeros(1, 2, 5, device=hu.DEV))
    with pytest.raises(PlyoloError, match="labels"):
        plugin(dm, torch.zeros(2, 2, 5, device=hu.DEV))                            # batch mismatch
    with pytest.raises(PlyoloError, match="labels"):
        plugin(dm, None)                                                           # training needs labels
    # an image without any label row trains (no foreground): finite loss, gradient of the objectness only
    leafs = [m.clone().requires_grad_(True) for m in dm]
    out = plugin(_nonleaf(leafs), torch.zeros(1, 0, 5, device=hu.DEV))
    out["loss"].backward()
    torch.cuda.synchronize()
    assert np.isfinite(float(out["loss"].detach())) and all(torch.isfinite(m.grad).all() for m in leafs)
    import copy
    c = copy.deepcopy(plugin)                                                      # ModelEMA deep-copies the detector: buffers stay behind
    assert len(c.__dict__["_edge"]) == 0 and len(plugin.__dict__["_edge"]) > 0


def test_eval_mode_submodule_does_not_cut_gradients_silently():
    """a sub-module in eval mode has no backward plan.  An input that requires a gradient raises; parameters that do are
    warned about once; under no_grad nothing is said."""
    from pl_yolo_amd.layers import BaseConv
    m = BaseConv(8, 16, 3, 1).to(hu.DEV).eval()
    x = torch.randn(1, 8, 8, 8, device=hu.DEV)
    with pytest.raises(PlyoloError, match="eval mode"):
        m(x.clone().requires_grad_(True))
    with pytest.warns(UserWarning, match="eval mode"):
        y = m(x)
    assert not y.requires_grad
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        m(x)                               # warned once per module
        with torch.no_grad():
            m(x.clone().requires_grad_(True))


def test_submodule_session_follows_requires_grad():
    """the recorded backward holds the gradient address of every parameter that required one at trace time -- freezing or
    unfreezing a parameter afterwards re-traces."""
    from pl_yolo_amd.layers import BaseConv
    torch.manual_seed(3)
    m = BaseConv(8, 16, 3, 1)
    m.compute_dtype = "fp32"
    m = m.to(hu.DEV).train()
    x = torch.randn(2, 8, 8, 8, device=hu.DEV)
    m(x).square().sum().backward()
    torch.cuda.synchronize()
    full = {n: p.grad.clone() for n, p in m.named_parameters()}
    assert all(v is not None for v in full.values())
    m.zero_grad(set_to_none=True)
    m.conv.weight.requires_grad_(False)
    m(x).square().sum().backward()
    torch.cuda.synchronize()
    assert m.conv.weight.grad is None
    for n, p in m.named_parameters():
        if p.requires_grad:
            assert hu.relerr(p.grad, full[n]) <= 1e-5, n
    m.zero_grad(set_to_none=True)
    m.conv.weight.requires_grad_(True)
    m(x).square().sum().backward()
    torch.cuda.synchronize()
    assert m.conv.weight.grad is not None and hu.relerr(m.conv.weight.grad, full["conv.weight"]) <= 1e-5
