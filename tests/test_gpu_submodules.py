"""-m gpu: the reference's TENSOR contract of the sub-modules (pl_yolo_amd/module_runner.py): `BaseConv(x)`, `CSPLayer(x)`,
`model.backbone(x) -> list`, `model.neck(list) -> list`, `model.head(list) -> list of raw NCHW maps`, each differentiable in training
mode (reference models/layers/network_blocks.py, models/backbones/darknet_csp.py:61-75, models/necks/pafpn_csp.py:60-86,
models/heads/decoupled_head.py:77-95).

  * every block of tests/golden/blocks.npz (the REFERENCE's own modules run on seeded inputs: output, input gradient, every parameter
    gradient, running statistics) through `module(x)` + `backward` in the fp32 parity mode, and the output in bf16;
  * backbone -> neck -> head called one after the other == the raw maps of the whole detector (same launches, another trace), and the
    gradients of a functional of those maps == the detector's (labels=None path);
  * CPU tensors and malformed inputs keep refusing loudly (the loss plugins' own call: tests/test_gpu_loss_plugins.py)."""
import os

import numpy as np
import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu

import pl_yolo_amd  # noqa: E402
from pl_yolo_amd.layers import BaseConv, Bottleneck, CSPLayer, Focus, SPPBottleneck  # noqa: E402
from conftest import ROOT, load_golden  # noqa: E402
import hiputil as hu  # noqa: E402

BLOCKS = {
    "conv3s2": lambda: BaseConv(8, 16, 3, 2),
    "conv3s1": lambda: BaseConv(8, 16, 3, 1),
    "conv1": lambda: BaseConv(16, 8, 1, 1),
    "focus": lambda: Focus(3, 8, ksize=3),
    "bottleneck": lambda: Bottleneck(8, 8, True, 1.0),
    "csp": lambda: CSPLayer(16, 16, num_bottle=2),
    "csp_noshort": lambda: CSPLayer(32, 16, num_bottle=1, shortcut=False),
    "spp": lambda: SPPBottleneck(32, 32),
}


@pytest.mark.parametrize("tag", sorted(BLOCKS))
def test_block_call_matches_reference_fixture(tag):
    g = load_golden("blocks")
    sd = {k[len(tag) + 7:]: torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith(tag + "/state/")}
    m = BLOCKS[tag]()
    m.load_state_dict(sd)
    m.compute_dtype = "fp32"
    m = m.to(hu.DEV).train()
    x = torch.from_numpy(g[tag + "/x"]).to(hu.DEV)
    is_image = tag == "focus"
    xin = x.clone().requires_grad_(not is_image)
    y = m(xin)
    want = torch.from_numpy(g[tag + "/y"]).to(hu.DEV)
    assert tuple(y.shape) == tuple(want.shape)
    err = hu.relerr(y.detach(), want)
    r = torch.from_numpy(g[tag + "/r"]).to(hu.DEV)
    (y * r).sum().backward()
    torch.cuda.synchronize()
    print("block %-12s fp32 output relerr %.3g" % (tag, err))
    assert err <= 1e-4
    if not is_image:
        dx = torch.from_numpy(g[tag + "/dx"]).to(hu.DEV)
        e = hu.relerr(xin.grad, dx)
        print("block %-12s input gradient relerr %.3g" % (tag, e))
        assert e <= 2e-4
    n = 0
    for name, p in m.named_parameters():
        k = "%s/grad/%s" % (tag, name)
        if k not in g:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name     # (the reference's unused Bottleneck.bn)
            continue
        e = hu.relerr(p.grad, torch.from_numpy(g[k]).to(hu.DEV))
        assert e <= 3e-4, (name, e)
        n += 1
    assert n > 0
    for k, v in g.items():
        if k.startswith(tag + "/state_after/"):
            got = m.state_dict()[k[len(tag) + 13:]]
            np.testing.assert_allclose(got.cpu().numpy(), v, rtol=1e-5, atol=1e-6, err_msg=k)
    # a second backward ACCUMULATES into .grad like autograd does
    g1 = {n_: p.grad.clone() for n_, p in m.named_parameters() if p.grad is not None}
    y2 = m(xin)
    (y2 * r).sum().backward()
    torch.cuda.synchronize()
    for n_, p in m.named_parameters():
        if p.grad is not None and n_ in g1 and "running" not in n_:
            assert hu.relerr(p.grad, 2.0 * g1[n_]) <= 2e-2, n_      # (the second forward ran on moved running statistics only; batch statistics are the same)
    # bf16 path, eval mode (no gradient): the activated output within the bf16 band
    m16 = BLOCKS[tag]()
    m16.load_state_dict(sd)
    m16 = m16.to(hu.DEV).train()
    with torch.no_grad():
        y16 = m16(x)
    e16 = hu.relerr(y16, want)
    print("block %-12s bf16 output relerr %.3g" % (tag, e16))
    assert e16 <= 3e-2


def _cfg(name, family="yolox"):
    with open(os.path.join(ROOT, "configs", "model", family, name + ".yaml")) as f:
        return yaml.safe_load(f)


@pytest.mark.parametrize("family", ["yolox", "yolov7"])
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_backbone_neck_head_called_one_by_one_match_the_detector(dtype, family):
    """CSPDarkNet + CSPPAFPN + DecoupledHead, and EELAN + YOLOv7NECK + ImplicitHead (models/backbones/eelan.py, models/necks/yolov7_neck.py,
    models/heads/implicit_head.py)."""
    g = load_golden("network_%s_test" % family)
    nc = int(g["num_classes"])
    sd = {k[6:]: torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith("state/")}
    x = torch.from_numpy(g["x"]).to(hu.DEV)

    def build():
        m = pl_yolo_amd.build_model(dict(_cfg(family + "_test", family), compute_dtype=dtype), nc)
        m.load_state_dict(sd)
        return m.to(hu.DEV).train()
    whole = build()
    maps_w = whole(x)                     # labels=None: raw NCHW head maps, differentiable
    gen = torch.Generator().manual_seed(5)
    cot = [torch.randn(mp.shape, generator=gen).to(hu.DEV) for mp in maps_w]
    sum((mp * c).sum() for mp, c in zip(maps_w, cot)).backward()
    torch.cuda.synchronize()
    gw = {n: p.grad.clone() for n, p in whole.named_parameters() if p.grad is not None}
    parts = build()
    for sub in (parts.backbone, parts.neck, parts.head):
        sub.compute_dtype = dtype
    feats = parts.backbone(x)
    assert isinstance(feats, list) and len(feats) == 3 and all(f.dim() == 4 for f in feats)
    necked = parts.neck(feats)
    assert isinstance(necked, list) and len(necked) == 3
    maps_p = parts.head(necked)
    assert [tuple(a.shape) for a in maps_p] == [tuple(b.shape) for b in maps_w]
    tol = 2e-5 if dtype == "fp32" else 3e-2       # bf16: the features cross the module boundaries as fp32 tensors and are re-rounded
    for i, (a, b) in enumerate(zip(maps_p, maps_w)):
        e = hu.relerr(a.detach(), b.detach())
        print("sub-modules one by one, %s: map %d relerr %.3g" % (dtype, i, e))
        assert e <= tol
    sum((mp * c).sum() for mp, c in zip(maps_p, cot)).backward()
    torch.cuda.synchronize()
    worst = 0.0
    for n, p in parts.named_parameters():
        if n in gw:
            assert p.grad is not None, n
            u, v = p.grad.double().reshape(-1), gw[n].double().reshape(-1)
            c = float((u * v).sum() / (u.norm() * v.norm() + 1e-300))
            worst = max(worst, 1.0 - c)
            assert c >= (0.99999 if dtype == "fp32" else 0.98), (n, c)
    print("sub-modules one by one, %s: worst 1 - cosine of a parameter gradient against the detector's %.3g" % (dtype, worst))
    # eval mode, no gradient: plain tensors out
    parts.eval()
    with torch.no_grad():
        f2 = parts.backbone(x)
    assert all(not t.requires_grad for t in f2)
    # the detector adopts the parameters into its flat buffers (their storage moves): the sub-module's traced sessions notice and re-trace
    parts.train()
    before = [t.detach().clone() for t in parts.backbone(x)]
    parts(x)
    after = parts.backbone(x)
    for a, b in zip(after, before):
        assert hu.relerr(a.detach(), b) <= (1e-6 if dtype == "fp32" else 2e-2)


def test_submodule_refusals():
    from pl_yolo_amd._lib import PlyoloError
    m = BaseConv(8, 16, 3, 1)
    with pytest.raises(PlyoloError, match="MI355X"):
        m(torch.zeros(1, 8, 8, 8))                       # CPU tensor
    m = m.to(hu.DEV)
    with pytest.raises(PlyoloError, match="multiple of"):
        m.conv = torch.nn.Conv2d(6, 16, 3, 1, 1, bias=False).to(hu.DEV)
        m(torch.zeros(1, 6, 8, 8, device=hu.DEV))        # 6 channels: no whole channel vector
    model = pl_yolo_amd.build_model(_cfg("yolox_test"), 3).to(hu.DEV)
    with pytest.raises(PlyoloError, match="strides"):       # one map for a three-level loss
        model.loss([torch.zeros(1, 8, 4, 4, device=hu.DEV)], torch.zeros(1, 1, 5, device=hu.DEV))
