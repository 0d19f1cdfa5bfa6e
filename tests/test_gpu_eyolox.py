"""-m gpu: the e-yolox plugin family: depthwise 3x3 and bicubic kernels through the C ABI against plain
PyTorch fp32, and the whole ecmnet + al_pafpn + decoupled_head + yolox detector against the reference-generated fixture."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F
import yaml

pytestmark = pytest.mark.gpu

import pl_yolo_amd  # noqa: E402
from pl_yolo_amd import _lib  # noqa: E402
from pl_yolo_amd._lib import BF16, F32, call  # noqa: E402
from conftest import load_golden, ROOT  # noqa: E402
import hiputil as hu  # noqa: E402


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("N,H,W,Cc,ld", [(2, 12, 10, 16, 16), (1, 7, 9, 48, 64), (3, 20, 20, 128, 128)])
def test_depthwise3x3_kernels_vs_torch(N, H, W, Cc, ld, dtype):
    dt = BF16 if dtype == "bf16" else F32
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(N, Cc, H, W, generator=gen).to(hu.DEV)
    w = (torch.randn(Cc, 1, 3, 3, generator=gen) * 0.3).to(hu.DEV)
    r = torch.randn(N, Cc, H, W, generator=gen).to(hu.DEV)
    if dtype == "bf16":
        x, r = hu.rnd_bf16(x), hu.rnd_bf16(r)
    wq = hu.rnd_bf16(w) if dtype == "bf16" else w
    xr, wr = x.clone().requires_grad_(True), wq.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, None, 1, 1, 1, Cc)
    (y_ref * r).sum().backward()
    xm = hu.to_nhwc(x, dt, ld)
    ym = torch.zeros(N * H * W, ld, dtype=hu.tdtype(dt), device=hu.DEV)
    stats = torch.zeros(8 * 2 * Cc, dtype=torch.float64, device=hu.DEV)
    call("plyolo_dwconv3x3_fwd", dt, N, H, W, Cc, xm.data_ptr(), ld, w.data_ptr(), ym.data_ptr(), ld, stats.data_ptr(), hu.stream())
    y = hu.from_nhwc(ym, N, H, W, Cc)
    tol = 1e-2 if dtype == "bf16" else 1e-5
    assert hu.relerr(y, y_ref.detach()) <= tol
    st = stats.view(8, 2, Cc).sum(0)
    np.testing.assert_allclose(st[0].cpu().numpy(), y_ref.detach().double().sum((0, 2, 3)).cpu().numpy(), rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(st[1].cpu().numpy(), (y_ref.detach().double() ** 2).sum((0, 2, 3)).cpu().numpy(), rtol=1e-4, atol=1e-3)
    rm = hu.to_nhwc(r, dt, ld)
    dxm = torch.full((N * H * W, ld), 1.0, dtype=hu.tdtype(dt), device=hu.DEV)
    call("plyolo_dwconv3x3_dgrad", dt, N, H, W, Cc, rm.data_ptr(), ld, w.data_ptr(), dxm.data_ptr(), ld, 1, hu.stream())     # accumulate onto ones
    dx = hu.from_nhwc(dxm, N, H, W, Cc) - 1.0
    assert hu.relerr(dx, xr.grad) <= (2e-2 if dtype == "bf16" else 1e-5)
    nb = _lib.lib().plyolo_dwconv3x3_wgrad_blocks(dt, N, H, W, Cc)
    partial = torch.empty(nb * Cc * 9, device=hu.DEV)
    dw = torch.zeros(Cc, 1, 3, 3, device=hu.DEV)
    call("plyolo_dwconv3x3_wgrad", dt, N, H, W, Cc, xm.data_ptr(), ld, rm.data_ptr(), ld, partial.data_ptr(), dw.data_ptr(), 0, hu.stream())
    torch.cuda.synchronize()
    assert hu.relerr(dw, wr.grad) <= (1e-2 if dtype == "bf16" else 2e-5)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("N,H,W,Cc", [(2, 5, 7, 16), (1, 20, 20, 64), (2, 1, 3, 8)])
def test_bicubic2x_vs_torch(N, H, W, Cc, dtype):
    dt = BF16 if dtype == "bf16" else F32
    gen = torch.Generator().manual_seed(8)
    x = torch.randn(N, Cc, H, W, generator=gen).to(hu.DEV)
    r = torch.randn(N, Cc, 2 * H, 2 * W, generator=gen).to(hu.DEV)
    if dtype == "bf16":
        x, r = hu.rnd_bf16(x), hu.rnd_bf16(r)
    xr = x.clone().requires_grad_(True)
    y_ref = F.interpolate(xr, scale_factor=2, mode="bicubic")
    (y_ref * r).sum().backward()
    xm = hu.to_nhwc(x, dt)
    ym = torch.zeros(N * 4 * H * W, Cc, dtype=hu.tdtype(dt), device=hu.DEV)
    call("plyolo_bicubic2x_fwd", dt, N, H, W, Cc, xm.data_ptr(), Cc, ym.data_ptr(), Cc, hu.stream())
    y = hu.from_nhwc(ym, N, 2 * H, 2 * W, Cc)
    assert hu.relerr(y, y_ref.detach()) <= (1e-2 if dtype == "bf16" else 2e-6)
    rm = hu.to_nhwc(r, dt)
    dxm = torch.zeros(N * H * W, Cc, dtype=hu.tdtype(dt), device=hu.DEV)
    call("plyolo_bicubic2x_bwd", dt, N, H, W, Cc, rm.data_ptr(), Cc, dxm.data_ptr(), Cc, 0, hu.stream())
    torch.cuda.synchronize()
    dx = hu.from_nhwc(dxm, N, H, W, Cc)
    assert hu.relerr(dx, xr.grad) <= (1e-2 if dtype == "bf16" else 2e-6)


def _cfg():
    with open(os.path.join(ROOT, "configs", "model", "e-yolox", "e-yolox_test.yaml")) as f:
        return yaml.safe_load(f)


def _golden_model(dtype):
    g = load_golden("network_eyolox_test")
    model = pl_yolo_amd.build_model(_cfg(), int(g["num_classes"]))
    sd = {k[6:]: torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith("state/")}
    assert set(sd) == set(model.state_dict().keys())
    model.load_state_dict(sd)
    model.compute_dtype = dtype
    return g, model.to(hu.DEV)


def test_eyolox_fp32_train_step_vs_reference():
    """Parity mode: losses within 1e-4, every gradient within 2e-4 of the largest entry, running statistics, raw maps, eval."""
    g, model = _golden_model("fp32")
    x, labels = torch.from_numpy(g["x"]).to(hu.DEV), torch.from_numpy(g["labels"]).to(hu.DEV)
    model.train()
    with torch.no_grad():
        maps = model(x, None)
    for i, m in enumerate(maps):
        np.testing.assert_allclose(m.cpu().numpy(), g["maps_train%d" % i], rtol=1e-3, atol=2e-4)
    g, model = _golden_model("fp32")
    model.train()
    out = model(x, labels)
    out["loss"].backward()
    torch.cuda.synchronize()
    for k in ("loss", "loss_iou", "loss_obj", "loss_cls"):
        got, want = float(out[k]), float(g["out/" + k])
        print("e-yolox", k, got, want)
        assert abs(got - want) <= 1e-4 * max(1.0, abs(want)), k
    params = dict(model.named_parameters())
    gmax = max(float(np.abs(v).max()) for k, v in g.items() if k.startswith("grad/"))
    worst = 0.0
    for k, v in g.items():
        if k.startswith("grad/"):
            assert params[k[5:]].grad is not None, k
            err = float((params[k[5:]].grad.cpu() - torch.from_numpy(v)).abs().max()) / max(float(np.abs(v).max()), 1e-3 * gmax)
            worst = max(worst, err)
            assert err <= 2e-4, (k, err)
    print("e-yolox worst relative gradient error %.3g" % worst)
    sd = model.state_dict()
    for k, v in g.items():
        if k.startswith("state_after/") and "running" in k:
            np.testing.assert_allclose(sd[k[12:]].cpu().numpy(), v, rtol=1e-4, atol=1e-5, err_msg=k)
    model.eval()
    with torch.no_grad():
        ev = model(x, labels)
    np.testing.assert_allclose(ev.cpu().numpy(), g["eval_out"], rtol=1e-3, atol=2e-3)


def test_eyolox_bf16_step_and_shipped_configs():
    g, model = _golden_model("bf16")
    x, labels = torch.from_numpy(g["x"]).to(hu.DEV), torch.from_numpy(g["labels"]).to(hu.DEV)
    model.train()
    out = model(x, labels)
    out["loss"].backward()
    torch.cuda.synchronize()
    rel = abs(float(out["loss"]) - float(g["out/loss"])) / float(g["out/loss"])
    print("e-yolox bf16 loss %.5f vs %.5f (rel %.3g)" % (float(out["loss"]), float(g["out/loss"]), rel))
    assert rel <= 3e-2
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in model.parameters())
    # the shipped e-yolox-s configuration: one bf16 training step + eval at 256x256
    with open(os.path.join(ROOT, "configs", "model", "e-yolox", "e-yolox-s.yaml")) as f:
        cfg = yaml.safe_load(f)
    torch.manual_seed(96)
    m = pl_yolo_amd.build_model(cfg, 80).to(hu.DEV).train()
    gen = torch.Generator().manual_seed(5)
    xi = (torch.rand(2, 3, 256, 256, generator=gen) * 255).to(hu.DEV)
    lab = torch.zeros(2, 20, 5)
    lab[:, :6, 0] = torch.randint(0, 80, (2, 6), generator=gen).float()
    lab[:, :6, 1:3] = 40 + torch.rand(2, 6, 2, generator=gen) * 170
    lab[:, :6, 3:5] = 16 + torch.rand(2, 6, 2, generator=gen) * 90
    o = m(xi, lab.to(hu.DEV))
    o["loss"].backward()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(o["loss"]))
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in m.parameters())
    m.eval()
    with torch.no_grad():
        pred = m(xi, torch.zeros(2, 1, 5, device=hu.DEV))
    assert tuple(pred.shape) == (2, 32 * 32 + 16 * 16 + 8 * 8, 85)
