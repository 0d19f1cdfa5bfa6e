"""-m gpu: convolution kernels through the C ABI vs plain PyTorch fp32 references.

bf16 path: operands are rounded to bf16 first, so products are exact and the
only differences are the fp32 accumulation order and the final bf16 store
(tolerance: 1 bf16 ulp of the largest output, 2^-7 relative).  fp32 path and
fp32-output epilogues: 2e-5 relative."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from pl_yolo_amd._lib import BF16, F32, call  # noqa: E402
import hiputil as hu  # noqa: E402

# (N, H, W, Cin, Cout, k, stride)
SHAPES = [
    (2, 16, 16, 16, 32, 3, 1),
    (2, 16, 32, 32, 64, 3, 1),
    (1, 24, 20, 64, 128, 3, 1),     # partial tiles in both directions
    (2, 20, 20, 128, 128, 3, 1),
    (1, 16, 16, 256, 256, 3, 1),    # two cout tiles, four cin chunks
    (2, 32, 32, 32, 64, 3, 2),
    (1, 40, 40, 128, 256, 3, 2),
    (3, 18, 22, 16, 32, 3, 2),      # odd tile counts
    (2, 16, 16, 64, 64, 1, 1),      # pointwise -> flattened pixel rows
    (2, 20, 20, 512, 256, 1, 1),
    (1, 12, 12, 8, 16, 1, 1),       # K tail (Cin 8 < MFMA K 16), 144 pixels (flatten ok)
    (1, 10, 10, 24, 40, 1, 1),      # 100 pixels: not a multiple of 16 -> 2-D tiling of a 1x1
    (2, 16, 16, 8, 16, 3, 2),
    (2, 32, 32, 32, 8, 3, 2),       # Cout 8: a single 16-byte output vector per pixel
    (2, 16, 16, 16, 8, 1, 1),
    (1, 160, 160, 32, 8, 3, 2),
    (16, 80, 80, 64, 128, 3, 1),    # >= 384 tiles of 16x16: the 256-pixel tile (8 fragments per wave)
    (4, 160, 160, 32, 64, 3, 1),    # 16x16 tile, BN 64 (2x2 waves)
    (8, 160, 160, 16, 32, 3, 1),    # 16x16 tile, BN 32 (4 waves along pixels)
    (16, 80, 80, 256, 128, 1, 1),   # pointwise, 16x16 tile, 4 chunks
    (6, 72, 88, 48, 80, 3, 1),      # 16x16 tile with ragged edges, Cin tail chunk (48 = 32 + 16), Cout 80
    (2, 160, 160, 64, 64, 3, 2),    # stride 2 dgrad parity classes on 16x16 tiles
]


def _ref_conv(x, w, s, bias=None):
    k = w.shape[-1]
    return F.conv2d(x, w, bias, s, (k - 1) // 2)


@pytest.mark.parametrize("dt", [BF16, F32], ids=["bf16", "fp32"])
@pytest.mark.parametrize("shape", SHAPES, ids=[str(s) for s in SHAPES])
def test_conv_fwd_stats(dt, shape):
    N, H, W, Cin, Cout, k, s = shape
    torch.manual_seed(sum(shape))
    x = torch.randn(N, Cin, H, W, device=hu.DEV)
    w = torch.randn(Cout, Cin, k, k, device=hu.DEV) / (Cin * k * k) ** 0.5
    if dt == BF16:
        x, w = hu.rnd_bf16(x), hu.rnd_bf16(w)
    ref = _ref_conv(x, w, s)
    OH, OW = ref.shape[2:]
    x_ld, y_ld = Cin + 16, Cout + 8          # exercise concat-slice pitches
    xm = hu.to_nhwc(x, dt, x_ld)
    pk = hu.Packed(w, dt)
    y = torch.full((N * OH * OW, y_ld), 3.0, dtype=hu.tdtype(dt), device=hu.DEV)
    d = hu.conv_desc(dt, N, H, W, Cin, Cout, k, s, x_ld, y_ld)
    stats = torch.zeros(hu._lib.STAT_SLOTS, 2, Cout, dtype=torch.float64, device=hu.DEV)   # fp64 stat slots
    call("plyolo_conv2d_fwd", C.byref(d), xm.data_ptr(), pk.wp.data_ptr(), None, y.data_ptr(), stats.data_ptr(), hu.stream())
    torch.cuda.synchronize()
    got = hu.from_nhwc(y, N, OH, OW, Cout)
    tol = 2.0 ** -7 if dt == BF16 else 2e-5
    err = hu.relerr(got, ref)
    print("conv_fwd", shape, "relerr %.3g" % err)
    assert err <= tol
    assert torch.all(y[:, Cout:].float() == 3.0), "pad columns of the output pitch were overwritten"
    # fused BatchNorm statistics (computed from the fp32 accumulators)
    s1 = stats[:, 0].sum(0)
    s2 = stats[:, 1].sum(0)
    r1 = ref.double().sum((0, 2, 3))
    r2 = (ref.double() ** 2).sum((0, 2, 3))
    assert float((s1 - r1).abs().max()) <= 2e-4 * float(ref.abs().sum((0, 2, 3)).max())
    assert float((s2 - r2).abs().max()) <= 2e-4 * float(r2.max())


@pytest.mark.parametrize("dt", [BF16, F32], ids=["bf16", "fp32"])
@pytest.mark.parametrize("cout", [80, 5])
def test_conv_fwd_f32_out_bias(dt, cout):
    """Head prediction convs: fp32 output with bias into a channel slice of 5+C rows."""
    N, H, W, Cin = 2, 20, 20, 128
    torch.manual_seed(cout)
    x = torch.randn(N, Cin, H, W, device=hu.DEV)
    w = torch.randn(cout, Cin, 1, 1, device=hu.DEV) / Cin ** 0.5
    b = torch.randn(cout, device=hu.DEV)
    if dt == BF16:
        x, w = hu.rnd_bf16(x), hu.rnd_bf16(w)
    ref = _ref_conv(x, w, 1, b)
    xm = hu.to_nhwc(x, dt, Cin)
    pk = hu.Packed(w, dt, bias=b)
    nch, off = 85, (5 if cout == 80 else 0)
    raw = torch.full((N * H * W, nch), -9.0, dtype=torch.float32, device=hu.DEV)
    d = hu.conv_desc(dt, N, H, W, Cin, cout, 1, 1, Cin, nch, 1)
    call("plyolo_conv2d_fwd", C.byref(d), xm.data_ptr(), pk.wp.data_ptr(), pk.bp.data_ptr(), raw.data_ptr() + off * 4, None, hu.stream())
    torch.cuda.synchronize()
    got = raw[:, off:off + cout].reshape(N, H, W, cout).permute(0, 3, 1, 2)
    err = hu.relerr(got, ref)
    print("conv_fwd_f32out", cout, "relerr %.3g" % err)
    assert err <= 2e-5
    untouched = torch.ones(nch, dtype=torch.bool)
    untouched[off:off + cout] = False
    assert torch.all(raw[:, untouched] == -9.0)


@pytest.mark.parametrize("dt", [BF16, F32], ids=["bf16", "fp32"])
@pytest.mark.parametrize("shape", SHAPES, ids=[str(s) for s in SHAPES])
def test_conv_dgrad(dt, shape):
    N, H, W, Cin, Cout, k, s = shape
    torch.manual_seed(sum(shape) + 1)
    x = torch.randn(N, Cin, H, W, device=hu.DEV, requires_grad=True)
    w = torch.randn(Cout, Cin, k, k, device=hu.DEV) / (Cout * k * k) ** 0.5
    if dt == BF16:
        w = hu.rnd_bf16(w)
    y = _ref_conv(x, w, s)
    OH, OW = y.shape[2:]
    dy = torch.randn_like(y)
    if dt == BF16:
        dy = hu.rnd_bf16(dy)
    (ref,) = torch.autograd.grad(y, x, dy)
    x_ld, y_ld = Cin + 8, Cout + 16
    dym = hu.to_nhwc(dy, dt, y_ld)
    pk = hu.Packed(w, dt)
    base = torch.randn(N, Cin, H, W, device=hu.DEV)
    if dt == BF16:
        base = hu.rnd_bf16(base)
    d = hu.conv_desc(dt, N, H, W, Cin, Cout, k, s, x_ld, y_ld)
    for acc in (0, 1):
        dx = hu.to_nhwc(base, dt, x_ld, fill=5.0)
        call("plyolo_conv2d_dgrad", C.byref(d), dym.data_ptr(), pk.wpd.data_ptr(), dx.data_ptr(), acc, hu.stream())
        torch.cuda.synchronize()
        got = hu.from_nhwc(dx, N, H, W, Cin)
        want = ref + base if acc else ref
        tol = 2.0 ** -6 if dt == BF16 else 2e-5
        err = hu.relerr(got, want)
        print("conv_dgrad", shape, "acc", acc, "relerr %.3g" % err)
        assert err <= tol
        assert torch.all(dx[:, Cin:].float() == 5.0)


@pytest.mark.parametrize("dt", [BF16, F32], ids=["bf16", "fp32"])
@pytest.mark.parametrize("shape", SHAPES, ids=[str(s) for s in SHAPES])
def test_conv_wgrad(dt, shape):
    N, H, W, Cin, Cout, k, s = shape
    torch.manual_seed(sum(shape) + 2)
    x = torch.randn(N, Cin, H, W, device=hu.DEV)
    w = (torch.randn(Cout, Cin, k, k, device=hu.DEV) / (Cin * k * k) ** 0.5).requires_grad_(True)
    if dt == BF16:
        x = hu.rnd_bf16(x)
    y = _ref_conv(x, w, s)
    OH, OW = y.shape[2:]
    dy = torch.randn_like(y)
    if dt == BF16:
        dy = hu.rnd_bf16(dy)
    (ref,) = torch.autograd.grad(y, w, dy)
    x_ld, y_ld = Cin + 8, Cout + 8
    xm, dym = hu.to_nhwc(x, dt, x_ld), hu.to_nhwc(dy, dt, y_ld)
    pk = hu.Packed(w.detach(), dt)
    d = hu.conv_desc(dt, N, H, W, Cin, Cout, k, s, x_ld, y_ld)
    pk.set_slabs(d)
    call("plyolo_conv2d_wgrad", C.byref(d), xm.data_ptr(), dym.data_ptr(), pk.dwp.data_ptr(), hu.stream())
    got = pk.unpack()
    torch.cuda.synchronize()
    err = hu.relerr(got, ref)
    print("conv_wgrad", shape, "relerr %.3g" % err)
    assert err <= 1e-4


@pytest.mark.parametrize("cout", [80, 5])
def test_head_pred_backward_bf16(cout):
    """dgrad / wgrad / bias grad of the head prediction convs from the bf16 gradient
    buffers the loss backward writes (Cout 5 lives in 16-channel rows)."""
    N, H, W, Cin = 2, 20, 20, 128
    torch.manual_seed(cout + 3)
    x = hu.rnd_bf16(torch.randn(N, Cin, H, W, device=hu.DEV)).requires_grad_(True)
    w = hu.rnd_bf16(torch.randn(cout, Cin, 1, 1, device=hu.DEV) / Cin ** 0.5).requires_grad_(True)
    b = torch.zeros(cout, device=hu.DEV, requires_grad=True)
    y = F.conv2d(x, w, b)
    dy = hu.rnd_bf16(torch.randn_like(y) * 0.01)
    gx, gw, gb = torch.autograd.grad(y, (x, w, b), dy)
    ld = 16 if cout == 5 else 80
    dym = torch.zeros(N * H * W, ld, dtype=torch.bfloat16, device=hu.DEV)
    dym[:, :cout] = dy.permute(0, 2, 3, 1).reshape(-1, cout).to(torch.bfloat16)
    xm = hu.to_nhwc(x.detach(), BF16, Cin)
    pk = hu.Packed(w.detach(), BF16, bias=b.detach())
    d = hu.conv_desc(BF16, N, H, W, Cin, cout, 1, 1, Cin, ld)
    dx = torch.zeros(N * H * W, Cin, dtype=torch.bfloat16, device=hu.DEV)
    call("plyolo_conv2d_dgrad", C.byref(d), dym.data_ptr(), pk.wpd.data_ptr(), dx.data_ptr(), 0, hu.stream())
    pk.set_slabs(d)
    call("plyolo_conv2d_wgrad", C.byref(d), xm.data_ptr(), dym.data_ptr(), pk.dwp.data_ptr(), hu.stream())
    call("plyolo_bias_grad", BF16, dym.data_ptr(), N * H * W, cout, ld, pk.dbp.data_ptr(), hu.stream())
    dw = pk.unpack()
    torch.cuda.synchronize()
    e1 = hu.relerr(hu.from_nhwc(dx, N, H, W, Cin), gx)
    e2 = hu.relerr(dw, gw)
    e3 = hu.relerr(pk.db, gb)
    print("head_pred_bwd", cout, "dgrad %.3g wgrad %.3g bias %.3g" % (e1, e2, e3))
    assert e1 <= 2.0 ** -6 and e2 <= 1e-4 and e3 <= 1e-4

@pytest.mark.parametrize("shape", [(2, 20, 20, 128, 128, 3, 1), (2, 32, 32, 32, 64, 3, 2), (2, 16, 16, 64, 64, 1, 1), (16, 80, 80, 64, 128, 3, 1)], ids=str)
@pytest.mark.parametrize("with_res", [False, True])
def test_conv_fwd_fused_bn_act_inference(shape, with_res):
    """plyolo_conv2d_fwd_bn_act: eval-mode BaseConv (+ Bottleneck shortcut) in one launch ==
    x_res + silu(batch_norm_eval(conv(x)))."""
    N, H, W, Cin, Cout, k, s = shape
    torch.manual_seed(sum(shape) + with_res)
    x = hu.rnd_bf16(torch.randn(N, Cin, H, W, device=hu.DEV))
    w = hu.rnd_bf16(torch.randn(Cout, Cin, k, k, device=hu.DEV) / (Cin * k * k) ** 0.5)
    gamma, beta = torch.rand(Cout, device=hu.DEV) + 0.5, torch.rand(Cout, device=hu.DEV) - 0.5
    rm, rv = torch.randn(Cout, device=hu.DEV) * 0.1, torch.rand(Cout, device=hu.DEV) + 0.5
    ref = F.silu(F.batch_norm(_ref_conv(x, w, s), rm, rv, gamma, beta, False, 0.03, 1e-3))
    OH, OW = ref.shape[2:]
    res = hu.rnd_bf16(torch.randn_like(ref)) if with_res else None
    if with_res:
        ref = ref + res
    coef = torch.zeros(4 * Cout, device=hu.DEV)
    call("plyolo_bn_eval_coef", Cout, gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr(), 1e-3, coef.data_ptr(), hu.stream())
    x_ld, y_ld = Cin + 8, Cout + 16
    xm = hu.to_nhwc(x, BF16, x_ld)
    resm = hu.to_nhwc(res, BF16, Cout + 8) if with_res else None
    pk = hu.Packed(w, BF16)
    y = torch.full((N * OH * OW, y_ld), 3.0, dtype=torch.bfloat16, device=hu.DEV)
    d = hu.conv_desc(BF16, N, H, W, Cin, Cout, k, s, x_ld, y_ld)
    call("plyolo_conv2d_fwd_bn_act", C.byref(d), xm.data_ptr(), pk.wp.data_ptr(), coef.data_ptr(), 1,
         resm.data_ptr() if with_res else None, Cout + 8 if with_res else 0, y.data_ptr(), hu.stream())
    torch.cuda.synchronize()
    err = hu.relerr(hu.from_nhwc(y, N, OH, OW, Cout), ref)
    assert err <= 2.0 ** -6, err
    assert torch.all(y[:, Cout:].float() == 3.0)


@pytest.mark.parametrize("shape", [s for s in SHAPES if s[5] == 3 and s[6] == 1 and s[3] % 16 == 0 and s[3] <= 128] + [(3, 20, 20, 128, 40, 3, 1), (2, 13, 29, 16, 104, 3, 1)], ids=str)
def test_conv3ws_opt_in_kernel(shape, monkeypatch):
    """The weights-stationary 3x3 kernel (csrc/conv3ws.hip, PLYOLO_CONV3WS=1, off by default): forward + BatchNorm statistics
    and the data gradient (overwrite and accumulate) on every 3x3 stride-1 shape it accepts, ragged maps included."""
    monkeypatch.setenv("PLYOLO_CONV3WS", "1")
    test_conv_fwd_stats(BF16, shape)
    test_conv_dgrad(BF16, shape)
