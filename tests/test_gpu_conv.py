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
    (3, 26, 38, 64, 128, 3, 2),     # stride-2 forward instances (4-row tiles, de-interleaved halo): ragged 13 x 19 output, two chunks
    (2, 22, 46, 96, 64, 3, 2),      # ... Cin 96 = three chunks, 11 x 23 output, 64-channel block
    # one small-spatial case per instance class the WIDE plans select (yolox_x.yaml: 320 / 640 / 1280 / 2560 channels; yolox_l.yaml and
    # yolov7.yaml: 1024 / 2048): many input chunks x several output blocks, channel counts that are not powers of two
    (2, 20, 20, 320, 320, 3, 1),    # ten chunks, three output blocks (the last one 64 of 128 channels)
    (1, 10, 10, 640, 640, 3, 1),
    (1, 8, 8, 1024, 1024, 3, 1),
    (1, 12, 12, 2560, 1280, 1, 1),  # SPP bottleneck conv2 of YOLOX-x
    (1, 12, 12, 2048, 1024, 1, 1),
    (2, 20, 20, 80, 160, 1, 1),     # 80 = a 64 + 16 channel chunk tail
    (1, 16, 16, 640, 1280, 3, 2),
    # row-flattened tiles of the 20- / 40-wide maps (conv_mfma_flat.hip): 80- and 160-pixel tiles of four whole rows, forward and data gradient
    (2, 40, 40, 128, 128, 3, 1),
    (1, 12, 40, 64, 192, 3, 1),     # Cout 192 = a block and a half; dgrad: dx 64 channels -> rectangular tiles
    (3, 8, 20, 96, 256, 3, 1),      # three input chunks, two output blocks
    (2, 20, 20, 256, 256, 3, 1),
    # ragged output-channel blocks (conv_mfma_rag.hip): full 128-channel blocks + one 32- / 64-channel block, forward and data gradient
    (2, 24, 24, 160, 160, 3, 1),    # 128 + 32 both ways (YOLOX-x dark3)
    (1, 16, 48, 96, 288, 3, 1),     # 2 x 128 + 32 forward; the data gradient's 96 channels keep one whole block
    (1, 24, 40, 192, 192, 3, 1),    # 128 + 64 both ways
    (1, 20, 20, 320, 320, 1, 1),    # pointwise, 2 x 128 + 64 both ways (conv_pw_rag_kernel)
    (2, 12, 16, 96, 160, 1, 1),     # pointwise, 128 + 32 forward
    (2, 34, 30, 80, 160, 3, 2),     # stride 2: forward 128 + (32 of a 64-channel block)
    (1, 28, 36, 160, 320, 3, 2),    # stride 2: forward 2 x 128 + 64, data gradient (four parity jobs) 128 + 32
    (1, 20, 24, 320, 320, 3, 2),    # stride 2: data gradient 2 x 128 + 64
    # one 96-channel block of three waves (65 .. 96 output channels)
    (2, 40, 48, 80, 80, 3, 1),      # YOLOX-x dark2 bottleneck: both ways
    (2, 32, 32, 16, 80, 3, 1),      # YOLOX-x stem: a single 16-channel chunk
    (1, 30, 34, 96, 96, 3, 1),      # YOLOX-m width
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


@pytest.mark.parametrize("shape", [(2, 40, 40, 128, 128), (1, 12, 40, 192, 192), (3, 8, 20, 96, 256), (2, 20, 20, 256, 256), (5, 24, 20, 128, 160)], ids=str)
def test_row_flattened_tiles_match_rectangular_tiles(shape, monkeypatch):
    """conv_mfma_flat.hip (tiles of four whole rows on 20- / 40-wide maps) against the rectangular 8 x 16 / 4 x 16 tiles (PLYOLO_FLAT=0): every
    output element is the same sum in the same order (chunk, tap, k), so forward, data gradient (overwrite and accumulate) and the folded
    BatchNorm reduction's dx are bit-identical; the BatchNorm statistics agree to their fp64 grouping."""
    from pl_yolo_amd._lib import STAT_SLOTS
    N, H, W, Cin, Cout = shape
    dt, dev = BF16, hu.DEV
    torch.manual_seed(sum(shape) + 21)
    M = N * H * W
    x = torch.randn(M, Cin + 8, device=dev).to(torch.bfloat16)
    w = hu.rnd_bf16(torch.randn(Cout, Cin, 3, 3, device=dev) / (9 * Cin) ** 0.5)
    pk = hu.Packed(w, dt)
    dy = torch.randn(M, Cout + 8, device=dev).to(torch.bfloat16)
    d = hu.conv_desc(dt, N, H, W, Cin, Cout, 3, 1, Cin + 8, Cout + 8)
    res = {}
    for flat in ("1", "0"):
        monkeypatch.setenv("PLYOLO_FLAT", "2" if flat == "1" else "0")     # 2: the 40-wide maps too (by default only the 20-wide ones)
        y = torch.full((M, Cout + 8), 3.0, dtype=torch.bfloat16, device=dev)
        stats = torch.zeros(STAT_SLOTS, 2, Cout, dtype=torch.float64, device=dev)
        call("plyolo_conv2d_fwd", C.byref(d), x.data_ptr(), pk.wp.data_ptr(), None, y.data_ptr(), stats.data_ptr(), hu.stream())
        dxs = []
        for acc in (0, 1):
            dx = torch.full((M, Cin + 8), 0.5, dtype=torch.bfloat16, device=dev)
            call("plyolo_conv2d_dgrad", C.byref(d), dy.data_ptr(), pk.wpd.data_ptr(), dx.data_ptr(), acc, hu.stream())
            dxs.append(dx)
        torch.cuda.synchronize()
        res[flat] = (y, stats.sum(0), dxs)
    (y1, s1, dx1), (y0, s0, dx0) = res["1"], res["0"]
    assert torch.equal(y1.view(torch.int16), y0.view(torch.int16)), "forward differs"
    assert float((s1 - s0).abs().max()) <= 1e-6 * float(s0.abs().max())      # (fp32 partial sums per tile, another tile shape)
    for a, b in zip(dx1, dx0):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16)), "data gradient differs"
    assert float(y1[:, :Cout].float().abs().max()) > 0 and torch.all(y1[:, Cout:].float() == 3.0)


# (N, H, W, Cin, Cout): wide pointwise layers on the deep-K GEMM tiles of csrc/conv_wgrad1w.hip -- 256 x 256 and 192 x 192 tiles, several
# channel tiles per side, channel tails inside a tile (160 = 5 x 32 of 192; 200 of 256), a pixel count that is no multiple of the
# 32-pixel stage, one and many pixel ranges (slabs)
WGRAD1W_SHAPES = [(1, 13, 9, 160, 160), (2, 20, 20, 320, 160), (1, 25, 25, 640, 320), (3, 7, 11, 168, 200), (1, 40, 40, 256, 512), (4, 40, 40, 320, 320),
                  (2, 16, 16, 1280, 640)]


@pytest.mark.parametrize("shape", [(2, 24, 24, 160, 160, 3, 1), (1, 16, 48, 96, 288, 3, 1), (1, 24, 40, 192, 192, 3, 1), (3, 17, 29, 64, 320, 3, 1), (2, 20, 20, 320, 320, 1, 1),
                                   (2, 12, 16, 96, 160, 1, 1), (1, 33, 7, 200, 448, 1, 1), (2, 34, 30, 80, 160, 3, 2), (1, 28, 36, 160, 320, 3, 2), (1, 21, 25, 320, 192, 3, 2),
                                   (2, 40, 48, 80, 80, 3, 1), (2, 32, 32, 16, 80, 3, 1), (1, 30, 34, 96, 96, 3, 1), (3, 18, 22, 72, 96, 3, 2)], ids=str)
def test_ragged_channel_blocks_match_whole_blocks(shape, monkeypatch):
    """conv_mfma_rag.hip / conv_pw_rag_kernel / the stride-2 forward and four-job data gradient: the last output-channel block of a 160- /
    320-channel layer as a 32- / 64-channel instance instead of a whole 128-channel block, and a 65 .. 96-channel layer as one block of three waves.  Same contraction order per output element:
    forward, data gradient (plain and accumulating) bit for bit against PLYOLO_RAG=0; the BatchNorm statistics (another
    fragment-to-wave assignment) to fp32 rounding."""
    N, H, W, Cin, Cout, k, st = shape
    torch.manual_seed(sum(shape))
    x = hu.rnd_bf16(torch.randn(N, Cin, H, W, device=hu.DEV))
    w = hu.rnd_bf16(torch.randn(Cout, Cin, k, k, device=hu.DEV) / (Cin * k * k) ** 0.5)
    ref = _ref_conv(x, w, st)
    OH, OW = ref.shape[2:]
    dy = hu.rnd_bf16(torch.randn(N, Cout, OH, OW, device=hu.DEV))
    x_ld, y_ld = Cin + 8, Cout + 16
    xm, dym = hu.to_nhwc(x, BF16, x_ld), hu.to_nhwc(dy, BF16, y_ld)
    pk = hu.Packed(w, BF16)
    d = hu.conv_desc(BF16, N, H, W, Cin, Cout, k, st, x_ld, y_ld)
    out = {}
    for rag in ("0", "1"):
        monkeypatch.setenv("PLYOLO_RAG", rag)
        y = torch.full((N * OH * OW, y_ld), 3.0, dtype=torch.bfloat16, device=hu.DEV)
        stats = torch.zeros(hu._lib.STAT_SLOTS, 2, Cout, dtype=torch.float64, device=hu.DEV)
        call("plyolo_conv2d_fwd", C.byref(d), xm.data_ptr(), pk.wp.data_ptr(), None, y.data_ptr(), stats.data_ptr(), hu.stream())
        dx = torch.full((N * H * W, x_ld), 2.0, dtype=torch.bfloat16, device=hu.DEV)
        call("plyolo_conv2d_dgrad", C.byref(d), dym.data_ptr(), pk.wpd.data_ptr(), dx.data_ptr(), 0, hu.stream())
        dxa = dx.clone()
        call("plyolo_conv2d_dgrad", C.byref(d), dym.data_ptr(), pk.wpd.data_ptr(), dxa.data_ptr(), 1, hu.stream())
        torch.cuda.synchronize()
        out[rag] = (y, stats.sum(0), dx, dxa)
    assert torch.equal(out["0"][0], out["1"][0])
    assert torch.equal(out["0"][2], out["1"][2]) and torch.equal(out["0"][3], out["1"][3])
    assert float((out["0"][1] - out["1"][1]).abs().max()) <= 1e-6 * float(out["0"][1].abs().max())
    assert hu.relerr(hu.from_nhwc(out["1"][0], N, OH, OW, Cout), ref) <= 2.0 ** -7


@pytest.mark.parametrize("shape", WGRAD1W_SHAPES, ids=str)
@pytest.mark.parametrize("wgs", [0, 24])
def test_wide_pointwise_wgrad_tiles(shape, wgs, monkeypatch):
    """plyolo_conv2d_wgrad of a wide 1x1 layer (conv_wgrad1w.hip) against torch's fp32 weight gradient of the same bf16 operands, with the
    planned number of pixel ranges and with a small one (many stages per range, ranges that end inside the last stage), and against
    the 128 x 128-slab kernel it replaces (PLYOLO_WG1W=0) -- same products, another summation order."""
    N, H, W, Cin, Cout = shape
    if wgs:
        monkeypatch.setenv("PLYOLO_WG1W_WGS", str(wgs))
    full = (N, H, W, Cin, Cout, 1, 1)
    test_conv_wgrad(BF16, full)
    d = hu.conv_desc(BF16, N, H, W, Cin, Cout, 1, 1, Cin + 8, Cout + 8)
    ns = hu._lib.lib().plyolo_conv2d_wgrad_slabs(C.byref(d))
    assert ns >= 1
    print("wide wgrad", shape, "slabs", ns)



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

@pytest.mark.parametrize("shape", [(2, 20, 20, 128, 128, 3, 1), (2, 32, 32, 32, 64, 3, 2), (2, 16, 16, 64, 64, 1, 1), (16, 80, 80, 64, 128, 3, 1),
                                   (2, 24, 24, 160, 160, 3, 1), (1, 16, 16, 96, 320, 1, 1), (2, 12, 12, 64, 160, 1, 1), (1, 16, 24, 64, 192, 3, 1)], ids=str)
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
    if not (hu._lib.lib().plyolo_build_flags() & 1):
        pytest.skip("opt-in kernel: libplyolo_hip.so built without OPTIN=1")
    monkeypatch.setenv("PLYOLO_CONV3WS", "1")
    test_conv_fwd_stats(BF16, shape)
    test_conv_dgrad(BF16, shape)


# (N, H, W, Cin, Cout, split): pointwise units; Cin 256 = two output blocks of the kernel, Cout 96 = a 64 + 32 channel chunk tail
@pytest.mark.parametrize("shape", [(2, 20, 20, 128, 128, 0), (3, 20, 20, 256, 96, 0), (2, 40, 40, 64, 128, 64), (1, 13, 9, 32, 64, 24), (4, 40, 40, 16, 32, 0)], ids=str)
@pytest.mark.parametrize("act", ["silu", "lrelu", None])
def test_pointwise_dgrad_with_fused_bn_backward(shape, act):
    """plyolo_conv2d_dgrad_bn == plyolo_bn_act_bwd_dz + plyolo_conv2d_dgrad, bit for bit (dz, dx, dgamma, dbeta), also for a merged pair
    (output gradient in two matrices, two BatchNorm parameter sets) and with accumulation into dx."""
    from pl_yolo_amd._lib import ACT, BnBwdFuse, Split, BnBwdSplit, STAT_SLOTS
    N, H, W, Cin, Cout, split = shape
    dt, M = BF16, N * H * W
    torch.manual_seed(sum(shape) + 3)
    dev = hu.DEV
    w = hu.rnd_bf16(torch.randn(Cout, Cin, 1, 1, device=dev) / Cout ** 0.5)
    pk = hu.Packed(w, dt)
    z = (torch.randn(M, Cout, device=dev) * 1.5).to(torch.bfloat16)
    Ca = split if split else Cout
    d_ld = Ca + 8
    dout = torch.randn(M, d_ld, device=dev).to(torch.bfloat16)
    dout2 = torch.randn(M, Cout - Ca + 16, device=dev).to(torch.bfloat16) if split else None
    gamma, gamma2 = torch.rand(Cout, device=dev) + 0.5, torch.rand(Cout, device=dev) + 0.5
    mean, invstd = torch.randn(Cout, device=dev) * 0.1, torch.rand(Cout, device=dev) + 0.5
    beta = torch.randn(Cout, device=dev) * 0.1
    g_eff = torch.cat([gamma[:Ca], gamma2[:Cout - Ca]]) if split else gamma
    scale = g_eff * invstd
    coef = torch.cat([scale, beta - mean * scale, mean, invstd]).contiguous()
    bslots = torch.zeros(STAT_SLOTS * 2 * Cout, dtype=torch.float64, device=dev)
    sp = Split()
    if split:
        sp.split, sp.p2, sp.ld2 = Ca, dout2.data_ptr(), dout2.shape[1]
    a = ACT[act]
    call("plyolo_bn_act_bwd_reduce", dt, M, Cout, dout.data_ptr(), d_ld, z.data_ptr(), Cout, coef.data_ptr(), a, bslots.data_ptr(),
         C.byref(sp) if split else None, hu.stream())
    x_ld = Cin + 8
    d = hu.conv_desc(dt, N, H, W, Cin, Cout, 1, 1, x_ld, Cout)
    assert hu._lib.lib().plyolo_conv2d_dgrad_bn_fits(C.byref(d), a) == 1
    base = torch.randn(M, x_ld, device=dev).to(torch.bfloat16)
    for acc in (0, 1):
        # reference: the two separate launches
        dz0 = torch.zeros(M, Cout, dtype=torch.bfloat16, device=dev)
        dg0, db0, dg0b, db0b = (torch.zeros(Cout, device=dev) for _ in range(4))
        p2 = BnBwdSplit()
        if split:
            p2.split, p2.gamma2, p2.dgamma2, p2.dbeta2 = Ca, gamma2.data_ptr(), dg0b.data_ptr(), db0b.data_ptr()
        call("plyolo_bn_act_bwd_dz", dt, M, Cout, dout.data_ptr(), d_ld, z.data_ptr(), Cout, coef.data_ptr(), bslots.data_ptr(), gamma.data_ptr(),
             dg0.data_ptr(), db0.data_ptr(), 0, a, dz0.data_ptr(), Cout, C.byref(sp) if split else None, C.byref(p2) if split else None, hu.stream())
        dx0 = base.clone()
        call("plyolo_conv2d_dgrad", C.byref(d), dz0.data_ptr(), pk.wpd.data_ptr(), dx0.data_ptr(), acc, hu.stream())
        # fused
        dz1 = torch.zeros(M, Cout, dtype=torch.bfloat16, device=dev)
        dg1, db1, dg1b, db1b = (torch.zeros(Cout, device=dev) for _ in range(4))
        f = BnBwdFuse()
        f.dout, f.dout_ld, f.z, f.z_ld, f.coef, f.bslots = dout.data_ptr(), d_ld, z.data_ptr(), Cout, coef.data_ptr(), bslots.data_ptr()
        if split:
            f.dout2, f.dout2_ld, f.dout_split = dout2.data_ptr(), dout2.shape[1], Ca
            f.par_split, f.gamma2, f.dgamma2, f.dbeta2 = Ca, gamma2.data_ptr(), dg1b.data_ptr(), db1b.data_ptr()
        f.gamma, f.dgamma, f.dbeta, f.act, f.dz, f.dz_ld = gamma.data_ptr(), dg1.data_ptr(), db1.data_ptr(), a, dz1.data_ptr(), Cout
        dx1 = base.clone()
        call("plyolo_conv2d_dgrad_bn", C.byref(d), C.byref(f), pk.wpd.data_ptr(), dx1.data_ptr(), acc, hu.stream())
        torch.cuda.synchronize()
        assert torch.equal(dz0.view(torch.int16), dz1.view(torch.int16)), "dz differs"
        assert torch.equal(dx0[:, :Cin].view(torch.int16), dx1[:, :Cin].view(torch.int16)), "dx differs"
        assert torch.equal(dx1[:, Cin:], base[:, Cin:]), "pad columns of dx touched"
        for u, v in ((dg0, dg1), (db0, db1), (dg0b, dg1b), (db0b, db1b)):
            assert torch.equal(u, v)
        assert float(dz1.float().abs().max()) > 0


# (N, H, W, Cin, Cout, split, TH4 allowed): 3x3 stride-1 units, one per loader instance -- 128 / 64 / 32 channel blocks of dx, 8- and 4-row
# tiles, one chunk and several, two output blocks (only block 0 writes dz), merged pairs (two output-gradient matrices), ragged maps
BNB3_SHAPES = [(2, 20, 20, 128, 128, 0, 1), (2, 20, 20, 128, 128, 0, 0), (3, 24, 20, 64, 64, 0, 0), (3, 24, 20, 64, 64, 0, 1), (2, 40, 40, 256, 128, 0, 0),
               (2, 40, 40, 128, 256, 128, 0), (1, 13, 29, 64, 96, 32, 1), (2, 48, 40, 32, 32, 0, 0), (2, 37, 45, 32, 64, 0, 0), (1, 21, 19, 24, 40, 0, 0)]


@pytest.mark.parametrize("shape", BNB3_SHAPES, ids=str)
def test_conv3x3_dgrad_with_fused_bn_backward(shape, monkeypatch):
    """plyolo_conv2d_dgrad_bn on a 3x3 stride-1 SiLU unit == plyolo_bn_act_bwd_dz + plyolo_conv2d_dgrad (+ the shortcut's copy), bit for
    bit: dz (written once, by the workgroups of output block 0), dx (overwrite and accumulate), dgamma, dbeta, the forwarded shortcut
    gradient; padding pixels of the halo stage zeros (a zero dout would stage Cc); dz == NULL writes nothing; and with the
    BatchNorm-backward reduction of the unit upstream folded into the same launch."""
    from pl_yolo_amd._lib import ACT, BnBwdFuse, Split, BnBwdSplit, STAT_SLOTS, BnRed
    N, H, W, Cin, Cout, split, t4 = shape
    if not t4:
        monkeypatch.setenv("PLYOLO_TH4_MAX_WG", "0")
    dt, M = BF16, N * H * W
    torch.manual_seed(sum(shape) + 11)
    dev = hu.DEV
    w = hu.rnd_bf16(torch.randn(Cout, Cin, 3, 3, device=dev) / (9 * Cout) ** 0.5)
    pk = hu.Packed(w, dt)
    z = (torch.randn(M, Cout, device=dev) * 1.5).to(torch.bfloat16)
    Ca = split if split else Cout
    d_ld = Ca + 8
    dout = torch.randn(M, d_ld, device=dev).to(torch.bfloat16)
    dout2 = torch.randn(M, Cout - Ca + 16, device=dev).to(torch.bfloat16) if split else None
    gamma, gamma2 = torch.rand(Cout, device=dev) + 0.5, torch.rand(Cout, device=dev) + 0.5
    mean, invstd = torch.randn(Cout, device=dev) * 0.1, torch.rand(Cout, device=dev) + 0.5
    beta = torch.randn(Cout, device=dev) * 0.1
    g_eff = torch.cat([gamma[:Ca], gamma2[:Cout - Ca]]) if split else gamma
    scale = g_eff * invstd
    coef = torch.cat([scale, beta - mean * scale, mean, invstd]).contiguous()
    bslots = torch.zeros(STAT_SLOTS * 2 * Cout, dtype=torch.float64, device=dev)
    sp = Split()
    if split:
        sp.split, sp.p2, sp.ld2 = Ca, dout2.data_ptr(), dout2.shape[1]
    a = ACT["silu"]
    call("plyolo_bn_act_bwd_reduce", dt, M, Cout, dout.data_ptr(), d_ld, z.data_ptr(), Cout, coef.data_ptr(), a, bslots.data_ptr(),
         C.byref(sp) if split else None, hu.stream())
    x_ld = Cin + 8
    d = hu.conv_desc(dt, N, H, W, Cin, Cout, 3, 1, x_ld, Cout)
    lib = hu._lib.lib()
    assert lib.plyolo_conv2d_dgrad_bn_fits(C.byref(d), a) == 1
    assert lib.plyolo_conv2d_dgrad_bn_fits(C.byref(d), ACT["lrelu"]) == 0      # the 3x3 loader carries SiLU only
    base = torch.randn(M, x_ld, device=dev).to(torch.bfloat16)
    fwd_ld = Cout + 24
    for acc in (0, 1):
        # reference: the separate launches
        dz0 = torch.zeros(M, Cout, dtype=torch.bfloat16, device=dev)
        dg0, db0, dg0b, db0b = (torch.zeros(Cout, device=dev) for _ in range(4))
        p2 = BnBwdSplit()
        if split:
            p2.split, p2.gamma2, p2.dgamma2, p2.dbeta2 = Ca, gamma2.data_ptr(), dg0b.data_ptr(), db0b.data_ptr()
        call("plyolo_bn_act_bwd_dz", dt, M, Cout, dout.data_ptr(), d_ld, z.data_ptr(), Cout, coef.data_ptr(), bslots.data_ptr(), gamma.data_ptr(),
             dg0.data_ptr(), db0.data_ptr(), 0, a, dz0.data_ptr(), Cout, C.byref(sp) if split else None, C.byref(p2) if split else None, hu.stream())
        dx0 = base.clone()
        call("plyolo_conv2d_dgrad", C.byref(d), dz0.data_ptr(), pk.wpd.data_ptr(), dx0.data_ptr(), acc, hu.stream())
        # fused
        dz1 = torch.full((M, Cout), 7.0, dtype=torch.bfloat16, device=dev)
        fw1 = torch.full((M, fwd_ld), 5.0, dtype=torch.bfloat16, device=dev)
        dg1, db1, dg1b, db1b = (torch.zeros(Cout, device=dev) for _ in range(4))
        f = BnBwdFuse()
        f.dout, f.dout_ld, f.z, f.z_ld, f.coef, f.bslots = dout.data_ptr(), d_ld, z.data_ptr(), Cout, coef.data_ptr(), bslots.data_ptr()
        if split:
            f.dout2, f.dout2_ld, f.dout_split = dout2.data_ptr(), dout2.shape[1], Ca
            f.par_split, f.gamma2, f.dgamma2, f.dbeta2 = Ca, gamma2.data_ptr(), dg1b.data_ptr(), db1b.data_ptr()
        else:
            f.fwd_to, f.fwd_ld = fw1.data_ptr(), fwd_ld
        f.gamma, f.dgamma, f.dbeta, f.act, f.dz, f.dz_ld = gamma.data_ptr(), dg1.data_ptr(), db1.data_ptr(), a, dz1.data_ptr(), Cout
        dx1 = base.clone()
        call("plyolo_conv2d_dgrad_bn", C.byref(d), C.byref(f), pk.wpd.data_ptr(), dx1.data_ptr(), acc, hu.stream())
        torch.cuda.synchronize()
        assert torch.equal(dz0.view(torch.int16), dz1.view(torch.int16)), "dz differs"
        assert torch.equal(dx0[:, :Cin].view(torch.int16), dx1[:, :Cin].view(torch.int16)), "dx differs"
        assert torch.equal(dx1[:, Cin:], base[:, Cin:]), "pad columns of dx touched"
        for u, v in ((dg0, dg1), (db0, db1), (dg0b, dg1b), (db0b, db1b)):
            assert torch.equal(u, v)
        assert float(dz1.float().abs().max()) > 0
        if not split:
            assert torch.equal(fw1[:, :Cout].view(torch.int16), dout[:, :Cout].view(torch.int16)), "forwarded shortcut gradient differs"
            assert torch.all(fw1[:, Cout:].float() == 5.0)
        # no dz reader: nothing is written there
        f.dz, f.dz_ld, f.fwd_to, f.fwd_ld = None, 0, None, 0
        dx2 = base.clone()
        call("plyolo_conv2d_dgrad_bn", C.byref(d), C.byref(f), pk.wpd.data_ptr(), dx2.data_ptr(), acc, hu.stream())
        # ... and with the reduction of the unit that produced x folded in (two segments with a gap when Cin allows)
        f.dz, f.dz_ld = dz1.data_ptr(), Cout
        zu = (torch.randn(M, Cin, device=dev) * 1.5).to(torch.bfloat16)
        cu = torch.cat([torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.2, torch.randn(Cin, device=dev) * 0.1, torch.rand(Cin, device=dev) + 0.5]).contiguous()
        su = torch.zeros(STAT_SLOTS * 2 * Cin, dtype=torch.float64, device=dev)
        red = BnRed()
        red.n = 1
        sg = red.seg[0]
        sg.c0, sg.c1, sg.z, sg.z_ld, sg.coef, sg.coef_ld, sg.bslots, sg.slot_ld, sg.act = 0, Cin, zu.data_ptr(), Cin, cu.data_ptr(), Cin, su.data_ptr(), Cin, ACT["silu"]
        dx3 = base.clone()
        call("plyolo_conv2d_dgrad_bn_red", C.byref(d), C.byref(f), pk.wpd.data_ptr(), dx3.data_ptr(), acc, C.byref(red), hu.stream())
        ref = torch.zeros(STAT_SLOTS * 2 * Cin, dtype=torch.float64, device=dev)
        dxs = dx3[:, :Cin].contiguous()
        call("plyolo_bn_act_bwd_reduce", dt, M, Cin, dxs.data_ptr(), Cin, zu.data_ptr(), Cin, cu.data_ptr(), ACT["silu"], ref.data_ptr(), None, hu.stream())
        torch.cuda.synchronize()
        assert torch.equal(dx2.view(torch.int16), dx1.view(torch.int16)), "dx differs without a dz reader"
        assert torch.equal(dx3.view(torch.int16), dx1.view(torch.int16)), "dx differs with the folded reduction"
        got, want = su.view(STAT_SLOTS, 2, Cin).sum(0), ref.view(STAT_SLOTS, 2, Cin).sum(0)
        assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max())


# (N, H, W, Cin, Cout, split): pointwise units of the shapes plyolo_conv2d_bwd_pw instantiates; 3x20x20 and 1x13x9 end in a ragged pixel tile
PWBWD_SHAPES = [(2, 40, 40, 128, 128, 0), (3, 20, 20, 64, 64, 0), (2, 40, 40, 64, 64, 32), (1, 13, 9, 32, 32, 8), (2, 24, 24, 64, 128, 0),
                (2, 24, 24, 128, 64, 32), (1, 40, 40, 32, 64, 0), (1, 40, 40, 64, 32, 0)]


@pytest.mark.parametrize("shape", PWBWD_SHAPES, ids=str)
@pytest.mark.parametrize("act", ["silu", "lrelu", None])
@pytest.mark.parametrize("grid", [0, 3])
def test_pointwise_unit_backward_in_one_launch(shape, act, grid, monkeypatch):
    """plyolo_conv2d_bwd_pw == plyolo_bn_act_bwd_dz + plyolo_conv2d_dgrad + plyolo_conv2d_wgrad: dx, dgamma, dbeta bit for bit (same dz
    bits, same MFMA order), the weight gradient up to the order of its fp32 sums and against torch's fp32 matmul of the SAME dz;
    merged pairs (output gradient in two matrices, two BatchNorm parameter sets), accumulation into dx, ragged last tile, a grid of
    three persistent workgroups (many tiles per workgroup) and the planned one."""
    from pl_yolo_amd._lib import ACT, BnBwdFuse, Split, BnBwdSplit, STAT_SLOTS
    monkeypatch.setenv("PLYOLO_PWBWD_MIN_MB", "0")
    if grid:
        monkeypatch.setenv("PLYOLO_PWBWD_G", str(grid))
    N, H, W, Cin, Cout, split = shape
    dt, M = BF16, N * H * W
    torch.manual_seed(sum(shape) + 5)
    dev = hu.DEV
    w = hu.rnd_bf16(torch.randn(Cout, Cin, 1, 1, device=dev) / Cout ** 0.5)
    pk = hu.Packed(w, dt)
    z = (torch.randn(M, Cout, device=dev) * 1.5).to(torch.bfloat16)
    Ca = split if split else Cout
    d_ld = Ca + 8
    dout = torch.randn(M, d_ld, device=dev).to(torch.bfloat16)
    dout2 = torch.randn(M, Cout - Ca + 16, device=dev).to(torch.bfloat16) if split else None
    gamma, gamma2 = torch.rand(Cout, device=dev) + 0.5, torch.rand(Cout, device=dev) + 0.5
    mean, invstd = torch.randn(Cout, device=dev) * 0.1, torch.rand(Cout, device=dev) + 0.5
    beta = torch.randn(Cout, device=dev) * 0.1
    g_eff = torch.cat([gamma[:Ca], gamma2[:Cout - Ca]]) if split else gamma
    scale = g_eff * invstd
    coef = torch.cat([scale, beta - mean * scale, mean, invstd]).contiguous()
    bslots = torch.zeros(STAT_SLOTS * 2 * Cout, dtype=torch.float64, device=dev)
    sp = Split()
    if split:
        sp.split, sp.p2, sp.ld2 = Ca, dout2.data_ptr(), dout2.shape[1]
    a = ACT[act]
    call("plyolo_bn_act_bwd_reduce", dt, M, Cout, dout.data_ptr(), d_ld, z.data_ptr(), Cout, coef.data_ptr(), a, bslots.data_ptr(),
         C.byref(sp) if split else None, hu.stream())
    x_ld = Cin + 8
    xm = torch.randn(M, x_ld, device=dev).to(torch.bfloat16)
    d = hu.conv_desc(dt, N, H, W, Cin, Cout, 1, 1, x_ld, Cout)
    lib = hu._lib.lib()
    assert lib.plyolo_conv2d_bwd_pw_fits(C.byref(d), a) == 1
    base = torch.randn(M, x_ld, device=dev).to(torch.bfloat16)
    for acc in (0, 1):
        # reference: the three separate launches
        dz0 = torch.zeros(M, Cout, dtype=torch.bfloat16, device=dev)
        dg0, db0, dg0b, db0b = (torch.zeros(Cout, device=dev) for _ in range(4))
        p2 = BnBwdSplit()
        if split:
            p2.split, p2.gamma2, p2.dgamma2, p2.dbeta2 = Ca, gamma2.data_ptr(), dg0b.data_ptr(), db0b.data_ptr()
        call("plyolo_bn_act_bwd_dz", dt, M, Cout, dout.data_ptr(), d_ld, z.data_ptr(), Cout, coef.data_ptr(), bslots.data_ptr(), gamma.data_ptr(),
             dg0.data_ptr(), db0.data_ptr(), 0, a, dz0.data_ptr(), Cout, C.byref(sp) if split else None, C.byref(p2) if split else None, hu.stream())
        dx0 = base.clone()
        call("plyolo_conv2d_dgrad", C.byref(d), dz0.data_ptr(), pk.wpd.data_ptr(), dx0.data_ptr(), acc, hu.stream())
        pk0 = hu.Packed(w, dt)
        pk0.set_slabs(d)
        call("plyolo_conv2d_wgrad", C.byref(d), xm.data_ptr(), dz0.data_ptr(), pk0.dwp.data_ptr(), hu.stream())
        dw0 = pk0.unpack().clone()
        # one launch
        ns = lib.plyolo_conv2d_bwd_pw_slabs(C.byref(d))
        assert 1 <= ns <= (grid or 256)
        pk1 = hu.Packed(w, dt, nslab=ns)
        pk1.entry.nslab = ns
        pk1.table = torch.frombuffer(bytearray(bytes((type(pk1.entry) * 1)(pk1.entry))), dtype=torch.uint8).to(dev)
        pk1.dwp.fill_(float("nan"))       # every slab element must be written
        dg1, db1, dg1b, db1b = (torch.zeros(Cout, device=dev) for _ in range(4))
        f = BnBwdFuse()
        f.dout, f.dout_ld, f.z, f.z_ld, f.coef, f.bslots = dout.data_ptr(), d_ld, z.data_ptr(), Cout, coef.data_ptr(), bslots.data_ptr()
        if split:
            f.dout2, f.dout2_ld, f.dout_split = dout2.data_ptr(), dout2.shape[1], Ca
            f.par_split, f.gamma2, f.dgamma2, f.dbeta2 = Ca, gamma2.data_ptr(), dg1b.data_ptr(), db1b.data_ptr()
        f.gamma, f.dgamma, f.dbeta, f.act = gamma.data_ptr(), dg1.data_ptr(), db1.data_ptr(), a
        dx1 = base.clone()
        call("plyolo_conv2d_bwd_pw", C.byref(d), C.byref(f), xm.data_ptr(), pk.wpd.data_ptr(), dx1.data_ptr(), acc, pk1.dwp.data_ptr(), hu.stream())
        dw1 = pk1.unpack().clone()
        torch.cuda.synchronize()
        assert torch.equal(dx0[:, :Cin].view(torch.int16), dx1[:, :Cin].view(torch.int16)), "dx differs"
        assert torch.equal(dx1[:, Cin:], base[:, Cin:]), "pad columns of dx touched"
        for u, v in ((dg0, dg1), (db0, db1), (dg0b, dg1b), (db0b, db1b)):
            assert torch.equal(u, v)
        ref = (dz0.float().t() @ xm[:, :Cin].float()).view(Cout, Cin, 1, 1)
        e0, e1 = hu.relerr(dw0, ref), hu.relerr(dw1, ref)
        print("pw_bwd", shape, act, "acc", acc, "slabs", ns, "dW relerr separate %.3g fused %.3g" % (e0, e1))
        assert torch.isfinite(dw1).all() and e1 <= 1e-4 and hu.relerr(dw1, dw0) <= 1e-4
        assert float(dx1.float().abs().max()) > 0
        # ... and with the BatchNorm-backward reduction of the unit that produced x folded into the same launch
        from pl_yolo_amd._lib import BnRed
        zu = (torch.randn(M, Cin, device=dev) * 1.5).to(torch.bfloat16)
        cu = torch.cat([torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.2, torch.randn(Cin, device=dev) * 0.1, torch.rand(Cin, device=dev) + 0.5]).contiguous()
        su = torch.zeros(STAT_SLOTS * 2 * Cin, dtype=torch.float64, device=dev)
        red = BnRed()
        red.n = 1
        sg = red.seg[0]
        sg.c0, sg.c1, sg.z, sg.z_ld, sg.coef, sg.coef_ld, sg.bslots, sg.slot_ld, sg.act = 0, Cin, zu.data_ptr(), Cin, cu.data_ptr(), Cin, su.data_ptr(), Cin, ACT["silu"]
        dx2 = base.clone()
        pk1.dwp.fill_(float("nan"))
        call("plyolo_conv2d_bwd_pw_red", C.byref(d), C.byref(f), xm.data_ptr(), pk.wpd.data_ptr(), dx2.data_ptr(), acc, pk1.dwp.data_ptr(), C.byref(red), hu.stream())
        dw2 = pk1.unpack().clone()
        ref = torch.zeros(STAT_SLOTS * 2 * Cin, dtype=torch.float64, device=dev)
        dxs = dx2[:, :Cin].contiguous()
        call("plyolo_bn_act_bwd_reduce", dt, M, Cin, dxs.data_ptr(), Cin, zu.data_ptr(), Cin, cu.data_ptr(), ACT["silu"], ref.data_ptr(), None, hu.stream())
        torch.cuda.synchronize()
        assert torch.equal(dx2.view(torch.int16), dx1.view(torch.int16)) and torch.equal(dw2, dw1)
        got, want = su.view(STAT_SLOTS, 2, Cin).sum(0), ref.view(STAT_SLOTS, 2, Cin).sum(0)
        assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max())


# (N, H, W, Cin, Cout, k, s, segments of dx channels): data gradients whose store loop folds the BatchNorm-backward reduction of the
# upstream unit(s); two segments = a concatenated input (one BatchNorm unit per part), a gap = a part without a BatchNorm unit
@pytest.mark.parametrize("shape", [(2, 48, 40, 16, 32), (3, 37, 45, 24, 64), (1, 160, 160, 16, 32), (2, 21, 19, 8, 24), (1, 64, 64, 32, 40)], ids=str)
def test_weight_gradient_with_bn_backward_in_its_loader(shape):
    """plyolo_conv2d_wgrad_bn == plyolo_bn_act_bwd_dz + plyolo_conv2d_wgrad (a unit without a data gradient: its dz is read by the weight
    gradient only): the folded weight gradient, dgamma and dbeta bit for bit -- the loader forms the same bf16 dz values the separate
    pass would have written, absent pixels of ragged tiles stage zeros -- both slab shapes, odd maps, channel tails."""
    from pl_yolo_amd._lib import ACT, BnBwdFuse, STAT_SLOTS
    N, H, W, Cin, Cout = shape
    dt, a = BF16, ACT["silu"]
    torch.manual_seed(sum(shape) + 3)
    dev = hu.DEV
    M = N * H * W
    x_ld, d_ld = (Cin + 7) // 8 * 8, (Cout + 7) // 8 * 8 + 8
    w = hu.rnd_bf16(torch.randn(Cout, Cin, 3, 3, device=dev) / (Cin * 9) ** 0.5)
    xm = torch.zeros(M, x_ld, device=dev)
    xm[:, :Cin] = torch.randn(M, Cin, device=dev)
    xm = xm.to(torch.bfloat16)
    z = (torch.randn(M, Cout, device=dev) * 1.5).to(torch.bfloat16)
    dout = torch.zeros(M, d_ld, device=dev)
    dout[:, :Cout] = torch.randn(M, Cout, device=dev)
    dout = dout.to(torch.bfloat16)
    mean, var = z.float().mean(0), z.float().var(0, unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-3)
    gamma = (torch.rand(Cout, device=dev) + 0.5).contiguous()
    beta = (torch.randn(Cout, device=dev) * 0.2).contiguous()
    coef = torch.cat([gamma * invstd, beta - mean * gamma * invstd, mean, invstd]).contiguous()
    d = hu.conv_desc(dt, N, H, W, Cin, Cout, 3, 1, x_ld, Cout)
    assert hu._lib.lib().plyolo_conv2d_wgrad_bn_fits(C.byref(d), a) == 1
    bslots = torch.zeros(STAT_SLOTS * 2 * Cout, dtype=torch.float64, device=dev)
    call("plyolo_bn_act_bwd_reduce", dt, M, Cout, dout.data_ptr(), d_ld, z.data_ptr(), Cout, coef.data_ptr(), a, bslots.data_ptr(), None, hu.stream())
    # separate launches
    dz0 = torch.empty(M, Cout, dtype=torch.bfloat16, device=dev)
    dg0, db0 = torch.zeros(Cout, device=dev), torch.zeros(Cout, device=dev)
    call("plyolo_bn_act_bwd_dz", dt, M, Cout, dout.data_ptr(), d_ld, z.data_ptr(), Cout, coef.data_ptr(), bslots.data_ptr(), gamma.data_ptr(),
         dg0.data_ptr(), db0.data_ptr(), 0, a, dz0.data_ptr(), Cout, None, None, hu.stream())
    pk0 = hu.Packed(w, dt)
    pk0.set_slabs(d)
    call("plyolo_conv2d_wgrad", C.byref(d), xm.data_ptr(), dz0.data_ptr(), pk0.dwp.data_ptr(), hu.stream())
    dw0 = pk0.unpack().clone()
    # one launch
    dg1, db1 = torch.full((Cout,), 7.0, device=dev), torch.full((Cout,), 7.0, device=dev)
    f = BnBwdFuse()
    f.dout, f.dout_ld, f.z, f.z_ld, f.coef, f.bslots = dout.data_ptr(), d_ld, z.data_ptr(), Cout, coef.data_ptr(), bslots.data_ptr()
    f.gamma, f.dgamma, f.dbeta, f.act = gamma.data_ptr(), dg1.data_ptr(), db1.data_ptr(), a
    pk1 = hu.Packed(w, dt)
    pk1.set_slabs(d)
    call("plyolo_conv2d_wgrad_bn", C.byref(d), C.byref(f), xm.data_ptr(), pk1.dwp.data_ptr(), hu.stream())
    dw1 = pk1.unpack().clone()
    torch.cuda.synchronize()
    assert float(dw0.abs().max()) > 0
    assert torch.equal(dw0, dw1), "weight gradient differs: max %.3e" % float((dw0 - dw1).abs().max())
    assert torch.equal(dg0, dg1) and torch.equal(db0, db1)


@pytest.mark.parametrize("shape", [(2, 32, 32, 64, 128, 3, 2), (1, 64, 64, 32, 64, 3, 2), (2, 26, 38, 128, 256, 3, 2), (3, 17, 23, 24, 40, 3, 2),
                                   (2, 21, 30, 96, 72, 3, 2), (1, 160, 160, 32, 64, 3, 2)], ids=str)
@pytest.mark.parametrize("acc", [0, 1])
def test_stride2_dgrad_fused_parity_classes_equal_the_four_job_launch(shape, acc, monkeypatch):
    """conv_s2d.hip (the four parity classes of a 3x3 stride-2 data gradient on one staged dZ tile per workgroup; default for at most
    64 input channels, forced here for every width) against the four-job launch of conv_mfma.hip: the same taps in the same order into
    every accumulator -- dx bit for bit, odd maps, channel tails, accumulation into dx."""
    N, H, W, Cin, Cout, k, st = shape
    dt = BF16
    torch.manual_seed(sum(shape) + 5)
    dev = hu.DEV
    OH, OW = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    Mi, Mo = N * H * W, N * OH * OW
    w = hu.rnd_bf16(torch.randn(Cout, Cin, k, k, device=dev) / (Cout * k * k) ** 0.5)
    pk = hu.Packed(w, dt)
    y_ld, x_ld = (Cout + 7) // 8 * 8 + 8, (Cin + 7) // 8 * 8 + 8
    dy = torch.zeros(Mo, y_ld, device=dev)
    dy[:, :Cout] = torch.randn(Mo, Cout, device=dev)
    dy = dy.to(torch.bfloat16)
    d = hu.conv_desc(dt, N, H, W, Cin, Cout, k, st, x_ld, y_ld)
    base = torch.randn(Mi, x_ld, device=dev).to(torch.bfloat16)
    outs = []
    for s2d in ("0", "1"):
        monkeypatch.setenv("PLYOLO_S2D", s2d)
        monkeypatch.setenv("PLYOLO_S2D_MAXC", "4096")
        dx = base.clone()
        call("plyolo_conv2d_dgrad", C.byref(d), dy.data_ptr(), pk.wpd.data_ptr(), dx.data_ptr(), acc, hu.stream())
        torch.cuda.synchronize()
        outs.append(dx)
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
    assert float((outs[1][:, :Cin].float() - (base[:, :Cin].float() if acc else 0)).abs().max()) > 0
    assert torch.equal(outs[1][:, Cin:].view(torch.int16), base[:, Cin:].view(torch.int16))      # the pitch padding is never written


RED_SHAPES = [(2, 20, 20, 128, 128, 3, 1, [(0, 128)]), (3, 24, 18, 64, 64, 3, 1, [(0, 32), (32, 64)]), (1, 36, 40, 32, 32, 3, 1, [(0, 32)]),
              (2, 32, 32, 64, 128, 3, 2, [(0, 64)]), (1, 64, 64, 32, 64, 3, 2, [(0, 32)]), (2, 26, 38, 128, 256, 3, 2, [(0, 64), (64, 128)]),
              (2, 20, 20, 128, 128, 1, 1, [(0, 128)]), (3, 13, 9, 64, 96, 1, 1, [(8, 40)]), (2, 40, 40, 256, 80, 1, 1, [(0, 128), (128, 256)]),
              (4, 20, 20, 128, 16, 1, 1, [(0, 128)]),
              # row-flattened tiles (conv_mfma_flat.hip): 40- and 20-wide maps, dx blocks of 128 channels (a 192-channel dx = a block and a half)
              (2, 40, 40, 128, 128, 3, 1, [(0, 128)]), (1, 12, 40, 192, 64, 3, 1, [(0, 64), (64, 192)]), (3, 8, 20, 256, 96, 3, 1, [(0, 128), (128, 256)])]


@pytest.mark.parametrize("shape", RED_SHAPES, ids=str)
@pytest.mark.parametrize("acc", [0, 1])
def test_dgrad_folds_upstream_bn_reduction(shape, acc):
    """plyolo_conv2d_dgrad_red == plyolo_conv2d_dgrad (dx bit for bit) + plyolo_bn_act_bwd_reduce of the FINAL dx against the upstream
    unit's z, per segment: the sums land in the segment's own slots (other channels of the unit untouched), 3x3 stride 1 / 2 and
    pointwise kernels, ragged maps, accumulation into dx."""
    from pl_yolo_amd._lib import ACT, BnRed, STAT_SLOTS
    N, H, W, Cin, Cout, k, st, segs = shape
    dt = BF16
    torch.manual_seed(sum(shape[:7]) + 11)
    dev = hu.DEV
    pad = (k - 1) // 2
    OH, OW = (H + 2 * pad - k) // st + 1, (W + 2 * pad - k) // st + 1
    Mi, Mo = N * H * W, N * OH * OW
    w = hu.rnd_bf16(torch.randn(Cout, Cin, k, k, device=dev) / (Cout * k * k) ** 0.5)
    pk = hu.Packed(w, dt)
    y_ld, x_ld = (Cout + 7) // 8 * 8 + 8, Cin + 8
    dy = torch.zeros(Mo, y_ld, device=dev)
    dy[:, :Cout] = torch.randn(Mo, Cout, device=dev)
    dy = dy.to(torch.bfloat16)
    d = hu.conv_desc(dt, N, H, W, Cin, Cout, k, st, x_ld, y_ld)
    lib = hu._lib.lib()
    assert lib.plyolo_conv2d_dgrad_red_fits(C.byref(d)) == 1
    base = torch.randn(Mi, x_ld, device=dev).to(torch.bfloat16)
    dx0 = base.clone()
    call("plyolo_conv2d_dgrad", C.byref(d), dy.data_ptr(), pk.wpd.data_ptr(), dx0.data_ptr(), acc, hu.stream())
    red = BnRed()
    red.n = len(segs)
    keep = []
    for i, (c0, c1) in enumerate(segs):
        Cu, off = (c1 - c0) + 16, 8                # the unit has more channels than this segment: [off, off + c1 - c0) of them
        z = (torch.randn(Mi, Cu, device=dev) * 1.5).to(torch.bfloat16)
        coef = torch.cat([torch.rand(Cu, device=dev) + 0.5, torch.randn(Cu, device=dev) * 0.2, torch.randn(Cu, device=dev) * 0.1, torch.rand(Cu, device=dev) + 0.5]).contiguous()
        slots = torch.zeros(STAT_SLOTS * 2 * Cu, dtype=torch.float64, device=dev)
        act = ACT["silu" if i == 0 else "lrelu"]
        sg = red.seg[i]
        sg.c0, sg.c1, sg.z, sg.z_ld = c0, c1, z.data_ptr() + off * 2, Cu
        sg.coef, sg.coef_ld, sg.bslots, sg.slot_ld, sg.act = coef.data_ptr() + off * 4, Cu, slots.data_ptr() + off * 8, Cu, act
        keep.append((z, coef, slots, act, Cu, off, c0, c1))
    dx1 = base.clone()
    call("plyolo_conv2d_dgrad_red", C.byref(d), dy.data_ptr(), pk.wpd.data_ptr(), dx1.data_ptr(), acc, C.byref(red), hu.stream())
    torch.cuda.synchronize()
    assert torch.equal(dx0.view(torch.int16), dx1.view(torch.int16)), "dx differs"
    for (z, coef, slots, act, Cu, off, c0, c1) in keep:
        n = c1 - c0
        ref = torch.zeros(STAT_SLOTS * 2 * n, dtype=torch.float64, device=dev)
        dxs = dx1[:, c0:c1].contiguous()
        zs = z[:, off:off + n].contiguous()
        cf = coef.view(4, Cu)[:, off:off + n].contiguous()
        call("plyolo_bn_act_bwd_reduce", dt, Mi, n, dxs.data_ptr(), n, zs.data_ptr(), n, cf.data_ptr(), act, ref.data_ptr(), None, hu.stream())
        torch.cuda.synchronize()
        got = slots.view(STAT_SLOTS, 2, Cu).sum(0)
        want = ref.view(STAT_SLOTS, 2, n).sum(0)
        assert float(got[:, :off].abs().max()) == 0.0 and float(got[:, off + n:].abs().max()) == 0.0, "slots outside the segment touched"
        err = float((got[:, off:off + n] - want).abs().max()) / max(float(want.abs().max()), 1e-9)
        print("dgrad_red", shape[:7], (c0, c1), "acc", acc, "slot sums rel err %.2e" % err)
        assert err <= 2e-5 and float(want.abs().max()) > 0
