"""-m gpu: BatchNorm/activation, data-movement and optimizer kernels vs PyTorch fp32."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from pl_yolo_amd._lib import BF16, F32, call  # noqa: E402
import hiputil as hu  # noqa: E402

DTS = [BF16, F32]
IDS = ["bf16", "fp32"]


def _tol(dt, f32=2e-5, bf=2.0 ** -7):
    return bf if dt == BF16 else f32


@pytest.mark.parametrize("dt", DTS, ids=IDS)
@pytest.mark.parametrize("act", ["silu", "relu", "lrelu", None])
def test_bn_act_fwd_bwd(dt, act):
    """conv-epilogue partials -> bn_finalize -> bn_act_fwd (+residual), then the
    three-kernel backward, against F.batch_norm + activation under autograd."""
    N, Cc, H, W = 3, 48, 14, 18
    torch.manual_seed(5)
    z = torch.randn(N, Cc, H, W, device=hu.DEV) * 2 + 0.5
    res = torch.randn(N, Cc, H, W, device=hu.DEV)
    if dt == BF16:
        z, res = hu.rnd_bf16(z), hu.rnd_bf16(res)
    z.requires_grad_(True)
    gamma = (torch.rand(Cc, device=hu.DEV) + 0.5).requires_grad_(True)
    beta = (torch.rand(Cc, device=hu.DEV) - 0.5).requires_grad_(True)
    rm, rv = torch.zeros(Cc, device=hu.DEV), torch.ones(Cc, device=hu.DEV)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    u = F.batch_norm(z, rm_ref, rv_ref, gamma, beta, True, 0.03, 1e-3)
    a = {"silu": F.silu, "relu": F.relu, "lrelu": lambda t: F.leaky_relu(t, 0.1), None: lambda t: t}[act](u)
    out_ref = a + res
    dout = torch.randn_like(out_ref)
    if dt == BF16:
        dout = hu.rnd_bf16(dout)
    gz, gg, gb = torch.autograd.grad(out_ref, (z, gamma, beta), dout)

    M = N * H * W
    zm = hu.to_nhwc(z.detach(), dt, Cc)
    resm = hu.to_nhwc(res, dt, Cc + 8)
    # statistics as the conv epilogue leaves them: partial sums spread over the fp64 stat slots
    zz = zm.double()
    S = hu._lib.STAT_SLOTS
    stats = torch.zeros(S, 2, Cc, dtype=torch.float64, device=hu.DEV)
    for r, chunk in enumerate(torch.chunk(zz, 3 * S, 0)):
        stats[r % S, 0] += chunk.sum(0)
        stats[r % S, 1] += (chunk * chunk).sum(0)
    coef = torch.zeros(4 * Cc, device=hu.DEV)
    coef2 = torch.zeros(4 * Cc, device=hu.DEV)
    nbt = torch.zeros(1, dtype=torch.int64, device=hu.DEV)
    st = hu._lib.BnStats()
    st.slots, st.count, st.gamma, st.beta = stats.data_ptr(), float(M), gamma.data_ptr(), beta.data_ptr()
    st.eps, st.momentum = 1e-3, 0.03
    # standalone finalize (no running statistics) ...
    call("plyolo_bn_finalize", C.byref(st), Cc, coef2.data_ptr(), hu.stream())
    # ... and the fused form: bn_act_fwd derives scale/shift itself and publishes coef + running statistics
    st.running_mean, st.running_var, st.num_batches_tracked = rm.data_ptr(), rv.data_ptr(), nbt.data_ptr()
    out = torch.full((M, Cc + 16), 2.0, dtype=hu.tdtype(dt), device=hu.DEV)
    call("plyolo_bn_act_fwd", dt, M, Cc, zm.data_ptr(), Cc, coef.data_ptr(), hu._lib.ACT[act], resm.data_ptr(), Cc + 8,
         out.data_ptr(), Cc + 16, C.byref(st), None, hu.stream())
    torch.cuda.synchronize()
    assert torch.equal(coef, coef2)
    out_b = torch.full((M, Cc + 16), 2.0, dtype=hu.tdtype(dt), device=hu.DEV)   # coef as an input gives the same bytes
    call("plyolo_bn_act_fwd", dt, M, Cc, zm.data_ptr(), Cc, coef.data_ptr(), hu._lib.ACT[act], resm.data_ptr(), Cc + 8,
         out_b.data_ptr(), Cc + 16, None, None, hu.stream())
    torch.cuda.synchronize()
    assert torch.equal(out, out_b)
    torch.cuda.synchronize()
    assert int(nbt) == 1
    assert hu.relerr(rm, rm_ref) < 1e-5 and hu.relerr(rv, rv_ref) < 1e-5
    e = hu.relerr(hu.from_nhwc(out, N, H, W, Cc), out_ref)
    print("bn_act_fwd", act, "relerr %.3g" % e)
    assert e <= _tol(dt, 1e-5)
    assert torch.all(out[:, Cc:].float() == 2.0)
    # backward
    dm = hu.to_nhwc(dout, dt, Cc + 8)
    bslots = torch.zeros(hu._lib.STAT_SLOTS, 2, Cc, dtype=torch.float64, device=hu.DEV)
    dg, db = torch.zeros(Cc, device=hu.DEV), torch.zeros(Cc, device=hu.DEV)
    dz = torch.zeros(M, Cc, dtype=hu.tdtype(dt), device=hu.DEV)
    call("plyolo_bn_act_bwd_reduce", dt, M, Cc, dm.data_ptr(), Cc + 8, zm.data_ptr(), Cc, coef.data_ptr(), hu._lib.ACT[act], bslots.data_ptr(), None, hu.stream())
    call("plyolo_bn_act_bwd_dz", dt, M, Cc, dm.data_ptr(), Cc + 8, zm.data_ptr(), Cc, coef.data_ptr(), bslots.data_ptr(), gamma.data_ptr(),
         dg.data_ptr(), db.data_ptr(), 0, hu._lib.ACT[act], dz.data_ptr(), Cc, None, None, hu.stream())
    torch.cuda.synchronize()
    e1, e2, e3 = hu.relerr(hu.from_nhwc(dz, N, H, W, Cc), gz), hu.relerr(dg, gg), hu.relerr(db, gb)
    print("bn_act_bwd", act, "dz %.3g dgamma %.3g dbeta %.3g" % (e1, e2, e3))
    assert e1 <= _tol(dt, 5e-5) and e2 <= 1e-4 and e3 <= 1e-4
    # the shortcut's share of dout forwarded by the same pass (plyolo_split.fwd_to) == a plyolo_copy_add launch, copy and accumulate
    from pl_yolo_amd._lib import Split
    for acc in (0, 1):
        base = torch.randn(M, Cc + 16, device=hu.DEV).to(hu.tdtype(dt))
        want = base.clone()
        call("plyolo_copy_add", dt, M, Cc, dm.data_ptr(), Cc + 8, want.data_ptr(), Cc + 16, acc, hu.stream())
        got, dz2 = base.clone(), torch.zeros_like(dz)
        sp = Split()
        sp.fwd_to, sp.fwd_ld, sp.fwd_acc = got.data_ptr(), Cc + 16, acc
        call("plyolo_bn_act_bwd_dz", dt, M, Cc, dm.data_ptr(), Cc + 8, zm.data_ptr(), Cc, coef.data_ptr(), bslots.data_ptr(), gamma.data_ptr(),
             dg.data_ptr(), db.data_ptr(), 0, hu._lib.ACT[act], dz2.data_ptr(), Cc, C.byref(sp), None, hu.stream())
        torch.cuda.synchronize()
        assert torch.equal(got, want) and torch.equal(dz2, dz)


@pytest.mark.parametrize("dt", DTS, ids=IDS)
def test_bn_eval_coef(dt):
    Cc = 40
    torch.manual_seed(1)
    g, b = torch.rand(Cc, device=hu.DEV) + 0.5, torch.rand(Cc, device=hu.DEV)
    rm, rv = torch.randn(Cc, device=hu.DEV), torch.rand(Cc, device=hu.DEV) + 0.1
    coef = torch.zeros(4 * Cc, device=hu.DEV)
    call("plyolo_bn_eval_coef", Cc, g.data_ptr(), b.data_ptr(), rm.data_ptr(), rv.data_ptr(), 1e-3, coef.data_ptr(), hu.stream())
    x = torch.randn(2, Cc, 4, 4, device=hu.DEV)
    ref = F.batch_norm(x, rm, rv, g, b, False, 0.03, 1e-3)
    got = x * coef[:Cc].view(1, -1, 1, 1) + coef[Cc:2 * Cc].view(1, -1, 1, 1)
    assert hu.relerr(got, ref) < 1e-5


@pytest.mark.parametrize("dt", DTS, ids=IDS)
def test_focus(dt):
    N, H, W = 2, 32, 48
    torch.manual_seed(2)
    img = torch.rand(N, 3, H, W, device=hu.DEV) * 255
    ref = torch.cat((img[..., ::2, ::2], img[..., 1::2, ::2], img[..., ::2, 1::2], img[..., 1::2, 1::2]), 1)
    cp = 16 if dt == BF16 else 12
    out = torch.full((N * (H // 2) * (W // 2), cp), 9.0, dtype=hu.tdtype(dt), device=hu.DEV)
    call("plyolo_focus_s2d", dt, img.data_ptr(), N, H, W, out.data_ptr(), cp, hu.stream())
    torch.cuda.synchronize()
    got = hu.from_nhwc(out, N, H // 2, W // 2, 12)
    want = hu.rnd_bf16(ref) if dt == BF16 else ref
    assert torch.equal(got, want)
    if cp > 12:
        assert torch.all(out[:, 12:].float() == 0)


@pytest.mark.parametrize("dt", DTS, ids=IDS)
def test_copy_upsample_pool(dt):
    N, Cc, H, W = 2, 24, 10, 12
    torch.manual_seed(3)
    x = torch.randn(N, Cc, H, W, device=hu.DEV)
    if dt == BF16:
        x = hu.rnd_bf16(x)
    xm = hu.to_nhwc(x, dt, Cc + 8)
    # copy / accumulate / zero-fill
    dst = hu.to_nhwc(torch.ones_like(x), dt, Cc + 16)
    call("plyolo_copy_add", dt, N * H * W, Cc, xm.data_ptr(), Cc + 8, dst.data_ptr(), Cc + 16, 1, hu.stream())
    assert hu.relerr(hu.from_nhwc(dst, N, H, W, Cc), x + 1) <= _tol(dt, 1e-6)
    call("plyolo_copy_add", dt, N * H * W, Cc, None, 0, dst.data_ptr(), Cc + 16, 0, hu.stream())
    assert float(dst[:, :Cc].float().abs().max()) == 0.0 and torch.all(dst[:, Cc:].float() == 7.0)
    # upsample fwd / bwd
    up = torch.zeros(N * 4 * H * W, Cc, dtype=hu.tdtype(dt), device=hu.DEV)
    call("plyolo_upsample2x_fwd", dt, N, H, W, Cc, xm.data_ptr(), Cc + 8, up.data_ptr(), Cc, hu.stream())
    assert torch.equal(hu.from_nhwc(up, N, 2 * H, 2 * W, Cc), F.interpolate(x, scale_factor=2, mode="nearest"))
    g = torch.randn(N, Cc, 2 * H, 2 * W, device=hu.DEV)
    if dt == BF16:
        g = hu.rnd_bf16(g)
    gm = hu.to_nhwc(g, dt, Cc)
    din = hu.to_nhwc(torch.ones_like(x), dt, Cc)
    call("plyolo_upsample2x_bwd", dt, N, H, W, Cc, gm.data_ptr(), Cc, din.data_ptr(), Cc, 1, hu.stream())
    want = 1 + g.reshape(N, Cc, H, 2, W, 2).sum((3, 5))
    assert hu.relerr(hu.from_nhwc(din, N, H, W, Cc), want) <= _tol(dt, 1e-6, 2.0 ** -6)
    # max pools: forward values and ATen-style first-max gradient routing
    for k in (5, 9, 13):
        out = torch.zeros(N * H * W, Cc, dtype=hu.tdtype(dt), device=hu.DEV)
        call("plyolo_maxpool_s1_fwd", dt, N, H, W, Cc, k, xm.data_ptr(), Cc + 8, out.data_ptr(), Cc, hu.stream())
        assert torch.equal(hu.from_nhwc(out, N, H, W, Cc), F.max_pool2d(x, k, 1, k // 2))
        xr = x.clone().requires_grad_(True)
        y = F.max_pool2d(xr, k, 1, k // 2)
        gy = torch.randn_like(y)
        if dt == BF16:
            gy = hu.rnd_bf16(gy)
        (gx,) = torch.autograd.grad(y, xr, gy)
        gym = hu.to_nhwc(gy, dt, Cc)
        acc = torch.zeros(N * H * W, Cc, device=hu.DEV)
        call("plyolo_maxpool_s1_bwd", dt, N, H, W, Cc, k, xm.data_ptr(), Cc + 8, gym.data_ptr(), Cc, acc.data_ptr(), hu.stream())
        torch.cuda.synchronize()
        assert hu.relerr(acc.reshape(N, H, W, Cc).permute(0, 3, 1, 2), gx) <= 1e-5, k
    # all three SPP pools routed in one launch (with ties: bf16 inputs repeat values), accumulate on/off
    import ctypes as C
    gys, want = [], torch.zeros_like(x)
    for k in (5, 9, 13):
        xr = x.clone().requires_grad_(True)
        y = F.max_pool2d(xr, k, 1, k // 2)
        gy = hu.rnd_bf16(torch.randn_like(y)) if dt == BF16 else torch.randn_like(y)
        want += torch.autograd.grad(y, xr, gy)[0]
        gys.append(hu.to_nhwc(gy, dt, Cc + 8))
    assert hu._lib.lib().plyolo_spp_pools_bwd_fits(dt, H, W) == 1
    for accumulate in (0, 1):
        din = hu.to_nhwc(torch.ones_like(x), dt, Cc)
        call("plyolo_spp_pools_bwd", dt, N, H, W, Cc, 3, (C.c_int * 3)(5, 9, 13), xm.data_ptr(), Cc + 8,
             (C.c_void_p * 3)(*[t.data_ptr() for t in gys]), (C.c_int * 3)(Cc + 8, Cc + 8, Cc + 8), din.data_ptr(), Cc,
             accumulate, hu.stream())
        torch.cuda.synchronize()
        assert hu.relerr(hu.from_nhwc(din, N, H, W, Cc), want + accumulate) <= _tol(dt, 1e-5, 2.0 ** -6), accumulate
    # cascade identity used by the SPP forward: pool9(x) == pool5(pool5(x))
    p5 = torch.zeros(N * H * W, Cc, dtype=hu.tdtype(dt), device=hu.DEV)
    p9 = torch.zeros_like(p5)
    call("plyolo_maxpool_s1_fwd", dt, N, H, W, Cc, 5, xm.data_ptr(), Cc + 8, p5.data_ptr(), Cc, hu.stream())
    call("plyolo_maxpool_s1_fwd", dt, N, H, W, Cc, 5, p5.data_ptr(), Cc, p9.data_ptr(), Cc, hu.stream())
    assert torch.equal(hu.from_nhwc(p9, N, H, W, Cc), F.max_pool2d(x, 9, 1, 4))
    # ... and the three pools of the SPP forward in ONE launch (plyolo_spp_pools_fwd): bit-identical to F.max_pool2d, pad columns untouched
    ks3 = (C.c_int * 3)(5, 9, 13)
    if hu._lib.lib().plyolo_spp_pools_fwd_fits(dt, H, W, Cc, 3, ks3) == 1:
        outs = [torch.full((N * H * W, Cc + 16), 3.0, dtype=hu.tdtype(dt), device=hu.DEV) for _ in range(3)]
        call("plyolo_spp_pools_fwd", dt, N, H, W, Cc, 3, ks3, xm.data_ptr(), Cc + 8, (C.c_void_p * 3)(*[t.data_ptr() for t in outs]),
             (C.c_int * 3)(Cc + 16, Cc + 16, Cc + 16), hu.stream())
        torch.cuda.synchronize()
        for k, o in zip((5, 9, 13), outs):
            assert torch.equal(hu.from_nhwc(o, N, H, W, Cc), F.max_pool2d(x, k, 1, k // 2)), k
            assert torch.all(o[:, Cc:].float() == 3.0)
    else:
        assert Cc % (8 if dt == BF16 else 4) != 0
    # f32 -> act conversion with accumulate, layout converters
    f = torch.randn(N * H * W, Cc, device=hu.DEV)
    tgt = hu.to_nhwc(x, dt, Cc + 8)
    call("plyolo_f32_to_act", dt, N * H * W, Cc, f.data_ptr(), tgt.data_ptr(), Cc + 8, 1, hu.stream())
    want = x.permute(0, 2, 3, 1).reshape(-1, Cc) + f
    assert hu.relerr(tgt[:, :Cc], want) <= _tol(dt, 1e-6)
    nchw = torch.zeros(N, Cc, H, W, device=hu.DEV)
    call("plyolo_nhwc_to_nchw_f32", dt, N, H, W, Cc, xm.data_ptr(), Cc + 8, nchw.data_ptr(), hu.stream())
    assert torch.equal(nchw, x)
    back = torch.zeros(N * H * W, Cc, dtype=hu.tdtype(dt), device=hu.DEV)
    call("plyolo_nchw_f32_to_nhwc", dt, N, H, W, Cc, x.contiguous().data_ptr(), back.data_ptr(), Cc, hu.stream())
    assert torch.equal(hu.from_nhwc(back, N, H, W, Cc), x)


def test_sgd_ema():
    torch.manual_seed(4)
    n = 100003
    p = torch.randn(n, device=hu.DEV)
    g = torch.randn(n, device=hu.DEV)
    ref_p = torch.nn.Parameter(p.clone())
    opt = torch.optim.SGD([ref_p], lr=0.01, momentum=0.9)
    mom = torch.zeros(n, device=hu.DEV)
    ema = torch.randn(n, device=hu.DEV)
    ema_ref = ema.clone()
    for step in range(3):
        ref_p.grad = g.clone() * (step + 1)
        opt.step()
        call("plyolo_sgd_momentum", p.data_ptr(), (g * (step + 1)).data_ptr(), mom.data_ptr(), n, None, 0.01, 0.9, int(step == 0), hu.stream())
        d = 0.9998 * (1 - 2.718281828459045 ** (-(step + 1) / 2000))
        ema_ref.mul_(d).add_((1 - d) * ref_p.data)
        call("plyolo_ema_update", ema.data_ptr(), p.data_ptr(), n, d, hu.stream())
    torch.cuda.synchronize()
    assert hu.relerr(p, ref_p.data) < 1e-6
    assert hu.relerr(ema, ema_ref) < 1e-6


@pytest.mark.parametrize("nslab,elems", [(1, 1024), (5, 4100), (16, 9216), (100, 9216), (1024, 2304), (54, 147456), (333, 36)])
def test_reduce_slabs(nslab, elems):
    """slab 0 += slabs 1..n-1 through the grouped two-launch fold; the other slabs' contents are scratch."""
    torch.manual_seed(nslab + elems)
    slabs = torch.randn(nslab, elems, device=hu.DEV)
    ref = slabs.double().sum(0)
    call("plyolo_reduce_slabs", slabs.data_ptr(), nslab, elems, hu.stream())
    torch.cuda.synchronize()
    scale = float(ref.abs().max()) + 1e-6
    assert float((slabs[0].double() - ref).abs().max()) <= 2e-6 * scale * max(1.0, nslab ** 0.5)
    # deterministic: a second run on the same data gives the same bits
    torch.manual_seed(nslab + elems)
    slabs2 = torch.randn(nslab, elems, device=hu.DEV)
    call("plyolo_reduce_slabs", slabs2.data_ptr(), nslab, elems, hu.stream())
    torch.cuda.synchronize()
    assert torch.equal(slabs[0], slabs2[0])
