"""-m gpu: deploy-time folding and the hswish / gelu activations of the BaseConv surface.

  * plyolo_repconv_fuse / plyolo_fold_conv_bn through RepConv.get_equivalent_kernel_bias / fuse_repvgg_block and
    BaseConv.fuse against the fixture the reference's own methods wrote (tests/golden/deploy_fold.npz);
  * the deploy forms through the launch plans (conv + bias + activation in ONE launch on the bf16 path) against the
    reference's fused outputs;
  * OneStageD.fuse(): the whole YOLOv7 toy detector with RepConv n3/n4/n5 before vs after folding;
  * BaseConv(act="hswish" | "gelu"): forward / backward against plain PyTorch fp32 on the same device."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F
import yaml

pytestmark = pytest.mark.gpu

import pl_yolo_amd  # noqa: E402
from pl_yolo_amd import graph as G  # noqa: E402
from pl_yolo_amd._lib import BF16, F32, call  # noqa: E402
from conftest import load_golden, ROOT  # noqa: E402
import hiputil as hu  # noqa: E402


def _state(g, prefix):
    return {k[len(prefix):]: torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith(prefix)}


def run_block(m, x, dtype, training=False, r=None):
    """Run one describer module through recorded launch plans: returns y [N,C,H,W] (and, with an upstream gradient r in
    training mode, dx and {parameter: gradient})."""
    dt = BF16 if dtype == "bf16" else F32
    N, c1, H, W = x.shape
    gr = G.Graph(dt, training, torch.device(hu.DEV))
    gr.use_lanes = False
    grads = {id(p): torch.zeros_like(p) for p in m.parameters()}
    gr.grad_ptr_of = lambda p: grads[id(p)].data_ptr() if p is not None and id(p) in grads else None
    xin = gr.new_act(N, H, W, c1, "x")
    out = m.emit(gr, xin)
    gr.allocate()
    gr.build_pack_table(gr.grad_ptr_of)
    xin.storage.tensor.view(-1, xin.ld)[:, :c1] = x.permute(0, 2, 3, 1).reshape(-1, c1).to(gr.tdtype)
    fwd = G.Plan()
    with fwd:
        gr.plan = fwd
        call("plyolo_pack_weights", gr.pack_table.data_ptr(), gr.n_pack, gr.dtype, gr.max_pack_elems, None)
        gr.zero_fwd_stats()
        G.record_ops(gr, fwd, gr.ops, "fwd")
    fwd.run(hu.stream())
    c2, OH, OW = out.C, out.H, out.W
    y = out.storage.tensor.view(-1, out.ld)[:, out.c_off:out.c_off + c2].float().reshape(N, OH, OW, c2).permute(0, 3, 1, 2).contiguous()
    if r is None:
        torch.cuda.synchronize()
        return y
    gout = gr.grad_storage(out.storage)
    gout.view(-1, out.ld)[:, out.c_off:out.c_off + c2] = r.permute(0, 2, 3, 1).reshape(-1, c2).to(gr.tdtype)
    for i in range(out.c_off, out.c_off + c2):
        out.storage.ginit[i] = True
    bwd = G.Plan()
    with bwd:
        gr.plan = bwd
        if dt != BF16:
            call("plyolo_memset_async", gr.dwp_arena.data_ptr(), 0, gr.dwp_arena.numel() * 4, None)
        gr.zero_bwd_stats()
        G.record_ops(gr, bwd, list(reversed(gr.ops)), "bwd")
        call("plyolo_unpack_wgrads", gr.pack_table.data_ptr(), gr.n_pack, gr.max_pack_elems, 0, None)
    bwd.run(hu.stream())
    torch.cuda.synchronize()
    dx = gr.grad_storage(xin.storage).view(-1, xin.ld)[:, :c1].float().reshape(N, H, W, c1).permute(0, 3, 1, 2).contiguous()
    return y, dx, {n: grads[id(p)] for n, p in m.named_parameters()}


@pytest.mark.parametrize("tag", ["ne", "id"])
def test_repconv_reparameterisation_vs_reference(tag):
    from pl_yolo_amd.necks import RepConv
    g = load_golden("deploy_fold")
    sd = _state(g, "rep_%s/state/" % tag)
    x = torch.from_numpy(g["rep_%s/x" % tag]).to(hu.DEV)
    c1, c2 = x.shape[1], g["rep_%s/y_eval" % tag].shape[1]
    m = RepConv(c1, c2, 3, 1)
    m.load_state_dict(sd)
    m = m.to(hu.DEV).eval()
    y_eval = run_block(m, x, "fp32")                      # three-branch inference form through the plans
    assert hu.relerr(y_eval, torch.from_numpy(g["rep_%s/y_eval" % tag]).to(hu.DEV)) <= 1e-5
    kernel, bias = m.get_equivalent_kernel_bias()          # plyolo_repconv_fuse
    np.testing.assert_allclose(kernel.cpu().numpy(), g["rep_%s/kernel" % tag], rtol=2e-6, atol=2e-7)
    np.testing.assert_allclose(bias.cpu().numpy(), g["rep_%s/bias" % tag], rtol=2e-6, atol=2e-7)
    k2, b2 = m.repvgg_convert()
    assert isinstance(k2, np.ndarray) and k2.shape == (c2, c1, 3, 3) and b2.shape == (c2,)
    m.fuse_repvgg_block()
    assert m.deploy and set(m.state_dict()) == {"rbr_reparam.weight", "rbr_reparam.bias"}
    np.testing.assert_allclose(m.rbr_reparam.weight.detach().cpu().numpy(), g["rep_%s/reparam_weight" % tag], rtol=1e-5, atol=1e-6)
    want = torch.from_numpy(g["rep_%s/y_fused" % tag]).to(hu.DEV)
    y32 = run_block(m, x, "fp32")
    y16 = run_block(m, x, "bf16")                          # conv + bias + SiLU in one launch (fused epilogue)
    print("repconv deploy", tag, "fp32 relerr %.3g bf16 relerr %.3g" % (hu.relerr(y32, want), hu.relerr(y16, want)))
    assert hu.relerr(y32, want) <= 1e-5 and hu.relerr(y16, want) <= 2e-2
    m2 = RepConv(c1, c2, 3, 1, deploy=True)                # constructed in deploy form (yolov7_neck.py:187-188)
    m2.load_state_dict(m.state_dict())
    assert hu.relerr(run_block(m2.to(hu.DEV).eval(), x, "fp32"), want) <= 1e-5


@pytest.mark.parametrize("tag", ["k3", "k1", "k3s2"])
def test_baseconv_bn_fold_vs_reference(tag):
    from pl_yolo_amd.layers import BaseConv
    g = load_golden("deploy_fold")
    sd = _state(g, "base_%s/state/" % tag)
    x = torch.from_numpy(g["base_%s/x" % tag]).to(hu.DEV)
    cout, cin, k = sd["conv.weight"].shape[:3]
    m = BaseConv(cin, cout, int(k), int(g["base_%s/stride" % tag]))
    m.load_state_dict(sd)
    m = m.to(hu.DEV).eval()
    want_eval = torch.from_numpy(g["base_%s/y_eval" % tag]).to(hu.DEV)
    assert hu.relerr(run_block(m, x, "fp32"), want_eval) <= 1e-5
    m.fuse()                                               # plyolo_fold_conv_bn
    assert m.norm is None and set(m.state_dict()) == {"conv.weight", "conv.bias"}
    np.testing.assert_allclose(m.conv.weight.detach().cpu().numpy(), g["base_%s/fused_weight" % tag], rtol=2e-6, atol=2e-7)
    np.testing.assert_allclose(m.conv.bias.detach().cpu().numpy(), g["base_%s/fused_bias" % tag], rtol=2e-6, atol=2e-7)
    want = torch.from_numpy(g["base_%s/y_fused" % tag]).to(hu.DEV)
    assert hu.relerr(run_block(m, x, "fp32"), want) <= 1e-5
    assert hu.relerr(run_block(m, x, "bf16"), want) <= 2e-2


def test_detector_fuse_matches_unfused_eval():
    """OneStageD.fuse() on the YOLOv7 toy detector with RepConv n3/n4/n5: every BatchNorm folded, every RepConv collapsed;
    the eval output is unchanged (fp32 1e-4 of the output scale) and the bf16 deploy path agrees with it."""
    with open(os.path.join(ROOT, "configs", "model", "yolov7", "yolov7_test.yaml")) as f:
        cfg = yaml.safe_load(f)
    cfg["neck"]["repconv"] = True
    torch.manual_seed(5)
    model = pl_yolo_amd.build_model(cfg, 3)
    gen = torch.Generator().manual_seed(9)
    for mod in model.modules():                      # non-trivial running statistics / affine parameters
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.data = torch.randn(mod.running_mean.shape, generator=gen) * 0.1
            mod.running_var.data = 0.5 + torch.rand(mod.running_var.shape, generator=gen)
            mod.weight.data = 0.5 + torch.rand(mod.weight.shape, generator=gen)
            mod.bias.data = torch.rand(mod.bias.shape, generator=gen) - 0.5
    model.compute_dtype = "fp32"
    model = model.to(hu.DEV).eval()
    x = (torch.rand(2, 3, 128, 128, generator=gen) * 255).to(hu.DEV)
    dummy = torch.zeros(2, 1, 5, device=hu.DEV)
    with torch.no_grad():
        before = model(x, dummy).clone()
    n_bn = sum(1 for k in model.state_dict() if "running_mean" in k)
    model.fuse()
    keys = list(model.state_dict())
    assert n_bn > 50 and not [k for k in keys if "running_mean" in k or ".norm." in k]
    assert any("rbr_reparam" in k for k in keys) and not any("rbr_dense" in k for k in keys)
    with torch.no_grad():
        after = model(x, dummy).clone()
        model.compute_dtype = "bf16"
        after16 = model(x, dummy).clone()
    scale = float(before.abs().max())
    e32, e16 = float((after - before).abs().max()) / scale, float((after16 - before).abs().max()) / scale
    print("fused vs unfused eval output: fp32 %.3g, bf16 deploy %.3g (of the output scale %.3g)" % (e32, e16, scale))
    assert e32 <= 1e-4 and e16 <= 5e-2


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("act", ["hswish", "gelu"])
def test_baseconv_hswish_gelu_vs_torch(act, dtype):
    """The two activations of models/layers/activation.py the YOLOX / YOLOv7 configs do not use: BaseConv(act=...) train-mode
    forward + backward through the launch plans against conv2d -> batch_norm -> activation in plain PyTorch fp32."""
    from pl_yolo_amd.layers import BaseConv
    torch.manual_seed(4)
    m = BaseConv(16, 24, 3, 1, act=act).to(hu.DEV).train()
    with torch.no_grad():
        m.norm.weight.uniform_(0.5, 1.5)
        m.norm.bias.uniform_(-1.0, 1.0)       # spread the pre-activations over both kinks of hswish
    gen = torch.Generator().manual_seed(6)
    x = (torch.randn(2, 16, 12, 10, generator=gen) * 2).to(hu.DEV)
    r = torch.randn(2, 24, 12, 10, generator=gen).to(hu.DEV)
    if dtype == "bf16":   # compare on the values the bf16 path actually stores
        x = hu.rnd_bf16(x)
    w = m.conv.weight.detach().clone().requires_grad_(True)
    gw = m.norm.weight.detach().clone().requires_grad_(True)
    gb = m.norm.bias.detach().clone().requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    z = F.conv2d(xr, hu.rnd_bf16(w) if dtype == "bf16" else w, None, 1, 1)
    u = F.batch_norm(z, None, None, gw, gb, True, 0.03, 1e-3)
    y_ref = (u * F.relu6(u + 3) / 6) if act == "hswish" else F.gelu(u)
    (y_ref * r).sum().backward()
    y, dx, grads = run_block(m, x, dtype, training=True, r=r)
    tol = 2e-2 if dtype == "bf16" else 2e-5
    print(act, dtype, "y %.3g dx %.3g dw %.3g" % (hu.relerr(y, y_ref), hu.relerr(dx, xr.grad), hu.relerr(grads["conv.weight"], w.grad)))
    assert hu.relerr(y, y_ref) <= tol
    assert hu.relerr(dx, xr.grad) <= (6e-2 if dtype == "bf16" else 2e-4)
    assert hu.relerr(grads["conv.weight"], w.grad) <= (6e-2 if dtype == "bf16" else 5e-4)
    assert hu.relerr(grads["norm.weight"], gw.grad) <= (6e-2 if dtype == "bf16" else 5e-4)
    assert hu.relerr(grads["norm.bias"], gb.grad) <= (6e-2 if dtype == "bf16" else 5e-4)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("act", ["silu", "lrelu"])
def test_baseconv_layernorm_over_width_vs_torch(act, dtype):
    """norm="ln" (models/layers/normalization.py:9-10): nn.LayerNorm(out_channels) on the NCHW conv output normalises the WIDTH axis
    (torch requires W == out_channels).  BaseConv(norm="ln") forward + backward through the launch plans against
    conv2d -> F.layer_norm -> activation in plain PyTorch fp32; a width != out_channels raises torch's error."""
    from pl_yolo_amd.layers import BaseConv
    torch.manual_seed(9)
    Cout, Wd = 24, 24
    m = BaseConv(16, Cout, 3, 1, norm="ln", act=act).to(hu.DEV).train()
    with torch.no_grad():
        m.norm.weight.uniform_(0.5, 1.5)
        m.norm.bias.uniform_(-0.5, 0.5)
    gen = torch.Generator().manual_seed(10)
    x = (torch.randn(2, 16, 10, Wd, generator=gen) * 2).to(hu.DEV)
    r = torch.randn(2, Cout, 10, Wd, generator=gen).to(hu.DEV)
    if dtype == "bf16":
        x = hu.rnd_bf16(x)
    w = m.conv.weight.detach().clone().requires_grad_(True)
    gw = m.norm.weight.detach().clone().requires_grad_(True)
    gb = m.norm.bias.detach().clone().requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    z = F.conv2d(xr, hu.rnd_bf16(w) if dtype == "bf16" else w, None, 1, 1)
    u = F.layer_norm(z, (Wd,), gw, gb, 1e-5)
    y_ref = F.silu(u) if act == "silu" else F.leaky_relu(u, 0.1)
    (y_ref * r).sum().backward()
    y, dx, grads = run_block(m, x, dtype, training=True, r=r)
    e = [hu.relerr(y, y_ref), hu.relerr(dx, xr.grad), hu.relerr(grads["conv.weight"], w.grad), hu.relerr(grads["norm.weight"], gw.grad),
         hu.relerr(grads["norm.bias"], gb.grad)]
    print("ln", act, dtype, " ".join("%.3g" % v for v in e))
    tol = [2e-2, 6e-2, 6e-2, 3e-2, 3e-2] if dtype == "bf16" else [2e-5, 2e-4, 5e-4, 2e-4, 2e-4]
    assert all(a <= b for a, b in zip(e, tol)), e
    bad = BaseConv(16, 20, 3, 1, norm="ln", act=act).to(hu.DEV)
    with pytest.raises(RuntimeError):
        run_block(bad, x, dtype)
