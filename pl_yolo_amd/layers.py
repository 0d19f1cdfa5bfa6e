"""Network blocks with the reference's names, constructor signatures and state_dict
key layout (reference models/layers/network_blocks.py, activation.py,
normalization.py).  The modules own the parameters (plain nn.Conv2d /
nn.BatchNorm2d containers, so checkpoints load unchanged) and describe their
computation to the HIP launch-plan builder through `emit(g, x)`; they never run
torch compute kernels themselves.
"""
import torch
import torch.nn as nn

from . import graph as G
from ._lib import PlyoloError

_ACTS = ("silu", "relu", "lrelu", "hswish", "gelu")


def get_activation(name="silu", inplace=True):
    """activation.py:5-26.  Returns a marker module (the activation is fused into the
    BatchNorm-apply / conv-epilogue kernels: csrc/common.h act_fwd / act_grad)."""
    if name is None:
        return None
    if name in _ACTS:
        m = nn.Identity()
        m.act_name = name
        return m
    raise AttributeError("Unsupported activation function type: {}".format(name))


def get_normalization(name, out_channels):
    """normalization.py:4-13 (bn = BatchNorm2d(eps=1e-3, momentum=0.03))."""
    if name is None:
        return None
    if name == "bn":
        return nn.BatchNorm2d(out_channels, eps=1e-3, momentum=0.03)
    if name == "ln":
        return nn.LayerNorm(out_channels)     # over the LAST axis of the NCHW tensor, i.e. the width (G.LnWidthOp)
    raise AttributeError("Unsupported normalization function type: {}".format(name))


class HipModule(nn.Module):
    """Base: sub-modules are graph describers (`emit`); the detector traces the whole network through them once (runner.py).
    Called directly they keep the reference's tensor contract -- backbone `Tensor -> list`, neck `list -> list`, head
    `list -> list of raw NCHW maps`, blocks `Tensor -> Tensor` -- through a traced session of their own (module_runner.py):
    forward plan, and in training mode an autograd node that replays the recorded backward plan.  That is the boundary for
    callers that mix these modules with foreign ones, not the hot path.  Modules without an `emit` of that shape (the loss
    plugins: their `(list, labels) -> dict` has no caller outside OneStageD) refuse a direct call loudly."""

    def forward(self, *a, **k):
        if k or not hasattr(self, "emit") or type(self).__name__ in ("YOLOXLoss", "YOLOv7Loss"):
            raise PlyoloError(
                "%s is a graph describer for the HIP launch plan; run it through build_model()/OneStageD "
                "(its tensor contract has no stand-alone path)" % type(self).__name__
            )
        from . import module_runner
        return module_runner.run(self, *a)


class BaseConv(HipModule):
    """A Convolution2d -> Normalization -> Activation (network_blocks.py:7-40)."""

    def __init__(self, in_channels, out_channels, ksize, stride, padding=None, groups=1, bias=False, norm="bn", act="silu"):
        super().__init__()
        pad = (ksize - 1) // 2 if padding is None else padding
        # groups: 1 (MFMA kernels) or depthwise 3x3 stride 1 (groups == in == out: the e-yolox Bottleneck, csrc/dwconv.hip)
        self.depthwise = groups != 1 and groups == in_channels == out_channels and ksize == 3 and stride == 1
        if (groups != 1 and not self.depthwise) or bias or pad != (ksize - 1) // 2 or ksize not in (1, 3) or stride not in (1, 2):
            raise NotImplementedError("HIP conv supports k in {1,3}, stride in {1,2}, same padding, no bias, groups=1 or depthwise 3x3 stride 1")
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=ksize, stride=stride, padding=pad, groups=groups, bias=bias)
        self.norm = get_normalization(norm, out_channels)
        self.act = get_activation(act, inplace=True)
        self.stride = stride

    def emit(self, g, x, residual=None, need_dgrad=True, cin_pad=None):
        act = self.act.act_name if self.act is not None else None
        if getattr(self, "depthwise", False):
            if self.norm is None:
                raise NotImplementedError("depthwise BaseConv without BatchNorm (fold it back or keep norm='bn')")
            return G.DwConvUnitOp(g, x, self.conv.weight, self.norm, act, residual).out
        if isinstance(self.norm, nn.LayerNorm):
            if residual is not None:
                raise NotImplementedError("a shortcut on a BaseConv with norm='ln'")
            z = G.ConvUnitOp(g, x, self.conv.weight, None, None, self.stride, None, need_dgrad, cin_pad, conv_b=self.conv.bias).out
            return G.LnWidthOp(g, z, self.norm, act).out
        op = G.ConvUnitOp(g, x, self.conv.weight, self.norm, act, self.stride, residual, need_dgrad, cin_pad, conv_b=self.conv.bias)
        return op.out

    def fuse(self):
        """Deploy form: fold the BatchNorm into the convolution (weights scaled per output channel, a bias appears) and drop
        it, so that forward == the reference's `fuseforward` = act(conv(x)) (network_blocks.py:39-40; the folding rule is
        RepConv.fuse_conv_bn, yolov7_neck.py:265-286).  One HIP launch (plyolo_fold_conv_bn); inference only."""
        if self.norm is None or getattr(self, "depthwise", False):   # depthwise units keep their BatchNorm (no fused deploy kernel)
            return self
        if not isinstance(self.norm, nn.BatchNorm2d):
            raise NotImplementedError("only BatchNorm2d folds into a convolution")
        self.conv = fold_conv_bn(self.conv, self.norm)
        self.norm = None
        return self

    fuseforward = HipModule.forward   # same contract as forward: the module only describes its launch-plan ops


def _bn_params(bn):
    from ._lib import BnParams, ptr
    b = BnParams()
    b.gamma, b.beta = ptr(bn.weight), ptr(bn.bias)
    b.running_mean, b.running_var, b.eps = ptr(bn.running_mean), ptr(bn.running_var), float(bn.eps)
    return b


def fold_conv_bn(conv, bn):
    """nn.Conv2d (+ optional bias) followed by an inference-mode BatchNorm2d -> one nn.Conv2d with bias
    (RepConv.fuse_conv_bn, yolov7_neck.py:265-286), computed on the device by plyolo_fold_conv_bn."""
    import ctypes as C
    from ._lib import call, ptr
    w = conv.weight.detach()
    if not w.is_cuda:
        raise PlyoloError("BatchNorm folding runs on the MI355X (move the model to the device first); there is no CPU path")
    w = w.float().contiguous()
    Cout, K = w.shape[0], w[0].numel()
    w_out = torch.empty_like(w)
    b_out = torch.empty(Cout, dtype=torch.float32, device=w.device)
    bp = _bn_params(bn)
    call("plyolo_fold_conv_bn", w.data_ptr(), ptr(conv.bias), C.byref(bp), Cout, K, w_out.data_ptr(), b_out.data_ptr(),
         torch.cuda.current_stream().cuda_stream)
    fused = nn.Conv2d(conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding, conv.dilation, conv.groups,
                      bias=True, padding_mode=conv.padding_mode, device=w.device)
    fused.weight = nn.Parameter(w_out)
    fused.bias = nn.Parameter(b_out)
    return fused


def emit_pair(g, x, unit_a, unit_b):
    """Two BaseConv units applied to the same input.  When they agree in kernel size, stride, activation
    and BatchNorm hyper-parameters they are lowered to ONE merged convolution (graph.ConvPairOp); the two
    modules keep their own parameters and state_dict keys.  Returns (out_a, out_b)."""
    ca, cb = unit_a.conv, unit_b.conv
    na, nb = unit_a.norm, unit_b.norm
    act_a = unit_a.act.act_name if unit_a.act is not None else None
    act_b = unit_b.act.act_name if unit_b.act is not None else None
    same = (g.pair_convs and ca.kernel_size == cb.kernel_size and unit_a.stride == unit_b.stride and act_a == act_b
            and na is not None and nb is not None and na.eps == nb.eps and na.momentum == nb.momentum
            and ca.out_channels % g.vec == 0 and cb.out_channels % g.vec == 0)
    if not same:
        return unit_a.emit(g, x), unit_b.emit(g, x)
    op = G.ConvPairOp(g, x, ca.weight, na, cb.weight, nb, act_a, unit_a.stride)
    return op.out_a, op.out_b


class Focus(HipModule):
    """Focus width and height information into channel space (network_blocks.py:43-65)."""

    def __init__(self, in_channels, out_channels, ksize=1, stride=1, norm="bn", act="silu"):
        super().__init__()
        if in_channels != 3:
            raise NotImplementedError("Focus kernel is written for RGB input")
        self.conv = BaseConv(in_channels * 4, out_channels, ksize, stride, norm=norm, act=act)

    def emit(self, g, image_act):
        # image_act: NHWC space-to-depth tensor produced by plyolo_focus_s2d (12 real channels)
        return self.conv.emit(g, image_act, need_dgrad=False, cin_pad=image_act.C)


class Bottleneck(HipModule):
    """network_blocks.py:68-91 (carries the reference's unused `bn`, line 81)."""

    def __init__(self, in_channels, out_channels, shortcut=True, expansion=0.5, norm="bn", act="silu"):
        super().__init__()
        hidden_channels = int(out_channels * expansion)
        self.bn = get_normalization(norm, out_channels)
        self.act = get_activation(act, inplace=True)
        self.conv1 = BaseConv(in_channels, hidden_channels, 1, stride=1, norm=norm, act=act)
        self.conv2 = BaseConv(hidden_channels, out_channels, 3, stride=1, norm=norm, act=act)
        self.use_add = shortcut and in_channels == out_channels

    def emit(self, g, x):
        y = self.conv1.emit(g, x)
        return self.conv2.emit(g, y, residual=x if self.use_add else None)

    def dead_parameters(self):
        """The reference's Bottleneck carries a BatchNorm it never calls (network_blocks.py:81): its affine pair never gets a
        gradient.  The runner keeps such parameters behind the live ones in the flat buffers, outside the exchanged buckets."""
        return list(self.bn.parameters()) if self.bn is not None else []


class CSPLayer(HipModule):
    """network_blocks.py:94-131."""

    def __init__(self, in_channels, out_channels, num_bottle=1, shortcut=True, expansion=0.5, norm="bn", act="silu"):
        super().__init__()
        hidden_channels = int(out_channels * expansion)
        self.conv1 = BaseConv(in_channels, hidden_channels, 1, stride=1, norm=norm, act=act)
        self.conv2 = BaseConv(in_channels, hidden_channels, 1, stride=1, norm=norm, act=act)
        self.conv3 = BaseConv(2 * hidden_channels, out_channels, 1, stride=1, norm=norm, act=act)
        self.m = nn.Sequential(*[Bottleneck(hidden_channels, hidden_channels, shortcut, 1.0, norm=norm, act=act) for _ in range(num_bottle)])

    def emit(self, g, x):
        x_1, x_2 = emit_pair(g, x, self.conv1, self.conv2)
        for b in self.m:
            x_1 = b.emit(g, x_1)
        return self.conv3.emit(g, g.concat([x_1, x_2]))


class SPPBottleneck(HipModule):
    """Spatial pyramid pooling layer used in YOLOv3-SPP (network_blocks.py:134-155)."""

    def __init__(self, in_channels, out_channels, kernel_sizes=(5, 9, 13), norm="bn", act="silu"):
        super().__init__()
        hidden_channels = in_channels // 2
        self.conv1 = BaseConv(in_channels, hidden_channels, 1, stride=1, norm=norm, act=act)
        self.m = nn.ModuleList([nn.MaxPool2d(kernel_size=ks, stride=1, padding=ks // 2) for ks in kernel_sizes])
        self.kernel_sizes = tuple(kernel_sizes)
        conv2_channels = hidden_channels * (len(kernel_sizes) + 1)
        self.conv2 = BaseConv(conv2_channels, out_channels, 1, stride=1, act=act)  # norm arg not forwarded (line 149)

    def emit(self, g, x):
        x = self.conv1.emit(g, x)
        pools = G.SppPoolsOp(g, x, self.kernel_sizes)
        return self.conv2.emit(g, g.concat([x] + pools.outs))


class SPPCSPC(HipModule):
    """YOLOv7 SPP-CSP block (network_blocks.py:158-175): all convs use the default bn/silu."""

    def __init__(self, c1, c2, k=(5, 9, 13)):
        super().__init__()
        self.cv1 = BaseConv(c1, c2, 1, 1)
        self.cv2 = BaseConv(c1, c2, 1, 1)
        self.cv3 = BaseConv(c2, c2, 3, 1)
        self.cv4 = BaseConv(c2, c2, 1, 1)
        self.m = nn.ModuleList([nn.MaxPool2d(kernel_size=x, stride=1, padding=x // 2) for x in k])
        self.kernel_sizes = tuple(k)
        self.cv5 = BaseConv(4 * c2, c2, 1, 1)
        self.cv6 = BaseConv(c2, c2, 3, 1)
        self.cv7 = BaseConv(2 * c2, c2, 1, 1)

    def emit(self, g, x):
        t1, y2 = emit_pair(g, x, self.cv1, self.cv2)
        x1 = self.cv4.emit(g, self.cv3.emit(g, t1))
        pools = G.SppPoolsOp(g, x1, self.kernel_sizes)
        y1 = self.cv6.emit(g, self.cv5.emit(g, g.concat([x1] + pools.outs)))
        return self.cv7.emit(g, g.concat([y1, y2]))
