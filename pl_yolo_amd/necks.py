"""CSP-PAFPN neck (reference models/necks/pafpn_csp.py:7-86)."""
import os

import torch
import torch.nn as nn

from . import graph as G
from ._lib import PlyoloError
from .layers import emit_pair, BaseConv, CSPLayer, HipModule


# Launch lane of the bottom-up path (n3 -> n4 -> n5): its 40x40 / 20x20 kernels are short and latency-bound, and nothing on the
# 80x80 head level depends on them -- on a side lane they run beside that level instead of in front of it (heads.py: PLYOLO_HEAD_LANES).
_NECK_LANE = int(os.environ.get("PLYOLO_NECK_LANE", "2"))


class CSPPAFPN(HipModule):
    """Only proceeds 3-level input (stage2, stage3, stage4).  NOTE: like the reference,
    all four CSPLayers use depths[0] (pafpn_csp.py:26,34,46,54)."""

    def __init__(self, depths=(1, 1, 1, 1), in_channels=(256, 512, 1024), norm="bn", act="silu"):
        super().__init__()
        self.shrink_conv1 = BaseConv(in_channels[2], in_channels[1], 1, 1, norm=norm, act=act)
        self.shrink_conv2 = BaseConv(in_channels[1], in_channels[0], 1, 1, norm=norm, act=act)
        self.upsample = nn.Upsample(scale_factor=2, mode="nearest")
        self.p5_p4 = CSPLayer(2 * in_channels[1], in_channels[1], num_bottle=depths[0], shortcut=False, norm=norm, act=act)
        self.p4_p3 = CSPLayer(2 * in_channels[0], in_channels[0], num_bottle=depths[0], shortcut=False, norm=norm, act=act)
        self.downsample_conv1 = BaseConv(int(in_channels[0]), int(in_channels[0]), 3, 2, norm=norm, act=act)
        self.downsample_conv2 = BaseConv(int(in_channels[1]), int(in_channels[1]), 3, 2, norm=norm, act=act)
        self.n3_n4 = CSPLayer(2 * in_channels[0], in_channels[1], num_bottle=depths[0], shortcut=False, norm=norm, act=act)
        self.n4_n5 = CSPLayer(2 * in_channels[1], in_channels[2], num_bottle=depths[0], shortcut=False, norm=norm, act=act)

    def emit(self, g, inputs):
        c3, c4, c5 = inputs
        # top-down
        p5_expand = self.shrink_conv1.emit(g, c5)
        p5_upsample = G.UpsampleOp(g, p5_expand).out
        p4 = self.p5_p4.emit(g, g.concat([p5_upsample, c4]))
        p4_expand = self.shrink_conv2.emit(g, p4)
        p4_upsample = G.UpsampleOp(g, p4_expand).out
        p3 = self.p4_p3.emit(g, g.concat([p4_upsample, c3]))
        # bottom-up
        n3 = p3
        with g.on_lane(_NECK_LANE):
            n3_downsample = self.downsample_conv1.emit(g, n3)
            n4 = self.n3_n4.emit(g, g.concat([n3_downsample, p4_expand]))
            n4_downsample = self.downsample_conv2.emit(g, n4)
            n5 = self.n4_n5.emit(g, g.concat([n4_downsample, p5_expand]))
        return (n3, n4, n5)


class ELANWLayer(HipModule):
    """The `CSPLayer` of models/necks/yolov7_neck.py:104-146 (ELAN-W: (num_bottle+3)-way concat)."""

    def __init__(self, in_channel, out_channel, expansion=0.5, num_bottle=1, norm="bn", act="silu"):
        super().__init__()
        hi_channel = int(in_channel * expansion)
        self.num_conv = num_bottle
        self.conv1 = BaseConv(in_channel, hi_channel, 1, stride=1, norm=norm, act=act)
        self.conv2 = BaseConv(in_channel, hi_channel, 1, stride=1, norm=norm, act=act)
        self.conv3 = BaseConv(hi_channel, hi_channel // 2, 1, stride=1, norm=norm, act=act)
        self.conv4 = nn.ModuleList([BaseConv(hi_channel // 2, hi_channel // 2, 3, stride=1, norm=norm, act=act) for _ in range(num_bottle)])
        cat_channel = hi_channel // 2 * (num_bottle + 1) + hi_channel * 2
        self.conv5 = BaseConv(cat_channel, out_channel, 1, stride=1, norm=norm, act=act)

    def emit(self, g, x):
        x_1, x_2 = emit_pair(g, x, self.conv1, self.conv2)
        x_3 = self.conv3.emit(g, x_2)
        x_all = [x_1, x_2, x_3]
        for m in self.conv4:
            x_3 = m.emit(g, x_3)
            x_all.append(x_3)
        return self.conv5.emit(g, g.concat(x_all))


class NeckTransition(HipModule):
    """`Transition` of models/necks/yolov7_neck.py:149-164."""

    def __init__(self, in_channel, out_channel, mpk=2, norm="bn", act="silu"):
        super().__init__()
        if mpk != 2:
            raise NotImplementedError("Transition max-pool kernel is 2x2 stride 2")
        self.mp = nn.MaxPool2d(kernel_size=mpk, stride=mpk)
        self.conv1 = BaseConv(in_channel, out_channel // 2, 1, 1)
        self.conv2 = BaseConv(in_channel, out_channel // 2, 1, 1)
        self.conv3 = BaseConv(out_channel // 2, out_channel // 2, 3, 2, norm=norm, act=act)

    def emit(self, g, x):
        x_1 = self.conv1.emit(g, G.MaxPool2x2Op(g, x).out)
        x_2 = self.conv3.emit(g, self.conv2.emit(g, x))
        return g.concat([x_2, x_1])


class RepConv(HipModule):
    """RepConv, train-time form (reference models/necks/yolov7_neck.py:167-211):
    act(bn(conv3x3(x)) + bn(conv1x1(x)) [+ bn(x) when c1 == c2 and s == 1]), plain nn.BatchNorm2d defaults.
    The reference defines it next to the YOLOv7 neck but wires it into no config (n3/n4/n5 are BaseConv,
    yolov7_neck.py:67-69); here it is an optional block (`YOLOv7NECK(..., repconv=True)`).  Same constructor
    signature and state_dict keys (rbr_dense.0/1, rbr_1x1.0/1, rbr_identity).  The deploy-time
    re-parameterisation (:213-348) is export tooling and not part of the hot path."""

    def __init__(self, c1, c2, k=3, s=1, p=None, g=1, act=True, deploy=False):
        super().__init__()
        if k != 3 or (p is not None and p != 1):
            raise AssertionError("RepConv: k == 3 and padding 1 (yolov7_neck.py:180-181)")
        if g != 1 or s != 1 or act is not True:
            raise NotImplementedError("RepConv: stride 1, groups 1 and SiLU only (what the HIP conv kernels cover)")
        self.in_channels, self.out_channels, self.groups, self.deploy = c1, c2, g, deploy
        self.act = nn.SiLU()
        if deploy:      # yolov7_neck.py:187-188
            self.rbr_reparam = nn.Conv2d(c1, c2, 3, s, 1, groups=g, bias=True)
        else:
            self.rbr_identity = nn.BatchNorm2d(num_features=c1) if c2 == c1 and s == 1 else None
            self.rbr_dense = nn.Sequential(nn.Conv2d(c1, c2, 3, s, 1, groups=g, bias=False), nn.BatchNorm2d(num_features=c2))
            self.rbr_1x1 = nn.Sequential(nn.Conv2d(c1, c2, 1, s, 0, groups=g, bias=False), nn.BatchNorm2d(num_features=c2))

    def emit(self, g, x):
        if hasattr(self, "rbr_reparam"):   # deploy form: act(conv3x3(x) + bias), yolov7_neck.py:203-204
            return G.ConvUnitOp(g, x, self.rbr_reparam.weight, None, "silu", 1, conv_b=self.rbr_reparam.bias).out
        # three BatchNorm branches accumulated through the residual input of the BN-apply kernel, then SiLU
        t = G.ConvUnitOp(g, x, self.rbr_dense[0].weight, self.rbr_dense[1], None, 1).out
        t = G.ConvUnitOp(g, x, self.rbr_1x1[0].weight, self.rbr_1x1[1], None, 1, residual=t).out
        if self.rbr_identity is not None:
            t = G.BnOnlyOp(g, x, self.rbr_identity, residual=t).out
        return G.ActOp(g, t, "silu").out

    # ---- deploy-time re-parameterisation (yolov7_neck.py:213-348), one HIP launch (plyolo_repconv_fuse)
    def get_equivalent_kernel_bias(self):
        """(kernel [c2, c1, 3, 3], bias [c2]) of the single 3x3 convolution equivalent to the three inference-mode branches:
        kernel3x3 + pad(kernel1x1) + kernel_id, bias3x3 + bias1x1 + bias_id (yolov7_neck.py:213-220)."""
        import ctypes as C
        from ._lib import call
        from .layers import _bn_params
        w3, w1 = self.rbr_dense[0].weight.detach(), self.rbr_1x1[0].weight.detach()
        if not w3.is_cuda:
            raise PlyoloError("RepConv re-parameterisation runs on the MI355X (move the model to the device first); there is no CPU path")
        w3, w1 = w3.float().contiguous(), w1.float().contiguous()
        c2, c1 = w3.shape[0], w3.shape[1]
        kernel = torch.empty(c2, c1, 3, 3, dtype=torch.float32, device=w3.device)
        bias = torch.empty(c2, dtype=torch.float32, device=w3.device)
        b3, b1 = _bn_params(self.rbr_dense[1]), _bn_params(self.rbr_1x1[1])
        bid = _bn_params(self.rbr_identity) if self.rbr_identity is not None else None
        call("plyolo_repconv_fuse", w3.data_ptr(), C.byref(b3), w1.data_ptr(), C.byref(b1), C.byref(bid) if bid is not None else None,
             c2, c1, kernel.data_ptr(), bias.data_ptr(), torch.cuda.current_stream().cuda_stream)
        return kernel, bias

    def repvgg_convert(self):
        kernel, bias = self.get_equivalent_kernel_bias()
        return kernel.detach().cpu().numpy(), bias.detach().cpu().numpy()

    def fuse_conv_bn(self, conv, bn):
        from .layers import fold_conv_bn
        return fold_conv_bn(conv, bn)

    def fuse_repvgg_block(self):
        """Switch to the deploy form in place (yolov7_neck.py:288-348): rbr_reparam = one 3x3 conv with bias, branches deleted."""
        if self.deploy:
            return
        kernel, bias = self.get_equivalent_kernel_bias()
        conv = nn.Conv2d(self.in_channels, self.out_channels, 3, 1, 1, groups=self.groups, bias=True, device=kernel.device)
        conv.weight = nn.Parameter(kernel)
        conv.bias = nn.Parameter(bias)
        self.rbr_reparam = conv
        self.deploy = True
        self.rbr_identity = None
        self.rbr_1x1 = None
        self.rbr_dense = None


class YOLOv7NECK(HipModule):
    """models/necks/yolov7_neck.py:7-101.  (RepConv, :167-348, is defined upstream but never
    instantiated by any config -- n3/n4/n5 are plain BaseConv, :67-69.)"""

    def __init__(self, depths=(1, 1, 1, 1), in_channels=(512, 1024, 1024), norm="bn", act="silu", repconv=False):
        super().__init__()
        from .layers import SPPCSPC
        c = in_channels
        self.spp = SPPCSPC(c[2], c[2] // 2, k=(5, 9, 13))
        self.conv_for_P5 = BaseConv(c[2] // 2, c[2] // 4, 1, 1, norm=norm, act=act)
        self.upsample = nn.Upsample(scale_factor=2, mode="nearest")
        self.conv_for_C4 = BaseConv(c[1], c[2] // 4, 1, 1, norm=norm, act=act)
        self.p5_p4 = ELANWLayer(c[2] // 2, c[2] // 4, expansion=0.5, num_bottle=depths[0], norm=norm, act=act)
        self.conv_for_P4 = BaseConv(c[2] // 4, c[2] // 8, 1, 1, norm=norm, act=act)
        self.conv_for_C3 = BaseConv(c[0], c[2] // 8, 1, 1, norm=norm, act=act)
        self.p4_p3 = ELANWLayer(c[2] // 4, c[2] // 8, expansion=0.5, num_bottle=depths[0], norm=norm, act=act)
        self.downsample_conv1 = NeckTransition(c[2] // 8, c[2] // 4, mpk=2, norm=norm, act=act)
        self.n3_n4 = ELANWLayer(c[2] // 2, c[2] // 4, expansion=0.5, num_bottle=depths[0], norm=norm, act=act)
        self.downsample_conv2 = NeckTransition(c[2] // 4, c[2] // 2, mpk=2, norm=norm, act=act)
        self.n4_n5 = ELANWLayer(c[2], c[2] // 2, expansion=0.5, num_bottle=depths[0], norm=norm, act=act)
        if repconv:   # BASELINE cfg 3 names "ELAN backbone + RepConv": optional, the reference itself uses BaseConv here
            self.n3 = RepConv(c[2] // 8, c[2] // 4, 3, 1)
            self.n4 = RepConv(c[2] // 4, c[2] // 2, 3, 1)
            self.n5 = RepConv(c[2] // 2, c[2], 3, 1)
        else:
            self.n3 = BaseConv(c[2] // 8, c[2] // 4, 3, 1, norm=norm, act=act)
            self.n4 = BaseConv(c[2] // 4, c[2] // 2, 3, 1, norm=norm, act=act)
            self.n5 = BaseConv(c[2] // 2, c[2], 3, 1, norm=norm, act=act)

    def emit(self, g, inputs):
        c3, c4, c5 = inputs
        p5 = self.spp.emit(g, c5)
        p5_shrink = self.conv_for_P5.emit(g, p5)
        p5_upsample = G.UpsampleOp(g, p5_shrink).out
        p4 = self.p5_p4.emit(g, g.concat([p5_upsample, self.conv_for_C4.emit(g, c4)]))
        p4_shrink = self.conv_for_P4.emit(g, p4)
        p4_upsample = G.UpsampleOp(g, p4_shrink).out
        p3 = self.p4_p3.emit(g, g.concat([p4_upsample, self.conv_for_C3.emit(g, c3)]))
        n3 = p3
        with g.on_lane(_NECK_LANE):
            n3_downsample = self.downsample_conv1.emit(g, n3)
            n4 = self.n3_n4.emit(g, g.concat([n3_downsample, p4]))
            n4_downsample = self.downsample_conv2.emit(g, n4)
            n5 = self.n4_n5.emit(g, g.concat([n4_downsample, p5]))
            o4, o5 = self.n4.emit(g, n4), self.n5.emit(g, n5)
        return (self.n3.emit(g, n3), o4, o5)
