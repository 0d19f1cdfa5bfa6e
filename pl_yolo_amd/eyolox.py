"""The e-yolox plugin family: ECMNet backbone + AL_PAFPN neck (reference models/backbones/ecmnet.py:17-177,
models/necks/pafpn_al.py:7-182, configs/model/e-yolox/*.yaml).  Same describer contract as the other plugins: the modules
own the parameters with the reference's state_dict key layout and emit launch-plan ops.

Their building block is an inverted bottleneck of depthwise 3x3 and pointwise convolutions with ONE activation per block:
    conv0 dw3x3+BN -> conv1 1x1 (no BN) + act -> conv2 1x1+BN -> conv3 dw3x3+BN (+ x) -> act
and a CSP layer that concatenates four taps and has no fusing convolution behind the concat."""
import torch.nn as nn

from . import graph as G
from .layers import emit_pair, BaseConv, Focus, SPPBottleneck, HipModule, get_activation


class Bottleneck(HipModule):
    """ecmnet.py:142-177 == pafpn_al.py:147-182."""

    def __init__(self, in_channels, out_channels, stride=1, shortcut=True, expansion=0.5, norm="bn", act="silu", attn=None):
        super().__init__()
        if attn is not None or stride != 1:
            raise NotImplementedError("e-yolox Bottleneck: stride 1, no attention (the shipped configs)")
        hidden_channels = int(out_channels * expansion)
        self.conv0 = BaseConv(in_channels, in_channels, 3, stride=stride, groups=in_channels, norm=norm, act=None)
        self.conv1 = BaseConv(in_channels, hidden_channels, 1, stride=1, norm=None, act=act)
        self.conv2 = BaseConv(hidden_channels, out_channels, 1, stride=1, norm=norm, act=None)
        self.conv3 = BaseConv(out_channels, out_channels, 3, stride=stride, groups=out_channels, norm=norm, act=None)
        self.nonlinearity = get_activation(act)
        self.use_add = shortcut and in_channels == out_channels
        self.use_attn = False

    def emit(self, g, x):
        y = self.conv0.emit(g, x)
        y = self.conv1.emit(g, y)
        y = self.conv2.emit(g, y)
        y = self.conv3.emit(g, y, residual=x if self.use_add else None)      # bn(conv3(y)) + x in one BN-apply launch
        if self.nonlinearity is None:
            return y
        return G.ActOp(g, y, self.nonlinearity.act_name).out


class CSPLayer(HipModule):
    """ecmnet.py:91-139 == pafpn_al.py:96-144: four taps concatenated, no convolution behind the concat."""

    def __init__(self, in_channels, num_bottle=1, expansion=1, shortcut=True, norm="bn", act="silu", attn=None):
        super().__init__()
        if attn is not None:
            raise NotImplementedError("e-yolox CSPLayer: no attention (the shipped configs)")
        in_ch = in_channels // 4
        num_conv = num_bottle // 2 if num_bottle > 2 else 1
        self.conv1 = BaseConv(in_channels, in_ch, 1, stride=1, norm=norm, act=act)
        self.conv2 = BaseConv(in_channels, in_ch, 1, stride=1, norm=norm, act=act)
        self.conv3 = nn.Sequential(*[Bottleneck(in_ch, in_ch, stride=1, shortcut=True, expansion=2, norm=norm, act=act) for _ in range(num_conv)])
        self.conv4 = nn.Sequential(*[Bottleneck(in_ch, in_ch, stride=1, shortcut=True, expansion=2, norm=norm, act=act) for _ in range(num_conv)])
        self.nonlinearity = get_activation(act)
        self.use_attn = False

    def emit(self, g, x):
        x_1, x_2 = emit_pair(g, x, self.conv1, self.conv2)
        x_3 = x_2
        for m in self.conv3:
            x_3 = m.emit(g, x_3)
        x_4 = x_3
        for m in self.conv4:
            x_4 = m.emit(g, x_4)
        return g.concat([x_1, x_2, x_3, x_4])


class ECMNet(HipModule):
    """ecmnet.py:17-88: MobileNext + CSPNet + inverted bottleneck, Focus stem."""
    stem_kind = "focus"

    def __init__(self, depths=(3, 9, 9, 3), channels=(64, 128, 256, 512, 1024), out_features=("stage2", "stage3", "stage4"),
                 norm="bn", act="silu"):
        super().__init__()
        assert out_features, "please provide output features of CSPMobileNext!"
        self.out_features = out_features
        self.stem = Focus(3, channels[0], ksize=3, norm=norm, act=act)
        expand = 0.5
        self.stage1 = nn.Sequential(
            BaseConv(channels[0], channels[1], 3, 2, norm=norm, act=act),
            CSPLayer(channels[1], num_bottle=depths[0], expansion=expand, norm=norm, act=act, attn=None),
        )
        self.stage2 = nn.Sequential(
            BaseConv(channels[1], channels[2], 3, 2, norm=norm, act=act),
            CSPLayer(channels[2], num_bottle=depths[1], expansion=expand, norm=norm, act=act, attn=None),
        )
        self.stage3 = nn.Sequential(
            BaseConv(channels[2], channels[3], 3, 2, norm=norm, act=act),
            CSPLayer(channels[3], num_bottle=depths[2], expansion=expand, norm=norm, act=act, attn=None),
        )
        self.stage4 = nn.Sequential(
            BaseConv(channels[3], channels[4], 3, 2, norm=norm, act=act),
            SPPBottleneck(channels[4], channels[4], norm=norm, act=act),
            CSPLayer(channels[4], num_bottle=depths[3], expansion=expand, shortcut=False, norm=norm, act=act, attn=None),
        )

    def emit(self, g, image_act):
        outputs = {}
        x = self.stem.emit(g, image_act)
        outputs["stem"] = x
        for name in ("stage1", "stage2", "stage3", "stage4"):
            for m in getattr(self, name):
                x = m.emit(g, x)
            outputs[name] = x
        if len(self.out_features) <= 1:
            return x
        return [v for k, v in outputs.items() if k in self.out_features]


class AL_PAFPN(HipModule):
    """pafpn_al.py:7-93.  Only proceeds 3-level input (stage2, stage3, stage4); bicubic x2 upsampling."""

    def __init__(self, depths=(1, 1, 1, 1), in_channels=(256, 512, 1024), norm="bn", act="silu"):
        super().__init__()
        c = in_channels
        self.shrink_conv1 = BaseConv(c[2], c[1], 1, 1, norm=norm, act=act)
        self.shrink_conv2 = BaseConv(c[2], c[1], 1, 1, norm=norm, act=act)
        self.shrink_conv3 = BaseConv(c[1], c[0], 1, 1, norm=norm, act=act)
        self.shrink_conv4 = BaseConv(c[1], c[0], 1, 1, norm=norm, act=act)
        self.upsample = nn.Upsample(scale_factor=2, mode="bicubic")
        self.p5_p4 = CSPLayer(c[1], num_bottle=depths[0], shortcut=False, norm=norm, act=act)
        self.p4_p3 = CSPLayer(c[0], num_bottle=depths[0], shortcut=False, norm=norm, act=act)
        self.downsample_conv1 = BaseConv(int(c[0]), int(c[0]), 3, 2, norm=norm, act=act)
        self.downsample_conv2 = BaseConv(int(c[1]), int(c[1]), 3, 2, norm=norm, act=act)
        self.n3_n4 = CSPLayer(c[1], num_bottle=depths[0], shortcut=False, norm=norm, act=act)
        self.n4_n5 = CSPLayer(c[2], num_bottle=depths[0], shortcut=False, norm=norm, act=act)

    def emit(self, g, inputs):
        c3, c4, c5 = inputs
        p5_expand = self.shrink_conv1.emit(g, c5)
        p5_upsample = G.BicubicUpsampleOp(g, p5_expand).out
        p4 = self.shrink_conv2.emit(g, g.concat([p5_upsample, c4]))
        p4 = self.p5_p4.emit(g, p4)
        p4_expand = self.shrink_conv3.emit(g, p4)
        p4_upsample = G.BicubicUpsampleOp(g, p4_expand).out
        p3 = self.shrink_conv4.emit(g, g.concat([p4_upsample, c3]))
        p3 = self.p4_p3.emit(g, p3)
        n3 = p3
        n4 = self.n3_n4.emit(g, g.concat([self.downsample_conv1.emit(g, n3), p4_expand]))
        n5 = self.n4_n5.emit(g, g.concat([self.downsample_conv2.emit(g, n4), p5_expand]))
        return (n3, n4, n5)
