"""CSPDarkNet backbone (reference models/backbones/darknet_csp.py:15-75)."""
from torch import nn

from .layers import emit_pair, Focus, BaseConv, CSPLayer, SPPBottleneck, HipModule


class CSPDarkNet(HipModule):
    stem_kind = "focus"

    """stem (Focus) + four stages; returns the features named in `out_features`."""

    def __init__(self, depths=(3, 9, 9, 3), channels=(64, 128, 256, 512, 1024),
                 out_features=("stage2", "stage3", "stage4"), norm="bn", act="silu"):
        super().__init__()
        assert out_features, "please provide output features of Darknet!"
        self.out_features = out_features
        self.stem = Focus(3, channels[0], ksize=3, norm=norm, act=act)
        self.stage1 = nn.Sequential(
            BaseConv(channels[0], channels[1], 3, 2, norm=norm, act=act),
            CSPLayer(channels[1], channels[1], num_bottle=depths[0], norm=norm, act=act),
        )
        self.stage2 = nn.Sequential(
            BaseConv(channels[1], channels[2], 3, 2, norm=norm, act=act),
            CSPLayer(channels[2], channels[2], num_bottle=depths[1], norm=norm, act=act),
        )
        self.stage3 = nn.Sequential(
            BaseConv(channels[2], channels[3], 3, 2, norm=norm, act=act),
            CSPLayer(channels[3], channels[3], num_bottle=depths[2], norm=norm, act=act),
        )
        self.stage4 = nn.Sequential(
            BaseConv(channels[3], channels[4], 3, 2, norm=norm, act=act),
            SPPBottleneck(channels[4], channels[4], norm=norm, act=act),
            CSPLayer(channels[4], channels[4], num_bottle=depths[3], shortcut=False, norm=norm, act=act),
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


class ELANLayer(HipModule):
    """The `CSPLayer` of models/backbones/eelan.py:82-123 (ELAN block: 4-way concat)."""

    def __init__(self, in_channel, out_channel, expansion=0.5, num_bottle=1, norm="bn", act="silu"):
        super().__init__()
        hi_channel = int(in_channel * expansion)
        self.num_conv = num_bottle // 2 if num_bottle > 2 else 1
        self.conv1 = BaseConv(in_channel, hi_channel, 1, stride=1, norm=norm, act=act)
        self.conv2 = BaseConv(in_channel, hi_channel, 1, stride=1, norm=norm, act=act)
        self.conv3 = nn.Sequential(*[BaseConv(hi_channel, hi_channel, 3, stride=1, norm=norm, act=act) for _ in range(self.num_conv)])
        self.conv4 = nn.Sequential(*[BaseConv(hi_channel, hi_channel, 3, stride=1, norm=norm, act=act) for _ in range(self.num_conv)])
        self.conv5 = BaseConv(4 * hi_channel, out_channel, 1, stride=1, norm=norm, act=act)

    def emit(self, g, x):
        x_1, x_2 = emit_pair(g, x, self.conv1, self.conv2)
        x_3 = x_2
        for m in self.conv3:
            x_3 = m.emit(g, x_3)
        x_4 = x_3
        for m in self.conv4:
            x_4 = m.emit(g, x_4)
        return self.conv5.emit(g, g.concat([x_1, x_2, x_3, x_4]))


class ELANTransition(HipModule):
    """`Transition` of models/backbones/eelan.py:126-141 (conv1/conv2 ignore norm/act, lines 130-131)."""

    def __init__(self, in_channel, mpk=2, norm="bn", act="silu"):
        super().__init__()
        if mpk != 2:
            raise NotImplementedError("Transition max-pool kernel is 2x2 stride 2")
        self.mp = nn.MaxPool2d(kernel_size=mpk, stride=mpk)
        self.conv1 = BaseConv(in_channel, in_channel // 2, 1, 1)
        self.conv2 = BaseConv(in_channel, in_channel // 2, 1, 1)
        self.conv3 = BaseConv(in_channel // 2, in_channel // 2, 3, 2, norm=norm, act=act)

    def emit(self, g, x):
        from . import graph as G
        x_1 = self.conv1.emit(g, G.MaxPool2x2Op(g, x).out)
        x_2 = self.conv3.emit(g, self.conv2.emit(g, x))
        return g.concat([x_2, x_1])


class EELAN(HipModule):
    """Extended efficient layer aggregation network (models/backbones/eelan.py:15-79)."""

    def __init__(self, depths=(4, 4, 4, 4), channels=(64, 128, 256, 512, 1024),
                 out_features=("stage2", "stage3", "stage4"), norm="bn", act="silu"):
        super().__init__()
        assert out_features, "please provide output features of EELAN!"
        self.out_features = out_features
        self.stem = nn.Sequential(
            BaseConv(3, 32, 3, 1, norm=norm, act=act),
            BaseConv(32, channels[0], 3, 2, norm=norm, act=act),
            BaseConv(channels[0], channels[0], 3, 1, norm=norm, act=act),
        )
        self.stage1 = nn.Sequential(
            BaseConv(channels[0], channels[1], 3, 2, norm=norm, act=act),
            ELANLayer(channels[1], channels[2], expansion=0.5, num_bottle=depths[0], norm=norm, act=act),
        )
        self.stage2 = nn.Sequential(
            ELANTransition(channels[2], mpk=2, norm=norm, act=act),
            ELANLayer(channels[2], channels[3], expansion=0.5, num_bottle=depths[1], norm=norm, act=act),
        )
        self.stage3 = nn.Sequential(
            ELANTransition(channels[3], mpk=2, norm=norm, act=act),
            ELANLayer(channels[3], channels[4], expansion=0.5, num_bottle=depths[2], norm=norm, act=act),
        )
        self.stage4 = nn.Sequential(
            ELANTransition(channels[4], mpk=2, norm=norm, act=act),
            SPPBottleneck(channels[4], channels[4], norm=norm, act=act),
            ELANLayer(channels[4], channels[4], expansion=0.5, num_bottle=depths[3], norm=norm, act=act),
        )
    stem_kind = "rgb"   # consumes the plain NHWC image (CSPDarkNet consumes the Focus gather)

    def emit(self, g, image_act):
        outputs = {}
        x = self.stem[0].emit(g, image_act, need_dgrad=False, cin_pad=image_act.C)
        x = self.stem[1].emit(g, x)
        x = self.stem[2].emit(g, x)
        outputs["stem"] = x
        for name in ("stage1", "stage2", "stage3", "stage4"):
            for m in getattr(self, name):
                x = m.emit(g, x)
            outputs[name] = x
        if len(self.out_features) <= 1:
            return x
        return [v for k, v in outputs.items() if k in self.out_features]
