"""CSPDarkNet backbone (reference models/backbones/darknet_csp.py:15-75)."""
from torch import nn

from .layers import Focus, BaseConv, CSPLayer, SPPBottleneck, HipModule


class CSPDarkNet(HipModule):
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
