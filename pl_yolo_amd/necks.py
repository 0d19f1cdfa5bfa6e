"""CSP-PAFPN neck (reference models/necks/pafpn_csp.py:7-86)."""
import torch.nn as nn

from . import graph as G
from .layers import BaseConv, CSPLayer, HipModule


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
        n3_downsample = self.downsample_conv1.emit(g, n3)
        n4 = self.n3_n4.emit(g, g.concat([n3_downsample, p4_expand]))
        n4_downsample = self.downsample_conv2.emit(g, n4)
        n5 = self.n4_n5.emit(g, g.concat([n4_downsample, p5_expand]))
        return (n3, n4, n5)
