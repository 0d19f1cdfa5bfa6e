"""YOLOX decoupled head (reference models/heads/decoupled_head.py:7-95)."""
import math
import os

import torch
import torch.nn as nn

from . import graph as G
from .layers import emit_pair, BaseConv, HipModule


# Launch lanes (graph.Graph.on_lane) of the head levels: PLYOLO_HEAD_LANES = one lane per level, the last entry repeats.
# Default "0,0,2": the 80x80 and the 40x40 level on the main lane, the 20x20 level on the side lane that also carries the
# neck's bottom-up path (necks.py) -- the big level-0 kernels then run beside the short, latency-bound 40x40 / 20x20 kernels of
# the neck instead of after them.  "0,2,2" is the round-2 layout (with PLYOLO_NECK_LANE=0), "0,0,0" keeps every level on lane 0.
_LEGACY = os.environ.get("PLYOLO_HEAD_ONE_LANE")
_HEAD_LANES = [int(v) for v in os.environ.get("PLYOLO_HEAD_LANES", {"0": "0,2,3", "1": "0,2,2", "2": "0,0,0"}.get(_LEGACY, "0,0,2")).split(",")]


_HEAD_LANES_FWD = [int(v) for v in os.environ["PLYOLO_HEAD_LANES_FWD"].split(",")] if os.environ.get("PLYOLO_HEAD_LANES_FWD") else _HEAD_LANES


def head_lane(k, fwd=False):
    t = _HEAD_LANES_FWD if fwd else _HEAD_LANES
    return t[k] if k < len(t) else t[-1]


class DecoupledHead(HipModule):
    def __init__(self, num_classes=80, n_anchors=1, in_channels=None, norm="bn", act="silu"):
        super().__init__()
        if n_anchors != 1:
            raise NotImplementedError("the YOLOX loss kernels assume one anchor per cell (yolox_loss.py:12)")
        self.n_anchors = n_anchors
        self.num_classes = num_classes
        ch = self.n_anchors * self.num_classes
        self.stems = nn.ModuleList()
        self.cls_convs = nn.ModuleList()
        self.cls_preds = nn.ModuleList()
        self.reg_convs = nn.ModuleList()
        self.reg_preds = nn.ModuleList()
        self.obj_preds = nn.ModuleList()
        for i in range(len(in_channels)):
            # the stem ignores `norm` in the reference (decoupled_head.py:29-31)
            self.stems.append(BaseConv(in_channels[i], in_channels[0], ksize=1, stride=1, act=act))
            self.cls_convs.append(nn.Sequential(
                BaseConv(in_channels[0], in_channels[0], ksize=3, stride=1, norm=norm, act=act),
                BaseConv(in_channels[0], in_channels[0], ksize=3, stride=1, norm=norm, act=act)))
            self.cls_preds.append(nn.Conv2d(in_channels[0], ch, kernel_size=(1, 1), stride=(1, 1), padding=0))
            self.reg_convs.append(nn.Sequential(
                BaseConv(in_channels[0], in_channels[0], ksize=3, stride=1, norm=norm, act=act),
                BaseConv(in_channels[0], in_channels[0], ksize=3, stride=1, norm=norm, act=act)))
            self.reg_preds.append(nn.Conv2d(in_channels[0], self.n_anchors * 4, kernel_size=(1, 1), stride=(1, 1), padding=0))
            self.obj_preds.append(nn.Conv2d(in_channels[0], self.n_anchors * 1, kernel_size=(1, 1), stride=(1, 1), padding=0))
        self.initialize_biases(1e-2)

    def initialize_biases(self, prior_prob):
        v = -math.log((1 - prior_prob) / prior_prob)
        for conv in list(self.cls_preds) + list(self.obj_preds):
            with torch.no_grad():
                conv.bias.fill_(v)

    def emit(self, g, inputs, head_buffers):
        """Writes the raw predictions of every level into `head_buffers.raw`
        (channel order reg(4), obj(1), cls(C) -- decoupled_head.py:93)."""
        # the levels are independent from the stem conv to the prediction convs (and back, in the backward plan): each runs
        # on the lane head_lane() names; graph.record_ops() orders them against the neck from the tensors they touch
        for k, x in enumerate(inputs):
            with g.on_lanes(head_lane(k, True), head_lane(k)):
                x = self.stems[k].emit(g, x)
                # the first conv of the cls and of the reg branch read the same stem output: one merged conv
                cls_feat, reg_feat = emit_pair(g, x, self.cls_convs[k][0], self.reg_convs[k][0])
                for m in list(self.cls_convs[k])[1:]:
                    cls_feat = m.emit(g, cls_feat)
                for m in list(self.reg_convs[k])[1:]:
                    reg_feat = m.emit(g, reg_feat)
                G.HeadPredOp(g, head_buffers, k, cls_feat, reg_feat, self.cls_preds[k], self.reg_preds[k], self.obj_preds[k])
        return head_buffers


class ImplicitA(nn.Module):
    def __init__(self, channel, mean=0., std=.02):
        super().__init__()
        self.channel, self.mean, self.std = channel, mean, std
        self.implicit = nn.Parameter(torch.zeros(1, channel, 1, 1))
        nn.init.normal_(self.implicit, mean=self.mean, std=self.std)


class ImplicitM(nn.Module):
    def __init__(self, channel, mean=0., std=.02):
        super().__init__()
        self.channel, self.mean, self.std = channel, mean, std
        self.implicit = nn.Parameter(torch.ones(1, channel, 1, 1))
        nn.init.normal_(self.implicit, mean=self.mean, std=self.std)  # mean 0 as coded upstream (implicit_head.py:53-59)


class ImplicitHead(HipModule):
    """YOLOv7 head (models/heads/implicit_head.py:5-36): per level  im * conv1x1(x + ia)."""

    def __init__(self, num_classes, num_anchors, in_channels):
        super().__init__()
        self.n_anchors = num_anchors
        self.num_classes = num_classes
        ch = self.n_anchors * (5 + num_classes)
        self.conv = nn.ModuleList()
        self.ia = nn.ModuleList()
        self.im = nn.ModuleList()
        for i in range(len(in_channels)):
            self.ia.append(ImplicitA(in_channels[i]))
            self.conv.append(nn.Conv2d(in_channels[i], ch, 1))
            self.im.append(ImplicitM(ch))

    def emit(self, g, inputs, head_buffers):
        for k, x in enumerate(inputs):
            G.ImplicitHeadOp(g, head_buffers, k, x, self.conv[k], self.ia[k].implicit, self.im[k].implicit)
        return head_buffers
