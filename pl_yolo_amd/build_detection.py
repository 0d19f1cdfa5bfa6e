"""Model+loss provider with the reference's plugin API
(reference PL_Modules/build_detection.py:23-144).

    build_model(cfg_models, num_classes) -> OneStageD
    OneStageD(backbone, neck, head, loss).forward(x, labels=None)

Plugin factories are looked up by the YAML `name` exactly like the reference (which
uses eval(name)); unknown names raise NameError, missing YAML keys KeyError.  The
returned module exposes .backbone/.neck/.head/.loss with the reference's state_dict
key layout, so `LitDetection` (PL_Modules/pl_detection.py) can use it unchanged as
`self.model`: training_step -> model(imgs, labels) -> dict of losses whose 'loss'
entry back-propagates into ordinary nn.Parameter .grad tensors.
"""
import os

import torch
import torch.nn as nn

from ._lib import PlyoloError
from .backbones import CSPDarkNet, EELAN
from .necks import CSPPAFPN, YOLOv7NECK
from .heads import DecoupledHead, ImplicitHead
from .losses import YOLOXLoss, YOLOv7Loss
from .eyolox import ECMNet, AL_PAFPN
from . import runner as R


def build_model(cfg_models, num_classes):
    cb = cfg_models['backbone']
    cn = cfg_models['neck']
    ch = cfg_models['head']
    cl = cfg_models['loss']
    backbone = _plugin(cb['name'])(cb)
    neck = _plugin(cn['name'])(cn)
    head = _plugin(ch['name'])(ch, num_classes)
    loss = _plugin(cl['name'])(cl, num_classes)
    # optional key of the model YAML (the reference has no such key: it always computes in fp32, train.py:40-42):
    #   compute_dtype: bf16 | fp32      -- bf16 = the MFMA kernels (default), fp32 = the parity mode
    return OneStageD(backbone, neck, head, loss, compute_dtype=cfg_models.get('compute_dtype'))


class OneStageD(nn.Module):
    """backbone -> neck -> head -> (loss if labels is not None), executed as HIP launch
    plans on one MI355X.  `compute_dtype`: "bf16" (MFMA path, default) or "fp32"
    (parity mode); env PLYOLO_DTYPE overrides the default."""

    def __init__(self, backbone=None, neck=None, head=None, loss=None, compute_dtype=None):
        super().__init__()
        self.backbone = backbone
        self.neck = neck
        self.head = head
        self.loss = loss
        if compute_dtype is not None and compute_dtype not in ("bf16", "fp32"):
            raise PlyoloError("compute_dtype must be 'bf16' or 'fp32' (got %r)" % (compute_dtype,))
        self.compute_dtype = compute_dtype or os.environ.get("PLYOLO_DTYPE", "bf16")
        self.__dict__['_runner'] = None
        self.__dict__['_ddp'] = None

    def runner(self):
        r = self.__dict__.get('_runner')
        want = self.compute_dtype
        if r is None or r.dtype_name != want:
            if want not in ("bf16", "fp32"):
                raise PlyoloError("compute_dtype must be 'bf16' or 'fp32'")
            r = R.DetectorRunner(self, want)
            r.dtype_name = want
            r.ddp = self.__dict__.get('_ddp')   # the gradient exchange belongs to the model (pl_yolo_amd.ddp.attach), not to one runner
            self.__dict__['_runner'] = r
        return r

    def fuse(self):
        """Inference export: fold every BaseConv's BatchNorm into its convolution and collapse every RepConv into its single
        3x3 convolution (the reference provides the per-module pieces -- BaseConv.fuseforward, network_blocks.py:39-40;
        RepConv.fuse_repvgg_block, yolov7_neck.py:288-348 -- this walks the tree).  The fused model is eval-only; its traced
        plans are rebuilt on the next call."""
        from .layers import BaseConv
        from .necks import RepConv
        for m in self.modules():
            if isinstance(m, RepConv):
                m.fuse_repvgg_block()
            elif isinstance(m, BaseConv):
                m.fuse()
        self.__dict__['_runner'] = None
        return self.eval()

    def forward(self, x, labels=None):
        r = self.runner()
        if labels is None:                     # list of raw NCHW head maps (build_detection.py:51-52)
            if self.training and torch.is_grad_enabled():
                return R.maps_step(r, x)       # differentiable w.r.t. the parameters
            return r.forward_maps(x)
        if not self.training:
            return r.forward_eval(x)           # [B, A, 5+C]: x1,y1,x2,y2,sig(obj),sig(cls) (yolox_loss.py:25-36)
        out = R.train_step(r, x, labels)       # fp32 loss vector, differentiable
        if isinstance(self.loss, YOLOv7Loss):  # yolov7_loss.py:150-153 returns {"loss": tensor[1]}
            return {"loss": out[0:1]}
        return self.loss.loss_dict(out)


# ---- plugin registry (names as in the reference YAMLs) --------------------------
def cspdarknet(cfg):
    return CSPDarkNet(cfg['depths'], cfg['channels'], cfg['outputs'], cfg['norm'], cfg['act'])


def eelan(cfg):
    return EELAN(cfg['depths'], cfg['channels'], cfg['outputs'], cfg['norm'], cfg['act'])


def yolov7neck(cfg):
    return YOLOv7NECK(cfg['depths'], cfg['channels'], cfg['norm'], cfg['act'], repconv=bool(cfg.get('repconv', False)))


def implicit_head(cfg, num_classes):
    return ImplicitHead(num_classes, cfg['num_anchor'], cfg['channels'])


def yolov7(cfg, num_classes):
    return YOLOv7Loss(num_classes, cfg['stride'], cfg['anchors'])


def ecmnet(cfg):
    return ECMNet(cfg['depths'], cfg['channels'], cfg['outputs'], cfg['norm'], cfg['act'])


def al_pafpn(cfg):
    return AL_PAFPN(cfg['depths'], cfg['channels'], cfg['norm'], cfg['act'])


def csppafpn(cfg):
    return CSPPAFPN(cfg['depths'], cfg['channels'], cfg['norm'], cfg['act'])


def none(cfg):
    return None


def decoupled_head(cfg, num_classes):
    return DecoupledHead(num_classes, cfg['num_anchor'], cfg['channels'], cfg['norm'], cfg['act'])


def yolox(cfg, num_classes):
    return YOLOXLoss(num_classes, cfg['stride'])


_REGISTRY = {f.__name__: f for f in (cspdarknet, eelan, ecmnet, csppafpn, yolov7neck, al_pafpn, none, decoupled_head, implicit_head, yolox, yolov7)}
# reference plugins that exist upstream but are outside this build's hot path
_KNOWN_UNBUILT = ("cspmobilenext", "shufflenetv2", "mobilenetv3s", "mobilenetv3l", "vision_transformer", "swin_transformer")


def _plugin(name):
    if name in _REGISTRY:
        return _REGISTRY[name]
    if name in _KNOWN_UNBUILT:
        raise NotImplementedError("plugin '%s' is part of the reference but has no HIP path in this build yet" % name)
    raise NameError("name '%s' is not defined" % name)
