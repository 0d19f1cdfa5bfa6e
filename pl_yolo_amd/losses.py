"""Loss plugins (reference models/losses/yolox/yolox_loss.py:7-228, models/losses/yolov7/yolov7_loss.py:9-415).

Like the reference classes they override __call__ (module hooks never fire on them) and switch between the training branch and
the eval decode on `self.training`.  Two ways in:

  * `emit(g, head_buffers, training)`: the detector's launch plan (OneStageD.forward) -- the hot path; the loss launches are
    recorded behind the head's prediction convolutions and read the level-major raw tensor those wrote;
  * `loss(inputs, labels)`: the reference's own contract (`self.loss(self.head(...), labels)`, PL_Modules/build_detection.py:46-53)
    on the caller's list of raw NCHW head maps -- train: the reference's dict, differentiable w.r.t. the maps (one autograd node
    around plyolo_yolox_loss_fwd / _bwd, plyolo_yolov7_loss_fwd / _bwd); eval: the decoded [B, A, 5+C] tensor
    (plyolo_yolox_eval_decode / plyolo_yolov7_eval_decode).  API edge, not the hot path: the maps are staged NCHW -> level-major NHWC
    fp32 and the gradient comes back the same way.  The reference's side effects on its arguments are kept: YOLOXLoss leaves DECODED
    boxes in channels 0..3 of the caller's maps (its decode writes through a view), YOLOv7Loss replaces the entries of the caller's list.

The arithmetic is in csrc/yolox_loss.hip / csrc/yolov7_loss.hip; there is no CPU path (a CPU tensor raises PlyoloError)."""
import ctypes as C

import torch
import torch.nn as nn

from . import graph as G
from ._lib import F32, call, PlyoloError


class _Edge:
    """What HeadBuffers / V7HeadBuffers need from a Graph when the loss runs on a caller's maps: a device, and the fp32 gradient form
    (`draw`: the loss arithmetic is fp32 in every mode; the bf16 [rows,16] / [rows,C8] matrices only exist to feed the MFMA head backward)."""

    def __init__(self, device):
        self.device, self.dtype = device, F32


class _EdgeCache(dict):
    """Device buffers of the stand-alone loss calls, per (map shapes, label rows, mode): never copied or pickled with the module
    (ModelEMA deep-copies the detector, utils/ema.py:41)."""

    def __deepcopy__(self, memo):
        return _EdgeCache()

    def __reduce__(self):
        return (_EdgeCache, ())


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _check_maps(inputs, nch, who):
    if not isinstance(inputs, (list, tuple)) or not inputs:
        raise PlyoloError("%s expects the list of raw head maps, got %r" % (who, type(inputs)))
    B = None
    for m in inputs:
        if not torch.is_tensor(m) or m.dim() != 4:
            raise PlyoloError("%s expects NCHW head maps [B, %d, h, w], got %r" % (who, nch, type(m) if not torch.is_tensor(m) else tuple(m.shape)))
        if not m.is_cuda:
            raise PlyoloError("pl_yolo_amd runs on an MI355X device tensor (got a %s tensor); there is no CPU path" % m.device.type)
        if m.shape[1] != nch:
            raise PlyoloError("%s: head map with %d channels, expected n_anchors*(5+num_classes) = %d" % (who, m.shape[1], nch))
        if B is None:
            B = m.shape[0]
        elif m.shape[0] != B:
            raise PlyoloError("%s: head maps of different batch sizes" % who)
    return B


def _check_labels(labels, B, who):
    if not torch.is_tensor(labels) or labels.dim() != 3 or labels.shape[2] != 5 or labels.shape[0] != B:
        raise PlyoloError("%s: labels must be [B, M, 5] rows (cls, cx, cy, w, h); got %s" % (who, tuple(labels.shape) if torch.is_tensor(labels) else type(labels)))
    if not labels.is_cuda:
        raise PlyoloError("pl_yolo_amd runs on an MI355X device tensor (labels are on %s); there is no CPU path" % labels.device.type)


def _stage(hd, maps):
    """NCHW maps -> the level-major raw tensor (level l = dense NHWC [B, h, w, nch] at row lvl_row[l])."""
    rawv = hd.raw.view(hd.rows, hd.nch)
    for (h, w), r0, m in zip(hd.sizes, hd.lvl_row, maps):
        n = hd.B * h * w
        rawv[r0:r0 + n].copy_(m.detach().permute(0, 2, 3, 1).reshape(n, hd.nch))


def _stage_labels(hd, labels):
    hd.labels.zero_()
    if labels.shape[1]:
        hd.labels.view(hd.B, -1, 5)[:, :labels.shape[1]].copy_(labels.detach())


def _map_grads(hd, dtypes):
    drawv = hd.draw.view(hd.rows, hd.nch)
    out = []
    for (h, w), r0, dt in zip(hd.sizes, hd.lvl_row, dtypes):
        n = hd.B * h * w
        out.append(drawv[r0:r0 + n].view(hd.B, h, w, hd.nch).permute(0, 3, 1, 2).to(dt).contiguous())
    return out


def _decode_boxes_in_place(maps, strides):
    """The side effect of the reference's YOLOXLoss.decode (yolox_loss.py:204-219) on the CALLER's tensors: `pred.view(...).permute(0, 1, 3,
    4, 2).reshape(B, h*w, -1)` of an NCHW map with one anchor is a VIEW (h and w merge), so `pred[..., :2] = (pred[..., :2] + grid) *
    stride` and `pred[..., 2:4] = exp(pred[..., 2:4]) * stride` write THROUGH into channels 0..3 of the head maps it was handed -- after
    `loss(maps, labels)` the caller's maps hold decoded boxes (cx, cy, w, h in pixels), in training and in eval mode alike.  Mirrored here
    (outside autograd: the returned gradient is the one with respect to the raw values, which is where the reference's CopySlices
    node sends it too).  Grid: the reference builds it with meshgrid(arange(h), arange(w), indexing='xy') and re-views the (w, h, 2) stack
    as (h, w, 2) -- cell k = gy*w + gx gets (k mod h, k div h), which is (gx, gy) on square maps only (the non-square quirk the loss kernels
    reproduce as well, loss fixture D)."""
    with torch.no_grad():
        for m, s in zip(maps, strides):
            h, w = m.shape[2], m.shape[3]
            k = torch.arange(h * w, device=m.device)
            gx = (k % h).view(1, h, w).to(m.dtype)
            gy = torch.div(k, h, rounding_mode="floor").view(1, h, w).to(m.dtype)
            m[:, 0] = (m[:, 0] + gx) * s
            m[:, 1] = (m[:, 1] + gy) * s
            m[:, 2:4] = torch.exp(m[:, 2:4]) * s


def _check_in_place_allowed(maps, who):
    for m in maps:
        if m.requires_grad and m.is_leaf and torch.is_grad_enabled():
            # what torch says when the reference's decode hits such a tensor
            raise PlyoloError("%s: a view of a leaf Variable that requires grad is being used in an in-place operation (the reference "
                              "decodes the box channels of the head maps it is given in place, yolox_loss.py:214-217): pass the head's "
                              "outputs (non-leaf tensors), or detached maps" % who)


class _LossFn(torch.autograd.Function):
    """One autograd node around the loss launches of a plugin: forward = stage + loss forward (fp32 loss vector out), backward =
    d(sum_i gout[i] * losses[i]) / d(maps).  The loss workspace (assignments) of a session belongs to its LAST forward: a backward that
    finds another forward in between re-runs its own forward from the raw values it set aside first (API edge: correctness over speed).
    The maps themselves are NOT saved: YOLOXLoss decodes the caller's box channels in place afterwards, as the reference does."""

    @staticmethod
    def forward(ctx, plugin, hd, labels, *maps):
        _stage(hd, maps)
        _stage_labels(hd, labels)
        plugin._fwd(hd)
        hd.generation = getattr(hd, "generation", 0) + 1
        ctx.plugin, ctx.hd, ctx.generation = plugin, hd, hd.generation
        ctx.dtypes = [m.dtype for m in maps]
        ctx.save_for_backward(hd.raw.clone(), hd.labels.clone())
        return hd.losses.clone()

    @staticmethod
    def backward(ctx, gout):
        plugin, hd = ctx.plugin, ctx.hd
        if ctx.generation != hd.generation:
            raw, labels = ctx.saved_tensors
            hd.raw.copy_(raw)
            hd.labels.copy_(labels)
            plugin._fwd(hd)
            hd.generation += 1
            ctx.generation = hd.generation
        hd.gout.zero_()
        hd.gout[:gout.numel()].copy_(gout.reshape(-1).float())
        plugin._bwd(hd)
        return (None, None, None) + tuple(_map_grads(hd, ctx.dtypes))


class YOLOXLoss(nn.Module):
    def __init__(self, num_classes, strides, use_l1=False):
        super().__init__()
        self.num_classes = num_classes
        self.strides = strides
        self.n_anchors = 1
        self.use_l1 = use_l1
        self.__dict__['_edge'] = _EdgeCache()

    # ---- the reference's contract on a caller's maps (yolox_loss.py:20-36; decode :180-228, including its write-through into the box
    # channels of the caller's maps: _decode_boxes_in_place)
    def _edge_buffers(self, maps, M, training):
        dev = maps[0].device
        key = (tuple(tuple(m.shape) for m in maps), M, training, bool(self.use_l1), dev)
        hd = self.__dict__['_edge'].get(key)
        if hd is None:
            if len(maps) != len(self.strides):
                raise PlyoloError("YOLOXLoss: %d head maps for %d strides" % (len(maps), len(self.strides)))
            hd = G.HeadBuffers(_Edge(dev), maps[0].shape[0], self.num_classes, [tuple(m.shape[2:]) for m in maps], list(self.strides), max(M, 1))
            hd.desc.use_l1 = 1 if self.use_l1 else 0
            if training:
                hd.alloc_loss()
            else:
                hd.alloc_eval()
            if len(self.__dict__['_edge']) >= 8:      # a few shapes stay resident (multi-scale training); not a cache to grow without bound
                self.__dict__['_edge'].pop(next(iter(self.__dict__['_edge'])))
            self.__dict__['_edge'][key] = hd
        return hd

    def _fwd(self, hd):
        call("plyolo_yolox_loss_fwd", C.byref(hd.desc), hd.raw.data_ptr(), hd.labels.data_ptr(), hd.fg.data_ptr(), hd.mgt.data_ptr(),
             hd.miou.data_ptr(), hd.losses.data_ptr(), hd.ws.data_ptr(), hd.ws_bytes, _stream())

    def _bwd(self, hd):
        call("plyolo_yolox_loss_bwd", C.byref(hd.desc), hd.raw.data_ptr(), hd.labels.data_ptr(), hd.fg.data_ptr(), hd.mgt.data_ptr(),
             hd.miou.data_ptr(), hd.losses.data_ptr(), hd.gout.data_ptr(), hd.draw.data_ptr(), None, None, 0, _stream())

    def loss_dict(self, out):
        """fp32 loss vector {loss, loss_iou, loss_obj, loss_cls, num_fg, num_gt, proportion, loss_l1} -> the reference's dict
        (yolox_loss.py:150-166)."""
        return {
            "loss": out[0],
            "loss_iou": out[1],
            "loss_obj": out[2],
            "loss_cls": out[3],
            "loss_l1": out[7] if self.use_l1 else 0.0,   # the python float 0.0 without use_l1, as the reference (yolox_loss.py:159-160)
            "proportion": out[6].detach(),
        }

    def __call__(self, inputs, labels=None):
        nch = self.n_anchors * (5 + self.num_classes)
        B = _check_maps(inputs, nch, "YOLOXLoss")
        maps = list(inputs)
        _check_in_place_allowed(maps, "YOLOXLoss")
        if not self.training:
            hd = self._edge_buffers(maps, 1, False)
            _stage(hd, maps)
            call("plyolo_yolox_eval_decode", C.byref(hd.desc), hd.raw.data_ptr(), hd.eval_out.data_ptr(), _stream())
            _decode_boxes_in_place(maps, self.strides)
            return hd.eval_out.view(hd.eval_shape).clone()     # x1, y1, x2, y2, sig(obj), sig(cls) (yolox_loss.py:25-36)
        _check_labels(labels, B, "YOLOXLoss")
        hd = self._edge_buffers(maps, labels.shape[1], True)
        out = _LossFn.apply(self, hd, labels, *maps)
        _decode_boxes_in_place(maps, self.strides)
        return self.loss_dict(out)

    def emit(self, g, head_buffers, training):
        if training:
            # use_l1 (yolox_loss.py:128-135,157-158; a constructor argument the plugin factory never sets): the L1 term of the
            # raw box outputs rides the same loss / gradient kernels
            head_buffers.desc.use_l1 = 1 if self.use_l1 else 0
            G.YoloxLossOp(g, head_buffers)
        else:
            G.YoloxEvalDecodeOp(g, head_buffers)


class YOLOv7Loss(nn.Module):
    """YOLOv7 loss plugin (reference models/losses/yolov7/yolov7_loss.py:9-415).  Eval decode
    (:50-78) runs on the device (csrc/yolox_loss.hip: k_v7_eval_decode), the training branch
    (find_3_positive / build_targets / CIoU + obj + cls losses, :80-368) in csrc/yolov7_loss.hip."""

    def __init__(self, num_classes, strides, anchors, label_smoothing=0, focal_g=0.0):
        super().__init__()
        self.num_classes = num_classes
        self.strides = strides
        self.anchors_list = anchors
        self.nl = len(strides)
        self.na = len(anchors[0])
        self.ch = 5 + num_classes
        self.__dict__['_edge'] = _EdgeCache()

    def _edge_buffers(self, maps, M, training):
        dev = maps[0].device
        key = (tuple(tuple(m.shape) for m in maps), M, training, dev)
        hd = self.__dict__['_edge'].get(key)
        if hd is None:
            if len(maps) != self.nl:
                raise PlyoloError("YOLOv7Loss: %d head maps for %d strides" % (len(maps), self.nl))
            hd = G.V7HeadBuffers(_Edge(dev), maps[0].shape[0], self.num_classes, self.na, [tuple(m.shape[2:]) for m in maps],
                                 list(self.strides), self.anchors_list)
            if training:
                hd.alloc_loss(M)
            else:
                hd.alloc_eval()
            if len(self.__dict__['_edge']) >= 8:
                self.__dict__['_edge'].pop(next(iter(self.__dict__['_edge'])))
            self.__dict__['_edge'][key] = hd
        return hd

    def _fwd(self, hd):
        call("plyolo_yolov7_loss_fwd", C.byref(hd.desc), hd.raw.data_ptr(), hd.labels.data_ptr(), hd.losses.data_ptr(),
             hd.ws.data_ptr(), hd.ws_bytes, _stream())

    def _bwd(self, hd):
        call("plyolo_yolov7_loss_bwd", C.byref(hd.desc), hd.raw.data_ptr(), hd.labels.data_ptr(), hd.gout.data_ptr(),
             hd.draw.data_ptr(), hd.ws.data_ptr(), hd.ws_bytes, _stream())

    def __call__(self, inputs, targets=None):
        B = _check_maps(inputs, self.na * self.ch, "YOLOv7Loss")
        maps = list(inputs)
        if isinstance(inputs, list):
            # the reference REPLACES the entries of the caller's list with the [B, na, h, w, ch] views it works on (yolov7_loss.py:43-47)
            for i, m in enumerate(maps):
                inputs[i] = m.view(B, self.na, self.ch, m.shape[2], m.shape[3]).permute(0, 1, 3, 4, 2).contiguous()
        if not self.training:
            hd = self._edge_buffers(maps, 1, False)
            _stage(hd, maps)
            for l, ((h, w), s) in enumerate(zip(hd.sizes, hd.strides)):
                call("plyolo_yolov7_eval_decode", hd.raw.data_ptr() + hd.lvl_row[l] * hd.nch * 4, hd.B, h, w, hd.na, hd.nc, int(s),
                     hd.anchor_t.data_ptr() + l * hd.na * 2 * 4, hd.eval_out.data_ptr(), hd.A, hd.lvl_off[l], _stream())
            return hd.eval_out.view(hd.eval_shape).clone()
        _check_labels(targets, B, "YOLOv7Loss")
        hd = self._edge_buffers(maps, max(int(targets.shape[1]), 1), True)
        out = _LossFn.apply(self, hd, targets, *maps)
        return {"loss": out[0:1]}      # yolov7_loss.py:150-153 returns {"loss": tensor[1]}

    def emit(self, g, head_buffers, training):
        if training:
            G.YoloV7LossOp(g, head_buffers)
        else:
            G.YoloV7EvalDecodeOp(g, head_buffers)
