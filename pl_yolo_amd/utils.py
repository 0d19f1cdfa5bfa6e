"""Model summary without thop (reference utils/flops.py:5-9, called by LitDetection.on_train_start, PL_Modules/pl_detection.py:49).

thop counts with forward hooks on the leaf nn.Conv2d / nn.BatchNorm2d modules; here the module tree only DESCRIBES its launches and
those leaves never run, so thop reports 0 operations for a pl_yolo_amd model.  `model_summary` prints the reference's line from the
traced launch graph instead: parameters = every nn.Parameter of the model, operations = 2 x the multiply-accumulates of every
convolution (what the upstream YOLOX tables and BASELINE.md count: 26.69 G for yolox_s at 640x640, 155.29 G for yolox_l)."""
import torch

from . import graph as G
from ._lib import F32, PlyoloError


def _clever(v):
    for unit, scale in (("T", 1e12), ("G", 1e9), ("M", 1e6), ("K", 1e3)):
        if v >= scale:
            return "%.3f%s" % (v / scale, unit)
    return "%.3f" % v


def conv_macs(model, height, width):
    """Multiply-accumulates of one image through backbone -> neck -> head (every convolution, bias-free), from a CPU dry trace:
    nothing is allocated on a device and no kernel runs."""
    if height % 32 or width % 32:
        raise PlyoloError("input size must be a multiple of 32 (got %dx%d)" % (height, width))
    g = G.Graph(F32, False, torch.device("cpu"))
    if getattr(model.backbone, "stem_kind", "focus") == "focus":
        image = g.new_act(1, height // 2, width // 2, 12, "focus")
    else:
        image = g.new_act(1, height, width, 4, "rgb")
    feats = model.backbone.emit(g, image)
    if model.neck is not None:
        feats = model.neck.emit(g, feats)
    if not isinstance(feats, (list, tuple)):
        feats = [feats]
    strides = list(model.loss.strides) if model.loss is not None else [width // f.W for f in feats]
    sizes = [(f.H, f.W) for f in feats]
    if model.head.n_anchors == 1:
        head = G.HeadBuffers(g, 1, model.head.num_classes, sizes, strides, 1)
    else:
        anchors = getattr(model.loss, "anchors_list", None) or [[[10.0, 13.0]] * model.head.n_anchors for _ in feats]
        head = G.V7HeadBuffers(g, 1, model.head.num_classes, model.head.n_anchors, sizes, strides, anchors)
    model.head.emit(g, feats, head)
    macs = 0
    for op in g.ops:
        pcs = [v for v in vars(op).values() if isinstance(v, G.PackedConv)]
        if hasattr(op, "OH"):
            pos = op.OH * op.OW
        elif hasattr(op, "cls_feat"):
            pos = op.cls_feat.H * op.cls_feat.W
        elif hasattr(op, "x") and hasattr(op.x, "H"):
            pos = op.x.H * op.x.W
        else:
            pos = 0
        for pc in pcs:
            macs += pos * sum(int(w.numel()) for (w, _, _) in pc.sources)
        if isinstance(op, G.DwConvUnitOp):      # depthwise weights are not packed
            macs += op.x.H * op.x.W * int(op.w.numel())
    return macs


def model_summary(model, train_size, device=None):
    """Drop-in for utils/flops.py: prints ' ------- params: ... ------- flops: ...' and returns None like the reference (device is accepted
    and ignored: the count is a host-side trace)."""
    params = sum(int(p.numel()) for p in model.parameters())
    flops = 2.0 * conv_macs(model, int(train_size[0]), int(train_size[1]))
    print(" ------- params: %s ------- flops: %s" % (_clever(params), _clever(flops)))
    return None
