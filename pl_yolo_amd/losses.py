"""YOLOX loss plugin (reference models/losses/yolox/yolox_loss.py:7-228).

Like the reference class it overrides __call__ (module hooks never fire on it) and
switches between the training branch and the eval decode on `self.training`.
The arithmetic is in csrc/yolox_loss.hip; this object only carries the config."""
import torch.nn as nn

from . import graph as G


class YOLOXLoss(nn.Module):
    def __init__(self, num_classes, strides, use_l1=False):
        super().__init__()
        self.num_classes = num_classes
        self.strides = strides
        self.n_anchors = 1
        self.use_l1 = use_l1

    def __call__(self, inputs, labels):
        raise RuntimeError("YOLOXLoss is driven by the detector's launch plan (OneStageD.forward); "
                           "it has no stand-alone tensor path")

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

    def __call__(self, inputs, targets):
        raise RuntimeError("YOLOv7Loss is driven by the detector's launch plan (OneStageD.forward)")

    def emit(self, g, head_buffers, training):
        if training:
            G.YoloV7LossOp(g, head_buffers)
        else:
            G.YoloV7EvalDecodeOp(g, head_buffers)
