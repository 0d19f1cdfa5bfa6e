"""GPU input pipeline: the per-image transforms of the reference's data loaders as one HIP launch per batch
(reference models/data/augmentation/data_augments.py:6-141).

    TrainTransform(max_labels, flip_prob, hsv_prob)(image, targets, input_dim) -> (image [3,H,W] fp32 device tensor, labels [max_labels,5])
    ValTransform(swap, legacy, max_labels)(img, targets, input_size)            -> (image, labels)
    preproc(img, input_size)                                                     -> (image, ratio)
    ... and the batch forms `TrainTransform.batch / ValTransform.batch / preproc_batch` (ONE launch for B images).

Images are uint8 HWC BGR DEVICE tensors (decode / upload them however the host likes); the pixel work -- HSV jitter,
mirror, letterbox resize, pad with 114, HWC->CHW fp32 -- runs in csrc/augment.hip (plyolo_preproc_batch).  The label
arithmetic (a few dozen floats per image) stays on the host in numpy, line by line as in the reference, and draws from
python's `random` / numpy's global RNG in the reference's order, so the same seeds take the same decisions.
Mosaic / random-affine / mixup (models/data/mosaic_detection.py) are not built yet."""
import ctypes as C
import random

import numpy as np
import torch

from ._lib import AugImage, PlyoloError, call


def _check_image(img):
    if not torch.is_tensor(img) or not img.is_cuda:
        raise PlyoloError("the input pipeline takes uint8 HWC device tensors (there is no CPU path)")
    if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3:
        raise PlyoloError("expected a uint8 [H, W, 3] BGR image, got %s %s" % (img.dtype, tuple(img.shape)))
    return img.contiguous()


def preproc_batch(images, input_size, flips=None, gains=None):
    """images: list of uint8 [h,w,3] device tensors -> (fp32 [B,3,H,W] device tensor, [ratio per image]).
    flips[i]: mirror image i; gains[i]: None or the three HSV factors (r of data_augments.py:114)."""
    B = len(images)
    if B == 0:
        raise PlyoloError("empty batch")
    imgs = [_check_image(im) for im in images]
    dev = imgs[0].device
    arr = (AugImage * B)()
    ratios = []
    for i, im in enumerate(imgs):
        h, w = int(im.shape[0]), int(im.shape[1])
        r = min(input_size[0] / h, input_size[1] / w)
        ratios.append(r)
        a = arr[i]
        a.src, a.h, a.w, a.dh, a.dw = im.data_ptr(), h, w, int(h * r), int(w * r)
        if a.dh < 1 or a.dw < 1:
            raise PlyoloError("image %dx%d vanishes at input size %s" % (h, w, tuple(input_size)))
        a.flip = int(bool(flips[i])) if flips is not None else 0
        g = gains[i] if gains is not None else None
        a.hsv = int(g is not None)
        if g is not None:
            a.hgain, a.sgain, a.vgain = float(g[0]), float(g[1]), float(g[2])
    table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
    out = torch.empty(B, 3, int(input_size[0]), int(input_size[1]), dtype=torch.float32, device=dev)
    call("plyolo_preproc_batch", table.data_ptr(), B, int(input_size[0]), int(input_size[1]), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    table.record_stream(torch.cuda.current_stream())
    return out, ratios


def preproc(img, input_size, swap=(2, 0, 1)):
    """data_augments.py:88-107."""
    if tuple(swap) != (2, 0, 1):
        raise NotImplementedError("preproc writes CHW (swap (2, 0, 1)), what the detector takes")
    out, r = preproc_batch([img], input_size)
    return out[0], r[0]


def _mirror_boxes(width, boxes):
    boxes[:, 0::2] = width - boxes[:, 2::-2]       # data_augments.py:132
    return boxes


def xyxy2cxcywh(bboxes):
    bboxes[:, 2] = bboxes[:, 2] - bboxes[:, 0]
    bboxes[:, 3] = bboxes[:, 3] - bboxes[:, 1]
    bboxes[:, 0] = bboxes[:, 0] + bboxes[:, 2] * 0.5
    bboxes[:, 1] = bboxes[:, 1] + bboxes[:, 3] * 0.5
    return bboxes


class TrainTransform:
    """data_augments.py:6-48."""

    def __init__(self, max_labels=50, flip_prob=0.5, hsv_prob=1.0):
        self.max_labels, self.flip_prob, self.hsv_prob = max_labels, flip_prob, hsv_prob

    def _decide(self, image, targets, input_dim):
        """Host side of one image: the random decisions and the label arithmetic.  Returns (flip, gains, labels, retry) where
        `retry` means "every box vanished: fall back to the un-augmented image" (data_augments.py:37-41)."""
        h, w = int(image.shape[0]), int(image.shape[1])
        if len(targets) == 0:
            return False, None, np.zeros((self.max_labels, 5), dtype=np.float32)
        targets_process = targets.copy()
        gains, flip = None, False
        if random.random() < self.hsv_prob:
            gains = np.random.uniform(-1, 1, 3) * [0.015, 0.7, 0.4] + 1          # augment_hsv :114
        if random.random() < self.flip_prob:
            flip = True
            targets_process[:, :4] = _mirror_boxes(w, targets_process[:, :4])
        r = min(input_dim[0] / h, input_dim[1] / w)
        targets_process[:, :4] = xyxy2cxcywh(targets_process[:, :4])
        targets_process[:, :4] *= r
        mask_b = np.minimum(targets_process[:, 2], targets_process[:, 3]) > 1
        targets_process = targets_process[mask_b]
        if len(targets_process) == 0:      # :37-41 -- the ORIGINAL image and boxes (the reference mutates `targets` here too)
            gains, flip = None, False
            targets_process = targets
            targets_process[:, :4] = r * targets_process[:, :4]
            targets_process[:, :4] = xyxy2cxcywh(targets_process[:, :4])
        label_process = np.expand_dims(targets_process[:, 4], 1)
        t = np.hstack((label_process, targets_process[:, :4]))
        padded_labels = np.zeros((self.max_labels, 5))
        padded_labels[range(len(t))[: self.max_labels]] = t[: self.max_labels]
        return flip, gains, np.ascontiguousarray(padded_labels, dtype=np.float32)

    def batch(self, images, targets_list, input_dim):
        dec = [self._decide(im, t, input_dim) for im, t in zip(images, targets_list)]
        out, _ = preproc_batch(images, input_dim, flips=[d[0] for d in dec], gains=[d[1] for d in dec])
        return out, np.stack([d[2] for d in dec], 0)

    def __call__(self, image, targets, input_dim):
        out, labels = self.batch([image], [targets], input_dim)
        return out[0], labels[0]


class ValTransform:
    """data_augments.py:51-85."""

    def __init__(self, swap=(2, 0, 1), legacy=False, max_labels=50):
        if legacy:
            raise NotImplementedError("ValTransform(legacy=True) (RGB + ImageNet normalisation) is not built")
        self.swap, self.legacy, self.max_labels = swap, legacy, max_labels

    def _labels(self, targets):
        boxes = xyxy2cxcywh(targets[:, :4].copy())
        labels = np.expand_dims(targets[:, 4].copy(), 1)
        targets_t = np.hstack((labels, boxes))
        padded_labels = np.zeros((self.max_labels, 5))
        padded_labels[range(len(targets_t))[:self.max_labels]] = targets_t[:self.max_labels]
        return padded_labels

    def batch(self, images, targets_list, input_size):
        out, _ = preproc_batch(images, input_size)
        return out, np.stack([self._labels(t) for t in targets_list], 0)

    def __call__(self, img, targets, input_size):
        out, labels = self.batch([img], [targets], input_size)
        return out[0], labels[0]
