"""CPU restatement of the per-image input transforms (TEST ORACLE -- imported by tests only).

Follows /root/reference/models/data/augmentation/data_augments.py:
    TrainTransform.__call__ :6-48, ValTransform.__call__ :51-85, preproc :88-107, augment_hsv :110-126, _mirror :129-133,
    xyxy2cxcywh :136-141
with numpy only.  The three OpenCV calls of that file are RESTATED from OpenCV's published 8-bit algorithms (cv2 is not
installable in the build container): PARITY UNPINNED against cv2 itself, and therefore against the reference for the
pixel values (the label arithmetic is plain numpy and is followed line by line):
  * cv2.resize(..., INTER_LINEAR) on uint8 -- imgproc/resize.cpp: source coordinate (d + 0.5) * scale - 0.5, clamped taps,
    11-bit fixed-point coefficients (saturate_cast<short>(f * 2048)), (S.b + 2^21) >> 22;
  * cv2.cvtColor(BGR2HSV) on uint8 -- color_hsv.simd.hpp RGB2HSV_b: 12-bit division tables, H in [0, 180);
  * cv2.cvtColor(HSV2BGR) on uint8 -- HSV2RGB_b: float sector arithmetic, saturate_cast<uchar>(x * 255);
  * cv2.LUT -- table look-up.
csrc/augment.hip implements the same arithmetic; tests compare the two bit for bit."""
import random

import numpy as np


def _rint(v):
    return np.rint(v).astype(np.int64)   # cvRound: nearest, ties to even


def resize_linear_u8(img, dw, dh):
    h, w = img.shape[:2]
    sx, sy = w / dw, h / dh

    def taps(n_dst, n_src, scale):
        f = ((np.arange(n_dst) + 0.5) * scale - 0.5).astype(np.float32)
        i0 = np.floor(f).astype(np.int64)
        f = f - i0.astype(np.float32)
        lo = i0 < 0
        i0[lo], f[lo] = 0, 0.0
        hi = i0 >= n_src - 1
        i0[hi], f[hi] = n_src - 1, 0.0
        i1 = np.minimum(i0 + 1, n_src - 1)
        a1 = _rint(f * np.float32(2048.0))
        a0 = _rint((np.float32(1.0) - f) * np.float32(2048.0))
        return i0, i1, a0, a1

    x0, x1, ax0, ax1 = taps(dw, w, sx)
    y0, y1, by0, by1 = taps(dh, h, sy)
    s = img.astype(np.int64)
    rows = s[:, x0] * ax0[None, :, None] + s[:, x1] * ax1[None, :, None]          # [h, dw, 3]  (x 2^11)
    q = (rows[y0] * by0[:, None, None] + rows[y1] * by1[:, None, None] + (1 << 21)) >> 22
    return np.clip(q, 0, 255).astype(np.uint8)


def preproc(img, input_size, swap=(2, 0, 1)):
    """data_augments.py:88-107."""
    padded_img = np.ones((input_size[0], input_size[1], 3), dtype=np.uint8) * 114
    r = min(input_size[0] / img.shape[0], input_size[1] / img.shape[1])
    dw, dh = int(img.shape[1] * r), int(img.shape[0] * r)
    padded_img[:dh, :dw] = resize_linear_u8(img, dw, dh)
    padded_img = padded_img.transpose(swap)
    return np.ascontiguousarray(padded_img, dtype=np.float32), r


def bgr2hsv_u8(img):
    b, g, r = [img[..., c].astype(np.int64) for c in range(3)]
    v = np.maximum(np.maximum(b, g), r)
    vmin = np.minimum(np.minimum(b, g), r)
    diff = v - vmin
    i = np.arange(256)
    with np.errstate(divide="ignore"):
        sdiv = np.where(i == 0, 0, _rint((255 << 12) / np.maximum(i, 1)))
        hdiv = np.where(i == 0, 0, _rint((180 << 12) / (6.0 * np.maximum(i, 1))))
    vr = np.where(v == r, -1, 0)
    vg = np.where(v == g, -1, 0)
    s = (diff * sdiv[v] + (1 << 11)) >> 12
    h = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))))
    h = (h * hdiv[diff] + (1 << 11)) >> 12
    h = h + np.where(h < 0, 180, 0)
    return h, s, v


def hsv2bgr_u8(h, s, v):
    hf = h.astype(np.float32) * np.float32(6.0 / 180.0)
    sf = s.astype(np.float32) * np.float32(1.0 / 255.0)
    vf = v.astype(np.float32) * np.float32(1.0 / 255.0)
    hf = np.where(hf >= 6, hf - 6, hf)            # H < 180 -> h < 6 already; kept for symmetry with the kernel
    sector = np.floor(hf).astype(np.int64)
    fr = hf - sector.astype(np.float32)
    bad = (sector < 0) | (sector >= 6)
    sector = np.where(bad, 0, sector)
    fr = np.where(bad, np.float32(0), fr)
    one = np.float32(1.0)
    tab = np.stack([vf, vf * (one - sf), vf * (one - sf * fr), vf * (one - sf * (one - fr))], -1)
    sd = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])
    idx = sd[sector]                               # [..., 3] -> b, g, r table slots
    out = np.take_along_axis(tab, idx, -1)
    out = np.where((sf == 0)[..., None], vf[..., None], out)
    return np.clip(_rint(out * np.float32(255.0)), 0, 255).astype(np.uint8)


def augment_hsv(img, hgain=0.015, sgain=0.7, vgain=0.4, gains=None):
    """data_augments.py:110-126 (in place).  `gains`: use these three factors instead of drawing them (tests)."""
    r = np.random.uniform(-1, 1, 3) * [hgain, sgain, vgain] + 1 if gains is None else np.asarray(gains, dtype=np.float64)
    hue, sat, val = bgr2hsv_u8(img)
    x = np.arange(0, 256, dtype=np.int16)
    lut_hue = ((x * r[0]) % 180).astype(np.uint8)
    lut_sat = np.clip(x * r[1], 0, 255).astype(np.uint8)
    lut_val = np.clip(x * r[2], 0, 255).astype(np.uint8)
    img[...] = hsv2bgr_u8(lut_hue[hue], lut_sat[sat], lut_val[val])
    return r


def _mirror(image, boxes):
    _, width, _ = image.shape
    image = image[:, ::-1]
    boxes[:, 0::2] = width - boxes[:, 2::-2]
    return image, boxes


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

    def __call__(self, image, targets, input_dim):
        if len(targets) == 0:
            targets = np.zeros((self.max_labels, 5), dtype=np.float32)
            image, r_o = preproc(image, input_dim)
            return image, targets
        image_process = image.copy()
        targets_process = targets.copy()
        if random.random() < self.hsv_prob:
            augment_hsv(image_process)
        if random.random() < self.flip_prob:
            image_process, targets_process[:, :4] = _mirror(image_process, targets_process[:, :4])
        image_process, r = preproc(image_process, input_dim)
        targets_process[:, :4] = xyxy2cxcywh(targets_process[:, :4])
        targets_process[:, :4] *= r
        mask_b = np.minimum(targets_process[:, 2], targets_process[:, 3]) > 1
        targets_process = targets_process[mask_b]
        if len(targets_process) == 0:
            image_process, r_o = preproc(image, input_dim)
            targets_process = targets
            targets_process[:, :4] = r_o * targets_process[:, :4]
            targets_process[:, :4] = xyxy2cxcywh(targets_process[:, :4])
        label_process = np.expand_dims(targets_process[:, 4], 1)
        targets = np.hstack((label_process, targets_process[:, :4]))
        padded_labels = np.zeros((self.max_labels, 5))
        padded_labels[range(len(targets))[: self.max_labels]] = targets[: self.max_labels]
        return image_process, np.ascontiguousarray(padded_labels, dtype=np.float32)


class ValTransform:
    """data_augments.py:51-85."""

    def __init__(self, swap=(2, 0, 1), legacy=False, max_labels=50):
        self.swap, self.legacy, self.max_labels = swap, legacy, max_labels

    def __call__(self, img, targets, input_size):
        img, _ = preproc(img, input_size, self.swap)
        if self.legacy:      # :72-76: BGR -> RGB, /255, ImageNet mean / std; float32 array, float64 constants (numpy rounds per step)
            img = img[::-1, :, :].copy()
            img /= 255.0
            img -= np.array([0.485, 0.456, 0.406]).reshape(3, 1, 1)
            img /= np.array([0.229, 0.224, 0.225]).reshape(3, 1, 1)
        boxes = xyxy2cxcywh(targets[:, :4].copy())
        labels = np.expand_dims(targets[:, 4].copy(), 1)
        targets_t = np.hstack((labels, boxes))
        padded_labels = np.zeros((self.max_labels, 5))
        padded_labels[range(len(targets_t))[:self.max_labels]] = targets_t[:self.max_labels]
        return img, padded_labels
