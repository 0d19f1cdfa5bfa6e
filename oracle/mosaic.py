"""CPU restatement of the mosaic / random-affine / mixup sample builder (TEST ORACLE -- imported by tests only).

Follows /root/reference/models/data/mosaic_detection.py:
    MosaicDetection.__getitem__ :60-167 (mosaic branch :61-146, plain branch :148-167), mixup :169-247,
    adjust_box_anns :250-253, get_mosaic_coordinate :256-274, random_perspective :277-371, box_candidates :374-387
with numpy only.  The control flow, the order of the draws from python's `random` and the label arithmetic are followed line
by line and ARE pinned: tests/golden/mosaic_samples.npz holds what the reference's own MosaicDetection returns in this
container (tools/gen_golden.py mosaic) when its `cv2` import is served by the three functions below.  The pixel arithmetic of
those OpenCV calls is RESTATED from OpenCV's published 8-bit algorithms (cv2 is not installable here): PARITY UNPINNED
against cv2 itself --
  * cv2.resize(INTER_LINEAR), uint8: oracle/augment.py resize_linear_u8;
  * cv2.getRotationMatrix2D: alpha = s cos a, beta = s sin a, [[alpha, beta, (1-alpha) cx - beta cy], [-beta, alpha, beta cx + (1-alpha) cy]];
  * cv2.warpAffine(INTER_LINEAR, BORDER_CONSTANT), uint8 -- imgproc/imgwarp.cpp: M inverted in double (invertAffineTransform);
    per column adelta/bdelta = cvRound(M[0] x 2^10), cvRound(M[3] x 2^10); per row X0 = cvRound((M[1] y + M[2]) 2^10) + 16 (same
    for Y0); X = (X0 + adelta[x]) >> 5: integer part X >> 5 (saturated to int16), 5 fraction bits; bilinear weights from
    the 32 x 32 table, short(w * 32768) (exact: the products are multiples of 2^-10); pixel = (sum of 4 taps * weights + 2^14) >> 15;
    a tap outside the source reads the border value.
Rounding cut-out (models/data/augmentation/cutout_round.py:6-55, models/utils/bbox.py:76-94; pure numpy, so PINNED: the
reference function itself runs in this container -- tests/golden/cutout_round.npz, tools/gen_golden.py cutout) is restated below;
copy-paste / cut-paste read `dataset.object_cls` / `.back_cls`, which no dataset class of the reference defines (they cannot run
there either): a non-zero probability raises.
csrc/augment.hip implements the same arithmetic; tests compare the two bit for bit."""
import math
import random

import numpy as np

from .augment import resize_linear_u8


# ---- the three OpenCV calls ------------------------------------------------------------------------------------------------
def get_rotation_matrix_2d(center, angle, scale):
    a = angle * math.pi / 180.0
    alpha, beta = math.cos(a) * scale, math.sin(a) * scale
    cx, cy = center
    return np.array([[alpha, beta, (1 - alpha) * cx - beta * cy], [-beta, alpha, beta * cx + (1 - alpha) * cy]], dtype=np.float64)


def invert_affine(M):
    m = np.asarray(M, dtype=np.float64).reshape(2, 3).copy().ravel()
    D = m[0] * m[4] - m[1] * m[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = m[4] * D, m[0] * D
    m[0] = A11
    m[1] *= -D
    m[3] *= -D
    m[4] = A22
    b1 = -m[0] * m[2] - m[1] * m[5]
    b2 = -m[3] * m[2] - m[4] * m[5]
    m[2], m[5] = b1, b2
    return m


def _rint(v):
    return np.rint(v).astype(np.int64)


def warp_affine_u8(img, M, dsize, border_value=(114, 114, 114)):
    """cv2.warpAffine(img, M, dsize=(width, height), borderValue=...) for uint8 HWC, INTER_LINEAR, BORDER_CONSTANT."""
    width, height = int(dsize[0]), int(dsize[1])
    sh, sw = img.shape[:2]
    m = invert_affine(M)
    xs = np.arange(width, dtype=np.float64)
    ys = np.arange(height, dtype=np.float64)
    adelta = _rint(m[0] * xs * 1024.0)
    bdelta = _rint(m[3] * xs * 1024.0)
    X0 = _rint((m[1] * ys + m[2]) * 1024.0) + 16
    Y0 = _rint((m[4] * ys + m[5]) * 1024.0) + 16
    X = (X0[:, None] + adelta[None, :]) >> 5
    Y = (Y0[:, None] + bdelta[None, :]) >> 5
    sx = np.clip(X >> 5, -32768, 32767)
    sy = np.clip(Y >> 5, -32768, 32767)
    fx, fy = X & 31, Y & 31
    w00 = (32 - fy) * (32 - fx) * 32          # (1 - fy/32)(1 - fx/32) * 32768, exact
    w01 = (32 - fy) * fx * 32
    w10 = fy * (32 - fx) * 32
    w11 = fy * fx * 32
    src = img.astype(np.int64)
    cval = np.asarray(border_value, dtype=np.int64)[: img.shape[2]]

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < sh) & (xx >= 0) & (xx < sw)
        v = src[np.clip(yy, 0, sh - 1), np.clip(xx, 0, sw - 1)]
        return np.where(ok[..., None], v, cval[None, None, :])

    acc = (tap(sy, sx) * w00[..., None] + tap(sy, sx + 1) * w01[..., None]
           + tap(sy + 1, sx) * w10[..., None] + tap(sy + 1, sx + 1) * w11[..., None])
    out = (acc + (1 << 14)) >> 15
    return np.clip(out, 0, 255).astype(np.uint8)


def invert_3x3(M):
    """cv::invert of a 3x3 double matrix (the closed form OpenCV uses up to 3x3: cofactors times 1 / det3)."""
    S = np.asarray(M, dtype=np.float64).reshape(3, 3)
    d = (S[0, 0] * (S[1, 1] * S[2, 2] - S[1, 2] * S[2, 1]) - S[0, 1] * (S[1, 0] * S[2, 2] - S[1, 2] * S[2, 0])
         + S[0, 2] * (S[1, 0] * S[2, 1] - S[1, 1] * S[2, 0]))
    if d == 0.0:
        return np.zeros(9)
    d = 1.0 / d
    return np.array([
        (S[1, 1] * S[2, 2] - S[1, 2] * S[2, 1]) * d, (S[0, 2] * S[2, 1] - S[0, 1] * S[2, 2]) * d, (S[0, 1] * S[1, 2] - S[0, 2] * S[1, 1]) * d,
        (S[1, 2] * S[2, 0] - S[1, 0] * S[2, 2]) * d, (S[0, 0] * S[2, 2] - S[0, 2] * S[2, 0]) * d, (S[0, 2] * S[1, 0] - S[0, 0] * S[1, 2]) * d,
        (S[1, 0] * S[2, 1] - S[1, 1] * S[2, 0]) * d, (S[0, 1] * S[2, 0] - S[0, 0] * S[2, 1]) * d, (S[0, 0] * S[1, 1] - S[0, 1] * S[1, 0]) * d])


def warp_perspective_u8(img, M, dsize, border_value=(114, 114, 114)):
    """cv2.warpPerspective(img, M, dsize=(width, height), borderValue=...) for uint8 HWC, INTER_LINEAR, BORDER_CONSTANT
    (mosaic_detection.py:320-323 calls it with the AFFINE 3x3 matrix whenever `perspective` is non-zero).  OpenCV's
    WarpPerspectiveInvoker, restated: M is inverted, destination blocks are 64 columns wide, inside a block
    X0 = M0*x_block + M1*y + M2 (doubles, left to right), per pixel W = W0 + M6*x1, W = W ? 32/W : 0,
    X = cvRound(clamp((X0 + M0*x1) * W)); integer part X >> 5, 5-bit bilinear fractions, the same 15-bit weights and
    rounding as warpAffine's remap.  Parity unpinned (cv2 is absent), like warp_affine_u8."""
    width, height = int(dsize[0]), int(dsize[1])
    sh, sw = img.shape[:2]
    m = invert_3x3(M)
    bw = min(64, width)
    xs = np.arange(width, dtype=np.int64)
    xb = ((xs // bw) * bw).astype(np.float64)[None, :]
    x1 = (xs % bw).astype(np.float64)[None, :]
    ys = np.arange(height, dtype=np.float64)[:, None]
    X0 = m[0] * xb + m[1] * ys + m[2]
    Y0 = m[3] * xb + m[4] * ys + m[5]
    W0 = m[6] * xb + m[7] * ys + m[8]
    W = W0 + m[6] * x1
    with np.errstate(divide="ignore"):
        W = np.where(W != 0.0, 32.0 / np.where(W != 0.0, W, 1.0), 0.0)
    lim_lo, lim_hi = float(-2 ** 31), float(2 ** 31 - 1)
    fX = np.maximum(lim_lo, np.minimum(lim_hi, (X0 + m[0] * x1) * W))
    fY = np.maximum(lim_lo, np.minimum(lim_hi, (Y0 + m[3] * x1) * W))
    X = np.clip(_rint(fX), -2 ** 31, 2 ** 31 - 1)
    Y = np.clip(_rint(fY), -2 ** 31, 2 ** 31 - 1)
    sx = np.clip(X >> 5, -32768, 32767)
    sy = np.clip(Y >> 5, -32768, 32767)
    fx, fy = X & 31, Y & 31
    w00 = (32 - fy) * (32 - fx) * 32
    w01 = (32 - fy) * fx * 32
    w10 = fy * (32 - fx) * 32
    w11 = fy * fx * 32
    src = img.astype(np.int64)
    cval = np.asarray(border_value, dtype=np.int64)[: img.shape[2]]

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < sh) & (xx >= 0) & (xx < sw)
        v = src[np.clip(yy, 0, sh - 1), np.clip(xx, 0, sw - 1)]
        return np.where(ok[..., None], v, cval[None, None, :])

    acc = (tap(sy, sx) * w00[..., None] + tap(sy, sx + 1) * w01[..., None]
           + tap(sy + 1, sx) * w10[..., None] + tap(sy + 1, sx + 1) * w11[..., None])
    out = (acc + (1 << 14)) >> 15
    return np.clip(out, 0, 255).astype(np.uint8)


def resize(img, dsize):
    """cv2.resize(img, (w, h), interpolation=INTER_LINEAR) for uint8 HWC."""
    return resize_linear_u8(img, int(dsize[0]), int(dsize[1]))


# ---- mosaic_detection.py, line by line -------------------------------------------------------------------------------------
def adjust_box_anns(bbox, scale_ratio, padw, padh, w_max, h_max):
    bbox[:, 0::2] = np.clip(bbox[:, 0::2] * scale_ratio + padw, 0, w_max)
    bbox[:, 1::2] = np.clip(bbox[:, 1::2] * scale_ratio + padh, 0, h_max)
    return bbox


def get_mosaic_coordinate(k, xc, yc, w, h, input_h, input_w):
    """:256-274.  Quadrant k (0 TL, 1 TR, 2 BL, 3 BR) of the 2H x 2W canvas around the centre (xc, yc): the canvas rectangle the
    w x h image lands in (clipped to the canvas) and the part of the image that is visible there."""
    left, top = k in (0, 2), k in (0, 1)
    x1, x2 = (max(xc - w, 0), xc) if left else (xc, min(xc + w, input_w * 2))
    y1, y2 = (max(yc - h, 0), yc) if top else (yc, min(input_h * 2, yc + h))
    sx = (w - (x2 - x1), w) if left else (0, min(w, x2 - x1))      # left quadrants keep the image's right edge at the centre
    sy = (h - (y2 - y1), h) if top else (0, min(y2 - y1, h))
    return (x1, y1, x2, y2), (sx[0], sy[0], sx[1], sy[1])


def box_candidates(box1, box2, wh_thr=2, ar_thr=20, area_thr=0.2):
    w1, h1 = box1[2] - box1[0], box1[3] - box1[1]
    w2, h2 = box2[2] - box2[0], box2[3] - box2[1]
    ar = np.maximum(w2 / (h2 + 1e-16), h2 / (w2 + 1e-16))
    return (w2 > wh_thr) & (h2 > wh_thr) & (w2 * h2 / (w1 * h1 + 1e-16) > area_thr) & (ar < ar_thr)


def affine_decision(shape_hw, degrees, translate, scale, shear, border):
    """The draws and the matrix of random_perspective :288-326 (perspective = 0).  Returns (M 3x3, s, width, height)."""
    height = shape_hw[0] + border[0] * 2
    width = shape_hw[1] + border[1] * 2
    C = np.eye(3)
    C[0, 2] = -shape_hw[1] / 2
    C[1, 2] = -shape_hw[0] / 2
    R = np.eye(3)
    a = random.uniform(-degrees, degrees)
    s = random.uniform(scale[0], scale[1])
    R[:2] = get_rotation_matrix_2d((0, 0), a, s)
    S = np.eye(3)
    S[0, 1] = math.tan(random.uniform(-shear, shear) * math.pi / 180)
    S[1, 0] = math.tan(random.uniform(-shear, shear) * math.pi / 180)
    T = np.eye(3)
    T[0, 2] = random.uniform(0.5 - translate, 0.5 + translate) * width
    T[1, 2] = random.uniform(0.5 - translate, 0.5 + translate) * height
    M = T @ S @ R @ C
    return M, s, width, height


def affine_labels(targets, M, s, width, height):
    """random_perspective :346-369."""
    n = len(targets)
    if n:
        xy = np.ones((n * 4, 3))
        xy[:, :2] = targets[:, [0, 1, 2, 3, 0, 3, 2, 1]].reshape(n * 4, 2)
        xy = xy @ M.T
        xy = xy[:, :2].reshape(n, 8)
        x = xy[:, [0, 2, 4, 6]]
        y = xy[:, [1, 3, 5, 7]]
        xy = np.concatenate((x.min(1), y.min(1), x.max(1), y.max(1))).reshape(4, n).T
        xy[:, [0, 2]] = xy[:, [0, 2]].clip(0, width)
        xy[:, [1, 3]] = xy[:, [1, 3]].clip(0, height)
        i = box_candidates(box1=targets[:, :4].T * s, box2=xy.T)
        targets = targets[i]
        targets[:, :4] = xy[i]
    return targets


def random_perspective(img, targets=(), degrees=10, translate=0.1, scale=(0.5, 1.5), shear=10, perspective=0.0, border=(0, 0)):
    """mosaic_detection.py:269-371.  `perspective` only selects the OpenCV entry point (:319-327): the matrix stays the affine
    T S R C, so the box arithmetic (division by a third coordinate that is exactly 1, :341-342) is unchanged."""
    M, s, width, height = affine_decision(img.shape[:2], degrees, translate, scale, shear, border)
    if (border[0] != 0) or (border[1] != 0) or (M != np.eye(3)).any():
        if perspective:
            img = warp_perspective_u8(img, M, (width, height), (114, 114, 114))
        else:
            img = warp_affine_u8(img, M[:2], (width, height), (114, 114, 114))
    return img, affine_labels(targets, M, s, width, height)


# ---- rounding cut-out ------------------------------------------------------------------------------------------------------
def bbox_ioa(box1, box2):
    """models/utils/bbox.py:76-94: intersection of box1 [4] with every row of box2 [n,4], over the area of the box2 row."""
    box2 = box2.transpose()
    b1_x1, b1_y1, b1_x2, b1_y2 = box1[0], box1[1], box1[2], box1[3]
    b2_x1, b2_y1, b2_x2, b2_y2 = box2[0], box2[1], box2[2], box2[3]
    inter = (np.minimum(b1_x2, b2_x2) - np.maximum(b1_x1, b2_x1)).clip(0) * (np.minimum(b1_y2, b2_y2) - np.maximum(b1_y1, b2_y1)).clip(0)
    return inter / ((b2_x2 - b2_x1) * (b2_y2 - b2_y1) + 1e-16)


def cutout_strips(labels, h, w):
    """The one-pixel strips around every label box whose mean colours make up the fill colour (cutout_round.py:14-31), as
    python slices (rows, cols) in the reference's order: beside the left edge, beside the right edge, above, below -- each only
    where the box keeps more than a pixel from that image border.  (The reference names them left / top / right / bottom.)"""
    out = []
    for i in range(len(labels)):
        x0, y0, x1, y1 = (int(labels[i, k]) for k in range(4))
        if labels[i, 0] > 1:
            out.append((slice(y0, y1), slice(x0 - 1, x0), 0))
        if labels[i, 2] < w - 1:
            out.append((slice(y0, y1), slice(x1, x1 + 1), 0))
        if labels[i, 1] > 1:
            out.append((slice(y0 - 1, y0), slice(x0, x1), 1))
        if labels[i, 3] < h - 1:
            out.append((slice(y1, y1 + 1), slice(x0, x1), 1))
    return out


def cutout_rounding(img, labels, n_hole, cutout_ratio, mixup, ioa_thre):
    """cutout_round.py:6-55.  Fill colour = mean over the strips of their mean colours (114 without strips); 1..3 rectangles at
    numpy-random places, each blended `mixup` : 1 - mixup into the image (float64, truncated into the uint8 image, IN ORDER: a later
    hole blends over an earlier one) unless it covers `ioa_thre` or more of some label box.  Draws: randint for the count, then
    three randints per hole.  Works on `img` in place (the caller copied it, mosaic_detection.py:80) and returns it."""
    h, w = img.shape[:2]
    if len(labels) == 0:
        return img.astype(np.uint8)
    fills = [img[rs, cs].mean(ax) for rs, cs, ax in cutout_strips(labels, h, w)]
    fill_in = np.array(fills).mean(0).reshape(3) if len(fills) != 0 else np.array([114, 114, 114])
    n = np.random.randint(n_hole[0], n_hole[1] + 1)
    for _ in range(n):
        x1 = np.random.randint(0, w)
        y1 = np.random.randint(0, h)
        index = np.random.randint(0, len(cutout_ratio))
        x2 = int(np.clip(x1 + cutout_ratio[index][0] * w, x1, w))
        y2 = int(np.clip(y1 + cutout_ratio[index][1] * h, y1, h))
        if bbox_ioa([x1, y1, x2, y2], labels[:, :4]).max() < ioa_thre:
            cut = np.ones(img[y1:y2, x1:x2, :].shape) * fill_in
            img[y1:y2, x1:x2, :] = mixup * cut + (1 - mixup) * img[y1:y2, x1:x2, :]
    return img.astype(np.uint8)


CR_NHOLE, CR_RATIO, CR_MIXUP, CR_IOA = (1, 3), [[0.1, 0.1], [0.3, 0.1], [0.1, 0.3], [0.2, 0.2], [0.3, 0.3]], 0.7, 0.2   # mosaic_detection.py:52-55


class MosaicDetection:
    """mosaic_detection.py:12-247 over a dataset object with `.annotations[i] = (labels [n,5] xyxy+cls, img_hw, resized_info,
    name)`, `.imgs` (list of uint8 HWC arrays, or None) / `.load_resized_img(i)`, `.img_size`."""

    def __init__(self, dataset, img_size, preprocess=None, mosaic_prob=1.0, mosaic_scale=(0.5, 1.5), degrees=10, translate=0.1,
                 shear=2.0, perspective=0.0, mixup_prob=1.0, mixup_scale=(0.5, 1.5), copypaste_prob=0.0,
                 copypaste_scale=(0.5, 1.5), cutpaste_prob=0.0, cutoutR_prob=0.0):
        if copypaste_prob or cutpaste_prob:
            raise NotImplementedError("copy-paste / cut-paste are not restated (they cannot run in the reference either)")
        self._dataset, self.img_size, self.preprocess = dataset, img_size, preprocess
        self.mosaic_prob, self.scale = mosaic_prob, mosaic_scale
        self.degrees, self.translate, self.shear, self.perspective = degrees, translate, shear, perspective
        self.mixup_prob, self.mixup_scale = mixup_prob, mixup_scale
        self.copypaste_scale = copypaste_scale
        self.off_probs = (copypaste_prob, cutpaste_prob, cutoutR_prob)

    def __len__(self):
        return len(self._dataset)

    def _img(self, index):
        ds = self._dataset
        return ds.imgs[index] if ds.imgs is not None else ds.load_resized_img(index)

    def _per_image(self, img, labels, mosaic):
        """The three per-image augmentations (:86-91, :156-161): copy-paste and cut-paste (probability 0) only consume their
        draws -- in the mosaic branch the first one is skipped for an image without labels (short-circuit `and`) -- the rounding
        cut-out runs when its draw says so."""
        for k, prob in enumerate(self.off_probs):
            if k == 0 and mosaic and len(labels) == 0:
                continue
            if random.random() < prob:
                if k < 2:
                    raise NotImplementedError
                img = cutout_rounding(img, labels, CR_NHOLE, CR_RATIO, CR_MIXUP, CR_IOA)
        return img

    def __getitem__(self, idx):
        ds = self._dataset
        if not random.random() < self.mosaic_prob:                                    # plain branch :148-167
            res, img_hw, _, img_name = ds.annotations[idx]
            ds.img_size = self.img_size
            img = self._per_image(self._img(idx).copy(), res, mosaic=False)
            target = res
            if self.preprocess is not None:
                img, target = self.preprocess(img, res, self.img_size)
            return img, target, img_hw, np.array([idx]), img_name
        H, W = ds.img_size[0], ds.img_size[1]                                         # mosaic branch :61-146
        yc = int(random.uniform(0.5 * H, 1.5 * H))
        xc = int(random.uniform(0.5 * W, 1.5 * W))
        members = [idx] + [random.randint(0, len(ds) - 1) for _ in range(3)]
        canvas = np.full((2 * H, 2 * W, 3), 114, dtype=np.uint8)
        parts = []
        for k, index in enumerate(members):
            boxes, _, _, img_name = ds.annotations[index]
            img = self._per_image(self._img(index).copy(), boxes, mosaic=True)
            h0, w0 = img.shape[:2]
            scale = min(1. * H / h0, 1. * W / w0)
            img = resize(img, (int(w0 * scale), int(h0 * scale)))
            h, w = img.shape[:2]
            (lx1, ly1, lx2, ly2), (sx1, sy1, sx2, sy2) = get_mosaic_coordinate(k, xc, yc, w, h, H, W)
            canvas[ly1:ly2, lx1:lx2] = img[sy1:sy2, sx1:sx2]
            moved = boxes.copy()
            if boxes.size > 0:
                moved[:, 0] = scale * boxes[:, 0] + (lx1 - sx1)
                moved[:, 1] = scale * boxes[:, 1] + (ly1 - sy1)
                moved[:, 2] = scale * boxes[:, 2] + (lx1 - sx1)
                moved[:, 3] = scale * boxes[:, 3] + (ly1 - sy1)
            parts.append(moved)
        labels = np.concatenate(parts, 0)
        for col, hi in ((0, 2 * W), (1, 2 * H), (2, 2 * W), (3, 2 * H)):
            np.clip(labels[:, col], 0, hi, out=labels[:, col])
        canvas, labels = random_perspective(canvas, labels, degrees=self.degrees, translate=self.translate, scale=self.scale,
                                            shear=self.shear, perspective=self.perspective, border=[-H // 2, -W // 2])
        if not len(labels) == 0 and random.random() < self.mixup_prob:
            canvas, labels = self.mixup(canvas, labels, self.img_size)
        mix_img, padded_labels = self.preprocess(canvas, labels, self.img_size)
        return mix_img, padded_labels, (mix_img.shape[1], mix_img.shape[2]), np.array([idx]), img_name

    def mixup(self, origin_img, origin_labels, input_dim):
        """:169-247."""
        jit = random.uniform(*self.copypaste_scale)             # sic: the reference jitters by copypaste_scale (:170)
        flip = random.uniform(0, 1) > 0.5
        other = []
        while len(other) == 0:                                  # an image WITH labels
            k = random.randint(0, len(self) - 1)
            other = self._dataset.annotations[k][0]
        img = self._img(k)
        # letterbox into input_dim (pad 114), then rescale the WHOLE padded image by the jitter factor
        r = min(input_dim[0] / img.shape[0], input_dim[1] / img.shape[1])
        boxed = np.ones((input_dim[0], input_dim[1], 3), dtype=np.uint8) * 114
        boxed[:int(img.shape[0] * r), :int(img.shape[1] * r)] = resize(img, (int(img.shape[1] * r), int(img.shape[0] * r)))
        boxed = resize(boxed, (int(boxed.shape[1] * jit), int(boxed.shape[0] * jit)))
        r *= jit
        if flip:
            boxed = boxed[:, ::-1, :]
        bh, bw = boxed.shape[:2]
        th, tw = origin_img.shape[:2]
        pad = np.zeros((max(bh, th), max(bw, tw), 3), dtype=np.uint8)       # zeros, not 114 (:206-209)
        pad[:bh, :bw] = boxed
        x_off = y_off = 0
        if pad.shape[0] > th:
            y_off = random.randint(0, pad.shape[0] - th - 1)
        if pad.shape[1] > tw:
            x_off = random.randint(0, pad.shape[1] - tw - 1)
        crop = pad[y_off:y_off + th, x_off:x_off + tw]
        b = adjust_box_anns(other[:, :4].copy(), r, 0, 0, bw, bh)
        if flip:
            b[:, 0::2] = bw - b[:, 0::2][:, ::-1]
        b[:, 0::2] = np.clip(b[:, 0::2] - x_off, 0, tw)
        b[:, 1::2] = np.clip(b[:, 1::2] - y_off, 0, th)
        origin_labels = np.vstack((origin_labels, np.hstack((b, other[:, 4:5].copy()))))
        blend = 0.5 * origin_img.astype(np.float32) + 0.5 * crop.astype(np.float32)
        return blend.astype(np.uint8), origin_labels
