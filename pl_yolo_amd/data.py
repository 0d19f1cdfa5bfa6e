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
`MosaicDetection` (models/data/mosaic_detection.py:12-247) builds mosaic / random-affine / mixup samples the same way: four
launches for the pixels (paste four letterboxed images on the 2H x 2W canvas, warpAffine, letterbox + jitter-rescale the
mixup partner, blend), the decisions and the box arithmetic on the host."""
import ctypes as C
import math
import random

import numpy as np
import torch

from ._lib import MAX_HOLES, AugImage, MosaicTile, PlyoloError, Rect, call


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
        if self.legacy:
            # data_augments.py:72-76: BGR -> RGB, /255 in fp32, then ImageNet mean / std as float64 constants on the fp32 array
            # (numpy computes those two steps in double and rounds each back to fp32)
            # (the /255 through fp64 as well: torch turns a division by a scalar into a multiplication by its reciprocal on the
            # device; pixel values are integers, for which rounding the fp64 quotient equals the fp32 division numpy performs)
            out = (out.flip(1).double() / 255.0).float()
            mean = torch.tensor([0.485, 0.456, 0.406], dtype=torch.float64, device=out.device).view(1, 3, 1, 1)
            std = torch.tensor([0.229, 0.224, 0.225], dtype=torch.float64, device=out.device).view(1, 3, 1, 1)
            out = (out.double() - mean).float()
            out = (out.double() / std).float()
        return out, np.stack([self._labels(t) for t in targets_list], 0)

    def __call__(self, img, targets, input_size):
        out, labels = self.batch([img], [targets], input_size)
        return out[0], labels[0]


# ---------------------------------------------------------------------------------------------------------------------------
# mosaic / random affine / mixup
def _stream():
    return torch.cuda.current_stream().cuda_stream


def mosaic_coordinate(k, xc, yc, w, h, input_h, input_w):
    """get_mosaic_coordinate, mosaic_detection.py:256-274: quadrant k (0 top-left, 1 top-right, 2 bottom-left, 3 bottom-right)
    around the centre (xc, yc) -> (canvas rectangle, visible rectangle of the w x h image)."""
    if k in (0, 2):
        x1, x2 = max(xc - w, 0), xc
        sx1, sx2 = w - (x2 - x1), w
    else:
        x1, x2 = xc, min(xc + w, input_w * 2)
        sx1, sx2 = 0, min(w, x2 - x1)
    if k in (0, 1):
        y1, y2 = max(yc - h, 0), yc
        sy1, sy2 = h - (y2 - y1), h
    else:
        y1, y2 = yc, min(input_h * 2, yc + h)
        sy1, sy2 = 0, min(y2 - y1, h)
    return (x1, y1, x2, y2), (sx1, sy1, sx2, sy2)


def invert_affine(M):
    """cv::invertAffineTransform on a 2x3 matrix, in float64 (what cv2.warpAffine does to M before sampling)."""
    a, b, c, d, e, f = [float(v) for v in np.asarray(M, dtype=np.float64).reshape(6)]
    det = a * e - b * d
    det = 1.0 / det if det != 0 else 0.0
    ia, ie = e * det, a * det
    ib, id_ = b * -det, d * -det
    return [ia, ib, -ia * c - ib * f, id_, ie, -id_ * c - ie * f]


def warp_affine(img, M, dsize, border_value=114):
    """cv2.warpAffine(img, M, dsize=(width, height), borderValue=(v, v, v)) on a uint8 HWC device tensor."""
    img = _check_image(img)
    width, height = int(dsize[0]), int(dsize[1])
    out = torch.empty(height, width, 3, dtype=torch.uint8, device=img.device)
    inv = (C.c_double * 6)(*invert_affine(M))
    call("plyolo_warp_affine_u8", img.data_ptr(), int(img.shape[0]), int(img.shape[1]), inv, out.data_ptr(), height, width, int(border_value), _stream())
    return out


def invert_3x3(M):
    """cv::invert of a 3x3 double matrix (OpenCV's closed form up to 3x3: cofactors times 1 / det), what cv2.warpPerspective
    does to M before sampling."""
    S = [[float(v) for v in row] for row in np.asarray(M, dtype=np.float64).reshape(3, 3)]
    d = (S[0][0] * (S[1][1] * S[2][2] - S[1][2] * S[2][1]) - S[0][1] * (S[1][0] * S[2][2] - S[1][2] * S[2][0])
         + S[0][2] * (S[1][0] * S[2][1] - S[1][1] * S[2][0]))
    if d == 0.0:
        return [0.0] * 9
    d = 1.0 / d
    return [(S[1][1] * S[2][2] - S[1][2] * S[2][1]) * d, (S[0][2] * S[2][1] - S[0][1] * S[2][2]) * d, (S[0][1] * S[1][2] - S[0][2] * S[1][1]) * d,
            (S[1][2] * S[2][0] - S[1][0] * S[2][2]) * d, (S[0][0] * S[2][2] - S[0][2] * S[2][0]) * d, (S[0][2] * S[1][0] - S[0][0] * S[1][2]) * d,
            (S[1][0] * S[2][1] - S[1][1] * S[2][0]) * d, (S[0][1] * S[2][0] - S[0][0] * S[2][1]) * d, (S[0][0] * S[1][1] - S[0][1] * S[1][0]) * d]


def warp_perspective(img, M, dsize, border_value=114):
    """cv2.warpPerspective(img, M, dsize=(width, height), borderValue=(v, v, v)) on a uint8 HWC device tensor, M 3x3."""
    img = _check_image(img)
    width, height = int(dsize[0]), int(dsize[1])
    out = torch.empty(height, width, 3, dtype=torch.uint8, device=img.device)
    inv = (C.c_double * 9)(*invert_3x3(M))
    call("plyolo_warp_perspective_u8", img.data_ptr(), int(img.shape[0]), int(img.shape[1]), inv, out.data_ptr(), height, width, int(border_value), _stream())
    return out


def resize_pad(img, dsize, out_hw=None, pad=114):
    """cv2.resize(img, (w, h)) into the top-left corner of a [out_h, out_w, 3] image filled with `pad` (default: no padding)."""
    img = _check_image(img)
    dw, dh = int(dsize[0]), int(dsize[1])
    oh, ow = (dh, dw) if out_hw is None else (int(out_hw[0]), int(out_hw[1]))
    out = torch.empty(oh, ow, 3, dtype=torch.uint8, device=img.device)
    call("plyolo_resize_pad_u8", img.data_ptr(), int(img.shape[0]), int(img.shape[1]), dh, dw, out.data_ptr(), oh, ow, int(pad), _stream())
    return out


def _box_candidates(box1, box2, wh_thr=2, ar_thr=20, area_thr=0.2):   # mosaic_detection.py:374-387
    w1, h1 = box1[2] - box1[0], box1[3] - box1[1]
    w2, h2 = box2[2] - box2[0], box2[3] - box2[1]
    ar = np.maximum(w2 / (h2 + 1e-16), h2 / (w2 + 1e-16))
    return (w2 > wh_thr) & (h2 > wh_thr) & (w2 * h2 / (w1 * h1 + 1e-16) > area_thr) & (ar < ar_thr)


def random_perspective(img, targets=(), degrees=10, translate=0.1, scale=(0.5, 1.5), shear=10, perspective=0.0, border=(0, 0)):
    """mosaic_detection.py:269-371: five draws, M = T S R C, warpAffine -- or, when `perspective` is non-zero, warpPerspective with
    the same (affine) matrix, :319-327 -- with border 114, boxes through M (the perspective division :341-342 divides by exactly 1),
    clipped, filtered by box_candidates."""
    sh, sw = int(img.shape[0]), int(img.shape[1])
    height, width = sh + border[0] * 2, sw + border[1] * 2
    Cm = np.eye(3)
    Cm[0, 2], Cm[1, 2] = -sw / 2, -sh / 2
    a = random.uniform(-degrees, degrees)
    s = random.uniform(scale[0], scale[1])
    ang = a * math.pi / 180.0                                  # cv2.getRotationMatrix2D(angle=a, center=(0, 0), scale=s)
    alpha, beta = math.cos(ang) * s, math.sin(ang) * s
    R = np.array([[alpha, beta, 0.0], [-beta, alpha, 0.0], [0.0, 0.0, 1.0]])
    S = np.eye(3)
    S[0, 1] = math.tan(random.uniform(-shear, shear) * math.pi / 180)
    S[1, 0] = math.tan(random.uniform(-shear, shear) * math.pi / 180)
    T = np.eye(3)
    T[0, 2] = random.uniform(0.5 - translate, 0.5 + translate) * width
    T[1, 2] = random.uniform(0.5 - translate, 0.5 + translate) * height
    M = T @ S @ R @ Cm
    if (border[0] != 0) or (border[1] != 0) or (M != np.eye(3)).any():
        img = warp_perspective(img, M, (width, height), 114) if perspective else warp_affine(img, M[:2], (width, height), 114)
    n = len(targets)
    if n:
        xy = np.ones((n * 4, 3))
        xy[:, :2] = targets[:, [0, 1, 2, 3, 0, 3, 2, 1]].reshape(n * 4, 2)     # the four corners of every box
        xy = (xy @ M.T)[:, :2].reshape(n, 8)
        x, y = xy[:, [0, 2, 4, 6]], xy[:, [1, 3, 5, 7]]
        xy = np.concatenate((x.min(1), y.min(1), x.max(1), y.max(1))).reshape(4, n).T
        xy[:, [0, 2]] = xy[:, [0, 2]].clip(0, width)
        xy[:, [1, 3]] = xy[:, [1, 3]].clip(0, height)
        keep = _box_candidates(box1=targets[:, :4].T * s, box2=xy.T)
        targets = targets[keep]
        targets[:, :4] = xy[keep]
    return img, targets


CR_NHOLE, CR_RATIO, CR_MIXUP, CR_IOA = (1, 3), [[0.1, 0.1], [0.3, 0.1], [0.1, 0.3], [0.2, 0.2], [0.3, 0.3]], 0.7, 0.2   # mosaic_detection.py:52-55


def _bbox_ioa(box1, box2):
    """models/utils/bbox.py:76-94: intersection of box1 with every row of box2 over the area of that row."""
    box2 = box2.transpose()
    inter = (np.minimum(box1[2], box2[2]) - np.maximum(box1[0], box2[0])).clip(0) * (np.minimum(box1[3], box2[3]) - np.maximum(box1[1], box2[1])).clip(0)
    return inter / ((box2[2] - box2[0]) * (box2[3] - box2[1]) + 1e-16)


def cutout_rounding(img, labels, n_hole=CR_NHOLE, cutout_ratio=CR_RATIO, mixup=CR_MIXUP, ioa_thre=CR_IOA):
    """models/data/augmentation/cutout_round.py:6-55 on a uint8 HWC device image, IN PLACE (the caller owns a copy, as the reference's
    callers do, mosaic_detection.py:80).  The fill colour is the mean over the one-pixel strips around the label boxes of their mean
    colours: the strip sums come from the device (plyolo_rect_sums_u8, exact integers), the float64 means from numpy like the
    reference's; the holes are drawn from numpy.random in the reference's order and tested against the boxes here, the accepted ones
    are blended into the image by ONE launch in their order (plyolo_cutout_holes_u8)."""
    _check_image(img)
    h, w = int(img.shape[0]), int(img.shape[1])
    if len(labels) == 0:
        return img
    strips = []      # (rows, cols, number of pixels the reference's mean divides by) in its order: :20-31
    for i in range(len(labels)):
        x0, y0, x1, y1 = (int(labels[i, k]) for k in range(4))
        cand = []
        if labels[i, 0] > 1:
            cand.append((slice(y0, y1), slice(x0 - 1, x0)))
        if labels[i, 2] < w - 1:
            cand.append((slice(y0, y1), slice(x1, x1 + 1)))
        if labels[i, 1] > 1:
            cand.append((slice(y0 - 1, y0), slice(x0, x1)))
        if labels[i, 3] < h - 1:
            cand.append((slice(y1, y1 + 1), slice(x0, x1)))
        for rs, cs in cand:
            ra, rb, _ = rs.indices(h)
            ca, cb, _ = cs.indices(w)
            strips.append((ra, max(ra, rb), ca, max(ca, cb)))
    if strips:
        rects = torch.tensor(strips, dtype=torch.int32, device=img.device)
        sums = torch.zeros(len(strips), 3, dtype=torch.int64, device=img.device)
        call("plyolo_rect_sums_u8", img.data_ptr(), h, w, rects.data_ptr(), len(strips), sums.data_ptr(), _stream())
        sums = sums.cpu().numpy().astype(np.float64)       # the one synchronisation of this augmentation
        with np.errstate(invalid="ignore", divide="ignore"):
            fills = [(sums[k] / float((r[1] - r[0]) * (r[3] - r[2]))).reshape(1, 3) for k, r in enumerate(strips)]   # an empty strip: nan, as numpy's mean
        fill_in = np.array(fills).mean(0).reshape(3)
    else:
        fill_in = np.array([114, 114, 114])
    holes = []
    for _ in range(np.random.randint(n_hole[0], n_hole[1] + 1)):
        x1 = np.random.randint(0, w)
        y1 = np.random.randint(0, h)
        index = np.random.randint(0, len(cutout_ratio))
        x2 = int(np.clip(x1 + cutout_ratio[index][0] * w, x1, w))
        y2 = int(np.clip(y1 + cutout_ratio[index][1] * h, y1, h))
        if _bbox_ioa([x1, y1, x2, y2], labels[:, :4]).max() < ioa_thre:
            holes.append((x1, y1, x2, y2))
    fill = (C.c_double * 3)(*[float(v) for v in fill_in])
    for k in range(0, len(holes), MAX_HOLES):
        part = holes[k:k + MAX_HOLES]
        arr = (Rect * len(part))(*[Rect(*hh) for hh in part])
        call("plyolo_cutout_holes_u8", img.data_ptr(), h, w, arr, len(part), fill, float(mixup), _stream())
    return img


class MosaicDetection:
    """mosaic_detection.py:12-247 on the device.  `dataset` has the reference's shape -- `.annotations[i] = (labels [n,5]
    xyxy + class, img_hw, resized_info, name)`, `.img_size`, `.imgs` (list, or None) / `.load_resized_img(i)` -- with uint8
    HWC BGR DEVICE tensors as images; `preprocess` is a pl_yolo_amd.data.TrainTransform.  `md[idx]` returns the reference's
    tuple (image [3,H,W] fp32 device tensor, padded labels, (H, W), array([idx]), name) and draws from `random` /
    `numpy.random` in the reference's order; `md.batch(indices)` builds the samples in that same order and runs the final
    transform of the whole batch as one launch."""

    def __init__(self, dataset, img_size, preprocess=None, mosaic_prob=1.0, mosaic_scale=(0.5, 1.5), degrees=10, translate=0.1,
                 shear=2.0, perspective=0.0, mixup_prob=1.0, mixup_scale=(0.5, 1.5), copypaste_prob=0.0,
                 copypaste_scale=(0.5, 1.5), cutpaste_prob=0.0, cutoutR_prob=0.0):
        if copypaste_prob or cutpaste_prob:
            # both read dataset.object_cls / .back_cls (mosaic_detection.py:87-89), which no dataset class of the reference defines:
            # they cannot run there either
            raise NotImplementedError("copy-paste / cut-paste (probability 0 in every shipped config) are not built")
        self._dataset, self.img_size, self.preprocess = dataset, img_size, preprocess
        self.mosaic_prob, self.scale = mosaic_prob, mosaic_scale
        self.degrees, self.translate, self.shear, self.perspective = degrees, translate, shear, perspective
        self.mixup_prob, self.mixup_scale = mixup_prob, mixup_scale
        self.copypaste_prob, self.copypaste_scale = copypaste_prob, copypaste_scale
        self.cutpaste_prob, self.cutoutR_prob = cutpaste_prob, cutoutR_prob

    def __len__(self):
        return len(self._dataset)

    def _img(self, index):
        ds = self._dataset
        return _check_image(ds.imgs[index] if ds.imgs is not None else ds.load_resized_img(index))

    def _per_image(self, img, labels, mosaic):
        # the per-image augmentations (:86-91, :156-161): copy-paste and cut-paste (probability 0) only draw -- the copy-paste draw
        # of the mosaic branch sits behind `not len(_labels) == 0 and`, so an image without labels skips it -- the rounding cut-out
        # works on a copy of the image (:80, :154)
        if len(labels) != 0 or not mosaic:
            random.random()
        random.random()
        if random.random() < self.cutoutR_prob:
            img = cutout_rounding(img.clone(), labels, CR_NHOLE, CR_RATIO, CR_MIXUP, CR_IOA)
        return img

    # ---- one sample up to (and excluding) the final transform: (uint8 HWC device image, labels [n,5] xyxy+cls, extra)
    def _build(self, idx):
        ds = self._dataset
        if not random.random() < self.mosaic_prob:
            res, img_hw, _, name = ds.annotations[idx]
            ds.img_size = self.img_size
            img = self._per_image(self._img(idx), res, mosaic=False)
            return img, res, img_hw, name, False
        H, W = int(ds.img_size[0]), int(ds.img_size[1])
        yc = int(random.uniform(0.5 * H, 1.5 * H))
        xc = int(random.uniform(0.5 * W, 1.5 * W))
        members = [idx] + [random.randint(0, len(ds) - 1) for _ in range(3)]
        tiles = (MosaicTile * 4)()
        keep, parts, name = [], [], None
        for k, index in enumerate(members):
            boxes, _, _, name = ds.annotations[index]
            img = self._per_image(self._img(index), boxes, mosaic=True)
            keep.append(img)
            h0, w0 = int(img.shape[0]), int(img.shape[1])
            scale = min(1. * H / h0, 1. * W / w0)
            w, h = int(w0 * scale), int(h0 * scale)
            (lx1, ly1, lx2, ly2), (sx1, sy1, _, _) = mosaic_coordinate(k, xc, yc, w, h, H, W)
            t = tiles[k]
            t.src, t.h, t.w, t.dh, t.dw = img.data_ptr(), h0, w0, h, w
            t.lx1, t.ly1, t.lx2, t.ly2, t.sx1, t.sy1 = lx1, ly1, lx2, ly2, sx1, sy1
            moved = boxes.copy()
            if boxes.size > 0:
                moved[:, 0] = scale * boxes[:, 0] + (lx1 - sx1)
                moved[:, 1] = scale * boxes[:, 1] + (ly1 - sy1)
                moved[:, 2] = scale * boxes[:, 2] + (lx1 - sx1)
                moved[:, 3] = scale * boxes[:, 3] + (ly1 - sy1)
            parts.append(moved)
        canvas = torch.empty(2 * H, 2 * W, 3, dtype=torch.uint8, device=keep[0].device)
        call("plyolo_mosaic4", tiles, 2 * H, 2 * W, canvas.data_ptr(), _stream())
        labels = np.concatenate(parts, 0)
        for col, hi in ((0, 2 * W), (1, 2 * H), (2, 2 * W), (3, 2 * H)):
            np.clip(labels[:, col], 0, hi, out=labels[:, col])
        canvas, labels = random_perspective(canvas, labels, degrees=self.degrees, translate=self.translate, scale=self.scale,
                                            shear=self.shear, perspective=self.perspective, border=[-H // 2, -W // 2])
        if not len(labels) == 0 and random.random() < self.mixup_prob:
            canvas, labels = self.mixup(canvas, labels, self.img_size)
        return canvas, labels, None, name, True

    def mixup(self, origin_img, origin_labels, input_dim):
        """:169-247.  The partner image is letterboxed into input_dim (pad 114), the whole padded image rescaled by the jitter
        factor, mirrored on a coin flip, padded with zeros up to the target, cropped at a random offset and blended 0.5 / 0.5."""
        jit = random.uniform(*self.copypaste_scale)             # sic (:170): the reference's mixup jitters by copypaste_scale
        flip = random.uniform(0, 1) > 0.5
        other = []
        while len(other) == 0:
            k = random.randint(0, len(self) - 1)
            other = self._dataset.annotations[k][0]
        img = self._img(k)
        ih, iw = int(img.shape[0]), int(img.shape[1])
        r = min(input_dim[0] / ih, input_dim[1] / iw)
        boxed = resize_pad(img, (int(iw * r), int(ih * r)), (int(input_dim[0]), int(input_dim[1])), 114)
        bw, bh = int(boxed.shape[1] * jit), int(boxed.shape[0] * jit)
        boxed = resize_pad(boxed, (bw, bh))
        r *= jit
        th, tw = int(origin_img.shape[0]), int(origin_img.shape[1])
        x_off = y_off = 0
        if max(bh, th) > th:
            y_off = random.randint(0, max(bh, th) - th - 1)
        if max(bw, tw) > tw:
            x_off = random.randint(0, max(bw, tw) - tw - 1)
        out = torch.empty_like(origin_img)
        call("plyolo_mixup_blend_u8", origin_img.data_ptr(), th, tw, boxed.data_ptr(), bh, bw, int(flip), x_off, y_off, out.data_ptr(), _stream())
        b = other[:, :4].copy()
        b[:, 0::2] = np.clip(b[:, 0::2] * r, 0, bw)             # adjust_box_anns :250-253 (no padding offsets)
        b[:, 1::2] = np.clip(b[:, 1::2] * r, 0, bh)
        if flip:
            b[:, 0::2] = bw - b[:, 0::2][:, ::-1]
        b[:, 0::2] = np.clip(b[:, 0::2] - x_off, 0, tw)
        b[:, 1::2] = np.clip(b[:, 1::2] - y_off, 0, th)
        return out, np.vstack((origin_labels, np.hstack((b, other[:, 4:5].copy()))))

    def __getitem__(self, idx):
        img, labels, img_hw, name, mosaic = self._build(idx)
        if mosaic:
            mix_img, padded = self.preprocess(img, labels, self.img_size)
            return mix_img, padded, (int(mix_img.shape[1]), int(mix_img.shape[2])), np.array([idx]), name
        if self.preprocess is not None:
            img, labels = self.preprocess(img, labels, self.img_size)
        return img, labels, img_hw, np.array([idx]), name

    def batch(self, indices):
        """The samples `[self[i] for i in indices]` with ONE final-transform launch: (images [B,3,H,W] fp32, labels
        [B,max_labels,5], [info], indices, [names]).  Needs a TrainTransform as `preprocess`."""
        if not isinstance(self.preprocess, TrainTransform):
            raise PlyoloError("MosaicDetection.batch needs preprocess=TrainTransform(...)")
        imgs, decided, infos, names = [], [], [], []
        for idx in indices:
            img, labels, img_hw, name, mosaic = self._build(idx)
            decided.append(self.preprocess._decide(img, labels, self.img_size))    # its draws come right after the sample's
            imgs.append(img)
            infos.append((int(self.img_size[0]), int(self.img_size[1])) if mosaic else img_hw)
            names.append(name)
        out, _ = preproc_batch(imgs, self.img_size, flips=[d[0] for d in decided], gains=[d[1] for d in decided])
        return out, np.stack([d[2] for d in decided], 0), infos, np.asarray(list(indices)), names
