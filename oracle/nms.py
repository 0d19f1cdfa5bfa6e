"""CPU restatement of the eval post-processing (TEST ORACLE).

`postprocess` follows /root/reference/models/evaluators/postprocess.py:7-48.
The NMS arithmetic itself lives in an UN-VENDORED, UNPINNED third party:
`torchvision.ops.batched_nms` / `torchvision.ops.nms` (requirements.txt:6 of
the reference names `torchvision` without a version; it is not installable in
the build container).  PARITY UNPINNED against the reference for this file:
the reference holds no test or golden vector at that boundary.  The algorithm
restated here is torchvision's published one (torchvision/ops/boxes.py and
csrc/ops/cpu/nms_kernel.cpp, stable since 0.9):

  nms(boxes, scores, thr): order = stable argsort(scores, descending);
      greedy: keep i, then suppress every later j with
      inter/(area_i + area_j - inter) > thr   (strict >; areas (x2-x1)(y2-y1),
      inter = max(0,xx2-xx1)*max(0,yy2-yy1));  result in descending score order.
  batched_nms(boxes, scores, idxs, thr):
      if boxes.numel() > (4000 on CPU | 20000 on GPU): per-class nms, kept
          indices re-sorted by descending score  ("vanilla")
      else: offsets = idxs * (boxes.max() + 1); nms(boxes + offsets[:,None])
          ("coordinate trick", fp32 arithmetic)

The product runs on the GPU, so `numel_threshold` defaults to 20000.
Known-answer tests are build-authored (tests/test_oracle_nms.py: hand cases +
an independent brute-force formulation).
"""
import numpy as np


def _areas(b):
    return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])


def nms(boxes, scores, thr):
    """Greedy NMS; boxes [n,4] float32 xyxy, scores [n] float32 -> int64 keep
    indices in descending-score order (ties: lower index first)."""
    boxes = np.asarray(boxes, dtype=np.float32)
    scores = np.asarray(scores, dtype=np.float32)
    n = boxes.shape[0]
    if n == 0:
        return np.zeros((0,), dtype=np.int64)
    order = np.argsort(-scores, kind="stable")
    b = boxes[order]
    areas = _areas(b)
    dead = np.zeros(n, dtype=bool)
    keep = []
    thr32 = np.float32(thr)
    for i in range(n):
        if dead[i]:
            continue
        keep.append(order[i])
        if i + 1 == n:
            break
        xx1 = np.maximum(b[i, 0], b[i + 1:, 0])
        yy1 = np.maximum(b[i, 1], b[i + 1:, 1])
        xx2 = np.minimum(b[i, 2], b[i + 1:, 2])
        yy2 = np.minimum(b[i, 3], b[i + 1:, 3])
        w = np.maximum(np.float32(0), xx2 - xx1)
        h = np.maximum(np.float32(0), yy2 - yy1)
        inter = w * h
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = inter / (areas[i] + areas[i + 1:] - inter)
        dead[i + 1:] |= ovr > thr32
    return np.asarray(keep, dtype=np.int64)


def batched_nms(boxes, scores, idxs, thr, numel_threshold=20000):
    boxes = np.asarray(boxes, dtype=np.float32)
    scores = np.asarray(scores, dtype=np.float32)
    idxs = np.asarray(idxs)
    if boxes.shape[0] == 0:
        return np.zeros((0,), dtype=np.int64)
    if boxes.size > numel_threshold:
        keep_mask = np.zeros(boxes.shape[0], dtype=bool)
        for c in np.unique(idxs):
            sel = np.nonzero(idxs == c)[0]
            keep_mask[sel[nms(boxes[sel], scores[sel], thr)]] = True
        kept = np.nonzero(keep_mask)[0]
        return kept[np.argsort(-scores[kept], kind="stable")].astype(np.int64)
    max_coordinate = boxes.max()
    offsets = idxs.astype(np.float32) * (max_coordinate + np.float32(1))
    return nms(boxes + offsets[:, None], scores, thr)


def postprocess(predictions, conf_thre=0.7, nms_thre=0.45, class_agnostic=False,
                numel_threshold=20000, max_det=300, max_nms=10000):
    """predictions [B,A,5+C] float32 (x1,y1,x2,y2,obj,cls...) as produced by the
    eval decode.  Returns a list (len B) of float32 [n,6] arrays
    (x1,y1,x2,y2,conf,cls) or None -- postprocess.py:11-48."""
    predictions = np.asarray(predictions, dtype=np.float32)
    out = [None] * predictions.shape[0]
    for i in range(predictions.shape[0]):
        p = predictions[i]
        if p.shape[0] == 0:
            continue
        cls_pred = p[:, 5:].argmax(1)
        cls_conf = p[np.arange(p.shape[0]), 5 + cls_pred]
        conf = p[:, 4] * cls_conf
        det = np.concatenate([p[:, :4], conf[:, None], cls_pred[:, None].astype(np.float32)], 1)
        det = det[conf >= np.float32(conf_thre)]
        if det.shape[0] > max_nms:
            det = det[:max_nms]
        if det.shape[0] == 0:
            continue
        if class_agnostic:
            keep = nms(det[:, :4], det[:, 4], nms_thre)
        else:
            keep = batched_nms(det[:, :4], det[:, 4], det[:, 5], nms_thre, numel_threshold)
        det = det[keep]
        if det.shape[0] > max_det:
            det = det[:max_det]
        out[i] = det
    return out


def nms_bruteforce(boxes, scores, thr):
    """Independent formulation used only to cross-check `nms` in the tests:
    a box is kept iff no KEPT box that precedes it in the stable descending
    order overlaps it by more than thr (fixed-point definition, evaluated with
    a full pairwise IoU matrix)."""
    boxes = np.asarray(boxes, dtype=np.float32)
    scores = np.asarray(scores, dtype=np.float32)
    n = boxes.shape[0]
    order = np.argsort(-scores, kind="stable")
    b = boxes[order]
    a = _areas(b)
    x1 = np.maximum(b[:, None, 0], b[None, :, 0])
    y1 = np.maximum(b[:, None, 1], b[None, :, 1])
    x2 = np.minimum(b[:, None, 2], b[None, :, 2])
    y2 = np.minimum(b[:, None, 3], b[None, :, 3])
    inter = np.maximum(np.float32(0), x2 - x1) * np.maximum(np.float32(0), y2 - y1)
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = inter / (a[:, None] + a[None, :] - inter)
    kept = np.zeros(n, dtype=bool)
    for j in range(n):
        kept[j] = not np.any(kept[:j] & (iou[:j, j] > np.float32(thr)))
    return order[kept].astype(np.int64)
