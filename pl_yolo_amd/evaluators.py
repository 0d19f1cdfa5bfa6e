"""The COCO evaluation feed of the validation loop (reference models/evaluators/eval_coco.py:8-36, called from
PL_Modules/pl_detection.py:90-91 with the json records `format_outputs` produced).

    COCOEvaluator(json_list, val_dataset) -> (AP@[.5:.95], AP@.5, summary text)

With pycocotools installed this is the reference's code path (loadRes + COCOeval "bbox").  pycocotools is not available
in every image (not in the build container), so the same protocol is also implemented here on numpy (`coco_bbox_eval`):
per image and category the detections in score order (at most 100), greedy matching at IoU .50:.05:.95 with crowd
regions matchable many times and ignored, precision envelope sampled at 101 recall points, mean over categories and
thresholds.  That implementation is restated from pycocotools' published algorithm (cocoeval.py: evaluateImg /
accumulate) and is PARITY-UNPINNED against pycocotools itself; it is CPU accuracy tooling, not the hot path."""
import contextlib
import io
import json
import tempfile

import numpy as np


def _iou_xywh(d, g, crowd):
    """IoU of detections d [n,4] with ground truths g [m,4] (x, y, w, h); for crowd gts the union is the detection area."""
    if len(d) == 0 or len(g) == 0:
        return np.zeros((len(d), len(g)))
    dx2, dy2 = d[:, 0] + d[:, 2], d[:, 1] + d[:, 3]
    gx2, gy2 = g[:, 0] + g[:, 2], g[:, 1] + g[:, 3]
    iw = np.clip(np.minimum(dx2[:, None], gx2[None]) - np.maximum(d[:, 0][:, None], g[:, 0][None]), 0, None)
    ih = np.clip(np.minimum(dy2[:, None], gy2[None]) - np.maximum(d[:, 1][:, None], g[:, 1][None]), 0, None)
    inter = iw * ih
    da, ga = d[:, 2] * d[:, 3], g[:, 2] * g[:, 3]
    union = np.where(np.asarray(crowd, bool)[None], da[:, None], da[:, None] + ga[None] - inter)
    return inter / np.maximum(union, 1e-12)


def coco_bbox_eval(json_list, gt, max_dets=100):
    """gt: COCO dict ({"images", "annotations", "categories"}).  Returns (AP, AP50, summary)."""
    iou_thrs = np.linspace(0.5, 0.95, 10)
    rec_thrs = np.linspace(0.0, 1.0, 101)
    cats = sorted(c["id"] for c in gt["categories"])
    imgs = sorted(im["id"] for im in gt["images"])
    gts, dts = {}, {}
    for a in gt["annotations"]:
        gts.setdefault((a["image_id"], a["category_id"]), []).append(a)
    for dct in json_list:
        dts.setdefault((dct["image_id"], dct["category_id"]), []).append(dct)
    T = len(iou_thrs)
    precision = -np.ones((T, len(cats)))
    for ci, cat in enumerate(cats):
        scores, matched, ignored, npos = [], [], [], 0
        for img in imgs:
            g = gts.get((img, cat), [])
            d = sorted(dts.get((img, cat), []), key=lambda x: -x["score"])[:max_dets]
            if not g and not d:
                continue
            # bbox evaluation: pycocotools' _prepare overwrites every annotation's 'ignore' with its 'iscrowd', so a dataset's own
            # 'ignore' flags do not count (cocoeval.py: gt['ignore'] = 'iscrowd' in gt and gt['iscrowd'])
            g_ig = np.array([int(a.get("iscrowd", 0)) for a in g], dtype=int)
            order = np.argsort(g_ig, kind="mergesort")               # non-ignored gts first
            g = [g[k] for k in order]
            g_ig = g_ig[order]
            crowd = [int(a.get("iscrowd", 0)) for a in g]
            ious = _iou_xywh(np.array([x["bbox"] for x in d], float).reshape(-1, 4), np.array([a["bbox"] for a in g], float).reshape(-1, 4), crowd)
            gtm = -np.ones((T, len(g)), dtype=int)
            dtm = -np.ones((T, len(d)), dtype=int)
            dt_ig = np.zeros((T, len(d)), dtype=bool)
            for t, thr in enumerate(iou_thrs):
                for di in range(len(d)):
                    best, m = min(thr, 1 - 1e-10), -1
                    for gi in range(len(g)):
                        if gtm[t, gi] >= 0 and not crowd[gi]:
                            continue
                        if m > -1 and g_ig[m] == 0 and g_ig[gi] == 1:
                            break
                        if ious[di, gi] < best:
                            continue
                        best, m = ious[di, gi], gi
                    if m == -1:
                        continue
                    dt_ig[t, di] = bool(g_ig[m])
                    dtm[t, di] = m
                    gtm[t, m] = di
            scores.append(np.array([x["score"] for x in d], float))
            matched.append(dtm >= 0)
            ignored.append(dt_ig)
            npos += int((g_ig == 0).sum())
        if npos == 0:
            continue
        sc = np.concatenate(scores) if scores else np.zeros(0)
        inds = np.argsort(-sc, kind="mergesort")
        dm = np.concatenate(matched, 1)[:, inds] if matched else np.zeros((T, 0), bool)
        di_ = np.concatenate(ignored, 1)[:, inds] if ignored else np.zeros((T, 0), bool)
        tps = np.cumsum(dm & ~di_, 1).astype(float)
        fps = np.cumsum(~dm & ~di_, 1).astype(float)
        for t in range(T):
            tp, fp = tps[t], fps[t]
            rc = tp / npos
            pr = tp / np.maximum(tp + fp, np.spacing(1))
            pr = pr.tolist()
            for k in range(len(pr) - 1, 0, -1):          # precision envelope
                if pr[k] > pr[k - 1]:
                    pr[k - 1] = pr[k]
            idx = np.searchsorted(rc, rec_thrs, side="left")
            q = np.zeros(len(rec_thrs))
            for ri, pi in enumerate(idx):
                if pi < len(pr):
                    q[ri] = pr[pi]
            precision[t, ci] = q.mean()
    valid = precision[:, (precision > -1).any(0)]
    ap = float(valid.mean()) if valid.size else -1.0
    ap50 = float(valid[0].mean()) if valid.size else -1.0
    info = (" Average Precision  (AP) @[ IoU=0.50:0.95 | area=   all | maxDets=%d ] = %.3f\n"
            " Average Precision  (AP) @[ IoU=0.50      | area=   all | maxDets=%d ] = %.3f\n" % (max_dets, ap, max_dets, ap50))
    return ap, ap50, info


def _gt_dict(coco):
    """A COCO ground-truth dict from a pycocotools COCO object or a plain dict."""
    if isinstance(coco, dict):
        return coco
    return getattr(coco, "dataset")


def COCOEvaluator(json_list, val_dataset):
    """eval_coco.py:8-36.  detections: the json records of format_outputs (bbox = x1, y1, w, h)."""
    cocoGt = val_dataset.coco
    if len(json_list) == 0:
        return 0.0, 0.0, "No detection!"
    try:
        from pycocotools.cocoeval import COCOeval
    except ImportError:
        import warnings
        warnings.warn("pycocotools is not installed: mAP comes from pl_yolo_amd.evaluators.coco_bbox_eval, a restatement of "
                      "COCOeval('bbox') that is not pinned against pycocotools itself", RuntimeWarning, stacklevel=2)
        return coco_bbox_eval(json_list, _gt_dict(cocoGt))
    annType = ["segm", "bbox", "keypoints"]
    _, tmp = tempfile.mkstemp()
    json.dump(json_list, open(tmp, "w"), skipkeys=True, ensure_ascii=True)
    cocoDt = cocoGt.loadRes(tmp)
    cocoEval = COCOeval(cocoGt, cocoDt, annType[1])
    cocoEval.evaluate()
    cocoEval.accumulate()
    redirect_string = io.StringIO()
    with contextlib.redirect_stdout(redirect_string):
        cocoEval.summarize()
    return cocoEval.stats[0], cocoEval.stats[1], redirect_string.getvalue()
