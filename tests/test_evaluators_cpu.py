"""CPU: the COCO evaluation feed (pl_yolo_amd/evaluators.py) -- the reference wraps pycocotools (eval_coco.py:8-36), which is
not installed here, so the built-in numpy implementation of the same protocol is checked on hand-computable cases; and the
CPU restatement of the input transforms (oracle/augment.py) on its structural properties."""
import numpy as np
import pytest

from pl_yolo_amd.evaluators import COCOEvaluator, coco_bbox_eval


def _gt():
    return {"images": [{"id": 1}, {"id": 2}],
            "categories": [{"id": 1}, {"id": 5}],
            "annotations": [
                {"id": 1, "image_id": 1, "category_id": 1, "bbox": [10, 10, 50, 40], "area": 2000, "iscrowd": 0},
                {"id": 2, "image_id": 1, "category_id": 5, "bbox": [100, 100, 30, 30], "area": 900, "iscrowd": 0},
                {"id": 3, "image_id": 2, "category_id": 1, "bbox": [20, 30, 60, 60], "area": 3600, "iscrowd": 0}]}


def test_perfect_detections_score_one():
    gt = _gt()
    dets = [{"image_id": a["image_id"], "category_id": a["category_id"], "bbox": list(a["bbox"]), "score": 0.9, "segmentation": []}
            for a in gt["annotations"]]
    ap, ap50, info = coco_bbox_eval(dets, gt)
    assert ap == pytest.approx(1.0) and ap50 == pytest.approx(1.0) and "Average Precision" in info


def test_known_precision_recall_curve():
    """Category 1: two ground truths; detections in score order TP, FP, TP -> precision envelope 1 up to recall .5, 2/3 up to 1:
    AP = (51 * 1 + 50 * 2/3) / 101 at every IoU threshold the boxes pass.  A box shifted to IoU 0.6 passes thresholds .5 and .55 only."""
    gt = _gt()
    gt["annotations"] = [a for a in gt["annotations"] if a["category_id"] == 1]
    gt["categories"] = [{"id": 1}]
    dets = [{"image_id": 1, "category_id": 1, "bbox": [10, 10, 50, 40], "score": 0.9},
            {"image_id": 1, "category_id": 1, "bbox": [300, 300, 20, 20], "score": 0.8},
            {"image_id": 2, "category_id": 1, "bbox": [20, 30, 60, 60], "score": 0.7}]
    ap, ap50, _ = coco_bbox_eval(dets, gt)
    want = (51 * 1.0 + 50 * (2.0 / 3.0)) / 101
    assert ap == pytest.approx(want) and ap50 == pytest.approx(want)
    # second true positive shifted by 16 px: IoU = 44*60 / (2*3600 - 44*60) = 0.579 -> matches at .50 / .55 only
    dets[2]["bbox"] = [36, 30, 60, 60]
    ap2, ap50_2, _ = coco_bbox_eval(dets, gt)
    one_tp = 51 / 101          # recall .5 reached with precision 1, nothing beyond
    assert ap50_2 == pytest.approx(want)
    assert ap2 == pytest.approx((2 * want + 8 * one_tp) / 10)


def test_crowd_region_absorbs_detections_and_evaluator_feed():
    gt = _gt()
    gt["annotations"] = [a for a in gt["annotations"] if a["category_id"] == 5]
    gt["categories"] = [{"id": 5}]
    gt["annotations"].append({"id": 4, "image_id": 2, "category_id": 5, "bbox": [0, 0, 200, 200], "area": 40000, "iscrowd": 1})
    dets = [{"image_id": 1, "category_id": 5, "bbox": [100, 100, 30, 30], "score": 0.9},
            {"image_id": 2, "category_id": 5, "bbox": [10, 10, 20, 20], "score": 0.95},      # inside the crowd region: ignored, not a FP
            {"image_id": 2, "category_id": 5, "bbox": [50, 50, 20, 20], "score": 0.85}]
    ap, ap50, _ = coco_bbox_eval(dets, gt)
    assert ap == pytest.approx(1.0)

    class DS:
        coco = gt
    assert COCOEvaluator([], DS) == (0.0, 0.0, "No detection!")
    a, b, info = COCOEvaluator(dets, DS)
    assert a == pytest.approx(1.0) and b == pytest.approx(1.0)


# ---- oracle/augment.py: structural properties of the restated OpenCV algorithms -------------------------------------
def test_augment_oracle_properties():
    import random
    from oracle import augment as oa
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    # identity resize reproduces the image; constant images stay constant under any resize
    assert np.array_equal(oa.resize_linear_u8(img, 53, 37), img)
    flat = np.full((20, 30, 3), 77, np.uint8)
    assert np.all(oa.resize_linear_u8(flat, 47, 11) == 77)
    # letterbox: ratio, pad value, CHW float32
    out, r = oa.preproc(img, (64, 64))
    assert out.shape == (3, 64, 64) and out.dtype == np.float32 and r == pytest.approx(64 / 53)
    dh, dw = int(37 * r), int(53 * r)
    assert np.all(out[:, dh:, :] == 114) and np.all(out[:, :, dw:] == 114)
    # HSV round trip with unit gains: within the quantisation of 8-bit H (2 degrees) / S / V
    rt = img.copy()
    oa.augment_hsv(rt, gains=(1.0, 1.0, 1.0))
    assert np.abs(rt.astype(int) - img.astype(int)).max() <= 8
    grey = np.repeat(rng.integers(0, 256, (5, 5, 1), dtype=np.uint8), 3, 2)
    g2 = grey.copy()
    oa.augment_hsv(g2, gains=(1.3, 0.5, 1.0))
    assert np.array_equal(g2, grey)                      # no saturation -> hue / saturation gains change nothing
    # value gain 0.5 halves a pure colour
    px = np.array([[[0, 0, 200]]], np.uint8)
    oa.augment_hsv(px, gains=(1.0, 1.0, 0.5))
    assert px[0, 0].tolist() == [0, 0, 100]
    # labels: the train transform keeps (cls, cx, cy, w, h) * r and mirrors boxes with the image
    random.seed(3)
    np.random.seed(3)
    t = oa.TrainTransform(max_labels=6, flip_prob=1.0, hsv_prob=0.0)
    targets = np.array([[5.0, 6.0, 25.0, 30.0, 2.0]], np.float32)
    im, lab = t(img, targets.copy(), (64, 64))
    x1, x2 = 53 - 25.0, 53 - 5.0
    np.testing.assert_allclose(lab[0], [2.0, (x1 + x2) / 2 * r, 18.0 * r, 20.0 * r, 24.0 * r], rtol=1e-6)
    assert np.all(lab[1:] == 0) and im.shape == (3, 64, 64)
    assert np.array_equal(im, oa.preproc(img[:, ::-1], (64, 64))[0])
