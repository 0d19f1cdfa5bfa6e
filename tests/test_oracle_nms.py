"""Known-answer + property tests for oracle/nms.py (torchvision semantics;
parity unpinned against the reference -- see the module header)."""
import numpy as np
import pytest
import torch

from oracle import nms as onms


def test_hand_cases():
    boxes = np.array([[0, 0, 10, 10], [1, 1, 11, 11], [20, 20, 30, 30], [0, 0, 10, 10]], dtype=np.float32)
    scores = np.array([0.9, 0.8, 0.7, 0.95], dtype=np.float32)
    # box3 (0.95) suppresses box0 (iou 1) and box1 (iou 81/119=0.68>0.5)
    assert onms.nms(boxes, scores, 0.5).tolist() == [3, 2]
    # strict '>' : iou exactly equal to thr is kept
    b = np.array([[0, 0, 2, 2], [1, 0, 3, 2]], dtype=np.float32)  # inter 2, union 6 -> 1/3
    s = np.array([0.9, 0.8], dtype=np.float32)
    assert onms.nms(b, s, np.float32(2.0) / np.float32(6.0)).tolist() == [0, 1]
    assert onms.nms(b, s, 0.33).tolist() == [0]
    # ties in score: lower index first
    assert onms.nms(boxes[[0, 2]], np.array([0.5, 0.5], np.float32), 0.5).tolist() == [0, 1]
    assert onms.nms(np.zeros((0, 4), np.float32), np.zeros((0,), np.float32), 0.5).shape == (0,)


def test_batched_is_class_aware_and_modes_agree():
    rng = np.random.default_rng(0)
    ctr = rng.uniform(50, 600, size=(60, 2))
    boxes = []
    for c in ctr:
        for _ in range(5):
            cc = c + rng.normal(0, 4, 2)
            wh = np.exp(rng.uniform(np.log(16), np.log(128), 2))
            boxes.append([cc[0] - wh[0] / 2, cc[1] - wh[1] / 2, cc[0] + wh[0] / 2, cc[1] + wh[1] / 2])
    boxes = np.asarray(boxes, np.float32)
    scores = rng.uniform(0.01, 1, len(boxes)).astype(np.float32)
    cls = rng.integers(0, 5, len(boxes)).astype(np.float32)
    trick = onms.batched_nms(boxes, scores, cls, 0.65, numel_threshold=10 ** 9)
    vanilla = onms.batched_nms(boxes, scores, cls, 0.65, numel_threshold=0)
    assert trick.tolist() == vanilla.tolist()
    # per-class brute force agrees
    keep = np.zeros(len(boxes), bool)
    for c in np.unique(cls):
        sel = np.nonzero(cls == c)[0]
        keep[sel[onms.nms_bruteforce(boxes[sel], scores[sel], 0.65)]] = True
    assert sorted(trick.tolist()) == np.nonzero(keep)[0].tolist()
    assert np.all(np.diff(scores[trick]) <= 0)


def test_greedy_vs_bruteforce_random():
    rng = np.random.default_rng(1)
    for n in (1, 2, 17, 200):
        xy = rng.uniform(0, 100, (n, 2))
        wh = rng.uniform(5, 60, (n, 2))
        b = np.concatenate([xy, xy + wh], 1).astype(np.float32)
        s = rng.uniform(0, 1, n).astype(np.float32)
        for thr in (0.3, 0.65):
            assert onms.nms(b, s, thr).tolist() == onms.nms_bruteforce(b, s, thr).tolist()


def test_postprocess_contract():
    rng = np.random.default_rng(2)
    B, A, C = 3, 500, 7
    p = np.zeros((B, A, 5 + C), np.float32)
    xy = rng.uniform(0, 300, (B, A, 2))
    wh = rng.uniform(10, 80, (B, A, 2))
    p[..., 0:2] = xy
    p[..., 2:4] = xy + wh
    p[..., 4] = rng.uniform(0, 1, (B, A))
    p[..., 5:] = rng.uniform(0, 1, (B, A, C))
    p[1, :, 4] = 0.0  # image 1: nothing passes -> None
    out = onms.postprocess(p, 0.3, 0.65)
    assert out[1] is None
    for i in (0, 2):
        d = out[i]
        assert d.shape[1] == 6 and d.shape[0] <= 300
        assert np.all(d[:, 4] >= 0.3) and np.all(np.diff(d[:, 4]) <= 0)
        cls = p[i, :, 5:].argmax(1)
        conf = p[i, :, 4] * p[i, :, 5:].max(1)
        # every output row is one of the input anchors with its conf/class
        for row in d[:5]:
            j = np.nonzero((p[i, :, 0] == row[0]) & (p[i, :, 1] == row[1]))[0][0]
            assert row[4] == conf[j] and row[5] == cls[j]
    # max_det cap
    out = onms.postprocess(p, 0.0, 1.0, max_det=10)
    assert out[0].shape[0] == 10


def _rand_dets(seed, counts, n_cls):
    import torch
    gen = torch.Generator().manual_seed(seed)
    outs = []
    for n in counts:
        if n is None:
            outs.append(None)
            continue
        xy = torch.rand(n, 2, generator=gen) * 500
        wh = torch.rand(n, 2, generator=gen) * 120 + 1
        conf = torch.rand(n, 1, generator=gen)
        cls = torch.randint(0, n_cls, (n, 1), generator=gen).float()
        outs.append(torch.cat([xy, xy + wh, conf, cls], 1))
    return outs


def test_format_outputs_hand_case():
    """postprocess.py:95-138 on numbers worked out by hand: a 480x640 (h x w) image letterboxed into 640x640
    has scale min(640/640, 640/480) = 1.0; a 960x1280 image has scale 0.5."""
    import torch
    from oracle import formatting as ofmt
    outs = [torch.tensor([[10.0, 20.0, 110.0, 220.0, 0.9, 2.0], [0.0, 0.0, 50.0, 40.0, 0.25, 0.0]]),
            None,
            torch.tensor([[100.0, 50.0, 300.0, 250.0, 0.5, 1.0]])]
    ids, hws = [7, 8, 9], ([480, 480, 960], [640, 640, 1280])
    class_ids = [1, 2, 3]
    js, det = ofmt.format_outputs(outs, ids, hws, (640, 640), class_ids, None)
    assert js == [
        {"image_id": 7, "category_id": 3, "bbox": [10.0, 20.0, 100.0, 200.0], "score": 0.8999999761581421, "segmentation": []},
        {"image_id": 7, "category_id": 1, "bbox": [0.0, 0.0, 50.0, 40.0], "score": 0.25, "segmentation": []},
        {"image_id": 9, "category_id": 2, "bbox": [200.0, 100.0, 400.0, 400.0], "score": 0.5, "segmentation": []},
    ]
    assert det[1][0].shape == (0, 5) and det[1][0].dtype == np.float64      # untouched default (postprocess.py:103)
    np.testing.assert_array_equal(det[0][2], np.array([[10.0, 20.0, 110.0, 220.0, 0.9]], dtype=np.float32))
    np.testing.assert_array_equal(det[2][1], np.array([[200.0, 100.0, 600.0, 500.0, 0.5]], dtype=np.float32))
    assert det[0][1].shape == (0, 5) and det[0][1].dtype == np.float32
    # the reference rescales the caller's tensors in place (:112-113)
    assert outs[2][0, :4].tolist() == [200.0, 100.0, 600.0, 500.0]


def test_format_outputs_product_refuses_cpu_tensors():
    """The product formats on the device (plyolo_format_detections); host tensors are refused, not silently handled."""
    import pl_yolo_amd
    from pl_yolo_amd.postprocess import format_outputs
    outs = _rand_dets(3, [4], 6)
    with pytest.raises(pl_yolo_amd.PlyoloError):
        format_outputs(outs, [1], ([480], [640]), (640, 640), [1, 2, 3, 5, 8, 13], None)


# ---- format_outputs: pinned to the reference function (tools/gen_golden.py: gen_format_outputs) ------------------
def _fmt_fixture():
    from conftest import load_golden
    g = load_golden("format_outputs")
    outs = [torch.from_numpy(g["in%d" % i].copy()) if ("in%d" % i) in g else None for i in range(4)]
    hws = ([int(v) for v in g["hs"]], [int(v) for v in g["ws"]])
    return g, outs, [int(v) for v in g["ids"]], hws, tuple(int(v) for v in g["val_size"]), [int(v) for v in g["class_ids"]]


def check_format_result(g, outs, json_list, det_list, n_cls):
    assert [j["image_id"] for j in json_list] == g["json_image_id"].tolist()
    assert [j["category_id"] for j in json_list] == g["json_category_id"].tolist()
    np.testing.assert_array_equal(np.asarray([j["bbox"] for j in json_list], dtype=np.float64), g["json_bbox"])
    np.testing.assert_array_equal(np.asarray([j["score"] for j in json_list], dtype=np.float64), g["json_score"])
    assert all(j["segmentation"] == [] for j in json_list)
    for i, o in enumerate(outs):
        if o is not None:
            np.testing.assert_array_equal(o.cpu().numpy(), g["after%d" % i])        # rescaled IN PLACE like the reference
        for c in range(n_cls):
            np.testing.assert_array_equal(np.asarray(det_list[i][c], dtype=np.float64).reshape(-1, 5), g["det_%d_%d" % (i, c)])


def test_format_outputs_vs_reference_fixture():
    from oracle import formatting as ofmt
    g, outs, ids, hws, val_size, class_ids = _fmt_fixture()
    json_list, det_list = ofmt.format_outputs(outs, ids, hws, val_size, class_ids, None)
    check_format_result(g, outs, json_list, det_list, len(class_ids))
