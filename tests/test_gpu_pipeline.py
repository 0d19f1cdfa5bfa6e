"""-m gpu: the stages either side of the detector and the NMS edge cases.

  * format_outputs on the device (plyolo_format_detections) against the fixture written by the REFERENCE function;
  * the input pipeline (plyolo_preproc_batch: HSV jitter + mirror + letterbox + pad 114 + CHW fp32) against oracle/augment.py,
    bit for bit, and TrainTransform / ValTransform label parity under identical RNG seeds;
  * NMS: both scan paths (LDS-resident for <= 1024 candidates, global for more), score ties, IoU exactly at the threshold,
    zero-area boxes, the per-class branch above 20 000 coordinates, class offsets at fp32 rounding scale."""
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import pl_yolo_amd  # noqa: E402
from pl_yolo_amd import data as pdata  # noqa: E402
from pl_yolo_amd.postprocess import format_outputs, postprocess  # noqa: E402
from oracle import augment as oa, nms as onms  # noqa: E402
import hiputil as hu  # noqa: E402
from test_oracle_nms import _fmt_fixture, check_format_result  # noqa: E402


def test_format_outputs_device_vs_reference_fixture():
    g, outs, ids, hws, val_size, class_ids = _fmt_fixture()
    dev_outs = [o.to(hu.DEV) if o is not None else None for o in outs]
    json_list, det_list = format_outputs(dev_outs, ids, hws, val_size, class_ids, None)
    check_format_result(g, dev_outs, json_list, det_list, len(class_ids))
    # views of ONE [B, 300, 6] buffer, as postprocess returns them (row pitch 6, different bases)
    base = torch.zeros(4, 300, 6, device=hu.DEV)
    views = []
    for i, o in enumerate(outs):
        if o is None:
            views.append(None)
        else:
            base[i, :o.shape[0]] = torch.from_numpy(g["in%d" % i]).to(hu.DEV)
            views.append(base[i, :o.shape[0]])
    json_list, det_list = format_outputs(views, ids, hws, val_size, class_ids, None)
    check_format_result(g, views, json_list, det_list, len(class_ids))
    assert format_outputs([None, None], [1, 2], ([10, 10], [10, 10]), (416, 416), class_ids, None)[0] == []


@pytest.mark.parametrize("size", [(64, 64), (416, 416), (96, 160)])
def test_preproc_batch_vs_oracle_bit_exact(size):
    rng = np.random.default_rng(5)
    shapes = [(37, 53), (480, 640), (500, 375), (64, 64), (1, 7)]
    imgs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in shapes]
    flips = [False, True, False, True, True]
    gains = [None, (1.01, 0.6, 1.3), (0.99, 1.5, 0.7), None, (1.0, 1.0, 1.0)]
    dev = [torch.from_numpy(im).to(hu.DEV) for im in imgs]
    out, ratios = pdata.preproc_batch(dev, size, flips=flips, gains=gains)
    torch.cuda.synchronize()
    assert tuple(out.shape) == (5, 3, size[0], size[1]) and out.dtype == torch.float32
    for i, im in enumerate(imgs):
        ref = im.copy()
        if gains[i] is not None:
            oa.augment_hsv(ref, gains=gains[i])
        if flips[i]:
            ref = ref[:, ::-1]
        want, r = oa.preproc(np.ascontiguousarray(ref), size)
        assert r == pytest.approx(ratios[i])
        got = out[i].cpu().numpy()
        bad = int((got != want).sum())
        assert bad == 0, "image %d: %d of %d values differ (max %.0f)" % (i, bad, want.size, np.abs(got - want).max())


def test_transforms_labels_and_pixels_follow_the_oracle_under_the_same_seeds():
    rng = np.random.default_rng(9)
    imgs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in ((120, 90), (60, 200), (333, 500))]
    tgts = [np.array([[10, 12, 60, 70, 3], [30, 5, 80, 50, 7]], np.float32), np.zeros((0, 5), np.float32),
            np.array([[100, 100, 101.5, 300, 1], [20, 40, 400, 300, 2]], np.float32)]
    for seed in (0, 1, 2, 3):
        random.seed(seed)
        np.random.seed(seed)
        t_ref = oa.TrainTransform(max_labels=8, flip_prob=0.5, hsv_prob=0.7)
        want = [t_ref(im.copy(), t.copy(), (128, 128)) for im, t in zip(imgs, tgts)]
        random.seed(seed)
        np.random.seed(seed)
        t_hip = pdata.TrainTransform(max_labels=8, flip_prob=0.5, hsv_prob=0.7)
        got = [t_hip(torch.from_numpy(im).to(hu.DEV), t.copy(), (128, 128)) for im, t in zip(imgs, tgts)]
        for (wi, wl), (gi, gl) in zip(want, got):
            np.testing.assert_array_equal(gl, wl)
            assert np.array_equal(gi.cpu().numpy(), wi)
    v_ref, v_hip = oa.ValTransform(max_labels=8), pdata.ValTransform(max_labels=8)
    out, labels = v_hip.batch([torch.from_numpy(im).to(hu.DEV) for im in imgs], tgts, (96, 96))
    for i, (im, t) in enumerate(zip(imgs, tgts)):
        wi, wl = v_ref(im, t, (96, 96))
        assert np.array_equal(out[i].cpu().numpy(), wi)
        np.testing.assert_array_equal(labels[i], wl)
    # legacy=True (data_augments.py:72-76): RGB + /255 + ImageNet mean / std, bit for bit what numpy leaves in the fp32 array
    v_ref, v_hip = oa.ValTransform(legacy=True, max_labels=8), pdata.ValTransform(legacy=True, max_labels=8)
    out, labels = v_hip.batch([torch.from_numpy(im).to(hu.DEV) for im in imgs], tgts, (96, 96))
    for i, (im, t) in enumerate(zip(imgs, tgts)):
        wi, wl = v_ref(im, t, (96, 96))
        assert np.array_equal(out[i].cpu().numpy(), wi)
        np.testing.assert_array_equal(labels[i], wl)
    with pytest.raises(pl_yolo_amd.PlyoloError):
        pdata.preproc(torch.zeros(8, 8, 3, dtype=torch.uint8), (32, 32))          # CPU tensor: no fallback


# ---- NMS edge cases: HIP == oracle on adversarial inputs -------------------------------------------------------------
def _pred_from(dets, C_):
    """[n, 6] (x1,y1,x2,y2,score,cls) -> prediction rows with obj = 1 and the class score at its slot."""
    n = dets.shape[0]
    p = np.zeros((1, n, 5 + C_), np.float32)
    p[0, :, :4] = dets[:, :4]
    p[0, :, 4] = 1.0
    p[0, np.arange(n), 5 + dets[:, 5].astype(int)] = dets[:, 4]
    return p


def _check(dets, C_, conf=0.001, thr=0.5, agnostic=False):
    pred = _pred_from(dets, C_)
    want = onms.postprocess(pred, conf, thr, class_agnostic=agnostic)[0]
    got = postprocess(torch.from_numpy(pred).to(hu.DEV), conf, thr, class_agnostic=agnostic)[0]
    if want is None:
        assert got is None
        return 0
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    return want.shape[0]


@pytest.mark.parametrize("n", [700, 1024, 1025, 3000, 5400])
def test_nms_both_scan_paths_random_clusters(n):
    rng = np.random.default_rng(n)
    c = np.repeat(rng.uniform(50, 600, (n // 5 + 1, 2)), 5, 0)[:n] + rng.normal(0, 3, (n, 2))
    wh = np.exp(rng.uniform(np.log(10), np.log(120), (n, 2)))
    dets = np.concatenate([c - wh / 2, c + wh / 2, rng.uniform(0.01, 1, (n, 1)), rng.integers(0, 7, (n, 1))], 1).astype(np.float32)
    kept = _check(dets, 7, thr=0.65)
    assert 0 < kept <= 300
    _check(dets, 7, thr=0.65, agnostic=True)


def test_nms_score_ties_keep_the_lower_index():
    """Equal scores: torchvision sorts stably descending, i.e. the earlier box wins the tie and suppresses the later one."""
    b = np.array([[10, 10, 60, 60], [12, 12, 62, 62], [200, 200, 240, 240], [11, 11, 61, 61], [201, 201, 241, 241]], np.float32)
    dets = np.concatenate([b, np.full((5, 1), 0.5, np.float32), np.zeros((5, 1), np.float32)], 1)
    pred = _pred_from(dets, 3)
    got = postprocess(torch.from_numpy(pred).to(hu.DEV), 0.01, 0.5)[0].cpu().numpy()
    np.testing.assert_array_equal(got[:, :4], b[[0, 2]])
    _check(dets, 3, conf=0.01)
    many = np.tile(dets, (300, 1))           # 1500 tied boxes: both kernels, same rule
    many[:, :4] += np.repeat(np.arange(300, dtype=np.float32)[:, None] * 0.0, 5, 0)
    _check(many, 3, conf=0.01)


def test_nms_iou_exactly_at_threshold_is_kept():
    """Suppression needs IoU > thr (strict): two boxes with IoU exactly 0.5 both survive at thr 0.5, the second dies at 0.4999."""
    dets = np.array([[0, 0, 2, 2, 0.9, 1], [0, 0, 2, 1, 0.8, 1]], np.float32)      # inter 2, union 4
    assert _check(dets, 2, thr=0.5) == 2
    assert _check(dets, 2, thr=0.4999) == 1


def test_nms_zero_area_boxes_and_empty_images():
    """Degenerate boxes: IoU with anything is 0/x = 0 (or 0/0 = NaN for two coincident points: NaN > thr is false) -> kept."""
    dets = np.array([[5, 5, 5, 5, 0.9, 0], [5, 5, 5, 5, 0.8, 0], [0, 0, 10, 10, 0.7, 0], [3, 3, 3, 9, 0.6, 0]], np.float32)
    assert _check(dets, 1, thr=0.3) == 4
    pred = np.zeros((2, 50, 8), np.float32)                       # nothing passes the confidence filter
    assert postprocess(torch.from_numpy(pred).to(hu.DEV), 0.5, 0.5) == [None, None]


def test_nms_class_offsets_at_fp32_rounding_scale():
    """Coordinate trick: boxes + cls * (max_coord + 1) in fp32.  With coordinates ~1e5 and 80 classes the offsets reach 8e6,
    where fp32 resolves 0.5-1 px: the shifted boxes of a high class lose their sub-pixel part, exactly as torchvision's would."""
    rng = np.random.default_rng(2)
    n = 800
    c = np.repeat(rng.uniform(9.0e4, 1.0e5, (n // 4, 2)), 4, 0) + rng.normal(0, 2.5, (n, 2))
    wh = rng.uniform(8, 40, (n, 2))
    dets = np.concatenate([c - wh / 2, c + wh / 2, rng.uniform(0.05, 1, (n, 1)), rng.integers(60, 80, (n, 1))], 1).astype(np.float32)
    _check(dets, 80, thr=0.5)


def test_nms_per_class_branch_above_20000_coordinates():
    """More than 20 000 box coordinates (5001+ boxes): torchvision's batched_nms loops over the classes instead of shifting."""
    rng = np.random.default_rng(4)
    n = 5200
    c = np.repeat(rng.uniform(20, 500, (n // 8, 2)), 8, 0) + rng.normal(0, 2, (n, 2))
    wh = rng.uniform(10, 60, (n, 2))
    dets = np.concatenate([c - wh / 2, c + wh / 2, rng.uniform(0.05, 1, (n, 1)), rng.integers(0, 4, (n, 1))], 1).astype(np.float32)
    assert 4 * n > 20000
    _check(dets, 4, thr=0.5)
