"""oracle/mosaic.py (mosaic + random affine + mixup) against tests/golden/mosaic_samples.npz: what the reference's own
MosaicDetection / TrainTransform return in the build container when OpenCV's calls are served by the oracle's restatements
(tools/gen_golden.py mosaic).  Pins control flow, draw order from `random` / `numpy.random`, label arithmetic, padding and
blending; the OpenCV pixel primitives themselves are unpinned and get structural checks here."""
import os
import random
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import augment as oa, mosaic as om  # noqa: E402

FIX = np.load(os.path.join(ROOT, "tests", "golden", "mosaic_samples.npz"))
CASES = {"mix": dict(mosaic_prob=1.0, mixup_prob=1.0), "nomix": dict(mosaic_prob=1.0, mixup_prob=0.0),
         "plain": dict(mosaic_prob=0.0, mixup_prob=1.0), "coin": dict(mosaic_prob=0.5, mixup_prob=0.5),
         "persp": dict(mosaic_prob=1.0, mixup_prob=0.0, perspective=0.001)}      # cv2.warpPerspective with the affine matrix


class ToyDataset:
    """The data set stored in the fixture, in the shape MosaicDetection reads."""

    def __init__(self, size=(48, 64)):
        n = len([k for k in FIX.files if k.startswith("ds_img")])
        self.imgs = [FIX["ds_img%d" % i].copy() for i in range(n)]
        self.annotations = [(FIX["ds_lab%d" % i].copy(), self.imgs[i].shape[:2], self.imgs[i].shape[:2], "img%d" % i) for i in range(n)]
        self.img_size = size

    def __len__(self):
        return len(self.imgs)


@pytest.mark.parametrize("tag", sorted(CASES))
def test_oracle_matches_reference_samples(tag):
    md = om.MosaicDetection(ToyDataset(), (48, 64), preprocess=oa.TrainTransform(max_labels=20, flip_prob=0.5, hsv_prob=1.0), **CASES[tag])
    seed = int(FIX["%s_seed" % tag])
    random.seed(seed)
    np.random.seed(seed)
    for k, idx in enumerate(FIX["%s_idx" % tag]):
        img, lab, info, ids, name = md[int(idx)]
        assert np.array_equal(np.asarray(img, dtype=np.float32), FIX["%s_%d_img" % (tag, k)]), (tag, k)
        assert np.array_equal(np.asarray(lab, dtype=np.float32), FIX["%s_%d_labels" % (tag, k)]), (tag, k)
        assert tuple(info) == tuple(FIX["%s_%d_info" % (tag, k)])
        assert int(ids[0]) == int(idx)
    assert random.random() == float(FIX["%s_state" % tag])      # same number of draws as the reference


def test_unbuilt_augmentations_raise():
    with pytest.raises(NotImplementedError):
        om.MosaicDetection(ToyDataset(), (48, 64), copypaste_prob=0.5)
    with pytest.raises(NotImplementedError):
        om.MosaicDetection(ToyDataset(), (48, 64), cutpaste_prob=0.5)


# ---- rounding cut-out: the reference function itself ran here (numpy only) -- tests/golden/cutout_round.npz, tools/gen_golden.py cutout
CUT = np.load(os.path.join(ROOT, "tests", "golden", "cutout_round.npz"))
CUT_FN = ("interior", "edges", "none", "many", "tiny")
CUT_MD = {"cmosaic": dict(mosaic_prob=1.0, mixup_prob=0.0, cutoutR_prob=1.0), "cplain": dict(mosaic_prob=0.0, cutoutR_prob=1.0),
          "ccoin": dict(mosaic_prob=0.5, mixup_prob=0.5, cutoutR_prob=0.6)}


@pytest.mark.parametrize("tag", CUT_FN)
def test_cutout_rounding_matches_the_reference_function(tag):
    """Four calls on one numpy-random stream, each on the previous result: fill colour from the strips around the boxes (boxes on
    the image borders lose strips; no boxes: untouched), holes over holes, holes rejected by the box overlap, draw count."""
    lab = CUT["f_%s_labels" % tag]
    cur = CUT["f_%s_img" % tag].copy()
    np.random.seed(int(CUT["f_%s_seed" % tag]))
    for rep in range(4):
        cur = om.cutout_rounding(cur, lab, om.CR_NHOLE, om.CR_RATIO, om.CR_MIXUP, om.CR_IOA)
        assert cur.dtype == np.uint8 and np.array_equal(cur, CUT["f_%s_out%d" % (tag, rep)]), (tag, rep)
    assert np.random.randint(0, 1 << 30) == int(CUT["f_%s_state" % tag])


def test_cutout_fixture_exercises_accepts_rejects_and_overlaps():
    """The fixture is not vacuous: holes were accepted (pixels changed), some draws were rejected (fewer changed rectangles than
    drawn holes in at least one call), and the no-label case is the identity."""
    assert np.array_equal(CUT["f_none_out3"], CUT["f_none_img"])
    for tag in ("interior", "edges", "many", "tiny"):
        assert (CUT["f_%s_out3" % tag] != CUT["f_%s_img" % tag]).any()
    lab = CUT["f_many_labels"]
    np.random.seed(int(CUT["f_many_seed"]))
    h, w = CUT["f_many_img"].shape[:2]
    drawn = accepted = 0
    for rep in range(4):
        for _ in range(np.random.randint(om.CR_NHOLE[0], om.CR_NHOLE[1] + 1)):
            x1, y1, index = np.random.randint(0, w), np.random.randint(0, h), np.random.randint(0, len(om.CR_RATIO))
            x2 = int(np.clip(x1 + om.CR_RATIO[index][0] * w, x1, w)); y2 = int(np.clip(y1 + om.CR_RATIO[index][1] * h, y1, h))
            drawn += 1
            accepted += bool(om.bbox_ioa([x1, y1, x2, y2], lab[:, :4]).max() < om.CR_IOA)
    assert 0 < accepted < drawn


@pytest.mark.parametrize("tag", sorted(CUT_MD))
def test_oracle_mosaic_with_cutout_matches_reference_samples(tag):
    md = om.MosaicDetection(ToyDataset(), (48, 64), preprocess=oa.TrainTransform(max_labels=20, flip_prob=0.5, hsv_prob=1.0), **CUT_MD[tag])
    seed = int(CUT["%s_seed" % tag])
    random.seed(seed)
    np.random.seed(seed)
    for k, idx in enumerate(CUT["%s_idx" % tag]):
        img, lab, info, ids, name = md[int(idx)]
        assert np.array_equal(np.asarray(img, dtype=np.float32), CUT["%s_%d_img" % (tag, k)]), (tag, k)
        assert np.array_equal(np.asarray(lab, dtype=np.float32), CUT["%s_%d_labels" % (tag, k)]), (tag, k)
    assert [random.random(), float(np.random.randint(0, 1 << 30))] == [float(v) for v in CUT["%s_state" % tag]]


def test_mosaic_coordinates_tile_the_canvas():
    """The four quadrants meet at (xc, yc), never overlap, and each shows the part of its image nearest the centre."""
    H, W = 48, 64
    for (xc, yc, w, h) in [(40, 30, 64, 40), (90, 70, 30, 48), (64, 48, 64, 48), (33, 71, 50, 20)]:
        seen = np.zeros((2 * H, 2 * W), dtype=np.int32)
        for k in range(4):
            (x1, y1, x2, y2), (sx1, sy1, sx2, sy2) = om.get_mosaic_coordinate(k, xc, yc, w, h, H, W)
            assert x2 - x1 == sx2 - sx1 and y2 - y1 == sy2 - sy1
            assert 0 <= x1 <= x2 <= 2 * W and 0 <= y1 <= y2 <= 2 * H and 0 <= sx1 <= sx2 <= w and 0 <= sy1 <= sy2 <= h
            assert (x2 == xc if k in (0, 2) else x1 == xc) and (y2 == yc if k in (0, 1) else y1 == yc)
            seen[y1:y2, x1:x2] += 1
        assert seen.max() <= 1


def test_warp_affine_structure():
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, (20, 30, 3)).astype(np.uint8)
    ident = np.array([[1.0, 0, 0], [0, 1.0, 0]])
    assert np.array_equal(om.warp_affine_u8(img, ident, (30, 20)), img)
    # integer translation: a shifted copy, the uncovered part filled with the border value
    out = om.warp_affine_u8(img, np.array([[1.0, 0, 5], [0, 1.0, 3]]), (30, 20), (114, 114, 114))
    assert np.array_equal(out[3:, 5:], img[:-3, :-5]) and (out[:3] == 114).all() and (out[:, :5] == 114).all()
    # half-pixel shift: the rounded mean of neighbouring columns (weights 16/32 each)
    out = om.warp_affine_u8(img, np.array([[1.0, 0, 0.5], [0, 1.0, 0]]), (30, 20), (0, 0, 0))
    want = (img[:, :-1].astype(np.int64) * 16384 + img[:, 1:].astype(np.int64) * 16384 + (1 << 14)) >> 15
    assert np.array_equal(out[:, 1:], want.astype(np.uint8))
    # an output larger than the source is border outside the source
    out = om.warp_affine_u8(img, ident, (40, 25), (7, 8, 9))
    assert np.array_equal(out[:20, :30], img) and (out[22:, :, 0] == 7).all() and (out[:, 32:, 2] == 9).all()
    # inverse of the inverse
    M = np.array([[0.9, -0.2, 3.0], [0.15, 1.1, -2.0]])
    inv = om.invert_affine(M).reshape(2, 3)
    assert np.allclose(om.invert_affine(inv).reshape(2, 3), M, atol=1e-12)
    R = om.get_rotation_matrix_2d((0, 0), 90.0, 2.0)
    assert np.allclose(R, [[0, 2, 0], [-2, 0, 0]], atol=1e-12)


def test_warp_perspective_structure():
    """The warpPerspective restatement: exact on identity / integer shifts, the projective division is honoured, and on an
    AFFINE matrix (all the reference ever passes, mosaic_detection.py:319-323) it lands within one coordinate quantum
    (1/32 px) of warpAffine -- same picture, different rounding path."""
    rng = np.random.RandomState(1)
    img = rng.randint(0, 256, (40, 150, 3)).astype(np.uint8)        # wider than two 64-column blocks
    assert np.array_equal(om.warp_perspective_u8(img, np.eye(3), (150, 40)), img)
    T = np.eye(3); T[0, 2], T[1, 2] = 7.0, 2.0
    out = om.warp_perspective_u8(img, T, (150, 40), (114, 114, 114))
    assert np.array_equal(out[2:, 7:], img[:-2, :-7]) and (out[:2] == 114).all() and (out[:, :7] == 114).all()
    # a uniform scale written into the projective row: M = diag(1, 1, 0.5) doubles every coordinate
    P = np.diag([1.0, 1.0, 0.5])
    out = om.warp_perspective_u8(img, P, (150, 40), (0, 0, 0))
    S2 = np.diag([2.0, 2.0, 1.0])
    assert np.array_equal(out, om.warp_perspective_u8(img, S2, (150, 40), (0, 0, 0)))
    assert np.array_equal(out[::2, ::2][:20, :75], img[:20, :75])
    # smooth image + affine matrix: the two entry points agree to the interpolation quantum
    yy, xx = np.mgrid[0:40, 0:150]
    smooth = np.stack([xx + yy, 2 * yy + 40, 255 - xx], -1).astype(np.uint8)
    random.seed(3)
    for _ in range(4):
        M, s, width, height = om.affine_decision((40, 150), degrees=10, translate=0.1, scale=(0.7, 1.3), shear=4, border=(0, 0))
        a = om.warp_affine_u8(smooth, M[:2], (150, 40), (114, 114, 114)).astype(int)
        b = om.warp_perspective_u8(smooth, M, (150, 40), (114, 114, 114)).astype(int)
        inner = np.abs(a - b)[(a != 114).all(-1) & (b != 114).all(-1)]
        assert np.percentile(inner, 99) <= 2, np.percentile(inner, 99)
    inv = om.invert_3x3(M).reshape(3, 3)
    assert np.allclose(inv @ M, np.eye(3), atol=1e-12)


def test_legacy_val_transform():
    """data_augments.py:72-76: RGB order, /255, ImageNet mean / std."""
    rng = np.random.RandomState(2)
    img = rng.randint(0, 256, (30, 40, 3)).astype(np.uint8)
    t = np.array([[2.0, 3, 20, 25, 1]])
    plain, lab = oa.ValTransform(max_labels=4)(img, t, (32, 48))
    leg, lab2 = oa.ValTransform(legacy=True, max_labels=4)(img, t, (32, 48))
    assert np.array_equal(lab, lab2) and leg.dtype == np.float32
    mean, std = np.array([0.485, 0.456, 0.406]).reshape(3, 1, 1), np.array([0.229, 0.224, 0.225]).reshape(3, 1, 1)
    np.testing.assert_allclose(leg, (plain[::-1] / 255.0 - mean) / std, rtol=0, atol=2e-6)


def test_affine_labels_keep_boxes_under_identity_and_drop_slivers():
    t = np.array([[10.0, 10, 30, 40, 3], [5, 5, 6, 30, 1]])
    out = om.affine_labels(t.copy(), np.eye(3), 1.0, 64, 48)
    assert np.array_equal(out, t[:1])                        # the 1-px-wide box fails the 2-px test of box_candidates
    M = np.eye(3); M[0, 2] = 100.0
    assert len(om.affine_labels(t.copy(), M, 1.0, 64, 48)) == 0   # pushed out of the frame: clipped to zero width
