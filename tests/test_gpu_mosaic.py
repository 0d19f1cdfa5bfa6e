"""Mosaic / random-affine / mixup on the device (pl_yolo_amd.data.MosaicDetection, csrc/augment.hip) -- bit for bit against
  * tests/golden/mosaic_samples.npz: what the reference's own MosaicDetection + TrainTransform return (tools/gen_golden.py mosaic;
    OpenCV's calls served by the oracle's restatements), including the position of the `random` stream afterwards;
  * oracle/mosaic.py on random matrices / sizes the fixture does not hold."""
import os
import random
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import augment as oa, mosaic as om  # noqa: E402
from pl_yolo_amd import data as pdata  # noqa: E402
from pl_yolo_amd._lib import PlyoloError  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
FIX = np.load(os.path.join(ROOT, "tests", "golden", "mosaic_samples.npz"))
CASES = {"mix": dict(mosaic_prob=1.0, mixup_prob=1.0), "nomix": dict(mosaic_prob=1.0, mixup_prob=0.0),
         "plain": dict(mosaic_prob=0.0, mixup_prob=1.0), "coin": dict(mosaic_prob=0.5, mixup_prob=0.5),
         "persp": dict(mosaic_prob=1.0, mixup_prob=0.0, perspective=0.001)}      # cv2.warpPerspective with the affine matrix


class ToyDataset:
    def __init__(self, device=None, size=(48, 64)):
        n = len([k for k in FIX.files if k.startswith("ds_img")])
        host = [FIX["ds_img%d" % i].copy() for i in range(n)]
        self.imgs = [torch.from_numpy(h).to(device) for h in host] if device else host
        self.annotations = [(FIX["ds_lab%d" % i].copy(), host[i].shape[:2], host[i].shape[:2], "img%d" % i) for i in range(n)]
        self.img_size = size

    def __len__(self):
        return len(self.imgs)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.mark.parametrize("tag", sorted(CASES))
def test_device_samples_match_reference_fixture(tag):
    md = pdata.MosaicDetection(ToyDataset(DEV), (48, 64), preprocess=pdata.TrainTransform(max_labels=20, flip_prob=0.5, hsv_prob=1.0), **CASES[tag])
    seed = int(FIX["%s_seed" % tag])
    random.seed(seed)
    np.random.seed(seed)
    for k, idx in enumerate(FIX["%s_idx" % tag]):
        img, lab, info, ids, name = md[int(idx)]
        assert img.is_cuda and img.dtype == torch.float32
        assert np.array_equal(img.cpu().numpy(), FIX["%s_%d_img" % (tag, k)]), (tag, k)
        assert np.array_equal(np.asarray(lab, dtype=np.float32), FIX["%s_%d_labels" % (tag, k)]), (tag, k)
        assert tuple(info) == tuple(FIX["%s_%d_info" % (tag, k)]) and int(ids[0]) == int(idx)
    assert random.random() == float(FIX["%s_state" % tag])


def test_batch_equals_sequential_items():
    kw = dict(mosaic_prob=0.7, mixup_prob=0.6)
    tt = pdata.TrainTransform(max_labels=20, flip_prob=0.5, hsv_prob=1.0)
    idxs = [1, 4, 0, 6, 2, 5]
    random.seed(3); np.random.seed(3)
    md = pdata.MosaicDetection(ToyDataset(DEV), (48, 64), preprocess=tt, **kw)
    seq = [md[i] for i in idxs]
    random.seed(3); np.random.seed(3)
    md = pdata.MosaicDetection(ToyDataset(DEV), (48, 64), preprocess=tt, **kw)
    imgs, labels, infos, ids, names = md.batch(idxs)
    assert imgs.shape == (len(idxs), 3, 48, 64)
    for k, s in enumerate(seq):
        assert torch.equal(imgs[k], s[0]) and np.array_equal(labels[k], s[1]) and names[k] == s[4]
    # and the oracle agrees on the same seeds (a case the fixture does not hold)
    random.seed(3); np.random.seed(3)
    mo = om.MosaicDetection(ToyDataset(None), (48, 64), preprocess=oa.TrainTransform(max_labels=20, flip_prob=0.5, hsv_prob=1.0), **kw)
    for k, i in enumerate(idxs):
        want = mo[i]
        assert np.array_equal(imgs[k].cpu().numpy(), np.asarray(want[0], dtype=np.float32)), k
        assert np.array_equal(labels[k], np.asarray(want[1], dtype=np.float32)), k


@pytest.mark.parametrize("shape,dsize", [((37, 53), (64, 48)), ((96, 128), (64, 48)), ((20, 20), (31, 57)), ((640, 480), (320, 320))])
def test_warp_affine_vs_oracle_random_matrices(shape, dsize):
    rng = np.random.RandomState(shape[0] * 7 + dsize[0])
    img = rng.randint(0, 256, shape + (3,)).astype(np.uint8)
    random.seed(shape[1])
    for rep in range(6):
        M, s, width, height = om.affine_decision(shape, degrees=25, translate=0.2, scale=(0.4, 1.8), shear=8, border=(0, 0))
        got = pdata.warp_affine(dev(img), M[:2], dsize, 114).cpu().numpy()
        want = om.warp_affine_u8(img, M[:2], dsize, (114, 114, 114))
        assert np.array_equal(got, want), (rep, np.abs(got.astype(int) - want.astype(int)).max())
    ident = np.array([[1.0, 0, 0], [0, 1.0, 0]])
    assert np.array_equal(pdata.warp_affine(dev(img), ident, (shape[1], shape[0]), 0).cpu().numpy(), img)
    far = np.array([[1.0, 0, 1.0e6], [0, 1.0, -1.0e6]])          # everything maps outside: border only (int16 saturation path)
    assert (pdata.warp_affine(dev(img), far, dsize, 9).cpu().numpy() == 9).all()
    assert pdata.invert_affine(M[:2]) == list(om.invert_affine(M[:2]))


@pytest.mark.parametrize("shape,dsize", [((37, 53), (64, 48)), ((96, 128), (200, 48)), ((20, 20), (31, 57)), ((640, 480), (320, 320))])
def test_warp_perspective_vs_oracle_random_matrices(shape, dsize):
    """plyolo_warp_perspective_u8 == oracle bit for bit: the affine matrices the reference passes AND true projective ones."""
    rng = np.random.RandomState(shape[0] * 5 + dsize[0])
    img = rng.randint(0, 256, shape + (3,)).astype(np.uint8)
    random.seed(shape[1] + 1)
    for rep in range(6):
        M, s, width, height = om.affine_decision(shape, degrees=25, translate=0.2, scale=(0.4, 1.8), shear=8, border=(0, 0))
        if rep >= 3:      # a real perspective row
            M = M.copy()
            M[2, 0], M[2, 1] = rng.uniform(-2e-3, 2e-3, 2)
        got = pdata.warp_perspective(dev(img), M, dsize, 114).cpu().numpy()
        want = om.warp_perspective_u8(img, M, dsize, (114, 114, 114))
        assert np.array_equal(got, want), (rep, np.abs(got.astype(int) - want.astype(int)).max())
        assert pdata.invert_3x3(M) == list(om.invert_3x3(M))
    assert np.array_equal(pdata.warp_perspective(dev(img), np.eye(3), (shape[1], shape[0]), 0).cpu().numpy(), img)
    sing = np.zeros((3, 3))                                       # singular: cv::invert returns zeros, W = 0 everywhere -> pixel (0, 0)
    got = pdata.warp_perspective(dev(img), sing, dsize, 9).cpu().numpy()
    assert np.array_equal(got, om.warp_perspective_u8(img, sing, dsize, (9, 9, 9)))


@pytest.mark.parametrize("shape,dsize,out_hw", [((40, 60), (30, 20), (48, 64)), ((17, 91), (91, 17), None), ((50, 50), (125, 124), (130, 130))])
def test_resize_pad_vs_oracle(shape, dsize, out_hw):
    rng = np.random.RandomState(shape[0])
    img = rng.randint(0, 256, shape + (3,)).astype(np.uint8)
    got = pdata.resize_pad(dev(img), dsize, out_hw, 114).cpu().numpy()
    rs = om.resize(img, dsize)
    want = rs if out_hw is None else np.full(out_hw + (3,), 114, dtype=np.uint8)
    if out_hw is not None:
        want[:dsize[1], :dsize[0]] = rs
    assert np.array_equal(got, want)


def test_mixup_offsets_flip_and_zero_padding():
    """mixup on its own against the oracle's, with a partner both larger and smaller than the target."""
    for seed in range(8):
        ds_d, ds_h = ToyDataset(DEV), ToyDataset(None)
        origin = np.random.RandomState(seed).randint(0, 256, (48, 64, 3)).astype(np.uint8)
        lab = np.array([[3.0, 4, 30, 40, 7]])
        md = pdata.MosaicDetection(ds_d, (48, 64), copypaste_scale=(0.4, 1.9))
        mo = om.MosaicDetection(ds_h, (48, 64), copypaste_scale=(0.4, 1.9))
        random.seed(seed)
        got_img, got_lab = md.mixup(dev(origin), lab.copy(), (48, 64))
        random.seed(seed)
        want_img, want_lab = mo.mixup(origin.copy(), lab.copy(), (48, 64))
        assert np.array_equal(got_img.cpu().numpy(), want_img) and np.array_equal(got_lab, want_lab)


# ---- rounding cut-out: tests/golden/cutout_round.npz holds what the reference's own function / MosaicDetection return (tools/gen_golden.py cutout)
CUT = np.load(os.path.join(ROOT, "tests", "golden", "cutout_round.npz"))
CUT_MD = {"cmosaic": dict(mosaic_prob=1.0, mixup_prob=0.0, cutoutR_prob=1.0), "cplain": dict(mosaic_prob=0.0, cutoutR_prob=1.0),
          "ccoin": dict(mosaic_prob=0.5, mixup_prob=0.5, cutoutR_prob=0.6)}


@pytest.mark.parametrize("tag", ["interior", "edges", "none", "many", "tiny"])
def test_device_cutout_rounding_matches_the_reference_function(tag):
    lab = CUT["f_%s_labels" % tag]
    cur = dev(CUT["f_%s_img" % tag])
    np.random.seed(int(CUT["f_%s_seed" % tag]))
    for rep in range(4):
        cur = pdata.cutout_rounding(cur, lab)
        assert cur.dtype == torch.uint8 and np.array_equal(cur.cpu().numpy(), CUT["f_%s_out%d" % (tag, rep)]), (tag, rep)
    assert np.random.randint(0, 1 << 30) == int(CUT["f_%s_state" % tag])


@pytest.mark.parametrize("tag", sorted(CUT_MD))
def test_device_mosaic_with_cutout_matches_reference_samples(tag):
    ds = ToyDataset(DEV)
    before = [im.clone() for im in ds.imgs]
    md = pdata.MosaicDetection(ds, (48, 64), preprocess=pdata.TrainTransform(max_labels=20, flip_prob=0.5, hsv_prob=1.0), **CUT_MD[tag])
    seed = int(CUT["%s_seed" % tag])
    random.seed(seed)
    np.random.seed(seed)
    for k, idx in enumerate(CUT["%s_idx" % tag]):
        img, lab, info, ids, name = md[int(idx)]
        assert np.array_equal(img.cpu().numpy(), CUT["%s_%d_img" % (tag, k)]), (tag, k)
        assert np.array_equal(np.asarray(lab, dtype=np.float32), CUT["%s_%d_labels" % (tag, k)]), (tag, k)
    assert [random.random(), float(np.random.randint(0, 1 << 30))] == [float(v) for v in CUT["%s_state" % tag]]
    assert all(torch.equal(a, b) for a, b in zip(before, ds.imgs))      # the data set's images are never written (the cut-out works on copies)


def test_device_cutout_vs_oracle_at_640_with_many_boxes_and_holes():
    """A 640 x 480 image, 40 boxes (some on the borders), 30 calls on one stream with up to 12 holes per call (more
    than one launch's PLYOLO_MAX_HOLES): device == oracle bit for bit, same draws."""
    rng = np.random.RandomState(3)
    h, w = 480, 640
    img = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
    x1 = rng.uniform(0, w * 0.8, 40); y1 = rng.uniform(0, h * 0.8, 40)
    lab = np.stack([x1, y1, np.minimum(x1 + rng.uniform(4, 60, 40), w), np.minimum(y1 + rng.uniform(4, 60, 40), h), rng.randint(0, 80, 40).astype(np.float64)], 1)
    lab[0, :4] = [0.0, 0.5, 30.0, 20.0]
    lab[1, :4] = [600.0, 440.0, 640.0, 480.0]
    ratio = [[0.02, 0.03], [0.05, 0.02], [0.01, 0.01], [0.04, 0.04]]
    a, b = img.copy(), dev(img)
    np.random.seed(9)
    for _ in range(30):
        a = om.cutout_rounding(a, lab, (4, 12), ratio, 0.7, 0.2)
    st = np.random.randint(0, 1 << 30)
    np.random.seed(9)
    for _ in range(30):
        b = pdata.cutout_rounding(b, lab, (4, 12), ratio, 0.7, 0.2)
    assert np.random.randint(0, 1 << 30) == st
    assert (a != img).any() and np.array_equal(b.cpu().numpy(), a)


def test_refusals():
    with pytest.raises(NotImplementedError):
        pdata.MosaicDetection(ToyDataset(DEV), (48, 64), cutpaste_prob=0.1)
    with pytest.raises(NotImplementedError):
        pdata.MosaicDetection(ToyDataset(DEV), (48, 64), copypaste_prob=0.1)
    md = pdata.MosaicDetection(ToyDataset(None), (48, 64), preprocess=pdata.TrainTransform())   # host arrays: no CPU path
    with pytest.raises(PlyoloError):
        md[0]


def test_full_size_640_samples_vs_oracle_and_throughput():
    """Real sizes (images up to 640x640, 1280x1280 canvas): the device sample builder against the oracle, bit for bit, and a
    throughput print (samples per second of one host process; the reference runs cv2 on 6 DataLoader workers, coco.py:85)."""
    import time
    rng = np.random.RandomState(77)
    host = []
    ann = []
    for i in range(8):
        h, w = int(rng.randint(360, 641)), int(rng.randint(360, 641))
        host.append(rng.randint(0, 256, (h, w, 3)).astype(np.uint8))
        k = int(rng.randint(1, 9))
        x1 = rng.uniform(0, w * 0.7, k); y1 = rng.uniform(0, h * 0.7, k)
        lab = np.stack([x1, y1, np.minimum(x1 + rng.uniform(8, w * 0.3, k), w), np.minimum(y1 + rng.uniform(8, h * 0.3, k), h),
                        rng.randint(0, 80, k).astype(np.float64)], 1)
        ann.append((lab, (h, w), (h, w), "im%d" % i))

    class DS:
        def __init__(self, imgs):
            self.imgs, self.annotations, self.img_size = imgs, [(a[0].copy(),) + a[1:] for a in ann], (640, 640)

        def __len__(self):
            return len(self.imgs)
    kw = dict(mosaic_prob=1.0, mixup_prob=1.0)
    md = pdata.MosaicDetection(DS([torch.from_numpy(h).to(DEV) for h in host]), (640, 640), preprocess=pdata.TrainTransform(max_labels=120), **kw)
    mo = om.MosaicDetection(DS([h.copy() for h in host]), (640, 640), preprocess=oa.TrainTransform(max_labels=120), **kw)
    random.seed(5); np.random.seed(5)
    imgs, labels, infos, ids, names = md.batch([0, 5, 2])
    random.seed(5); np.random.seed(5)
    for k, i in enumerate([0, 5, 2]):
        want = mo[i]
        assert np.array_equal(imgs[k].cpu().numpy(), np.asarray(want[0], dtype=np.float32)), k
        assert np.array_equal(labels[k], np.asarray(want[1], dtype=np.float32)), k
    torch.cuda.synchronize()
    idx = [i % 8 for i in range(64)]
    md.batch(idx[:8])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in range(0, 64, 32):
        md.batch(idx[b:b + 32])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("mosaic + affine + mixup + train transform, 640x640: %.0f samples/s in one host process (%.2f ms per sample)" % (64 / dt, dt / 64 * 1e3))
