"""-m gpu: every BASELINE.json configuration at its REAL size (the oracle needs minutes to hours there, so the
checks are size-independent properties; the small-size fixture parity lives in test_gpu_network / test_gpu_yolov7):

  cfg1  YOLOX-nano 416x416 batch 4     HIP fp32 parity mode vs the reference-generated fixture (loss 1e-4) --
                                       the CPU side of this config is tests/test_oracle_cfg1.py
  cfg3  YOLOv7 640x640 batch 32        with and without `neck.repconv`
  cfg4  YOLOX-l 640x640 batch 16/GPU
  cfg5  YOLOX-x 1280x1280 batch 16/GPU train step + eval decode + postprocess on the SURVEY 8d set-(i) head maps

  * finite losses / a finite, non-zero gradient for every trained parameter
  * repeatability          same state + batch -> same losses, same gradients (fixed-order or fp64 reductions)
  * backward linearity     d(2*loss) == 2*d(loss)   (power-of-two scale commutes with every rounding)
  * batch permutation      fp32 parity mode: bit-identical head maps, equal losses / gradients
  * closed-form decode     model.eval() output == decode of the raw maps of the same weights

These are also the launches that exercise the 32-bit offset guards of the conv loaders (api.hip: check_conv) and
the weight-gradient slab planning at 1280x1280.
"""
import os

import numpy as np
import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu

import pl_yolo_amd  # noqa: E402
from conftest import ROOT, load_golden  # noqa: E402
import hiputil as hu  # noqa: E402

NC = 80


def _cfg(family, name, repconv=False):
    with open(os.path.join(ROOT, "configs", "model", family, name + ".yaml")) as f:
        cfg = yaml.safe_load(f)
    if repconv:
        cfg["neck"]["repconv"] = True
    return cfg


def _batch(B, size, seed, num_gt=30, max_gt=100):
    gen = torch.Generator().manual_seed(seed)
    imgs = torch.rand(B, 3, size, size, generator=gen) * 255
    labels = torch.zeros(B, max_gt, 5)
    labels[:, :num_gt, 0] = torch.randint(0, NC, (B, num_gt), generator=gen).float()
    labels[:, :num_gt, 1:3] = (0.15 + 0.7 * torch.rand(B, num_gt, 2, generator=gen)) * size
    labels[:, :num_gt, 3:5] = 8 + torch.rand(B, num_gt, 2, generator=gen) * 0.3 * size
    return imgs.to(hu.DEV), labels.to(hu.DEV)


def _build(cfg, dtype="bf16"):
    torch.manual_seed(96)
    model = pl_yolo_amd.build_model(cfg, NC)
    model.compute_dtype = dtype
    model = model.to(hu.DEV)
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    return model, sd0


def _step(model, sd0, imgs, labels, scale=1.0):
    model.load_state_dict(sd0)
    model.train()
    model.zero_grad(set_to_none=True)
    out = model(imgs, labels)
    (out["loss"] * scale).backward()
    torch.cuda.synchronize()
    trained = [(n, p) for n, p in model.named_parameters() if p.grad is not None]
    g = torch.cat([p.grad.flatten() for _, p in trained]).clone()
    losses = {k: float(v.detach().reshape(-1)[0]) if torch.is_tensor(v) else float(v) for k, v in out.items()}
    return losses, g, trained


def _dead(name):
    # the reference's unused Bottleneck.bn (network_blocks.py:81) never receives a gradient
    return name.endswith(".bn.weight") or name.endswith(".bn.bias")


def _check_step_properties(model, sd0, imgs, labels, tag):
    l1, g1, trained = _step(model, sd0, imgs, labels)
    print(tag, "losses", l1)
    assert all(np.isfinite(v) for v in l1.values()), l1
    assert bool(torch.isfinite(g1).all())
    missing = [n for n, p in model.named_parameters() if p.grad is None and not _dead(n)]
    assert not missing, missing[:5]
    zero = [n for n, p in trained if float(p.grad.abs().max()) == 0.0]
    assert not zero, zero[:5]
    gmax = float(g1.abs().max())
    l2, g2, _ = _step(model, sd0, imgs, labels)                 # repeatability
    for k in l1:
        assert abs(l1[k] - l2[k]) <= 1e-6 * max(1.0, abs(l1[k])), (k, l1[k], l2[k])
    assert float((g1 - g2).abs().max()) <= 1e-5 * gmax
    l3, g3, _ = _step(model, sd0, imgs, labels, scale=2.0)      # backward linearity
    assert float((g3 - 2 * g1).abs().max()) <= 2e-5 * gmax
    return l1, g1


def _check_permutation_fp32(model, sd0, imgs, labels, tag):
    B = imgs.shape[0]
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(8)).to(hu.DEV)
    pimgs, plabels = imgs[perm].contiguous(), labels[perm].contiguous()
    prev = model.compute_dtype
    model.compute_dtype = "fp32"
    try:
        model.load_state_dict(sd0)
        model.train()
        with torch.no_grad():
            m1 = [m.float().clone() for m in model(imgs, None)]
            model.load_state_dict(sd0)
            m2 = [m.float().clone() for m in model(pimgs, None)]
        for a, b in zip(m1, m2):
            assert torch.equal(a[perm], b)
        l1, g1, _ = _step(model, sd0, imgs, labels)
        l2, g2, _ = _step(model, sd0, pimgs, plabels)
        for k in l1:
            assert abs(l1[k] - l2[k]) <= 5e-6 * max(1.0, abs(l1[k])), (k, l1[k], l2[k])
        cs = hu.cossim(g1, g2)
        print(tag, "permuted batch (fp32): loss %.6f vs %.6f, gradient cosine %.7f" % (l1["loss"], l2["loss"], cs))
        assert cs >= 0.99999 and float((g1 - g2).abs().max()) <= 2e-4 * float(g1.abs().max())
    finally:
        model.compute_dtype = prev


def _decode_yolox(maps, strides):
    rows = []
    for m, s in zip(maps, strides):
        b, c, h, w = m.shape
        ys, xs = torch.meshgrid(torch.arange(h, device=m.device), torch.arange(w, device=m.device), indexing="ij")
        t = m.float().permute(0, 2, 3, 1).reshape(b, h * w, c)
        xy = (t[..., 0:2] + torch.stack([xs, ys], -1).reshape(1, h * w, 2)) * s
        wh = torch.exp(t[..., 2:4]) * s
        rows.append(torch.cat([xy - wh / 2, xy + wh / 2, torch.sigmoid(t[..., 4:])], -1))
    return torch.cat(rows, 1)


def _check_eval_decode_yolox(model, sd0, imgs, A):
    model.load_state_dict(sd0)
    model.eval()
    with torch.no_grad():
        pred = model(imgs, torch.zeros(imgs.shape[0], 1, 5, device=hu.DEV))
        maps = model(imgs, None)
    assert tuple(pred.shape) == (imgs.shape[0], A, 5 + NC)
    want = _decode_yolox(maps, (8, 16, 32))
    fin = torch.isfinite(want)
    err = float(((pred - want)[fin]).abs().max() / want[fin].abs().max())
    print("eval decode vs closed form: rel max err %.3g" % err)
    assert err <= 1e-5
    model.train()


# ---------------------------------------------------------------------------------------------- cfg1
CFG1_BF16_LOSS_TOL = 5e-3     # bf16 loss of the random-initialised nano net against the reference's fp32 loss (round 3 allowed 3e-2)
CFG1_BF16_COS_MIN = 0.95      # class-prediction bias gradient (a sum over all anchors), bf16 against the reference's fp32


def test_cfg1_nano416_b4_fp32_vs_reference_fixture():
    """YOLOX-nano 416x416 batch 4 (the reference's CPU-runnable configuration) through the HIP fp32 parity mode:
    same seeded weights and synthetic batch as the reference run that wrote tests/golden/cfg1_nano416.npz."""
    g = load_golden("cfg1_nano416")
    cfg = _cfg("yolox", "yolox_nano")
    model, sd0 = _build(cfg, "fp32")
    w = model.backbone.stem.conv.conv.weight.detach().cpu().numpy()
    assert np.array_equal(w, g["stem_weight"]), "seed-96 initialisation differs from the reference's"
    psum = float(sum(p.double().sum() for p in model.parameters()))
    assert abs(psum - float(g["param_sum"])) <= 1e-6 * float(g["param_abs_sum"])
    B, S = int(g["batch"]), int(g["size"])
    gen = torch.Generator().manual_seed(int(g["seed_data"]))
    imgs = torch.rand(B, 3, S, S, generator=gen) * 255
    labels = torch.zeros(B, 100, 5)
    labels[:, :30, 0] = torch.randint(0, NC, (B, 30), generator=gen).float()
    labels[:, :30, 1:3] = (0.15 + 0.7 * torch.rand(B, 30, 2, generator=gen)) * S
    labels[:, :30, 3:5] = 8 + torch.rand(B, 30, 2, generator=gen) * 0.3 * S
    model.train()
    out = model(imgs.to(hu.DEV), labels.to(hu.DEV))
    out["loss"].backward()
    torch.cuda.synchronize()
    for k in ("loss", "loss_iou", "loss_obj", "loss_cls"):
        got, want = float(out[k]), float(g["out/" + k])
        print("cfg1", k, got, want)
        assert abs(got - want) <= 1e-4 * max(1.0, abs(want)), k
    assert abs(float(out["proportion"]) - float(np.asarray(g["out/proportion"]).reshape(-1)[0])) <= 1e-5
    params = dict(model.named_parameters())
    for k in [k for k in g if k.startswith("grad/")]:
        ref = torch.from_numpy(g[k])
        got = params[k[5:]].grad.cpu()
        assert float((got - ref).abs().max()) <= 2e-4 * max(float(ref.abs().max()), 1e-6), k
    gsq = float(sum((p.grad.double() ** 2).sum() for p in model.parameters() if p.grad is not None))
    assert abs(gsq - float(g["grad_sq_sum"])) <= 1e-3 * float(g["grad_sq_sum"])
    # the bf16 MFMA path at this configuration: same batch, loss within the bf16 band
    m16, _ = _build(cfg, "bf16")
    m16.train()
    o16 = m16(imgs.to(hu.DEV), labels.to(hu.DEV))
    o16["loss"].backward()
    torch.cuda.synchronize()
    l16 = float(o16["loss"])
    # ... and its gradients.  This net is RANDOM-initialised at batch 4: its SimOTA costs are near-ties, bf16 storage noise flips
    # assignments (the foreground proportion moves 0.925 -> 0.892) and the BatchNorm chain amplifies the difference layer by layer
    # -- measured with tools/diag_cfg1.py: bf16 against the HIP fp32 step (== the fixture) cosine 1.000 / 0.979 on the objectness /
    # class prediction biases, 0.3 ... 0.9 inside the head, ~0 in the backbone.  So the asserted instruments are the loss (5e-3,
    # round 3 allowed 3e-2), the foreground proportion and the two bias gradients, which sum over all anchors; the convolution
    # gradients of the bf16 path are pinned where the comparison is meaningful: the warm fixtures (test_warm_yolox_s_bf16_end_to_end:
    # all-parameter cosine >= 0.9995 against the reference's fp32 step)
    p16 = dict(m16.named_parameters())
    cosb = {}
    for k in [k for k in g if k.startswith("grad/")]:
        ref, got = torch.from_numpy(g[k]).double(), p16[k[5:]].grad.cpu().double()
        c = float((ref * got).sum()) / max(float(ref.norm() * got.norm()), 1e-30)
        cosb[k[5:]] = c
        print("cfg1 bf16 gradient cosine vs reference %-40s %.5f" % (k[5:], c))
    print("cfg1 bf16 loss %.5f vs reference %.5f (rel %.2e); proportion %.4f vs %.4f"
          % (l16, float(g["out/loss"]), abs(l16 - float(g["out/loss"])) / float(g["out/loss"]), float(o16["proportion"]), float(g["out/proportion"])))
    assert abs(l16 - float(g["out/loss"])) <= CFG1_BF16_LOSS_TOL * float(g["out/loss"])
    assert abs(float(o16["proportion"]) - float(g["out/proportion"])) <= 0.06
    assert cosb["head.obj_preds.2.bias"] >= 0.99 and cosb["head.cls_preds.0.bias"] >= CFG1_BF16_COS_MIN


# ---------------------------------------------------------------------------------------------- cfg3 / cfg4 / cfg5 at full width, pinned
WIDE_FP32_MAP_TOL = 1e-3       # fp32 parity mode, raw head maps, max error relative to the largest value: random-initialised nets 100-200 convolutions
                               # deep amplify the summation-order difference between the plain-FMA kernels and ATen (measured 6e-5 ... 5.4e-4; the losses hold 1e-4)
WIDE_FP32_GRAD_TOL = 1e-2      # ... and the stored gradients, max error of an element relative to the tensor's largest (measured: prediction biases 2e-4, the
                               # first convolution -- the end of the longest backward chain -- 1e-3 ... 3.5e-3, one neck convolution of YOLOv7 4.5e-3) ...
WIDE_FP32_GRAD_COS = 1.0e-5    # ... 1 - cosine of every stored gradient, and
WIDE_FP32_NORM_TOL = 2e-3      # the L2 norm of EVERY parameter's gradient (measured <= 9e-4)
# bf16 MFMA path on these RANDOM-initialised full-width nets.  Yardstick: the REFERENCE itself under torch.autocast(cpu, bfloat16) against its own
# fp32 run on the same weights and batch (tools/diag_wide_bf16_ref.py) -- raw head maps rel-rms 0.064 / 0.077 / 0.080 (yolox_l), 0.076 / 0.084 / 0.085
# (yolox_x), 0.65 / 0.87 / 0.95 (yolov7: its maps are ~0 at initialisation); loss 1.1e-2 / 1.8e-2 / 5e-7; and its GRADIENTS decorrelate completely
# (all-parameter cosine 0.14 / -0.04 / 0.03, prediction-bias cosines down to -1: the deep random-initialised BatchNorm chain amplifies the rounding noise
# and flips label assignments).  So the maps and the loss are asserted at the reference's own bf16 level, gradient cosines are printed only; the bf16
# gradients of the wide kernels are asserted where the comparison means something: on warm weights (test_gpu_yolov7.py::test_full_width_warm_bf16_tracks_fp32_and_does_not_depend_on_fusions)
WIDE_BF16_LOSS_TOL = 3e-2      # measured 1.4e-2 (yolox_l)
WIDE_BF16_MAP_RMS = {"yolox": 0.12, "yolov7": 1.2}     # measured 0.054 ... 0.086 (yolox_l / yolox_x)


def _wide_cfg(name):
    return _cfg("yolov7" if name.startswith("yolov7") else "yolox", name)


@pytest.mark.parametrize("name", ["yolox_l", "yolox_x", "yolov7"])
def test_wide_models_vs_reference_fixture(name):
    """yolox_l.yaml / yolox_x.yaml / yolov7.yaml at their FULL width through the HIP plans on the small batch the REFERENCE ran
    (tests/golden/wide_<name>.npz, tools/gen_golden.py: gen_wide): fp32 parity mode -- raw head maps 5e-4 of their largest value, losses 1e-4, the stored
    gradients 1e-2 of their largest element and 1e-5 in cosine, the L2 norm of EVERY parameter's gradient 2e-3; bf16 MFMA path -- head maps and loss at the level of the reference's own bf16 autocast (0.12 rel-rms,
    3e-2).  These are the 320 / 640 / 1024 / 1280 / 2560-channel kernel instances no YOLOX-s fixture reaches."""
    g = load_golden("wide_" + name)
    cfg = _wide_cfg(name)
    model, sd0 = _build(cfg, "fp32")
    first = next(iter(model.named_parameters()))
    assert np.array_equal(first[1].detach().cpu().numpy(), g["first_weight"]), "seed-96 initialisation differs from the reference's"
    psum = float(sum(p.double().sum() for p in model.parameters()))
    assert abs(psum - float(g["param_sum"])) <= 1e-6 * float(g["param_abs_sum"])
    B, S, ngt, mgt = int(g["batch"]), int(g["size"]), int(g["num_gt"]), int(g["max_gt"])
    gen = torch.Generator().manual_seed(int(g["seed_data"]))
    imgs = torch.rand(B, 3, S, S, generator=gen) * 255
    labels = torch.zeros(B, mgt, 5)
    labels[:, :ngt, 0] = torch.randint(0, NC, (B, ngt), generator=gen).float()
    labels[:, :ngt, 1:3] = (0.15 + 0.7 * torch.rand(B, ngt, 2, generator=gen)) * S
    labels[:, :ngt, 3:5] = 8 + torch.rand(B, ngt, 2, generator=gen) * 0.3 * S
    imgs, labels = imgs.to(hu.DEV), labels.to(hu.DEV)
    names, norms = [str(n) for n in g["grad_names"]], g["grad_norms"]
    gscale = float(np.sqrt(float(g["grad_sq_sum"])))

    def step(m):
        m.load_state_dict(sd0)
        m.train()
        with torch.no_grad():
            maps = [t.float().cpu() for t in m(imgs, None)]
        m.load_state_dict(sd0)
        out = m(imgs, labels)
        out["loss"].sum().backward()
        torch.cuda.synchronize()
        return maps, out, dict(m.named_parameters())

    # ---- fp32 parity mode
    maps, out, params = step(model)
    for i, mp in enumerate(maps):
        ref = torch.from_numpy(g["maps/%d" % i])
        err = float((mp - ref).abs().max() / ref.abs().max())
        print("wide %s fp32 map %d: rel max err %.3g" % (name, i, err))
        assert err <= WIDE_FP32_MAP_TOL
    for k in [k for k in g if k.startswith("out/") and k != "out/proportion"]:
        got, want = float(torch.as_tensor(out[k[4:]]).sum()), float(np.asarray(g[k]).sum())
        print("wide", name, k, got, want)
        assert abs(got - want) <= 1e-4 * max(1.0, abs(want)), k
    if "out/proportion" in g:
        assert abs(float(out["proportion"]) - float(np.asarray(g["out/proportion"]).reshape(-1)[0])) <= 1e-5
    for k in [k for k in g if k.startswith("grad/")]:
        ref = torch.from_numpy(g[k])
        got = params[k[5:]].grad.cpu()
        e = float((got - ref).abs().max()) / max(float(ref.abs().max()), 1e-6)
        c1 = 1.0 - float((got.double() * ref.double()).sum() / max(float(got.double().norm() * ref.double().norm()), 1e-300))
        print("wide %s fp32 gradient %-44s rel max err %.3g, 1 - cosine %.3g" % (name, k[5:], e, c1))
        assert e <= WIDE_FP32_GRAD_TOL and c1 <= WIDE_FP32_GRAD_COS, k
    worst = 0.0
    for k, want in zip(names, norms):
        if want < 0:
            continue
        got = float(params[k].grad.double().norm())
        rel = abs(got - want) / max(want, 1e-4 * gscale)
        worst = max(worst, rel)
        assert rel <= WIDE_FP32_NORM_TOL, (k, got, want)
    print("wide %s fp32: worst per-parameter gradient-norm deviation %.3g over %d parameters" % (name, worst, int((norms >= 0).sum())))
    # ---- bf16 MFMA path (the benchmarked kernels), same state and batch
    m16, _ = _build(cfg, "bf16")
    maps16, o16, p16 = step(m16)
    for i, mp in enumerate(maps16):
        ref = torch.from_numpy(g["maps/%d" % i])
        rms = float((mp - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
        print("wide %s bf16 map %d: rel rms %.3g" % (name, i, rms))
        assert rms <= WIDE_BF16_MAP_RMS["yolov7" if name.startswith("yolov7") else "yolox"]
    l16, lref = float(o16["loss"].sum()), float(np.asarray(g["out/loss"]).sum())
    print("wide %s bf16 loss %.5f vs reference %.5f (rel %.2e)" % (name, l16, lref, abs(l16 - lref) / lref))
    assert abs(l16 - lref) <= WIDE_BF16_LOSS_TOL * lref
    for k in [k for k in g if k.startswith("grad/") and (k.endswith(".bias") and "norm" not in k)]:
        ref, got = torch.from_numpy(g[k]).double(), p16[k[5:]].grad.cpu().double()
        c = float((ref * got).sum()) / max(float(ref.norm() * got.norm()), 1e-30)
        print("wide %s bf16 gradient cosine vs reference %-28s %.5f (printed only: the reference's own bf16 autocast decorrelates here)" % (name, k[5:], c))


# ---------------------------------------------------------------------------------------------- cfg3
@pytest.mark.parametrize("name", ["yolox_m", "yolox_tiny", "yolox_x"])
def test_ragged_channel_blocks_end_to_end(name, monkeypatch):
    """The reference's other widths through the column tiling that fits them (csrc/conv_mfma_rag.hip, conv_pw_rag_kernel): yolox_m (96 / 192 / 384 /
    768 channels: one 96-channel block of three waves, 128 + 64), yolox_tiny (96 / 192 / 384), yolox_x (80 / 160 / 320).  EVAL forward (BatchNorm
    folded into the convolution epilogues, every concat pitch / merged pair / shortcut of the real graph) with PLYOLO_RAG=1 (default) against
    PLYOLO_RAG=0: bit-identical predictions.  In training mode the two differ only in how the fp32 partials of the BatchNorm statistics are grouped
    -- which a randomly initialised 100-layer network amplifies to tens of percent at its outputs, exactly as forcing 64-channel whole blocks does
    (tools/diag_ragged.py, profiles/r05_ab_ragged.txt) -- so the training step is checked for what it must be: finite, every parameter reached."""
    cfg = _cfg("yolox", name)
    imgs, labels = _batch(2, 128, 4321, num_gt=6)
    imgs, labels = imgs.to(hu.DEV), labels.to(hu.DEV)
    res = {}
    for rag in ("0", "1"):
        monkeypatch.setenv("PLYOLO_RAG", rag)
        model, sd0 = _build(cfg, "bf16")
        model.eval()
        with torch.no_grad():
            res[rag] = model(imgs, labels).float().cpu()
        if rag == "1":
            losses, g, trained = _step(model, sd0, imgs, labels)
            assert all(np.isfinite(v) for v in losses.values()) and bool(torch.isfinite(g).all())
            missing = [n for n, p in model.named_parameters() if p.grad is None and not _dead(n)]
            assert not missing, missing[:5]
        del model
    assert bool(torch.isfinite(res["1"]).all())
    assert torch.equal(res["0"], res["1"])


RAG_WARM_LOSS_TOL, RAG_WARM_COS, RAG_WARM_STATS = 1e-3, 0.9995, 5e-3   # measured 2e-5 / 8e-5, 0.99997 / 0.99998, 2e-4 (yolox_tiny / yolox_m)


@pytest.mark.parametrize("name", ["yolox_tiny", "yolox_m"])
def test_ragged_channel_blocks_training_step_on_warm_weights(name, monkeypatch):
    """the TRAINING path of the ragged / 96-channel column blocks (BatchNorm-statistics epilogue of the 32 / 64 / 96-channel
    blocks, merged pairs, folded reductions) at network level, on something stable: the net is first warmed by 30 bf16 SGD steps (the bf16
    mode is deterministic; a warm net no longer amplifies the regrouped fp32 partials of the statistics the way a random initialisation
    does), then ONE step from that state with PLYOLO_RAG=0 (whole 128-channel blocks) and with PLYOLO_RAG=1: loss, every BatchNorm's
    running statistics after the step, and the gradients must agree within the bf16 band.  A wrong statistics epilogue, a mis-indexed
    short block or a stale slab would show as O(1)."""
    from pl_yolo_amd.trainer import Trainer
    cfg = _cfg("yolox", name)
    monkeypatch.setenv("PLYOLO_RAG", "1")
    warm, _ = _build(cfg, "bf16")
    data = [_batch(4, 128, 700 + i, num_gt=5, max_gt=8) for i in range(3)]
    tr = Trainer(warm, learning_rate=0.01, momentum=0.9, warmup=0.1, total_steps=300, ema=False)
    losses = [float(tr.train_step(*data[i % 3])["loss"].detach()) for i in range(30)]
    assert all(np.isfinite(losses)) and sum(losses[-5:]) < sum(losses[:5])
    state = {k: v.detach().clone() for k, v in warm.state_dict().items()}
    del warm, tr
    imgs, labels = _batch(4, 128, 799, num_gt=5, max_gt=8)
    res = {}
    for rag in ("0", "1"):
        monkeypatch.setenv("PLYOLO_RAG", rag)
        m, _ = _build(cfg, "bf16")
        ls, g, trained = _step(m, state, imgs, labels)
        stats = {k: v.detach().double().cpu().clone() for k, v in m.state_dict().items() if k.endswith(("running_mean", "running_var"))}
        res[rag] = (ls, g.double().cpu(), [n for n, _ in trained], stats)
        del m
    (l0, g0, n0, s0), (l1, g1, n1, s1) = res["0"], res["1"]
    assert n0 == n1 and set(s0) == set(s1)
    cs = float((g0 * g1).sum() / (g0.norm() * g1.norm()))
    worst = max(float((s0[k] - s1[k]).abs().max() / (s0[k].abs().max() + 1e-6)) for k in s0)
    print("ragged blocks, warm %s: loss RAG=0 %.5f RAG=1 %.5f | all-parameter gradient cosine %.6f | worst running-statistic deviation %.3g"
          " | warm-up %.3f -> %.3f" % (name, l0["loss"], l1["loss"], cs, worst, sum(losses[:5]) / 5, sum(losses[-5:]) / 5))
    assert abs(l0["loss"] - l1["loss"]) <= RAG_WARM_LOSS_TOL * abs(l0["loss"])
    assert cs >= RAG_WARM_COS
    assert worst <= RAG_WARM_STATS


@pytest.mark.parametrize("repconv", [False, True])
def test_cfg3_yolov7_640_b32(repconv):
    cfg = _cfg("yolov7", "yolov7", repconv)
    model, sd0 = _build(cfg)
    imgs, labels = _batch(32, 640, 1234)
    _check_step_properties(model, sd0, imgs, labels, "yolov7%s" % ("+repconv" if repconv else ""))
    # eval: [B, 3*(80^2+40^2+20^2), 85], finite scores, closed-form decode of the raw maps (yolov7_loss.py:50-78)
    model.load_state_dict(sd0)
    model.eval()
    with torch.no_grad():
        pred = model(imgs, torch.zeros(32, 1, 5, device=hu.DEV))
        maps = model(imgs, None)
    A = 3 * (80 * 80 + 40 * 40 + 20 * 20)
    assert tuple(pred.shape) == (32, A, 5 + NC)
    anchors = torch.tensor(cfg["loss"]["anchors"], dtype=torch.float32, device=hu.DEV).reshape(3, 3, 2)
    rows = []
    for l, (m, s) in enumerate(zip(maps, (8, 16, 32))):
        b, c, h, w = m.shape
        t = torch.sigmoid(m.float().reshape(b, 3, 5 + NC, h, w).permute(0, 1, 3, 4, 2))
        ys, xs = torch.meshgrid(torch.arange(h, device=hu.DEV), torch.arange(w, device=hu.DEV), indexing="ij")
        grid = torch.stack([xs, ys], -1).reshape(1, 1, h, w, 2).float()
        xy = (t[..., 0:2] * 2.0 - 0.5 + grid) * s
        wh = (t[..., 2:4] * 2.0) ** 2 * anchors[l].reshape(1, 3, 1, 1, 2)
        rows.append(torch.cat([xy - wh / 2, xy + wh / 2, t[..., 4:]], -1).reshape(b, 3 * h * w, 5 + NC))
    want = torch.cat(rows, 1)
    err = float((pred - want).abs().max() / want.abs().max())
    print("yolov7 eval decode vs closed form: rel max err %.3g" % err)
    assert err <= 1e-5


def test_cfg3_yolov7_640_b32_permutation_fp32():
    cfg = _cfg("yolov7", "yolov7")
    model, sd0 = _build(cfg)
    imgs, labels = _batch(32, 640, 4321)
    _check_permutation_fp32(model, sd0, imgs, labels, "yolov7")


# ---------------------------------------------------------------------------------------------- cfg4
def test_cfg4_yolox_l_640_b16():
    cfg = _cfg("yolox", "yolox_l")
    model, sd0 = _build(cfg)
    imgs, labels = _batch(16, 640, 1234)
    _check_step_properties(model, sd0, imgs, labels, "yolox_l")
    _check_eval_decode_yolox(model, sd0, imgs, 8400)
    _check_permutation_fp32(model, sd0, imgs, labels, "yolox_l")


# ---------------------------------------------------------------------------------------------- cfg5
def test_cfg5_yolox_x_1280_b16_train_step():
    cfg = _cfg("yolox", "yolox_x")
    model, sd0 = _build(cfg)
    imgs, labels = _batch(16, 1280, 1234)
    _check_step_properties(model, sd0, imgs, labels, "yolox_x@1280")
    _check_eval_decode_yolox(model, sd0, imgs, 33600)


def _set_i_head_maps(B, gen):
    """SURVEY 8d cfg5 set (i): raw head maps [B,85,160,160] / [B,85,80,80] / [B,85,40,40] with background objectness
    logit -12 and, per image, 200 random sites x 5 adjacent anchors = 1000 anchors carrying obj ~U(0,4), one class
    logit ~U(0,4) (others -8), box logits ~N(0, 0.5)."""
    sizes = [(160, 160), (80, 80), (40, 40)]
    A = sum(h * w for h, w in sizes)
    flat = torch.zeros(B, A, 5 + NC)
    flat[..., 0:4] = torch.randn(B, A, 4, generator=gen) * 0.5
    flat[..., 4] = -12.0
    flat[..., 5:] = -8.0
    for b in range(B):
        sites = torch.randint(0, A - 5, (200,), generator=gen)
        idx = (sites[:, None] + torch.arange(5)[None, :]).reshape(-1)
        flat[b, idx, 4] = torch.rand(1000, generator=gen) * 4
        cls = torch.randint(0, NC, (1000,), generator=gen)
        flat[b, idx, 5 + cls] = torch.rand(1000, generator=gen) * 4
    maps, a0 = [], 0
    for (h, w) in sizes:
        maps.append(flat[:, a0:a0 + h * w].reshape(B, h, w, 5 + NC).permute(0, 3, 1, 2).contiguous())
        a0 += h * w
    return maps, sizes


def test_cfg5_eval_decode_and_postprocess_set_i():
    """The eval leg of cfg5 on the synthetic set-(i) head maps: device eval decode == closed form, then `postprocess`
    (conf 0.01, NMS 0.65, pl_detection.py:24-25) == the CPU oracle's post-processing, image by image, box by box."""
    import ctypes as C
    from pl_yolo_amd._lib import call
    from pl_yolo_amd.postprocess import postprocess
    from oracle import nms as onms
    B = 16
    maps, sizes = _set_i_head_maps(B, torch.Generator().manual_seed(2025))
    d, rows = hu.yolox_desc(B, NC, 1, sizes, (8, 16, 32))
    raw = hu.maps_to_raw([m.to(hu.DEV) for m in maps])
    ev = torch.zeros(B * d.A * (5 + NC), device=hu.DEV)
    call("plyolo_yolox_eval_decode", C.byref(d), raw.data_ptr(), ev.data_ptr(), hu.stream())
    pred = ev.view(B, d.A, 5 + NC)
    want = _decode_yolox([m.to(hu.DEV) for m in maps], (8, 16, 32))
    err = float((pred - want).abs().max() / want.abs().max())
    assert err <= 1e-5, err
    dets = postprocess(pred, conf_thre=0.01, nms_thre=0.65)
    ref = onms.postprocess(pred.cpu().numpy(), 0.01, 0.65)
    n_in = int((pred[..., 4] * pred[..., 5:].max(-1).values >= 0.01).sum())
    kept = 0
    for b in range(B):
        assert (dets[b] is None) == (ref[b] is None)
        if ref[b] is None:
            continue
        got = dets[b].cpu().numpy()
        assert got.shape == ref[b].shape, (b, got.shape, ref[b].shape)
        np.testing.assert_array_equal(got[:, 5], ref[b][:, 5])
        np.testing.assert_allclose(got[:, :5], ref[b][:, :5], rtol=1e-6, atol=1e-5)
        kept += got.shape[0]
    print("set (i): %d candidates into NMS (%.0f / image), %d kept" % (n_in, n_in / B, kept))
    assert 900 * B <= n_in <= 1000 * B and 0 < kept <= 300 * B
