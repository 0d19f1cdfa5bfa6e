"""-m gpu: the whole detector through the plugin API (build_model / OneStageD) on the
MI355X against the reference-generated golden fixture and the CPU oracle.

fp32 (parity mode): losses within 1e-4, gradients within 2e-4 of the largest entry.
bf16 (MFMA mode): loss within 3e-2 relative, gradient cosine >= 0.98 per tensor group
(bf16 activations move head logits by ~1e-2, SURVEY.md section 7 hard part 2)."""
import copy
import os

import numpy as np
import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu

import pl_yolo_amd  # noqa: E402
from conftest import load_golden, warm_s_state, ROOT  # noqa: E402
from oracle import net as onet, detector as odet  # noqa: E402
import hiputil as hu  # noqa: E402


def _cfg(name):
    with open(os.path.join(ROOT, "configs", "model", "yolox", name + ".yaml")) as f:
        return yaml.safe_load(f)


def _golden_model(dtype):
    g = load_golden("network_yolox_test")
    model = pl_yolo_amd.build_model(_cfg("yolox_test"), int(g["num_classes"]))
    sd = {k[6:]: torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith("state/")}
    assert set(sd) == set(model.state_dict().keys())
    model.load_state_dict(sd)
    model.compute_dtype = dtype
    return g, model.to(hu.DEV)


def test_fp32_train_step_vs_golden():
    g, model = _golden_model("fp32")
    model.train()
    x = torch.from_numpy(g["x"]).to(hu.DEV)
    labels = torch.from_numpy(g["labels"]).to(hu.DEV)
    out = model(x, labels)
    for k in ("loss", "loss_iou", "loss_obj", "loss_cls"):
        print(k, float(out[k]), float(g["out/" + k]))
        assert abs(float(out[k]) - float(g["out/" + k])) <= 1e-4 * max(1.0, abs(float(g["out/" + k]))), k
    assert out["loss_l1"] == 0.0
    assert abs(float(out["proportion"]) - float(g["out/proportion"])) < 1e-5
    out["loss"].backward()
    torch.cuda.synchronize()
    nograd = set(str(n) for n in g["nograd_names"])
    worst = 0.0
    for name, p in model.named_parameters():
        if name in nograd:
            assert p.grad is None, name
            continue
        ref = g["grad/" + name]
        assert p.grad is not None, name
        err = float(np.abs(p.grad.cpu().numpy() - ref).max()) / max(1e-3, float(np.abs(ref).max()))
        worst = max(worst, err)
        assert err <= 2e-4, (name, err)
    print("worst relative gradient error %.3g" % worst)
    sd = model.state_dict()
    for k, v in g.items():
        if k.startswith("state_after/"):
            np.testing.assert_allclose(sd[k[12:]].cpu().numpy(), v, rtol=1e-4, atol=1e-5, err_msg=k)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_train_step_and_eval_on_a_rectangular_image_vs_golden(dtype):
    """A 64 x 96 batch through the whole detector (head maps 8x12 / 4x6 / 2x3; ragged tiles in every convolution, the reference's non-square
    grid quirk in decode / SimOTA / eval decode, yolox_loss.py:198-200) against the reference itself (tests/golden/network_yolox_rect.npz,
    the weights of network_yolox_test.npz): fp32 parity mode at the fixture bars (losses 1e-4, gradients 2e-4, running statistics, eval
    output); bf16: loss within 3e-2 (random-initialised toy net), finite gradients."""
    g = load_golden("network_yolox_rect")
    _, model = _golden_model(dtype)
    model.train()
    x, labels = torch.from_numpy(g["x"]).to(hu.DEV), torch.from_numpy(g["labels"]).to(hu.DEV)
    maps = model(x)
    assert [tuple(m.shape[2:]) for m in maps] == [(8, 12), (4, 6), (2, 3)]
    if dtype == "fp32":
        for i, m in enumerate(maps):
            np.testing.assert_allclose(m.detach().cpu().numpy(), g["maps_train%d" % i], rtol=1e-3, atol=2e-4)
    _, model = _golden_model(dtype)
    model.train()
    out = model(x, labels)
    out["loss"].backward()
    torch.cuda.synchronize()
    if dtype == "bf16":
        rel = abs(float(out["loss"]) - float(g["out/loss"])) / float(g["out/loss"])
        print("rectangular image, bf16 loss %.5f vs %.5f (rel %.3g)" % (float(out["loss"]), float(g["out/loss"]), rel))
        assert rel <= 3e-2 and all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
        return
    for k in ("loss", "loss_iou", "loss_obj", "loss_cls"):
        assert abs(float(out[k]) - float(g["out/" + k])) <= 1e-4 * max(1.0, abs(float(g["out/" + k]))), k
    assert abs(float(out["proportion"]) - float(g["out/proportion"])) < 1e-5
    worst = 0.0
    for name, p in model.named_parameters():
        if "grad/" + name not in g:
            assert p.grad is None, name
            continue
        ref = g["grad/" + name]
        err = float(np.abs(p.grad.cpu().numpy() - ref).max()) / max(1e-3, float(np.abs(ref).max()))
        worst = max(worst, err)
        assert err <= 2e-4, (name, err)
    print("rectangular image, fp32: worst relative gradient error %.3g" % worst)
    sd = model.state_dict()
    for k, v in g.items():
        if k.startswith("state_after/"):
            np.testing.assert_allclose(sd[k[12:]].cpu().numpy(), v, rtol=1e-4, atol=1e-5, err_msg=k)
    model.eval()
    with torch.no_grad():
        ev = model(x, labels)
    np.testing.assert_allclose(ev.cpu().numpy(), g["eval_out"], rtol=1e-3, atol=2e-3)


def test_fp32_maps_and_eval_vs_golden():
    g, model = _golden_model("fp32")
    x = torch.from_numpy(g["x"]).to(hu.DEV)
    model.train()
    maps = model(x)  # labels=None -> raw NCHW head maps, BN in batch-stat mode
    for i, m in enumerate(maps):
        np.testing.assert_allclose(m.detach().cpu().numpy(), g["maps_train%d" % i], rtol=1e-3, atol=2e-4)
    # eval: running stats as they were after the reference's single training step
    sd = model.state_dict()
    for k, v in g.items():
        if k.startswith("state_after/"):
            sd[k[12:]].copy_(torch.from_numpy(v.copy()))
    model.eval()
    with torch.no_grad():
        out = model(x, torch.from_numpy(g["labels"]).to(hu.DEV))
        maps = model(x)
    for i, m in enumerate(maps):
        np.testing.assert_allclose(m.cpu().numpy(), g["maps_eval%d" % i], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(out.cpu().numpy(), g["eval_out"], rtol=1e-3, atol=2e-3)


def _maps_grads(model, x, rs):
    model.zero_grad(set_to_none=True)
    maps = model(x)
    sum((m * r).sum() for m, r in zip(maps, rs)).backward()
    torch.cuda.synchronize()
    return [m.detach() for m in maps], {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}


def test_bf16_train_step_vs_golden():
    """bf16 MFMA mode.  SimOTA is a discrete assignment: bf16 activations perturb the head
    logits by ~1e-2, which legitimately flips a few anchors of a random-init net, so the
    full-step gradients are only sanity-checked here; the precision of the bf16 backward
    is measured with the assignment taken out (labels=None path, fixed upstream grads)."""
    g, model = _golden_model("bf16")
    model.train()
    x = torch.from_numpy(g["x"]).to(hu.DEV)
    labels = torch.from_numpy(g["labels"]).to(hu.DEV)
    out = model(x, labels)
    rel = abs(float(out["loss"]) - float(g["out/loss"])) / float(g["out/loss"])
    print("bf16 loss %.5f vs %.5f (rel %.3g)" % (float(out["loss"]), float(g["out/loss"]), rel))
    assert rel <= 3e-2
    out["loss"].backward()
    torch.cuda.synchronize()
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
    # head maps + parameter gradients for identical upstream gradients: bf16 vs fp32 HIP
    _, m32 = _golden_model("fp32")
    m32.train()
    gen = torch.Generator().manual_seed(3)
    # 160x160 input: the coarsest level is 5x5, i.e. 50 samples per BatchNorm channel (at 64x64 it
    # would be 8 samples, where a single bf16 rounding flip moves the statistics by percents)
    x = (torch.rand(2, 3, 160, 160, generator=gen) * 255).to(hu.DEV)
    rs = [torch.randn(m.shape, generator=gen).to(hu.DEV) for m in [torch.empty(2, 8, 20, 20), torch.empty(2, 8, 10, 10), torch.empty(2, 8, 5, 5)]]
    maps16, g16 = _maps_grads(model, x, rs)
    maps32, g32 = _maps_grads(m32, x, rs)
    st = {k[6:]: torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith("state/")}
    with onet.emulate_bf16(), torch.no_grad():
        emu = odet.forward(st, _cfg("yolox_test"), int(g["num_classes"]), x.cpu(), None, training=True)
    for a, b, c in zip(maps16, maps32, emu):
        e, r, re_ = hu.relerr(a, b), hu.relrms(a, b), hu.relrms(a.cpu(), c)
        print("bf16 head map: vs fp32 HIP max %.3g rms %.3g | vs bf16-emulating oracle rms %.3g" % (e, r, re_))
        assert r <= 8e-2 and re_ <= 8e-2  # level 2 is a 2x2 map: BatchNorm over 8 samples amplifies rounding flips
    worst = 1.0
    for n in g32:
        c = hu.cossim(g16[n], g32[n])
        worst = min(worst, c)
        if c < 0.99:
            print("low cosine", n, c)
    allc = hu.cossim(torch.cat([g16[n].flatten() for n in g32]), torch.cat([g32[n].flatten() for n in g32]))
    print("bf16 vs fp32 gradient cosine: all %.5f worst tensor %.5f" % (allc, worst))
    # Printed, not asserted: a random-initialised 8-channel BatchNorm net amplifies a bf16 rounding ~1.1x per layer (DESIGN.md,
    # "bf16 mode"), so this cosine says nothing about the kernels (0.8 / 0.4 used to be "asserted" here).  The bf16 backward is
    # held to >= 0.99 on trained weights instead: test_warm_weights_* (toy net) and test_warm_yolox_s_* (the benchmarked net).


def test_hipgraph_replay_matches_eager():
    g, model = _golden_model("bf16")
    model.train()
    x = torch.from_numpy(g["x"]).to(hu.DEV)
    labels = torch.from_numpy(g["labels"]).to(hu.DEV)
    sd0 = copy.deepcopy(model.state_dict())
    res = []
    for use_graph in (False, True, True, False, True):   # eager multi-stream replay and hipGraph replays, interleaved
        model.load_state_dict(sd0)
        model.runner().use_graph = use_graph
        model.zero_grad(set_to_none=True)
        out = model(x, labels)
        out["loss"].backward()
        torch.cuda.synchronize()
        res.append((float(out["loss"]), torch.cat([p.grad.flatten() for p in model.parameters() if p.grad is not None]).clone()))
    for r in res[1:]:
        assert abs(r[0] - res[0][0]) <= 1e-5 * abs(res[0][0])
        # same launches, same order per lane: equal up to the fp32 atomics of the six bias-gradient reductions
        assert float((r[1] - res[0][1]).abs().max()) <= 1e-5 * float(res[0][1].abs().max())


def test_module_contract():
    g, model = _golden_model("bf16")
    # deep copy (ModelEMA does this), optimizer over .parameters(), repeated steps
    ema = copy.deepcopy(model).eval()
    assert set(ema.state_dict()) == set(model.state_dict())
    opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9)
    x = torch.from_numpy(g["x"]).to(hu.DEV)
    labels = torch.from_numpy(g["labels"]).to(hu.DEV)
    model.train()
    losses = []
    for _ in range(3):
        out = model(x, labels)
        opt.zero_grad()
        out["loss"].backward()
        opt.step()
        losses.append(float(out["loss"]))
    print("losses over 3 SGD steps", losses)
    assert all(np.isfinite(losses))
    assert int(model.state_dict()["backbone.stem.conv.norm.num_batches_tracked"]) == 3
    with torch.no_grad():
        d = ema(x, labels)
    assert tuple(d.shape) == (2, 84, 8)
    with pytest.raises(pl_yolo_amd.PlyoloError):
        model(torch.zeros(1, 3, 64, 64), None)  # CPU tensor: no fallback


def test_yolox_s_bf16_vs_oracle():
    """YOLOX-s at 320x320, B=2: HIP bf16 vs the fp32 CPU oracle on identical weights:
    loss of the full step, and (labels=None path) head maps + parameter gradients for
    identical upstream gradients."""
    cfg = _cfg("yolox_s")
    torch.manual_seed(96)
    model = pl_yolo_amd.build_model(cfg, 80)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    imgs, labels = odet.synthetic_batch(2, 320, 80, num_gt=12, max_gt=20, seed=1234)
    st1 = {k: v.clone() for k, v in state.items()}
    out_ref, _ = odet.train_step_grads(st1, cfg, 80, imgs, labels)
    names = onet.param_names(state)
    for k in names:
        state[k].requires_grad_(True)
    maps_ref = odet.forward(state, cfg, 80, imgs, None, training=True)
    gen = torch.Generator().manual_seed(4)
    rs = [torch.randn(m.shape, generator=gen) for m in maps_ref]
    sum((m * r).sum() for m, r in zip(maps_ref, rs)).backward()
    model = model.to(hu.DEV).train()
    maps, grads = _maps_grads(model, imgs.to(hu.DEV), [r.to(hu.DEV) for r in rs])
    st2 = {k: v.detach().clone() for k, v in state.items()}
    with onet.emulate_bf16(), torch.no_grad():
        emu = odet.forward(st2, cfg, 80, imgs, None, training=True)
    for a, b, c in zip(maps, maps_ref, emu):
        e, r, re_ = hu.relerr(a.cpu(), b.detach()), hu.relrms(a.cpu(), b.detach()), hu.relrms(a.cpu(), c)
        print("yolox_s head map: vs fp32 oracle max %.3g rms %.3g | vs bf16-emulating oracle rms %.3g" % (e, r, re_))
        assert r <= 4e-2 and re_ <= 1.5e-2
    a = torch.cat([grads[n].flatten().cpu() for n in grads])
    b = torch.cat([state[n].grad.flatten() for n in grads])
    cs = hu.cossim(a, b)
    worst = min(hu.cossim(grads[n].cpu(), state[n].grad) for n in grads)
    print("yolox_s gradient cosine vs oracle: all %.5f worst tensor %.5f" % (cs, worst))
    # printed only (random-init amplification, see above); the asserted bounds live in test_warm_yolox_s_bf16_end_to_end
    out = model(imgs.to(hu.DEV), labels.to(hu.DEV))
    rel = abs(float(out["loss"]) - float(out_ref["loss"])) / float(out_ref["loss"])
    print("yolox_s loss hip %.5f oracle %.5f rel %.3g" % (float(out["loss"]), float(out_ref["loss"]), rel))
    assert rel <= 3e-2


def test_trainer_vs_reference_trajectory():
    """3 x (fwd, bwd, SGD-momentum, EMA, LR step): pl_yolo_amd.trainer.Trainer in fp32 mode vs the
    trajectory recorded from the reference's optimizer / scheduler / ModelEMA (a25)."""
    from pl_yolo_amd.trainer import Trainer
    g, model = _golden_model("fp32")
    h = load_golden("harness_trajectory")
    x = torch.from_numpy(h["x"]).to(hu.DEV)
    labels = torch.from_numpy(h["labels"]).to(hu.DEV)
    tr = Trainer(model, learning_rate=0.01, momentum=0.9, warmup=0.1, total_steps=20, ema=True)
    for step in range(3):
        assert abs(tr.current_lr() - float(h["lrs"][step])) < 1e-12
        out = tr.train_step(x, labels)
        assert abs(float(out["loss"]) - float(h["loss%d" % step])) <= 3e-4 * float(h["loss%d" % step]), step
    torch.cuda.synchronize()
    sd, esd = model.state_dict(), tr.ema_model.state_dict()
    for k in sd:
        ref, refe = h["final/" + k], h["ema/" + k]
        if ref.dtype.kind != "f":
            assert int(sd[k]) == int(ref), k
            continue
        init = g["state/" + k]
        # error measured against the size of the 3-step update itself
        upd = max(float(np.abs(ref - init).max()), 1e-4 * max(1.0, float(np.abs(ref).max())))
        e = float(np.abs(sd[k].cpu().numpy() - ref).max()) / upd
        updE = max(float(np.abs(refe - init).max()), 1e-4 * max(1.0, float(np.abs(refe).max())))
        ee = float(np.abs(esd[k].cpu().numpy() - refe).max()) / updE
        assert e <= 2e-2 and ee <= 2e-2, (k, e, ee)


def test_postprocess_api_vs_oracle():
    from pl_yolo_amd.postprocess import postprocess
    from oracle import nms as onms
    g, model = _golden_model("fp32")
    model.eval()
    gen = torch.Generator().manual_seed(9)
    x = (torch.rand(3, 3, 128, 128, generator=gen) * 255).to(hu.DEV)
    with torch.no_grad():
        pred = model(x, torch.zeros(3, 1, 5, device=hu.DEV))
    assert tuple(pred.shape) == (3, 16 * 16 + 8 * 8 + 4 * 4, 8)
    got = postprocess(pred, conf_thre=0.001, nms_thre=0.65)
    want = onms.postprocess(pred.cpu().numpy(), 0.001, 0.65)
    assert len(got) == 3
    for a, b in zip(got, want):
        if b is None:
            assert a is None
        else:
            np.testing.assert_array_equal(a.cpu().numpy(), b)
    assert postprocess(pred, conf_thre=2.0) == [None, None, None]


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_overfit_one_batch(dtype):
    """End-to-end sanity of forward + loss + backward + optimizer through the Trainer: 60 SGD steps on one fixed
    batch drive the loss down (any sign or scaling error in a gradient kernel shows up here as divergence)."""
    from pl_yolo_amd.trainer import Trainer
    g, model = _golden_model(dtype)
    x = torch.from_numpy(g["x"]).to(hu.DEV)
    labels = torch.from_numpy(g["labels"]).to(hu.DEV)
    tr = Trainer(model, learning_rate=0.02, momentum=0.9, warmup=0.1, total_steps=400, ema=True)
    losses = []
    for _ in range(60):
        out = tr.train_step(x, labels)
        losses.append(float(out["loss"].detach()))
    head, tail = sum(losses[:5]) / 5, sum(losses[-5:]) / 5
    print("overfit %s: loss %.3f -> %.3f (min %.3f)" % (dtype, head, tail, min(losses)))
    assert all(np.isfinite(losses))
    assert tail < 0.8 * head
    # the EMA copy runs as an eval model (its BatchNorm running statistics are 60 steps old at momentum 0.03, so
    # the exp() of the box decode may overflow exactly as it would in the reference; scores must stay numbers)
    ema = tr.eval_model().eval()
    with torch.no_grad():
        pred = ema(x, torch.zeros(x.shape[0], 1, 5, device=hu.DEV))
    assert not bool(torch.isnan(pred[..., 4:]).any())


def test_bench_rccl_path_single_rank():
    """bench.py under torchrun with one rank and PLYOLO_BENCH_FORCE_DDP=1: RCCL process-group init, weight
    broadcast, the gradient all-reduce after every backward and the barriers of the timing protocol all execute
    on this GPU (the multi-GPU runs are the driver's; this keeps their code path from rotting)."""
    import json, socket, subprocess, sys
    env = dict(os.environ, PLYOLO_BENCH_FORCE_DDP="1", MASTER_ADDR="127.0.0.1")
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))        # a port that is free now (a fixed one can still sit in another test's or another user's hands)
    port = sk.getsockname()[1]
    sk.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--batch", "4", "--size", "320", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["value"] > 0 and np.isfinite(out["config"]["loss"])
    # the multi-GPU report fields: per-rank step time and the exposed part of the gradient exchange
    pr = out["per_rank"]
    assert len(pr["ms_per_step"]) == 1 and pr["ms_per_step"][0] > 0 and 0 <= pr["exposed_comm_ms_per_step"][0] < pr["ms_per_step"][0]


def test_bench_two_gpus_rccl():
    """`python bench.py --gpus 2`: two real ranks, RCCL ReduceOp.AVG over xGMI, the bucket hooks of the backward plan under real
    traffic.  SKIPPED on a one-GPU box (every gpurun box of this pool); the driver's 8-GPU node runs it on first contact."""
    import json, subprocess, sys
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (this box exposes %d)" % torch.cuda.device_count())
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["rccl_world_size"] == 2 and out["value"] > 0 and np.isfinite(out["config"]["loss"])
    pr = out["per_rank"]
    assert len(pr["ms_per_step"]) == 2 and all(v > 0 for v in pr["ms_per_step"])
    assert all(0 <= e < t for e, t in zip(pr["exposed_comm_ms_per_step"], pr["ms_per_step"]))


def test_bench_two_ranks_share_one_gpu_dry_run():
    """`PLYOLO_BENCH_SHARE_GPU=1 python bench.py --gpus 2` on a ONE-GPU box: two real processes share the device and gloo carries the
    exchange -- the whole N-rank path of the bench (self-launch, weight broadcast, bucket hooks in the backward plan, barrier +
    max-over-ranks timing, per_rank.exposed_comm_ms_per_step gathered over the ranks) runs end to end across process boundaries.  A
    plumbing check: the line is marked `dry_run_shared_gpu` and its value is no scaling figure."""
    import json, subprocess, sys
    if torch.cuda.device_count() >= 2:
        pytest.skip("a multi-GPU node runs the real thing (test_bench_two_gpus_rccl)")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--no-cpu-baseline",
           "--model", "yolox_test", "--size", "64", "--batch", "2"]
    env = dict(os.environ, PLYOLO_BENCH_SHARE_GPU="1")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["dry_run_shared_gpu"] is True and out["value"] > 0 and np.isfinite(out["config"]["loss"])
    assert out["config"]["global_batch"] == 4 and "gloo" in out["config"]["parallelism"]
    pr = out["per_rank"]
    assert len(pr["ms_per_step"]) == 2 and all(v > 0 for v in pr["ms_per_step"])
    assert len(pr["exposed_comm_ms_per_step"]) == 2 and all(0 <= e < t for e, t in zip(pr["exposed_comm_ms_per_step"], pr["ms_per_step"]))
    # without the switch the same command refuses instead of benchmarking one GPU twice
    r2 = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, PLYOLO_BENCH_SHARE_GPU="0"), capture_output=True, text=True, timeout=300)
    assert r2.returncode != 0 and "exposes" in r2.stderr


# ----------------------------------------------------------------------------------------------------------------
# bf16 GRADIENT parity: the bf16-emulating oracle run through its backward, and the warm-weights fixture
# ----------------------------------------------------------------------------------------------------------------
def _emu_grads(cfg, nc, state, x, rs):
    """Head maps + parameter gradients of the CPU oracle with the HIP bf16 pipeline's storage roundings emulated in
    BOTH directions (oracle/net.py: emulate_bf16) for the upstream gradients `rs` (labels=None path)."""
    st = {k: v.detach().clone() for k, v in state.items()}
    names = onet.param_names(st)
    for k in names:
        st[k].requires_grad_(True)
    with onet.emulate_bf16():
        maps = odet.forward(st, cfg, nc, x, None, training=True)
        sum((m * r).sum() for m, r in zip(maps, rs)).backward()
    return [m.detach() for m in maps], {k: st[k].grad for k in names if st[k].grad is not None}


def _grad_report(tag, got, want):
    """Per-tensor relative rms error and cosine of two gradient dicts (tensors whose reference gradient is numerically
    zero are compared on an absolute scale)."""
    gmax = max(float(v.abs().max()) for v in want.values())
    worst_rms, worst_cos, worst_name = 0.0, 1.0, None
    for n, ref in want.items():
        a, b = got[n].detach().float().cpu(), ref.float()
        if float(b.abs().max()) <= 1e-6 * gmax:
            assert float(a.abs().max()) <= 1e-4 * gmax, n
            continue
        r, c = hu.relrms(a, b), hu.cossim(a, b)
        if r > worst_rms:
            worst_rms, worst_name = r, n
        worst_cos = min(worst_cos, c)
    a = torch.cat([got[n].detach().float().cpu().flatten() for n in want])
    b = torch.cat([want[n].float().flatten() for n in want])
    allc, allr = hu.cossim(a, b), hu.relrms(a, b)
    print("%s: all-parameter gradient cosine %.5f rel-rms %.4f | worst tensor rel-rms %.4f (%s) cosine %.5f"
          % (tag, allc, allr, worst_rms, worst_name, worst_cos))
    return allc, allr, worst_rms, worst_cos


def test_bf16_gradients_vs_bf16_emulating_oracle_yolox_s():
    """YOLOX-s at 320x320, batch 2, labels=None path with fixed upstream gradients: the HIP bf16 head maps AND every
    parameter gradient against the CPU oracle that rounds to bf16 at the same storage points, forward and backward.
    What is left between the two is the accumulation order of the fp32 sums (an MFMA tile vs ATen's blocking), i.e. a
    few bf16 roundings that flip -- amplified by the random-initialised BatchNorm stack, which is why the fp32 oracle
    (second column) is much further away than the emulating one."""
    cfg = _cfg("yolox_s")
    torch.manual_seed(96)
    model = pl_yolo_amd.build_model(cfg, 80)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    imgs, _ = odet.synthetic_batch(2, 320, 80, num_gt=12, max_gt=20, seed=1234)
    gen = torch.Generator().manual_seed(4)
    rs = [torch.randn(2, 85, 320 // s, 320 // s, generator=gen) for s in (8, 16, 32)]
    emu_maps, emu_grads = _emu_grads(cfg, 80, state, imgs, rs)
    model = model.to(hu.DEV).train()
    maps, grads = _maps_grads(model, imgs.to(hu.DEV), [r.to(hu.DEV) for r in rs])
    for a, c in zip(maps, emu_maps):
        assert hu.relrms(a.cpu(), c) <= 1.5e-2
    allc, allr, worst_rms, worst_cos = _grad_report("yolox_s bf16 vs bf16-emulating oracle", grads, emu_grads)
    # measured 0.9625 / 0.942 (the fp32 oracle: 0.904 / 0.837): what remains is the random-initialised net's amplification
    # of the few roundings that flip with the summation order -- printed only; the same comparison on the reference's warm
    # weights is asserted at >= 0.99 in test_warm_yolox_s_bf16_gradients_vs_emulating_oracle


def _warm_model(dtype):
    g = load_golden("network_yolox_warm")
    model = pl_yolo_amd.build_model(_cfg("yolox_test"), int(g["num_classes"]))
    sd = {k[6:]: torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith("state/")}
    model.load_state_dict(sd)
    model.compute_dtype = dtype
    return g, model.to(hu.DEV).train()


def test_warm_weights_fp32_vs_reference():
    """The toy YOLOX after 50 SGD steps of the reference (tools/gen_golden.py: gen_network_warm): parity mode,
    losses within 1e-4 and every gradient within 2e-4 of the reference's."""
    g, model = _warm_model("fp32")
    out = model(torch.from_numpy(g["x"]).to(hu.DEV), torch.from_numpy(g["labels"]).to(hu.DEV))
    out["loss"].backward()
    torch.cuda.synchronize()
    for k in ("loss", "loss_iou", "loss_obj", "loss_cls"):
        got, want = float(out[k]), float(g["out/" + k])
        assert abs(got - want) <= 1e-4 * max(1.0, abs(want)), (k, got, want)
    params = dict(model.named_parameters())
    gmax = max(float(np.abs(v).max()) for k, v in g.items() if k.startswith("grad/"))
    for k, v in g.items():
        if k.startswith("grad/"):
            err = float((params[k[5:]].grad.cpu() - torch.from_numpy(v)).abs().max())
            assert err <= 2e-4 * max(float(np.abs(v).max()), 1e-3 * gmax), (k, err)


def test_use_l1_flipped_on_a_live_model_vs_oracle():
    """YOLOXLoss.use_l1 (yolox_loss.py:14,128-135,157-160) switched on after the model has already stepped -- what YOLOX does for
    its last epochs: the runner traces a new session (the flag is baked into the recorded loss launches), the dict carries
    `loss_l1` as a tensor, losses and every parameter gradient match the oracle detector (pinned against the reference by
    loss_case_F / G) on the warm weights."""
    from oracle import detector as od
    g, model = _warm_model("fp32")
    x, labels = torch.from_numpy(g["x"]), torch.from_numpy(g["labels"])
    out0 = model(x.to(hu.DEV), labels.to(hu.DEV))
    assert out0["loss_l1"] == 0.0
    out0["loss"].backward()
    model.zero_grad()
    model.loss.use_l1 = True
    out = model(x.to(hu.DEV), labels.to(hu.DEV))
    out["loss"].backward()
    torch.cuda.synchronize()
    cfg, nc = _cfg("yolox_test"), int(g["num_classes"])
    state = {k[6:]: torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith("state/")}
    ref, rgrads = od.train_step_grads(state, cfg, nc, x, labels, use_l1=True)
    assert float(ref["loss_l1"].detach()) > 0.05
    for k in ("loss", "loss_iou", "loss_obj", "loss_cls", "loss_l1"):
        got, want = float(out[k].detach()), float(ref[k].detach())
        print("use_l1", k, got, want)
        assert abs(got - want) <= 1e-4 * max(1.0, abs(want)), (k, got, want)
    assert abs(float(out["loss"]) - float(out0["loss"]) - float(out["loss_l1"])) <= 1e-4 * float(out["loss"])
    params = dict(model.named_parameters())
    gmax = max(float(v.abs().max()) for v in rgrads.values())
    for k, v in rgrads.items():
        err = float((params[k].grad.cpu() - v).abs().max())
        assert err <= 2e-4 * max(float(v.abs().max()), 1e-3 * gmax), (k, err)


def test_warm_weights_bf16_end_to_end():
    """bf16 MFMA mode on the warm weights, the WHOLE training step (SimOTA included) against the reference's fp32
    numbers: a net that has trained for a while is not the perturbation amplifier a random-initialised one is, so
    the end-to-end bounds are tight here (the random-init fixtures only admit cosine 0.8)."""
    g, model = _warm_model("bf16")
    x, labels = torch.from_numpy(g["x"]).to(hu.DEV), torch.from_numpy(g["labels"]).to(hu.DEV)
    with torch.no_grad():
        maps = model(x, None)
    for i, m in enumerate(maps):
        r = hu.relrms(m.float().cpu(), torch.from_numpy(g["maps_train%d" % i]))
        print("warm bf16 head map %d rel-rms vs reference %.4f" % (i, r))
        assert r <= 3e-2
    g2, model = _warm_model("bf16")
    out = model(x, labels)
    out["loss"].backward()
    torch.cuda.synchronize()
    for k in ("loss", "loss_iou", "loss_obj", "loss_cls"):
        got, want = float(out[k]), float(g["out/" + k])
        print("warm bf16", k, got, want)
        assert abs(got - want) <= 2e-3 * max(1.0, abs(want)), (k, got, want)
    want = {k[5:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("grad/")}
    got = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    allc, allr, worst_rms, worst_cos = _grad_report("warm weights bf16 vs reference fp32", got, want)
    assert allc >= 0.999 and worst_cos >= 0.99 and allr <= 3e-2     # measured 0.99989 / 0.9973 / 0.015


def test_warm_weights_bf16_gradients_vs_emulating_oracle():
    """labels=None path on the warm weights, fixed upstream gradients: every HIP bf16 parameter gradient against the
    bf16-emulating oracle run through its backward -- per-tensor relative rms at the 1e-2 level."""
    g, model = _warm_model("bf16")
    cfg, nc = _cfg("yolox_test"), int(g["num_classes"])
    state = {k[6:]: torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith("state/")}
    x = torch.from_numpy(g["x"])
    gen = torch.Generator().manual_seed(11)
    rs = [torch.randn(m.shape, generator=gen) for m in (g["maps_train0"], g["maps_train1"], g["maps_train2"])]
    emu_maps, emu_grads = _emu_grads(cfg, nc, state, x, rs)
    maps, grads = _maps_grads(model, x.to(hu.DEV), [r.to(hu.DEV) for r in rs])
    for a, c in zip(maps, emu_maps):
        r = hu.relrms(a.cpu(), c)
        print("warm bf16 head map vs emulating oracle rel-rms %.5f" % r)
        assert r <= 5e-3
    allc, allr, worst_rms, worst_cos = _grad_report("warm weights bf16 vs bf16-emulating oracle", grads, emu_grads)
    assert allc >= 0.9995 and allr <= 2e-2 and worst_rms <= 6e-2


def _warm_s_model(dtype):
    g = load_golden("network_yolox_s_warm")
    model = pl_yolo_amd.build_model(_cfg("yolox_s"), int(g["num_classes"]))
    model.load_state_dict(warm_s_state(g))
    model.compute_dtype = dtype
    return g, model.to(hu.DEV).train()


def _fixture_grad_checks(tag, model, g):
    """Stored gradients (every tensor up to 40k elements + the 128-channel 3x3 layers) per tensor, all-parameter L2 norms."""
    params = dict(model.named_parameters())
    want = {k[5:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("grad/")}
    got = {n: params[n].grad for n in want}
    allc, allr, worst_rms, worst_cos = _grad_report(tag, got, want)
    gmax = max(float(v) for k, v in g.items() if k.startswith("gnorm/"))
    worst_norm = 0.0
    for k, v in g.items():
        if k.startswith("gnorm/") and float(v) > 1e-3 * gmax:
            worst_norm = max(worst_norm, abs(float(params[k[6:]].grad.double().norm()) - float(v)) / float(v))
    print("%s: worst relative error of a gradient L2 norm %.4f" % (tag, worst_norm))
    return allc, allr, worst_rms, worst_cos, worst_norm


def test_warm_yolox_s_fp32_vs_reference():
    """yolox_s.yaml itself on the reference's warm weights (tools/gen_golden.py: gen_network_warm_s), parity mode: losses within
    1e-4, the stored gradients within 2e-4 of the reference's."""
    g, model = _warm_s_model("fp32")
    x, labels = torch.from_numpy(g["x"]).to(hu.DEV), torch.from_numpy(g["labels"]).to(hu.DEV)
    out = model(x, labels)
    out["loss"].backward()
    torch.cuda.synchronize()
    for k in ("loss", "loss_iou", "loss_obj", "loss_cls"):
        got, want = float(out[k]), float(g["out/" + k])
        assert abs(got - want) <= 1e-4 * max(1.0, abs(want)), (k, got, want)
    params = dict(model.named_parameters())
    gmax = max(float(np.abs(v).max()) for k, v in g.items() if k.startswith("grad/"))
    for k, v in g.items():
        if k.startswith("grad/"):
            err = float((params[k[5:]].grad.cpu() - torch.from_numpy(v)).abs().max())
            assert err <= 2e-4 * max(float(np.abs(v).max()), 1e-3 * gmax), (k, err)


def test_warm_yolox_s_bf16_end_to_end():
    """The benchmarked kernels (128-channel blocks, 32-channel chunks, 8-row tiles, 64x64 weight-gradient slabs, the pointwise
    kernel with the fused BatchNorm backward) end to end in bf16 -- SimOTA included -- against the REFERENCE's fp32 step on its
    own warm yolox_s weights: losses 2e-3, gradient cosine >= 0.999 over the stored tensors, every tensor >= 0.99."""
    g, model = _warm_s_model("bf16")
    x, labels = torch.from_numpy(g["x"]).to(hu.DEV), torch.from_numpy(g["labels"]).to(hu.DEV)
    with torch.no_grad():
        maps = model(x, None)
    for i, m in enumerate(maps):
        r = hu.relrms(m.float().cpu(), torch.from_numpy(g["maps_train%d" % i]))
        print("warm yolox_s bf16 head map %d rel-rms vs reference %.4f" % (i, r))
        assert r <= 3e-2
    g, model = _warm_s_model("bf16")
    out = model(x, labels)
    out["loss"].backward()
    torch.cuda.synchronize()
    for k in ("loss", "loss_iou", "loss_obj", "loss_cls"):
        got, want = float(out[k]), float(g["out/" + k])
        print("warm yolox_s bf16", k, got, want)
        assert abs(got - want) <= 2e-3 * max(1.0, abs(want)), (k, got, want)
    allc, allr, worst_rms, worst_cos, worst_norm = _fixture_grad_checks("warm yolox_s bf16 vs reference fp32", model, g)
    assert allc >= 0.9995 and worst_cos >= 0.995 and allr <= 2e-2 and worst_norm <= 3e-2   # measured 0.99998 / 0.99969 / 0.0067 / 0.0104


def _warm_s_step(env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        g, model = _warm_s_model("bf16")
        x, labels = torch.from_numpy(g["x"]).to(hu.DEV), torch.from_numpy(g["labels"]).to(hu.DEV)
        out = model(x, labels)
        out["loss"].backward()
        torch.cuda.synchronize()
        sess = [s for k, s in model.runner().sessions.items() if k[4] == "train"][0]
        n_one = sum(1 for op in sess.g.ops if getattr(op, "pw_slabs", 0))
        return g, model, {k: float(out[k]) for k in ("loss", "loss_iou", "loss_obj", "loss_cls")}, n_one
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_warm_yolox_s_bf16_pointwise_backward_in_one_launch():
    """plyolo_conv2d_bwd_pw (dz + data gradient + weight gradient of a pointwise unit in one persistent launch; on the benchmarked
    640x640 batch the 160x160 / 80x80 units take it) forced onto EVERY covered unit of yolox_s at the fixture's 160x160 input
    (PLYOLO_PWBWD_MIN_MB=0): against the plan with the separate launches -- same losses, data-gradient chain bit-identical, so every
    gradient that is not one of the fused units' own weight gradients is unchanged and those agree to the order of their fp32 sums
    -- and against the REFERENCE's step with the bounds of test_warm_yolox_s_bf16_end_to_end."""
    # (PLYOLO_FUSE_BNRED=0 in both: with it the two plans also group the fp32 partials of the BatchNorm sums differently)
    g, m0, l0, n0 = _warm_s_step({"PLYOLO_FUSE_PWBWD": "0", "PLYOLO_FUSE_BNRED": "0"})
    _, m1, l1, n1 = _warm_s_step({"PLYOLO_FUSE_PWBWD": "1", "PLYOLO_PWBWD_MIN_MB": "0", "PLYOLO_FUSE_BNRED": "0"})
    assert n0 == 0 and n1 >= 15, (n0, n1)
    assert l0 == l1, (l0, l1)
    p0, p1 = dict(m0.named_parameters()), dict(m1.named_parameters())
    nexact = 0
    for n in p0:
        if p0[n].grad is None:
            continue
        a, b = p0[n].grad, p1[n].grad
        if torch.equal(a, b):
            nexact += 1
            continue
        assert n.endswith("conv.weight") and a.shape[2] == 1, n      # only pointwise weight gradients may differ at all
        assert float((a - b).abs().max()) <= 2e-5 * max(float(a.abs().max()), 1e-6), n
    print("pointwise backward in one launch: %d units, %d gradient tensors bit-identical to the separate launches" % (n1, nexact))
    for k in ("loss", "loss_iou", "loss_obj", "loss_cls"):
        assert abs(l1[k] - float(g["out/" + k])) <= 2e-3 * max(1.0, abs(float(g["out/" + k]))), k
    allc, allr, worst_rms, worst_cos, worst_norm = _fixture_grad_checks("warm yolox_s bf16, pointwise backward in one launch, vs reference fp32", m1, g)
    assert allc >= 0.9995 and worst_cos >= 0.995 and allr <= 2e-2 and worst_norm <= 3e-2


def test_warm_yolox_s_bf16_bn_reduction_inside_data_gradients():
    """PLYOLO_FUSE_BNRED (default): the bn_act_bwd_reduce pass of a unit rides the store loop of the data gradient that writes the
    unit's output gradient last.  Against the plan with one reduce launch per unit: same losses; the sums differ only in how their
    fp32 partials are grouped (a last-bit difference in a few bf16 dz values), so every gradient agrees to 3e-2 of its tensor's range and
    the all-parameter cosine is 1 to five digits; most units are taken; and against the REFERENCE's step the bounds of
    test_warm_yolox_s_bf16_end_to_end hold."""
    g, m0, l0, _ = _warm_s_step({"PLYOLO_FUSE_BNRED": "0"})
    _, m1, l1, _ = _warm_s_step({"PLYOLO_FUSE_BNRED": "1"})
    s0 = [s for k, s in m0.runner().sessions.items() if k[4] == "train"][0]
    s1 = [s for k, s in m1.runner().sessions.items() if k[4] == "train"][0]
    n0 = sum(1 for op in s0.g.ops if getattr(op, "red_done", False))
    n1 = sum(1 for op in s1.g.ops if getattr(op, "red_done", False))
    nbn = sum(1 for op in s1.g.ops if hasattr(op, "red_done"))
    print("bn_act_bwd_reduce folded into data gradients: %d of %d units" % (n1, nbn))
    assert n0 == 0 and n1 >= 0.6 * nbn, (n0, n1, nbn)
    assert l0 == l1, (l0, l1)
    p0, p1 = dict(m0.named_parameters()), dict(m1.named_parameters())
    dot = na = nb = worst = 0.0
    for n in p0:
        if p0[n].grad is None:
            continue
        a, b = p0[n].grad.double(), p1[n].grad.double()
        e = float((a - b).abs().max()) / max(float(a.abs().max()), 1e-6)
        worst = max(worst, e)
        assert e <= 3e-2, (n, e)      # a flipped last bit of a bf16 dz is 2^-8 of that value; on this fixture's small maps (50 ... 3200 rows per BatchNorm) the
                                      # per-unit sums of the two plans drift apart from 1e-7 at the head to 1e-2 at the stem (tools/diag_bnred.py)
        dot, na, nb = dot + float((a * b).sum()), na + float((a * a).sum()), nb + float((b * b).sum())
    print("worst per-tensor difference (relative to the tensor's largest gradient) %.2e, all-parameter cosine %.8f" % (worst, dot / (na * nb) ** 0.5))
    assert dot / (na * nb) ** 0.5 >= 0.9999
    for k in ("loss", "loss_iou", "loss_obj", "loss_cls"):
        assert abs(l1[k] - float(g["out/" + k])) <= 2e-3 * max(1.0, abs(float(g["out/" + k]))), k
    allc, allr, worst_rms, worst_cos, worst_norm = _fixture_grad_checks("warm yolox_s bf16, reduction inside the data gradients, vs reference fp32", m1, g)
    assert allc >= 0.9995 and worst_cos >= 0.995 and allr <= 2e-2 and worst_norm <= 3e-2


def test_warm_yolox_s_bf16_gradients_vs_emulating_oracle():
    """labels=None path on the warm yolox_s weights, fixed upstream gradients: every HIP bf16 parameter gradient against the
    bf16-emulating oracle (replaces the random-init comparison whose bound was 0.95 / 0.9)."""
    g, model = _warm_s_model("bf16")
    cfg, nc = _cfg("yolox_s"), int(g["num_classes"])
    state = warm_s_state(g)
    x = torch.from_numpy(g["x"])
    gen = torch.Generator().manual_seed(11)
    rs = [torch.randn(m.shape, generator=gen) for m in (g["maps_train0"], g["maps_train1"], g["maps_train2"])]
    emu_maps, emu_grads = _emu_grads(cfg, nc, state, x, rs)
    maps, grads = _maps_grads(model, x.to(hu.DEV), [r.to(hu.DEV) for r in rs])
    for a, c in zip(maps, emu_maps):
        r = hu.relrms(a.cpu(), c)
        print("warm yolox_s bf16 head map vs emulating oracle rel-rms %.5f" % r)
        assert r <= 1e-2
    allc, allr, worst_rms, worst_cos = _grad_report("warm yolox_s bf16 vs bf16-emulating oracle", grads, emu_grads)
    assert allc >= 0.9995 and worst_cos >= 0.995 and allr <= 2.5e-2    # measured 0.99994 / 0.99981 / 0.0107


def test_multi_scale_sessions_are_evicted_and_retraced(monkeypatch):
    """Multi-scale training: one traced session per input shape, least recently used ones dropped beyond the memory budget
    (PLYOLO_SESSION_BUDGET).  With a budget of ~nothing every new shape evicts the previous one; coming back to a shape
    retraces it and gives the same numbers; a backward whose session was evicted in between still runs (autograd holds it)."""
    g, model = _golden_model("fp32")
    model.train()
    gen = torch.Generator().manual_seed(5)
    batches = {}
    for size in (64, 96, 128):
        x = torch.rand(2, 3, size, size, generator=gen) * 255
        lab = torch.zeros(2, 4, 5)
        lab[:, :2, 0] = torch.randint(0, int(g["num_classes"]), (2, 2), generator=gen).float()
        lab[:, :2, 1:3] = (0.3 + 0.4 * torch.rand(2, 2, 2, generator=gen)) * size
        lab[:, :2, 3:5] = 8 + torch.rand(2, 2, 2, generator=gen) * 0.3 * size
        batches[size] = (x.to(hu.DEV), lab.to(hu.DEV))

    def step(size):
        model.zero_grad(set_to_none=True)
        out = model(*batches[size])
        out["loss"].backward()
        torch.cuda.synchronize()
        return float(out["loss"]), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    ref = {size: step(size) for size in (64, 96, 128)}           # default budget: all three stay resident
    r = model.runner()
    assert len(r.sessions) == 3 and all(v.bytes > 0 for v in r.sessions.values())
    monkeypatch.setenv("PLYOLO_SESSION_BUDGET", "1e-12")
    r.sessions.clear()
    for size in (64, 96, 128, 64, 128, 96):
        loss, grads = step(size)
        assert len(r.sessions) == 1                               # only the shape just used
        assert loss == ref[size][0]
        for n, v in grads.items():
            assert float((v - ref[size][1][n]).abs().max()) <= 2e-5 * max(float(ref[size][1][n].abs().max()), 1e-6), (size, n)
    # forward at 64, forward at 96 (evicts the 64 session), THEN the backward of the 64 forward
    model.zero_grad(set_to_none=True)
    out_a = model(*batches[64])
    out_b = model(*batches[96])
    assert [k[1] for k in r.sessions] == [96]
    out_a["loss"].backward()
    torch.cuda.synchronize()
    for n, p in model.named_parameters():
        if p.grad is not None:
            assert float((p.grad - ref[64][1][n]).abs().max()) <= 2e-5 * max(float(ref[64][1][n].abs().max()), 1e-6), n
    monkeypatch.delenv("PLYOLO_SESSION_BUDGET")
    # recently used shapes move to the back of the queue
    r.sessions.clear()
    for size in (64, 96, 64):
        step(size)
    assert [k[1] for k in r.sessions] == [96, 64]


def test_stale_forward_is_refused_and_gradients_accumulate():
    """One set of activation buffers per traced shape: a backward is only valid for the LAST forward of its session (raises
    otherwise).  A backward without zero_grad in between ACCUMULATES like autograd does (Lightning's accumulate_grad_batches):
    the plan overwrites the flat gradient buffer, the runner sets the previous gradient aside and adds it back."""
    from pl_yolo_amd._lib import PlyoloError
    g, model = _golden_model("fp32")
    model.train()
    x = torch.from_numpy(g["x"]).to(hu.DEV)
    labels = torch.from_numpy(g["labels"]).to(hu.DEV)
    out1 = model(x, labels)
    out2 = model(x, labels)
    with pytest.raises(PlyoloError, match="stale forward"):
        out1["loss"].backward()
    out2["loss"].backward()
    g2 = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    out3 = model(x, labels)
    out3["loss"].backward()                 # .grad still holds the previous micro-batch: the same batch again doubles it
    torch.cuda.synchronize()
    for n, p in model.named_parameters():   # (the fp32 parity weight gradient sums with atomics: equal up to summation order)
        if p.grad is not None:
            assert float((p.grad - 2.0 * g2[n]).abs().max()) <= 2e-5 * max(float(g2[n].abs().max()), 1e-6), n
    # another traced shape shares the flat buffer: its backward accumulates as well
    x2, l2 = x[:1].contiguous(), labels[:1].contiguous()
    before = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    out5 = model(x2, l2)
    out5["loss"].backward()
    torch.cuda.synchronize()
    _, solo = _golden_model("fp32")
    solo.train()
    o = solo(x2, l2)
    o["loss"].backward()
    torch.cuda.synchronize()
    gs = {n: p.grad for n, p in solo.named_parameters() if p.grad is not None}
    for n, p in model.named_parameters():
        if p.grad is not None:
            want = before[n] + gs[n]
            assert float((p.grad - want).abs().max()) <= 2e-5 * max(float(want.abs().max()), 1e-6), n
    model.zero_grad(set_to_none=False)      # zeroed in place: the views stay, overwrite == add-to-zero
    out4 = model(x, labels)
    out4["loss"].backward()
    torch.cuda.synchronize()
    for n, p in model.named_parameters():
        if p.grad is not None:
            assert float((p.grad - g2[n]).abs().max()) <= 1e-5 * max(float(g2[n].abs().max()), 1e-6), n


def test_partially_cleared_gradients_accumulate_per_parameter():
    """A backward over PARTIALLY cleared gradients behaves like autograd, parameter by parameter: kept gradients (the previous backward's
    views, untouched) accumulate, gradients zeroed in place or dropped (zero_grad on one parameter group, a frozen stem, a second
    optimizer) are overwritten -- the three cases of runner._accumulating: all kept, all cleared, partial."""
    g, model = _golden_model("fp32")
    model.train()
    x = torch.from_numpy(g["x"]).to(hu.DEV)
    labels = torch.from_numpy(g["labels"]).to(hu.DEV)
    model(x, labels)["loss"].backward()
    torch.cuda.synchronize()
    g1 = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    names = sorted(g1)
    dropped = [n for n in names if n.startswith("backbone.stem")]
    zeroed = [n for n in names if n.startswith("head.")]
    kept = [n for n in names if n not in dropped and n not in zeroed]
    assert dropped and zeroed and kept
    params = dict(model.named_parameters())
    for n in dropped:
        params[n].grad = None
    for n in zeroed:
        params[n].grad.zero_()
    model(x, labels)["loss"].backward()       # same batch again
    torch.cuda.synchronize()
    for n in names:      # (the fp32 parity weight gradient sums with atomics: equal up to summation order)
        want = 2.0 * g1[n] if n in kept else g1[n]
        assert float((params[n].grad - want).abs().max()) <= 2e-5 * max(float(g1[n].abs().max()), 1e-6), n
    # all cleared -> plain overwrite; all kept -> everything doubles again
    model.zero_grad(set_to_none=True)
    model(x, labels)["loss"].backward()
    model(x, labels)["loss"].backward()
    torch.cuda.synchronize()
    for n in names:
        assert float((params[n].grad - 2.0 * g1[n]).abs().max()) <= 2e-5 * max(float(g1[n].abs().max()), 1e-6), n


def _step_with_env(env, dtype, g):
    """One training step of the golden toy model with environment switches applied while its plans are recorded."""
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        _, model = _golden_model(dtype)
        model.train()
        x = torch.from_numpy(g["x"]).to(hu.DEV)
        labels = torch.from_numpy(g["labels"]).to(hu.DEV)
        out = model(x, labels)
        out["loss"].backward()
        torch.cuda.synchronize()
        sess = [s for k, s in model.runner().sessions.items() if k[4] == "train"][0]
        grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
        return {k: float(out[k]) for k in ("loss", "loss_iou", "loss_obj", "loss_cls")}, grads, sess.g.n_lazy
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_fused_bn_backward_and_lane_placements_match_default():
    """Switches that only change HOW the same arithmetic is issued must not change a bit of the result: PLYOLO_FUSE_BNBWD=1 (dz of
    the pointwise units formed inside their data gradient's loader) and every launch-lane placement of neck / head levels (the
    cross-lane events are derived from the tensors the ops touch: a missing one shows up as a wrong or run-to-run different gradient)."""
    g = load_golden("network_yolox_test")
    # (PLYOLO_FUSE_BNRED=0 throughout: with it the BatchNorm sums of a unit are folded by whichever data-gradient kernel writes its
    # output gradient last, and PLYOLO_FUSE_BNBWD swaps that kernel -- same sums, fp32 partials grouped differently)
    base = {"PLYOLO_FUSE_BNRED": "0"}
    l0, g0, _ = _step_with_env(dict(base, PLYOLO_FUSE_BNBWD="0"), "bf16", g)
    import pl_yolo_amd.heads as heads_mod, pl_yolo_amd.necks as necks_mod
    cases = [(dict(base, PLYOLO_FUSE_BNBWD="1"), None, None), (base, [0, 2, 2], 0), (base, [0, 1, 2], 2), (base, [2, 0, 1], 1), (base, [0, 0, 0], 0)]
    old = (heads_mod._HEAD_LANES, heads_mod._HEAD_LANES_FWD, necks_mod._NECK_LANE)
    try:
        for env, hl, nl in cases:
            if hl is not None:
                heads_mod._HEAD_LANES = heads_mod._HEAD_LANES_FWD = hl
                necks_mod._NECK_LANE = nl
            for rep in range(2):
                l1, g1, _ = _step_with_env(env, "bf16", g)
                assert l1 == l0, (env, hl, nl, l0, l1)
                for n in g0:
                    if "_preds" in n and n.endswith(".bias"):   # fp32 atomics in bias_grad: run-to-run last-bit differences
                        assert torch.allclose(g0[n], g1[n], rtol=1e-5, atol=1e-7), (env, hl, nl, n)
                    else:
                        assert torch.equal(g0[n], g1[n]), (env, hl, nl, n)
    finally:
        heads_mod._HEAD_LANES, heads_mod._HEAD_LANES_FWD, necks_mod._NECK_LANE = old


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_lazy_activations_match_materialised(dtype):
    """PLYOLO_LAZY=1: BaseConv outputs that only feed convolutions are never written; the consumers' loaders (pointwise
    and 3x3 forward, weight gradient) apply BatchNorm + SiLU to the producer's raw output instead.  Same arithmetic on
    the same bf16 / fp32 values: losses and every gradient equal the materialised plan's -- and in fp32 the golden
    fixture's (1e-4), so the lazy lowering is pinned to the reference too."""
    from pl_yolo_amd import _lib
    if dtype == "bf16" and not (_lib.lib().plyolo_build_flags() & 1):
        pytest.skip("opt-in kernels: libplyolo_hip.so built without OPTIN=1 (the fp32 parity kernels always take lazy inputs)")
    g = load_golden("network_yolox_test")
    l0, g0, n0 = _step_with_env({"PLYOLO_LAZY": "0"}, dtype, g)
    l1, g1, n1 = _step_with_env({"PLYOLO_LAZY": "1"}, dtype, g)
    assert n0 == 0 and n1 >= 20, (n0, n1)
    for k in l0:
        assert abs(l0[k] - l1[k]) <= 1e-6 * max(1.0, abs(l0[k])), (k, l0[k], l1[k])
    gmax = max(float(v.abs().max()) for v in g0.values())
    for n in g0:
        tol = 1e-6 if dtype == "bf16" else 2e-5   # fp32: the parity weight gradient sums with atomics
        assert float((g0[n] - g1[n]).abs().max()) <= tol * max(float(g0[n].abs().max()), 1e-3 * gmax), n
    if dtype == "fp32":
        for k in l1:
            assert abs(l1[k] - float(g["out/" + k])) <= 1e-4 * max(1.0, abs(float(g["out/" + k]))), k
