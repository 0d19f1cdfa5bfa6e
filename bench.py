#!/usr/bin/env python3
"""Headline benchmark: images/sec of one YOLOX-s 640x640 training step (forward +
SimOTA/loss + backward [+ RCCL gradient all-reduce for N > 1]) at batch 32 per GPU,
bf16 MFMA path, synthetic data.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU; weak scaling)

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     : the dominant kernel of the step (largest total device time in the
                 per-launch hipEvent profile of the recorded plans), priced on its
                 ALGORITHMIC flops/bytes, plus the whole-step fraction of
                 the section-8d conv roofline in "step";
  cpu_baseline : the CPU oracle (a pure-PyTorch fp32 port of the reference path) timed
                 on this box's host cores on a bounded sample of the same workload;
  nms          : boxes/ms of the device post-processing on 1000 boxes/image.
"""
import argparse
import json
import os
import sys
import time


import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# SURVEY.md section 8d, per image, bf16: 3 x fwd conv FLOPs ; 3 x (sum conv-in + conv-out elems) x 2 B
ALGO = {"yolox_s": dict(flops=80.07e9, bytes=443.6e6), "yolox_l": dict(flops=465.9e9, bytes=1229e6),
        "yolox_x": dict(flops=3376.9e9 / 4, bytes=7020e6 / 4), "yolox_nano": dict(flops=8.73e9 * (640 / 416) ** 2, bytes=96.2e6 * (640 / 416) ** 2),
        "yolov7": dict(flops=337.4e9, bytes=1266e6), "yolox_test": dict(flops=1e9, bytes=1e8)}
PEAK_HBM = 8000.0      # GB/s  (MI355X_MICROARCH.md: HBM3E spec)
PEAK_MFMA = 2500.0     # TFLOP/s dense bf16


def synthetic(batch, size, nc, seed, num_gt=30, max_gt=100):
    g = torch.Generator().manual_seed(seed)
    imgs = torch.rand(batch, 3, size, size, generator=g) * 255
    labels = torch.zeros(batch, max_gt, 5)
    labels[:, :num_gt, 0] = torch.randint(0, nc, (batch, num_gt), generator=g).float()
    labels[:, :num_gt, 1:3] = (0.15 + 0.7 * torch.rand(batch, num_gt, 2, generator=g)) * size
    labels[:, :num_gt, 3:5] = 8 + torch.rand(batch, num_gt, 2, generator=g) * 0.3 * size
    return imgs, labels


def _fresh_state(cfg, nc):
    from oracle import net as onet
    torch.manual_seed(96)
    return onet.build_state(cfg, nc)


def _thread_counts(spec):
    """Thread counts of the CPU-baseline sweep: the oracle at batch 4 does not scale to 128 threads (oversubscribed
    it is SLOWER than the reference on 8 cores), so the baseline reports the best of a short sweep."""
    if spec:
        return [int(v) for v in spec.split(",") if v]
    n = os.cpu_count() or 1
    return sorted({t for t in (8, 16, 32, 64) if t <= n} | {min(n, 8)})


def cpu_baseline(cfg, nc, size, b=4, budget_s=25.0, threads=""):
    """The oracle (port of the reference path) on the host cores, bounded sample; thread count = best of a sweep."""
    from oracle import net as onet, detector as odet
    torch.manual_seed(96)
    state = onet.build_state(cfg, nc)
    imgs, labels = synthetic(b, size, nc, 1234)
    t_start = time.time()
    odet.train_step_grads(state, cfg, nc, imgs, labels)  # warm-up (allocator, thread pool)
    sweep, best = {}, None
    n0 = torch.get_num_threads()
    counts = _thread_counts(threads)
    for i, t in enumerate(counts):
        if time.time() - t_start > budget_s and best is not None:
            break
        torch.set_num_threads(t)
        odet.train_step_grads(state, cfg, nc, imgs, labels)
        t0 = time.time()
        out = odet.train_step_grads(state, cfg, nc, imgs, labels)
        dt = time.time() - t0
        sweep[t] = b / dt
        if best is None or b / dt > best[1]:
            best = (t, b / dt)
    # the reported value: MEDIAN of individually timed steps at the best thread count (the sweep's own step included), with the
    # spread beside it -- one timed step per count only ranks the counts (round-3 review: the max of two estimators moved
    # 12.2 -> 10.4 img/s between rounds with no code change)
    torch.set_num_threads(best[0])
    samples = [b / best[1]]
    while len(samples) < 6 and (time.time() - t_start) < budget_s:
        t0 = time.time()
        out = odet.train_step_grads(state, cfg, nc, imgs, labels)
        samples.append(time.time() - t0)
    srt = sorted(samples)
    med = srt[len(srt) // 2] if len(srt) % 2 else 0.5 * (srt[len(srt) // 2 - 1] + srt[len(srt) // 2])
    ref_loss = float(odet.train_step_grads(_fresh_state(cfg, nc), cfg, nc, imgs, labels)[0]["loss"].detach())
    torch.set_num_threads(n0)
    return {"value": b / med, "unit": "images/sec", "cores": best[0], "kind": "port", "loss_b4": ref_loss,
            "steps_timed": len(samples), "min": b / srt[-1], "max": b / srt[0],
            "host_cpus": os.cpu_count(), "thread_sweep_img_per_s": {str(k): v for k, v in sweep.items()},
            "sample": "oracle (pure-PyTorch fp32 port of OneStageD fwd+loss+bwd), %s %dx%d, batch %d; thread count = best of a %s-thread "
                      "sweep (1 warm-up + 1 timed step each); value = median of %d individually timed steps at that count, min / max "
                      "beside it" % (cfg.get("_name", "model"), size, size, b, "/".join(str(c) for c in sweep), len(samples))}


def nms_boxes(B, n):
    """SURVEY 8d set (ii): 200 cluster centres x 5 jittered copies per image -> [B, n, 6] (x1,y1,x2,y2,score,class)."""
    import numpy as np
    rng = np.random.default_rng(0)
    boxes = np.zeros((B, n, 6), np.float32)
    for b in range(B):
        c = np.repeat(rng.uniform(50, 1230, (n // 5, 2)), 5, 0) + rng.normal(0, 4, (n, 2))
        wh = np.exp(rng.uniform(np.log(16), np.log(256), (n, 2)))
        boxes[b, :, 0:2], boxes[b, :, 2:4] = c - wh / 2, c + wh / 2
        boxes[b, :, 4] = rng.uniform(0.01, 1, n)
        boxes[b, :, 5] = rng.integers(0, 80, n)
    return boxes


def nms_cpu_baseline(boxes, nms_thre=0.65, max_det=300, reps=3):
    """The oracle's greedy class-NMS (numpy port of torchvision's published rule, oracle/nms.py) on the same boxes,
    one host thread -- the CPU figure beside the device boxes/ms."""
    from oracle import nms as onms
    B, n = boxes.shape[0], boxes.shape[1]
    t0 = time.time()
    kept = 0
    for _ in range(reps):
        kept = 0
        for b in range(B):
            k = onms.batched_nms(boxes[b, :, 0:4], boxes[b, :, 4], boxes[b, :, 5].astype("int64"), nms_thre)
            kept += min(len(k), max_det)
    ms = (time.time() - t0) * 1e3 / reps
    return {"boxes_per_ms": B * n / ms, "ms_per_batch": ms, "cores": 1, "kind": "port", "kept_mean": kept / B,
            "sample": "oracle.nms.batched_nms (numpy), %d x %d boxes, %d repetitions" % (B, n, reps)}


def nms_bench(device, B=16, n=1000, reps=20, cpu=True):
    import ctypes as C
    from pl_yolo_amd import _lib
    from pl_yolo_amd._lib import NmsDesc, call
    boxes = nms_boxes(B, n)
    d = NmsDesc()
    d.B, d.A, d.C, d.conf_thre, d.nms_thre, d.class_agnostic, d.max_nms, d.max_det, d.numel_threshold = B, n, 80, 0.01, 0.65, 0, 10000, 300, 20000
    wsb = _lib.lib().plyolo_postprocess_workspace(C.byref(d))
    ws = torch.zeros(wsb, dtype=torch.uint8, device=device)
    bt = torch.as_tensor(boxes, device=device)
    nb = torch.full((B,), n, dtype=torch.int32, device=device)
    det = torch.zeros(B, 300, 6, device=device)
    cnt = torch.zeros(B, dtype=torch.int32, device=device)
    st = torch.cuda.current_stream().cuda_stream

    def run():
        call("plyolo_batched_nms", C.byref(d), bt.data_ptr(), n, nb.data_ptr(), det.data_ptr(), cnt.data_ptr(), ws.data_ptr(), wsb, st)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    ms = (time.time() - t0) * 1e3 / reps
    out = {"boxes_per_ms": B * n / ms, "ms_per_batch": ms, "batch": B, "boxes_per_image": n, "kept_mean": float(cnt.float().mean())}
    if cpu:
        out["cpu_baseline"] = nms_cpu_baseline(boxes)
        out["speedup_vs_cpu"] = out["boxes_per_ms"] / out["cpu_baseline"]["boxes_per_ms"]
    return out


# plan-profile label -> the __global__ function that launch runs (csrc/*.hip); forward, data-gradient and the
# stride-2 parity-class jobs are all instances of conv_mfma_kernel / conv_mfma_jobs_kernel
_KERNEL_OF = {"conv_mfma_fwd": "conv_mfma_kernel", "conv_mfma_dgrad": "conv_mfma_kernel", "conv_mfma_dgrad_s2": "conv_mfma_jobs_kernel",
              "conv_wgrad": "conv_wgrad_kernel", "conv3ws_fwd": "conv3ws_kernel", "conv3ws_dgrad": "conv3ws_kernel", "conv_pw_fwd": "conv_pw_kernel", "conv_pw_dgrad": "conv_pw_kernel",
              "conv_mfma_fwd_s2": "conv_mfma_kernel", "conv_pw_dgrad_bn": "conv_pw_kernel", "conv_pw_bwd": "conv_pw_bwd_kernel", "conv_s2d_dgrad": "conv_s2d_kernel"}


def lib_md5():
    """Identity of the HIP library this run loaded (the PMC summaries under profiles/ carry the same field)."""
    import hashlib
    try:
        with open(os.path.join(ROOT, "pl_yolo_amd", "libplyolo_hip.so"), "rb") as f:
            return hashlib.md5(f.read()).hexdigest()
    except OSError:
        return None


def kernel_of(label):
    base = label.split("<")[0]
    if base == "conv_wgrad" and ",k3>" in label and os.environ.get("PLYOLO_WG3", "1") != "0" and os.environ.get("PLYOLO_WG_TRS", "1") == "1":
        return "conv_wgrad3_kernel"      # the 3x3 weight gradient has its own __global__ function (conv_wgrad_mfma.hip)
    return _KERNEL_OF.get(base, base)


def load_pmc_traffic(args):
    """HBM traffic per launch (FETCH_SIZE x2 + WRITE_SIZE, MI355X_MICROARCH.md section HBM) is collected by separate
    rocprofv3 --pmc passes (tools/collect_evidence.sh) and committed under profiles/; it is NOT measured in this run,
    and the entry says which build it belongs to."""
    if (args.model, args.size, args.batch) != ("yolox_s", 640, 32):
        return None
    for tag in ("r06", "r05", "r04", "r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", "%s_pmc_hbm_traffic.json" % tag)
        try:
            with open(path) as f:
                pm = json.load(f)
        except OSError:
            continue
        build = pm.get("build")
        if not isinstance(build, dict) or build.get("lib_md5") != lib_md5():
            # counters of ANOTHER build say nothing about the kernels this run launched: no traffic rather than stale traffic
            return {"kernels": {}, "source": "profiles/%s_pmc_hbm_traffic.json" % tag, "build": build,
                    "stale": "lib_md5 of the profiled build differs from the library loaded by this run; re-run tools/collect_evidence.sh"}
        kernels = {}
        for k, v in pm.items():   # rocpd family names drop the "_kernel" suffix of the __global__ functions
            if isinstance(v, dict) and "read_MB_per_launch" in v:
                kernels[k + "_kernel" if k.startswith("conv_") and not k.endswith("_kernel") else k] = v
        return {"kernels": kernels, "source": "profiles/%s_pmc_hbm_traffic.json" % tag, "build": pm.get("build", tag)}
    return None


def warm_fixture_check(dev):
    """The meaningful bf16 figure: yolox_s.yaml after 50 SGD steps of the REFERENCE (tests/golden/network_yolox_s_warm.npz, made by
    tools/gen_golden.py warm_s: state, batch and the reference's own fp32 losses of that step), stepped once by the benchmarked bf16
    kernels and by the fp32 parity mode -- losses against the reference's (tests/test_gpu_network.py asserts the gradients too)."""
    import numpy as np
    import yaml
    import pl_yolo_amd
    path = os.path.join(ROOT, "tests", "golden", "network_yolox_s_warm.npz")
    try:
        g = dict(np.load(path, allow_pickle=False))
    except OSError as e:
        return {"error": repr(e)}
    with open(os.path.join(ROOT, "configs", "model", "yolox", "yolox_s.yaml")) as f:
        cfg = yaml.safe_load(f)
    sd = {}
    for k, v in g.items():      # float tensors are stored as bf16 bit patterns (tests/conftest.py: warm_s_state)
        if k.startswith("state16/"):
            sd[k[8:]] = torch.from_numpy(v.copy()).view(torch.bfloat16).float()
        elif k.startswith("state/"):
            sd[k[6:]] = torch.from_numpy(v.copy())
    x, labels = torch.from_numpy(g["x"]).to(dev), torch.from_numpy(g["labels"]).to(dev)
    out = {"fixture": "tests/golden/network_yolox_s_warm.npz", "reference_fp32_loss": float(g["out/loss"])}
    for dt in ("bf16", "fp32"):
        m = pl_yolo_amd.build_model(cfg, int(g["num_classes"]))
        m.load_state_dict(sd)
        m.compute_dtype = dt
        m = m.to(dev).train()
        l = m(x, labels)
        worst = 0.0
        for k in ("loss", "loss_iou", "loss_obj", "loss_cls"):
            want = float(g["out/" + k])
            worst = max(worst, abs(float(l[k].detach()) - want) / max(1.0, abs(want)))
        out["hip_%s_loss" % dt] = float(l["loss"].detach())
        out["rel_diff_%s" % dt] = worst      # worst of the four loss terms, relative to max(1, |reference|)
        del m
    return out


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: run torch.distributed.run as a CHILD process (one rank per
    GPU, RCCL over xGMI), relay its output and return its exit code.  Nothing in this process touches the GPU."""
    import socket
    import subprocess
    have = torch.cuda.device_count()   # counting devices does not initialise HIP
    if have < n and os.environ.get("PLYOLO_BENCH_SHARE_GPU", "0") == "1" and have >= 1:
        have = n    # dry run: the N ranks share the GPU(s) that exist, gloo carries the exchange (see main)
    if have < n:
        sys.stderr.write("bench.py: --gpus %d but this node exposes %d GPU(s)\n" % (n, have))
        return 2
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, text=True)
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{")]
    for l in proc.stdout.splitlines():
        if not l.startswith("{"):
            sys.stderr.write(l + "\n")
    if proc.returncode != 0 or not lines:
        sys.stderr.write("bench.py: the %d-rank run failed (exit code %d)\n" % (n, proc.returncode))
        return proc.returncode or 1
    out = json.loads(lines[-1])
    if out.get("n_gpus") != n:
        sys.stderr.write("bench.py: asked for %d GPUs, the ranks report n_gpus=%r\n" % (n, out.get("n_gpus")))
        return 1
    print(lines[-1])
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", default="yolox_s")
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch")
    ap.add_argument("--graph", action="store_true", help="replay the plans as hipGraphs (single lane) instead of the eager multi-stream replay")
    ap.add_argument("--no-graph", action="store_true", help="replay every plan eagerly (default: single-lane plans as hipGraphs, multi-lane plans eagerly)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-out", default=None, help="write the per-launch plan profile (JSON) here")
    ap.add_argument("--cpu-threads", default="", help="comma-separated thread counts of the CPU-baseline sweep (default: a bounded sweep up to the host's cores)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # Not under a launcher: start the N ranks ourselves.  This happens BEFORE anything touches the GPU (a process
        # that has initialised HIP must never exec or fork GPU work), as child processes; rank 0's JSON line is relayed.
        raise SystemExit(self_launch(args.gpus))

    import yaml
    import pl_yolo_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d is running with WORLD_SIZE=%d: launch it as\n  python -m torch.distributed.run --nnodes=1 "
                         "--nproc-per-node %d --master-addr 127.0.0.1 --master-port 29500 bench.py --gpus %d ...\n(or plain "
                         "`python bench.py --gpus %d`, which starts those ranks itself)" % (args.gpus, world, args.gpus, args.gpus, args.gpus))
    # PLYOLO_BENCH_SHARE_GPU=1: a DRY RUN of the N-rank path on a box with fewer GPUs -- the ranks share the device(s) that exist and
    # gloo carries the collectives (two RCCL ranks cannot sit on one device).  Everything the driver's N = 2 ... 8 runs read is
    # exercised end to end across real process boundaries (weight broadcast, bucket hooks, per_rank.exposed_comm_ms_per_step, the
    # barrier + max-over-ranks timing); the VALUE of such a run says nothing about scaling and the line is marked `dry_run_shared_gpu`.
    share = os.environ.get("PLYOLO_BENCH_SHARE_GPU", "0") == "1" and world > 1 and torch.cuda.device_count() < world
    local_dev = local % max(torch.cuda.device_count(), 1) if share else local
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    dist = None
    # PLYOLO_BENCH_FORCE_DDP=1 (under torchrun with one rank) runs the multi-GPU code path -- RCCL init, weight
    # broadcast, gradient all-reduce, barriers -- on a single GPU: a self-test of that path on a 1-GPU box
    ddp_on = world > 1 or (os.environ.get("PLYOLO_BENCH_FORCE_DDP", "0") == "1" and "RANK" in os.environ)
    if ddp_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI

    family = "yolov7" if args.model.startswith("yolov7") else "yolox"
    with open(os.path.join(ROOT, "configs", "model", family, args.model + ".yaml")) as f:
        cfg = yaml.safe_load(f)
    cfg["_name"] = args.model
    nc = 80
    torch.manual_seed(96)  # identical weights on every rank (reference train.py:22)
    model = pl_yolo_amd.build_model(cfg, nc)
    model.compute_dtype = "bf16"
    model = model.to(dev).train()
    runner = model.runner()
    runner.use_graph = True if args.graph else (False if args.no_graph else "auto")
    if ddp_on and os.environ.get("PLYOLO_BENCH_PG_ONLY", "0") != "1":   # PG_ONLY: process group up, no data-parallel schedule (diagnostics)
        from pl_yolo_amd import ddp
        ddp.FORCE_COLLECTIVE = world == 1
        ddp.TIME_EXPOSED = True       # event pair around the end-of-backward wait for the gradient buckets
        ddp.attach(model)
    imgs, labels = synthetic(args.batch, args.size, nc, 1234 + rank)
    imgs, labels = imgs.to(dev), labels.to(dev)

    def step():
        out = model(imgs, labels)
        model.zero_grad(set_to_none=True)
        out["loss"].backward()
        return out

    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    scheds = [v.sched for v in runner.sessions.values() if getattr(v, "sched", None) is not None]
    for sc in scheds:
        sc.exposed = []
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # per rank: its own wall time per step and the communication it could not hide (time the stream that consumes the
    # gradients sat in wait_all() behind the last collectives, hipEvents on that stream)
    exposed_ms = sum(a.elapsed_time(b) for sc in scheds for (a, b) in sc.exposed) / max(args.steps, 1)
    per_rank = None
    if dist is not None:
        mine = torch.tensor([dt * 1e3 / args.steps, exposed_ms], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = {"ms_per_step": [float(v[0]) for v in allr], "exposed_comm_ms_per_step": [float(v[1]) for v in allr]}
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    loss = float(out["loss"].detach())
    ms_step = dt * 1e3 / args.steps
    value = world * args.batch * args.steps / dt

    result = None
    if rank == 0:
        # ---- per-launch profile of one step (eager replay with hipEvents on the launch stream)
        s = [v for k, v in runner.sessions.items() if k[4] == "train"][0]
        st = torch.cuda.current_stream().cuda_stream
        prof = []
        for rep in range(3):
            runner._focus(s, imgs)
            pf = s.fwd.profile(st)
            pb = s.bwd.profile(st)
            if rep:  # first pass warms caches
                prof.append(pf + pb)
                n_fwd_launches = len(pf)
        agg = {}
        for run in prof:
            for (label, ms, fl, by, _lane) in run:
                a = agg.setdefault(label, [0, 0.0, 0.0, 0.0])
                a[0] += 1; a[1] += ms; a[2] += fl; a[3] += by
        nrep = len(prof)
        table = sorted(((k, v[0] / nrep, v[1] / nrep, v[2] / nrep, v[3] / nrep) for k, v in agg.items()), key=lambda r: -r[2])
        total_ms = sum(r[2] for r in table)
        # dominant kernel = the __global__ FUNCTION (all template instances, every role it is launched in) with the
        # largest share of the step; roofline figures are per launch, averaged over that function's launches
        fam = {}
        for (label, cnt, ms_tot, fl_tot, by_tot) in table:
            f = fam.setdefault(kernel_of(label), [0.0, 0.0, 0.0, 0.0])
            f[0] += cnt; f[1] += ms_tot; f[2] += fl_tot; f[3] += by_tot
        pmc = load_pmc_traffic(args)

        def fam_entry(name, v):
            cnt, ms_tot, fl_tot, by_tot = v
            avg_ms = ms_tot / cnt
            tf = fl_tot / cnt / (avg_ms * 1e-3) / 1e12
            gbs = by_tot / cnt / (avg_ms * 1e-3) / 1e9
            mfma_bound = (fl_tot / PEAK_MFMA / 1e12) >= (by_tot / PEAK_HBM / 1e9)
            e = {"kernel": name, "launches_per_step": cnt, "avg_ms": avg_ms, "share_of_step": ms_tot / total_ms,
                 "bound": "mfma" if mfma_bound else "hbm",
                 "achieved": tf if mfma_bound else gbs, "peak": PEAK_MFMA if mfma_bound else PEAK_HBM,
                 "unit": "TFLOP/s" if mfma_bound else "GB/s",
                 "frac": (tf / PEAK_MFMA) if mfma_bound else (gbs / PEAK_HBM),
                 "achieved_tflops": tf, "achieved_gbs": gbs, "algorithmic_bytes_per_launch": by_tot / cnt,
                 "traffic": None}
            t = pmc.get("kernels", {}).get(name) if pmc else None
            if t:  # HBM bytes per launch of this function from the rocprofv3 --pmc passes of the SAME build
                e["traffic"] = (t["read_MB_per_launch"] + t["write_MB_per_launch"]) * 1e6
                e["traffic_detail"] = {"read_bytes": t["read_MB_per_launch"] * 1e6, "write_bytes": t["write_MB_per_launch"] * 1e6,
                                       "measured_in_run": False, "source": pmc.get("source"), "build": pmc.get("build")}
            return e
        fams = sorted((fam_entry(k, v) for k, v in fam.items()), key=lambda e: -e["share_of_step"])
        roof = dict(fams[0])
        roof["families"] = [{k: e[k] for k in ("kernel", "launches_per_step", "avg_ms", "share_of_step", "bound", "achieved", "unit", "frac", "traffic")}
                            for e in fams[:8]]
        a = ALGO.get(args.model, ALGO["yolox_s"])
        scale = (args.size / 640.0) ** 2 if args.model != "yolox_nano" else (args.size / 640.0) ** 2
        t_roof = max(a["flops"] * scale * args.batch / (PEAK_MFMA * 1e12), a["bytes"] * scale * args.batch / (PEAK_HBM * 1e9))
        roof["step"] = {"t_roof_ms": t_roof * 1e3, "t_measured_ms": ms_step, "frac": t_roof * 1e3 / ms_step,
                        "sum_of_launch_ms": total_ms}
        # yardstick, measured in this run on this box: the vendor library's best-case bf16 GEMM (a square 4096^3 torch.mm = hipBLASLt).
        # `peak` above stays the 2.5 PFLOP/s of the guide; this is what the matrix pipes deliver to a tuned library kernel here.
        try:
            if args.no_cpu_baseline:      # (the profiling passes of tools/collect_evidence.sh: keep foreign kernels out of their counter sums)
                raise RuntimeError("skipped with --no-cpu-baseline")
            ya = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
            yb = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
            for _ in range(3):
                torch.mm(ya, yb)
            torch.cuda.synchronize()
            y0, y1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            y0.record()
            for _ in range(20):
                torch.mm(ya, yb)
            y1.record()
            torch.cuda.synchronize()
            lib_tf = 2.0 * 4096 ** 3 * 20 / (y0.elapsed_time(y1) * 1e-3) / 1e12
            roof["library_gemm_yardstick"] = {"kernel": "torch.mm bf16 4096x4096x4096 (hipBLASLt)", "achieved": lib_tf, "unit": "TFLOP/s",
                                              "frac_of_peak": lib_tf / PEAK_MFMA}
            del ya, yb
        except Exception as e:   # the headline line must still print
            roof["library_gemm_yardstick"] = {"error": repr(e)}
        if args.profile_out:
            os.makedirs(os.path.dirname(os.path.abspath(args.profile_out)), exist_ok=True)
            with open(args.profile_out, "w") as f:
                nf = n_fwd_launches
                ops = [[("fwd" if i < nf else "bwd"), l, sum(r[i][1] for r in prof) / nrep, fl, by, ln] for i, (l, _, fl, by, ln) in enumerate(prof[0])]
                json.dump({"columns": ["kernel", "launches", "total_ms", "algo_flops", "algo_bytes"], "rows": table,
                           "sum_ms": total_ms, "ops_columns": ["plan", "kernel", "ms", "algo_flops", "algo_bytes", "lane"], "ops": ops}, f, indent=1)
        result = {
            "metric": ("images/sec fwd+bwd YOLOX-s 640x640 bs32" if (args.model, args.size, args.batch) == ("yolox_s", 640, 32)
                       else "images/sec fwd+bwd %s %dx%d bs%d" % (args.model, args.size, args.size, args.batch)),
            "value": value, "unit": "images/sec",
            "n_gpus": world, "rccl_world_size": (dist.get_world_size() if dist is not None and not share else 1), "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            # what "bf16" means for the headline (round-5 review, weak 1): the contract's 1e-4 on the losses is met by the fp32 parity
            # mode (same graph, plain-FMA kernels); the benchmarked path is asserted at 2e-3 on warm weights (parity_check.warm_fixture)
            "numerics": "bf16 storage + bf16 MFMA with fp32 accumulation, BatchNorm statistics in fp64, BN / SiLU / loss arithmetic fp32; losses "
                        "within 2e-3 of the reference's fp32 step on warm weights (measured ~3e-4), all-parameter gradient cosine >= 0.9995; "
                        "the fp32 parity mode (compute_dtype='fp32') meets the 1e-4 loss contract",
            "config": {"workload": "%s %dx%d, batch %d per GPU, 30 GT/img, fwd + SimOTA/loss + bwd%s" % (
                args.model, args.size, args.size, args.batch, " + RCCL grad all-reduce" if world > 1 else ""),
                "global_batch": world * args.batch, "parallelism": "dp%d" % world,
                "replay": "hipgraph" if args.graph else "eager multi-stream (main, weight-gradient and neck/head side lane)",
                "loss": loss},
            "roofline": roof,
            "lib_md5": lib_md5(),
        }
        if per_rank is not None:
            result["per_rank"] = per_rank          # the driver's scaling runs: which rank is slow, and how much of it is exchange
        if share:
            result["dry_run_shared_gpu"] = True    # N ranks on fewer GPUs over gloo: plumbing check, NOT a scaling measurement
            result["config"]["parallelism"] += " (dry run: %d ranks share %d GPU(s), gloo)" % (world, torch.cuda.device_count())
        if pmc and pmc.get("stale"):
            result["roofline"]["traffic_note"] = pmc["stale"]
        try:
            result["nms"] = nms_bench(dev, cpu=(world == 1 and not args.no_cpu_baseline))
        except Exception as e:  # the headline number must still print
            result["nms"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(cfg, nc, args.size, threads=args.cpu_threads)
            # parity spot-check at the benchmark's own resolution: same weights (seed 96), same
            # synthetic batch of 4 -> HIP bf16 loss vs the oracle's fp32 loss
            torch.manual_seed(96)
            m4 = pl_yolo_amd.build_model(cfg, nc)
            m4.compute_dtype = "bf16"
            m4 = m4.to(dev).train()
            i4, l4 = synthetic(4, args.size, nc, 1234)
            hip_loss = float(m4(i4.to(dev), l4.to(dev))["loss"].detach())
            ref_loss = result["cpu_baseline"].pop("loss_b4")
            # ... and the HIP fp32 parity mode on the same weights and batch: the loss contract (1e-4) is stated for fp32; the bf16
            # number beside it is rounding plus the SimOTA assignments that flip with it on a random-initialised batch of 4 (the warm
            # fixtures of tests/test_gpu_network.py are the instrument for the bf16 path)
            torch.manual_seed(96)
            m4f = pl_yolo_amd.build_model(cfg, nc)
            m4f.compute_dtype = "fp32"
            m4f = m4f.to(dev).train()
            hip_f32 = float(m4f(i4.to(dev), l4.to(dev))["loss"].detach())
            del m4f
            result["parity_check"] = {"batch": 4, "hip_bf16_loss": hip_loss, "hip_fp32_loss": hip_f32, "oracle_fp32_loss": ref_loss,
                                      # bf16 on a RANDOM-INITIALISED batch of 4: rounding + the SimOTA assignments that flip with it (3e-4 ... 2e-3 run to run)
                                      "rel_diff_bf16_random_init": abs(hip_loss - ref_loss) / abs(ref_loss),
                                      "rel_diff_fp32": abs(hip_f32 - ref_loss) / abs(ref_loss)}
            if args.model == "yolox_s":
                result["parity_check"]["warm_fixture"] = warm_fixture_check(dev)
            if args.model == "yolox_s":
                # BASELINE.json configs[0] (the reference's CPU-runnable case): YOLOX-nano 416x416 batch 4 on the host
                # cores, with the HIP fp32 parity mode on the same weights and batch beside it (loss contract 1e-4)
                with open(os.path.join(ROOT, "configs", "model", "yolox", "yolox_nano.yaml")) as f:
                    cfg1 = yaml.safe_load(f)
                cfg1["_name"] = "yolox_nano"
                c1 = cpu_baseline(cfg1, nc, 416, budget_s=10.0, threads=args.cpu_threads)
                torch.manual_seed(96)
                m1 = pl_yolo_amd.build_model(cfg1, nc)
                m1.compute_dtype = "fp32"
                m1 = m1.to(dev).train()
                i1, l1 = synthetic(4, 416, nc, 1234)
                hip1 = float(m1(i1.to(dev), l1.to(dev))["loss"].detach())
                ref1 = c1.pop("loss_b4")
                c1["hip_fp32_loss"], c1["oracle_fp32_loss"], c1["rel_diff"] = hip1, ref1, abs(hip1 - ref1) / abs(ref1)
                result["cpu_baseline_cfg1"] = c1
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
