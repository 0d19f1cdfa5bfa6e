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

import torch

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


def cpu_baseline(cfg, nc, size, budget_s=25.0):
    """The oracle (port of the reference path) on the host cores, bounded sample."""
    from oracle import net as onet, detector as odet
    torch.manual_seed(96)
    state = onet.build_state(cfg, nc)
    b = 4
    imgs, labels = synthetic(b, size, nc, 1234)
    t0 = time.time()
    odet.train_step_grads(state, cfg, nc, imgs, labels)  # warm-up
    warm = time.time() - t0
    steps = max(1, min(5, int((budget_s - warm) / max(warm, 1e-3))))
    t0 = time.time()
    for _ in range(steps):
        odet.train_step_grads(state, cfg, nc, imgs, labels)
    dt = (time.time() - t0) / steps
    ref_loss = float(odet.train_step_grads(onet.build_state(cfg, nc) if False else _fresh_state(cfg, nc), cfg, nc, imgs, labels)[0]["loss"].detach())
    return {"value": b / dt, "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port", "loss_b4": ref_loss,
            "sample": "oracle (pure-PyTorch fp32 port of OneStageD fwd+loss+bwd), %s %dx%d, batch %d, %d timed steps after 1 warm-up"
                      % (cfg.get("_name", "model"), size, size, b, steps)}


def nms_bench(device, B=16, n=1000, reps=20):
    import ctypes as C
    import numpy as np
    from pl_yolo_amd import _lib
    from pl_yolo_amd._lib import NmsDesc, call
    rng = np.random.default_rng(0)
    boxes = np.zeros((B, n, 6), np.float32)
    for b in range(B):
        c = np.repeat(rng.uniform(50, 1230, (n // 5, 2)), 5, 0) + rng.normal(0, 4, (n, 2))
        wh = np.exp(rng.uniform(np.log(16), np.log(256), (n, 2)))
        boxes[b, :, 0:2], boxes[b, :, 2:4] = c - wh / 2, c + wh / 2
        boxes[b, :, 4] = rng.uniform(0.01, 1, n)
        boxes[b, :, 5] = rng.integers(0, 80, n)
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
    return {"boxes_per_ms": B * n / ms, "ms_per_batch": ms, "batch": B, "boxes_per_image": n, "kept_mean": float(cnt.float().mean())}


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
    args = ap.parse_args()

    import yaml
    import pl_yolo_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    # PLYOLO_BENCH_FORCE_DDP=1 (under torchrun with one rank) runs the multi-GPU code path -- RCCL init, weight
    # broadcast, gradient all-reduce, barriers -- on a single GPU: a self-test of that path on a 1-GPU box
    ddp_on = world > 1 or (os.environ.get("PLYOLO_BENCH_FORCE_DDP", "0") == "1" and "RANK" in os.environ)
    if ddp_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
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
    if ddp_on:
        from pl_yolo_amd import ddp
        ddp.FORCE_COLLECTIVE = world == 1
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
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    loss = float(out["loss"])
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
        agg = {}
        for run in prof:
            for (label, ms, fl, by) in run:
                a = agg.setdefault(label, [0, 0.0, 0.0, 0.0])
                a[0] += 1; a[1] += ms; a[2] += fl; a[3] += by
        nrep = len(prof)
        table = sorted(((k, v[0] / nrep, v[1] / nrep, v[2] / nrep, v[3] / nrep) for k, v in agg.items()), key=lambda r: -r[2])
        total_ms = sum(r[2] for r in table)
        # dominant kernel = the kernel FAMILY (all template instances of one __global__ function) with the
        # largest share of the step; roofline figures are per launch, averaged over the family's launches
        fam = {}
        for (label, cnt, ms_tot, fl_tot, by_tot) in table:
            f = fam.setdefault(label.split("<")[0], [0.0, 0.0, 0.0, 0.0])
            f[0] += cnt; f[1] += ms_tot; f[2] += fl_tot; f[3] += by_tot
        label, (cnt, ms_tot, fl_tot, by_tot) = max(fam.items(), key=lambda kv: kv[1][1])
        avg_ms = ms_tot / cnt
        tf = fl_tot / cnt / (avg_ms * 1e-3) / 1e12
        gbs = by_tot / cnt / (avg_ms * 1e-3) / 1e9
        mfma_bound = (fl_tot / PEAK_MFMA / 1e12) >= (by_tot / PEAK_HBM / 1e9)
        traffic = None
        try:  # HBM bytes per launch from the committed rocprofv3 --pmc passes (FETCH_SIZE x2 + WRITE_SIZE, tools/rocpd_pmc.py)
            with open(os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.json")) as f:
                pm = json.load(f)
            key = {"conv_mfma_fwd": "conv_mfma", "conv_mfma_dgrad": "conv_mfma", "conv_mfma_dgrad_s2": "conv_mfma"}.get(label, label)
            if args.model == "yolox_s" and args.size == 640 and args.batch == 32 and key in pm:
                traffic = {"bytes_per_launch": (pm[key]["read_MB_per_launch"] + pm[key]["write_MB_per_launch"]) * 1e6,
                           "algorithmic_bytes_per_launch": by_tot / cnt,
                           "source": "profiles/r01_pmc_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, family %s)" % key}
        except OSError:
            pass
        roof = {"kernel": label, "launches_per_step": cnt, "avg_ms": avg_ms, "share_of_step": ms_tot / total_ms,
                "bound": "mfma" if mfma_bound else "hbm",
                "achieved": tf if mfma_bound else gbs, "peak": PEAK_MFMA if mfma_bound else PEAK_HBM,
                "unit": "TFLOP/s" if mfma_bound else "GB/s",
                "frac": (tf / PEAK_MFMA) if mfma_bound else (gbs / PEAK_HBM),
                # HBM bytes per launch from the PMC passes (a number, like `achieved` per launch), or null
                "traffic": traffic["bytes_per_launch"] if traffic else None, "traffic_detail": traffic,
                "achieved_tflops": tf, "achieved_gbs": gbs}
        a = ALGO.get(args.model, ALGO["yolox_s"])
        scale = (args.size / 640.0) ** 2 if args.model != "yolox_nano" else (args.size / 640.0) ** 2
        t_roof = max(a["flops"] * scale * args.batch / (PEAK_MFMA * 1e12), a["bytes"] * scale * args.batch / (PEAK_HBM * 1e9))
        roof["step"] = {"t_roof_ms": t_roof * 1e3, "t_measured_ms": ms_step, "frac": t_roof * 1e3 / ms_step,
                        "sum_of_launch_ms": total_ms}
        if args.profile_out:
            os.makedirs(os.path.dirname(os.path.abspath(args.profile_out)), exist_ok=True)
            with open(args.profile_out, "w") as f:
                nf = s.fwd.size()
                ops = [[("fwd" if i < nf else "bwd"), l, sum(r[i][1] for r in prof) / nrep, fl, by] for i, (l, _, fl, by) in enumerate(prof[0])]
                json.dump({"columns": ["kernel", "launches", "total_ms", "algo_flops", "algo_bytes"], "rows": table,
                           "sum_ms": total_ms, "ops_columns": ["plan", "kernel", "ms", "algo_flops", "algo_bytes"], "ops": ops}, f, indent=1)
        result = {
            "metric": ("images/sec fwd+bwd YOLOX-s 640x640 bs32" if (args.model, args.size, args.batch) == ("yolox_s", 640, 32)
                       else "images/sec fwd+bwd %s %dx%d bs%d" % (args.model, args.size, args.size, args.batch)),
            "value": value, "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "%s %dx%d, batch %d per GPU, 30 GT/img, fwd + SimOTA/loss + bwd%s" % (
                args.model, args.size, args.size, args.batch, " + RCCL grad all-reduce" if world > 1 else ""),
                "global_batch": world * args.batch, "parallelism": "dp%d" % world,
                "replay": "hipgraph" if args.graph else "eager multi-stream (weight-gradient + head-level lanes)",
                "loss": loss},
            "roofline": roof,
        }
        try:
            result["nms"] = nms_bench(dev)
        except Exception as e:  # the headline number must still print
            result["nms"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(cfg, nc, args.size)
            # parity spot-check at the benchmark's own resolution: same weights (seed 96), same
            # synthetic batch of 4 -> HIP bf16 loss vs the oracle's fp32 loss
            torch.manual_seed(96)
            m4 = pl_yolo_amd.build_model(cfg, nc)
            m4.compute_dtype = "bf16"
            m4 = m4.to(dev).train()
            i4, l4 = synthetic(4, args.size, nc, 1234)
            hip_loss = float(m4(i4.to(dev), l4.to(dev))["loss"].detach())
            ref_loss = result["cpu_baseline"].pop("loss_b4")
            result["parity_check"] = {"batch": 4, "hip_bf16_loss": hip_loss, "oracle_fp32_loss": ref_loss,
                                      "rel_diff": abs(hip_loss - ref_loss) / abs(ref_loss)}
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
