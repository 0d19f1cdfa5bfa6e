"""Multi-process (world_size 2, gloo) checks of the data-parallel plumbing, on CPU (the kernels themselves need the GPU):
identical start weights after attach(); the gradient exchange = mean of the per-rank flat gradient buffers; the BUCKETED
exchange equals the single-bucket one bit for bit; a data-parallel backward plan records one host hook per bucket.
Plus the bucket planner on its own."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import yaml

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_plan_buckets_covers_live_range_in_reverse_order():
    from pl_yolo_amd.ddp import plan_buckets
    sizes = [64, 128, 64, 1024, 64, 2048, 64, 64]
    offs = list(np.cumsum([0] + sizes[:-1]))
    n_live = int(sum(sizes))
    ready = [0, 0, 1, 2, 2, 5, 7, None]          # owning layer (forward index); the last parameter is used by no layer
    b = plan_buckets(offs, sizes, ready, n_live, target_bytes=4096)
    # emission order = end of the buffer first, contiguous, covering [0, n_live)
    assert b[0][1] == n_live and b[-1][0] == 0
    for (s0, e0, _), (s1, e1, _) in zip(b, b[1:]):
        assert e1 == s0 and s1 < e1
    assert all(e - s >= 1024 or s == 0 for s, e, _ in b)           # ~target-sized (elements * 4 bytes), except the last one
    r = [x[2] for x in b]
    assert r == sorted(r, reverse=True)                              # a bucket never leaves before the one behind it
    # a bucket is ready when the backward has passed the EARLIEST layer owning one of its parameters
    for s, e, rdy in b:
        owners = [ready[i] for i in range(len(offs)) if s <= offs[i] < e and ready[i] is not None]
        assert not owners or rdy <= min(owners)
    one = plan_buckets(offs, sizes, ready, n_live, target_bytes=1 << 40)
    assert one == [(0, n_live, 0)]
    # tail bucket: the first parameters (the last to be ready) get a small bucket of their own, the rest is cut as before
    t = plan_buckets(offs, sizes, ready, n_live, target_bytes=1 << 40, tail_bytes=700)
    assert t == [(offs[2], n_live, 1), (0, offs[2], 0)]              # 64 + 128 elements = 768 bytes >= 700
    t2 = plan_buckets(offs, sizes, ready, n_live, target_bytes=4096, tail_bytes=700)
    assert t2[-1] == (0, offs[2], 0) and t2[0][1] == n_live
    for (s0, e0, _), (s1, e1, _) in zip(t2, t2[1:]):
        assert e1 == s0 and s1 < e1
    assert [x[2] for x in t2] == sorted([x[2] for x in t2], reverse=True)
    assert plan_buckets(offs, sizes, ready, n_live, target_bytes=1 << 40, tail_bytes=1 << 30) == one   # tail >= everything: no split


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pl_yolo_amd
    from pl_yolo_amd import ddp
    with open(os.path.join(ROOT, "configs", "model", "yolox", "yolox_test.yaml")) as f:
        cfg = yaml.safe_load(f)
    torch.manual_seed(100 + rank)  # deliberately different init per rank
    model = pl_yolo_amd.build_model(cfg, 3)
    r = model.runner()
    r.adopt(torch.device("cpu"))
    w_before = r.flat["w"].clone()
    ddp.attach(model)
    w = r.flat["w"]
    gathered = [torch.zeros_like(w) for _ in range(world)]
    dist.all_gather(gathered, w)
    same = all(torch.equal(gathered[0], g) for g in gathered)
    # per-rank gradients -> mean after the exchange
    g = torch.Generator().manual_seed(7 + rank)
    r.flat["g"].copy_(torch.randn(r.flat["n"], generator=g))
    mine = r.flat["g"].clone()
    allg = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allg, mine)
    r.ddp.all_reduce_(r.flat["g"])
    want = sum(allg) / world
    err = float((r.flat["g"] - want).abs().max())
    single = r.flat["g"].clone()
    # bucketed exchange over the live range, small buckets: bit-identical to the single bucket there, dead tail untouched
    os.environ["PLYOLO_BUCKET_MB"] = "0.02"
    model.train()
    s = r._build(2, 64, 64, 8, "train", torch.device("cpu"))          # records the data-parallel backward plan (dry run)
    buckets = s.sched.buckets
    r.flat["g"].copy_(mine)
    r.ddp.all_reduce_buckets_(r.flat["g"], buckets)
    n_live = r.flat["n_live"]
    bucket_equal = torch.equal(r.flat["g"][:n_live], single[:n_live]) and torch.equal(r.flat["g"][n_live:], mine[n_live:])
    covered = sorted((a, b) for a, b, _ in buckets)
    contiguous = covered[0][0] == 0 and covered[-1][1] == n_live and all(x[1] == y[0] for x, y in zip(covered, covered[1:]))
    hooks = s.bwd.hooks()
    # the dead Bottleneck.bn pairs sit behind the live range
    dead = [p for m in model.modules() if hasattr(m, "dead_parameters") for p in m.dead_parameters()]
    dead_behind = all(r.flat["off_of"][id(p)] >= n_live for p in dead) and len(dead) == 16 and n_live < r.flat["n"]
    # parameters are views of the flat buffer: the broadcast must be visible through them
    p0 = r.flat["params"][0]
    view_ok = p0.data_ptr() == w.data_ptr()
    if rank == 0:
        np.save(out, np.array([float(same), err, float(view_ok), float(torch.equal(w, w_before)), float(bucket_equal),
                               float(contiguous), float(len(buckets)), float(hooks), float(dead_behind)]))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_allreduce_world2(tmp_path):
    out = str(tmp_path / "res.npy")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    same, err, view_ok, rank0_unchanged, bucket_equal, contiguous, nbuckets, hooks, dead_behind = np.load(out)
    assert same == 1.0, "ranks disagree on the weights after attach()"
    assert err < 1e-6, "all-reduced gradient is not the mean of the per-rank gradients"
    assert view_ok == 1.0 and rank0_unchanged == 1.0
    assert bucket_equal == 1.0, "bucketed exchange differs from the single-bucket exchange"
    assert contiguous == 1.0 and nbuckets >= 3, "buckets must tile the live range"
    assert hooks == nbuckets, "one host hook per bucket in the data-parallel backward plan"
    assert dead_behind == 1.0


def test_attach_warns_about_a_hardware_queue_count_off_the_optimum(monkeypatch):
    """five or more hardware queues halve the step rate; a launch script that exports
    GPU_MAX_HW_QUEUES != 4 is told so where the data-parallel run starts."""
    import warnings
    from pl_yolo_amd import ddp
    monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        ddp._check_hw_queues()                       # unset: the runtime's default (4), silent
        monkeypatch.setenv("GPU_MAX_HW_QUEUES", "4")
        ddp._check_hw_queues()
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "8")
    with pytest.warns(RuntimeWarning, match="GPU_MAX_HW_QUEUES=8"):
        ddp._check_hw_queues()
