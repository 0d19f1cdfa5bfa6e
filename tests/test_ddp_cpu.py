"""Multi-process (world_size 2, gloo) checks of the data-parallel plumbing: identical
start weights after attach(), and the gradient exchange = mean of the per-rank flat
gradient buffers.  Runs on CPU (the kernels themselves need the GPU)."""
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
    # parameters are views of the flat buffer: the broadcast must be visible through them
    p0 = next(model.parameters())
    view_ok = p0.data_ptr() == w.data_ptr()
    if rank == 0:
        np.save(out, np.array([float(same), err, float(view_ok), float(torch.equal(w, w_before))]))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_allreduce_world2(tmp_path):
    out = str(tmp_path / "res.npy")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    same, err, view_ok, rank0_unchanged = np.load(out)
    assert same == 1.0, "ranks disagree on the weights after attach()"
    assert err < 1e-6, "all-reduced gradient is not the mean of the per-rank gradients"
    assert view_ok == 1.0 and rank0_unchanged == 1.0
