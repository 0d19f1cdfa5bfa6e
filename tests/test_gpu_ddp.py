"""-m gpu: the data-parallel backward plan on one MI355X (a one-rank RCCL group exercises every piece -- host hooks on
the communication lane, per-bucket unpack launches, collectives enqueued while the plan replays; the multi-GPU runs are the
driver's), and plyolo_rccl_allreduce_bucket through the C ABI on a communicator of its own."""
import ctypes as C
import glob
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import yaml

pytestmark = pytest.mark.gpu

import pl_yolo_amd  # noqa: E402
from pl_yolo_amd import ddp, _lib  # noqa: E402
from conftest import ROOT  # noqa: E402
from oracle import detector as odet  # noqa: E402
import hiputil as hu  # noqa: E402


@pytest.fixture()
def one_rank_rccl():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=torch.device(hu.DEV))
    ddp.FORCE_COLLECTIVE = True
    yield
    ddp.FORCE_COLLECTIVE = False
    dist.destroy_process_group()


@pytest.mark.parametrize("family,name,lanes", [("yolox", "yolox_s", "1"), ("yolov7", "yolov7_test", "1"), ("yolox", "yolox_s", "0")])
def test_bucketed_exchange_inside_the_backward_plan(one_rank_rccl, family, name, lanes, monkeypatch):
    """Same weights, same batch, with and without the data-parallel schedule: in a one-rank group the mean is the identity,
    so every gradient must come out bit-identical -- while the plan really issues one RCCL all-reduce per bucket from its
    host hooks (counted), after per-bucket unpack launches instead of the single final one."""
    monkeypatch.setenv("PLYOLO_LANES", lanes)
    monkeypatch.setenv("PLYOLO_BUCKET_MB", "4" if name == "yolox_s" else "0.05")
    with open(os.path.join(ROOT, "configs", "model", family, name + ".yaml")) as f:
        cfg = yaml.safe_load(f)
    nc = 80 if name == "yolox_s" else 3
    size = 320 if name == "yolox_s" else 128
    imgs, labels = odet.synthetic_batch(4, size, nc, num_gt=6, max_gt=10, seed=5)
    imgs, labels = imgs.to(hu.DEV), labels.to(hu.DEV)

    def run(with_ddp):
        torch.manual_seed(96)
        model = pl_yolo_amd.build_model(cfg, nc)
        model.compute_dtype = "bf16"
        model = model.to(hu.DEV).train()
        if with_ddp:
            ddp.attach(model)
        calls = []
        if with_ddp:
            orig = model.runner().ddp.all_reduce_
            model.runner().ddp.all_reduce_ = lambda flat, a=0, b=None, **kw: (calls.append((a, b)), orig(flat, a, b, **kw))[1]
        for _ in range(2):                       # second step: the hooks fire on every replay
            out = model(imgs, labels)
            model.zero_grad(set_to_none=True)
            out["loss"].backward()
        torch.cuda.synchronize()
        sess = [s for k, s in model.runner().sessions.items() if k[4] == "train"][0]
        return float(out["loss"]), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}, sess, calls

    l0, g0, s0, _ = run(False)
    l1, g1, s1, calls = run(True)
    assert s0.sched is None and s0.bwd.hooks() == 0
    nb = len(s1.sched.buckets)
    print("%s: %d buckets, %d hooks, %d collectives in 2 steps" % (name, nb, s1.bwd.hooks(), len(calls)))
    assert nb >= 3 and s1.bwd.hooks() == nb and len(calls) == 2 * nb
    flat = s1.sched.runner.flat
    assert sorted(c[:2] for c in calls[:nb]) == sorted((a, b) for a, b, _ in s1.sched.buckets)
    assert max(b for _, b, _ in s1.sched.buckets) == flat["n_live"] <= flat["n"]
    assert l0 == l1
    assert set(g0) == set(g1)
    for n in g0:
        assert torch.equal(g0[n], g1[n]), n


def test_hipgraph_replay_refuses_a_data_parallel_plan(one_rank_rccl):
    from pl_yolo_amd._lib import PlyoloError
    with open(os.path.join(ROOT, "configs", "model", "yolox", "yolox_test.yaml")) as f:
        cfg = yaml.safe_load(f)
    model = pl_yolo_amd.build_model(cfg, 3).to(hu.DEV).train()
    ddp.attach(model)
    model.runner().use_graph = True
    imgs, labels = odet.synthetic_batch(2, 64, 3, num_gt=2, max_gt=4, seed=1)
    out = model(imgs.to(hu.DEV), labels.to(hu.DEV))
    with pytest.raises(PlyoloError, match="host hooks"):
        out["loss"].backward()


def test_rccl_allreduce_bucket_through_the_c_abi():
    """plyolo_rccl_allreduce_bucket on a one-rank communicator created here with the RCCL library torch ships
    (ncclGetUniqueId / ncclCommInitRank through ctypes): sum and average of one rank leave the bucket unchanged, the
    call is ordered on the given stream, a null communicator is refused."""
    path = glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so*"))[0]
    rccl = C.CDLL(path)
    class UniqueId(C.Structure):                 # ncclUniqueId: 128 opaque bytes, passed BY VALUE to ncclCommInitRank
        _fields_ = [("internal", C.c_char * 128)]
    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    torch.cuda.set_device(0)
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    lib = _lib.lib()
    assert lib.plyolo_rccl_set_library(path.encode()) == 0
    torch.manual_seed(11)
    g = torch.randn(1 << 20, device=hu.DEV)
    want = g.clone()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())      # randn / clone were queued on the current stream: without this the side stream may double g before the clone has read it
    with torch.cuda.stream(st):
        g.mul_(2.0)                                                  # queued before the collective on the same stream
        _lib.call("plyolo_rccl_allreduce_bucket", comm, g.data_ptr() + 4 * 1024, g.numel() - 2048, 1, st.cuda_stream)
        _lib.call("plyolo_rccl_allreduce_bucket", comm, g.data_ptr(), 1024, 0, st.cuda_stream)
    st.synchronize()
    assert torch.equal(g, want * 2.0)
    assert lib.plyolo_rccl_allreduce_bucket(None, g.data_ptr(), 16, 1, st.cuda_stream) != 0
    assert b"null communicator" in lib.plyolo_last_error()
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    rccl.ncclCommDestroy(comm)


def test_two_processes_one_gpu_gloo(tmp_path, monkeypatch):
    """TWO ranks (two processes sharing cuda:0, gloo for the exchange): the only multi-process run of the bucketed backward
    this pool allows.  Ranks start from different weights and see different batches; after attach() + two steps both must
    hold rank 0's weights and bit-identical gradients equal to the mean of the two single-process gradients."""
    import subprocess
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", PLYOLO_BUCKET_MB="0.02")
    procs = []
    for r in range(2):
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "ddp_two_rank_worker.py"), str(tmp_path / ("r%d.npz" % r)), "2"],
                                      env=dict(env, RANK=str(r)), cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=240)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("the two-rank run did not finish in 240 s (a rank is waiting for a collective the other never issued)")
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-2000:] for o in outs)
    a, b = np.load(tmp_path / "r0.npz"), np.load(tmp_path / "r1.npz")
    assert int(a["buckets"]) >= 3 and int(a["hooks"]) == int(a["buckets"]) == int(b["buckets"])
    gkeys = [k for k in a.files if k.startswith("g/")]
    assert gkeys and set(a.files) == set(b.files)
    for k in a.files:
        if k.startswith("w/") or k.startswith("g/"):
            assert np.array_equal(a[k], b[k]), k                      # same weights (rank 0's), same averaged gradients
    # the single-process gradients of the two batches, from rank 0's initial weights
    monkeypatch.setenv("PLYOLO_BUCKET_MB", "0.02")
    with open(os.path.join(ROOT, "configs", "model", "yolox", "yolox_test.yaml")) as f:
        cfg = yaml.safe_load(f)
    grads = []
    for r in range(2):
        torch.manual_seed(96)
        model = pl_yolo_amd.build_model(cfg, 3)
        model.compute_dtype = "bf16"
        model = model.to(hu.DEV).train()
        imgs, labels = odet.synthetic_batch(2, 64, 3, num_gt=3, max_gt=6, seed=5 + r)
        for _ in range(2):
            out = model(imgs.to(hu.DEV), labels.to(hu.DEV))
            model.zero_grad(set_to_none=True)
            out["loss"].backward()
        torch.cuda.synchronize()
        grads.append({n: p.grad.detach().float().cpu().numpy() for n, p in model.named_parameters() if p.grad is not None})
        if r == 0:
            for n, p in model.named_parameters():
                assert np.array_equal(a["w/" + n], p.detach().float().cpu().numpy()), n
    for n in grads[0]:
        want = (grads[0][n] + grads[1][n]) * np.float32(0.5)
        assert np.array_equal(a["g/" + n], want), n


def test_two_gpus_rccl_weights_and_gradients_equal_across_ranks(tmp_path):
    """TWO ranks on TWO GPUs, RCCL (ReduceOp.AVG over xGMI) carrying the bucketed exchange of the backward plan: after attach() +
    two steps both ranks hold rank 0's weights and the same averaged gradients.  Skipped on a one-GPU box (every gpurun box of
    this pool exposes one GPU); the driver's multi-GPU node runs it."""
    import subprocess
    import sys
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (this box exposes %d)" % torch.cuda.device_count())
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", PLYOLO_BUCKET_MB="0.02", PLYOLO_TWO_RANK_BACKEND="nccl",
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "ddp_two_rank_worker.py"), str(tmp_path / ("r%d.npz" % r)), "2"],
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=300)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("the two-GPU run did not finish in 300 s (a rank is waiting for a collective the other never issued)")
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-2000:] for o in outs)
    a, b = np.load(tmp_path / "r0.npz"), np.load(tmp_path / "r1.npz")
    assert int(a["buckets"]) >= 3 and int(a["hooks"]) == int(a["buckets"]) == int(b["buckets"])
    for k in a.files:
        if k.startswith("w/") or k.startswith("g/"):
            assert np.array_equal(a[k], b[k]), k
