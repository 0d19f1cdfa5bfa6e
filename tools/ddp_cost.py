#!/usr/bin/env python3
"""Where the per-bucket cost of the data-parallel backward goes (GPU box, one-rank RCCL group): step time with the real
collective, with a no-op hook, and without the schedule.   python tools/ddp_cost.py [bucket_MB]"""
import os, socket, sys, time
import torch, torch.distributed as dist, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, pl_yolo_amd
from pl_yolo_amd import ddp
os.environ["PLYOLO_BUCKET_MB"] = sys.argv[1] if len(sys.argv) > 1 else "9"
dev = torch.device("cuda:0")
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "model", "yolox", "yolox_s.yaml")))
imgs, labels = bench.synthetic(32, 640, 80, 1234)
imgs, labels = imgs.to(dev), labels.to(dev)

def run(mode):
    torch.manual_seed(96)
    model = pl_yolo_amd.build_model(cfg, 80); model.compute_dtype = "bf16"; model = model.to(dev).train()
    host = []
    if mode != "plain":
        ddp.FORCE_COLLECTIVE = True
        ddp.attach(model)
        r = model.runner()
        orig = r.ddp.all_reduce_
        def timed(flat, a=0, b=None, **kw):
            t0 = time.perf_counter()
            if mode == "native":
                b_ = flat.numel() if b is None else b
                _lib.call("plyolo_rccl_allreduce_bucket", COMM, flat.data_ptr() + 4 * a, b_ - a, 1, torch.cuda.current_stream().cuda_stream)
                out = flat
            else:
                out = flat if mode == "noop" else orig(flat, a, b, **kw)
            host.append(time.perf_counter() - t0)
            return out
        r.ddp.all_reduce_ = timed
    def step():
        out = model(imgs, labels); model.zero_grad(set_to_none=True); out["loss"].backward(); return out
    for _ in range(5): step()
    torch.cuda.synchronize(); host.clear()
    t0 = time.perf_counter()
    for _ in range(40): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 40 * 1e3
    nb = len(host) // 40 if host else 0
    print("DBG=%s " % os.environ.get("PLYOLO_DDP_DBG", "0") + "%-6s %.3f ms/step  buckets %d  host time inside the hooks %.3f ms/step" % (mode, dt, nb, sum(host) / 40 * 1e3))
    r = model.runner()
    sess = [v for k, v in r.sessions.items() if k[4] == "train"][0]
    st = torch.cuda.current_stream().cuda_stream
    for rep in range(3):
        r._focus(sess, imgs)
        sess.fwd.lane_times(st)
        b = sess.bwd.lane_times(st)
    print("       bwd lane ends (ms): " + "  ".join("lane%d %.3f" % (i, x) for i, x in enumerate(b[:-1])) + "  join %.3f" % b[-1])
    ddp.FORCE_COLLECTIVE = False

import ctypes as C, glob
from pl_yolo_amd import _lib
path = glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so*"))[0]
rccl = C.CDLL(path)
class UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]
uid = UniqueId(); assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
COMM = C.c_void_p()
rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
assert rccl.ncclCommInitRank(C.byref(COMM), 1, uid, 0) == 0
assert _lib.lib().plyolo_rccl_set_library(path.encode()) == 0
for rep in range(2):
    os.environ["PLYOLO_DDP_DBG"] = "0"; run("plain"); run("noop")
    os.environ["PLYOLO_DDP_DBG"] = "1"; run("noop")
    os.environ["PLYOLO_DDP_DBG"] = "2"; run("noop")
os.environ["PLYOLO_DDP_DBG"] = "0"
dist.destroy_process_group()
