#!/usr/bin/env python3
"""BatchNorm/activation stream kernels through the C ABI at the YOLOX-s layer sizes (GPU box)."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from pl_yolo_amd import _lib
from pl_yolo_amd._lib import BF16, call
import hiputil as hu
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print("%-22s %10s %10s %10s   (us, GB/s algorithmic)" % ("M x C", "fwd", "bwd_reduce", "bwd_dz"))
for (M, Cc) in [(3276800, 32), (819200, 64), (204800, 64), (204800, 128), (204800, 256), (51200, 128), (51200, 256), (51200, 512), (12800, 256), (12800, 512), (12800, 1024)]:
    z = torch.randn(M, Cc, device="cuda").to(torch.bfloat16); y = torch.empty_like(z); dy = torch.randn(M, Cc, device="cuda").to(torch.bfloat16); dz = torch.empty_like(z)
    coef = torch.rand(4 * Cc, device="cuda") + 0.5
    slots = torch.rand(_lib.STAT_SLOTS, 2, Cc, dtype=torch.float64, device="cuda") * M
    slots[:, 1] += M
    g = torch.rand(Cc, device="cuda") + 0.5; b = torch.rand(Cc, device="cuda"); dg = torch.zeros(Cc, device="cuda"); db = torch.zeros(Cc, device="cuda")
    bs = _lib.BnStats(); bs.slots, bs.count, bs.gamma, bs.beta, bs.eps, bs.momentum = slots.data_ptr(), float(M), g.data_ptr(), b.data_ptr(), 1e-3, 0.03
    bsl = torch.zeros(_lib.STAT_SLOTS, 2, Cc, dtype=torch.float64, device="cuda")
    t1 = timeit(lambda: call("plyolo_bn_act_fwd", BF16, M, Cc, z.data_ptr(), Cc, coef.data_ptr(), 1, None, 0, y.data_ptr(), Cc, C.byref(bs), None, st))
    t2 = timeit(lambda: call("plyolo_bn_act_bwd_reduce", BF16, M, Cc, dy.data_ptr(), Cc, z.data_ptr(), Cc, coef.data_ptr(), 1, bsl.data_ptr(), None, st))
    t3 = timeit(lambda: call("plyolo_bn_act_bwd_dz", BF16, M, Cc, dy.data_ptr(), Cc, z.data_ptr(), Cc, coef.data_ptr(), bsl.data_ptr(), g.data_ptr(), dg.data_ptr(), db.data_ptr(), 0, 1, dz.data_ptr(), Cc, None, None, st))
    by = M * Cc * 2
    print("%-22s %5.1f %5.0f %5.1f %5.0f %5.1f %5.0f" % ("%d x %d (%.0f MB)" % (M, Cc, by / 1e6), t1, 2 * by / t1 / 1e3, t2, 2 * by / t2 / 1e3, t3, 3 * by / t3 / 1e3))
