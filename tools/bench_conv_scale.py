#!/usr/bin/env python3
"""How much of a small 3x3 launch is fixed cost?  fwd / dgrad of the three 15-GF layer shapes of YOLOX-s at batch 4 ... 128 through the C ABI
(GPU box): time against work -- the intercept is launch + prologue + epilogue + tail, the slope the matrix rate.   python tools/bench_conv_scale.py"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from pl_yolo_amd._lib import BF16, call
import hiputil as hu
SHAPES = [("128->128 @40^2", 40, 128), ("64->64 @80^2", 80, 64), ("256->256 @20^2", 20, 256), ("128->128 @80^2", 80, 128)]
def timeit(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print("%-18s %5s %8s %20s %20s" % ("layer", "N", "GFLOP", "fwd us (TF/s)", "dgrad us (TF/s)"))
for name, H, Cc in SHAPES:
    for N in (4, 8, 16, 32, 64, 128):
        x = torch.randn(N * H * H, Cc, device="cuda").to(torch.bfloat16)
        w = torch.randn(Cc, Cc, 3, 3, device="cuda") / (Cc * 9) ** 0.5
        y = torch.empty(N * H * H, Cc, dtype=torch.bfloat16, device="cuda")
        dx = torch.empty_like(x)
        d = hu.conv_desc(BF16, N, H, H, Cc, Cc, 3, 1, Cc, Cc)
        pk = hu.Packed(w, BF16)
        stats = torch.zeros(hu._lib.STAT_SLOTS * 2 * Cc, dtype=torch.float64, device="cuda")
        st = hu.stream()
        gf = 2.0 * N * H * H * Cc * Cc * 9 / 1e9
        tf = timeit(lambda: call("plyolo_conv2d_fwd", C.byref(d), x.data_ptr(), pk.wp.data_ptr(), None, y.data_ptr(), stats.data_ptr(), st))
        td = timeit(lambda: call("plyolo_conv2d_dgrad", C.byref(d), y.data_ptr(), pk.wpd.data_ptr(), dx.data_ptr(), 0, st))
        print("%-18s %5d %8.1f %12.1f (%6.1f) %12.1f (%6.1f)" % (name, N, gf, tf, gf / tf * 1e3, td, gf / td * 1e3))
