#!/usr/bin/env python3
"""Tail quantisation of the tiled convolution launches: the same layer at batch sizes around the point where its workgroup count crosses
the number of co-resident workgroups (3 per CU x 256 CUs = 768 for the 128-channel pointwise / 3x3 instances).  If a launch of 800
workgroups costs visibly more than one of 760, a row-balanced persistent grid would pay; if time is smooth in the count, it would not.
  python tools/bench_quant.py"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from pl_yolo_amd import _lib
from pl_yolo_amd._lib import BF16, call
import hiputil as hu


def timeit(fn, reps=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


CASES = [("1x1 256->256 @40^2", 256, 256, 40, 1, [20, 24, 28, 30, 32, 34, 36, 40, 48, 60, 62, 64, 68]),
         ("1x1 128->128 @80^2", 128, 128, 80, 1, [12, 14, 15, 16, 18, 24, 28, 30, 32, 34, 36, 40]),
         ("3x3 128->128 @40^2", 128, 128, 40, 3, [32, 44, 48, 51, 52, 56, 64, 96, 100, 104, 112]),
         ("3x3 256->256 @40^2", 256, 256, 40, 3, [16, 20, 24, 25, 26, 28, 32, 40, 48, 51, 52, 56])]
st = hu.stream()
for name, Ci, Co, H, k, batches in CASES:
    print("%s   (tiles of 128 positions x 128 output channels)" % name)
    w = torch.randn(Co, Ci, k, k, device="cuda") / (Ci * k * k) ** 0.5
    pk = hu.Packed(w, BF16)
    stats = torch.zeros(_lib.STAT_SLOTS * 2 * Co, dtype=torch.float64, device="cuda")
    for n in batches:
        M = n * H * H
        x = torch.randn(M, Ci, device="cuda").to(torch.bfloat16)
        y = torch.empty(M, Co, dtype=torch.bfloat16, device="cuda")
        d = hu.conv_desc(BF16, n, H, H, Ci, Co, k, 1, Ci, Co)
        if k == 1:
            nwg = ((M + 127) // 128) * ((Co + 127) // 128)
        else:
            nwg = n * ((H + 7) // 8) * ((H + 15) // 16) * ((Co + 127) // 128)
        t = timeit(lambda: call("plyolo_conv2d_fwd", C.byref(d), x.data_ptr(), pk.wp.data_ptr(), None, y.data_ptr(), stats.data_ptr(), st))
        print("   batch %3d  workgroups %5d (%.2f x 768)  %7.1f us   %6.3f us per 100 workgroups" % (n, nwg, nwg / 768.0, t, t / nwg * 100))
