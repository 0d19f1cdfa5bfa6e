#!/usr/bin/env python3
"""3x3 stride-1 layers of YOLOX-x (B=16, 1280^2) through the C ABI under forced output-channel tiles (PLYOLO_FORCE_BN): what the 128-channel
column blocks cost on 80 / 160 / 320-channel layers (160 = 1.25 blocks, 320 = 2.5).   python tools/bench_conv_x.py"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from pl_yolo_amd._lib import BF16, call
import hiputil as hu
SHAPES = [("80->80 @320^2", 320, 80), ("160->160 @160^2", 160, 160), ("320->320 @80^2", 80, 320), ("320->320 @160^2", 160, 320), ("640->640 @40^2", 40, 640),
          ("128->128 @160^2 (ref)", 160, 128), ("256->256 @80^2 (ref)", 80, 256)]
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
N = 16
print("%-24s %8s %8s %22s %22s" % ("layer (B=16)", "GFLOP", "BN", "fwd us (TF/s)", "dgrad us (TF/s)"))
for name, H, Cc in SHAPES:
    x = torch.randn(N * H * H, Cc, device="cuda").to(torch.bfloat16)
    w = torch.randn(Cc, Cc, 3, 3, device="cuda") / (Cc * 9) ** 0.5
    y = torch.empty(N * H * H, Cc, dtype=torch.bfloat16, device="cuda")
    dx = torch.empty_like(x)
    d = hu.conv_desc(BF16, N, H, H, Cc, Cc, 3, 1, Cc, Cc)
    pk = hu.Packed(w, BF16)
    stats = torch.zeros(hu._lib.STAT_SLOTS * 2 * Cc, dtype=torch.float64, device="cuda")
    st = hu.stream()
    gf = 2.0 * N * H * H * Cc * Cc * 9 / 1e9
    for bn in (os.environ.get("BNS", "128,64,32").split(",")):
        os.environ["PLYOLO_FORCE_BN"] = bn
        tf = timeit(lambda: call("plyolo_conv2d_fwd", C.byref(d), x.data_ptr(), pk.wp.data_ptr(), None, y.data_ptr(), stats.data_ptr(), st))
        td = timeit(lambda: call("plyolo_conv2d_dgrad", C.byref(d), y.data_ptr(), pk.wpd.data_ptr(), dx.data_ptr(), 0, st))
        print("%-24s %8.1f %8s %12.1f (%7.1f) %12.1f (%7.1f)" % (name, gf, bn, tf, gf / tf * 1e3, td, gf / td * 1e3))
    del os.environ["PLYOLO_FORCE_BN"]
