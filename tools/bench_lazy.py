#!/usr/bin/env python3
"""Forward / weight-gradient launches with and without a lazy input (plyolo_conv_desc.x_coef: BatchNorm + SiLU of the producer applied
while the input is staged) on the same shapes, through the C ABI; needs an OPTIN build (PLYOLO_LIB=...).  python tools/bench_lazy.py"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from pl_yolo_amd._lib import BF16, call
import hiputil as hu
LAYERS = [("pw_128_128_80", 32, 80, 80, 128, 128, 1, 1), ("pw_64_64_160", 32, 160, 160, 64, 64, 1, 1), ("pw_256_256_40", 32, 40, 40, 256, 256, 1, 1),
          ("c3_128_128_80", 32, 80, 80, 128, 128, 3, 1), ("c3_64_64_80", 32, 80, 80, 64, 64, 3, 1), ("c3_128_128_40", 32, 40, 40, 128, 128, 3, 1),
          ("c3_32_32_160", 32, 160, 160, 32, 32, 3, 1)]
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print("%-16s %10s %10s %10s %10s   (us)" % ("layer", "fwd", "fwd lazy", "wgrad", "wgrad lazy"))
for (name, N, H, W, Cin, Cout, k, s) in LAYERS:
    M = N * H * W
    # rotate over several input tensors so that the operands do not sit in the Infinity Cache
    nrot = max(2, int(600e6 // (M * Cin * 2)))
    xs = [torch.randn(M, Cin, device="cuda").to(torch.bfloat16) for _ in range(nrot)]
    w = torch.randn(Cout, Cin, k, k, device="cuda") / (Cin * k * k) ** 0.5
    y = torch.empty(M, Cout, dtype=torch.bfloat16, device="cuda")
    dy = torch.randn(M, Cout, device="cuda").to(torch.bfloat16)
    coef = torch.cat([torch.rand(Cin, device="cuda") + 0.5, torch.randn(Cin, device="cuda") * 0.2, torch.zeros(2 * Cin, device="cuda")]).contiguous()
    pk = hu.Packed(w, BF16)
    stats = torch.zeros(hu._lib.STAT_SLOTS * 2 * Cout, dtype=torch.float64, device="cuda")
    st = hu.stream()
    res = []
    for lazy in (0, 1):
        d = hu.conv_desc(BF16, N, H, W, Cin, Cout, k, s, Cin, Cout)
        if lazy:
            d.x_coef, d.x_coef_ld, d.x_act = coef.data_ptr(), Cin, 1
        pk.set_slabs(d)
        it = [0]
        def fwd():
            it[0] += 1
            call("plyolo_conv2d_fwd", C.byref(d), xs[it[0] % nrot].data_ptr(), pk.wp.data_ptr(), None, y.data_ptr(), stats.data_ptr(), st)
        def wg():
            it[0] += 1
            call("plyolo_conv2d_wgrad", C.byref(d), xs[it[0] % nrot].data_ptr(), dy.data_ptr(), pk.dwp.data_ptr(), st)
        res.append((timeit(fwd), timeit(wg)))
    print("%-16s %10.1f %10.1f %10.1f %10.1f" % (name, res[0][0], res[1][0], res[0][1], res[1][1]))
