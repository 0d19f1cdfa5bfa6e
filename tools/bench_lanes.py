#!/usr/bin/env python3
"""How much do independent mid-size launches overlap when issued on separate HIP streams?
(GPU box)  python tools/bench_lanes.py"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from pl_yolo_amd._lib import BF16, call
import hiputil as hu

def mk(N, H, W, Cin, Cout, k):
    x = torch.randn(N * H * W, Cin, device="cuda").to(torch.bfloat16)
    w = torch.randn(Cout, Cin, k, k, device="cuda") / (Cin * k * k) ** 0.5
    y = torch.empty(N * H * W, Cout, dtype=torch.bfloat16, device="cuda")
    d = hu.conv_desc(BF16, N, H, W, Cin, Cout, k, 1, Cin, Cout)
    pk = hu.Packed(w, BF16)
    z = torch.randn(N * H * W, Cout, device="cuda").to(torch.bfloat16)
    coef = torch.rand(4 * Cout, device="cuda")
    return dict(x=x, y=y, d=d, pk=pk, z=z, coef=coef, M=N * H * W, C=Cout)

def conv(b, st): call("plyolo_conv2d_fwd", C.byref(b["d"]), b["x"].data_ptr(), b["pk"].wp.data_ptr(), None, b["y"].data_ptr(), None, st)
def bn(b, st): call("plyolo_bn_act_fwd", BF16, b["M"], b["C"], b["z"].data_ptr(), b["C"], b["coef"].data_ptr(), 1, None, 0, b["y"].data_ptr(), b["C"], None, None, st)

for name, shape in [("3x3 128->128 @40", (32, 40, 40, 128, 128, 3)), ("1x1 256->128 @40", (32, 40, 40, 256, 128, 1)), ("3x3 128->128 @20", (32, 20, 20, 128, 128, 3)),
                    ("1x1 512->256 @20", (32, 20, 20, 512, 256, 1)), ("3x3 64->64 @80", (32, 80, 80, 64, 64, 3))]:
    for fn, fname in ((conv, "conv"), (bn, "bn_act")):
        for lanes in (1, 2, 3, 4):
            bufs = [mk(*shape) for _ in range(lanes)]
            streams = [torch.cuda.Stream() for _ in range(lanes)]
            reps = 20
            def run():
                for _ in range(reps):
                    for b, s in zip(bufs, streams):
                        fn(b, s.cuda_stream)
            run(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for s in streams: s.wait_stream(torch.cuda.current_stream())
            run()
            for s in streams: torch.cuda.current_stream().wait_stream(s)
            e1.record(); torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / reps * 1e3
            print("%-18s %-7s lanes %d: %7.1f us per round = %6.1f us per launch" % (name, fname, lanes, t, t / lanes))
