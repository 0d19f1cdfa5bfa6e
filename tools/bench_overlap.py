#!/usr/bin/env python3
"""What half-batch pipelining could return ('where a next round should look'): a matrix-bound 3x3 convolution of one half batch
beside the bandwidth-bound bn_act_fwd of the other half, on two streams, against the two launches one after the other -- YOLOX-x layer
shapes at HALF of its per-GPU batch (8 images), through the C ABI.   python tools/bench_overlap.py"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from pl_yolo_amd import _lib
from pl_yolo_amd._lib import BF16, call
import hiputil as hu

N = 8
SHAPES = [("3x3 160->160 @320^2", 320, 160, 3), ("3x3 320->320 @160^2", 160, 320, 3), ("3x3 640->640 @80^2", 80, 640, 3), ("3x3 320->320 @80^2", 80, 320, 3),
          ("1x1 320->320 @160^2", 160, 320, 1), ("3x3 128->128 @80^2 (YOLOX-s, 16 img)", 80, 128, 3)]
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()


def wall(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print("%-40s %10s %10s %12s %12s %8s" % ("layer, half batch", "conv us", "bn us", "in turn us", "side by side", "hidden"))
for name, H, Cc, k in SHAPES:
    n = 16 if "YOLOX-s" in name else N
    M = n * H * H
    # two sets of operands: the convolution works on half A, the BatchNorm stream on half B (distinct tensors, as in the pipeline)
    xa = torch.randn(M, Cc, device="cuda").to(torch.bfloat16)
    za = torch.empty(M, Cc, dtype=torch.bfloat16, device="cuda")
    zb = torch.randn(M, Cc, device="cuda").to(torch.bfloat16)
    yb = torch.empty_like(zb)
    w = torch.randn(Cc, Cc, k, k, device="cuda") / (Cc * k * k) ** 0.5
    d = hu.conv_desc(BF16, n, H, H, Cc, Cc, k, 1, Cc, Cc)
    pk = hu.Packed(w, BF16)
    stats = torch.zeros(_lib.STAT_SLOTS * 2 * Cc, dtype=torch.float64, device="cuda")
    coef = torch.rand(4 * Cc, device="cuda") + 0.5
    slots = torch.rand(_lib.STAT_SLOTS, 2, Cc, dtype=torch.float64, device="cuda") * M
    slots[:, 1] += M
    g = torch.rand(Cc, device="cuda") + 0.5
    b = torch.rand(Cc, device="cuda")
    bs = _lib.BnStats()
    bs.slots, bs.count, bs.gamma, bs.beta, bs.eps, bs.momentum = slots.data_ptr(), float(M), g.data_ptr(), b.data_ptr(), 1e-3, 0.03

    def conv(st):
        call("plyolo_conv2d_fwd", C.byref(d), xa.data_ptr(), pk.wp.data_ptr(), None, za.data_ptr(), stats.data_ptr(), st)

    def bn(st):
        call("plyolo_bn_act_fwd", BF16, M, Cc, zb.data_ptr(), Cc, coef.data_ptr(), 1, None, 0, yb.data_ptr(), Cc, C.byref(bs), None, st)

    cur = torch.cuda.current_stream()

    R = 10   # launches per stream between one fork and one join: the fork / join events (tens of microseconds in this harness) are amortised,
             # as they are in a launch plan whose lanes run long sequences between their cross-lane events

    def in_turn():
        for _ in range(R):
            conv(cur.cuda_stream)
            bn(cur.cuda_stream)

    def side_by_side():
        s0.wait_stream(cur); s1.wait_stream(cur)
        for _ in range(R):
            conv(s0.cuda_stream)
        for _ in range(R):
            bn(s1.cuda_stream)
        cur.wait_stream(s0); cur.wait_stream(s1)

    def side_by_side_matched():     # as many BatchNorm launches as fit into the convolutions' time: both streams busy for the whole window
        s0.wait_stream(cur); s1.wait_stream(cur)
        for _ in range(R):
            conv(s0.cuda_stream)
        for _ in range(RB):
            bn(s1.cuda_stream)
        cur.wait_stream(s0); cur.wait_stream(s1)

    tc = wall(lambda: conv(cur.cuda_stream))
    tb = wall(lambda: bn(cur.cuda_stream))
    RB = max(1, int(R * tc / tb))
    ts = wall(in_turn, 5) / R
    tp = wall(side_by_side, 5) / R
    tm = wall(side_by_side_matched, 5)
    # matched window: R convolutions + RB streams took tm; in turn they take R * tc + RB * tb
    print("%-40s %10.1f %10.1f %12.1f %12.1f %7.0f%%   matched: %d conv + %d bn in %.0f us against %.0f in turn (%.2fx)"
          % (name, tc, tb, ts, tp, 100.0 * (ts - tp) / min(tc, tb), R, RB, tm, R * tc + RB * tb, (R * tc + RB * tb) / tm))
