#!/usr/bin/env python3
"""Micro-benchmark of the pointwise-unit backward (GPU box): plyolo_conv2d_bwd_pw against the three launches it replaces
(bn_act_bwd_dz, conv2d_dgrad, conv2d_wgrad) on the YOLOX-s B=32 shapes, hipEvent-timed, operands rotated through NBUF buffer sets
(> 256 MiB in total) so that nothing is served from the Infinity Cache.   python tools/bench_pwbwd.py [filter] [acc]"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("PLYOLO_PWBWD_MIN_MB", "0")
from pl_yolo_amd._lib import BF16, ACT, BnBwdFuse, STAT_SLOTS, call
import hiputil as hu
SHAPES = [("c128_80", 32, 80, 80, 128), ("c64_80", 32, 80, 80, 64), ("c64_160", 32, 160, 160, 64), ("c32_160", 32, 160, 160, 32),
          ("c128_40", 32, 40, 40, 128)]
flt = sys.argv[1] if len(sys.argv) > 1 else ""
acc = int(sys.argv[2]) if len(sys.argv) > 2 else 0
reps = 10
lib = hu._lib.lib()
for (name, N, H, W, Cc) in SHAPES:
    if flt and flt not in name: continue
    M = N * H * W
    per_set = M * Cc * 2 * 5
    NBUF = max(2, int(600e6 // per_set) + 1)
    dev = "cuda"
    w = torch.randn(Cc, Cc, 1, 1, device=dev) / Cc ** 0.5
    pk = hu.Packed(w, BF16)
    d = hu.conv_desc(BF16, N, H, W, Cc, Cc, 1, 1, Cc, Cc)
    pk.set_slabs(d)
    ns = lib.plyolo_conv2d_bwd_pw_slabs(C.byref(d))
    dwp = torch.zeros(max(ns, 1) * Cc * Cc, dtype=torch.float32, device=dev)
    sets = []
    for _ in range(NBUF):
        sets.append(dict(z=(torch.randn(M, Cc, device=dev) * 1.5).to(torch.bfloat16), dout=torch.randn(M, Cc, device=dev).to(torch.bfloat16),
                         x=torch.randn(M, Cc, device=dev).to(torch.bfloat16), dx=torch.zeros(M, Cc, dtype=torch.bfloat16, device=dev),
                         dz=torch.zeros(M, Cc, dtype=torch.bfloat16, device=dev)))
    gamma = torch.rand(Cc, device=dev) + 0.5
    mean, invstd, beta = torch.randn(Cc, device=dev) * 0.1, torch.rand(Cc, device=dev) + 0.5, torch.randn(Cc, device=dev) * 0.1
    scale = gamma * invstd
    coef = torch.cat([scale, beta - mean * scale, mean, invstd]).contiguous()
    bslots = torch.zeros(STAT_SLOTS * 2 * Cc, dtype=torch.float64, device=dev)
    a = ACT["silu"]
    call("plyolo_bn_act_bwd_reduce", BF16, M, Cc, sets[0]["dout"].data_ptr(), Cc, sets[0]["z"].data_ptr(), Cc, coef.data_ptr(), a, bslots.data_ptr(), None, hu.stream())
    dg, db = torch.zeros(Cc, device=dev), torch.zeros(Cc, device=dev)
    fs = []
    for s in sets:
        f = BnBwdFuse()
        f.dout, f.dout_ld, f.z, f.z_ld, f.coef, f.bslots = s["dout"].data_ptr(), Cc, s["z"].data_ptr(), Cc, coef.data_ptr(), bslots.data_ptr()
        f.gamma, f.dgamma, f.dbeta, f.act = gamma.data_ptr(), dg.data_ptr(), db.data_ptr(), a
        fs.append(f)
    st = hu.stream()
    def one(i):
        s, f = sets[i % NBUF], fs[i % NBUF]
        call("plyolo_conv2d_bwd_pw", C.byref(d), C.byref(f), s["x"].data_ptr(), pk.wpd.data_ptr(), s["dx"].data_ptr(), acc, dwp.data_ptr(), st)
    def dzp(i):
        s = sets[i % NBUF]
        call("plyolo_bn_act_bwd_dz", BF16, M, Cc, s["dout"].data_ptr(), Cc, s["z"].data_ptr(), Cc, coef.data_ptr(), bslots.data_ptr(), gamma.data_ptr(),
             dg.data_ptr(), db.data_ptr(), 0, a, s["dz"].data_ptr(), Cc, None, None, st)
    def dgr(i):
        s = sets[i % NBUF]
        call("plyolo_conv2d_dgrad", C.byref(d), s["dz"].data_ptr(), pk.wpd.data_ptr(), s["dx"].data_ptr(), acc, st)
    def wgr(i):
        s = sets[i % NBUF]
        call("plyolo_conv2d_wgrad", C.byref(d), s["x"].data_ptr(), s["dz"].data_ptr(), pk.dwp.data_ptr(), st)
    def timeit(fn):
        for i in range(NBUF): fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps): fn(i)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    t1, tz, td, tw = timeit(one), timeit(dzp), timeit(dgr), timeit(wgr)
    mv = M * Cc * 2 * (4 + acc)
    print("%-8s acc %d  slabs %4d (%5.1f MB)  one launch %6.1f us = %.2f TB/s | dz %6.1f + dgrad %6.1f = %6.1f us, wgrad %6.1f us"
          % (name, acc, ns, ns * Cc * Cc * 4 / 1e6, t1, mv / t1 / 1e6, tz, td, tz + td, tw))
