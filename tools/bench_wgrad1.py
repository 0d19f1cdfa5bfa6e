#!/usr/bin/env python3
"""1x1 weight gradient micro-benchmark (GPU box): dW[Cout,Cin] = dz^T[Cout,M] . x[M,Cin] -- a plain deep-K GEMM.  plyolo_conv2d_wgrad
(+ its slab fold) against the library GEMM torch.mm dispatches (hipBLASLt / rocBLAS), on the pointwise shapes of YOLOX-x at 1280x1280
batch 16, YOLOX-l / YOLOv7 and YOLOX-s.   python tools/bench_wgrad1.py [env1;env2]"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from pl_yolo_amd._lib import BF16, call
import hiputil as hu
# (M pixels as N, H, W; Cin, Cout)
SHAPES = [(16, 160, 160, 160, 160), (16, 160, 160, 320, 320), (16, 80, 80, 320, 320), (16, 80, 80, 640, 640), (16, 40, 40, 640, 640),
          (16, 40, 40, 1280, 1280), (16, 40, 40, 2560, 1280), (16, 40, 40, 512, 512), (16, 20, 20, 1024, 1024), (32, 80, 80, 256, 256),
          (32, 80, 80, 128, 128), (32, 40, 40, 256, 256), (32, 20, 20, 512, 512), (32, 40, 40, 1024, 512)]
ENVS = [dict(kv.split("=") for kv in e.split(",")) if e else {} for e in (sys.argv[1].split(";") if len(sys.argv) > 1 else [""])]
reps = 10
def timeit(fn):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for env in ENVS:
    for k_ in [k for k in os.environ if k.startswith("PLYOLO_")]: del os.environ[k_]
    os.environ.update(env)
    print("ENV", env)
    print("%-28s %7s %26s %14s %22s %22s" % ("M x Cin -> Cout", "GFLOP", "plyolo wgrad us (TF/s)", "fold us/slabs", "torch.mm bf16 us (TF/s)", "torch.mm f32out (TF/s)"))
    for (N, H, W, Cin, Cout) in SHAPES:
        M = N * H * W
        x = torch.randn(M, Cin, device="cuda").to(torch.bfloat16)
        dy = torch.randn(M, Cout, device="cuda").to(torch.bfloat16)
        w = torch.randn(Cout, Cin, 1, 1, device="cuda")
        d = hu.conv_desc(BF16, N, H, W, Cin, Cout, 1, 1, Cin, Cout)
        pk = hu.Packed(w, BF16)
        pk.set_slabs(d)
        st = hu.stream()
        gf = 2.0 * M * Cout * Cin / 1e9
        t = timeit(lambda: call("plyolo_conv2d_wgrad", C.byref(d), x.data_ptr(), dy.data_ptr(), pk.dwp.data_ptr(), st))
        tu = timeit(lambda: pk.unpack())
        dyt = dy.t()
        tm = timeit(lambda: torch.mm(dyt, x))
        try:
            tf = timeit(lambda: torch.mm(dyt, x, out_dtype=torch.float32))
        except Exception as e:   # noqa
            tf = float("nan")
        print("%-28s %7.1f %14.1f (%7.1f) %8.1f /%4d %14.1f (%7.1f) %14.1f (%7.1f)" % ("%d x %d -> %d" % (M, Cin, Cout), gf, t, gf / t * 1e3, tu, pk.entry.nslab,
                                                                         tm, gf / tm * 1e3, tf, gf / tf * 1e3))
