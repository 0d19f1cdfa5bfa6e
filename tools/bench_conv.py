#!/usr/bin/env python3
"""Convolution micro-benchmark through the C ABI (GPU box): fwd / dgrad / wgrad of the
YOLOX-s layer shapes at batch 32, hipEvent-timed.   python tools/bench_conv.py [filter]"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from pl_yolo_amd._lib import BF16, call
import hiputil as hu
# (name, N, H, W, Cin, Cout, k, s)
LAYERS = [
    ("head3x3_80", 32, 80, 80, 128, 128, 3, 1),
    ("head3x3_40", 32, 40, 40, 128, 128, 3, 1),
    ("head3x3_20", 32, 20, 20, 128, 128, 3, 1),
    ("bb3x3_64_80", 32, 80, 80, 64, 64, 3, 1),
    ("bb3x3_32_160", 32, 160, 160, 32, 32, 3, 1),
    ("bb3x3_256_20", 32, 20, 20, 256, 256, 3, 1),
    ("stem", 32, 320, 320, 16, 32, 3, 1),
    ("s2_32_64", 32, 320, 320, 32, 64, 3, 2),
    ("s2_64_128", 32, 160, 160, 64, 128, 3, 2),
    ("s2_128_256", 32, 80, 80, 128, 256, 3, 2),
    ("s2_256_512", 32, 40, 40, 256, 512, 3, 2),
    ("pw_128_128_80", 32, 80, 80, 128, 128, 1, 1),
    ("pw_64_64_80", 32, 80, 80, 64, 64, 1, 1),
    ("pw_64_32_160", 32, 160, 160, 64, 32, 1, 1),
    ("pw_256_128_40", 32, 40, 40, 256, 128, 1, 1),
    ("pw_512_512_20", 32, 20, 20, 512, 512, 1, 1),
    ("pw_1024_512_20", 32, 20, 20, 1024, 512, 1, 1),
]
flt = sys.argv[1] if len(sys.argv) > 1 else ""
ENVS = [dict(kv.split("=") for kv in e.split(",")) if e else {} for e in (sys.argv[3].split(";") if len(sys.argv) > 3 else [""])]
what = sys.argv[2].split(",") if len(sys.argv) > 2 else ["fwd", "dgrad", "wgrad"]
reps = 20
def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print("%-16s %8s %22s %22s %22s" % ("layer", "GFLOP", "fwd us (TF/s)", "dgrad us (TF/s)", "wgrad us (TF/s)"))
for env in ENVS:
  for k_ in [k for k in os.environ if k.startswith("PLYOLO_")]: del os.environ[k_]
  os.environ.update(env)
  if env: print("ENV", env)
  for (name, N, H, W, Cin, Cout, k, s) in LAYERS:
    if flt and not any(f in name for f in flt.split("|")): continue
    pad = (k - 1) // 2
    OH, OW = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    x = torch.randn(N * H * W, Cin, device="cuda").to(torch.bfloat16)
    w = torch.randn(Cout, Cin, k, k, device="cuda") / (Cin * k * k) ** 0.5
    y = torch.empty(N * OH * OW, Cout, dtype=torch.bfloat16, device="cuda")
    dy = torch.randn(N * OH * OW, Cout, device="cuda").to(torch.bfloat16)
    dx = torch.empty(N * H * W, Cin, dtype=torch.bfloat16, device="cuda")
    d = hu.conv_desc(BF16, N, H, W, Cin, Cout, k, s, Cin, Cout)
    pk = hu.Packed(w, BF16)
    pk.set_slabs(d)
    stats = torch.zeros(hu._lib.STAT_SLOTS * 2 * Cout, dtype=torch.float64, device="cuda")
    st = hu.stream()
    gf = 2.0 * N * OH * OW * Cout * Cin * k * k / 1e9
    res = []
    if "fwd" in what:
          t = timeit(lambda: call("plyolo_conv2d_fwd", C.byref(d), x.data_ptr(), pk.wp.data_ptr(), None, y.data_ptr(), stats.data_ptr(), st)); res.append(t)
    else: res.append(float("nan"))
    if "dgrad" in what:
          t = timeit(lambda: call("plyolo_conv2d_dgrad", C.byref(d), dy.data_ptr(), pk.wpd.data_ptr(), dx.data_ptr(), 0, st)); res.append(t)
    else: res.append(float("nan"))
    if "wgrad" in what:
          t = timeit(lambda: call("plyolo_conv2d_wgrad", C.byref(d), x.data_ptr(), dy.data_ptr(), pk.dwp.data_ptr(), st)); res.append(t)
          tu = timeit(lambda: pk.unpack()); print("    unpack %.1f us (%d slabs)" % (tu, pk.entry.nslab))
    else: res.append(float("nan"))
    print("%-16s %8.1f %s" % (name, gf, " ".join("%12.1f (%7.1f)" % (t, gf / t * 1e3) for t in res)))
