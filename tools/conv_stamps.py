#!/usr/bin/env python3
"""In-kernel phase timing of the main conv tile (diagnostic instantiation PLYOLO_ABLATE=512): clock64 stamps
of wave 0 of every workgroup: start | halo 0 ready | halo 1 ready | taps done | barrier | end."""
import ctypes as C, os, sys
os.environ["PLYOLO_ABLATE"] = "512"
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from pl_yolo_amd._lib import BF16, call
import hiputil as hu
N, H, W, Cin, Cout, k = 32, 80, 80, 128, 128, 3
x = torch.randn(N * H * W, Cin, device="cuda").to(torch.bfloat16)
w = torch.randn(Cout, Cin, k, k, device="cuda") / (Cin * k * k) ** 0.5
y = torch.empty(N * H * W, Cout, dtype=torch.bfloat16, device="cuda")
d = hu.conv_desc(BF16, N, H, W, Cin, Cout, k, 1, Cin, Cout)
pk = hu.Packed(w, BF16)
ntile = N * 5 * 5
dbg = torch.zeros(ntile * 8, dtype=torch.int64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    call("plyolo_conv2d_fwd", C.byref(d), x.data_ptr(), pk.wp.data_ptr(), dbg.data_ptr(), y.data_ptr(), None, st)
torch.cuda.synchronize()
s = dbg.cpu().numpy().reshape(ntile, 8).astype(np.float64)
t0 = s[:, 0].min()
names = ["halo0 (load+barriers)", "chunk0 taps + halo1", "chunk1 taps", "final barrier", "epilogue"]
ph = np.diff(s[:, :6], axis=1)
print("tiles", ntile, "kernel span (cycles of clock64): %.0f" % (s[:, 5].max() - t0))
for i, n in enumerate(names):
    print("  %-24s median %8.0f  p10 %8.0f  p90 %8.0f" % (n, np.median(ph[:, i]), np.percentile(ph[:, i], 10), np.percentile(ph[:, i], 90)))
tot = s[:, 5] - s[:, 0]
print("  per-tile total           median %8.0f" % np.median(tot))
start = np.sort(s[:, 0] - t0)
print("  tile start times: first 512 by %.0f, tile 513 at %.0f, last at %.0f" % (start[min(511, ntile - 1)], start[min(512, ntile - 1)], start[-1]))
