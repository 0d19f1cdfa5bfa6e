#!/usr/bin/env python3
"""YOLOX loss kernels through the C ABI on the benchmark's shape (B=32, 640x640 -> A=8400, 30 GT/img, 80 classes), head
outputs as a freshly initialised head gives them (small box logits, obj / cls logits around the prior bias).  Run under
rocprofv3 --kernel-trace --stats for the per-kernel times:   python tools/bench_loss.py [reps]"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from pl_yolo_amd._lib import call
import hiputil as hu
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B, C_, size, G, M = 32, 80, 640, 30, 100
gen = torch.Generator().manual_seed(1234)
sizes = [(size // s, size // s) for s in (8, 16, 32)]
maps = []
for (h, w) in sizes:
    m = torch.randn(B, 5 + C_, h, w, generator=gen) * 0.3
    m[:, 4:] -= 4.6
    maps.append(m)
lab = torch.zeros(B, M, 5)
lab[:, :G, 0] = torch.randint(0, C_, (B, G), generator=gen).float()
lab[:, :G, 1:3] = (0.15 + 0.7 * torch.rand(B, G, 2, generator=gen)) * size
lab[:, :G, 3:5] = 8 + torch.rand(B, G, 2, generator=gen) * 0.3 * size
d, rows = hu.yolox_desc(B, C_, M, sizes, [8, 16, 32])
raw = hu.maps_to_raw([m.to(hu.DEV) for m in maps])
labd = lab.to(hu.DEV).contiguous()
BA = B * d.A
fg = torch.zeros(BA, dtype=torch.uint8, device=hu.DEV); mgt = torch.zeros(BA, dtype=torch.int32, device=hu.DEV)
miou = torch.zeros(BA, device=hu.DEV); losses = torch.zeros(8, device=hu.DEV)
wsb = hu._lib.lib().plyolo_yolox_workspace(C.byref(d)); ws = torch.zeros(wsb, dtype=torch.uint8, device=hu.DEV)
cls_ld = 80
dro = torch.zeros(rows, 16, dtype=torch.bfloat16, device=hu.DEV); dcl = torch.zeros(rows, cls_ld, dtype=torch.bfloat16, device=hu.DEV)
st = hu.stream()
def fwd():
    call("plyolo_yolox_loss_fwd", C.byref(d), raw.data_ptr(), labd.data_ptr(), fg.data_ptr(), mgt.data_ptr(), miou.data_ptr(), losses.data_ptr(), ws.data_ptr(), wsb, st)
def bwd():
    call("plyolo_yolox_loss_bwd", C.byref(d), raw.data_ptr(), labd.data_ptr(), fg.data_ptr(), mgt.data_ptr(), miou.data_ptr(), losses.data_ptr(), None, None, dro.data_ptr(), dcl.data_ptr(), cls_ld, st)
for _ in range(3): fwd(); bwd()
torch.cuda.synchronize()
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
e[0].record()
for _ in range(reps): fwd()
e[1].record()
for _ in range(reps): bwd()
e[2].record(); torch.cuda.synchronize()
print("loss fwd %.1f us   bwd %.1f us   (num_fg %d, candidates %.1f %%)" % (e[0].elapsed_time(e[1]) / reps * 1e3, e[1].elapsed_time(e[2]) / reps * 1e3,
      int(losses[4]), 100.0 * float(torch.frombuffer(ws.cpu().numpy(), dtype=torch.uint8)[:0].numel() or 0)))
