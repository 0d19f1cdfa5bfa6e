"""Helpers for the -m gpu tests: drive the C ABI (include/plyolo.h) on torch device tensors."""
import ctypes as C

import numpy as np
import torch

from pl_yolo_amd import _lib
from pl_yolo_amd._lib import BF16, F32, ConvDesc, PackEntry, YoloxDesc, NmsDesc, call

DEV = "cuda:0"


def stream():
    return torch.cuda.current_stream().cuda_stream


def tdtype(dt):
    return torch.bfloat16 if dt == BF16 else torch.float32


def to_nhwc(x, dt, ld=None, fill=7.0):
    """NCHW fp32 -> [N*H*W, ld] activation matrix (junk in the pad columns)."""
    N, Cc, H, W = x.shape
    ld = ld or Cc
    out = torch.full((N * H * W, ld), fill, dtype=tdtype(dt), device=x.device)
    out[:, :Cc] = x.permute(0, 2, 3, 1).reshape(-1, Cc).to(tdtype(dt))
    return out


def from_nhwc(t, N, H, W, Cc):
    return t[:, :Cc].float().reshape(N, H, W, Cc).permute(0, 3, 1, 2).contiguous()


def rnd_bf16(x):
    return x.to(torch.bfloat16).float()


class Packed:
    """Pack one conv weight (+bias) through plyolo_pack_weights."""

    def __init__(self, w, dt, bias=None, cin_p=None, nslab=1):
        Cout, Cin, k, _ = w.shape
        self.w = w.contiguous()
        self.Cin_p = cin_p or Cin
        self.Cout_p8 = (Cout + 7) // 8 * 8
        taps = k * k
        dev = w.device
        n_wp, n_wpd = _lib.pack_elems(dt, Cout, self.Cin_p, k)
        self.wp = torch.zeros(n_wp, dtype=tdtype(dt), device=dev)
        self.wpd = torch.zeros(n_wpd, dtype=tdtype(dt), device=dev)
        self.nslab = nslab  # set_slabs(desc) sizes the wgrad slab buffer for a concrete launch
        self.slab_elems = taps * Cout * self.Cin_p
        self.dwp = torch.zeros(nslab * self.slab_elems, dtype=torch.float32, device=dev)
        self.dw = torch.zeros_like(self.w)
        self.bias = bias.contiguous() if bias is not None else None
        self.bp = torch.zeros(max(Cout, 8), dtype=torch.float32, device=dev)
        self.dbp = torch.zeros(max(Cout, 8), dtype=torch.float32, device=dev)
        self.db = torch.zeros(Cout, dtype=torch.float32, device=dev)
        e = PackEntry()
        e.w, e.wp, e.wpd, e.dwp, e.dw = self.w.data_ptr(), self.wp.data_ptr(), self.wpd.data_ptr(), self.dwp.data_ptr(), self.dw.data_ptr()
        e.b = self.bias.data_ptr() if bias is not None else None
        e.bp, e.dbp, e.db = self.bp.data_ptr(), self.dbp.data_ptr(), self.db.data_ptr() if bias is not None else None
        e.Cout, e.Cin, e.Cin_p, e.ksize = Cout, Cin, self.Cin_p, k
        e.Cout_total, e.Cout_p8, e.co_off, e.nslab = Cout, self.Cout_p8, 0, 1
        self.entry = e
        arr = (PackEntry * 1)(e)
        self.table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        self.nelem = taps * Cout * self.Cin_p
        call("plyolo_pack_weights", self.table.data_ptr(), 1, dt, self.nelem, stream())

    def set_slabs(self, desc):
        n = _lib.lib().plyolo_conv2d_wgrad_slabs(C.byref(desc))
        assert n > 0, n
        if n > self.nslab:
            self.nslab = n
            self.dwp = torch.zeros(n * self.slab_elems, dtype=torch.float32, device=self.w.device)
            self.entry.dwp = self.dwp.data_ptr()
        self.entry.nslab = n
        arr = (PackEntry * 1)(self.entry)
        self.table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.w.device)
        return n

    def unpack(self, accumulate=0):
        call("plyolo_unpack_wgrads", self.table.data_ptr(), 1, self.nelem, accumulate, stream())
        return self.dw


def conv_desc(dt, N, H, W, Cin, Cout, k, s, x_ld, y_ld, y_f32=0):
    d = ConvDesc()
    d.dtype, d.N, d.H, d.W, d.Cin, d.Cout, d.ksize, d.stride, d.x_ld, d.y_ld, d.y_f32 = dt, N, H, W, Cin, Cout, k, s, x_ld, y_ld, y_f32
    return d


def yolox_desc(B, C_, M, sizes, strides):
    d = YoloxDesc()
    a = r = 0
    d.B, d.C, d.M, d.nlevels = B, C_, M, len(sizes)
    for i, ((h, w), s) in enumerate(zip(sizes, strides)):
        d.lvl_h[i], d.lvl_w[i], d.lvl_stride[i], d.lvl_off[i], d.lvl_row[i] = h, w, int(s), a, r
        a += h * w
        r += B * h * w
    d.A = a
    return d, r


def maps_to_raw(maps):
    """list of [B, 5+C, h, w] -> level-major raw [rows, 5+C] fp32."""
    return torch.cat([m.permute(0, 2, 3, 1).reshape(-1, m.shape[1]) for m in maps], 0).contiguous().float()


def raw_to_maps(raw, B, sizes):
    outs, r = [], 0
    for (h, w) in sizes:
        n = B * h * w
        outs.append(raw[r:r + n].reshape(B, h, w, -1).permute(0, 3, 1, 2).contiguous())
        r += n
    return outs


def relerr(a, b):
    a, b = a.float(), b.float()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def cossim(a, b):
    a, b = a.flatten().double(), b.flatten().double()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


def relrms(a, b):
    a, b = a.float(), b.float()
    return float((a - b).pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-30))
