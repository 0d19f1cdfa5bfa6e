// bf16 MFMA 3x3 stride-1 convolution, weights-stationary (forward and stride-1 data gradient) for gfx950.
//
// Replaces, for the layers it accepts, conv_mfma_kernel (conv_mfma_body.h) behind plyolo_conv2d_fwd / plyolo_conv2d_dgrad
// (nn.Conv2d inside BaseConv, reference models/layers/network_blocks.py:18-26).
//
// What conv_mfma_kernel pays for (measured): every 8x16-pixel workgroup streams ALL weights of its
// output channels from L2 into registers (295 KB per 128 pixels at 128->128: 460 MB of L2->L1 traffic per launch at 80x80,
// a third of the kernel's time), starts with an exposed halo load, ends with an LDS-staged epilogue, and on 20x20 / 40x40
// maps a launch has 100-400 short workgroups for 256 CUs.  Here
//   * a workgroup is PERSISTENT and owns one block of 32 output channels: its weights (9 taps x Cin x 32, <= 72 KB for
//     Cin <= 128) are copied into LDS once and stay; launches have (Cout / 32) x up to (256 / (Cout / 32)) workgroups;
//   * each of the four waves is an independent worker over its own 8x16-pixel tiles (strided through the batch): it stages
//     its own 10x18 halo, 16 input channels at a time, into a private double buffer -- the next chunk is written to LDS and
//     the one after is requested from HBM inside the MFMA phase, vector by vector (raw buffer loads: padding and "no
//     such tile" come back as zeros from the range check, no branches) -- so the main loop has NO barrier at all;
//   * the operands are swapped: A = weights (rows = output channels), B = pixels; a lane's accumulators are then 4 x 4
//     CONSECUTIVE channels of one pixel: 8-byte stores straight from registers, no LDS transpose;
//   * BatchNorm statistics accumulate in registers over all tiles of the wave and are reduced across lanes once per kernel.
// LDS image of a halo row: 18 pixels x 32 B, row pitch 784 B (== 16 mod 256): the two image rows a 32-pixel fragment spans
// fall on disjoint banks for ds_read_b128 (checked per 16-lane group), weights are read linearly (lane * 16 B).
#include <stdlib.h>

#include "common.h"

namespace {

struct WsP {
  const bf16_t* x;
  const bf16_t* w;      // fragment-ordered pack [tap][nnb][nkb][64 lanes][8] (conv_mfma_pack_elems)
  bf16_t* y;
  double* stats;        // fp64 stat slots [PLYOLO_STAT_SLOTS][2][Cout] or NULL
  int N, H, W, Cin, Cout, x_ld, y_ld;
  int nkb, nnb;
  int tiles_y, tiles_x, nsub;   // 8x16-pixel tiles per image column / row, in the batch
  int NB, GS;                   // 32-channel blocks; workgroups per block (each with 4 wave-workers)
  int accumulate;
  int x_bytes;                  // extent of x for the buffer descriptor
  unsigned long long taps_lo;   // 8 bits per tap: dy | dx << 2 | weight index << 4 (ConvP::taps_lo / taps_hi)
  unsigned int taps_hi;
};

constexpr int ROWP = 784, SUB = 10 * ROWP, WAVE_LDS = 2 * SUB;          // two halo buffers per wave
constexpr int NV = 6;                                                   // 360 16-byte vectors per halo chunk / 64 lanes

DEVINL unsigned tap_of(const WsP& p, int t) { return t < 8 ? (unsigned)((p.taps_lo >> (8 * t)) & 0xffull) : (p.taps_hi & 0xffu); }

// ABL (diagnostics, PLYOLO_CONV3WS_ABL): 1 no MFMA, 2 no halo requests after the prologue, 4 no epilogue, 8 no LDS fragment reads
template <int ABL>
__global__ __launch_bounds__(256) void conv3ws_kernel(const WsP p) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wg = (int)blockIdx.x;
  const int nb = wg % p.NB, gslot = wg / p.NB;
  const int cout0 = nb * 32;
  const int nkb = p.nkb;
  const int wbytes = 9 * nkb * 1024;

  // ---- the weights of this channel block -> LDS [tap][k-block][1 KiB], once
  for (int v = tid; v < 9 * nkb * 64; v += 256) {
    const int t = v / (nkb * 64), rem = v - t * nkb * 64;
    *(u32x4*)(smem + (size_t)v * 16) = *(const u32x4*)(p.w + ((size_t)(t * p.nnb + nb) * nkb) * 512 + (size_t)rem * 8);
  }
  __syncthreads();

  unsigned char* hb = smem + wbytes + wave * WAVE_LDS;
  // ---- tile-invariant staging descriptors of this lane's six vectors (pixel iy, ix of the 10x18 halo; channel half)
  constexpr unsigned NEVER = 0x7fffu;
  int rel[NV], lds[NV];
  unsigned yx[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int idx = lane + v * 64, pix = idx >> 1, half = idx & 1, iy = pix / 18, ix = pix - iy * 18;
    const bool live = idx < 360;
    rel[v] = ((iy * p.W + ix) * p.x_ld + half * 8) * 2;
    yx[v] = live ? (unsigned)iy | ((unsigned)ix << 16) : NEVER;
    // the 24 idle lanes of the sixth vector write into the unused tail of rows 0 / 1 (576..784 of each 784-byte row)
    const int idle = idx - 360;
    lds[v] = live ? iy * ROWP + ix * 32 + half * 16 : (idle / 13) * ROWP + 576 + (idle % 13) * 16;
  }
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  // fragment read offsets: B operand lane (j = pixel of the 2x16 fragment, kg = channel half); A operand linear
  const int j = lane & 31, kg = lane >> 5;
  const int boff = (j >> 4) * ROWP + (j & 15) * 32 + kg * 16;
  const unsigned char* wl = smem + lane * 16;

  struct Pos { int sub, ck; };
  const int stride = p.GS * 4;
  auto advance = [&](Pos a) { Pos b; b.ck = a.ck + 1 < nkb ? a.ck + 1 : 0; b.sub = a.ck + 1 < nkb ? a.sub : a.sub + stride; return b; };
  u32x4 R[NV];
  // Byte offsets of this lane's six vectors for one tile (chunk 0), computed when the pipeline moves to a new tile, NOT per
  // request: out-of-image pixels and tiles past the batch get an offset beyond the descriptor's range and read as zeros.
  // (A select between a computed offset and the marker next to each load became a branch around the offset arithmetic:
  // 200 scalar instructions per 36 MFMAs and a basic-block boundary per vector -- the wave could not even issue fast enough.)
  auto tile_offsets = [&](int q, int (&vb)[NV]) {
    const bool real = q < p.nsub;
    const int sub = real ? q : 0;
    const int tx = sub % p.tiles_x, t2 = sub / p.tiles_x, ty = t2 % p.tiles_y, n = t2 / p.tiles_y;
    const int iy0 = ty * 8 - 1, ix0 = tx * 16 - 1;
    const int base = ((n * p.H + iy0) * p.W + ix0) * p.x_ld * 2;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int y = iy0 + (int)(yx[v] & 0x7fffu), x = ix0 + (int)(yx[v] >> 16);
      const bool ok = real && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
      vb[v] = ok ? base + rel[v] : (int)0x80000000;
    }
  };
  auto request = [&](const int (&vb)[NV], int ck, int v) { R[v] = __builtin_amdgcn_raw_buffer_load_b128(rx, vb[v], ck * 32, 0); };
  auto stage = [&](int buf, int v) { *(u32x4*)(hb + buf * SUB + lds[v]) = R[v]; };
  // per-tap constants: halo offset of the tap shift, LDS offset of the tap's weight fragments
  int tap_b[9], tap_w[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const unsigned tc = tap_of(p, t);
    tap_b[t] = (int)(tc & 3u) * ROWP + (int)((tc >> 2) & 3u) * 32;
    tap_w[t] = (int)(tc >> 4) * nkb * 1024;
  }

  f32x16 acc[4];
  float s1[16], s2[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) s1[k] = s2[k] = 0.f;

  Pos c0; c0.sub = gslot * 4 + wave; c0.ck = 0;
  Pos c1 = advance(c0), c2 = advance(c1);
  int vb2[NV];
  {
    int vb0[NV];
    tile_offsets(c0.sub, vb0);
#pragma unroll
    for (int v = 0; v < NV; ++v) request(vb0, c0.ck, v);
#pragma unroll
    for (int v = 0; v < NV; ++v) stage(0, v);
    if (c1.sub != c0.sub) tile_offsets(c1.sub, vb0);
#pragma unroll
    for (int v = 0; v < NV; ++v) request(vb0, c1.ck, v);
    tile_offsets(c2.sub, vb2);
  }

  bf16x8 a0, a1, b0[4], b1[4];
  auto ldfrag = [&](int buf, int ck, int t, bf16x8& a, bf16x8 (&b)[4]) {
    a = *(const bf16x8*)(wl + tap_w[t] + ck * 1024);
    const unsigned char* bp = hb + buf * SUB + tap_b[t] + boff;
#pragma unroll
    for (int f = 0; f < 4; ++f) b[f] = *(const bf16x8*)(bp + 2 * f * ROWP);
  };
  auto mm = [&](const bf16x8& a, const bf16x8 (&b)[4]) {
    if (ABL & 1) { acc[0][0] += (float)a[0] + (float)b[0][0] + (float)b[1][0] + (float)b[2][0] + (float)b[3][0]; return; }
#pragma unroll
    for (int f = 0; f < 4; ++f) acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[f], acc[f], 0, 0, 0);
  };

  int buf = 0;
  ldfrag(0, 0, 0, a0, b0);
  while (c0.sub < p.nsub) {
    if (c0.ck == 0) {
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[f][i] = 0.f;
    }
    // ---- nine taps; beside tap t: the fragments of tap t+1 (after the last tap: tap 0 of the next chunk, which was staged
    // into the other buffer during taps 0-5), vector t of the next chunk staged, vector t of the chunk after requested
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      bf16x8& an = (t & 1) ? a0 : a1;
      bf16x8(&bn)[4] = (t & 1) ? b0 : b1;
      if (!(ABL & 8)) {
        if (t < 8) ldfrag(buf, c0.ck, t + 1, an, bn);
        else ldfrag(buf ^ 1, c1.ck, 0, an, bn);
      }
      mm((t & 1) ? a1 : a0, (t & 1) ? b1 : b0);
      if (t < NV) {
        if (!(ABL & 64)) stage(buf ^ 1, t);
        if (!(ABL & 2)) request(vb2, c2.ck, t);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // tap 8 used set 0 and loaded the next chunk's tap 0 into set 1: hand it over
    a0 = a1;
#pragma unroll
    for (int f = 0; f < 4; ++f) b0[f] = b1[f];

    if (c0.ck == nkb - 1 && !(ABL & 4)) {
      // ---- epilogue of this tile: lane (pixel j of fragment f, channel quarter h) holds channels 8q + 4h + (0..3), q = 0..3
      const int sub = c0.sub;
      const int tx = sub % p.tiles_x, t2 = sub / p.tiles_x, ty = t2 % p.tiles_y, n = t2 / p.tiles_y;
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const int oy = ty * 8 + 2 * f + (j >> 4), ox = tx * 16 + (j & 15);
        const bool valid = oy < p.H && ox < p.W;
        bf16_t* row = p.y + ((size_t)(n * p.H + oy) * p.W + ox) * p.y_ld + cout0 + 4 * kg;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = cout0 + 8 * q + 4 * kg;
          float v4[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) v4[i] = acc[f][4 * q + i];
          if (p.stats && !(ABL & 32)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float vv = valid ? v4[i] : 0.f;
              s1[4 * q + i] += vv;
              s2[4 * q + i] = fmaf(vv, vv, s2[4 * q + i]);
            }
          }
          if (valid && c < p.Cout && (!(ABL & 16) || v4[0] == 123.456f)) {
            unsigned lo = pack2bf(v4[0], v4[1]), hi = pack2bf(v4[2], v4[3]);
            unsigned* dst = (unsigned*)(row + 8 * q);
            if (p.accumulate) {
              const unsigned o0 = dst[0], o1 = dst[1];
              lo = pack2bf(__uint_as_float(o0 << 16) + __uint_as_float(lo << 16), __uint_as_float(o0 & 0xffff0000u) + __uint_as_float(lo & 0xffff0000u));
              hi = pack2bf(__uint_as_float(o1 << 16) + __uint_as_float(hi << 16), __uint_as_float(o1 & 0xffff0000u) + __uint_as_float(hi & 0xffff0000u));
            }
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            u32x2 pk = {lo, hi};
            *(u32x2*)dst = pk;
          }
        }
      }
    }
    c0 = c1; c1 = c2; c2 = advance(c2);
    if (c2.sub != c1.sub) tile_offsets(c2.sub, vb2);
    buf ^= 1;
  }

  if (p.stats) {
    // ---- per-channel sums: over the 32 pixels of a half-wave (xor shuffles stay inside lanes 0-31 / 32-63), then over the waves
#pragma unroll
    for (int k = 0; k < 16; ++k) {
#pragma unroll
      for (int m = 1; m < 32; m <<= 1) {
        s1[k] += __shfl_xor(s1[k], m);
        s2[k] += __shfl_xor(s2[k], m);
      }
    }
    __syncthreads();                       // every wave is out of its loop: the halo buffers are free
    float* red = (float*)(smem + wbytes);  // [4 waves][2][32 channels]
    if (j == 0) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int ch = 8 * (k >> 2) + 4 * kg + (k & 3);
        red[(wave * 2 + 0) * 32 + ch] = s1[k];
        red[(wave * 2 + 1) * 32 + ch] = s2[k];
      }
    }
    __syncthreads();
    if (tid < 64) {
      const int which = tid >> 5, ch = tid & 31, co = cout0 + ch;
      const float s = red[(0 * 2 + which) * 32 + ch] + red[(1 * 2 + which) * 32 + ch] + red[(2 * 2 + which) * 32 + ch] + red[(3 * 2 + which) * 32 + ch];
      if (co < p.Cout) {
        double* slot = p.stats + (size_t)(wg % PLYOLO_STAT_SLOTS) * 2 * p.Cout;
        __hip_atomic_fetch_add(slot + (size_t)which * p.Cout + co, (double)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

}  // namespace

namespace plyolo {

// 1 if conv3ws takes this convolution (3x3, stride 1, pad 1, bf16 in / out, no bias / fused epilogue / lazy input), else 0
int conv3ws_accepts(int N, int H, int W, int Cin, int Cout, int x_ld, int y_ld, const void* y) {
  // OPT-IN (PLYOLO_CONV3WS=1, read per call): correct on every test shape, but measured slower than conv_mfma_kernel
  // (3x3 128->128 @80x80 B=32: 87 vs 73 us forward) -- see the note at the end of this file and DESIGN.md section 8
  const char* e = getenv("PLYOLO_CONV3WS");
  if (!e || atoi(e) == 0) return 0;
  if (Cin % 16 != 0 || Cin > 128 || Cout % 4 != 0 || y_ld % 4 != 0 || ((size_t)y & 7) != 0) return 0;
  if (((double)N * H * W - 1.0) * x_ld * 2.0 + 256.0 >= 2.0e9) return 0;     // one 31-bit buffer descriptor over x
  if ((Cout + 31) / 32 > 256) return 0;
  return 1;
}

hipError_t conv3ws_launch(const void* x, const void* w, void* y, double* stats, int N, int H, int W, int Cin, int Cout, int x_ld, int y_ld,
                          int nkb, int nnb, int accumulate, unsigned long long taps_lo, unsigned taps_hi, hipStream_t s) {
  WsP p{};
  p.x = (const bf16_t*)x; p.w = (const bf16_t*)w; p.y = (bf16_t*)y; p.stats = stats;
  p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.x_ld = x_ld; p.y_ld = y_ld;
  p.nkb = nkb; p.nnb = nnb;
  p.tiles_y = (H + 7) / 8; p.tiles_x = (W + 15) / 16;
  p.nsub = N * p.tiles_y * p.tiles_x;
  p.NB = (Cout + 31) / 32;
  int gs = 256 / p.NB;
  if (gs < 1) gs = 1;
  const int groups = (p.nsub + 3) / 4;
  if (gs > groups) gs = groups;
  static const int gs_env = getenv("PLYOLO_CONV3WS_GS") ? atoi(getenv("PLYOLO_CONV3WS_GS")) : 0;
  if (gs_env > 0 && gs_env < gs) gs = gs_env;
  p.GS = gs;
  p.accumulate = accumulate;
  p.x_bytes = (int)(((size_t)N * H * W - 1) * x_ld * 2 + (size_t)((Cin + 7) & ~7) * 2);
  p.taps_lo = taps_lo; p.taps_hi = taps_hi;
  const size_t lds = (size_t)9 * nkb * 1024 + 4 * (size_t)WAVE_LDS;
  static const int abl = getenv("PLYOLO_CONV3WS_ABL") ? atoi(getenv("PLYOLO_CONV3WS_ABL")) : 0;
  auto kern = conv3ws_kernel<0>;
#ifdef PLYOLO_DIAG_ABLATE
  if (abl == 1) kern = conv3ws_kernel<1>;
  if (abl == 2) kern = conv3ws_kernel<2>;
  if (abl == 4) kern = conv3ws_kernel<4>;
  if (abl == 8) kern = conv3ws_kernel<8>;
  if (abl == 9) kern = conv3ws_kernel<9>;
  if (abl == 6) kern = conv3ws_kernel<6>;
  if (abl == 15) kern = conv3ws_kernel<15>;
  if (abl == 16) kern = conv3ws_kernel<16>;
  if (abl == 48) kern = conv3ws_kernel<48>;
  if (abl == 18) kern = conv3ws_kernel<18>;
  if (abl == 50) kern = conv3ws_kernel<50>;
  if (abl == 58) kern = conv3ws_kernel<58>;
  if (abl == 122) kern = conv3ws_kernel<122>;
#endif
  (void)abl;
  if (hipError_t e = ensure_dynamic_lds((const void*)kern, 160 * 1024); e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(p.NB * p.GS), dim3(256), lds, s, p);
  return hipGetLastError();
}

}  // namespace plyolo

// Measured (MI355X, 3x3 128->128 @80x80, B=32, 60.4 GFLOP, stand-alone, operands warm in the 256 MB Infinity Cache;
// `make DIAG=1` + PLYOLO_CONV3WS_ABL):
//   conv_mfma_kernel                                            73 us forward / 83 us data gradient
//   this kernel                                                 87 / 98 us
//   MFMAs only (no LDS reads, no staging, no loads, no stores)  49 us = 1.24 PFLOP/s: what one wave per SIMD sustains on
//                                                               256 CUs with back-to-back v_mfma_f32_32x32x16_bf16
//   + LDS fragment reads and staging writes                     67 us (45 ds_read_b128 + 6 ds_write_b128 per 36 MFMAs and wave:
//                                                               ~80 % of the LDS array at the MFMA-only rate)
//   + halo requests                                             72 us (295 MB of L2 -> LDS per launch -- every 32-channel block
//                                                               re-reads the halo -- at ~8.4 TB/s when run alone)
//   + 8-byte stores                                             76 us; + BatchNorm statistics 87 us
// The weights are gone from the L2 stream (545 -> 295 MB per launch), but 32-channel blocks need 1.25 LDS reads per MFMA
// plus the staging writes, and the wave cannot hide them: the kernel is LDS-bound where conv_mfma_kernel, with its weight
// fragments from L1/L2 and 3 waves per SIMD, is not.  A 64-channel block (0.75 reads per MFMA, half the halo traffic) needs
// 144 KB of weights at Cin = 128 and does not fit beside the halo buffers.
