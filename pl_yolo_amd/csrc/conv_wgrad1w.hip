// Weight gradient of a WIDE pointwise (1x1, stride 1) convolution: a deep-K GEMM on large output tiles (gfx950).
//
//     dW[Cout, Cin] = dz^T[Cout, M] . x[M, Cin]        (what ATen's convolution_backward computes for the weight of the 1x1 BaseConv
//                                                       layers, reference models/layers/network_blocks.py:18-26)
//
// with M = N*H*W pixels as the contraction index (10^4 .. 4*10^5) and 160 .. 2560 channels on both sides (yolox_l / yolox_x / yolov7:
// CSP and ELAN pointwise layers, SPP bottlenecks).  Both operands are pixel-major in HBM, so every MFMA fragment (8 consecutive k per
// lane) is transposed on its way out of LDS (ds_read_b64_tr_b16), and the LDS read port is the scarce resource: a wave that owns an
// (a x b)-fragment block of the output reads a + b fragments per a*b MFMAs.  The 128 x 128 slab of conv_wgrad_mfma.hip gives each of
// its four waves a 2 x 2 block -- one fragment read per v_mfma_f32_32x32x16_bf16, the LDS port saturated at half the matrix rate, and
// every operand tile re-read from L2 by Cout/128 (Cin/128) workgroups: 160 .. 340 TFLOP/s on these layers, 0.10 of the peak.  Here:
//   * one workgroup = a 256 x 256 (eight waves, 4 x 2 fragments each: 0.75 reads per MFMA, 128 accumulator registers) or a 192 x 192
//     (six waves, 2 x 3 fragments: yolox_x's 160 / 320 / 640-channel layers) tile of dW, ONE workgroup per CU, persistent over the
//     32-pixel stages of its pixel range; a stage is requested four stages ahead (four register sets), written into the other of two LDS
//     images while the current one is multiplied, ONE barrier per stage;
//   * fragments of k-step j+1 are read while the MFMAs of k-step j run (two waves per SIMD cover each other's LDS latency);
//   * split-K over pixel ranges: every range writes a private fp32 slab tile [Cout][Cin] (no atomics), folded in a fixed order by
//     plyolo_reduce_slabs like every other weight gradient; the tiles of one range are neighbours on one XCD (its L2 serves the
//     second .. n-th reader of an operand stage);
//   * channel tails (160 = 5 x 32 in a 192 tile, 320 in two of them) are zero columns in LDS and masked stores; a pixel tail is zero rows.
#include <stdlib.h>

#include "common.h"

namespace {

struct W1wP {
  const bf16_t* dz;      // [M][dz_ld]
  const bf16_t* x;       // [M][x_ld]
  float* dw;             // slabs [S][Cout][Cin]
  int dz_ld, x_ld, M, Cout, Cin, Cout8;
  int nci, ntile;        // channel tiles along ci, all channel tiles (nco * nci)
  int S, stages_per;     // pixel ranges (= slabs), 32-pixel stages per range
};

typedef __attribute__((address_space(3))) s16x4 lds_s16x4_w;
DEVINL s16x4 tr_read_w(const unsigned char* p) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_w*)p); }

constexpr int W1W_BM = 32;     // pixels per stage (two 16-pixel k-steps)

// WCO x WCI waves, each a (32*MTC) x (32*MTI) block of the CO_T x CI_T tile
template <int WCO, int WCI, int MTC, int MTI>
__global__ __launch_bounds__(64 * WCO * WCI, 1) void conv_wgrad1w_kernel(const W1wP p) {
  constexpr int NT = 64 * WCO * WCI, BM = W1W_BM;
  constexpr int CO_T = WCO * MTC * 32, CI_T = WCI * MTI * 32;
  constexpr int PD = CO_T * 2 + 16, PX = CI_T * 2 + 16;            // LDS row pitches (bytes): 4 * pitch = 64 mod 256 -> conflict-free transposed reads
  constexpr int DV = CO_T / 8, XV = CI_T / 8;                      // 16-byte vectors per row
  constexpr int NDV = (BM * DV + NT - 1) / NT, NXV = (BM * XV + NT - 1) / NT;
  static_assert((BM * DV) % NT == 0 && (BM * XV) % NT == 0, "whole vectors per thread");
  constexpr int BUF = BM * (PD + PX);
  extern __shared__ __align__(16) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // workgroup -> (pixel range, channel tile): the tiles of one range adjacent, ranges dealt over the XCDs in contiguous runs
  int wg = (int)blockIdx.x;
  {
    const int nwg = (int)gridDim.x, q = nwg >> 3, r8 = nwg & 7, xc = wg & 7;
    wg = (xc < r8 ? xc * (q + 1) : r8 * (q + 1) + (xc - r8) * q) + (wg >> 3);
  }
  const int split = wg / p.ntile, tile = wg % p.ntile;
  const int co0 = (tile / p.nci) * CO_T, ci0 = (tile % p.nci) * CI_T;
  const int st0 = split * p.stages_per;
  const int nstage_all = (p.M + BM - 1) / BM;
  int nst = nstage_all - st0;
  nst = nst < p.stages_per ? nst : p.stages_per;
  if (nst < 0) nst = 0;

  // loader: thread -> (row, channel vector) of each operand; the channel vector never changes, rows advance by NT / vectors-per-row
  int dcv[NDV], drow[NDV], xcv[NXV], xrow[NXV];
  bool dok[NDV], xok[NXV];
#pragma unroll
  for (int v = 0; v < NDV; ++v) {
    const int idx = tid + v * NT;
    drow[v] = idx / DV; dcv[v] = idx % DV;
    dok[v] = co0 + dcv[v] * 8 < p.Cout8;
  }
#pragma unroll
  for (int v = 0; v < NXV; ++v) {
    const int idx = tid + v * NT;
    xrow[v] = idx / XV; xcv[v] = idx % XV;
    xok[v] = ci0 + xcv[v] * 8 < p.Cin;
  }
  struct Rows { u32x4 av[NDV], xv[NXV]; };
  // requests are unconditional (rows beyond M / stages beyond the range read row 0, channel tails read channel 0): a fixed number of
  // vector-memory instructions per call keeps the compiler's vmcnt bookkeeping exact across the two register sets
  auto request = [&](Rows& R, const int s) {
    const int m0 = (st0 + s) * BM;
    const bool live = s < nst;
#pragma unroll
    for (int v = 0; v < NDV; ++v) {
      const int m = m0 + drow[v];
      const size_t off = (live && m < p.M && dok[v]) ? (size_t)m * p.dz_ld + co0 + dcv[v] * 8 : 0;
      R.av[v] = *(const u32x4*)(p.dz + off);
    }
#pragma unroll
    for (int v = 0; v < NXV; ++v) {
      const int m = m0 + xrow[v];
      const size_t off = (live && m < p.M && xok[v]) ? (size_t)m * p.x_ld + ci0 + xcv[v] * 8 : 0;
      R.xv[v] = *(const u32x4*)(p.x + off);
    }
  };
  auto commit = [&](const Rows& R, const int s, unsigned char* buf) {
    const int m0 = (st0 + s) * BM;
    const u32x4 zero = {0u, 0u, 0u, 0u};
    unsigned char* dz_s = buf;
    unsigned char* x_s = buf + BM * PD;
#pragma unroll
    for (int v = 0; v < NDV; ++v) *(u32x4*)(dz_s + drow[v] * PD + dcv[v] * 16) = (dok[v] && m0 + drow[v] < p.M) ? R.av[v] : zero;
#pragma unroll
    for (int v = 0; v < NXV; ++v) *(u32x4*)(x_s + xrow[v] * PX + xcv[v] * 16) = (xok[v] && m0 + xrow[v] < p.M) ? R.xv[v] : zero;
  };

  const int wco = wave / WCI, wci = wave % WCI;
  f32x16 acc[MTC][MTI];
#pragma unroll
  for (int a = 0; a < MTC; ++a)
#pragma unroll
    for (int b = 0; b < MTI; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  // transposed-read addresses (as conv_pw_bwd.hip / conv_wgrad1.hip): lane group g4 of a 16-pixel k-step reads physical rows trow,
  // trow + 1; A and B use the same pixel -> k permutation, a sum over pixels does not care about its order
  const int g4 = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, pp = i16 & 3;
  const int trow = 4 * q4 + 2 * (g4 >> 1);
  const int a_off = trow * PD + (wco * MTC * 32 + 16 * (g4 & 1) + 4 * pp) * 2;
  const int b_off = BM * PD + trow * PX + (wci * MTI * 32 + 16 * (g4 & 1) + 4 * pp) * 2;

  auto ldfrag = [&](const unsigned char* buf, const int j, s16x8 (&af)[MTC], s16x8 (&bfr)[MTI]) {
#pragma unroll
    for (int a = 0; a < MTC; ++a) {
      const unsigned char* ap = buf + j * 16 * PD + a_off + a * 64;
      const s16x4 lo = tr_read_w(ap), hi = tr_read_w(ap + PD);
      af[a] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
#pragma unroll
    for (int b = 0; b < MTI; ++b) {
      const unsigned char* bp = buf + j * 16 * PX + b_off + b * 64;
      const s16x4 lo = tr_read_w(bp), hi = tr_read_w(bp + PX);
      bfr[b] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
  };
  auto mm = [&](const s16x8 (&af)[MTC], const s16x8 (&bfr)[MTI]) {
#pragma unroll
    for (int a = 0; a < MTC; ++a)
#pragma unroll
      for (int b = 0; b < MTI; ++b)
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)&af[a], *(const bf16x8*)&bfr[b], acc[a][b], 0, 0, 0);
  };

  // NSET register sets = NSET stages in flight per workgroup: with one workgroup per CU the request depth IS the memory-level
  // parallelism of the CU (two sets, 64 KB in flight: 31 GB/s per CU at the ~2 us loaded latency -- the first build ran at that
  // number, 1.03 us per stage against 0.21 us of matrix-core time)
  constexpr int NSET = 4;
  Rows R[NSET];
#pragma unroll
  for (int k = 0; k < NSET; ++k) {
    request(R[k], k);
    __builtin_amdgcn_sched_barrier(0);
  }

  // stage s: its rows (requested NSET stages ago) -> LDS image s & 1; barrier; request stage s + NSET into the freed register set;
  // multiply image s & 1.  One barrier per stage: a wave can only write image (s + 1) & 1 behind barrier s, which every wave
  // reached after it had finished multiplying stage s - 1 out of that image.
  auto stage = [&](Rows& Rs, const int s) {
    unsigned char* buf = smem + (s & 1) * BUF;
    commit(Rs, s, buf);
    __syncthreads();
    request(Rs, s + NSET);
    __builtin_amdgcn_sched_barrier(0);
    s16x8 af0[MTC], bf0[MTI], af1[MTC], bf1[MTI];
    ldfrag(buf, 0, af0, bf0);
    ldfrag(buf, 1, af1, bf1);          // BM = 32: two k-steps; the second one's fragments land under the first one's MFMAs
    mm(af0, bf0);
    mm(af1, bf1);
  };
  int s = 0;
  for (; s + NSET <= nst; s += NSET) {
#pragma unroll
    for (int k = 0; k < NSET; ++k) stage(R[k], s + k);
  }
#pragma unroll
  for (int k = 0; k < NSET - 1; ++k)
    if (s + k < nst) stage(R[k], s + k);

  // private slab tile of this pixel range: D[row = co][col = ci], col = lane & 31, row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)
  float* slab = p.dw + (size_t)split * ((size_t)p.Cout * p.Cin);
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int b = 0; b < MTI; ++b) {
    const int ci = ci0 + (wci * MTI + b) * 32 + r;
    if (ci < p.Cin) {
#pragma unroll
      for (int a = 0; a < MTC; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int co = co0 + (wco * MTC + a) * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
          if (co < p.Cout) slab[(size_t)co * p.Cin + ci] = acc[a][b][i];
        }
    }
  }
}

template <int WCO, int WCI, int MTC, int MTI>
hipError_t launch_w1w(W1wP p, hipStream_t s) {
  constexpr int NT = 64 * WCO * WCI, CO_T = WCO * MTC * 32, CI_T = WCI * MTI * 32;
  constexpr size_t lds = 2 * (size_t)W1W_BM * ((CO_T * 2 + 16) + (CI_T * 2 + 16));
  p.nci = (p.Cin + CI_T - 1) / CI_T;
  p.ntile = ((p.Cout + CO_T - 1) / CO_T) * p.nci;
  auto kern = conv_wgrad1w_kernel<WCO, WCI, MTC, MTI>;
  if (hipError_t e = plyolo::ensure_dynamic_lds((const void*)kern, lds); e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(p.S * p.ntile), dim3(NT), lds, s, p);
  return hipGetLastError();
}

}  // namespace

namespace plyolo {

// tile edge (256 or 192) that wastes fewer matrix-core cycles on the channel tails of a Cout x Cin gradient, 0 = not a wide layer
int conv_wgrad1w_tile(int Cout, int Cin) {
  if (Cout < 160 || Cin < 160) return 0;
  auto waste = [&](int t) { return (double)(((Cout + t - 1) / t) * t) * (double)(((Cin + t - 1) / t) * t) / ((double)Cout * Cin); };
  return waste(192) < waste(256) * 0.95 ? 192 : 256;      // (ties and near-ties: the 256 tile reads less per MFMA)
}

// channel tiles of the launch (the caller plans S = pixel ranges = slabs with it)
int conv_wgrad1w_tiles(int Cout, int Cin) {
  const int t = conv_wgrad1w_tile(Cout, Cin);
  return t ? ((Cout + t - 1) / t) * ((Cin + t - 1) / t) : 0;
}

// dz [M][dz_ld], x [M][x_ld], S private slabs [S][Cout][Cin] at dw; S * 32-pixel stage ranges cover M
hipError_t conv_wgrad1w_launch(const void* x, const void* dz, float* dw, int M, int Cout, int Cin, int x_ld, int dz_ld, int S, hipStream_t s) {
  W1wP p{};
  p.dz = (const bf16_t*)dz; p.x = (const bf16_t*)x; p.dw = dw;
  p.dz_ld = dz_ld; p.x_ld = x_ld; p.M = M; p.Cout = Cout; p.Cin = Cin; p.Cout8 = (Cout + 7) & ~7;
  const int nstage = (M + W1W_BM - 1) / W1W_BM;
  if (S < 1 || S > nstage) return hipErrorInvalidValue;
  p.S = S;
  p.stages_per = (nstage + S - 1) / S;
  const int t = conv_wgrad1w_tile(Cout, Cin);
  if (t == 256) return launch_w1w<2, 4, 4, 2>(p, s);
  if (t == 192) return launch_w1w<3, 2, 2, 3>(p, s);
  return hipErrorInvalidValue;
}

}  // namespace plyolo
