// Weight gradient of a pointwise (1x1, stride 1) convolution with 128+ channels on both sides, as a persistent streaming launch (gfx950).
//
//     dW[Cout, Cin] = dz^T[Cout, M] . x[M, Cin]        (what ATen's convolution_backward computes for the weight of the 1x1 BaseConv
//                                                       layers, reference models/layers/network_blocks.py:18-26)
//
// These launches move two 128-channel operands once and are HBM-bound by a factor of ~6 over their MFMA time; the phase-alternating
// round-1 kernel (conv_wgrad_mfma.hip: one tile in LDS, the next in registers, four waves) left them at 1.4 - 1.8 TB/s.  This is the
// weight-gradient half of conv_pw_bwd.hip on its own: eight waves per workgroup, one workgroup per CU, persistent over the pixel tiles
// of its row group; TWO register sets, so the rows of tiles t+G and t+2G are in flight while tile t is turned into its LDS image and
// multiplied (requests unconditional and in pairs: exact vector-memory wait counts, see conv_pw_bwd.hip); both operands pixel-major
// in LDS (row pitch C*2+16), fragments transposed on the way out with ds_read_b64_tr_b16 under the k-permutation that makes those
// reads conflict-free; the accumulators (128 x 128 fp32 = 32 registers per lane) leave once, as the workgroup's private slab tile.
// Grid: (row groups = slabs) x (128 x 128 channel tiles); channel tails are masked (zero columns in LDS, no store).
#include <stdlib.h>

#include "common.h"

namespace {

struct W1P {
  const bf16_t* dz;      // [M][dz_ld]
  const bf16_t* x;       // [M][x_ld]
  float* dw;             // slabs [G][Cout][Cin]
  int dz_ld, x_ld, M, ntiles, Cout, Cin, nci;
};

typedef __attribute__((address_space(3))) s16x4 lds_s16x4_w1;
DEVINL s16x4 tr_read_w1(const unsigned char* p) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_w1*)p); }

constexpr int W1_NW = 8, W1_NT = W1_NW * 64;

template <int BM>
__global__ __launch_bounds__(W1_NT, 1) void conv_wgrad1_kernel(const W1P p) {
  constexpr int CO = 128, CI = 128, NT = W1_NT;
  constexpr int WCI = 4, WCO = 2, MTC = 2, MTI = 1;        // eight waves: 2 x 4 blocks of (64 co) x (32 ci)
  constexpr int KSW = BM / 16;
  constexpr int PD = CO * 2 + 16, PX = CI * 2 + 16;
  constexpr int DV = CO / 8, XV = CI / 8;
  constexpr int NDV = BM * DV / NT, NXV = BM * XV / NT;
  constexpr int DRP = NT / DV, XRP = NT / XV;
  static_assert((BM * DV) % NT == 0 && (BM * XV) % NT == 0, "whole vectors per thread");
  extern __shared__ __align__(16) unsigned char smem[];
  unsigned char* dz_s = smem;
  unsigned char* x_s = smem + BM * PD;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int G_ = (int)gridDim.x, wg = (int)blockIdx.x;
  const int co0 = ((int)blockIdx.y / p.nci) * CO, ci0 = ((int)blockIdx.y % p.nci) * CI;

  const int dcv = tid % DV, drow = tid / DV;
  const int xcv = tid % XV, xrow = tid / XV;
  const bool dok = co0 + dcv * 8 < p.Cout, xok = ci0 + xcv * 8 < p.Cin;      // channel tails: dz rows hold Cout rounded up to 8
  const bf16_t* dsrc = p.dz + (dok ? co0 + dcv * 8 : 0);
  const bf16_t* xsrc = p.x + (xok ? ci0 + xcv * 8 : 0);

  struct Rows { u32x4 av[NDV], xv[NXV]; };
  auto request = [&](Rows& R, const int tile) {
    const int m0 = tile * BM;
#pragma unroll
    for (int v = 0; v < NDV; ++v) {
      const int m = m0 + drow + v * DRP, mc = (tile < p.ntiles && m < p.M) ? m : 0;
      R.av[v] = *(const u32x4*)(dsrc + (size_t)mc * p.dz_ld);
    }
#pragma unroll
    for (int v = 0; v < NXV; ++v) {
      const int m = m0 + xrow + v * XRP, mc = (tile < p.ntiles && m < p.M) ? m : 0;
      R.xv[v] = *(const u32x4*)(xsrc + (size_t)mc * p.x_ld);
    }
  };
  Rows RA, RB;
  request(RA, wg);
  __builtin_amdgcn_sched_barrier(0);
  request(RB, wg + G_);
  __builtin_amdgcn_sched_barrier(0);

  const int wco = wave / WCI, wci = wave % WCI;
  f32x16 accw[MTC][MTI];
#pragma unroll
  for (int a = 0; a < MTC; ++a)
#pragma unroll
    for (int b = 0; b < MTI; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) accw[a][b][i] = 0.f;

  // transposed-read addresses (conv_pw_bwd.hip): physical row of pixel k inside a 16-pixel k-step = 4*(k&3) + (k>>2)
  const int g4 = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3;
  const int trow = 4 * q + 2 * (g4 >> 1);
  const int a_off = trow * PD + (wco * MTC * 32 + 16 * (g4 & 1) + 4 * pp) * 2;
  const int b_off = trow * PX + (wci * MTI * 32 + 16 * (g4 & 1) + 4 * pp) * 2;

  auto process = [&](Rows& R, const int tile) {
    const int m0 = tile * BM;
    const u32x4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int v = 0; v < NXV; ++v) {
      const int row = xrow + v * XRP;
      *(u32x4*)(x_s + row * PX + xcv * 16) = (xok && m0 + row < p.M) ? R.xv[v] : zero;
    }
#pragma unroll
    for (int v = 0; v < NDV; ++v) {
      const int row = drow + v * DRP;
      *(u32x4*)(dz_s + row * PD + dcv * 16) = (dok && m0 + row < p.M) ? R.av[v] : zero;
    }
    __syncthreads();                                            // the tile's LDS image is complete
    request(R, tile + 2 * G_);                                  // this register set is free again: the tile after next
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < KSW; ++j) {
      s16x8 af[MTC], bfr[MTI];
#pragma unroll
      for (int a = 0; a < MTC; ++a) {
        const unsigned char* ap = dz_s + j * 16 * PD + a_off + a * 64;
        const s16x4 lo = tr_read_w1(ap), hi = tr_read_w1(ap + PD);
        af[a] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int b = 0; b < MTI; ++b) {
        const unsigned char* bp = x_s + j * 16 * PX + b_off + b * 64;
        const s16x4 lo = tr_read_w1(bp), hi = tr_read_w1(bp + PX);
        bfr[b] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int a = 0; a < MTC; ++a)
#pragma unroll
        for (int b = 0; b < MTI; ++b)
          accw[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)&af[a], *(const bf16x8*)&bfr[b], accw[a][b], 0, 0, 0);
    }
    __syncthreads();                                            // every wave is done with dz_s / x_s
  };

  int tile = wg;
  for (; tile + G_ < p.ntiles; tile += 2 * G_) {
    process(RA, tile);
    process(RB, tile + G_);
  }
  if (tile < p.ntiles) process(RA, tile);

  // private slab of this row group: D[row = co][col = ci], col = lane & 31, row = (i&3) + 8*(i>>2) + 4*h
  float* slab = p.dw + (size_t)wg * ((size_t)p.Cout * p.Cin);
#pragma unroll
  for (int b = 0; b < MTI; ++b) {
    const int ci = ci0 + (wci * MTI + b) * 32 + r;
    if (ci < p.Cin) {
#pragma unroll
      for (int a = 0; a < MTC; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int co = co0 + (wco * MTC + a) * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
          if (co < p.Cout) slab[(size_t)co * p.Cin + ci] = accw[a][b][i];
        }
    }
  }
}

}  // namespace

namespace plyolo {

// the launch behind conv_mfma_wgrad for 1x1 stride-1 layers with Cout, Cin >= 128: G row groups (= slabs, the caller's split-K plan),
// dz [M][dz_ld], x [M][x_ld], slabs [G][Cout][Cin] at dw
hipError_t conv_wgrad1_launch(const void* x, const void* dz, float* dw, int M, int Cout, int Cin, int x_ld, int dz_ld, int G, hipStream_t s) {
  constexpr int BM = 64;
  W1P p{};
  p.dz = (const bf16_t*)dz; p.x = (const bf16_t*)x; p.dw = dw;
  p.dz_ld = dz_ld; p.x_ld = x_ld; p.M = M; p.Cout = Cout; p.Cin = Cin;
  p.ntiles = (M + BM - 1) / BM;
  const int nco = (Cout + 127) / 128;
  p.nci = (Cin + 127) / 128;
  if (G > p.ntiles) return hipErrorInvalidValue;
  const size_t lds = (size_t)BM * (128 * 2 + 16) * 2;
  auto kern = conv_wgrad1_kernel<BM>;
  if (hipError_t e = ensure_dynamic_lds((const void*)kern, lds); e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(G, nco * p.nci), dim3(W1_NT), lds, s, p);
  return hipGetLastError();
}

}  // namespace plyolo
