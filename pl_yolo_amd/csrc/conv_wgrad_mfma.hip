// bf16 MFMA weight-gradient for gfx950.
//
// Replaces the weight half of ATen convolution_backward for nn.Conv2d inside
// BaseConv (reference models/layers/network_blocks.py:18-26).
//
//   dW[tap][co][ci] = sum over output pixels  dY[pix][co] * X[pix*s + tap - pad][ci]
//
// is a GEMM whose contraction index is the PIXEL.  Both operands live pixel-major
// (NHWC) in HBM and in LDS, so the MFMA fragments (8 consecutive k per lane) are
// transposed on the way out of LDS with ds_read_b64_tr_b16 -- each lane supplies
// its own row address, so the tap shift and the stride-2 sampling of X cost nothing.
// One workgroup (4 waves) owns a CO_T x CI_T slab of dW for ALL taps (9*16 fp32
// accumulator registers per wave) and streams a strided share of the 8x16 output
// tiles: per tile the dY tile and the X halo tile are staged once in LDS and used
// for 8 k-steps x ntaps MFMAs.  Partial slabs from the spatial splits are combined
// with fp32 atomics shaped as full 128-byte row segments.
#include "common.h"

namespace {

constexpr int TH = 8, TW = 16;

struct WgP {
  const bf16_t* x;
  const bf16_t* dy;
  float* dw;
  int N, H, W, OH, OW, Cin, Cout, x_ld, dy_ld;
  int si, pad, ITH, ITW;
  int tiles_y, tiles_x, ntiles;
  int nci;  // number of ci tiles (blockIdx.y = co_tile * nci + ci_tile)
};

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

DEVINL s16x4 tr_read(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
}

constexpr int pitch_for(int ch) { return ch * 2 + ((ch * 2) % 128 == 0 ? 64 : 0); }

template <int CO_T, int CI_T, int KS>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgP p) {
  constexpr int NTAPS = KS * KS;
  constexpr int WCI = CI_T / 32;
  constexpr int DZB = pitch_for(CO_T), XB = pitch_for(CI_T);
  constexpr int DZV = CO_T / 8, XV = CI_T / 8;
  extern __shared__ __align__(16) unsigned char smem[];
  unsigned char* dz_s = smem;
  unsigned char* x_s = smem + TH * TW * DZB;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wco = wave / WCI, wci = wave % WCI;
  const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3;
  const int co_tile = blockIdx.y / p.nci, ci_tile = blockIdx.y % p.nci;
  const int co0 = co_tile * CO_T, ci0 = ci_tile * CI_T;

  // per-lane tr-read address pieces: pixel column within the 16-wide k-step, channel column
  const int kpix = 8 * (g >> 1) + q;                 // + 4 for the second read
  const int a_col = (wco * 32 + 16 * (g & 1) + 4 * pp) * 2;
  const int b_col = (wci * 32 + 16 * (g & 1) + 4 * pp) * 2;

  f32x16 acc[NTAPS];
#pragma unroll
  for (int t = 0; t < NTAPS; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    const int txi = tile % p.tiles_x;
    const int t2 = tile / p.tiles_x;
    const int tyi = t2 % p.tiles_y;
    const int n = t2 / p.tiles_y;
    const int oy0 = tyi * TH, ox0 = txi * TW;
    const int iy0 = oy0 * p.si - p.pad, ix0 = ox0 * p.si - p.pad;
    __syncthreads();  // previous tile fully consumed
    {
      // stage both tiles with all 16-byte loads of a batch in flight before the first LDS write
      const bf16_t* dyn = p.dy + (size_t)n * p.OH * p.OW * p.dy_ld;
      constexpr int DV = (TH * TW * DZV + 255) / 256;
      u32x4 dv[DV];
#pragma unroll
      for (int v = 0; v < DV; ++v) {
        const int idx = tid + v * 256;
        u32x4 val = {0u, 0u, 0u, 0u};
        if (idx < TH * TW * DZV) {
          const int m = idx / DZV, vv = idx - m * DZV;
          const int oy = oy0 + (m >> 4), ox = ox0 + (m & 15), co = co0 + vv * 8;
          if (oy < p.OH && ox < p.OW && co < p.Cout) val = *(const u32x4*)(dyn + ((size_t)oy * p.OW + ox) * p.dy_ld + co);
        }
        dv[v] = val;
      }
      const bf16_t* xn = p.x + (size_t)n * p.H * p.W * p.x_ld;
      const int nvec = p.ITH * p.ITW * XV;
      constexpr int HV = 6;
      for (int base = 0; base < nvec; base += HV * 256) {
        u32x4 hv[HV];
#pragma unroll
        for (int v = 0; v < HV; ++v) {
          const int idx = base + tid + v * 256;
          u32x4 val = {0u, 0u, 0u, 0u};
          if (idx < nvec) {
            const int pix = idx / XV, vv = idx - pix * XV;
            const int iy = pix / p.ITW, ix = pix - iy * p.ITW;
            const int gy = iy0 + iy, gx = ix0 + ix, ci = ci0 + vv * 8;
            if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W && ci < p.Cin)
              val = *(const u32x4*)(xn + ((size_t)gy * p.W + gx) * p.x_ld + ci);
          }
          hv[v] = val;
        }
        if (base == 0) {
#pragma unroll
          for (int v = 0; v < DV; ++v) {
            const int idx = tid + v * 256;
            if (idx < TH * TW * DZV) {
              const int m = idx / DZV, vv = idx - m * DZV;
              *(u32x4*)(dz_s + m * DZB + vv * 16) = dv[v];
            }
          }
        }
#pragma unroll
        for (int v = 0; v < HV; ++v) {
          const int idx = base + tid + v * 256;
          if (idx < nvec) {
            const int pix = idx / XV, vv = idx - pix * XV;
            *(u32x4*)(x_s + pix * XB + vv * 16) = hv[v];
          }
        }
      }
    }
    __syncthreads();
#pragma unroll 2
    for (int j = 0; j < TH; ++j) {
      // A = dY^T fragment: A[row = co][k = pixel (j, 8h..8h+7)]
      s16x8 af;
      {
        const unsigned char* ap = dz_s + (j * TW + kpix) * DZB + a_col;
        const s16x4 lo = tr_read(ap);
        const s16x4 hi = tr_read(ap + 4 * DZB);
        af = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int t = 0; t < NTAPS; ++t) {
        const int dy_ = t / KS, dx_ = t % KS;
        const unsigned char* bp = x_s + ((j * p.si + dy_) * p.ITW + kpix * p.si + dx_) * XB + b_col;
        const s16x4 lo = tr_read(bp);
        const s16x4 hi = tr_read(bp + 4 * p.si * XB);
        const s16x8 bfv = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)&af, *(const bf16x8*)&bfv, acc[t], 0, 0, 0);
      }
    }
  }

  // D[row = co][col = ci]: col = lane & 31, row = (i&3) + 8*(i>>2) + 4*(lane>>5)
  const int r = lane & 31, h = lane >> 5;
  const int ci = ci0 + wci * 32 + r;
  if (ci < p.Cin) {
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int co = co0 + wco * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (co < p.Cout) atomicAdd(p.dw + ((size_t)t * p.Cout + co) * p.Cin + ci, acc[t][i]);
      }
  }
}

template <int CO_T, int CI_T, int KS>
hipError_t launch_wg(const WgP& p, int S, hipStream_t s) {
  constexpr int DZB = pitch_for(CO_T), XB = pitch_for(CI_T);
  const size_t lds = (size_t)TH * TW * DZB + (size_t)p.ITH * p.ITW * XB;
  auto kern = conv_wgrad_kernel<CO_T, CI_T, KS>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  const int nco = (p.Cout + CO_T - 1) / CO_T;
  dim3 grid(S, nco * p.nci);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, p);
  return hipGetLastError();
}

}  // namespace

namespace plyolo {

int conv_mfma_wgrad(const plyolo_conv_desc* d, const void* x, const void* dy, float* dwp, void* stream) {
  const int pad = (d->ksize - 1) / 2;
  WgP p{};
  p.x = (const bf16_t*)x;
  p.dy = (const bf16_t*)dy;
  p.dw = dwp;
  p.N = d->N; p.H = d->H; p.W = d->W;
  p.OH = (d->H + 2 * pad - d->ksize) / d->stride + 1;
  p.OW = (d->W + 2 * pad - d->ksize) / d->stride + 1;
  p.Cin = d->Cin; p.Cout = (d->Cout + 7) & ~7;  // dy rows carry Cout rounded up to 8 channels (zeros)
  p.x_ld = d->x_ld; p.dy_ld = d->y_ld;
  p.si = d->stride; p.pad = pad;
  if (d->ksize == 1 && d->stride == 1 && ((size_t)d->N * d->H * d->W) % TW == 0) {
    const int rows = (int)((size_t)d->N * d->H * d->W / TW);
    p.N = 1; p.H = rows; p.W = TW; p.OH = rows; p.OW = TW;
  }
  p.ITH = (TH - 1) * p.si + d->ksize;
  p.ITW = (TW - 1) * p.si + d->ksize;
  p.tiles_y = (p.OH + TH - 1) / TH;
  p.tiles_x = (p.OW + TW - 1) / TW;
  p.ntiles = p.N * p.tiles_y * p.tiles_x;
  // NOTE: dwp is [tap][Cout][Cin] with the TRUE Cout of the conv; rows >= Cout never get written
  const int true_cout = d->Cout;
  const bool wide = (d->stride == 2 && d->ksize == 3) || d->Cin <= 32;  // 128x32 slab keeps the s2 halo tile small
  const int CO_T = wide ? 128 : 64, CI_T = wide ? 32 : 64;
  p.nci = (p.Cin + CI_T - 1) / CI_T;
  const int nco = (p.Cout + CO_T - 1) / CO_T;
  // spatial split: enough workgroups to cover the chip several times over, but the fp32
  // atomic combine moves S * |dW| bytes at ~1.3 TB/s chip-wide, so cap it at ~24 MB
  const double dw_bytes = 4.0 * d->ksize * d->ksize * (double)d->Cout * d->Cin;
  int S = 2048 / (nco * p.nci);
  const int s_budget = (int)(24.0e6 / dw_bytes);
  if (S > s_budget) S = s_budget;
  if (S < 1) S = 1;
  if (S > p.ntiles) S = p.ntiles;
  p.Cout = true_cout;
  const int ks = d->ksize;
  {
    char lab[64];
    snprintf(lab, sizeof(lab), "conv_wgrad<%dx%d,k%d>", CO_T, CI_T, ks);
    const double Mo = (double)p.N * p.OH * p.OW, Mi = (double)p.N * p.H * p.W;
    annotate(lab, 2.0 * Mo * d->Cout * d->Cin * ks * ks, (Mo * d->Cout + Mi * d->Cin) * 2.0 + 4.0 * ks * ks * d->Cout * d->Cin);
  }
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    if (wide) return ks == 3 ? launch_wg<128, 32, 3>(p, S, s) : launch_wg<128, 32, 1>(p, S, s);
    return ks == 3 ? launch_wg<64, 64, 3>(p, S, s) : launch_wg<64, 64, 1>(p, S, s);
  });
}

}  // namespace plyolo
