// Data gradient of a 3x3 stride-2 (pad 1) convolution with ALL FOUR output parity classes in one workgroup (gfx950, bf16).
//
// Replaces what ATen's convolution_backward computes for the input of the stride-2 BaseConv layers (reference
// models/layers/network_blocks.py:18-26; the downsampling convolutions of CSPDarknet and the bottom-up path of the PAFPN).
//
// dX[2a+py, 2b+px] collects the taps (kh, kw) with (py + 1 - kh) and (px + 1 - kw) even: class (0,0) one tap, (0,1) and (1,0) two,
// (1,1) four, every one of them reading dZ at (a + dy, b + dx) with dy, dx in {0, 1}.  conv_mfma.hip runs the classes as four jobs of
// one launch, each job staging its own dZ tile per 32-channel chunk: four tile loads for nine taps of MFMA work (a stride-1 3x3
// tile: one load for nine taps).  Here ONE workgroup stages the (TH+1) x 17 dZ tile of a chunk once (double-buffered like the
// stride-1 tiles: the next chunk's vectors are in flight during the taps) and runs all nine taps on it, each into the accumulators
// of its class (4 x MT x 16 fp32 registers per lane); the epilogue stores the four classes one after the other through the LDS
// transpose, whole 16-byte channel vectors at pixel stride 2.
//   * A fragments: tap-shifted ds_read_b128 rows of the tile (row pitch a multiple of 256 bytes, pixel pitch 80: conflict-free);
//   * B fragments: straight from the data-gradient fragment pack (one coalesced 1-KiB load per fragment, a tap ahead);
//   * RED instances fold the BatchNorm-backward reduction of the upstream unit(s) into the four store loops (bnred.h).
#include <type_traits>

#include "conv_mfma_body.h"

namespace {

constexpr int S2D_CK = 32, S2D_KS = 2, S2D_CV = 4, S2D_ROWB = S2D_CK * 2 + 16;
#ifndef S2D_PD
#define S2D_PD 2
#endif

// tap t = kh * 3 + kw (the pack's tap order): parity class and tile offset
constexpr int s2d_py(int t) { return ((t / 3) + 1) & 1; }
constexpr int s2d_px(int t) { return ((t % 3) + 1) & 1; }
constexpr int s2d_dy(int t) { return (t / 3) == 0 ? 1 : 0; }
constexpr int s2d_dx(int t) { return (t % 3) == 0 ? 1 : 0; }

// OCC: workgroups per CU the register budget is set for (2 everywhere: the 128-channel block with 8-row tiles and one wave per SIMD --
// 256 accumulator registers -- was measured at 125 us per launch against 84 for the four-job launch and 93 for this 4-row instance,
// which conv_mfma.hip does not use by default either: PLYOLO_S2D_MAXC)
template <int BN, int TH, bool RED, int OCC>
__global__ __launch_bounds__(256, OCC) void conv_s2d_kernel(const ConvP p) {
  constexpr int BM = TH * TW;
  constexpr int WN = BN / 32, WM = 4 / WN, MT = BM / (32 * WM);
  constexpr int KS = S2D_KS, CV = S2D_CV, ROWB = S2D_ROWB;
  constexpr int ITH = TH + 1, ITW = TW + 1;
  constexpr int HVT = (ITH * ITW * CV + 255) / 256;
  static_assert(MT >= 1 && BM % (32 * WM) == 0, "tile rows vs wave layout");
  extern __shared__ __align__(16) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;

  int tile;
  {
    const int nwg = (int)gridDim.x, bid = (int)blockIdx.x;
    const int q = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    tile = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
  }
  const int txi = tile % p.tiles_x;
  const int t2 = tile / p.tiles_x;
  const int tyi = t2 % p.tiles_y;
  const int n = t2 / p.tiles_y;
  const int a0 = tyi * TH, b0 = txi * TW;
  const int cout0 = blockIdx.y * BN;
  const int nb = blockIdx.y * WN + wn;

  int arow[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = (wm * MT + mt) * 32 + r;
    arow[mt] = (m >> 4) * p.rowp + (m & 15) * ROWB + h * 16;
  }

  f32x16 acc[4][MT];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[c][mt][i] = 0.f;

  const int nchunks = (p.Cin + S2D_CK - 1) / S2D_CK;
  const bool nb_ok = nb < p.nnb;
  const char* wbase = (const char*)p.w;
  const unsigned wvoff = (unsigned)((nb_ok ? nb : p.nnb - 1) * p.nkb) * 1024u + (unsigned)lane * 16u;
  const unsigned wtapB = (unsigned)p.nnb * (unsigned)p.nkb * 1024u;

  // weight fragments of (chunk, tap): k-blocks beyond Cin meet zero-filled tile columns -- clamp instead of masking
  auto load_b = [&](u32x4* dst, const int chunk, const int t) {
    const char* wt = wbase + (size_t)t * wtapB;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      const int kb = chunk * KS + kk;
      const unsigned kbc = (unsigned)(kb < p.nkb ? kb : p.nkb - 1);
      dst[kk] = *(const u32x4*)(wt + (size_t)(kbc * 1024u) + wvoff);
    }
  };

  // ---- dZ tile loader: the (pixel, channel vector) -> (global offset, LDS offset) map of this thread's vectors, once per tile
  const bf16_t* xn = p.x + (size_t)n * p.H * p.W * p.x_ld;
  const int cvt = tid % CV;
  int goff[HVT], loff[HVT];
#pragma unroll
  for (int v = 0; v < HVT; ++v) {
    const int idx = tid + v * 256;
    goff[v] = -1;
    loff[v] = -1;
    if (idx < ITH * ITW * CV) {
      const int pix = idx / CV;
      const int iy = pix / ITW, ix = pix - iy * ITW;
      const int gy = a0 + iy, gx = b0 + ix;
      loff[v] = iy * p.rowp + ix * ROWB + cvt * 16;
      if (gy < p.H && gx < p.W) goff[v] = (gy * p.W + gx) * p.x_ld + cvt * 8;
    }
  }
  // unconditional loads (out-of-range vectors read the image's first bytes and are stored as zeros): exact vmcnt bookkeeping
  auto halo_load = [&](const int c0, u32x4* hv) {
    const bool cok = c0 + cvt * 8 < p.Cin;
#pragma unroll
    for (int v = 0; v < HVT; ++v) hv[v] = *(const u32x4*)(xn + ((goff[v] >= 0 && cok) ? goff[v] + c0 : 0));
  };
  auto halo_store = [&](const int c0, const u32x4* hv, const int boff) {
    const bool cok = c0 + cvt * 8 < p.Cin;
    const u32x4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int v = 0; v < HVT; ++v)
      if (loff[v] >= 0) *(u32x4*)(smem + boff + loff[v]) = (goff[v] >= 0 && cok) ? hv[v] : zero;
  };

  // weight fragments PD taps ahead (a tap of this kernel is only 2 x MT MFMAs: one tap of distance does not cover an L2 round trip)
  constexpr int PD = S2D_PD;
  u32x4 bq[PD + 1][KS];
#pragma unroll
  for (int d = 0; d < PD; ++d) load_b(bq[d], d / 9 < nchunks ? d / 9 : nchunks - 1, d % 9);
  u32x4 hv[HVT];
  halo_load(0, hv);
  halo_store(0, hv, 0);
  __syncthreads();

  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const int boff = (chunk & 1) * p.bufsz;
    const int cnext = (chunk + 1) * S2D_CK;   // past Cin on the last chunk: every load collapses to the dummy address
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      // the fragments of the tap PD ahead (into the next chunk behind the last taps; parked on the last chunk)
      {
        constexpr int dummy_pd = PD;
        const int tn = (t + dummy_pd) % 9, cn = chunk + (t + dummy_pd) / 9;
        load_b(bq[PD], cn < nchunks ? cn : nchunks - 1, tn);
      }
      if (t == 0) halo_load(cnext, hv);
      __builtin_amdgcn_sched_barrier(0);   // the prefetch stays ABOVE the MFMA block
      const int toff = boff + s2d_dy(t) * p.rowp + s2d_dx(t) * ROWB;
      constexpr int KSTEPS = KS;
      bf16x8 a[MT], an[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) a[mt] = *(const bf16x8*)(smem + arow[mt] + toff);
#pragma unroll
      for (int kk = 0; kk < KSTEPS; ++kk) {
        const bf16x8 b = *(const bf16x8*)&bq[0][kk];
        if (kk + 1 < KSTEPS) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) an[mt] = *(const bf16x8*)(smem + arow[mt] + toff + (kk + 1) * 32);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc[s2d_py(t) * 2 + s2d_px(t)][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt], b, acc[s2d_py(t) * 2 + s2d_px(t)][mt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (kk + 1 < KSTEPS) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) a[mt] = an[mt];
        }
      }
#pragma unroll
      for (int d = 0; d < PD; ++d)
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) bq[d][kk] = bq[d + 1][kk];
    }
    if (chunk + 1 < nchunks) halo_store(cnext, hv, p.bufsz - boff);
    __syncthreads();   // next buffer complete; every wave is done with this one
  }

  // ---- epilogue: the four classes one after the other through the LDS transpose
  constexpr int SROW = BN * 2 + 16;
  constexpr int VPR = BN / 8;
  bf16_t* y = (bf16_t*)p.y;
  [[maybe_unused]] BnRedThread rt;
  [[maybe_unused]] const int rv = tid % VPR, rco = cout0 + rv * 8;
  if constexpr (RED) bnred_init(rt, p.red, rco);
  // (one call per class with the class as a compile-time constant: a loop the compiler declines to unroll would index the
  // accumulators dynamically and move all 256 of them to scratch)
  auto store_class = [&](auto cc) {
    constexpr int c = decltype(cc)::value;
    constexpr int py = c >> 1, px = c & 1;
    if (c) __syncthreads();   // the previous class's rows have been read
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int m = (wm * MT + mt) * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        *(bf16_t*)(smem + m * SROW + (wn * 32 + r) * 2) = f2bf(acc[c][mt][i]);
      }
    __syncthreads();
    if constexpr (RED) {
      constexpr int NIT = BM * VPR / 256, NB = NIT > 4 ? 4 : NIT;    // rows per thread, requested NB at a time
      static_assert(256 % VPR == 0 && (BM * VPR) % 256 == 0 && NIT % NB == 0, "RED: whole rows per thread");
#pragma unroll
      for (int it0 = 0; it0 < NIT; it0 += NB) {
        u32x4 zq[NB], old[NB];
        size_t pixs[NB];
        bool ok[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) {
          const int m = (tid + (it0 + k) * 256) / VPR;
          const int oy = (a0 + (m >> 4)) * 2 + py, ox = (b0 + (m & 15)) * 2 + px;
          ok[k] = oy < p.OHf && ox < p.OWf && rco < p.Cout;
          pixs[k] = ok[k] ? (size_t)(n * p.OHf + oy) * p.OWf + ox : 0;
          zq[k] = rt.z ? bnred_load(rt, pixs[k]) : u32x4{0u, 0u, 0u, 0u};
          old[k] = p.accumulate ? *(const u32x4*)(y + pixs[k] * p.y_ld + (rco < p.Cout ? rco : 0)) : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int k = 0; k < NB; ++k) {
          const int m = (tid + (it0 + k) * 256) / VPR;
          if (ok[k]) {
            u32x4 val = *(const u32x4*)(smem + m * SROW + rv * 16);
            if (p.accumulate) val = add_bf16x8(old[k], val);
            *(u32x4*)(y + pixs[k] * p.y_ld + rco) = val;
            if (rt.z) bnred_add_any(rt, val, zq[k]);
          }
        }
      }
    } else {
      for (int idx = tid; idx < BM * VPR; idx += 256) {
        const int m = idx / VPR, v = idx - m * VPR;
        const int oy = (a0 + (m >> 4)) * 2 + py, ox = (b0 + (m & 15)) * 2 + px, co = cout0 + v * 8;
        if (oy < p.OHf && ox < p.OWf && co < p.Cout) {
          u32x4 val = *(const u32x4*)(smem + m * SROW + v * 16);
          bf16_t* dst = y + ((size_t)(n * p.OHf + oy) * p.OWf + ox) * p.y_ld + co;
          if (p.accumulate) val = add_bf16x8(*(const u32x4*)dst, val);
          *(u32x4*)dst = val;
        }
      }
    }
  };
  store_class(std::integral_constant<int, 0>{});
  store_class(std::integral_constant<int, 1>{});
  store_class(std::integral_constant<int, 2>{});
  store_class(std::integral_constant<int, 3>{});
  if constexpr (RED) {
    __syncthreads();   // the last class's staging rows have been read: the fold reuses them
    bnred_flush<256, VPR>(rt, p.red, cout0, (float*)smem, tid, tile % PLYOLO_STAT_SLOTS);
  }
}

template <int BN, int TH, int OCC>
hipError_t s2d_launch_inst(ConvP p, bool red, hipStream_t s) {
  constexpr int BM = TH * TW, SROW = BN * 2 + 16;
  p.rowp = ((TW + 1) * S2D_ROWB + 255) & ~255;
  p.bufsz = (TH + 1) * p.rowp;
  size_t lds = 2 * (size_t)p.bufsz;
  const size_t lds_epi = (size_t)BM * SROW;
  lds = lds > lds_epi ? lds : lds_epi;
  lds = lds > 16384 ? lds : 16384;   // bnred_flush scratch
  auto kern = red ? conv_s2d_kernel<BN, TH, true, OCC> : conv_s2d_kernel<BN, TH, false, OCC>;
  if (hipError_t e = plyolo::ensure_dynamic_lds((const void*)kern, lds); e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(p.nmb, (p.Cout + BN - 1) / BN), dim3(256), lds, s, p);
  return hipGetLastError();
}

}  // namespace

namespace plyolo {

// rows of the class grid per tile for BN output channels (0: no instance)
int conv_s2d_th(int BN) { return BN == 128 ? 4 : (BN == 64 ? 8 : (BN == 32 ? 16 : 0)); }

// `convp`: x = dZ [N, H = OH, W = OW, Cin = K], y = dX [N, OHf, OWf, Cout], w = the data-gradient pack, tiles_y / tiles_x / nmb set
// for conv_s2d_th(BN) rows x 16 columns of the class grid (ceil(OHf / 2) x ceil(OWf / 2))
hipError_t conv_s2d_launch(const void* convp, int BN, int red, hipStream_t s) {
  const ConvP& p = *(const ConvP*)convp;
  if (BN == 128) return s2d_launch_inst<128, 4, 2>(p, red != 0, s);
  if (BN == 64) return s2d_launch_inst<64, 8, 2>(p, red != 0, s);
  if (BN == 32) return s2d_launch_inst<32, 16, 2>(p, red != 0, s);
  return hipErrorInvalidValue;
}

}  // namespace plyolo
