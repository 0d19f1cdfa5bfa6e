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
// for 8 k-steps x ntaps MFMAs.  Every spatial split writes its own PRIVATE fp32 slab (plain 128-byte
// row stores, no atomics); plyolo_reduce_slabs folds the slabs in a fixed order right behind this launch.
// The 3x3 variants run ONE workgroup per CU (168 VGPRs + 144 accumulator AGPRs): measured better for the
// whole step than two register-capped workgroups, the launches are deliberately under-filled
// (~20 MB of slabs per layer) and share the chip with the data-gradient lane.
#include <stdlib.h>

#include "common.h"

#include "conv_wgrad_common.h"

namespace {

// CO_T x CI_T: dW slab of the workgroup; MTC x MTI: 32x32 MFMA tiles per wave along co / ci;
// WK: waves that split the k-steps (tile rows) of one slab (small-channel layers); TH_: tile rows.
// TRS: 1 = the workgroup accumulates all KS*KS taps; 3 (KS == 3 only) = one kernel ROW per workgroup -- a third of the
// accumulator registers (48 instead of 144: 3-4 waves per SIMD instead of one), a halo tile without the two extra rows,
// three times as many workgroups for the same number of slabs.
template <int CO_T, int CI_T, int KS, int WK, int MTC, int MTI, int TH_, int SI, bool PRE, int TRS = 1>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgP p) {
  static_assert(TRS == 1 || (TRS == 3 && KS == 3), "tap-row split: 3x3 only");
  constexpr int NTAPS = KS * KS / TRS;     // taps accumulated by this workgroup
  constexpr int KSY = KS / TRS;            // kernel rows covered by this workgroup
  constexpr int WCO = CO_T / (32 * MTC), WCI = CI_T / (32 * MTI);
  static_assert(WCO * WCI * WK == 4, "four waves per workgroup");
  constexpr int DZB = pitch_for(CO_T), XB = pitch_for(CI_T);
  constexpr int DZV = CO_T / 8, XV = CI_T / 8;
  extern __shared__ __align__(16) unsigned char smem[];
  unsigned char* dz_s = smem;
  unsigned char* x_s = smem + TH_ * TW * DZB;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wk = wave / (WCO * WCI), wco = (wave / WCI) % WCO, wci = wave % WCI;
  const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3;
  // Workgroup -> (spatial split, slab tile).  The nco*nci workgroups of one split read the SAME pixel tiles (each its own
  // channel slice of dY and X); in the round-1 launch (grid (S, nco*nci)) they were dispatched far apart and every operand tile
  // came from HBM once per slab column -- 84 MB read per launch against 61 MB algorithmic.  Here they are neighbours in an order
  // that gives each XCD a contiguous run of workgroups (blocks with equal id % 8 share an XCD), so the second .. fourth reader of
  // a tile finds it in that XCD's L2.
  int wg = (int)blockIdx.x;
  if (p.xcd) {
    const int nwg = (int)gridDim.x, q = nwg >> 3, r8 = nwg & 7, x = wg & 7;
    wg = (x < r8 ? x * (q + 1) : r8 * (q + 1) + (x - r8) * q) + (wg >> 3);
  }
  const int split = p.xcd ? wg / p.nslabt : wg % p.S, slab_t0 = p.xcd ? wg % p.nslabt : wg / p.S;
  const int trow = slab_t0 % TRS, slab_tile = slab_t0 / TRS;    // kernel row of this workgroup (TRS == 3)
  const int co_tile = slab_tile / p.nci, ci_tile = slab_tile % p.nci;
  const int co0 = co_tile * CO_T, ci0 = ci_tile * CI_T;

  // per-lane tr-read address pieces: pixel column within the 16-wide k-step, channel column
  const int kpix = 8 * (g >> 1) + q;                 // + 4 for the second read
  const int a_col = (wco * 32 * MTC + 16 * (g & 1) + 4 * pp) * 2;
  const int b_col = (wci * 32 * MTI + 16 * (g & 1) + 4 * pp) * 2;

  f32x16 acc[NTAPS][MTC][MTI];
#pragma unroll
  for (int t = 0; t < NTAPS; ++t)
#pragma unroll
    for (int a = 0; a < MTC; ++a)
#pragma unroll
      for (int b = 0; b < MTI; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][a][b][i] = 0.f;

  // Software pipeline over the workgroup's tiles: the 16-byte global loads of tile i+1 are issued
  // into registers before the MFMAs of tile i and written to LDS after them.
  constexpr int ITH_ = (TH_ - 1) * SI + KSY, ITW_ = (TW - 1) * SI + KS;
  constexpr int DV = (TH_ * TW * DZV + 255) / 256;
  constexpr int NXV = ITH_ * ITW_ * XV;
  constexpr int HV = (NXV + 255) / 256;
  // two register sets: two tiles are in flight while a third is being multiplied (one set covered a single HBM round trip per
  // tile, ~2 us against ~1 us of MFMA work: the loop ran at the memory LATENCY, neither roof in sight)
  u32x4 dvA[DV], hvA[HV], dvB[DV], hvB[HV];
  // lazy input: a thread always stages the same 8 input channels (256 % XV == 0), so their BatchNorm coefficients sit in
  // registers for the whole kernel; `hmask` remembers which of the prefetched vectors are real pixels (padding stays 0)
  unsigned hmaskA = 0u, hmaskB = 0u;
  float psc[PRE ? 8 : 1], psh[PRE ? 8 : 1];
  if constexpr (PRE) {
    static_assert(256 % XV == 0 && HV <= 32, "lazy input staging assumes a fixed channel vector per thread");
    const int ci = ci0 + (tid % XV) * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      psc[i] = ci < p.Cin ? p.pre[ci + i] : 0.f;
      psh[i] = ci < p.Cin ? p.pre[p.pre_ld + ci + i] : 0.f;
    }
  }
  auto prefetch = [&](int tile, u32x4 (&dv)[DV], u32x4 (&hv)[HV], unsigned& hmask) {
    const int txi = tile % p.tiles_x;
    const int t2 = tile / p.tiles_x;
    const int tyi = t2 % p.tiles_y;
    const int n = t2 / p.tiles_y;
    const int oy0 = tyi * TH_, ox0 = txi * TW;
    const int iy0 = oy0 * SI - p.pad + (TRS == 3 ? trow : 0), ix0 = ox0 * SI - p.pad;
    const bf16_t* dyn = p.dy + (size_t)n * p.OH * p.OW * p.dy_ld;
#pragma unroll
    for (int v = 0; v < DV; ++v) {
      const int idx = tid + v * 256;
      u32x4 val = {0u, 0u, 0u, 0u};
      if (idx < TH_ * TW * DZV) {
        const int m = idx / DZV, vv = idx - m * DZV;
        const int oy = oy0 + (m >> 4), ox = ox0 + (m & 15), co = co0 + vv * 8;
        if (oy < p.OH && ox < p.OW && co < p.Cout) val = *(const u32x4*)(dyn + ((size_t)oy * p.OW + ox) * p.dy_ld + co);
      }
      dv[v] = val;
    }
    const bf16_t* xn = p.x + (size_t)n * p.H * p.W * p.x_ld;
    hmask = 0u;
#pragma unroll
    for (int v = 0; v < HV; ++v) {
      const int idx = tid + v * 256;
      u32x4 val = {0u, 0u, 0u, 0u};
      if (idx < NXV) {
        const int pix = idx / XV, vv = idx - pix * XV;
        const int iy = pix / ITW_, ix = pix - iy * ITW_;
        const int gy = iy0 + iy, gx = ix0 + ix, ci = ci0 + vv * 8;
        if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W && ci < p.Cin) {
          val = *(const u32x4*)(xn + ((size_t)gy * p.W + gx) * p.x_ld + ci);
          if constexpr (PRE) hmask |= 1u << v;
        }
      }
      hv[v] = val;
    }
  };
  auto commit = [&](const u32x4 (&dv)[DV], const u32x4 (&hv)[HV], const unsigned hmask) {
#pragma unroll
    for (int v = 0; v < DV; ++v) {
      const int idx = tid + v * 256;
      if (idx < TH_ * TW * DZV) {
        const int m = idx / DZV, vv = idx - m * DZV;
        *(u32x4*)(dz_s + m * DZB + vv * 16) = dv[v];
      }
    }
#pragma unroll
    for (int v = 0; v < HV; ++v) {
      const int idx = tid + v * 256;
      if (idx < NXV) {
        const int pix = idx / XV, vv = idx - pix * XV;
        u32x4 t = hv[v];
        if constexpr (PRE) {
          if (hmask & (1u << v)) {
            t = bn_act_vec8(t, psc, psh, p.pre_act);
          }
        }
        *(u32x4*)(x_s + pix * XB + vv * 16) = t;
      }
    }
  };

  auto multiply = [&]() {
    if (!(p.ablate & 4)) {
      // MFMA phase, software-pipelined by hand: the fragments of k-step j+1 are read from LDS (ds_read_b64_tr_b16) while the
      // MFMAs of k-step j run.  Left to itself the compiler keeps this a rolled loop that reads each B fragment right before
      // the MFMA using it: with one wave per SIMD (the 3x3 variants hold 144 accumulator registers) every k-step exposed the
      // LDS latency several times -- ~1000 cycles per 9 MFMAs (288 cycles of matrix-core time).
      constexpr int NJ = TH_ / WK;                       // k-steps (tile rows) of this wave
      constexpr int NRD = 2 * (MTC + NTAPS * MTI), NMM = NTAPS * MTI * MTC;
      s16x8 af0[MTC], af1[MTC], bf0[NTAPS][MTI], bf1[NTAPS][MTI];
      auto ldfrag = [&](int j, s16x8 (&af)[MTC], s16x8 (&bfv)[NTAPS][MTI]) {
#pragma unroll
        for (int a = 0; a < MTC; ++a) {          // A = dY^T fragments: A[row = co][k = pixel (j, 8h..8h+7)]
          const unsigned char* ap = dz_s + (j * TW + kpix) * DZB + a_col + a * 64;
          const s16x4 lo = tr_read(ap);
          const s16x4 hi = tr_read(ap + 4 * DZB);
          af[a] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) {
          const int dy_ = t / KS, dx_ = t % KS;     // dy_ is relative to the staged halo rows (0 under the tap-row split)
#pragma unroll
          for (int b = 0; b < MTI; ++b) {
            const unsigned char* bp = x_s + ((j * SI + dy_) * ITW_ + kpix * SI + dx_) * XB + b_col + b * 64;
            const s16x4 lo = tr_read(bp);
            const s16x4 hi = tr_read(bp + 4 * SI * XB);
            bfv[t][b] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          }
        }
      };
      auto mm = [&](const s16x8 (&af)[MTC], const s16x8 (&bfv)[NTAPS][MTI]) {
#pragma unroll
        for (int t = 0; t < NTAPS; ++t)
#pragma unroll
          for (int b = 0; b < MTI; ++b)
#pragma unroll
            for (int a = 0; a < MTC; ++a)
              acc[t][a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)&af[a], *(const bf16x8*)&bfv[t][b], acc[t][a][b], 0, 0, 0);
      };
      // every k-step is its own scheduling region: [MFMA, RPM reads] repeated, the reads of the NEXT step front-loaded into the
      // first half of this step's MFMAs so that they have landed when the next region starts
      constexpr int RPM = (NRD + (NMM > 1 ? NMM / 2 : 1) - 1) / (NMM > 1 ? NMM / 2 : 1);
      auto interleave = [&]() {
#pragma unroll
        for (int i = 0; i < NMM; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, RPM, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      ldfrag(wk, af0, bf0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int jj = 0; jj < NJ; jj += 2) {
        const int j = wk + jj * WK;
        if (jj + 1 < NJ) ldfrag(j + WK, af1, bf1);
        mm(af0, bf0);
        interleave();
        if (jj + 2 < NJ) ldfrag(j + 2 * WK, af0, bf0);
        if (jj + 1 < NJ) {
          mm(af1, bf1);
          interleave();
        }
      }
    }
  };

  if (split < p.ntiles) prefetch(split, dvA, hvA, hmaskA);
  if (split + p.S < p.ntiles && !(p.ablate & 2)) prefetch(split + p.S, dvB, hvB, hmaskB);
  for (int tile = split; tile < p.ntiles; tile += 2 * p.S) {
    __syncthreads();  // previous tile fully consumed
    commit(dvA, hvA, hmaskA);
    __syncthreads();
    if (tile + 2 * p.S < p.ntiles && !(p.ablate & 2)) prefetch(tile + 2 * p.S, dvA, hvA, hmaskA);
    multiply();
    if (tile + p.S >= p.ntiles) break;
    __syncthreads();
    commit(dvB, hvB, hmaskB);
    __syncthreads();
    if (tile + 3 * p.S < p.ntiles && !(p.ablate & 2)) prefetch(tile + 3 * p.S, dvB, hvB, hmaskB);
    multiply();
  }

  if (p.ablate & 1) { if (acc[0][0][0][0] == 123.456f) p.dw[0] = 1.f; return; }
  // Each (spatial split, k-split wave) owns a private slab [tap][Cout][Cin]: plain 128-byte row
  // stores (D[row = co][col = ci]: col = lane & 31, row = (i&3) + 8*(i>>2) + 4*(lane>>5)); the
  // slabs are summed in a fixed order by plyolo_unpack_wgrads (deterministic, and no fp32
  // atomics: same-address atomics serialise at ~1.6 us each on gfx950).
  float* slab = p.dw + ((size_t)split * WK + wk) * ((size_t)KS * KS * p.Cout * p.Cin) + (size_t)(TRS == 3 ? trow * KS : 0) * p.Cout * p.Cin;
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int b = 0; b < MTI; ++b) {
    const int ci = ci0 + (wci * MTI + b) * 32 + r;
    if (ci < p.Cin) {
#pragma unroll
      for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int a = 0; a < MTC; ++a)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int co = co0 + (wco * MTC + a) * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
            if (co < p.Cout) slab[((size_t)t * p.Cout + co) * p.Cin + ci] = acc[t][a][b][i];
          }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// 3x3 kernel, round 2: the tile pipeline runs INSIDE the MFMA phase.
//
// conv_wgrad_kernel above alternates phases -- wait for the prefetched tile, write it to LDS, barrier, issue the next
// prefetch (~200 address / bounds instructions), multiply -- and the 3x3 variants run one wave per SIMD (144 accumulator
// registers), so nothing overlaps: measured per 8x16 tile of the 64x64 variant 1.33 us of MFMA phase + 0.45 us of commit and
// barriers + 0.37 us of prefetch issue, against 0.96 us of matrix-core time.  Here
//   * LDS holds TWO tiles: while tile i is multiplied out of one buffer, tile i+1 is written into the other and tile i+2 is
//     requested from HBM -- vector by vector, spread over the k-steps, in the shadow of the MFMAs (a wave issues ~7 other
//     instructions per 32-cycle MFMA); one register set, recycled vector by vector; ONE barrier per tile;
//   * tile loads are raw buffer loads (one descriptor per image): padding, ragged edges, channel tails and "no such tile"
//     (a zero-record descriptor) all come back as zeros from the range check -- no branches, so every k-step stays one
//     scheduling region; the per-vector offsets and LDS addresses are computed once per kernel, a tile costs 6 VALU per vector.
// BNB (plyolo_conv2d_wgrad_bn; SiLU units without a data gradient -- the first convolution of a network): the separate
// bn_act_bwd_dz pass of such a unit writes a dz that ONLY this kernel reads.  Here the dY vectors of a tile arrive as (dout, z) pairs
// and are turned into dz between the register set and LDS: dout and z are read once, dz never reaches HBM (3 E_out of traffic less).
template <int CO_T, int CI_T, int WK, int TH_, int SI, bool PRE, bool BNB = false, bool WIN = true>
__global__ __launch_bounds__(256) void conv_wgrad3_kernel(const WgP p) {
  constexpr int KS = 3, NTAPS = 9;
  constexpr int WCO = CO_T / 32, WCI = CI_T / 32;
  static_assert(WCO * WCI * WK == 4, "four waves per workgroup");
  constexpr int DZB = pitch_for(CO_T), XB = pitch_for(CI_T);
  constexpr int DZV = CO_T / 8, XV = CI_T / 8;
  constexpr int ITH_ = (TH_ - 1) * SI + KS, ITW_ = (TW - 1) * SI + KS;
  constexpr int DZ_BYTES = TH_ * TW * DZB, BUF = DZ_BYTES + ITH_ * ITW_ * XB;
  constexpr int NDV = TH_ * TW * DZV, DV = (NDV + 255) / 256;
  constexpr int NXV = ITH_ * ITW_ * XV, HV = (NXV + 255) / 256;
  constexpr int NV = DV + HV;
  constexpr int NJ = TH_ / WK;                       // k-steps (tile rows) of this wave
  constexpr int DUMP = BUF;             // 4 KB behind each buffer: where the lanes of a partial last vector write
  constexpr int BUFP = BUF + 4096;
  static_assert(2 * BUFP <= 160 * 1024, "two tiles must fit in LDS");
  static_assert(!PRE || 256 % XV == 0, "lazy input staging assumes a fixed channel vector per thread");
  extern __shared__ __align__(16) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wk = wave / (WCO * WCI), wco = (wave / WCI) % WCO, wci = wave % WCI;
  const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3;
  int wg = (int)blockIdx.x;
  if (p.xcd) {
    const int nwg = (int)gridDim.x, qq = nwg >> 3, r8 = nwg & 7, x = wg & 7;
    wg = (x < r8 ? x * (qq + 1) : r8 * (qq + 1) + (x - r8) * qq) + (wg >> 3);
  }
  const int split = p.xcd ? wg / p.nslabt : wg % p.S, slab_tile = p.xcd ? wg % p.nslabt : wg / p.S;
  const int co_tile = slab_tile / p.nci, ci_tile = slab_tile % p.nci;
  const int co0 = co_tile * CO_T, ci0 = ci_tile * CI_T;

  const int kpix = 8 * (g >> 1) + q;
  const int a_off = kpix * DZB + (wco * 32 + 16 * (g & 1) + 4 * pp) * 2;
  const int b_off = DZ_BYTES + kpix * SI * XB + (wci * 32 + 16 * (g & 1) + 4 * pp) * 2;

  f32x16 acc[NTAPS];
#pragma unroll
  for (int t = 0; t < NTAPS; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  // ---- tile-invariant state of the vectors this thread stages (vector v < DV: dY, else X halo)
  constexpr unsigned NEVER = 0x7fffu;          // a row coordinate no image has: the vector is never loaded
  int rel[NV], lds[NV];
  unsigned yx[NV];
#pragma unroll
  for (int v = 0; v < DV; ++v) {
    const int idx = tid + v * 256, m = idx / DZV, vv = idx - m * DZV, co = co0 + vv * 8;
    const bool live = idx < NDV && co < p.Cout;
    rel[v] = (((m >> 4) * p.OW + (m & 15)) * p.dy_ld + co) * 2;
    yx[v] = live ? (unsigned)(m >> 4) | ((unsigned)(m & 15) << 16) : NEVER;
    lds[v] = idx < NDV ? m * DZB + vv * 16 : DUMP + tid * 16;
  }
#pragma unroll
  for (int v = 0; v < HV; ++v) {
    const int idx = tid + v * 256, pix = idx / XV, vv = idx - pix * XV, iy = pix / ITW_, ix = pix - iy * ITW_, ci = ci0 + vv * 8;
    const bool live = idx < NXV && ci < p.Cin;
    rel[DV + v] = ((iy * p.W + ix) * p.x_ld + ci) * 2;
    yx[DV + v] = live ? (unsigned)iy | ((unsigned)ix << 16) : NEVER;
    lds[DV + v] = idx < NXV ? DZ_BYTES + pix * XB + vv * 16 : DUMP + tid * 16;
  }
  // BNB: z offsets of the dY vectors, the (scale, shift, A, B, Cc) of this thread's 8 output channels (the channel vector of a
  // thread is the same for all its dY vectors: 256 % DZV == 0), dgamma / dbeta from workgroup 0
  [[maybe_unused]] int relz[BNB ? DV : 1];
  [[maybe_unused]] float bsc[BNB ? 8 : 1], bsh[BNB ? 8 : 1], bA[BNB ? 8 : 1], bB[BNB ? 8 : 1], bC[BNB ? 8 : 1];
  if constexpr (BNB) {
    static_assert(256 % DZV == 0 && !PRE, "BNB: a fixed channel vector per thread");
#pragma unroll
    for (int v = 0; v < DV; ++v) {
      const int idx = tid + v * 256, m = idx / DZV, vv = idx - m * DZV;
      relz[v] = (((m >> 4) * p.OW + (m & 15)) * p.bz_ld + co0 + vv * 8) * 2;
    }
    float* tab = (float*)(smem + 2 * BUFP);      // [5][CO_T], built once per workgroup
    for (int ch = tid; ch < CO_T; ch += 256) {
      const int c = co0 + ch;
      float vsc = 0.f, vsh = 0.f, A = 0.f, B = 0.f, Cc = 0.f;
      if (c < p.Cout) {
        double su = 0.0, suz = 0.0;
#pragma unroll
        for (int sl = 0; sl < PLYOLO_STAT_SLOTS; ++sl) {
          su += p.bslots[((size_t)sl * 2 + 0) * p.Cout + c];
          suz += p.bslots[((size_t)sl * 2 + 1) * p.Cout + c];
        }
        const float mean = p.bcoef[2 * p.Cout + c], invstd = p.bcoef[3 * p.Cout + c];
        A = (p.bgamma ? p.bgamma[c] : 1.f) * invstd;
        B = (float)(-(double)A * (suz / p.bcount) * (double)invstd);
        Cc = (float)(-(double)A * (su / p.bcount) - (double)B * (double)mean);
        vsc = p.bcoef[c]; vsh = p.bcoef[p.Cout + c];
        if (wg == 0) {
          if (p.bdbeta) p.bdbeta[c] = (float)su;
          if (p.bdgamma) p.bdgamma[c] = (float)suz;
        }
      }
      tab[ch] = vsc; tab[CO_T + ch] = vsh; tab[2 * CO_T + ch] = A; tab[3 * CO_T + ch] = B; tab[4 * CO_T + ch] = Cc;
    }
    __syncthreads();
    const int ch0 = (tid % DZV) * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      bsc[i] = tab[ch0 + i]; bsh[i] = tab[CO_T + ch0 + i]; bA[i] = tab[2 * CO_T + ch0 + i]; bB[i] = tab[3 * CO_T + ch0 + i]; bC[i] = tab[4 * CO_T + ch0 + i];
    }
  }
  float psc[PRE ? 8 : 1], psh[PRE ? 8 : 1];
  if constexpr (PRE) {
    const int ci = ci0 + (tid % XV) * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      psc[i] = ci < p.Cin ? p.pre[ci + i] : 0.f;
      psh[i] = ci < p.Cin ? p.pre[p.pre_ld + ci + i] : 0.f;
    }
  }

  struct Tile {
    __amdgpu_buffer_rsrc_t rd, rx, rz;
    int oy0, ox0, iy0, ix0, dbase, xbase, zbase;
  };
  const int d_img = ((p.OH * p.OW - 1) * p.dy_ld + ((p.Cout + 7) & ~7)) * 2, x_img = ((p.H * p.W - 1) * p.x_ld + ((p.Cin + 7) & ~7)) * 2;
  auto tile_at = [&](int tile) {      // wave-uniform (blockIdx and kernel arguments only): lives in SGPRs
    Tile t;
    const bool real = tile < p.ntiles;
    const int tt = real ? tile : 0;
    const int txi = tt % p.tiles_x, t2 = tt / p.tiles_x, tyi = t2 % p.tiles_y, n = t2 / p.tiles_y;
    t.oy0 = tyi * TH_; t.ox0 = txi * TW;
    t.iy0 = t.oy0 * SI - p.pad; t.ix0 = t.ox0 * SI - p.pad;
    t.dbase = (t.oy0 * p.OW + t.ox0) * p.dy_ld * 2;
    t.xbase = (t.iy0 * p.W + t.ix0) * p.x_ld * 2;
    t.rd = __builtin_amdgcn_make_buffer_rsrc((void*)(p.dy + (size_t)n * p.OH * p.OW * p.dy_ld), 0, real ? d_img : 0, 0x00020000);
    t.rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (size_t)n * p.H * p.W * p.x_ld), 0, real ? x_img : 0, 0x00020000);
    if constexpr (BNB) {
      const int z_img = ((p.OH * p.OW - 1) * p.bz_ld + ((p.Cout + 7) & ~7)) * 2;
      t.zbase = (t.oy0 * p.OW + t.ox0) * p.bz_ld * 2;
      t.rz = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bz + (size_t)n * p.OH * p.OW * p.bz_ld), 0, real ? z_img : 0, 0x00020000);
    }
    return t;
  };
  u32x4 R[NV];
  [[maybe_unused]] u32x4 RZ[BNB ? DV : 1];
  unsigned hmask = 0u;                 // lazy input: which X vectors in R are real pixels (padding stays zero); BNB: which dY vectors are
  auto request = [&](const Tile& t, int v) {   // v: compile-time after unrolling
    const bool isx = v >= DV;
    const int y = (isx ? t.iy0 : t.oy0) + (int)(yx[v] & 0xffffu), x = (isx ? t.ix0 : t.ox0) + (int)(yx[v] >> 16);
    const bool ok = (unsigned)y < (unsigned)(isx ? p.H : p.OH) && (unsigned)x < (unsigned)(isx ? p.W : p.OW);
    const int off = ok ? (isx ? t.xbase : t.dbase) + rel[v] : (int)0x80000000;   // past any image: the range check returns zeros
    R[v] = __builtin_amdgcn_raw_buffer_load_b128(isx ? t.rx : t.rd, off, 0, 0);
    if constexpr (PRE) { if (isx) hmask = ok ? hmask | (1u << (v - DV)) : hmask & ~(1u << (v - DV)); }
    if constexpr (BNB) {
      if (!isx) {
        // (mask arithmetic, no second use of `ok` as a condition: with two conditional uses the compiler splits the request into an
        // if / else with a load on either side, and behind a branch the tile pipeline's vector-memory waits become full drains)
        const unsigned m = 0u - (unsigned)ok;
        RZ[v < DV ? v : 0] = __builtin_amdgcn_raw_buffer_load_b128(t.rz, (int)(((unsigned)(t.zbase + relz[v < DV ? v : 0]) & m) | (0x80000000u & ~m)), 0, 0);
        hmask = (hmask & ~(1u << v)) | ((1u << v) & m);           // an absent pixel (or tile) must stage 0, not Cc
      }
    }
  };
  auto stage = [&](unsigned char* buf, int v) {
    u32x4 t = R[v];
    if constexpr (BNB) {
      if (v < DV) {
        const u32x4 zz = RZ[v < DV ? v : 0];
        // an all-ones / all-zeros word ANDed in, not a select: the compiler turns `real ? q : 0` into a branch around the arithmetic,
        // and behind a branch every vector-memory wait of the tile pipeline becomes a wait for everything in flight
        const unsigned keep = 0u - ((hmask >> v) & 1u);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float zl = __uint_as_float(zz[i] << 16), zh = __uint_as_float(zz[i] & 0xffff0000u);
          const float dl = __uint_as_float(t[i] << 16), dh = __uint_as_float(t[i] & 0xffff0000u);
          const float dul = dl * wg_silu_grad(fmaf(zl, bsc[2 * i], bsh[2 * i]));
          const float duh = dh * wg_silu_grad(fmaf(zh, bsc[2 * i + 1], bsh[2 * i + 1]));
          const unsigned q = pack2bf(fmaf(bA[2 * i], dul, fmaf(bB[2 * i], zl, bC[2 * i])), fmaf(bA[2 * i + 1], duh, fmaf(bB[2 * i + 1], zh, bC[2 * i + 1])));
          t[i] = q & keep;
        }
      }
    }
    if constexpr (PRE) {
      if (v >= DV) {
        const bool real = (hmask >> (v - DV)) & 1u;
        {
          const u32x4 a_ = bn_act_vec8(t, psc, psh, p.pre_act);
#pragma unroll
          for (int i = 0; i < 4; ++i) t[i] = real ? a_[i] : 0u;
        }
      }
    }
    *(u32x4*)(buf + lds[v]) = t;     // lanes beyond the tile write to a dump row: no branch inside a k-step
  };

  // ---- prologue: tile 0 into buffer 0, tile 1 requested
  {
    // (requests in the order the main loop re-issues them -- by k-step -- so that its vmcnt waits are exact from the first tile on)
    const Tile t0 = tile_at(split);
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
      for (int v = jj; v < NV; v += NJ) request(t0, v);
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
      for (int v = jj; v < NV; v += NJ) stage(smem, v);
    const Tile t1 = tile_at((p.ablate & 2) ? p.ntiles : split + p.S);
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
      for (int v = jj; v < NV; v += NJ) request(t1, v);
  }
  __syncthreads();

  int cur = 0;
  for (int tile = split; tile < p.ntiles; tile += p.S) {
    const unsigned char* bc = smem + cur * BUFP;
    unsigned char* bn = smem + (cur ^ 1) * BUFP;
    const Tile t2 = tile_at((p.ablate & 2) ? p.ntiles : tile + 2 * p.S);
    const unsigned char* ap = bc + a_off;
    const unsigned char* bp = bc + b_off;
    s16x8 af0, af1, bf0[NTAPS], bf1[NTAPS];
    auto ldfrag = [&](int j, s16x8& af, s16x8 (&bfv)[NTAPS]) {
      const s16x4 lo = tr_read(ap + j * TW * DZB);
      const s16x4 hi = tr_read(ap + j * TW * DZB + 4 * DZB);
      af = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      if constexpr (SI == 1 && WIN) {
        // stride 1: the three taps of one kernel row read the SAME halo row shifted by 0 / 1 / 2 pixels.  A lane's fragment is 8
        // consecutive pixels of one channel, so a 12-pixel window (three transposed reads) holds all three: dx = 0 and dx = 2 are
        // register renames, dx = 1 is four v_alignbit -- 9 LDS reads per k-step for the X operand instead of 18.  The LDS read port
        // is what bounds this kernel (one fragment read per MFMA with a 32 x 32 block per wave and tap); the window's last two
        // pixels of the upper half-wave (columns 18, 19) are the next halo row's first two -- read, never used
#pragma unroll
        for (int dy = 0; dy < KS; ++dy) {
          const unsigned char* rp = bp + (j + dy) * ITW_ * XB;
          const s16x4 w0 = tr_read(rp), w1 = tr_read(rp + 4 * XB), w2 = tr_read(rp + 8 * XB);
          typedef __attribute__((ext_vector_type(2))) unsigned u32x2_;
          const u32x2_ a0 = *(const u32x2_*)&w0, a1 = *(const u32x2_*)&w1, a2 = *(const u32x2_*)&w2;
          const unsigned D0 = a0[0], D1 = a0[1], D2 = a1[0], D3 = a1[1], D4 = a2[0];
          const u32x4 t0 = {D0, D1, D2, D3}, t2 = {D1, D2, D3, D4};
          const u32x4 t1 = {__builtin_amdgcn_alignbit(D1, D0, 16), __builtin_amdgcn_alignbit(D2, D1, 16), __builtin_amdgcn_alignbit(D3, D2, 16),
                            __builtin_amdgcn_alignbit(D4, D3, 16)};
          bfv[dy * KS + 0] = *(const s16x8*)&t0;
          bfv[dy * KS + 1] = *(const s16x8*)&t1;
          bfv[dy * KS + 2] = *(const s16x8*)&t2;
        }
      } else {
#pragma unroll
      for (int t = 0; t < NTAPS; ++t) {
        const int o = ((j * SI + t / KS) * ITW_ + t % KS) * XB;
        const s16x4 l2 = tr_read(bp + o);
        const s16x4 h2 = tr_read(bp + o + 4 * SI * XB);
        bfv[t] = __builtin_shufflevector(l2, h2, 0, 1, 2, 3, 4, 5, 6, 7);
      }
      }
    };
    auto mm = [&](const s16x8& af, const s16x8 (&bfv)[NTAPS]) {
#pragma unroll
      for (int t = 0; t < NTAPS; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)&af, *(const bf16x8*)&bfv[t], acc[t], 0, 0, 0);
    };
    // the vectors of tile i+1 / i+2 handled beside k-step jj: stage R[v] into the other buffer, then re-request it
    auto side = [&](int jj) {
#pragma unroll
      for (int v = 0; v < NV; ++v)
        if (v % NJ == jj) {
          stage(bn, v);
          request(t2, v);
        }
    };
    auto pattern = [&]() {   // [MFMA + 5 LDS reads] x 4 (windowed X operand: 3 reads): the next k-step's fragments first; then the tile pipeline's share
      constexpr int SV = (NV + NJ - 1) / NJ;
      constexpr int RD = (SI == 1 && WIN) ? 3 : 5;
#pragma unroll
      for (int i = 0; i < NTAPS; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (i < 4) __builtin_amdgcn_sched_group_barrier(0x100, RD, 0);
        else if (i - 4 < SV) {
          __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    ldfrag(wk, af0, bf0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int jj = 0; jj < NJ; jj += 2) {
      const int j = wk + jj * WK;
      if (jj + 1 < NJ) ldfrag(j + WK, af1, bf1);
      mm(af0, bf0);
      side(jj);
      pattern();
      if (jj + 2 < NJ) ldfrag(j + 2 * WK, af0, bf0);
      if (jj + 1 < NJ) {
        mm(af1, bf1);
        side(jj + 1);
        pattern();
      }
    }
    __syncthreads();   // tile i consumed by every wave, tile i+1 complete in the other buffer
    cur ^= 1;
  }

  if (p.ablate & 1) { if (acc[0][0] == 123.456f) p.dw[0] = 1.f; return; }
  float* slab = p.dw + ((size_t)split * WK + wk) * ((size_t)KS * KS * p.Cout * p.Cin);
  const int r = lane & 31, h = lane >> 5;
  const int ci = ci0 + wci * 32 + r;
  if (ci < p.Cin) {
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int co = co0 + wco * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (co < p.Cout) slab[((size_t)t * p.Cout + co) * p.Cin + ci] = acc[t][i];
      }
  }
}

template <int CO_T, int CI_T, int WK, int TH_, int SI>
hipError_t launch_wg3(const WgP& p, int S, hipStream_t s) {
  constexpr int DZB = pitch_for(CO_T), XB = pitch_for(CI_T);
  const size_t lds = 2 * ((size_t)TH_ * TW * DZB + (size_t)((TH_ - 1) * SI + 3) * ((TW - 1) * SI + 3) * XB + 4096);
#ifdef PLYOLO_OPTIN
  auto kern = p.pre ? conv_wgrad3_kernel<CO_T, CI_T, WK, TH_, SI, true> : conv_wgrad3_kernel<CO_T, CI_T, WK, TH_, SI, false>;
#else
  auto kern = conv_wgrad3_kernel<CO_T, CI_T, WK, TH_, SI, false>;   // lazy inputs are refused at the C ABI (api.hip: check_conv)
  if constexpr (SI == 1) {      // PLYOLO_WG_WIN=0: the X fragments of every tap read from LDS on their own (round 2 .. 4; A/B switch)
    static const int win = getenv("PLYOLO_WG_WIN") ? atoi(getenv("PLYOLO_WG_WIN")) : 1;
    if (!win) kern = conv_wgrad3_kernel<CO_T, CI_T, WK, TH_, SI, false, false, false>;
  }
#endif
  if (hipError_t e = plyolo::ensure_dynamic_lds((const void*)kern, 160 * 1024); e != hipSuccess) return e;
  const int nco = (p.Cout + CO_T - 1) / CO_T;
  WgP q = p;
  q.nslabt = nco * p.nci;
  q.S = S;
  static const int xcd = getenv("PLYOLO_WG_XCD") ? atoi(getenv("PLYOLO_WG_XCD")) : 1;
  q.xcd = xcd;
  hipLaunchKernelGGL(kern, dim3(S * q.nslabt), dim3(256), lds, s, q);
  return hipGetLastError();
}

// BNB instances (plyolo_conv2d_wgrad_bn): the per-channel table sits behind the two tile buffers
template <int CO_T, int CI_T, int WK, int TH_>
hipError_t launch_wg3_bnb(const WgP& p, int S, hipStream_t s) {
  constexpr int SI = 1;
  constexpr int DZB = pitch_for(CO_T), XB = pitch_for(CI_T);
  const size_t lds = 2 * ((size_t)TH_ * TW * DZB + (size_t)((TH_ - 1) * SI + 3) * ((TW - 1) * SI + 3) * XB + 4096) + 5 * CO_T * 4;
  auto kern = conv_wgrad3_kernel<CO_T, CI_T, WK, TH_, SI, false, true>;
  if (hipError_t e = plyolo::ensure_dynamic_lds((const void*)kern, 160 * 1024); e != hipSuccess) return e;
  const int nco = (p.Cout + CO_T - 1) / CO_T;
  WgP q = p;
  q.nslabt = nco * p.nci;
  q.S = S;
  static const int xcd = getenv("PLYOLO_WG_XCD") ? atoi(getenv("PLYOLO_WG_XCD")) : 1;
  q.xcd = xcd;
  hipLaunchKernelGGL(kern, dim3(S * q.nslabt), dim3(256), lds, s, q);
  return hipGetLastError();
}

template <int CO_T, int CI_T, int KS, int WK, int MTC, int MTI, int TH_, int SI, int TRS = 1>
hipError_t launch_wg(const WgP& p, int S, hipStream_t s) {
  constexpr int DZB = pitch_for(CO_T), XB = pitch_for(CI_T);
  const size_t lds = (size_t)TH_ * TW * DZB + (size_t)((TH_ - 1) * SI + KS / TRS) * ((TW - 1) * SI + KS) * XB;
#ifdef PLYOLO_OPTIN
  auto kern = p.pre ? conv_wgrad_kernel<CO_T, CI_T, KS, WK, MTC, MTI, TH_, SI, true, TRS> : conv_wgrad_kernel<CO_T, CI_T, KS, WK, MTC, MTI, TH_, SI, false, TRS>;
#else
  auto kern = conv_wgrad_kernel<CO_T, CI_T, KS, WK, MTC, MTI, TH_, SI, false, TRS>;
#endif
  if (hipError_t e = plyolo::ensure_dynamic_lds((const void*)kern, 160 * 1024); e != hipSuccess) return e;
  const int nco = (p.Cout + CO_T - 1) / CO_T;
  WgP q = p;
  q.nslabt = nco * p.nci * TRS;
  q.S = S;
  static const int xcd = getenv("PLYOLO_WG_XCD") ? atoi(getenv("PLYOLO_WG_XCD")) : 1;
  q.xcd = xcd;
  hipLaunchKernelGGL(kern, dim3(S * q.nslabt), dim3(256), lds, s, q);
  return hipGetLastError();
}

}  // namespace

namespace plyolo {

hipError_t conv_wgrad1_launch(const void* x, const void* dz, float* dw, int M, int Cout, int Cin, int x_ld, int dz_ld, int G, hipStream_t s);   // conv_wgrad1.hip
// conv_wgrad1w.hip: wide pointwise layers (160+ channels on both sides) as a deep-K GEMM on 256 x 256 / 192 x 192 tiles
int conv_wgrad1w_tiles(int Cout, int Cin);
int conv_wgrad1w_tile(int Cout, int Cin);
hipError_t conv_wgrad1w_launch(const void* x, const void* dz, float* dw, int M, int Cout, int Cin, int x_ld, int dz_ld, int S, hipStream_t s);
// conv_wgrad3r.hip: the 64 x 64 3x3 stride-1 slab tile with twelve waves per workgroup (one kernel row per wave triple)
hipError_t conv_wgrad3r_launch(const void* wgp, int S, hipStream_t s);

struct WgPlan { WgP p; int id, S, WK, CO_T, CI_T, th, trs; };

static WgPlan plan_wgrad(const plyolo_conv_desc* d) {
  const int pad = (d->ksize - 1) / 2;
  WgPlan w{};
  WgP& p = w.p;
  p.N = d->N; p.H = d->H; p.W = d->W;
  p.OH = (d->H + 2 * pad - d->ksize) / d->stride + 1;
  p.OW = (d->W + 2 * pad - d->ksize) / d->stride + 1;
  p.Cin = d->Cin; p.Cout = (d->Cout + 7) & ~7;  // dy rows carry Cout rounded up to 8 channels (zeros)
  p.x_ld = d->x_ld; p.dy_ld = d->y_ld;
  p.pre = d->x_coef; p.pre_ld = d->x_coef_ld; p.pre_act = d->x_act;
  p.si = d->stride; p.pad = pad;
  if (d->ksize == 1 && d->stride == 1 && ((size_t)d->N * d->H * d->W) % TW == 0) {
    const int rows = (int)((size_t)d->N * d->H * d->W / TW);
    p.N = 1; p.H = rows; p.W = TW; p.OH = rows; p.OW = TW;
  }
  // ---- variant selection (slab shape, waves splitting k, tile rows)
  //  id 0: 3x3  64x64 slab                      (Cin >= 64, stride 1)
  //  id 1: 3x3 128x32 slab                      (stride 2 or Cin <= 32, Cout > 64): small halo tile
  //  id 2: 3x3  64x32 slab, 2 waves split k     (Cout <= 64, Cin <= 32 or stride 2)
  //  id 3: 3x3  32x32 slab, 4 waves split k     (Cout <= 32, Cin <= 32)
  //  id 4: 1x1  64x64 slab
  //  id 5: 1x1 128x128 slab, 64-pixel tiles     (both operands read once)
  const int true_cout = d->Cout;
  int th = TH;
  w.WK = 1;
  if (d->ksize == 3) {
    const bool narrow = d->stride == 2 || d->Cin <= 32;
    if (!narrow) { w.id = 0; w.CO_T = 64; w.CI_T = 64; }  // stride 1 only
    else if (true_cout <= 32 && d->Cin <= 32) { w.id = 3; w.CO_T = 32; w.CI_T = 32; w.WK = 4; }
    else if (true_cout <= 64) { w.id = 2; w.CO_T = 64; w.CI_T = 32; w.WK = 2; }
    else { w.id = 1; w.CO_T = 128; w.CI_T = 32; }
  } else {
    if (true_cout >= 128 && d->Cin >= 128 && d->stride == 1) { w.id = 5; w.CO_T = 128; w.CI_T = 128; th = 4; }
    else { w.id = 4; w.CO_T = 64; w.CI_T = 64; }
  }
  // small-channel stride-1 3x3 layers (huge pixel counts, tiny per-pixel rows): 16-row tiles halve the
  // per-tile barriers and prefetch hand-offs per pixel
  if ((w.id == 2 || w.id == 3) && d->stride == 1 && getenv("PLYOLO_WG_TH8") == nullptr) th = 16;
  w.th = th;
  p.ITH = (th - 1) * p.si + d->ksize;
  p.ITW = (TW - 1) * p.si + d->ksize;
  p.tiles_y = (p.OH + th - 1) / th;
  p.tiles_x = (p.OW + TW - 1) / TW;
  p.ntiles = p.N * p.tiles_y * p.tiles_x;
  p.nci = (p.Cin + w.CI_T - 1) / w.CI_T;
  const int nco = (p.Cout + w.CO_T - 1) / w.CO_T;
  // spatial split: a few workgroups per CU; every split writes a private fp32 slab (S*WK*|dW|
  // bytes stored once and read once by the unpack pass), capped at ~20 MB and 1024 slabs per layer
  const double dw_bytes = 4.0 * d->ksize * d->ksize * (double)d->Cout * d->Cin;
  int target = 768;  // workgroups per launch (3 per CU)
  if (const char* e = getenv("PLYOLO_WG_TARGET")) { const int v = atoi(e); if (v >= 64) target = v; }
  // tap-row split for the 64x64 3x3 variant (PLYOLO_WG_TRS=3): three workgroups per slab tile, one kernel row each
  // Measured (YOLOX-s B=32, same box, 3 alternations): the 64x64 launches 1.53 -> 1.23 ms, all weight-gradient launches
  // 3.37 -> 3.07 ms, step 10.34 -> 10.25 ms.  PLYOLO_WG_TRS=1 restores one workgroup per slab tile; =13 also splits the 128x32 variant.
  // Round 2, later: with two tiles in flight and the hand-pipelined MFMA phase the unsplit kernel is the better co-runner again
  // (a third of the L2 traffic: the three row workgroups each re-read the dY tile): its launches are longer (3.15 vs 2.95 ms per
  // step) but the step is shorter, 10.32 / 10.36 / 10.37 vs 10.49 / 10.55 / 10.52 ms on one box.  Default 1.
#ifdef PLYOLO_OPTIN
  static const int trs_env = getenv("PLYOLO_WG_TRS") ? atoi(getenv("PLYOLO_WG_TRS")) : 1;
#else
  constexpr int trs_env = 1;    // the tap-row split instances are an opt-in build (make OPTIN=1)
#endif
  w.trs = ((w.id == 0 && trs_env >= 3) || (w.id == 1 && trs_env == 13)) ? 3 : 1;
  int S = target / (nco * p.nci * w.WK * w.trs);
  if (S * w.WK > 1024) S = 1024 / w.WK;
  // measured on the whole step: slab stores + folds compete with the main lane for HBM, and a lighter weight-gradient launch is
  // the better co-runner even when it takes longer.  Round 1 (phase-alternating kernels): 20 MB best of 8 .. 64.  Round 2 (tile
  // pipeline inside the MFMA phase), mean of 4 alternations on one box: 4 MB 14.3 ms, 6 11.5, 8 10.8, 10 10.23, 12 10.25,
  // 16 10.08, 20 10.22, 28 10.36, 40 10.63 -- flat between 10 and 20.
  double budget = 16.0e6;
  // wide layers (FLOPs per operand byte k*k*Cin*Cout/(Cin+Cout) >= 1300: 320+ channels at 3x3) are MFMA-bound and sit
  // on the critical lane of the large models (YOLOX-x: the weight-gradient lane is the longer one): they get twice
  // the slabs so that their launches fill the chip (+3 % on YOLOX-x 1280, nothing on YOLOX-s whose widest 3x3 is 256)
  if ((double)d->ksize * d->ksize * d->Cin * d->Cout / (double)(d->Cin + d->Cout) >= 1300.0) budget = 40.0e6;
  if (const char* e = getenv("PLYOLO_WG_BUDGET_MB")) { const double v = atof(e); if (v >= 1.0) budget = v * 1.0e6; }
  if (w.id == 0 && w.trs == 1) if (const char* e = getenv("PLYOLO_WG_BUDGET0_MB")) { const double v = atof(e); if (v >= 1.0) budget = v * 1.0e6; }
  if (d->ksize == 1) if (const char* e = getenv("PLYOLO_WG_BUDGET1_MB")) { const double v = atof(e); if (v >= 1.0) budget = v * 1.0e6; }   // 1x1 layers only
  const int s_budget = (int)(budget / (dw_bytes * w.WK));
  if (S > s_budget) S = s_budget;
  // the 3x3 variants hold 144 accumulator registers: one workgroup per CU, a 257th would wait for a whole first round
  if (d->ksize == 3 && w.trs == 1 && S * nco * p.nci > 256) S = 256 / (nco * p.nci);
  if (S < 1) S = 1;
  if (S > p.ntiles) S = p.ntiles;
  //  id 6: 1x1 wide (160+ channels on both sides): 256 x 256 / 192 x 192 tiles, one workgroup per CU (conv_wgrad1w.hip; PLYOLO_WG1W=0: id 5)
  {
    static const int wg1w = getenv("PLYOLO_WG1W") ? atoi(getenv("PLYOLO_WG1W")) : 1;
    const int wt = (wg1w && d->ksize == 1 && d->stride == 1 && !d->x_coef) ? conv_wgrad1w_tiles(true_cout, d->Cin) : 0;
    if (wt) {
      w.id = 6;
      w.CO_T = w.CI_T = conv_wgrad1w_tile(true_cout, d->Cin);
      const int nstage = (int)(((size_t)d->N * d->H * d->W + 31) / 32);
      int cus = 256;
      if (const char* e = getenv("PLYOLO_WG1W_WGS")) { const int v = atoi(e); if (v >= 8) cus = v; }
      S = cus / wt;
      // slab bytes per layer (written once, read once by the fold).  Same box, two alternations, ms/step at 16 / 40 / 80 MB: YOLOX-x 1280 98.8 /
      // 99.3 / 99.3 (128 x 128 slabs: 100.3), YOLOv7 27.8 / 28.2 / - (28.4), YOLOX-l 17.3 / 17.5 / - (17.3): the lighter launch wins again
      double b1 = 16.0e6;
      if (const char* e = getenv("PLYOLO_WG1W_BUDGET_MB")) { const double v = atof(e); if (v >= 1.0) b1 = v * 1.0e6; }
      const int sb = (int)(b1 / dw_bytes);
      if (S > sb) S = sb;
      if (S < 1) S = 1;
      if (S > nstage / 4) S = nstage / 4 > 0 ? nstage / 4 : 1;      // at least four stages per range
    }
  }
  if (const char* e = getenv("PLYOLO_WG_S")) { const int v = atoi(e); if (v > 0) S = v < p.ntiles ? v : p.ntiles; }
  if (const char* e = getenv("PLYOLO_ABLATE_WG")) p.ablate = atoi(e);
  if (d->ksize == 1) if (const char* e = getenv("PLYOLO_ABLATE_WG1")) p.ablate = atoi(e);   // diagnostics (results are wrong): the 1x1 weight gradients only
  p.Cout = true_cout;  // slabs are [tap][Cout][Cin] with the TRUE Cout
  w.S = S;
  return w;
}

int conv_mfma_wgrad_slabs(const plyolo_conv_desc* d) {
  const WgPlan w = plan_wgrad(d);
  return w.S * w.WK;
}

int conv_mfma_wgrad(const plyolo_conv_desc* d, const void* x, const void* dy, float* dwp, void* stream) {
  WgPlan w = plan_wgrad(d);
  WgP p = w.p;
  p.x = (const bf16_t*)x;
  p.dy = (const bf16_t*)dy;
  p.dw = dwp;
  const int id = w.id, S = w.S, ks = d->ksize, trs = w.trs;
  const bool th16 = w.th == 16;
  {
    char lab[64];
    snprintf(lab, sizeof(lab), "conv_wgrad<%dx%d,k%d>%s", w.CO_T, w.CI_T, ks, p.pre ? "+bnact" : "");
    const double Mo = (double)p.N * p.OH * p.OW, Mi = (double)p.N * p.H * p.W;
    annotate(lab, 2.0 * Mo * d->Cout * d->Cin * ks * ks, (Mo * d->Cout + Mi * d->Cin) * 2.0 + 4.0 * ks * ks * d->Cout * d->Cin);
  }
  if (id == 6 && !p.ablate) {
    const int M = d->N * d->H * d->W, cout = d->Cout, cin = d->Cin, xl = d->x_ld, yl = d->y_ld;
    const void* xp = x; const void* dyp = dy;
    return submit(stream, [=](hipStream_t s) -> hipError_t { return conv_wgrad1w_launch(xp, dyp, dwp, M, cout, cin, xl, yl, S, s); });
  }
#ifdef PLYOLO_OPTIN
  // 1x1 stride 1 with 128+ channels on both sides: the persistent streaming kernel (conv_wgrad1.hip), opt-in (make OPTIN=1,
  // PLYOLO_WG1=1).  Round 4, same box: its launches are 28 % shorter (24 launches of YOLOX-s 0.631 -> 0.455 ms, 1.5 -> 2.1 TB/s) and
  // the step is LONGER -- YOLOX-s 8.548 vs 8.498 ms, YOLOX-x 1280 98.6 vs 97.5, YOLOX-l 16.79 vs 16.86: the weight-gradient lane is
  // not the critical one, and a launch that streams harder takes HBM bandwidth from the data-gradient chain beside it
  {
    const int wg1 = getenv("PLYOLO_WG1") ? atoi(getenv("PLYOLO_WG1")) : 0;
    if (wg1 && id == 5 && !p.pre && !p.ablate) {
      const int M = d->N * d->H * d->W, cout = d->Cout, cin = d->Cin, xl = d->x_ld, yl = d->y_ld;
      const void* xp = x; const void* dyp = dy;
      return submit(stream, [=](hipStream_t s) -> hipError_t { return conv_wgrad1_launch(xp, dyp, dwp, M, cout, cin, xl, yl, S, s); });
    }
  }
#endif
  // 3x3, unsplit: the round-2 kernel with the tile pipeline inside the MFMA phase (PLYOLO_WG3=0: the phase-alternating kernel);
  // its buffer descriptors address one image with 31-bit offsets
  static const int wg3_env = getenv("PLYOLO_WG3") ? atoi(getenv("PLYOLO_WG3")) : 1;
  const bool wg3 = wg3_env != 0 && ks == 3 && trs == 1 && (double)p.H * p.W * p.x_ld * 2.0 < 2.0e9 && (double)p.OH * p.OW * p.dy_ld * 2.0 < 2.0e9;
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    const bool s2 = p.si == 2;
    if (wg3) switch (id) {
      case 0: {
        // PLYOLO_WG3R=1: twelve waves per workgroup, one kernel row per wave triple (conv_wgrad3r.hip).  Measured EQUAL per launch (60-GF layer
        // 115.4 / 111.8 us, 15-GF layers 40.8 / 40.6, 36.6 / 36.7, 64.4 / 65.0) and not better in the step (YOLOX-s 8.319 / 8.345 ms, YOLOX-x
        // 96.2 / 97.1): three waves per SIMD change nothing, like halving the LDS reads did -- at the clock the chip holds under this load the
        // four-wave kernel's 72 MFMAs per tile are ~3/4 of its 1.86 us already.  Off.
        static const int wg3r = getenv("PLYOLO_WG3R") ? atoi(getenv("PLYOLO_WG3R")) : 0;
        if (wg3r && !p.pre && !p.ablate) return conv_wgrad3r_launch(&p, S, s);
        return launch_wg3<64, 64, 1, 8, 1>(p, S, s);
      }
      case 1: return s2 ? launch_wg3<128, 32, 1, 8, 2>(p, S, s) : launch_wg3<128, 32, 1, 8, 1>(p, S, s);
      case 2: return s2 ? launch_wg3<64, 32, 2, 8, 2>(p, S, s) : (th16 ? launch_wg3<64, 32, 2, 16, 1>(p, S, s) : launch_wg3<64, 32, 2, 8, 1>(p, S, s));
      case 3: return s2 ? launch_wg3<32, 32, 4, 8, 2>(p, S, s) : (th16 ? launch_wg3<32, 32, 4, 16, 1>(p, S, s) : launch_wg3<32, 32, 4, 8, 1>(p, S, s));
      default: break;
    }
    switch (id) {
      case 0:
#ifdef PLYOLO_OPTIN
        if (trs == 3) return launch_wg<64, 64, 3, 1, 1, 1, 8, 1, 3>(p, S, s);
#endif
        return launch_wg<64, 64, 3, 1, 1, 1, 8, 1>(p, S, s);
      case 1:
#ifdef PLYOLO_OPTIN
        if (trs == 3) return s2 ? launch_wg<128, 32, 3, 1, 1, 1, 8, 2, 3>(p, S, s) : launch_wg<128, 32, 3, 1, 1, 1, 8, 1, 3>(p, S, s);
#endif
        return s2 ? launch_wg<128, 32, 3, 1, 1, 1, 8, 2>(p, S, s) : launch_wg<128, 32, 3, 1, 1, 1, 8, 1>(p, S, s);
      case 2: return s2 ? launch_wg<64, 32, 3, 2, 1, 1, 8, 2>(p, S, s) : (th16 ? launch_wg<64, 32, 3, 2, 1, 1, 16, 1>(p, S, s) : launch_wg<64, 32, 3, 2, 1, 1, 8, 1>(p, S, s));
      case 3: return s2 ? launch_wg<32, 32, 3, 4, 1, 1, 8, 2>(p, S, s) : (th16 ? launch_wg<32, 32, 3, 4, 1, 1, 16, 1>(p, S, s) : launch_wg<32, 32, 3, 4, 1, 1, 8, 1>(p, S, s));
      case 4: return s2 ? launch_wg<64, 64, 1, 1, 1, 1, 8, 2>(p, S, s) : launch_wg<64, 64, 1, 1, 1, 1, 8, 1>(p, S, s);
      default: return launch_wg<128, 128, 1, 1, 2, 2, 4, 1>(p, S, s);
    }
  });
}


// 1 when plyolo_conv2d_wgrad_bn covers this unit: bf16, 3x3 stride 1, SiLU, at most 64 output and 32 input channels with the 16-row
// tiles (the first convolution of a CSPDarknet / ELAN stem)
int conv_mfma_wgrad_bn_fits(const plyolo_conv_desc* d, int act) {
  if (d->dtype != PLYOLO_BF16 || d->ksize != 3 || d->stride != 1 || act != PLYOLO_ACT_SILU || d->x_coef) return 0;
  if (getenv("PLYOLO_WG3") && atoi(getenv("PLYOLO_WG3")) == 0) return 0;
  const WgPlan w = plan_wgrad(d);
  if (!((w.id == 2 || w.id == 3) && w.th == 16 && w.trs == 1)) return 0;
  return ((double)d->H * d->W * d->x_ld * 2.0 < 2.0e9 && (double)w.p.OH * w.p.OW * d->y_ld * 2.0 < 2.0e9) ? 1 : 0;
}

// the weight gradient of a unit behind plyolo_bn_act_bwd_reduce with the unit's bn_act_bwd_dz in its loader (f->dz is ignored):
// same slabs as plyolo_bn_act_bwd_dz + plyolo_conv2d_wgrad, bit for bit; dgamma / dbeta are written by workgroup 0
int conv_mfma_wgrad_bn(const plyolo_conv_desc* d, const plyolo_bn_bwd_fuse* f, const void* x, float* dwp, void* stream) {
  WgPlan w = plan_wgrad(d);
  WgP p = w.p;
  p.x = (const bf16_t*)x;
  p.dy = (const bf16_t*)f->dout;
  p.dy_ld = f->dout_ld;
  p.dw = dwp;
  p.bz = (const bf16_t*)f->z; p.bz_ld = f->z_ld;
  p.bcoef = f->coef; p.bslots = f->bslots;
  p.bgamma = f->gamma; p.bdgamma = f->dgamma; p.bdbeta = f->dbeta;
  p.bcount = (double)d->N * p.OH * p.OW;
  const int id = w.id, S = w.S;
  {
    char lab[64];
    snprintf(lab, sizeof(lab), "conv_wgrad_bn<%dx%d,k3>", w.CO_T, w.CI_T);
    const double Mo = (double)p.N * p.OH * p.OW, Mi = (double)p.N * p.H * p.W;
    annotate(lab, 2.0 * Mo * d->Cout * d->Cin * 9.0, (2.0 * Mo * d->Cout + Mi * d->Cin) * 2.0 + 4.0 * 9.0 * d->Cout * d->Cin);
  }
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    if (id == 3) return launch_wg3_bnb<32, 32, 4, 16>(p, S, s);
    return launch_wg3_bnb<64, 32, 2, 16>(p, S, s);
  });
}

}  // namespace plyolo
