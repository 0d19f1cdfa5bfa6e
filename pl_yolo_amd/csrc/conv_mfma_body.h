// Body of the bf16 MFMA direct convolution (shared by conv_mfma.hip and conv_mfma_pre.hip).
// The lazy-input (PRE) instantiations live in their OWN translation unit: co-compiled template instances share the
// compiler's inlining / register-allocation context, and adding them to conv_mfma.hip made the unchanged non-PRE kernels
// 6 % (forward) to 14 % (3x3 layers) slower (measured on one box against the round-1 build: 2.47 -> 2.61 ms of forward
// launches per step) although their source and register counts were identical.
#pragma once
#include <stdlib.h>

#include "common.h"
#include "bnred.h"

namespace {

constexpr int TW = 16;

struct ConvP {
  const bf16_t* x;
  const bf16_t* w;   // fragment-native packed weights (see frag_index)
  void* y;
  const float* bias;
  const float* ep_coef;  // inference: fused BatchNorm (scale[Cout], shift[Cout]) + activation in the epilogue, or NULL
  int ep_act;
  const bf16_t* ep_res;  // fused epilogue only: residual added after the activation (Bottleneck shortcut), or NULL
  int ep_res_ld;
  double* stats;  // fp64 stat slots [PLYOLO_STAT_SLOTS][2][Cout] or NULL
  const float* pre;  // lazy input (plyolo_conv_desc::x_coef): the halo tile is staged as act(x * pre[c] + pre[pre_ld + c]); padding stays 0
  int pre_ld, pre_act;
  int N, H, W, Cin, Cout, x_ld, y_ld;
  int OHt, OWt;  // extent of the output position grid handled by this launch
  int OHf, OWf;  // full output tensor spatial dims
  int so, oy_off, ox_off;
  int si, iy_off, ix_off;
  int ITH, ITW;
  int db, bufsz;  // double-buffered halo variant (PLYOLO_DB, default on): enabled / bytes of one halo buffer
  int rowp;  // LDS pitch of one halo-tile image row (bytes): a multiple of 256 when si == 1, so that the two
             // image rows a 32-pixel A fragment spans land on disjoint banks (conflict-free ds_read_b128)
  int ntaps;
  int tiles_y, tiles_x, nmb;
  int accumulate;
  int nkb, nnb;  // packed weight geometry: 16-channel k-blocks, 32-channel n-blocks
  signed char tap_dy[9], tap_dx[9], tap_w[9];  // host-side table
  // the same table packed 8 bits per tap (dy | dx<<2 | w<<4): decoded with scalar shifts in the
  // kernel -- indexing a kernarg ARRAY with a runtime tap index makes hipcc emit VMEM byte loads,
  // whose s_waitcnt vmcnt(0) would also drain the in-flight weight prefetch every tap
  unsigned long long taps_lo;
  unsigned int taps_hi;
  int ablate;  // diagnostics (PLYOLO_ABLATE): 1 skip epilogue, 4 reload no halo after chunk 0, 8 skip MFMA, 16 skip weight loads, 32 skip LDS fragment reads
  plyolo_bn_red red;  // RED instances (data gradients): BatchNorm-backward reduction of the unit(s) whose output gradient this launch completes
  // BNB instances (plyolo_conv2d_dgrad_bn on a 3x3 unit, conv_mfma_bnb.hip): x is the gradient of the unit's ACTIVATED output (channels
  // >= bsplit in a second matrix for merged pairs); the halo tile is staged as dz = A*du + B*z + Cc, du = x * act'(z*sc + sh) -- exactly
  // bn_act_bwd_dz (bn.hip) --, padding stays 0, and the workgroups of output block 0 write the dz of their own tile's pixels once for
  // the weight gradient (and forward the shortcut's share of x, bfwd)
  const bf16_t* bz;      // raw conv output z of the unit, pitch bz_ld
  int bz_ld;
  const bf16_t* bx2;
  int bx2_ld, bsplit;
  const float* bcoef;    // (scale | shift | mean | invstd) [4][Cin] of the unit's forward
  const double* bslots;  // fp64 backward stat slots [PLYOLO_STAT_SLOTS][2][Cin] (sum du, sum du*zhat)
  const float *bgamma, *bgamma2;
  float *bdgamma, *bdbeta, *bdgamma2, *bdbeta2;
  int bpsplit;
  bf16_t* bdz;           // out: dz [N*H*W][bdz_ld], or NULL (nothing reads it: the weight gradient forms its own)
  int bdz_ld;
  bf16_t* bfwd;          // out: x copied here (the Bottleneck shortcut's gradient, first writer), or NULL
  int bfwd_ld;
  int btab;              // LDS byte offset of the per-channel table (behind the halo buffers)
  // FLAT instances (conv_mfma_flat.hip): a tile is `trows` FULL image rows of a `tw`-wide map (tw = 20 / 40), its pixels numbered row-major;
  // twinv = ceil(65536 / tw): row of pixel m = (m * twinv) >> 16 for every m of a tile
  int tw, trows, twinv;
};

DEVINL unsigned tap_code(const ConvP& p, int t) {
  return t < 8 ? (unsigned)((p.taps_lo >> (8 * t)) & 0xffull) : (p.taps_hi & 0xffu);
}

DEVINL u32x4 add_bf16x8(u32x4 a, u32x4 b) {
  u32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float lo = __uint_as_float(a[i] << 16) + __uint_as_float(b[i] << 16);
    float hi = __uint_as_float(a[i] & 0xffff0000u) + __uint_as_float(b[i] & 0xffff0000u);
    r[i] = pack2bf(lo, hi);
  }
  return r;
}

// One workgroup (4 waves) = TH x 16 output positions x BN output channels.  Every wave owns ONE
// 32-channel block of the output (so BN = 32 * WN) and MT = TH*16/(32*WM) pixel fragments.
//   A (pixels x channels): the input halo tile, staged once per CK-channel chunk in LDS and read
//     as tap-shifted ds_read_b128 fragments;
//   B (weights): never touches LDS -- the weights are packed on the host side of the step in MFMA
//     fragment order, so a wave's B fragment is ONE coalesced 1-KiB global load (L2/L1 resident),
//     prefetched one tap ahead in registers.  No per-tap barrier: waves only meet when the halo
//     tile is replaced, so MFMA, LDS reads and the loads of the co-resident workgroup overlap.
// up to four independent convolutions of one tile configuration in ONE launch (the four parity classes of a
// stride-2 data gradient): job j owns workgroups [start[j], start[j+1])
struct ConvJobs {
  ConvP c[4];
  int n;
  int start[5];
};

// BatchNorm + activation of the producing layer applied to one staged 16-byte vector (8 channels c .. c+7)
DEVINL u32x4 pre_apply(u32x4 t, const float* __restrict__ pre, int pre_ld, int act, int c) {
  const f32x4 s0 = *(const f32x4*)(pre + c), s1 = *(const f32x4*)(pre + c + 4);
  const f32x4 h0 = *(const f32x4*)(pre + pre_ld + c), h1 = *(const f32x4*)(pre + pre_ld + c + 4);
  const float sc[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
  const float sh[8] = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
  return bn_act_vec8(t, sc, sh, act);
}

// MF16 (PLYOLO_MFMA16=1, A/B instances): the same tile on v_mfma_f32_16x16x32_bf16 -- per wave 2*MT pixel fragments of 16 (one image
// row of the tile each) x two 16-channel halves, ONE 32-deep k-step per tap and 32-channel chunk.  Same MFMA cycles, same LDS
// and L2 bytes; MI355X_MICROARCH.md (DVFS item 7) measured this shape holding a higher clock under load.  B fragments come from
// the SAME weight pack (a 16-channel half x 32-deep fragment = four 256-byte pieces of two 32x16 fragments); the pixel pitch grows
// to CK*2+32 bytes, which makes the 16-pixel x 4-k-group ds_read_b128 conflict-free.
// S2 (conv_mfma_s2.hip, 3x3 stride-2 forward): 4 x 16 output positions, the (2*4+1) x 33 halo tile of a chunk double-buffered like
// the stride-1 tiles (descriptor loader, next chunk in flight during the taps) and stored with its COLUMNS DE-INTERLEAVED -- even
// halo columns first (17 entries), odd ones behind them (16) -- so that the 16 pixels of a fragment row, two input columns apart,
// are neighbours in LDS again (pitch CK*2+16: conflict-free ds_read_b128; side by side as they lie in the image they are 2*pitch
// apart and every read is 2-way conflicted).  Tap (dy, dx) reads region dx & 1 at column offset dx >> 1.
// RED (conv_mfma_red.hip, data gradients): the store loop also folds the BatchNorm-backward reduction of the upstream unit(s) (bnred.h)
// BNB (conv_mfma_bnb.hip, 3x3 stride-1 data gradients of SiLU units): the halo loader takes (dout, z) pairs and stages dz (ConvP::bz ...)
// FLAT (conv_mfma_flat.hip, 3x3 stride 1 on 20- / 40-wide maps): the TH*16 pixels of a tile are `trows` whole image rows instead of a TH x 16
// rectangle -- the 16-pixel fragments run over the row-major pixel list (per-lane LDS row addresses are free in this kernel: the tap shifts
// already use them), so a 20-wide map is covered at 100 % instead of 62 % (4 x 16 tiles) and a 40-wide one at 100 % instead of 83 %
template <int BN, int CK, int TH, bool OUT_F32, int ABL, bool DB = false, bool PRE = false, bool MF16 = false, bool S2 = false, bool RED = false, bool BNB = false,
          bool FLAT = false>
DEVINL void conv_mfma_body(const ConvP& p, const int bid, const int nwg, const int cb32 = -1) {
  static_assert(!RED || !OUT_F32, "RED instances store bf16 gradients");
  static_assert(!BNB || (!OUT_F32 && !PRE && !S2), "BNB instances: stride-1 bf16 data gradients");
  static_assert(!FLAT || (MF16 && !S2 && !BNB && !OUT_F32), "FLAT instances: stride-1 bf16 tiles on the 16x16x32 path");
  static_assert(!MF16 || (CK == 32 && !OUT_F32 && !PRE && DB), "MF16 instances: 32-channel double-buffered bf16 tiles");
  static_assert(!S2 || (DB && !MF16 && !PRE && !OUT_F32), "S2 instances: double-buffered bf16 tiles");
  constexpr int BM = TH * TW;
  constexpr int WN = BN / 32, WM = 4 / WN, MT = BM / (32 * WM);
  constexpr int NT = 64 * WN * WM;     // threads per workgroup: 256, or 192 for the 96-channel block (three waves, one 32-channel block each)
  static_assert(BN % 32 == 0 && WM >= 1, "whole 32-channel blocks per wave");
  constexpr int MT16 = FLAT ? BM / (16 * WM) : 2 * MT;   // MF16: 16-pixel fragments per wave ...
  constexpr int HALF16 = (MT16 + 1) / 2, HALF16B = MT16 - HALF16;   // ... and per software-pipeline half (5 = 3 + 2 on the 80-pixel FLAT tile)
  // pixel m of a tile -> (row, column) inside the tile
  auto tile_rc = [&](const int m, int& rr, int& cc) {
    if constexpr (FLAT) { rr = (int)(((unsigned)m * (unsigned)p.twinv) >> 16); cc = m - rr * p.tw; }
    else { rr = m >> 4; cc = m & 15; }
  };
  constexpr int ROWB = CK * 2 + (MF16 ? 32 : 16);  // LDS row pitch in bytes (pad: 16 B; MF16: 32 B)
  constexpr int CV = CK / 8;         // 16-byte vectors per row
  constexpr int KS = CK / 16;        // k-steps per chunk
  extern __shared__ __align__(16) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;

  // XCD-aware bijective remap: workgroups with equal (blockIdx.x % 8) share an XCD's
  // L2, give each such group a contiguous run of tiles so halo re-reads hit L2.
  int tile;
  {
    const int q = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    tile = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
  }
  const int txi = tile % p.tiles_x;
  const int t2 = tile / p.tiles_x;
  const int tyi = t2 % p.tiles_y;
  const int n = t2 / p.tiles_y;
  const int oy0 = tyi * (FLAT ? p.trows : TH), ox0 = txi * TW;
  const int iy0 = oy0 * p.si + p.iy_off, ix0 = ox0 * p.si + p.ix_off;
  // first 32-channel block of this workgroup's output columns: blockIdx.y column blocks of BN channels, or handed in by a kernel that mixes
  // block widths (conv_mfma_rag.hip: full 128-channel blocks + one narrower block for the remainder of a 160- / 320-channel layer)
  const int nb0 = cb32 >= 0 ? cb32 : (int)blockIdx.y * WN;
  const int cout0 = nb0 * 32;
  const int nb = nb0 + wn;  // this wave's 32-channel block of the packed weights

  int arow[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = (wm * MT + mt) * 32 + r;
    arow[mt] = S2 ? ((m >> 4) * 2) * p.rowp + (m & 15) * ROWB + h * 16 : ((m >> 4) * p.si) * p.rowp + ((m & 15) * p.si) * ROWB + h * 16;
  }

  f32x16 acc[MF16 ? 1 : MT];
#pragma unroll
  for (int mt = 0; mt < (MF16 ? 1 : MT); ++mt)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;
  // MF16: fragment j of this wave = image row (wm*MT16 + j) of the tile, pixel lane&15, k-group lane>>4
  [[maybe_unused]] int arow16[MF16 ? MT16 : 1];
  [[maybe_unused]] f32x4 acc16[MF16 ? MT16 : 1][2];
  if constexpr (MF16) {
#pragma unroll
    for (int j = 0; j < MT16; ++j) {
      if constexpr (FLAT) {
        int rr, cc;
        tile_rc((wm * MT16 + j) * 16 + (lane & 15), rr, cc);
        arow16[j] = rr * p.rowp + cc * ROWB + (lane >> 4) * 16;
      } else
      arow16[j] = ((wm * MT16 + j) * p.si) * p.rowp + ((lane & 15) * p.si) * ROWB + (lane >> 4) * 16;
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) acc16[j][hh] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }

  const int nchunks = (p.Cin + CK - 1) / CK;
  const int total = nchunks * p.ntaps;
  const bool nb_ok = nb < p.nnb;
  // fragment (tap, nb, kb): 64 lanes x 8 bf16, contiguous.  Addressing = uniform 64-bit base (SGPR pair: pack base +
  // tap + k-block, scalar arithmetic) + a per-lane 32-bit byte offset that never changes (n-block + lane): the
  // loads take the saddr + voffset form and cost no vector ALU work (the pack is < 2^31 bytes, checked on the host)
  const char* wbase = (const char*)p.w;
  const unsigned wvoff = (unsigned)((nb_ok ? nb : p.nnb - 1) * p.nkb) * 1024u + (unsigned)lane * 16u;
  const unsigned wtapB = (unsigned)p.nnb * (unsigned)p.nkb * 1024u;

  // weight fragments in flight: PD taps ahead.  PD = 2 on the 32-channel double-buffered variant (a tap there is only
  // 256 cycles of MFMA) measured 1.5 % SLOWER on the whole step in round 2 (+8 VGPRs, same occupancy); round 4: launches 2-5 %
  // shorter, step unchanged when every instance has it, -0.03 ms when only the translation units of the FORWARD launches do
  // (conv_mfma.hip, conv_mfma_s2.hip define PLYOLO_CONV_PD 2: the forward has no co-runner whose share of the CU the extra registers
  // cost; the plain, un-folded data gradients are instances of conv_mfma.hip too and share the setting -- a training plan launches few)
#ifndef PLYOLO_CONV_PD
#define PLYOLO_CONV_PD 1
#endif
  constexpr int PD = PLYOLO_CONV_PD;
  u32x4 bq[PD + 1][KS];
  // The (chunk, tap) of the next prefetch is tracked incrementally (deriving it from the phase counter cost a scalar
  // integer division -- ~20 SALU instructions -- per tap).  Past the last tap the stream parks on the last fragments:
  // the loads stay unconditional, a branch around them makes the waitcnt insertion fall back to vmcnt(0).
  int pf_chunk = 0, pf_t = 0;
  auto load_b = [&](u32x4* dst) {
    const char* wt = wbase + (size_t)((tap_code(p, pf_t) >> 4) * wtapB);
    if constexpr (MF16) {
      // lane (n = lane&15, kq = lane>>4) of the 16-channel half hh needs W[co = nb*32 + hh*16 + n][ci = chunk*32 + kq*8 + j]:
      // entry [nb][kb = 2*chunk + (kq>>1)][lane' = (kq&1)*32 + hh*16 + n] of the pack
      const int kq = lane >> 4;
      int kbl = pf_chunk * 2 + (kq >> 1);
      kbl = kbl < p.nkb ? kbl : p.nkb - 1;     // beyond Cin: zero-filled halo columns (see below)
      const unsigned base = (unsigned)((nb_ok ? nb : p.nnb - 1) * p.nkb + kbl) * 1024u + (unsigned)((kq & 1) * 32 + (lane & 15)) * 16u;
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) dst[hh] = *(const u32x4*)(wt + (size_t)(base + (unsigned)hh * 256u));
      if (pf_t + 1 < p.ntaps) ++pf_t;
      else if (pf_chunk + 1 < nchunks) { pf_t = 0; ++pf_chunk; }
      return;
    }
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      const int kb = pf_chunk * KS + kk;
      // No masking is needed: a k-block beyond Cin meets zero-filled halo columns (finite weights x 0 = 0), and a
      // wave whose 32-channel block lies beyond Cout only produces accumulators that the epilogue never stores.
      const unsigned kbc = (unsigned)(kb < p.nkb ? kb : p.nkb - 1);
      dst[kk] = *(const u32x4*)(wt + (size_t)(kbc * 1024u) + wvoff);
    }
    if (pf_t + 1 < p.ntaps) ++pf_t;
    else if (pf_chunk + 1 < nchunks) { pf_t = 0; ++pf_chunk; }
  };

  constexpr int abl = ABL;  // compile-time diagnostics switch (see ConvP::ablate)
  load_b(bq[0]);
  if (PD == 2) load_b(bq[1]);

  // ---- halo-tile loader.  The (pixel, channel-vector) -> (global offset, LDS offset) mapping of a thread's
  // vectors is the same for every Cin chunk, so it is computed ONCE per tile: in-kernel stamps showed the halo
  // phases at ~10k cycles each, most of it the per-vector index arithmetic (a runtime division by the tile
  // width, bounds tests, 64-bit addresses) repeated for the loads, again for the LDS stores, and per chunk.
  constexpr int HVT = S2 ? ((2 * TH + 1) * 33 * CV + NT - 1) / NT : (FLAT ? (6 * 42 * CV + NT - 1) / NT : ((TH + 2) * 18 * CV + NT - 1) / NT);   // vectors per thread of a 3x3 halo tile
  static_assert(NT % CV == 0, "a thread always owns the same channel vector");
  const int nvec = p.ITH * p.ITW * CV;
  const bool fastpath = nvec <= HVT * NT;                 // stride-2 forward tiles take the generic loop
  const bf16_t* xn = p.x + (size_t)n * p.H * p.W * p.x_ld;
  const int cvt = tid % CV;                                // 256 % CV == 0: a thread always owns the same channel vector
  int goff[HVT], loff[HVT];
  [[maybe_unused]] unsigned own = 0u;   // BNB: bit v = vector v is a pixel of this tile's own output rows / columns (its dz is written here)
  if (fastpath) {
#pragma unroll
    for (int v = 0; v < HVT; ++v) {
      const int idx = tid + v * NT;
      goff[v] = -1;
      loff[v] = -1;
      if (idx < nvec) {
        const int pix = idx / CV;
        const int iy = pix / p.ITW, ix = pix - iy * p.ITW;
        const int gy = iy0 + iy, gx = ix0 + ix;
        loff[v] = iy * p.rowp + (S2 ? ((ix & 1) ? 17 + (ix >> 1) : (ix >> 1)) : ix) * ROWB + cvt * 16;
        if constexpr (BNB) {
          // the pixel index: the loader addresses up to four matrices of different pitch with it
          if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) {
            goff[v] = gy * p.W + gx;
            if (gy >= oy0 && gy < oy0 + TH && gx >= ox0 && gx < ox0 + TW) own |= 1u << v;
          }
        } else {
          if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) goff[v] = (gy * p.W + gx) * p.x_ld + cvt * 8;
        }
      }
    }
  }
  // loads are ALWAYS issued (border / out-of-range vectors read the image's first 16 bytes and are replaced by
  // zeros when they are written to LDS): a fixed number of VMEM instructions per call keeps the compiler's vmcnt
  // bookkeeping exact, which the double-buffered variant below depends on
  auto halo_load = [&](const int c0, u32x4* hv) {
    const bool cok = c0 + cvt * 8 < p.Cin;
#pragma unroll
    for (int v = 0; v < HVT; ++v) hv[v] = *(const u32x4*)(xn + ((goff[v] >= 0 && cok) ? goff[v] + c0 : 0));
  };
  auto halo_store = [&](const int c0, const u32x4* hv, const int boff) {
    const bool cok = c0 + cvt * 8 < p.Cin;
    const u32x4 zero = {0u, 0u, 0u, 0u};
    if constexpr (PRE) {
      // lazy input: the 8 channels of this thread's vectors are the same for the whole chunk; their coefficients are
      // read once (L1/L2 hits) and the affine + activation runs between the global load and the LDS write
      const int c = cok ? c0 + cvt * 8 : 0;
#pragma unroll
      for (int v = 0; v < HVT; ++v)
        if (loff[v] >= 0) *(u32x4*)(smem + boff + loff[v]) = (goff[v] >= 0 && cok) ? pre_apply(hv[v], p.pre, p.pre_ld, p.pre_act, c) : zero;
    } else {
#pragma unroll
      for (int v = 0; v < HVT; ++v)
        if (loff[v] >= 0) *(u32x4*)(smem + boff + loff[v]) = (goff[v] >= 0 && cok) ? hv[v] : zero;
    }
  };
  // ---- BNB loader: (dout, z) vector pairs -> dz.  Requests are unconditional like the plain loader's; the dz / shortcut stores
  // are raw buffer stores whose offset is pushed out of range for the vectors that are not written (dropped by the hardware):
  // no branch between the vector-memory instructions of the tap loop, the compiler's vmcnt bookkeeping stays exact
  [[maybe_unused]] const size_t img = (size_t)n * p.H * p.W;
  // every matrix of the loader is addressed through a per-image buffer descriptor (uniform) + a 32-bit byte offset per vector:
  // out-of-image / out-of-range vectors get an offset beyond the descriptor's extent -- loads return zeros, stores are dropped
  [[maybe_unused]] const unsigned ibytes = (unsigned)(p.H * p.W) * 2u;   // bytes of one image per channel of pitch
  [[maybe_unused]] auto halo_load_bnb = [&](const int c0, u32x4* hv, u32x4* zv) {
    if constexpr (BNB) {
      const int c = c0 + cvt * 8;
      const bool cok = c < p.Cin;
      const bool second = p.bsplit > 0 && c0 >= p.bsplit;     // wave-uniform: the split is a multiple of the chunk (checked on the host)
      const bf16_t* dbase = second ? p.bx2 + img * p.bx2_ld : xn;
      const int dld = second ? p.bx2_ld : p.x_ld;
      const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)dbase, 0, (int)(ibytes * (unsigned)dld), 0x00020000);
      const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bz + img * p.bz_ld), 0, (int)(ibytes * (unsigned)p.bz_ld), 0x00020000);
      const int cd2 = (second ? c - p.bsplit : c) * 2, dld2 = dld * 2, zld2 = p.bz_ld * 2;
#pragma unroll
      for (int v = 0; v < HVT; ++v) {
        const bool ok = goff[v] >= 0 && cok;
        hv[v] = __builtin_amdgcn_raw_buffer_load_b128(rd, ok ? goff[v] * dld2 + cd2 : (int)0x80000000, 0, 0);
        zv[v] = __builtin_amdgcn_raw_buffer_load_b128(rz, ok ? goff[v] * zld2 + c * 2 : (int)0x80000000, 0, 0);
      }
    }
  };
  [[maybe_unused]] auto halo_store_bnb = [&](const int c0, const u32x4* hv, u32x4* zv, const int boff) {
    if constexpr (BNB) {
      const int c = c0 + cvt * 8;
      const bool cok = c < p.Cin;
      const u32x4 zero = {0u, 0u, 0u, 0u};
      const float* tab = (const float*)(smem + p.btab);
      const int Kp = p.Cin, cc = cok ? c : 0;
      const bool wr = blockIdx.y == 0 && cok;
      const __amdgpu_buffer_rsrc_t rdz = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bdz + img * p.bdz_ld), 0, p.bdz ? (int)(ibytes * (unsigned)p.bdz_ld) : 0, 0x00020000);
      const __amdgpu_buffer_rsrc_t rfw = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bfwd + img * p.bfwd_ld), 0, p.bfwd ? (int)(ibytes * (unsigned)p.bfwd_ld) : 0, 0x00020000);
      // the shortcut's copy first: its data is the untouched request
#pragma unroll
      for (int v = 0; v < HVT; ++v) {
        const bool w = wr && goff[v] >= 0 && ((own >> v) & 1u);
        __builtin_amdgcn_raw_buffer_store_b128(hv[v], rfw, w ? (goff[v] * p.bfwd_ld + c) * 2 : (int)0x80000000, 0, 0);
      }
      // one channel PAIR (a dword of every vector) at a time, ten coefficients from the table per pair, results in place of the z
      // vectors: the kernel sits at the three-waves-per-SIMD edge (whole-vector coefficient sets and a third register set spilled)
      typedef __attribute__((ext_vector_type(2))) float f32x2;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x2 sc = *(const f32x2*)(tab + cc + 2 * i), sh = *(const f32x2*)(tab + Kp + cc + 2 * i), A = *(const f32x2*)(tab + 2 * Kp + cc + 2 * i),
                    B = *(const f32x2*)(tab + 3 * Kp + cc + 2 * i), Cc = *(const f32x2*)(tab + 4 * Kp + cc + 2 * i);
#pragma unroll
        for (int v = 0; v < HVT; ++v) {
          const unsigned zz = zv[v][i], dd = hv[v][i];
          const float zl = __uint_as_float(zz << 16), zh = __uint_as_float(zz & 0xffff0000u);
          const float dl = __uint_as_float(dd << 16), dh = __uint_as_float(dd & 0xffff0000u);
          const float dul = dl * bnred_act_grad(fmaf(zl, sc[0], sh[0]), PLYOLO_ACT_SILU);
          const float duh = dh * bnred_act_grad(fmaf(zh, sc[1], sh[1]), PLYOLO_ACT_SILU);
          zv[v][i] = pack2bf(fmaf(A[0], dul, fmaf(B[0], zl, Cc[0])), fmaf(A[1], duh, fmaf(B[1], zh, Cc[1])));
        }
      }
#pragma unroll
      for (int v = 0; v < HVT; ++v) {
        const bool ok = goff[v] >= 0 && cok;
        if (loff[v] >= 0) *(u32x4*)(smem + boff + loff[v]) = ok ? zv[v] : zero;
        const bool w = wr && ok && ((own >> v) & 1u);
        __builtin_amdgcn_raw_buffer_store_b128(zv[v], rdz, w ? (goff[v] * p.bdz_ld + c) * 2 : (int)0x80000000, 0, 0);
      }
    }
  };
  // per-channel table (scale, shift, A, B, Cc) of the unit, built by every workgroup from the fp64 slots while its first halo
  // vectors are in flight (conv_pw.hip does the same); tile 0 of output block 0 publishes dgamma / dbeta as bn_act_bwd_dz does
  [[maybe_unused]] auto bnb_table = [&]() {
    if constexpr (BNB) {
      float* tab = (float*)(smem + p.btab);
      const int Kp = p.Cin;
      const double cnt = (double)p.N * p.H * p.W;
      for (int ch = tid; ch < Kp; ch += NT) {
        double su = 0.0, suz = 0.0;
#pragma unroll
        for (int sl = 0; sl < PLYOLO_STAT_SLOTS; ++sl) {
          su += p.bslots[((size_t)sl * 2 + 0) * Kp + ch];
          suz += p.bslots[((size_t)sl * 2 + 1) * Kp + ch];
        }
        const float mean = p.bcoef[2 * Kp + ch], invstd = p.bcoef[3 * Kp + ch];
        const bool sec = p.bpsplit > 0 && ch >= p.bpsplit;
        const int cp = sec ? ch - p.bpsplit : ch;
        const float* gam = sec ? p.bgamma2 : p.bgamma;
        const float A = (gam ? gam[cp] : 1.f) * invstd;
        const float B = (float)(-(double)A * (suz / cnt) * (double)invstd);
        tab[ch] = p.bcoef[ch];
        tab[Kp + ch] = p.bcoef[Kp + ch];
        tab[2 * Kp + ch] = A;
        tab[3 * Kp + ch] = B;
        tab[4 * Kp + ch] = (float)(-(double)A * (su / cnt) - (double)B * (double)mean);
        if (tile == 0 && blockIdx.y == 0) {
          float* db_ = sec ? p.bdbeta2 : p.bdbeta;
          float* dg_ = sec ? p.bdgamma2 : p.bdgamma;
          if (db_) db_[cp] = (float)su;
          if (dg_) dg_[cp] = (float)suz;
        }
      }
    }
  };
  // generic loader (stride-2 forward tiles): batches of HV 16-byte loads in flight before the first LDS write
  auto halo_generic = [&](const int c0) {
    constexpr int HV = 6;
    for (int base = 0; base < nvec; base += HV * NT) {
      u32x4 hv[HV];
#pragma unroll
      for (int v = 0; v < HV; ++v) {
        const int idx = base + tid + v * NT;
        u32x4 val = {0u, 0u, 0u, 0u};
        if (idx < nvec) {
          const int pix = idx / CV, cv = idx - pix * CV;
          const int iy = pix / p.ITW, ix = pix - iy * p.ITW;
          const int gy = iy0 + iy, gx = ix0 + ix, c = c0 + cv * 8;
          if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W && c < p.Cin) {
            val = *(const u32x4*)(xn + ((size_t)gy * p.W + gx) * p.x_ld + c);
            if constexpr (PRE) val = pre_apply(val, p.pre, p.pre_ld, p.pre_act, c);
          }
        }
        hv[v] = val;
      }
#pragma unroll
      for (int v = 0; v < HV; ++v) {
        const int idx = base + tid + v * NT;
        if (idx < nvec) {
          const int pix = idx / CV, cv = idx - pix * CV;
          const int iy = pix / p.ITW, ix = pix - iy * p.ITW;
          *(u32x4*)(smem + iy * p.rowp + ix * ROWB + cv * 16) = hv[v];
        }
      }
    }
  };

  int phase = 0;
  // one filter tap of the current chunk: KS k-steps x MT MFMAs on the halo tile at LDS offset `boff`;
  // `extra_loads` is issued right after the weight prefetch (see the double-buffered loop)
  auto run_tap = [&](const int t, const int boff, auto&& extra_loads) {
    // unconditional (the last tap re-loads its own fragments): a branch around the loads makes the waitcnt
    // insertion fall back to vmcnt(0) at the join
    if (!(abl & 16)) load_b(bq[PD]);
    extra_loads();
    // keep the prefetch ABOVE the MFMA block: left alone, the scheduler sinks these loads to the end of the tap
    // (shorter live range) where the next tap's s_waitcnt vmcnt(0) exposes their full latency
    __builtin_amdgcn_sched_barrier(0);
    const unsigned tc = tap_code(p, t);
    const int tdx = (int)((tc >> 2) & 3u);
    const int toff = boff + (int)(tc & 3u) * p.rowp + (S2 ? ((tdx & 1) ? 17 : 0) + (tdx >> 1) : tdx) * ROWB;
    // software pipeline over the k-steps: the A fragments of step kk+1 are requested from LDS before the MFMAs
    // of step kk are issued, so that no MFMA waits on a ds_read issued just before it
    if constexpr (MF16) {
      // two halves of HALF16 fragments: the second half's A fragments are requested from LDS before the first half's MFMAs
      const bf16x8 b0 = *(const bf16x8*)&bq[0][0], b1 = *(const bf16x8*)&bq[0][1];
      if constexpr (BNB && MT16 >= 8) {
        // BNB instances with 8 fragments per wave hold the next chunk's (dout, z) vectors through the taps: the second half's A fragments are
        // requested in QUARTERS behind the first MFMAs instead of as a second register set (16 VGPRs: the three-waves-per-SIMD edge)
        constexpr int Q = HALF16 / 2;
        bf16x8 a16[Q], an16[Q];
#pragma unroll
        for (int j = 0; j < Q; ++j) a16[j] = *(const bf16x8*)(smem + arow16[j] + toff);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (q < 3) {
#pragma unroll
            for (int j = 0; j < Q; ++j) an16[j] = *(const bf16x8*)(smem + arow16[(q + 1) * Q + j] + toff);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < Q; ++j) {
            acc16[q * Q + j][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a16[j], b0, acc16[q * Q + j][0], 0, 0, 0);
            acc16[q * Q + j][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a16[j], b1, acc16[q * Q + j][1], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (q < 3) {
#pragma unroll
            for (int j = 0; j < Q; ++j) a16[j] = an16[j];
          }
        }
      } else if constexpr (FLAT) {
      // uneven halves (5 fragments = 3 + 2 on the 80-pixel tile, 10 = 5 + 5 on the 160-pixel one), ONE fragment set: the second half is
      // requested behind the first half's MFMAs (a second set costs the 160-pixel tile its third wave per SIMD: 179 -> 159 VGPRs)
      bf16x8 a16[HALF16];
#pragma unroll
      for (int j = 0; j < HALF16; ++j) a16[j] = *(const bf16x8*)(smem + arow16[j] + toff);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < HALF16; ++j) {
        acc16[j][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a16[j], b0, acc16[j][0], 0, 0, 0);
        acc16[j][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a16[j], b1, acc16[j][1], 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < HALF16B; ++j) a16[j] = *(const bf16x8*)(smem + arow16[HALF16 + j] + toff);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < HALF16B; ++j) {
        acc16[HALF16 + j][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a16[j], b0, acc16[HALF16 + j][0], 0, 0, 0);
        acc16[HALF16 + j][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a16[j], b1, acc16[HALF16 + j][1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      } else {
      bf16x8 a16[HALF16], an16[HALF16];
#pragma unroll
      for (int j = 0; j < HALF16; ++j) a16[j] = *(const bf16x8*)(smem + arow16[j] + toff);
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        if (half == 0) {
#pragma unroll
          for (int j = 0; j < HALF16; ++j) an16[j] = *(const bf16x8*)(smem + arow16[HALF16 + j] + toff);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < HALF16; ++j) {
          acc16[half * HALF16 + j][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a16[j], b0, acc16[half * HALF16 + j][0], 0, 0, 0);
          acc16[half * HALF16 + j][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a16[j], b1, acc16[half * HALF16 + j][1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (half == 0) {
#pragma unroll
          for (int j = 0; j < HALF16; ++j) a16[j] = an16[j];
        }
      }
      }
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
#pragma unroll
        for (int d = 0; d < PD; ++d) bq[d][kk] = bq[d + 1][kk];
      }
      ++phase;
      return;
    }
    constexpr bool PIPE = MT <= 4;   // the 8-fragment tile has no registers left for a second fragment set
    bf16x8 a[MT], an[PIPE ? MT : 1];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a[mt] = *(const bf16x8*)(smem + arow[mt] + toff);
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      const bf16x8 b = *(const bf16x8*)&bq[0][kk];
      if (PIPE && kk + 1 < KS) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) an[PIPE ? mt : 0] = *(const bf16x8*)(smem + arow[mt] + toff + (kk + 1) * 32);
      }
      // fence: all reads of step kk+1 are issued BEFORE the MFMAs of step kk (counted lgkmcnt then lets the MFMAs
      // start while those reads are still in flight); without it the scheduler pairs reads with the MFMAs again
      if (PIPE) __builtin_amdgcn_sched_barrier(0);
      if (!(abl & 8)) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt], b, acc[mt], 0, 0, 0);
      } else {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][0] += (float)a[mt][0];
      }
      if (PIPE) __builtin_amdgcn_sched_barrier(0);
      if (kk + 1 < KS) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[mt] = PIPE ? an[PIPE ? mt : 0] : *(const bf16x8*)(smem + arow[mt] + toff + (kk + 1) * 32);
      }
    }
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
#pragma unroll
      for (int d = 0; d < PD; ++d) bq[d][kk] = bq[d + 1][kk];
    }
    ++phase;
  };

  if constexpr (DB) {
    // Double-buffered halo tile: the next chunk's vectors are requested right after tap 0's weight prefetch, stay
    // in flight for the whole chunk (vmcnt is in-order: the first wait that covers them is tap 2's wait for its
    // weights) and are written to the OTHER LDS buffer after the last tap.  One barrier per chunk, no exposed
    // global-load latency between chunks.
    u32x4 hv[HVT];
    [[maybe_unused]] u32x4 zv[BNB ? HVT : 1];
    if constexpr (BNB) {
      halo_load_bnb(0, hv, zv);
      bnb_table();
      __syncthreads();
      halo_store_bnb(0, hv, zv, 0);
    } else {
      halo_load(0, hv);
      halo_store(0, hv, 0);
    }
    __syncthreads();
    for (int chunk = 0; chunk < nchunks; ++chunk) {
      const int boff = (chunk & 1) * p.bufsz;
      const int cnext = (chunk + 1) * CK;   // past Cin on the last chunk: every load collapses to the dummy address
      run_tap(0, boff, [&]() {
        if constexpr (BNB) halo_load_bnb(cnext, hv, zv);
        else halo_load(cnext, hv);
      });
      for (int t = 1; t < p.ntaps; ++t) run_tap(t, boff, []() {});
      if (chunk + 1 < nchunks) {
        if constexpr (BNB) halo_store_bnb(cnext, hv, zv, p.bufsz - boff);
        else halo_store(cnext, hv, p.bufsz - boff);
      }
      __syncthreads();  // next buffer complete; every wave is done with this one
    }
  } else {
    for (int chunk = 0; chunk < nchunks; ++chunk) {
      const int c0 = chunk * CK;
      __syncthreads();  // every wave is done reading the previous chunk's halo tile
      if (!(abl & 4) || chunk == 0) {
        if constexpr (BNB) {   // (the launcher only takes tiles the descriptor loader covers)
          u32x4 hv[HVT], zv[HVT];
          halo_load_bnb(c0, hv, zv);
          if (chunk == 0) {
            bnb_table();
            __syncthreads();
          }
          halo_store_bnb(c0, hv, zv, 0);
        } else if (fastpath) {
          u32x4 hv[HVT];
          halo_load(c0, hv);
          halo_store(c0, hv, 0);
        } else {
          halo_generic(c0);
        }
      }
      __syncthreads();  // halo tile visible
      for (int t = 0; t < p.ntaps; ++t) run_tap(t, 0, []() {});
    }
  }
  __syncthreads();  // all LDS operand reads retired; LDS is reused for the epilogue
  if (abl & 1) {  // keep every accumulator live, skip the epilogue
    float t = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int i = 0; i < 16; ++i) t += acc[mt][i];
    if (t == 12345.f) ((float*)p.y)[0] = t;
    return;
  }

  // ---- epilogue ---------------------------------------------------------------
  constexpr int SROW = OUT_F32 ? (BN + 4) * 4 : (BN * 2 + 16);  // staging row pitch (bytes)
  float* red = (float*)(smem + BM * SROW);                       // [WM][2][BN]

  if constexpr (MF16) {
    // accumulator (j, hh)[i]: pixel row (wm*MT16 + j) of the tile, pixel (lane>>4)*4 + i, channel wn*32 + hh*16 + (lane&15)
    const int n = lane & 15, kq = lane >> 4;
    if (p.stats != nullptr) {
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < MT16; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            bool valid;
            if constexpr (FLAT) {
              int rr, cc;
              tile_rc((wm * MT16 + j) * 16 + kq * 4 + i, rr, cc);
              valid = oy0 + rr < p.OHt;      // (every column of a full-width row is a pixel)
            } else
            valid = (oy0 + wm * MT16 + j < p.OHt) && (ox0 + kq * 4 + i < p.OWt);
            const float v = valid ? acc16[j][hh][i] : 0.f;
            s1 += v;
            s2 = fmaf(v, v, s2);
          }
        s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
        s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
        if (kq == 0) {
          red[(wm * 2 + 0) * BN + wn * 32 + hh * 16 + n] = s1;
          red[(wm * 2 + 1) * BN + wn * 32 + hh * 16 + n] = s2;
        }
      }
    }
    const bool fused16 = p.ep_coef != nullptr;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int col = wn * 32 + hh * 16 + n;
      float sc = 1.f, sh = 0.f;
      if (fused16 && cout0 + col < p.Cout) { sc = p.ep_coef[cout0 + col]; sh = p.ep_coef[p.Cout + cout0 + col]; }
#pragma unroll
      for (int j = 0; j < MT16; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int m = (wm * MT16 + j) * 16 + kq * 4 + i;
          const float v = acc16[j][hh][i];
          *(bf16_t*)(smem + m * SROW + col * 2) = f2bf(fused16 ? act_fwd_core(fmaf(v, sc, sh), p.ep_act) : v);
        }
    }
  } else {
  if (p.stats != nullptr && !(abl & 64)) {
    float s1 = 0.f, s2 = 0.f;
    if (oy0 + TH <= p.OHt && ox0 + TW <= p.OWt) {
      // interior tile (the common case): no per-element validity arithmetic
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          s1 += acc[mt][i];
          s2 = fmaf(acc[mt][i], acc[mt][i], s2);
        }
    } else {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int m = (wm * MT + mt) * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
          const bool valid = (oy0 + (m >> 4) < p.OHt) && (ox0 + (m & 15) < p.OWt);
          const float v = valid ? acc[mt][i] : 0.f;
          s1 += v;
          s2 = fmaf(v, v, s2);
        }
    }
    s1 += __shfl_xor(s1, 32);
    s2 += __shfl_xor(s2, 32);
    if (h == 0) {
      red[(wm * 2 + 0) * BN + wn * 32 + r] = s1;
      red[(wm * 2 + 1) * BN + wn * 32 + r] = s2;
    }
  }
  // inference-mode BaseConv: BatchNorm is a fixed per-channel affine, so BN + activation are applied to the
  // accumulators here and the activated tensor is the ONLY thing written (no z, no bn_act launch)
  const bool fused = !OUT_F32 && p.ep_coef != nullptr;
  float ep_sc = 1.f, ep_sh = 0.f;
  if (fused) {
    const int co = cout0 + wn * 32 + r;
    if (co < p.Cout) { ep_sc = p.ep_coef[co]; ep_sh = p.ep_coef[p.Cout + co]; }
  }
  if (!(abl & 256))
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int m = (wm * MT + mt) * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
      const int col = wn * 32 + r;
      if (OUT_F32)
        *(float*)(smem + m * SROW + col * 4) = acc[mt][i];
      else
        *(bf16_t*)(smem + m * SROW + col * 2) = f2bf(fused ? act_fwd_core(fmaf(acc[mt][i], ep_sc, ep_sh), p.ep_act) : acc[mt][i]);
    }
  }   // !MF16
  __syncthreads();

  if (abl & 256) {
    float t = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int i = 0; i < 16; ++i) t += acc[mt][i];
    if (t == 12345.f) ((float*)p.y)[0] = t;
  }
  if (p.stats != nullptr && tid < BN && !(abl & 64)) {
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int w = 0; w < WM; ++w) {
      s += red[(w * 2 + 0) * BN + tid];
      ss += red[(w * 2 + 1) * BN + tid];
    }
    const int co = cout0 + tid;
    if (co < p.Cout) {  // one fp64 add per workgroup and channel (agent scope); the order cannot change the fp32 result
      double* slot = p.stats + (size_t)(tile % PLYOLO_STAT_SLOTS) * 2 * p.Cout;
      __hip_atomic_fetch_add(slot + co, (double)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(slot + p.Cout + co, (double)ss, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }

  if (abl & 128) return;
  if (OUT_F32) {
    // fp32 rows (the raw head maps, pitch 5+C floats) are only 4-byte aligned: dwordx4 stores with a 4-byte
    // aligned type (global memory takes them) instead of one dword per thread and iteration
    typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
    float* y = (float*)p.y;
    constexpr int VPR4 = BN / 4;
    for (int idx = tid; idx < BM * VPR4; idx += NT) {
      const int m = idx / VPR4, v = idx - m * VPR4;
      const int a = oy0 + (m >> 4), b = ox0 + (m & 15), co = cout0 + v * 4;
      if (a < p.OHt && b < p.OWt && co < p.Cout) {
        const int oy = a * p.so + p.oy_off, ox = b * p.so + p.ox_off;
        f32x4 val = *(const f32x4*)(smem + m * SROW + v * 16);
        float* dst = y + ((size_t)(n * p.OHf + oy) * p.OWf + ox) * p.y_ld + co;
        if (co + 4 <= p.Cout) {
          if (p.bias) val += *(const f32x4_u*)(p.bias + co);
          if (p.accumulate) val += *(const f32x4_u*)dst;
          *(f32x4_u*)dst = val;
        } else {
          for (int j = 0; co + j < p.Cout; ++j) {
            float t = val[j];
            if (p.bias) t += p.bias[co + j];
            if (p.accumulate) t += dst[j];
            dst[j] = t;
          }
        }
      }
    }
  } else {
    bf16_t* y = (bf16_t*)p.y;
    constexpr int VPR = BN / 8;
    if constexpr (RED) {
      // same rows, same stores; every thread owns ONE channel vector (256 % VPR == 0) and NIT rows.  The upstream unit's z vectors
      // (and the old dx rows of an accumulating launch) are requested up front -- one memory round trip for the whole loop
      constexpr int NIT = BM * VPR / NT;
      static_assert(NT % VPR == 0 && (BM * VPR) % NT == 0, "RED: whole rows per thread");
      const int v = tid % VPR, co = cout0 + v * 8;
      BnRedThread rt;
      bnred_init(rt, p.red, co);
      // (FLAT tiles with ten rows per thread run the loop in two halves: ten z vectors + ten old rows in flight cost the 160-pixel tile its third
      // wave per SIMD -- 189 VGPRs)
      constexpr int NPASS = (FLAT && NIT > 8) ? 2 : 1, NIP = NIT / NPASS;
      static_assert(NIT % NPASS == 0, "RED: whole passes");
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
      u32x4 zq[NIP], old[NIP];
      size_t pixs[NIP];
      bool ok[NIP];
#pragma unroll
      for (int it = 0; it < NIP; ++it) {
        const int m = (tid + (ps * NIP + it) * NT) / VPR;
        int rr_, cc_;
        tile_rc(m, rr_, cc_);
        const int a = oy0 + rr_, b = ox0 + cc_;
        ok[it] = a < p.OHt && b < p.OWt && co < p.Cout;
        const int oy = a * p.so + p.oy_off, ox = b * p.so + p.ox_off;
        pixs[it] = ok[it] ? (size_t)(n * p.OHf + oy) * p.OWf + ox : 0;
        zq[it] = rt.z ? bnred_load(rt, pixs[it]) : u32x4{0u, 0u, 0u, 0u};
        old[it] = p.accumulate ? *(const u32x4*)(y + pixs[it] * p.y_ld + (co < p.Cout ? co : 0)) : u32x4{0u, 0u, 0u, 0u};
      }
#pragma unroll
      for (int it = 0; it < NIP; ++it) {
        const int m = (tid + (ps * NIP + it) * NT) / VPR;
        if (ok[it]) {
          u32x4 val = *(const u32x4*)(smem + m * SROW + v * 16);
          if (p.accumulate) val = add_bf16x8(old[it], val);
          *(u32x4*)(y + pixs[it] * p.y_ld + co) = val;
          if (rt.z) bnred_add_any(rt, val, zq[it]);
        }
      }
      }
      __syncthreads();   // every thread is done with the staging rows: the fold reuses them
      bnred_flush<NT, VPR>(rt, p.red, cout0, (float*)smem, tid, tile % PLYOLO_STAT_SLOTS);
    } else {
    for (int idx = tid; idx < BM * VPR; idx += NT) {
      const int m = idx / VPR, v = idx - m * VPR;
      int rr_, cc_;
      tile_rc(m, rr_, cc_);
      const int a = oy0 + rr_, b = ox0 + cc_, co = cout0 + v * 8;
      if (a < p.OHt && b < p.OWt && co < p.Cout) {
        const int oy = a * p.so + p.oy_off, ox = b * p.so + p.ox_off;
        u32x4 val = *(const u32x4*)(smem + m * SROW + v * 16);
        const size_t pix = (size_t)(n * p.OHf + oy) * p.OWf + ox;
        bf16_t* dst = y + pix * p.y_ld + co;
        if (p.ep_res) val = add_bf16x8(*(const u32x4*)(p.ep_res + pix * p.ep_res_ld + co), val);
        if (p.accumulate) val = add_bf16x8(*(const u32x4*)dst, val);
        *(u32x4*)dst = val;
      }
    }
    }
  }
}

template <int BN, int CK, int TH, bool OUT_F32, int ABL = 0, bool DB = false, bool PRE = false, bool MF16 = false, bool S2 = false, bool RED = false, bool BNB = false,
          bool FLAT = false>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(const ConvP p) {
  conv_mfma_body<BN, CK, TH, OUT_F32, ABL, DB, PRE, MF16, S2, RED, BNB, FLAT>(p, (int)blockIdx.x, (int)gridDim.x);
}

// MF16 instances (stride-1 tiles of 8 rows, 32-channel double-buffered chunks, bf16 output): launch with the wider pixel pitch
template <int BN>
hipError_t launch_inst_mf16(ConvP p, hipStream_t s) {
  constexpr int CK = 32, TH = 8, BM = TH * TW, WN = BN / 32, WM = 4 / WN;
  constexpr int ROWB = CK * 2 + 32, SROW = BN * 2 + 16;
  p.rowp = (p.ITW * ROWB + 255) & ~255;
  p.bufsz = p.ITH * p.rowp;
  const size_t lds_main = 2 * (size_t)p.bufsz, lds_epi = (size_t)BM * SROW + WM * 2 * BN * 4;
  const size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
  auto kern = conv_mfma_kernel<BN, CK, TH, false, 0, true, false, true>;
  if (hipError_t e = plyolo::ensure_dynamic_lds((const void*)kern, lds); e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(p.nmb, (p.Cout + BN - 1) / BN), dim3(256), lds, s, p);
  return hipGetLastError();
}

template <int BN, int CK, int TH, bool DB = false, bool RED = false>
__global__ __launch_bounds__(256, 2) void conv_mfma_jobs_kernel(const ConvJobs jobs) {
  int j = 0;
  for (int k = 1; k < 4; ++k)
    if (k < jobs.n && (int)blockIdx.x >= jobs.start[k]) j = k;
  conv_mfma_body<BN, CK, TH, false, 0, DB, false, false, false, RED>(jobs.c[j], (int)blockIdx.x - jobs.start[j], jobs.start[j + 1] - jobs.start[j]);
}

template <int BN, int CK, int TH, bool OUT_F32, bool PRE = false>
hipError_t launch_inst(ConvP p, hipStream_t s) {
  constexpr int BM = TH * TW, WN = BN / 32, WM = 4 / WN;
  constexpr int ROWB = CK * 2 + 16;
  constexpr int SROW = OUT_F32 ? (BN + 4) * 4 : (BN * 2 + 16);
  p.rowp = p.ITW * ROWB;
  if (p.si == 1) p.rowp = (p.rowp + 255) & ~255;
  size_t lds_main = (size_t)p.ITH * p.rowp;
  size_t lds_epi = (size_t)BM * SROW + WM * 2 * BN * 4;
  size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
  auto kern = conv_mfma_kernel<BN, CK, TH, OUT_F32, 0, false, PRE>;
  // double-buffered halo tile: stride-1 tiles with more than one Cin chunk
  constexpr bool HAS_DB = !OUT_F32 && ((TH == 8 && (CK == 32 || CK == 64)) || (TH == 16 && CK == 32));
  if constexpr (!OUT_F32 && !PRE && TH == 8 && CK == 32 && (BN == 128 || BN == 64)) {
    // default since round 3 (same box, three alternations: 9.31 -> 9.245 ms/step; forward launches of this shape -3.5 %, data
    // gradients -7.7 %; profiles/r03_ab_mfma16.txt).  PLYOLO_MFMA16=0 restores the 32x32x16 instances in an OPTIN build (the
    // shipped library does not carry both: co-compiled instances cost the hot kernels time, see the top of this file)
#ifdef PLYOLO_OPTIN
    static const bool mf16 = !(getenv("PLYOLO_MFMA16") && atoi(getenv("PLYOLO_MFMA16")) == 0);
#else
    constexpr bool mf16 = true;
#endif
    if (mf16 && p.db && p.si == 1 && p.Cin > CK && !p.ablate) return launch_inst_mf16<BN>(p, s);
  }
#ifndef PLYOLO_OPTIN
  // the 32x32x16 double-buffered instances of the two MF16 shapes are never launched by the shipped library: do not instantiate them
  constexpr bool MF16_SHAPE = !OUT_F32 && !PRE && TH == 8 && CK == 32 && (BN == 128 || BN == 64);
#else
  constexpr bool MF16_SHAPE = false;
#endif
  if constexpr (HAS_DB && !MF16_SHAPE) {
    if (p.db && p.si == 1 && p.Cin > CK && !p.ablate) {
      p.bufsz = p.ITH * p.rowp;
      lds_main = 2 * (size_t)p.bufsz;
      lds = lds_main > lds_epi ? lds_main : lds_epi;
      kern = conv_mfma_kernel<BN, CK, TH, OUT_F32, 0, true, PRE>;
    }
  }
#ifdef PLYOLO_DIAG_ABLATE   // diagnostic instantiations (results are wrong by design): make DIAG=1
  if (!PRE && BN == 128 && CK == 64 && TH == 16 && !OUT_F32 && p.ablate) {  // diagnostic instantiations of the main shape only
    switch (p.ablate) {
      case 1: kern = conv_mfma_kernel<128, 64, 16, false, 1>; break;
      case 8: kern = conv_mfma_kernel<128, 64, 16, false, 8>; break;
      case 9: kern = conv_mfma_kernel<128, 64, 16, false, 9>; break;
      case 41: kern = conv_mfma_kernel<128, 64, 16, false, 41>; break;
      case 57: kern = conv_mfma_kernel<128, 64, 16, false, 57>; break;
      case 61: kern = conv_mfma_kernel<128, 64, 16, false, 61>; break;
      case 33: kern = conv_mfma_kernel<128, 64, 16, false, 33>; break;
      case 5: kern = conv_mfma_kernel<128, 64, 16, false, 5>; break;
      case 16: kern = conv_mfma_kernel<128, 64, 16, false, 16>; break;
      case 37: kern = conv_mfma_kernel<128, 64, 16, false, 37>; break;
      case 64: kern = conv_mfma_kernel<128, 64, 16, false, 64>; break;
      case 128: kern = conv_mfma_kernel<128, 64, 16, false, 128>; break;
      case 192: kern = conv_mfma_kernel<128, 64, 16, false, 192>; break;
      case 256: kern = conv_mfma_kernel<128, 64, 16, false, 256>; break;
      case 448: kern = conv_mfma_kernel<128, 64, 16, false, 448>; break;
      default: break;
    }
  }
#endif
  if (hipError_t e = plyolo::ensure_dynamic_lds((const void*)kern, lds); e != hipSuccess) return e;
  dim3 grid(p.nmb, (p.Cout + BN - 1) / BN);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, p);
  return hipGetLastError();
}


}  // namespace
