// 3x3 stride-1 weight gradient, 64 x 64 slab tile, TWELVE waves per workgroup: one kernel ROW per wave triple (gfx950).
//
// conv_wgrad3_kernel (conv_wgrad_mfma.hip) gives each of its four waves a 32 x 32 block of the slab for ALL nine taps: 144 accumulator
// registers, ONE wave per SIMD -- every LDS latency, every barrier and every staging instruction of the tile pipeline sits in the one
// instruction stream that also has to keep the matrix pipe fed (rocprofv3: MFMA busy ~39 % on the CUs a launch occupies).  Here the same
// two LDS tiles are shared by three times as many waves: wave (wr, wco, wci) owns the 32 x 32 block (wco, wci) for the three taps of kernel
// row wr -- 48 accumulator registers, three waves per SIMD, so one wave's LDS read or barrier wait is another wave's MFMA time.  Unlike the
// tap-row split ACROSS workgroups (PLYOLO_WG_TRS=3: three workgroups each re-read the dY tile from L2, measured slower) nothing is read
// twice: the twelve waves stage the tiles together.  The X fragments of a row's three taps come from one 12-pixel window (three transposed
// reads, conv_wgrad3_kernel's PLYOLO_WG_WIN path).  Same tile pipeline (two tiles in LDS, tile i+2 requested while tile i multiplies, raw
// buffer loads through per-image descriptors, one barrier per tile), same private slabs, same summation order per slab element as the
// four-wave kernel (pixels of a tile in k-step order, tiles in stride order): the slabs are bit-identical to conv_wgrad3_kernel's.
#include <stdlib.h>

#include "conv_wgrad_common.h"

namespace {

template <int SI>
__global__ __launch_bounds__(768) void conv_wgrad3r_kernel(const WgP p) {
  static_assert(SI == 1, "row-split instance: stride 1");
  constexpr int CO_T = 64, CI_T = 64, TH_ = 8, KS = 3, NTHR = 768;
  constexpr int DZB = pitch_for(CO_T), XB = pitch_for(CI_T);
  constexpr int DZV = CO_T / 8, XV = CI_T / 8;
  constexpr int ITH_ = (TH_ - 1) * SI + KS, ITW_ = (TW - 1) * SI + KS;
  constexpr int NDV = TH_ * TW * DZV, NXV = ITH_ * ITW_ * XV;
  constexpr int DV = (NDV + NTHR - 1) / NTHR, HV = (NXV + NTHR - 1) / NTHR, NV = DV + HV;
  constexpr int NJ = TH_;
  constexpr int DZ_BYTES = TH_ * TW * DZB, X_BYTES = ITH_ * ITW_ * XB;
  constexpr int BUF = DZ_BYTES + X_BYTES, DUMP = BUF, BUFP = BUF + NTHR * 16;
  static_assert(2 * BUFP <= 160 * 1024, "two tiles must fit in LDS");
  extern __shared__ __align__(16) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wco = (wave >> 1) & 1, wci = wave & 1;       // kernel row, 32-channel block of co / ci
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
  const int b_off = DZ_BYTES + (wr * ITW_ + kpix) * XB + (wci * 32 + 16 * (g & 1) + 4 * pp) * 2;    // halo row j + wr, the window's first pixel

  f32x16 acc[KS];
#pragma unroll
  for (int t = 0; t < KS; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  // ---- tile-invariant state of the vectors this thread stages (vector v < DV: dY, else X halo)
  constexpr unsigned NEVER = 0x7fffu;
  int rel[NV], lds[NV];
  unsigned yx[NV];
#pragma unroll
  for (int v = 0; v < DV; ++v) {
    const int idx = tid + v * NTHR, m = idx / DZV, vv = idx - m * DZV, co = co0 + vv * 8;
    const bool live = idx < NDV && co < p.Cout;
    rel[v] = (((m >> 4) * p.OW + (m & 15)) * p.dy_ld + co) * 2;
    yx[v] = live ? (unsigned)(m >> 4) | ((unsigned)(m & 15) << 16) : NEVER;
    lds[v] = idx < NDV ? m * DZB + vv * 16 : DUMP + tid * 16;
  }
#pragma unroll
  for (int v = 0; v < HV; ++v) {
    const int idx = tid + v * NTHR, pix = idx / XV, vv = idx - pix * XV, iy = pix / ITW_, ix = pix - iy * ITW_, ci = ci0 + vv * 8;
    const bool live = idx < NXV && ci < p.Cin;
    rel[DV + v] = ((iy * p.W + ix) * p.x_ld + ci) * 2;
    yx[DV + v] = live ? (unsigned)iy | ((unsigned)ix << 16) : NEVER;
    lds[DV + v] = idx < NXV ? DZ_BYTES + pix * XB + vv * 16 : DUMP + tid * 16;
  }

  struct Tile {
    __amdgpu_buffer_rsrc_t rd, rx;
    int oy0, ox0, iy0, ix0, dbase, xbase;
  };
  const int d_img = ((p.OH * p.OW - 1) * p.dy_ld + ((p.Cout + 7) & ~7)) * 2, x_img = ((p.H * p.W - 1) * p.x_ld + ((p.Cin + 7) & ~7)) * 2;
  auto tile_at = [&](int tile) {      // wave-uniform: lives in SGPRs
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
    return t;
  };
  u32x4 R[NV];
  auto request = [&](const Tile& t, int v) {
    const bool isx = v >= DV;
    const int y = (isx ? t.iy0 : t.oy0) + (int)(yx[v] & 0xffffu), x = (isx ? t.ix0 : t.ox0) + (int)(yx[v] >> 16);
    const bool ok = (unsigned)y < (unsigned)(isx ? p.H : p.OH) && (unsigned)x < (unsigned)(isx ? p.W : p.OW);
    const int off = ok ? (isx ? t.xbase : t.dbase) + rel[v] : (int)0x80000000;   // past any image: the range check returns zeros
    R[v] = __builtin_amdgcn_raw_buffer_load_b128(isx ? t.rx : t.rd, off, 0, 0);
  };
  auto stage = [&](unsigned char* buf, int v) { *(u32x4*)(buf + lds[v]) = R[v]; };   // lanes beyond the tile write to a dump row

  // ---- prologue: tile 0 into buffer 0, tile 1 requested (requests in the order the main loop re-issues them)
  {
    const Tile t0 = tile_at(split);
#pragma unroll
    for (int v = 0; v < NV; ++v) request(t0, v);
#pragma unroll
    for (int v = 0; v < NV; ++v) stage(smem, v);
    const Tile t1 = tile_at(split + p.S);
#pragma unroll
    for (int v = 0; v < NV; ++v) request(t1, v);
  }
  __syncthreads();

  int cur = 0;
  for (int tile = split; tile < p.ntiles; tile += p.S) {
    const unsigned char* bc = smem + cur * BUFP;
    unsigned char* bn = smem + (cur ^ 1) * BUFP;
    const Tile t2 = tile_at(tile + 2 * p.S);
    const unsigned char* ap = bc + a_off;
    const unsigned char* bp = bc + b_off;
    s16x8 af0, af1, bf0[KS], bf1[KS];
    auto ldfrag = [&](int j, s16x8& af, s16x8 (&bfv)[KS]) {
      const s16x4 lo = tr_read(ap + j * TW * DZB);
      const s16x4 hi = tr_read(ap + j * TW * DZB + 4 * DZB);
      af = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      // the three taps of this wave's kernel row: one 12-pixel window of halo row j + wr (dx = 0 / 2 are register renames, dx = 1 four v_alignbit)
      const unsigned char* rp = bp + j * ITW_ * XB;
      const s16x4 w0 = tr_read(rp), w1 = tr_read(rp + 4 * XB), w2 = tr_read(rp + 8 * XB);
      typedef __attribute__((ext_vector_type(2))) unsigned u32x2_;
      const u32x2_ a0 = *(const u32x2_*)&w0, a1 = *(const u32x2_*)&w1, a2 = *(const u32x2_*)&w2;
      const unsigned D0 = a0[0], D1 = a0[1], D2 = a1[0], D3 = a1[1], D4 = a2[0];
      const u32x4 t0 = {D0, D1, D2, D3}, t2_ = {D1, D2, D3, D4};
      const u32x4 t1 = {__builtin_amdgcn_alignbit(D1, D0, 16), __builtin_amdgcn_alignbit(D2, D1, 16), __builtin_amdgcn_alignbit(D3, D2, 16),
                        __builtin_amdgcn_alignbit(D4, D3, 16)};
      bfv[0] = *(const s16x8*)&t0;
      bfv[1] = *(const s16x8*)&t1;
      bfv[2] = *(const s16x8*)&t2_;
    };
    auto mm = [&](const s16x8& af, const s16x8 (&bfv)[KS]) {
#pragma unroll
      for (int t = 0; t < KS; ++t)
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
    ldfrag(0, af0, bf0);
#pragma unroll
    for (int jj = 0; jj < NJ; jj += 2) {
      ldfrag(jj + 1, af1, bf1);
      mm(af0, bf0);
      side(jj);
      __builtin_amdgcn_sched_barrier(0);
      if (jj + 2 < NJ) ldfrag(jj + 2, af0, bf0);
      mm(af1, bf1);
      side(jj + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();   // tile i consumed by every wave, tile i+1 complete in the other buffer
    cur ^= 1;
  }

  float* slab = p.dw + (size_t)split * ((size_t)KS * KS * p.Cout * p.Cin);
  const int r = lane & 31, h = lane >> 5;
  const int ci = ci0 + wci * 32 + r;
  if (ci < p.Cin) {
#pragma unroll
    for (int t = 0; t < KS; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int co = co0 + wco * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (co < p.Cout) slab[((size_t)(wr * KS + t) * p.Cout + co) * p.Cin + ci] = acc[t][i];
      }
  }
}

}  // namespace

namespace plyolo {

// `wgp`: the WgP of a 3x3 stride-1 weight gradient planned for the 64 x 64 slab tile with 8-row tiles (conv_wgrad_mfma.hip: id 0); S spatial splits
hipError_t conv_wgrad3r_launch(const void* wgp, int S, hipStream_t s) {
  const WgP& p = *(const WgP*)wgp;
  constexpr int DZB = pitch_for(64), XB = pitch_for(64);
  const size_t lds = 2 * ((size_t)8 * TW * DZB + (size_t)10 * 18 * XB + 768 * 16);
  auto kern = conv_wgrad3r_kernel<1>;
  if (hipError_t e = ensure_dynamic_lds((const void*)kern, 160 * 1024); e != hipSuccess) return e;
  const int nco = (p.Cout + 63) / 64;
  WgP q = p;
  q.nslabt = nco * p.nci;
  q.S = S;
  static const int xcd = getenv("PLYOLO_WG_XCD") ? atoi(getenv("PLYOLO_WG_XCD")) : 1;
  q.xcd = xcd;
  hipLaunchKernelGGL(kern, dim3(S * q.nslabt), dim3(768), lds, s, q);
  return hipGetLastError();
}

}  // namespace plyolo
