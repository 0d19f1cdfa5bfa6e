// Whole backward of a pointwise (1x1, stride 1) BaseConv unit behind its BatchNorm reduction, in ONE persistent launch (gfx950).
//
// Replaces, for act(bn(conv1x1(x))) (reference models/layers/network_blocks.py:18-40), what autograd does between the gradient
// of the activated output and (dx, dW, dgamma, dbeta) once the two per-channel sums of the BatchNorm backward are known:
//
//     dz = A*du + B*z + Cc,  du = dout * act'(z*sc + sh)        (bn.hip: bn_act_bwd_dz, bit for bit)
//     dx[M, Cin]  (+)= dz[M, Cout] . W[Cout, Cin]                (conv_pw.hip: conv_pw_dgrad)
//     dW[Cout, Cin] = dz^T[Cout, M] . x[M, Cin]                  (conv_wgrad_mfma.hip, 1x1 variants)
//
// As separate launches that chain moves 3 E_out (dz pass) + E_out + E_in (data gradient) + E_out + E_in (weight gradient) bytes
// and writes dz to HBM only to read it twice.  Here every pixel tile of dout, z and x is read ONCE and dz lives in LDS only:
// 2 E_out + 2 E_in bytes, both GEMMs run out of the same LDS image (the kernel is HBM-bound: two GEMMs of 2*M*Cout*Cin FLOPs
// each leave the matrix cores ~80 % idle at 64 ... 128 channels).
//
//   * persistent workgroups (256 threads, <= 2 per CU) walk the pixel tiles round-robin; the weight gradient accumulates in
//     registers across ALL tiles of a workgroup (Cout x Cin fp32 = 64 registers per lane at 128 x 128) and leaves as one
//     private fp32 slab per workgroup (no atomics; plyolo_reduce_slabs folds them in a fixed order -- run-to-run identical);
//   * the loads of tile t+1 (16-byte vectors of whole rows -> registers) are issued behind tile t's LDS image and stay in
//     flight during its MFMAs and epilogue;
//   * ONE LDS image of dz serves both products: rows (ds_read_b128) as the A operand of dx = dz.W, columns
//     (ds_read_b64_tr_b16) as the A operand of dW = dz^T.x.  Row pitch C*2+16 bytes makes the row reads conflict-free; the
//     transposed reads are conflict-free with the contraction index permuted inside every 16-pixel k-step (k -> row
//     4*(k&3) + (k>>2): the four rows of a half-wave's block are 4 apart, 4*pitch = 64 mod 256 bytes) -- a sum over pixels
//     does not care about its order, and dz and x are read with the same permutation;
//   * W (data-gradient fragment pack, conv_mfma.hip order) sits in registers for the whole launch; the per-channel
//     coefficient table (sc, sh, A, B, Cc) is built once per workgroup from the fp64 stat slots.
#include <stdlib.h>

#include "common.h"
#include "bnred.h"

namespace {

struct PbP {
  const bf16_t* dout;    // gradient of the activated output [M][CO] (channels >= dsplit in dout2 for merged pairs)
  const bf16_t* dout2;
  const bf16_t* z;       // raw conv output of the forward [M][CO]
  const bf16_t* x;       // input of the forward [M][CI]
  const bf16_t* wpd;     // data-gradient fragment pack [CI/32][CO/16][64 lanes][8]
  bf16_t* dx;            // [M][CI], pitch dx_ld
  float* dw;             // slabs [G * WK][CO][CI]
  const float* coef;     // (scale | shift | mean | invstd) [4][CO]
  const double* bslots;  // [PLYOLO_STAT_SLOTS][2][CO]
  const float *gamma, *gamma2;
  float *dgamma, *dbeta, *dgamma2, *dbeta2;
  int dout_ld, dout2_ld, dsplit, psplit, z_ld, x_ld, dx_ld;
  int M, ntiles, act, accumulate;
  plyolo_bn_red red;     // RED instances: BatchNorm-backward reduction of the unit(s) that produced x (their output gradient is dx)
};

typedef __attribute__((address_space(3))) s16x4 lds_s16x4_b;
DEVINL s16x4 tr_read_b(const unsigned char* p) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_b*)p); }

DEVINL float pb_act_grad(float u, int act) {     // == pw_act_grad (conv_pw.hip) == act_grad<false> for the three cheap activations
  switch (act) {
    case PLYOLO_ACT_SILU: {
      const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-u));
      return s * (1.0f + u * (1.0f - s));
    }
    case PLYOLO_ACT_RELU: return u > 0.f ? 1.f : 0.f;
    case PLYOLO_ACT_LRELU: return u > 0.f ? 1.f : 0.1f;
    default: return 1.f;
  }
}

DEVINL u32x4 pb_add_bf16x8(u32x4 a, u32x4 b) {
  u32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float lo = __uint_as_float(a[i] << 16) + __uint_as_float(b[i] << 16);
    float hi = __uint_as_float(a[i] & 0xffff0000u) + __uint_as_float(b[i] & 0xffff0000u);
    r[i] = pack2bf(lo, hi);
  }
  return r;
}

constexpr int pb_min(int a, int b) { return a < b ? a : b; }

constexpr int NW = 8, NT = NW * 64;     // eight waves per workgroup, ONE workgroup per CU (two waves per SIMD)

template <int CO, int CI, int BM> struct PbGeom {
  // data gradient: dx tile [BM][CI], wave (wm, wn) owns 32 input channels x MT 32-row fragments
  static constexpr int WN = pb_min(CI / 32, 4), WM = NW / WN, MT = BM / (32 * WM);
  static constexpr int KSD = CO / 16;                  // k-steps of the data gradient (contraction over CO)
  // weight gradient: dW [CO][CI] in 32x32 tiles over the eight waves; fewer than eight tiles -> wave groups split the k-steps
  static constexpr int TCO = CO / 32, TCI = CI / 32;
  static constexpr int WCI = pb_min(TCI, 4), WCO = pb_min(TCO, NW / WCI), WK = NW / (WCO * WCI);
  static constexpr int MTC = TCO / WCO, MTI = TCI / WCI;
  static constexpr int KSW = BM / 16;                  // k-steps of the weight gradient per tile (contraction over pixels)
  static constexpr int PD = CO * 2 + 16, PX = CI * 2 + 16;   // LDS row pitches (bytes)
  static constexpr int DV = CO / 8, XV = CI / 8;       // 16-byte vectors per row
  static constexpr int NDV = BM * DV / NT, NXV = BM * XV / NT;   // vectors per thread and tile
  static constexpr int DZ_BYTES = BM * PD, X_BYTES = BM * PX, STG_BYTES = BM * PX;
  static constexpr int TAB_OFF = DZ_BYTES + X_BYTES + STG_BYTES;
  static constexpr int RTAB_OFF = TAB_OFF + 5 * CO * 4;          // RED: (scale | shift | mean | invstd) of the upstream unit(s), [4][CI]
  static constexpr int LDS = RTAB_OFF + 4 * CI * 4;
  static_assert(MT >= 1 && BM % (32 * WM) == 0, "tile rows vs wave layout");
  static_assert(WCO * WCI * WK == NW && KSW % WK == 0 && MTC >= 1 && MTI >= 1, "eight waves");
  static_assert(NT % DV == 0 && NT % XV == 0 && (BM * DV) % NT == 0 && (BM * XV) % NT == 0, "whole vectors per thread");
  static_assert((WK - 1) * WCO * WCI * MTC * MTI * 4096 <= DZ_BYTES + X_BYTES + STG_BYTES, "k-split partials fit in LDS");
};

// ACT: PLYOLO_ACT_SILU = the compile-time SiLU instance every shipped config runs (straight-line staging code: a switch on a run-time
// activation inside the unrolled element loops compiles to a branch per element); -1 = the activation is p.act (none / relu / lrelu)
// RED: the dx rows this launch completes are also the output gradient of the unit(s) that produced x: their BatchNorm-backward
// reduction (plyolo_bn_red, bnred.h) rides the dx store -- partial sums in registers across ALL tiles of a workgroup, one fold and one
// set of fp64 slot adds per workgroup at the end
template <int CO, int CI, int BM, int ACT, bool RED = false>
__global__ __launch_bounds__(NT, 1) void conv_pw_bwd_kernel(const PbP p) {
  using G = PbGeom<CO, CI, BM>;
  constexpr int WN = G::WN, WM = G::WM, MT = G::MT, KSD = G::KSD;
  constexpr int WCO = G::WCO, WCI = G::WCI, WK = G::WK, MTC = G::MTC, MTI = G::MTI, KSW = G::KSW;
  constexpr int PD = G::PD, PX = G::PX, DV = G::DV, XV = G::XV, NDV = G::NDV, NXV = G::NXV;
  extern __shared__ __align__(16) unsigned char smem[];
  unsigned char* dz_s = smem;
  unsigned char* x_s = smem + G::DZ_BYTES;
  unsigned char* stg = smem + G::DZ_BYTES + G::X_BYTES;
  float* tab = (float*)(smem + G::TAB_OFF);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int G_ = (int)gridDim.x, wg = (int)blockIdx.x;

  // ---- per-thread staging geometry (tile-invariant): vector v of the dz / x image = row drow + v*DRP, channel vector dcv
  constexpr int DRP = NT / DV, XRP = NT / XV;
  const int dcv = tid % DV, drow = tid / DV;
  const int xcv = tid % XV, xrow = tid / XV;
  const int dch = dcv * 8;
  const bool second = p.dsplit > 0 && dch >= p.dsplit;
  const bf16_t* dsrc = second ? p.dout2 + (dch - p.dsplit) : p.dout + dch;
  const int dld = second ? p.dout2_ld : p.dout_ld;

  // two register sets: while tile t is turned into its LDS image and multiplied, the rows of tiles t+G and t+2G are in flight
  struct Rows { u32x4 av[NDV], zv[NDV], xv[NXV]; };
  // Requests are UNCONDITIONAL (no branch around them): the compiler's vector-memory wait counts are exact only in straight-line
  // code -- a load behind a branch makes every later wait for an OLDER load a wait for everything in flight, prefetch included.
  // A tile index beyond the last tile (and a row beyond M) reads row 0 instead: one L2-resident row, never used.
  auto request = [&](Rows& R, const int tile) {
    const int m0 = tile * BM;
#pragma unroll
    for (int v = 0; v < NDV; ++v) {
      const int m = m0 + drow + v * DRP, mc = (tile < p.ntiles && m < p.M) ? m : 0;
      R.av[v] = *(const u32x4*)(dsrc + (size_t)mc * dld);
      R.zv[v] = *(const u32x4*)(p.z + (size_t)mc * p.z_ld + dch);
    }
#pragma unroll
    for (int v = 0; v < NXV; ++v) {
      const int m = m0 + xrow + v * XRP, mc = (tile < p.ntiles && m < p.M) ? m : 0;
      R.xv[v] = *(const u32x4*)(p.x + (size_t)mc * p.x_ld + xcv * 8);
    }
  };
  Rows RA, RB;

  // ---- data-gradient weights of this wave: 32 input channels x all CO, resident in registers.  Requested first and pinned as
  // "arrived" behind the table build: the vector-memory counter retires in order, so a wait for them inside the tile loop (where
  // their first use is) would have to be written as "everything in flight has landed", prefetched rows included
  const int wm = wave / WN, wn = wave % WN;
  u32x4 wq[KSD];
  {
    const char* wbase = (const char*)p.wpd + (size_t)(wn * KSD) * 1024u + (size_t)lane * 16u;
#pragma unroll
    for (int kk = 0; kk < KSD; ++kk) wq[kk] = *(const u32x4*)(wbase + (size_t)kk * 1024u);
  }

  // ---- per-channel table (scale, shift, A, B, Cc) (conv_pw.hip BNB, bit for bit)
  for (int ch = tid; ch < CO; ch += NT) {
    double su = 0.0, suz = 0.0;
#pragma unroll
    for (int sl = 0; sl < PLYOLO_STAT_SLOTS; ++sl) {
      su += p.bslots[((size_t)sl * 2 + 0) * CO + ch];
      suz += p.bslots[((size_t)sl * 2 + 1) * CO + ch];
    }
    const float mean = p.coef[2 * CO + ch], invstd = p.coef[3 * CO + ch];
    const bool sec = p.psplit > 0 && ch >= p.psplit;
    const int cp = sec ? ch - p.psplit : ch;
    const float* gam = sec ? p.gamma2 : p.gamma;
    const double cnt = (double)p.M;
    const float A = (gam ? gam[cp] : 1.f) * invstd;
    const float B = (float)(-(double)A * (suz / cnt) * (double)invstd);
    tab[ch] = p.coef[ch];
    tab[CO + ch] = p.coef[CO + ch];
    tab[2 * CO + ch] = A;
    tab[3 * CO + ch] = B;
    tab[4 * CO + ch] = (float)(-(double)A * (su / cnt) - (double)B * (double)mean);
    if (wg == 0) {   // dbeta = sum du, dgamma = sum du*zhat (bn_act_bwd_dz publishes them from its workgroup 0)
      float* db_ = sec ? p.dbeta2 : p.dbeta;
      float* dg_ = sec ? p.dgamma2 : p.dgamma;
      if (db_) db_[cp] = (float)su;
      if (dg_) dg_[cp] = (float)suz;
    }
  }

#pragma unroll
  for (int kk = 0; kk < KSD; ++kk) asm volatile("" : "+v"(wq[kk]));
  request(RA, wg);
  __builtin_amdgcn_sched_barrier(0);   // set A's loads stay in front of set B's: the loop head waits for "all but set B" by count
  request(RB, wg + G_);
  __builtin_amdgcn_sched_barrier(0);

  // ---- weight-gradient accumulators of this wave
  const int wk = wave / (WCO * WCI), wco = (wave / WCI) % WCO, wci = wave % WCI;
  f32x16 accw[MTC][MTI];
#pragma unroll
  for (int a = 0; a < MTC; ++a)
#pragma unroll
    for (int b = 0; b < MTI; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) accw[a][b][i] = 0.f;

  // transposed-read addresses: 16-lane group g = lane>>4 reads the block of rows k = 8*(g>>1) + {0..3} (+4 for the second
  // read), columns 16*(g&1) .. +15 of a 32-channel fragment; lane 4q+pp of the group supplies row q, columns 4pp .. 4pp+3.
  // Physical row of k inside the 16-pixel k-step: 4*(k&3) + (k>>2)  ->  4q + 2*(g>>1) (+1 for the second read)
  const int g4 = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3;
  const int trow = 4 * q + 2 * (g4 >> 1);
  const int a_off = trow * PD + (wco * MTC * 32 + 16 * (g4 & 1) + 4 * pp) * 2;
  const int b_off = trow * PX + (wci * MTI * 32 + 16 * (g4 & 1) + 4 * pp) * 2;
  int arow[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) arow[mt] = ((wm * MT + mt) * 32 + r) * PD + h * 16;

  // RED: this thread's 8 dx channels (xcv) belong to one segment; its coefficients go to an LDS table (read back per tile: the
  // 128 x 128 instance has no registers left to keep 32 of them resident), the 2 x 8 partial sums stay in registers
  [[maybe_unused]] BnRedThread rt;
  [[maybe_unused]] float* rtab = (float*)(smem + G::RTAB_OFF);
  if constexpr (RED) {
    bnred_init(rt, p.red, xcv * 8);
    if (xrow == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        rtab[0 * CI + xcv * 8 + i] = rt.sc[i]; rtab[1 * CI + xcv * 8 + i] = rt.sh[i];
        rtab[2 * CI + xcv * 8 + i] = rt.mu[i]; rtab[3 * CI + xcv * 8 + i] = rt.is[i];
      }
    }
  }

  __syncthreads();   // table complete

  // one tile: rows in R -> LDS image -> (R re-requested for tile + 2G) -> both products -> dx rows out
  auto process = [&](Rows& R, const int tile) {
    const int m0 = tile * BM;
    {
      float sc[8], sh[8], A[8], B[8], Cc[8];
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {
        const f32x4 a0 = *(const f32x4*)(tab + dch + 4 * qq), a1 = *(const f32x4*)(tab + CO + dch + 4 * qq);
        const f32x4 a2 = *(const f32x4*)(tab + 2 * CO + dch + 4 * qq), a3 = *(const f32x4*)(tab + 3 * CO + dch + 4 * qq);
        const f32x4 a4 = *(const f32x4*)(tab + 4 * CO + dch + 4 * qq);
#pragma unroll
        for (int i = 0; i < 4; ++i) { sc[4 * qq + i] = a0[i]; sh[4 * qq + i] = a1[i]; A[4 * qq + i] = a2[i]; B[4 * qq + i] = a3[i]; Cc[4 * qq + i] = a4[i]; }
      }
      const u32x4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int v = 0; v < NXV; ++v) {
        const int row = xrow + v * XRP;
        *(u32x4*)(x_s + row * PX + xcv * 16) = (m0 + row < p.M) ? R.xv[v] : zero;
      }
#pragma unroll
      for (int v = 0; v < NDV; ++v) {
        u32x4 t = R.av[v];
        const u32x4 zz = R.zv[v];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float zl = __uint_as_float(zz[i] << 16), zh = __uint_as_float(zz[i] & 0xffff0000u);
          const float dl = __uint_as_float(t[i] << 16), dh = __uint_as_float(t[i] & 0xffff0000u);
          const float dul = dl * pb_act_grad(fmaf(zl, sc[2 * i], sh[2 * i]), ACT >= 0 ? ACT : p.act);
          const float duh = dh * pb_act_grad(fmaf(zh, sc[2 * i + 1], sh[2 * i + 1]), ACT >= 0 ? ACT : p.act);
          t[i] = pack2bf(fmaf(A[2 * i], dul, fmaf(B[2 * i], zl, Cc[2 * i])), fmaf(A[2 * i + 1], duh, fmaf(B[2 * i + 1], zh, Cc[2 * i + 1])));
        }
        const int row = drow + v * DRP;
        *(u32x4*)(dz_s + row * PD + dcv * 16) = (m0 + row < p.M) ? t : zero;
      }
    }
    __syncthreads();                                            // B1: the tile's LDS image is complete
    // accumulating launches: the old dx rows are requested FIRST, so that the epilogue's wait for them does not cover the prefetch
    // (unconditional like every request: a plain launch reads row 0 four times per tile and ignores it)
    u32x4 oldx[NXV];
#pragma unroll
    for (int v = 0; v < NXV; ++v) {
      const int m = m0 + xrow + v * XRP, mc = (p.accumulate && m < p.M) ? m : 0;
      oldx[v] = *(const u32x4*)(p.dx + (size_t)mc * p.dx_ld + xcv * 8);
    }
    [[maybe_unused]] u32x4 zup[NXV];
    if constexpr (RED) {
#pragma unroll
      for (int v = 0; v < NXV; ++v) {
        const int m = m0 + xrow + v * XRP, mc = m < p.M ? m : 0;
        zup[v] = rt.z ? bnred_load(rt, (size_t)mc) : u32x4{0u, 0u, 0u, 0u};
      }
    }
    request(R, tile + 2 * G_);                                  // this register set is free again: the tile after next
    __builtin_amdgcn_sched_barrier(0);                           // ... requested ABOVE the MFMAs

    // ---- dx tile = dz . W
    f32x16 accd[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int i = 0; i < 16; ++i) accd[mt][i] = 0.f;
#pragma unroll
    for (int kk = 0; kk < KSD; ++kk) {
      const bf16x8 b = *(const bf16x8*)&wq[kk];
      bf16x8 a[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) a[mt] = *(const bf16x8*)(dz_s + arow[mt] + kk * 32);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) accd[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt], b, accd[mt], 0, 0, 0);
    }
    // ---- dW += dz^T . x  (k = the tile's pixels, 16 per step; this wave takes steps wk, wk + WK, ...)
#pragma unroll
    for (int js = 0; js < KSW / WK; ++js) {
      const int j = wk + js * WK;
      s16x8 af[MTC], bfr[MTI];
#pragma unroll
      for (int a = 0; a < MTC; ++a) {
        const unsigned char* ap = dz_s + j * 16 * PD + a_off + a * 64;
        const s16x4 lo = tr_read_b(ap), hi = tr_read_b(ap + PD);
        af[a] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int b = 0; b < MTI; ++b) {
        const unsigned char* bp = x_s + j * 16 * PX + b_off + b * 64;
        const s16x4 lo = tr_read_b(bp), hi = tr_read_b(bp + PX);
        bfr[b] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int a = 0; a < MTC; ++a)
#pragma unroll
        for (int b = 0; b < MTI; ++b)
          accw[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)&af[a], *(const bf16x8*)&bfr[b], accw[a][b], 0, 0, 0);
    }

    // ---- dx tile -> staging (bf16, pixel-major) -> whole 16-byte channel vectors to HBM
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int m = (wm * MT + mt) * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        *(bf16_t*)(stg + m * PX + (wn * 32 + r) * 2) = f2bf(accd[mt][i]);
      }
    __syncthreads();                                   // B2: staging complete; every wave is done with dz_s / x_s
#pragma unroll
    for (int v = 0; v < NXV; ++v) {
      const int row = xrow + v * XRP;
      if (m0 + row < p.M) {
        u32x4 val = *(const u32x4*)(stg + row * PX + xcv * 16);
        bf16_t* dst = p.dx + (size_t)(m0 + row) * p.dx_ld + xcv * 8;
        if (p.accumulate) val = pb_add_bf16x8(oldx[v], val);
        *(u32x4*)dst = val;
        if constexpr (RED) {
          if (rt.z) {
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
              const f32x4 c0 = *(const f32x4*)(rtab + 0 * CI + xcv * 8 + 4 * qq), c1 = *(const f32x4*)(rtab + 1 * CI + xcv * 8 + 4 * qq);
              const f32x4 c2 = *(const f32x4*)(rtab + 2 * CI + xcv * 8 + 4 * qq), c3 = *(const f32x4*)(rtab + 3 * CI + xcv * 8 + 4 * qq);
#pragma unroll
              for (int i = 0; i < 4; ++i) { rt.sc[4 * qq + i] = c0[i]; rt.sh[4 * qq + i] = c1[i]; rt.mu[4 * qq + i] = c2[i]; rt.is[4 * qq + i] = c3[i]; }
            }
            bnred_add_any(rt, val, zup[v]);
          }
        }
      }
    }
    // (the next tile writes dz_s / x_s, which every wave left before B2, and staging again only behind its own B1)
  };

  // Pairs of tiles in the loop, an odd last tile behind it: every path to the loop head passes BOTH sets' requests, so the wait
  // for set A at the head is "all but set B's loads" by count (with a conditional second tile inside the loop the compiler has to
  // assume a back edge that issued nothing behind set A's request, and that wait becomes a wait for everything in flight)
  int tile = wg;
  for (; tile + G_ < p.ntiles; tile += 2 * G_) {
    process(RA, tile);
    process(RB, tile + G_);
  }
  if (tile < p.ntiles) process(RA, tile);

  if constexpr (RED) {
    __syncthreads();                                   // the last tile's staging reads are done: LDS is free
    bnred_flush<NT, XV>(rt, p.red, 0, (float*)smem, tid, wg % PLYOLO_STAT_SLOTS);
  }

  // ---- k-split wave groups (narrow layers): fold the partial accumulators of groups 1 .. WK-1 into group 0 through LDS
  if constexpr (WK > 1) {
    __syncthreads();                                   // the last tile's staging reads are done: LDS is free
    float* part = (float*)smem;                        // [(wk-1)][wco][wci][MTC][MTI][16 regs][64 lanes]
    if (wk > 0) {
#pragma unroll
      for (int a = 0; a < MTC; ++a)
#pragma unroll
        for (int b = 0; b < MTI; ++b)
#pragma unroll
          for (int i = 0; i < 16; ++i)
            part[((((size_t)(wk - 1) * WCO + wco) * WCI + wci) * MTC * MTI + a * MTI + b) * 1024 + i * 64 + lane] = accw[a][b][i];
    }
    __syncthreads();
    if (wk == 0) {
#pragma unroll
      for (int k2 = 1; k2 < WK; ++k2)
#pragma unroll
        for (int a = 0; a < MTC; ++a)
#pragma unroll
          for (int b = 0; b < MTI; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i)
              accw[a][b][i] += part[((((size_t)(k2 - 1) * WCO + wco) * WCI + wci) * MTC * MTI + a * MTI + b) * 1024 + i * 64 + lane];
    }
  }

  // ---- private slab of this workgroup: D[row = co][col = ci], col = lane & 31, row = (i&3) + 8*(i>>2) + 4*h
  if (wk == 0) {
    float* slab = p.dw + (size_t)wg * ((size_t)CO * CI);
#pragma unroll
    for (int b = 0; b < MTI; ++b) {
      const int ci = (wci * MTI + b) * 32 + r;
#pragma unroll
      for (int a = 0; a < MTC; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int co = (wco * MTC + a) * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
          slab[(size_t)co * CI + ci] = accw[a][b][i];
        }
    }
  }
}

// pixel-tile rows per channel count: bigger tiles for the narrow layers (their rows are short, a tile should still be several
// tens of KB of loads in flight)
constexpr int pb_bm(int co, int ci) { return (co >= 128 && ci >= 128) ? 64 : ((co >= 128 || ci >= 128 || (co == 64 && ci == 64)) ? 128 : 256); }

template <int CO, int CI>
hipError_t pb_launch_inst(const PbP& p, int G_, hipStream_t s) {
  constexpr int BM = pb_bm(CO, CI);
  using G = PbGeom<CO, CI, BM>;
  auto kern = p.act == PLYOLO_ACT_SILU ? conv_pw_bwd_kernel<CO, CI, BM, PLYOLO_ACT_SILU> : conv_pw_bwd_kernel<CO, CI, BM, -1>;
  if (p.red.n > 0) kern = p.act == PLYOLO_ACT_SILU ? conv_pw_bwd_kernel<CO, CI, BM, PLYOLO_ACT_SILU, true> : conv_pw_bwd_kernel<CO, CI, BM, -1, true>;
  if (hipError_t e = plyolo::ensure_dynamic_lds((const void*)kern, G::LDS); e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(G_), dim3(NT), G::LDS, s, p);
  return hipGetLastError();
}

struct PbPlan { int ok, bm, ntiles, G, WK; };

PbPlan pb_plan(const plyolo_conv_desc* d) {
  PbPlan w{};
  const int co = d->Cout, ci = d->Cin;
  const bool shape = (co == 32 || co == 64 || co == 128) && (ci == 32 || ci == 64 || ci == 128);
  // instantiated: the square shapes and the 2:1 / 1:2 ones next to them
  const bool inst = shape && (co == ci || co == 2 * ci || ci == 2 * co);
  if (!inst) return w;
  w.ok = 1;
  w.bm = pb_bm(co, ci);
  const size_t M = (size_t)d->N * d->H * d->W;
  w.ntiles = (int)((M + w.bm - 1) / w.bm);
  int gmax = 256;      // one 512-thread workgroup per CU
  // every workgroup leaves a CO x CI fp32 slab: at most ~1/8 of the tensor bytes the launch moves
  const double moved = (double)M * (2.0 * co + 2.0 * ci) * 2.0, slab = 4.0 * co * ci;
  const double frac = getenv("PLYOLO_PWBWD_SLAB_FRAC") ? atof(getenv("PLYOLO_PWBWD_SLAB_FRAC")) : 0.125;
  w.WK = 1;            // k-split wave groups fold their partial sums inside the kernel: one slab per workgroup
  const int g_budget = (int)(moved * frac / (slab * w.WK));
  if (gmax > g_budget) gmax = g_budget;
  if (gmax < 1) gmax = 1;
  if (gmax > w.ntiles) gmax = w.ntiles;
  const int rounds = (w.ntiles + gmax - 1) / gmax;
  w.G = (w.ntiles + rounds - 1) / rounds;      // the same number of rounds with an even share per workgroup
  if (const char* e = getenv("PLYOLO_PWBWD_G")) { const int v = atoi(e); if (v > 0) w.G = v < w.ntiles ? v : w.ntiles; }   // tests: forced grid
  return w;
}

}  // namespace

namespace plyolo {

bool conv_pw_enabled();

// 1 when plyolo_conv2d_bwd_pw covers this unit: bf16 pointwise stride-1, a cheap activation, Cout and Cin in {32, 64, 128}
// (square or 2:1), and an output gradient of at least PLYOLO_PWBWD_MIN_MB (default 20: below that the three separate launches are
// as fast -- 13 MB: 28 us against 21 for dz + data gradient, 26 MB: 34 against 29, 52 MB: 47 against 60, 105 MB: 86 against 101 --
// and the step, same box, three alternations: off 9.15 ms, >= 12 MB 9.01, >= 20 MB 8.99, >= 40 MB 9.00)
int conv_pw_bwd_fits(const plyolo_conv_desc* d, int act) {
  if (!conv_pw_enabled() || d->dtype != PLYOLO_BF16 || d->ksize != 1 || d->stride != 1) return 0;
  if (act < PLYOLO_ACT_NONE || act > PLYOLO_ACT_LRELU) return 0;
  const double min_mb = getenv("PLYOLO_PWBWD_MIN_MB") ? atof(getenv("PLYOLO_PWBWD_MIN_MB")) : 20.0;
  if ((double)d->N * d->H * d->W * d->Cout * 2.0 < min_mb * 1.0e6) return 0;
  return pb_plan(d).ok;
}

int conv_pw_bwd_slabs(const plyolo_conv_desc* d) {
  const PbPlan w = pb_plan(d);
  return w.ok ? w.G * w.WK : 0;
}

int conv_pw_bwd(const plyolo_conv_desc* d, const plyolo_bn_bwd_fuse* f, const void* x, const void* wpd, void* dx, int accumulate,
                float* dwp, const plyolo_bn_red* red, void* stream) {
  const PbPlan w = pb_plan(d);
  PbP p{};
  p.dout = (const bf16_t*)f->dout; p.dout_ld = f->dout_ld;
  p.dout2 = (const bf16_t*)f->dout2; p.dout2_ld = f->dout2_ld; p.dsplit = f->dout2 ? f->dout_split : 0;
  p.z = (const bf16_t*)f->z; p.z_ld = f->z_ld;
  p.x = (const bf16_t*)x; p.x_ld = d->x_ld;
  p.wpd = (const bf16_t*)wpd;
  p.dx = (bf16_t*)dx; p.dx_ld = d->x_ld;       // the gradient matrix of x mirrors x (same pitch)
  p.dw = dwp;
  p.coef = f->coef; p.bslots = f->bslots;
  p.gamma = f->gamma; p.dgamma = f->dgamma; p.dbeta = f->dbeta;
  p.psplit = f->par_split; p.gamma2 = f->gamma2; p.dgamma2 = f->dgamma2; p.dbeta2 = f->dbeta2;
  p.M = d->N * d->H * d->W;
  p.ntiles = w.ntiles;
  p.act = f->act;
  p.accumulate = accumulate;
  const bool use_red = red && red->n > 0;
  if (use_red) p.red = *red;
  const int co = d->Cout, ci = d->Cin, G_ = w.G;
  {
    char lab[64];
    snprintf(lab, sizeof(lab), "conv_pw_bwd<%dx%d>%s", co, ci, use_red ? "+bnred" : "");
    annotate(lab, 4.0 * p.M * (double)co * ci, (double)p.M * (2.0 * co + ci * ((accumulate ? 3.0 : 2.0) + (use_red ? 1.0 : 0.0))) * 2.0 + 4.0 * co * ci);
  }
  return submit(stream, [=](hipStream_t s) -> hipError_t {
#define PB_CASE(a, b) if (co == a && ci == b) return pb_launch_inst<a, b>(p, G_, s);
    PB_CASE(32, 32) PB_CASE(64, 64) PB_CASE(128, 128)
    PB_CASE(64, 32) PB_CASE(32, 64) PB_CASE(128, 64) PB_CASE(64, 128)
#undef PB_CASE
    return hipErrorInvalidValue;
  });
}

}  // namespace plyolo
