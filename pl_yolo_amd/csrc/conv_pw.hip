// bf16 MFMA pointwise (1x1, stride 1) convolution for gfx950: forward and data gradient.
//
// Replaces nn.Conv2d(kernel_size=1) inside BaseConv (reference models/layers/network_blocks.py:18-26), the
// prediction convs of DecoupledHead (models/heads/decoupled_head.py:43-62) and what ATen's convolution_backward
// computes for their inputs.  41 of the 69 convolutions of YOLOX-s are pointwise, all of them HBM- or
// latency-bound:   Y[M, N] = X[M, K] . W[N, K]^T   with M = N*H*W pixels (12.8 k ... 3.3 M), K, N in 32 ... 1024.
//
// The 3x3 kernel (conv_mfma.hip) serves them as a degenerate one-tap case, but pays for its halo machinery on
// every launch: per-thread halo descriptors, 32-channel chunks with one barrier AND one exposed memory round trip
// per chunk (a chunk of a 1x1 layer is only 8 MFMAs per wave), 24 KB of loads in flight per CU.  Here:
//   * one workgroup = 128 pixel rows x BN output channels, K in chunks of 64 channels (16 KB of rows); the requests
//     of chunk i+1 -- weights and rows -- are issued before the MFMAs of chunk i, so only the first round trip of a
//     tile is exposed (whole-K tiles of 128 / 256 channels are instantiated too: equally fast alone, slower in the
//     step, see pw_tiles);
//   * A (pixels): 16-byte vectors of full rows -> registers -> [optional BatchNorm + activation of the producing
//     layer, "lazy input"] -> LDS (row pitch K*2+16 bytes: conflict-free ds_read_b128 fragments);
//   * B (weights): never touches LDS -- fragment-ordered pack (conv_mfma.hip), one coalesced KiB per fragment
//     straight into registers, requested BEFORE the pixel rows so both are in flight together;
//   * epilogue as in the 3x3 kernel: BatchNorm sum / sum-of-squares partials -> fp64 stat slots, bias, LDS
//     transpose, whole 16-byte channel vectors to HBM, strided store (y_ld) = free concat, accumulate (dgrad).
// Workgroups sharing a pixel tile (different BN blocks) are adjacent in the XCD-aware tile order: the rows are
// re-read from that XCD's L2, not from HBM.
#include <stdlib.h>

#include "common.h"
#include "bnred.h"

namespace {

struct PwP {
  const bf16_t* x;
  const bf16_t* w;       // fragment-native pack [nb = N/32][kb = K/16][64 lanes][8] (one tap)
  void* y;
  const float* bias;
  const float* ep_coef;  // inference: fused BatchNorm (scale | shift) + activation in the epilogue, or NULL
  int ep_act;
  const bf16_t* ep_res;  // fused epilogue only: residual added after the activation, or NULL
  int ep_res_ld;
  double* stats;         // fp64 stat slots [PLYOLO_STAT_SLOTS][2][N] or NULL
  const float* pre;      // lazy input: x is a raw conv output, rows are staged as act(x * pre[c] + pre[pre_ld + c])
  int pre_ld, pre_act;
  int M, K, N, x_ld, y_ld;
  int nkb, nnb;          // packed weight geometry: 16-channel k-blocks, 32-channel n-blocks
  int nmt, nnblk;        // pixel tiles, BN blocks
  int accumulate;
  int pf;                // 1: requests of chunk i+1 before the MFMAs of chunk i (PLYOLO_PW_PF, A/B switch)
  int pipe;              // 1: the software-pipelined chunk loop, 0: the plain one (PLYOLO_PW_LOOP, A/B switch)
  // BNB instances (plyolo_conv2d_dgrad_bn): x is the gradient of the unit's ACTIVATED output (channels >= bsplit in a second
  // matrix for merged pairs); the rows are staged as dz = A*du + B*z + Cc, du = x * act'(z*sc + sh)  -- exactly bn_act_bwd_dz
  // (bn.hip) -- and the workgroups of BN block 0 also write dz for the weight gradient
  const bf16_t* bz;      // raw conv output z of the unit [M][K], pitch bz_ld
  int bz_ld;
  const bf16_t* bx2;
  int bx2_ld, bsplit;
  const float* bcoef;    // (scale | shift | mean | invstd) [4][K] of the forward
  const double* bslots;  // fp64 backward stat slots [PLYOLO_STAT_SLOTS][2][K] (sum du, sum du*zhat)
  const float *bgamma, *bgamma2;
  float *bdgamma, *bdbeta, *bdgamma2, *bdbeta2;
  int bpsplit, bact;
  bf16_t* bdz;
  int bdz_ld;
  plyolo_bn_red red;     // RED instances (data gradients): BatchNorm-backward reduction of the unit(s) whose output gradient this launch completes
};

// derivative of the three cheap activations (the BNB loader; hswish / gelu units keep the separate bn_act_bwd_dz launch)
DEVINL float pw_act_grad(float u, int act) {
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

DEVINL u32x4 pw_add_bf16x8(u32x4 a, u32x4 b) {
  u32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float lo = __uint_as_float(a[i] << 16) + __uint_as_float(b[i] << 16);
    float hi = __uint_as_float(a[i] & 0xffff0000u) + __uint_as_float(b[i] & 0xffff0000u);
    r[i] = pack2bf(lo, hi);
  }
  return r;
}

constexpr int PW_BM = 128;

// BNB instances run 64-row tiles where the wave layout allows (BN >= 64): half the accumulators and half the rows in flight per
// thread pay for the loader's second operand and its coefficient registers (207 -> 3 waves per SIMD without spilling)
template <int BN, bool BNB> constexpr int pw_bm() { return (BNB && BN >= 64) ? 64 : PW_BM; }

// RED: the bf16 store loop also folds the BatchNorm-backward reduction of the upstream unit(s) (bnred.h)
// BACT (BNB instances): PLYOLO_ACT_SILU = the activation of the unit is SiLU at compile time (a switch on a run-time activation
// inside the unrolled element loop of the loader compiles to a branch per element), -1 = p.bact
// XCD-aware bijective remap (workgroups with equal blockIdx.x % 8 share an XCD's L2): every XCD gets a contiguous run
// of tiles, BN blocks of one pixel tile adjacent
DEVINL int pw_tile_of_block() {
  const int nwg = (int)gridDim.x, bid = (int)blockIdx.x;
  const int q = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  return (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
}

// tile: position in the XCD-aware order; mt_i: pixel tile; nb0: first 32-channel block of this workgroup's output columns (a kernel that
// mixes block widths hands in something else than nblk * WN: conv_pw_rag_kernel below)
template <int BN, int KC, bool OUT_F32, bool PRE, bool PIPE, bool BNB = false, bool RED = false, int BACT = -1>
DEVINL void conv_pw_body(const PwP& p, const int tile, const int nblk, const int mt_i, const int nb0) {
  static_assert(!RED || (!OUT_F32 && !PRE), "RED instances: bf16 data gradients");
  constexpr int BM = pw_bm<BN, BNB>();
  constexpr int WN = BN / 32, WM = 4 / WN, MT = BM / (32 * WM);
  constexpr int ROWB = KC * 2 + 16;   // LDS row pitch (bytes)
  constexpr int CV = KC / 8;          // 16-byte vectors per row
  constexpr int KS = KC / 16;         // k-steps per chunk
  constexpr int RPP = 256 / CV;       // rows covered by one pass of the 256 threads
  constexpr int NV = BM / RPP;        // vectors per thread and chunk
  extern __shared__ __align__(16) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;

  const int m0 = mt_i * BM;
  const int cout0 = nb0 * 32;
  const int nb = nb0 + wn;   // this wave's 32-channel block of the packed weights
  const bool nb_ok = nb < p.nnb;

  f32x16 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;

  const char* wbase = (const char*)p.w + (size_t)((nb_ok ? nb : p.nnb - 1) * p.nkb) * 1024u + (size_t)lane * 16u;
  const int cvt = tid % CV, row0 = tid / CV;
  const int nchunks = (p.K + KC - 1) / KC;
  int arow[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) arow[mt] = ((wm * MT + mt) * 32 + r) * ROWB + h * 16;

  if constexpr (!PIPE) {
  // plain loop: weights first (L2-resident, tiny), then the pixel rows of the chunk, all requested at the top of the iteration
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const int c0 = chunk * KC;
    u32x4 bq[KS];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      const int kb = chunk * KS + kk;
      bq[kk] = *(const u32x4*)(wbase + (size_t)(kb < p.nkb ? kb : p.nkb - 1) * 1024u);
    }
    const int c = c0 + cvt * 8;
    const bool cok = c < p.K;
    u32x4 av[NV];
    [[maybe_unused]] u32x4 zv[NV];
    if constexpr (BNB) {
      // this thread's channel vector of the output gradient lives in one of the two matrices of a merged pair
      const bool second = p.bsplit > 0 && c >= p.bsplit;
      const bf16_t* dsrc = second ? p.bx2 + (c - p.bsplit) : p.x + c;
      const int dld = second ? p.bx2_ld : p.x_ld;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int m = m0 + row0 + v * RPP;
        const bool ok = cok && m < p.M;
        av[v] = *(const u32x4*)(ok ? dsrc + (size_t)m * dld : p.x);
        zv[v] = *(const u32x4*)(p.bz + (ok ? (size_t)m * p.bz_ld + c : 0));
      }
      if (chunk == 0) {
        // per-channel table (scale, shift, A, B, Cc) behind the row tile, built while the first rows are in flight
        float* tab = (float*)(smem + BM * ROWB);
        const int Kp = p.K;
        for (int ch = tid; ch < Kp; ch += 256) {
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
          const double cnt = (double)p.M;
          const float A = (gam ? gam[cp] : 1.f) * invstd;
          const float B = (float)(-(double)A * (suz / cnt) * (double)invstd);
          tab[ch] = p.bcoef[ch];
          tab[Kp + ch] = p.bcoef[Kp + ch];
          tab[2 * Kp + ch] = A;
          tab[3 * Kp + ch] = B;
          tab[4 * Kp + ch] = (float)(-(double)A * (su / cnt) - (double)B * (double)mean);
          if (tile == 0) {   // dbeta = sum du, dgamma = sum du*zhat (bn_act_bwd_dz publishes them from its workgroup 0)
            float* db_ = sec ? p.bdbeta2 : p.bdbeta;
            float* dg_ = sec ? p.bdgamma2 : p.bdgamma;
            if (db_) db_[cp] = (float)su;
            if (dg_) dg_[cp] = (float)suz;
          }
        }
        __syncthreads();
      }
    } else {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int m = m0 + row0 + v * RPP;
      const bool ok = cok && m < p.M;
      av[v] = *(const u32x4*)(p.x + (ok ? (size_t)m * p.x_ld + c : 0));
    }
    }
    if (chunk) __syncthreads();   // every wave is done with the previous chunk's rows
    if constexpr (BNB) {
      const float* tab = (const float*)(smem + BM * ROWB);
      const int Kp = p.K, cc = cok ? c : 0;
      float sc[8], sh[8], A[8], B[8], Cc[8];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const f32x4 a0 = *(const f32x4*)(tab + cc + 4 * q), a1 = *(const f32x4*)(tab + Kp + cc + 4 * q);
        const f32x4 a2 = *(const f32x4*)(tab + 2 * Kp + cc + 4 * q), a3 = *(const f32x4*)(tab + 3 * Kp + cc + 4 * q);
        const f32x4 a4 = *(const f32x4*)(tab + 4 * Kp + cc + 4 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i) { sc[4 * q + i] = a0[i]; sh[4 * q + i] = a1[i]; A[4 * q + i] = a2[i]; B[4 * q + i] = a3[i]; Cc[4 * q + i] = a4[i]; }
      }
      const bool wr = nblk == 0;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        u32x4 t = av[v];
        const u32x4 zz = zv[v];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float zl = __uint_as_float(zz[i] << 16), zh = __uint_as_float(zz[i] & 0xffff0000u);
          const float dl = __uint_as_float(t[i] << 16), dh = __uint_as_float(t[i] & 0xffff0000u);
          const float dul = dl * pw_act_grad(fmaf(zl, sc[2 * i], sh[2 * i]), BACT >= 0 ? BACT : p.bact);
          const float duh = dh * pw_act_grad(fmaf(zh, sc[2 * i + 1], sh[2 * i + 1]), BACT >= 0 ? BACT : p.bact);
          t[i] = pack2bf(fmaf(A[2 * i], dul, fmaf(B[2 * i], zl, Cc[2 * i])), fmaf(A[2 * i + 1], duh, fmaf(B[2 * i + 1], zh, Cc[2 * i + 1])));
        }
        av[v] = t;
        const int m = m0 + row0 + v * RPP;
        if (wr && cok && m < p.M) *(u32x4*)(p.bdz + (size_t)m * p.bdz_ld + c) = t;
      }
    }
    if constexpr (PRE) {
      float sc[8], sh[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        sc[i] = cok ? p.pre[c + i] : 0.f;
        sh[i] = cok ? p.pre[p.pre_ld + c + i] : 0.f;
      }
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        av[v] = bn_act_vec8(av[v], sc, sh, p.pre_act);
      }
    }
    {
      const u32x4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int row = row0 + v * RPP;
        const bool ok = cok && (m0 + row) < p.M;
        *(u32x4*)(smem + row * ROWB + cvt * 16) = ok ? av[v] : zero;
      }
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      const bf16x8 b = *(const bf16x8*)&bq[kk];
      bf16x8 a[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) a[mt] = *(const bf16x8*)(smem + arow[mt] + kk * 32);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt], b, acc[mt], 0, 0, 0);
    }
  }
  } else {
  // one chunk = KC channels of the tile's 128 rows.  Requests of chunk i+1 (weights first -- L2-resident, tiny -- then the
  // pixel rows) are issued BEFORE the MFMAs of chunk i, so a memory round trip is only exposed for the first chunk
  u32x4 bq[KS], av[NV];
  auto request = [&](const int chunk) {
    const int c = chunk * KC + cvt * 8;
    const bool cok = c < p.K;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      const int kb = chunk * KS + kk;
      // a k-block beyond K meets zero-filled LDS columns (finite weights x 0 = 0): clamp instead of masking
      bq[kk] = *(const u32x4*)(wbase + (size_t)(kb < p.nkb ? kb : p.nkb - 1) * 1024u);
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int m = m0 + row0 + v * RPP;
      const bool ok = cok && m < p.M;
      av[v] = *(const u32x4*)(p.x + (ok ? (size_t)m * p.x_ld + c : 0));
    }
  };
  request(0);
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const int c = chunk * KC + cvt * 8;
    const bool cok = c < p.K;
    if (chunk) __syncthreads();   // every wave is done with the previous chunk's rows
    if constexpr (PRE) {
      float sc[8], sh[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        sc[i] = cok ? p.pre[c + i] : 0.f;
        sh[i] = cok ? p.pre[p.pre_ld + c + i] : 0.f;
      }
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        av[v] = bn_act_vec8(av[v], sc, sh, p.pre_act);
      }
    }
    {
      const u32x4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int row = row0 + v * RPP;
        const bool ok = cok && (m0 + row) < p.M;
        *(u32x4*)(smem + row * ROWB + cvt * 16) = ok ? av[v] : zero;
      }
    }
    u32x4 bc[KS];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) bc[kk] = bq[kk];
    __syncthreads();
    if (p.pf && chunk + 1 < nchunks) request(chunk + 1);
    __builtin_amdgcn_sched_barrier(0);   // keep the next chunk's requests ABOVE the MFMA block
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      const bf16x8 b = *(const bf16x8*)&bc[kk];
      bf16x8 a[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) a[mt] = *(const bf16x8*)(smem + arow[mt] + kk * 32);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt], b, acc[mt], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (!p.pf && chunk + 1 < nchunks) request(chunk + 1);
  }
  }
  __syncthreads();  // all LDS operand reads retired; LDS is reused for the epilogue

  // ---- epilogue ---------------------------------------------------------------
  constexpr int SROW = OUT_F32 ? (BN + 4) * 4 : (BN * 2 + 16);  // staging row pitch (bytes)
  float* red = (float*)(smem + BM * SROW);                       // [WM][2][BN]

  if (p.stats != nullptr) {
    float s1 = 0.f, s2 = 0.f;
    if (m0 + BM <= p.M) {
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
          const float v = (m0 + m < p.M) ? acc[mt][i] : 0.f;
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
  const bool fused = !OUT_F32 && p.ep_coef != nullptr;
  float ep_sc = 1.f, ep_sh = 0.f;
  if (fused) {
    const int co = cout0 + wn * 32 + r;
    if (co < p.N) { ep_sc = p.ep_coef[co]; ep_sh = p.ep_coef[p.N + co]; }
  }
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
  __syncthreads();

  if (p.stats != nullptr && tid < BN) {
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int w = 0; w < WM; ++w) {
      s += red[(w * 2 + 0) * BN + tid];
      ss += red[(w * 2 + 1) * BN + tid];
    }
    const int co = cout0 + tid;
    if (co < p.N) {  // one fp64 add per workgroup and channel (agent scope); the order cannot change the fp32 result
      double* slot = p.stats + (size_t)(mt_i % PLYOLO_STAT_SLOTS) * 2 * p.N;
      __hip_atomic_fetch_add(slot + co, (double)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(slot + p.N + co, (double)ss, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }

  if (OUT_F32) {
    // fp32 rows (the raw head maps, pitch 5+C floats) are only 4-byte aligned: dwordx4 stores with a 4-byte aligned type
    typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
    float* y = (float*)p.y;
    constexpr int VPR4 = BN / 4;
    for (int idx = tid; idx < BM * VPR4; idx += 256) {
      const int m = idx / VPR4, v = idx - m * VPR4;
      const int co = cout0 + v * 4;
      if (m0 + m < p.M && co < p.N) {
        f32x4 val = *(const f32x4*)(smem + m * SROW + v * 16);
        float* dst = y + (size_t)(m0 + m) * p.y_ld + co;
        if (co + 4 <= p.N) {
          if (p.bias) val += *(const f32x4_u*)(p.bias + co);
          if (p.accumulate) val += *(const f32x4_u*)dst;
          *(f32x4_u*)dst = val;
        } else {
          for (int j = 0; co + j < p.N; ++j) {
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
      // every thread owns ONE channel vector (256 % VPR == 0) and NIT rows; the upstream unit's z vectors and the old dx rows of an
      // accumulating launch are requested up front -- one memory round trip for the whole loop
      constexpr int NIT = BM * VPR / 256;
      static_assert(256 % VPR == 0 && (BM * VPR) % 256 == 0, "RED: whole rows per thread");
      const int v = tid % VPR, co = cout0 + v * 8;
      BnRedThread rt;
      bnred_init(rt, p.red, co);
      u32x4 zq[NIT], old[NIT];
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int m = (tid + it * 256) / VPR;
        const bool ok = m0 + m < p.M && co < p.N;
        const size_t pix = ok ? (size_t)(m0 + m) : 0;
        zq[it] = rt.z ? bnred_load(rt, pix) : u32x4{0u, 0u, 0u, 0u};
        old[it] = p.accumulate ? *(const u32x4*)(y + pix * p.y_ld + (co < p.N ? co : 0)) : u32x4{0u, 0u, 0u, 0u};
      }
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int m = (tid + it * 256) / VPR;
        if (m0 + m < p.M && co < p.N) {
          u32x4 val = *(const u32x4*)(smem + m * SROW + v * 16);
          if (p.accumulate) val = pw_add_bf16x8(old[it], val);
          *(u32x4*)(y + (size_t)(m0 + m) * p.y_ld + co) = val;
          if (rt.z) bnred_add_any(rt, val, zq[it]);
        }
      }
      __syncthreads();   // every thread is done with the staging rows: the fold reuses them
      bnred_flush<256, VPR>(rt, p.red, cout0, (float*)smem, tid, mt_i % PLYOLO_STAT_SLOTS);
    } else {
    for (int idx = tid; idx < BM * VPR; idx += 256) {
      const int m = idx / VPR, v = idx - m * VPR;
      const int co = cout0 + v * 8;
      if (m0 + m < p.M && co < p.N) {
        u32x4 val = *(const u32x4*)(smem + m * SROW + v * 16);
        const size_t pix = (size_t)(m0 + m);
        bf16_t* dst = y + pix * p.y_ld + co;
        if (p.ep_res) val = pw_add_bf16x8(*(const u32x4*)(p.ep_res + pix * p.ep_res_ld + co), val);
        if (p.accumulate) val = pw_add_bf16x8(*(const u32x4*)dst, val);
        *(u32x4*)dst = val;
      }
    }
    }
  }
}

template <int BN, int KC, bool OUT_F32, bool PRE, bool PIPE, bool BNB = false, bool RED = false, int BACT = -1>
__global__ __launch_bounds__(256, BNB ? 3 : 2) void conv_pw_kernel(const PwP p) {
  const int tile = pw_tile_of_block();
  const int nblk = tile % p.nnblk, mt_i = tile / p.nnblk;
  conv_pw_body<BN, KC, OUT_F32, PRE, PIPE, BNB, RED, BACT>(p, tile, nblk, mt_i, nblk * (BN / 32));
}

// RAGGED output-channel blocks (plain bf16 forward / data gradient): nnblk - 1 full 128-channel blocks and ONE narrower block (REM = 32 or 64
// channels) for what a 160- / 320-channel layer leaves behind them -- the plain launch runs a whole 128-channel block of MFMAs there for 32
// / 64 real columns (YOLOX-x: every pointwise layer).  Both widths in one kernel: the blocks of a pixel tile stay neighbours on one XCD
template <int REM, int KC>
__global__ __launch_bounds__(256, 2) void conv_pw_rag_kernel(const PwP p) {
  const int tile = pw_tile_of_block();
  const int nblk = tile % p.nnblk, mt_i = tile / p.nnblk;
  if (nblk + 1 < p.nnblk) conv_pw_body<128, KC, false, false, false>(p, tile, nblk, mt_i, nblk * 4);
  else conv_pw_body<REM, KC, false, false, false>(p, tile, nblk, mt_i, nblk * 4);
}

template <int REM, int KC>
hipError_t pw_launch_rag_inst(const PwP& p, hipStream_t s) {
  constexpr int ROWB = KC * 2 + 16;
  auto epi = [](int BN) { return (size_t)PW_BM * (BN * 2 + 16) + (size_t)(4 / (BN / 32)) * 2 * BN * 4; };
  const size_t lds_main = (size_t)PW_BM * ROWB, lds_epi = epi(128) > epi(REM) ? epi(128) : epi(REM);
  const size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
  auto kern = conv_pw_rag_kernel<REM, KC>;
  if (hipError_t e = plyolo::ensure_dynamic_lds((const void*)kern, lds); e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(p.nmt * p.nnblk), dim3(256), lds, s, p);
  return hipGetLastError();
}

// N = 128 * nfull + rem, nfull >= 1, 0 < rem <= 64, plain loop, 64-channel chunks, bf16 output, no lazy input (PLYOLO_RAG=0: off)
bool pw_use_rag(const PwP& p, int BN, int KC, bool plain_bf16) {
  const int on = getenv("PLYOLO_RAG") ? atoi(getenv("PLYOLO_RAG")) : 1;      // (read per call: the tests switch it)
  const int rem = p.N % 128;
  return on && plain_bf16 && BN == 128 && KC == 64 && !p.pipe && p.N > 128 && rem > 0 && rem <= 64;
}
hipError_t pw_launch_rag(const PwP& p, hipStream_t s) { return p.N % 128 <= 32 ? pw_launch_rag_inst<32, 64>(p, s) : pw_launch_rag_inst<64, 64>(p, s); }

template <int BN, int KC, bool OUT_F32, bool PRE>
hipError_t pw_launch_inst(const PwP& p, hipStream_t s) {
  constexpr int WN = BN / 32, WM = 4 / WN;
  constexpr int ROWB = KC * 2 + 16;
  constexpr int SROW = OUT_F32 ? (BN + 4) * 4 : (BN * 2 + 16);
  const size_t lds_main = (size_t)PW_BM * ROWB;
  const size_t lds_epi = (size_t)PW_BM * SROW + WM * 2 * BN * 4;
  const size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
  auto kern = p.pipe ? conv_pw_kernel<BN, KC, OUT_F32, PRE, true> : conv_pw_kernel<BN, KC, OUT_F32, PRE, false>;
  if (hipError_t e = plyolo::ensure_dynamic_lds((const void*)kern, lds); e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(p.nmt * p.nnblk), dim3(256), lds, s, p);
  return hipGetLastError();
}

// RED instances: plain chunk loop, 32- / 64-channel chunks (what pw_tiles picks for a data gradient unless PLYOLO_PW_KCMAX widens them)
template <int BN, int KC>
hipError_t pw_launch_red_inst(const PwP& p, hipStream_t s) {
  constexpr int WN = BN / 32, WM = 4 / WN;
  constexpr int ROWB = KC * 2 + 16, SROW = BN * 2 + 16;
  const size_t lds_main = (size_t)PW_BM * ROWB, lds_epi = (size_t)PW_BM * SROW + WM * 2 * BN * 4;
  size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
  lds = lds > 16384 ? lds : 16384;     // bnred_flush scratch (aliases the staging rows)
  auto kern = conv_pw_kernel<BN, KC, false, false, false, false, true>;
  if (hipError_t e = plyolo::ensure_dynamic_lds((const void*)kern, lds); e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(p.nmt * p.nnblk), dim3(256), lds, s, p);
  return hipGetLastError();
}
bool pw_red_has(int BN, int KC) { return (BN == 32 || BN == 64 || BN == 128) && (KC == 32 || KC == 64); }
hipError_t pw_launch_red(const PwP& p, int BN, int KC, hipStream_t s) {
#define PW_RCASE(bn, kc) \
  if (BN == bn && KC == kc) return pw_launch_red_inst<bn, kc>(p, s);
  PW_RCASE(32, 32) PW_RCASE(32, 64) PW_RCASE(64, 32) PW_RCASE(64, 64) PW_RCASE(128, 32) PW_RCASE(128, 64)
#undef PW_RCASE
  return hipErrorInvalidValue;
}

// BNB instances: bf16 output, plain chunk loop, 32- / 64-channel chunks
template <int BN, int KC>
hipError_t pw_launch_bnb_inst(const PwP& p, hipStream_t s) {
  constexpr int WN = BN / 32, WM = 4 / WN;
  constexpr int ROWB = KC * 2 + 16, SROW = BN * 2 + 16;
  constexpr int BM = pw_bm<BN, true>();
  const size_t lds_main = (size_t)BM * ROWB + (size_t)5 * p.K * 4;   // row tile + the per-channel table
  const size_t lds_epi = (size_t)BM * SROW + WM * 2 * BN * 4;
  const size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
  auto kern = p.bact == PLYOLO_ACT_SILU ? conv_pw_kernel<BN, KC, false, false, false, true, false, PLYOLO_ACT_SILU> : conv_pw_kernel<BN, KC, false, false, false, true>;
  size_t lds_ = lds;
  if (p.red.n > 0) {   // + the upstream unit's BatchNorm-backward reduction in the store loop
    kern = p.bact == PLYOLO_ACT_SILU ? conv_pw_kernel<BN, KC, false, false, false, true, true, PLYOLO_ACT_SILU> : conv_pw_kernel<BN, KC, false, false, false, true, true>;
    lds_ = lds > 16384 ? lds : 16384;
  }
  if (hipError_t e = plyolo::ensure_dynamic_lds((const void*)kern, lds_); e != hipSuccess) return e;
  const int nmt = (p.M + BM - 1) / BM;
  PwP q = p;
  q.nmt = nmt;
  hipLaunchKernelGGL(kern, dim3(nmt * p.nnblk), dim3(256), lds_, s, q);
  return hipGetLastError();
}

hipError_t pw_launch_bnb(const PwP& p, int BN, int KC, hipStream_t s) {
#define PW_BCASE(bn, kc) \
  if (BN == bn && KC == kc) return pw_launch_bnb_inst<bn, kc>(p, s);
  PW_BCASE(32, 32) PW_BCASE(32, 64) PW_BCASE(64, 32) PW_BCASE(64, 64) PW_BCASE(128, 32) PW_BCASE(128, 64)
#undef PW_BCASE
  return hipErrorInvalidValue;
}

template <bool OUT_F32, bool PRE>
hipError_t pw_launch(const PwP& p, int BN, int KC, hipStream_t s) {
#define PW_CASE(bn, kc) \
  if (BN == bn && KC == kc) return pw_launch_inst<bn, kc, OUT_F32, PRE>(p, s);
  PW_CASE(32, 32) PW_CASE(32, 64) PW_CASE(32, 128) PW_CASE(32, 256)
  PW_CASE(64, 32) PW_CASE(64, 64) PW_CASE(64, 128) PW_CASE(64, 256)
  if constexpr (!OUT_F32) { PW_CASE(128, 32) PW_CASE(128, 64) PW_CASE(128, 128) PW_CASE(128, 256) }
#undef PW_CASE
  return hipErrorInvalidValue;
}

void pw_tiles(PwP& p, bool out_f32, int* BN, int* KC) {
  int bn = p.N > 64 ? 128 : (p.N > 32 ? 64 : 32);
  if (out_f32 && bn > 64) bn = 64;
  // Measured on the whole training step (YOLOX-s B=32, PLYOLO_PW_KC sweep): 64-channel chunks beat 128 / 256 (9.96 vs
  // 10.29 / 10.33 ms) although the pointwise launches alone are equally fast -- the lighter workgroup (18 KB of LDS,
  // ~120 VGPRs) leaves more of each CU to the weight-gradient lane that runs beside the data-gradient chain
  int kc = p.K > 32 ? 64 : 32;
  if (const char* e = getenv("PLYOLO_PW_KCMAX")) { const int v = atoi(e); if (v == 128 || v == 256) kc = p.K > 128 ? (v == 256 ? 256 : 128) : (p.K > 64 ? 128 : kc); }
  if (const char* e = getenv("PLYOLO_PW_KC")) { const int v = atoi(e); if (v == 32 || v == 64 || v == 128 || v == 256) kc = v < kc ? v : kc; }
  *BN = bn;
  *KC = kc;
  static const int pf = getenv("PLYOLO_PW_PF") ? atoi(getenv("PLYOLO_PW_PF")) : 0;   // measured: requests ahead of the MFMAs make the pointwise launches 2.5 % faster alone and the step 0.8 % slower
  p.pf = pf;
  static const int pipe = getenv("PLYOLO_PW_LOOP") ? atoi(getenv("PLYOLO_PW_LOOP")) : 0;
  p.pipe = pipe;
  p.nmt = (p.M + PW_BM - 1) / PW_BM;
  p.nnblk = (p.N + bn - 1) / bn;
}

}  // namespace

namespace plyolo {

// PLYOLO_PW=0 sends the pointwise layers through the 3x3 kernel's one-tap path again (A/B switch)
bool conv_pw_enabled() {
  static const bool on = !(getenv("PLYOLO_PW") && atoi(getenv("PLYOLO_PW")) == 0);
  return on;
}

int conv_pw_fwd(const plyolo_conv_desc* d, const void* x, const void* wp, const float* bias, void* y, double* stats,
                const float* ep_coef, int ep_act, const void* ep_res, int ep_res_ld, void* stream) {
  PwP p{};
  p.x = (const bf16_t*)x;
  p.w = (const bf16_t*)wp;
  p.y = y;
  p.bias = bias;
  p.stats = stats;
  p.ep_coef = ep_coef; p.ep_act = ep_act; p.ep_res = (const bf16_t*)ep_res; p.ep_res_ld = ep_res_ld;
  p.pre = d->x_coef; p.pre_ld = d->x_coef_ld; p.pre_act = d->x_act;
  p.M = d->N * d->H * d->W;
  p.K = d->Cin; p.N = d->Cout; p.x_ld = d->x_ld; p.y_ld = d->y_ld;
  p.nkb = (d->Cin + 15) / 16;
  p.nnb = (d->Cout + 31) / 32;
  const bool f32 = d->y_f32 != 0, pre = p.pre != nullptr;
  int BN, KC;
  pw_tiles(p, f32, &BN, &KC);
  const bool rag = pw_use_rag(p, BN, KC, !f32 && !pre);
  {
    char lab[64];
    if (rag) snprintf(lab, sizeof(lab), "conv_pw_fwd<BN128+%d,KC64>", p.N % 128 <= 32 ? 32 : 64);
    else snprintf(lab, sizeof(lab), "conv_pw_fwd<BN%d,KC%d>%s%s", BN, KC, f32 ? "f32out" : "", pre ? "+bnact" : "");
    annotate(lab, 2.0 * p.M * (double)d->Cout * d->Cin, (double)p.M * (d->Cout * (f32 ? 4.0 : 2.0) + d->Cin * 2.0));
  }
  if (rag) return submit(stream, [=](hipStream_t s) { return pw_launch_rag(p, s); });
  return submit(stream, [=](hipStream_t s) {
#ifdef PLYOLO_OPTIN
    if (pre) return f32 ? pw_launch<true, true>(p, BN, KC, s) : pw_launch<false, true>(p, BN, KC, s);
#endif
    return f32 ? pw_launch<true, false>(p, BN, KC, s) : pw_launch<false, false>(p, BN, KC, s);
  });
}

// dx[M, Cin] (+)= dy[M, Cout_p8] . Wd   (weights in the dgrad fragment pack: n-blocks over Cin, k-blocks over Cout)
// red (optional): plyolo_bn_red folded into the store loop; red_fits != NULL: only report whether a RED instance serves this launch
int conv_pw_dgrad(const plyolo_conv_desc* d, const void* dy, const void* wpd, void* dx, int accumulate, const plyolo_bn_red* red, void* stream,
                  int* red_fits) {
  PwP p{};
  p.x = (const bf16_t*)dy;
  p.w = (const bf16_t*)wpd;
  p.y = dx;
  p.M = d->N * d->H * d->W;
  p.K = (d->Cout + 7) & ~7;   // dy rows are read in 16-byte vectors
  p.N = d->Cin; p.x_ld = d->y_ld; p.y_ld = d->x_ld;
  p.nkb = (d->Cout + 15) / 16;
  p.nnb = (d->Cin + 31) / 32;
  p.accumulate = accumulate;
  int BN, KC;
  pw_tiles(p, false, &BN, &KC);
  if (red_fits) { *red_fits = (!p.pipe && pw_red_has(BN, KC)) ? 1 : 0; return 0; }
  const bool use_red = red && red->n > 0;
  if (use_red && (p.pipe || !pw_red_has(BN, KC))) { set_error("conv_pw_dgrad: no RED instance for these tiles (ask plyolo_conv2d_dgrad_red_fits)"); return -1; }
  if (use_red) p.red = *red;
  const bool rag = !use_red && pw_use_rag(p, BN, KC, true);      // (a folded reduction keeps the whole-block RED instance)
  {
    char lab[64];
    if (rag) snprintf(lab, sizeof(lab), "conv_pw_dgrad<BN128+%d,KC64>", p.N % 128 <= 32 ? 32 : 64);
    else snprintf(lab, sizeof(lab), "conv_pw_dgrad<BN%d,KC%d>%s", BN, KC, use_red ? "+bnred" : "");
    annotate(lab, 2.0 * p.M * (double)d->Cout * d->Cin, (double)p.M * (p.K + d->Cin * ((accumulate ? 2.0 : 1.0) + (use_red ? 1.0 : 0.0))) * 2.0);
  }
  if (rag) return submit(stream, [=](hipStream_t s) { return pw_launch_rag(p, s); });
  if (use_red) return submit(stream, [=](hipStream_t s) { return pw_launch_red(p, BN, KC, s); });
  return submit(stream, [=](hipStream_t s) { return pw_launch<false, false>(p, BN, KC, s); });
}

// ---- data gradient with the unit's BatchNorm + activation backward in the loader (plyolo_conv2d_dgrad_bn)
// 1 when plyolo_conv2d_dgrad_bn covers this unit: pointwise stride-1 bf16, a cheap activation, every channel vector whole, and
// at most two BN blocks per pixel tile (every block re-derives dz for the whole contraction length; beyond two the separate
// bn_act_bwd_dz pass is cheaper)
int conv_pw_dgrad_bn_fits(const plyolo_conv_desc* d, int act) {
  if (!conv_pw_enabled() || d->dtype != PLYOLO_BF16 || d->ksize != 1 || d->stride != 1) return 0;
  if (act < PLYOLO_ACT_NONE || act > PLYOLO_ACT_LRELU || d->Cout % 8 != 0 || d->Cout > 1024) return 0;
  // ... and the output gradient is at most PLYOLO_FUSE_BNBWD_MB (default 64) MB: measured per launch on YOLOX-s (26 MB: 18.5 us
  // against 23 for the two launches, 52 MB: 48 against 54, 105 MB: 102 against 95) and on the whole step of YOLOX-x at 1280x1280,
  // whose layers are all above that (fused everywhere: 104.0 ms, never: 102.1) -- the big streams are better off in the
  // lighter, higher-occupancy pair of kernels
  static const double max_mb = getenv("PLYOLO_FUSE_BNBWD_MB") ? atof(getenv("PLYOLO_FUSE_BNBWD_MB")) : 64.0;
  if ((double)d->N * d->H * d->W * d->Cout * 2.0 > max_mb * 1.0e6) return 0;
  const int bn = d->Cin > 64 ? 128 : (d->Cin > 32 ? 64 : 32);
  static const int maxblk = getenv("PLYOLO_FUSE_BNBWD_BLK") ? atoi(getenv("PLYOLO_FUSE_BNBWD_BLK")) : 2;
  return (d->Cin + bn - 1) / bn <= maxblk ? 1 : 0;
}

int conv_pw_dgrad_bn(const plyolo_conv_desc* d, const plyolo_bn_bwd_fuse* f, const void* wpd, void* dx, int accumulate, const plyolo_bn_red* red,
                     void* stream) {
  PwP p{};
  if (red && red->n > 0) p.red = *red;
  p.x = (const bf16_t*)f->dout;
  p.w = (const bf16_t*)wpd;
  p.y = dx;
  p.M = d->N * d->H * d->W;
  p.K = d->Cout;
  p.N = d->Cin; p.x_ld = f->dout_ld; p.y_ld = d->x_ld;
  p.nkb = (d->Cout + 15) / 16;
  p.nnb = (d->Cin + 31) / 32;
  p.accumulate = accumulate;
  p.bz = (const bf16_t*)f->z; p.bz_ld = f->z_ld;
  p.bx2 = (const bf16_t*)f->dout2; p.bx2_ld = f->dout2_ld; p.bsplit = f->dout2 ? f->dout_split : 0;
  p.bcoef = f->coef; p.bslots = f->bslots;
  p.bgamma = f->gamma; p.bdgamma = f->dgamma; p.bdbeta = f->dbeta;
  p.bpsplit = f->par_split; p.bgamma2 = f->gamma2; p.bdgamma2 = f->dgamma2; p.bdbeta2 = f->dbeta2;
  p.bact = f->act;
  p.bdz = (bf16_t*)f->dz; p.bdz_ld = f->dz_ld;
  int BN, KC;
  pw_tiles(p, false, &BN, &KC);
  if (KC > 64) KC = 64;
  static const int kc_env = getenv("PLYOLO_BNB_KC") ? atoi(getenv("PLYOLO_BNB_KC")) : 0;   // A/B: 32-channel chunks (4 waves per SIMD)
  if (kc_env == 32) KC = 32;
  p.pipe = 0;
  {
    char lab[64];
    snprintf(lab, sizeof(lab), "conv_pw_dgrad_bn<BN%d,KC%d>%s", BN, KC, p.red.n > 0 ? "+bnred" : "");
    annotate(lab, 2.0 * p.M * (double)d->Cout * d->Cin, (double)p.M * (3.0 * p.K + d->Cin * ((accumulate ? 2.0 : 1.0) + (p.red.n > 0 ? 1.0 : 0.0))) * 2.0);
  }
  return submit(stream, [=](hipStream_t s) { return pw_launch_bnb(p, BN, KC, s); });
}

}  // namespace plyolo
