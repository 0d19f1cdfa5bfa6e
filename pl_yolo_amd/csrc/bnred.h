// BatchNorm-backward reduction folded into a data gradient's store loop (plyolo_bn_red, include/plyolo.h): device side.
//
// A data-gradient epilogue hands every thread whole 16-byte channel vectors of FINAL dx rows; a thread always owns the same 8
// channels (the row loop strides over pixels only).  BnRedThread keeps that thread's coefficients and its 2 x 8 partial sums in
// registers; bnred_flush() folds the partials of the workgroup in a fixed order through LDS (run-to-run identical) and adds one
// fp64 value per channel and sum to the unit's stat slots (agent-scope atomics; fp64 makes their order irrelevant to the fp32
// result -- the scheme of the forward statistics, conv_mfma_body.h).
#pragma once
#include "common.h"

struct BnRedThread {
  const bf16_t* z;   // segment's z, pre-offset to this thread's 8 channels (NULL: no reduction for these channels)
  int z_ld, act;
  double* slots;     // segment's slots, pre-offset to this thread's 8 channels
  int slot_ld;
  float sc[8], sh[8], mu[8], is[8], s1[8], s2[8];
};

// co = first of the thread's 8 channels inside dx
DEVINL void bnred_init(BnRedThread& t, const plyolo_bn_red& r, const int co) {
  t.z = nullptr; t.z_ld = 0; t.act = 0; t.slots = nullptr; t.slot_ld = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) { t.sc[i] = t.sh[i] = t.mu[i] = t.is[i] = 0.f; t.s1[i] = t.s2[i] = 0.f; }
#pragma unroll
  for (int k = 0; k < PLYOLO_BN_RED_SEGS; ++k) {
    if (k < r.n && co >= r.seg[k].c0 && co < r.seg[k].c1) {
      const plyolo_bn_red_seg& g = r.seg[k];
      const int c = co - g.c0;
      t.z = (const bf16_t*)g.z + c; t.z_ld = g.z_ld; t.act = g.act;
      t.slots = g.bslots + c; t.slot_ld = g.slot_ld;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        t.sc[i] = g.coef[c + i]; t.sh[i] = g.coef[g.coef_ld + c + i];
        t.mu[i] = g.coef[2 * g.coef_ld + c + i]; t.is[i] = g.coef[3 * g.coef_ld + c + i];
      }
    }
  }
}

DEVINL u32x4 bnred_load(const BnRedThread& t, const size_t pix) { return *(const u32x4*)(t.z + pix * t.z_ld); }

// derivative of the activations a folded reduction carries (== act_grad<false> for them).  hswish / gelu stay out: this runs in the
// unrolled store loops of the convolution kernels, 8 rows x 8 channels per thread, and the erff expansion in every element of a
// run-time switch made the 128-channel pointwise RED instance 15 k instructions long (61 KB: the instruction cache of a CU pair)
// although no shipped config ever takes that branch; such units keep their plyolo_bn_act_bwd_reduce launch (api.hip: check_red)
DEVINL float bnred_act_grad(const float u, const int act) {
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

// dx = the stored (bf16-rounded) gradient vector, zz = the unit's z at the same pixel and channels
template <int ACT>
DEVINL void bnred_add(BnRedThread& t, const u32x4 dx, const u32x4 zz) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float zl = __uint_as_float(zz[i] << 16), zh = __uint_as_float(zz[i] & 0xffff0000u);
    const float dl = __uint_as_float(dx[i] << 16), dh = __uint_as_float(dx[i] & 0xffff0000u);
    const float dul = dl * bnred_act_grad(fmaf(zl, t.sc[2 * i], t.sh[2 * i]), ACT >= 0 ? ACT : t.act);
    const float duh = dh * bnred_act_grad(fmaf(zh, t.sc[2 * i + 1], t.sh[2 * i + 1]), ACT >= 0 ? ACT : t.act);
    t.s1[2 * i] += dul;
    t.s1[2 * i + 1] += duh;
    t.s2[2 * i] += dul * ((zl - t.mu[2 * i]) * t.is[2 * i]);
    t.s2[2 * i + 1] += duh * ((zh - t.mu[2 * i + 1]) * t.is[2 * i + 1]);
  }
}

// run-time activation, dispatched once per VECTOR: SiLU (every shipped config) takes the straight-line instance -- a switch on the
// activation inside the unrolled element loop compiles to a branch per element
DEVINL void bnred_add_any(BnRedThread& t, const u32x4 dx, const u32x4 zz) {
  if (t.act == PLYOLO_ACT_SILU) bnred_add<PLYOLO_ACT_SILU>(t, dx, zz);
  else bnred_add<-1>(t, dx, zz);
}

// Workgroup fold + slot adds.  `lds` = NT / VPR x 2 x (VPR * 8) floats of scratch (may alias the store loop's staging: the caller
// has synchronised behind it); thread layout of the partials: channel vector tid % VPR, row group tid / VPR.  The 2 * VPR * 8 column
// sums are dealt over the threads (one fixed-order sum of NT / VPR partials and one atomic each); cbase = first dx channel of the
// workgroup's VPR vectors; `slot` picks one of the stat slots.
template <int NT, int VPR>
DEVINL void bnred_flush(const BnRedThread& t, const plyolo_bn_red& r, const int cbase, float* lds, const int tid, const int slot) {
  constexpr int G = NT / VPR, NC = VPR * 8;
  const int v = tid % VPR, g = tid / VPR;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    lds[(g * 2 + 0) * NC + v * 8 + i] = t.s1[i];
    lds[(g * 2 + 1) * NC + v * 8 + i] = t.s2[i];
  }
  __syncthreads();
  for (int j = tid; j < 2 * NC; j += NT) {
    const int which = j / NC, ch = j - which * NC, co = cbase + ch;
    double* dst = nullptr;
#pragma unroll
    for (int k = 0; k < PLYOLO_BN_RED_SEGS; ++k)
      if (k < r.n && co >= r.seg[k].c0 && co < r.seg[k].c1)
        dst = r.seg[k].bslots + (size_t)slot * 2 * r.seg[k].slot_ld + (size_t)which * r.seg[k].slot_ld + (co - r.seg[k].c0);
    if (dst != nullptr) {
      float a = 0.f;
#pragma unroll 4
      for (int k = 0; k < G; ++k) a += lds[(k * 2 + which) * NC + ch];
      __hip_atomic_fetch_add(dst, (double)a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
