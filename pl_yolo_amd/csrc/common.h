// Shared host/device helpers for libplyolo_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <functional>
#include <string>
#include <vector>

#include "../../include/plyolo.h"

typedef unsigned short bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define DEVINL __device__ __forceinline__

DEVINL float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
DEVINL bf16_t f2bf(float f) {
  // round-to-nearest-even; NaN stays NaN (plain cast -> v_cvt_pk_bf16_f32 on gfx950)
  __bf16 b = (__bf16)f;
  return *(bf16_t*)&b;
}
DEVINL unsigned pack2bf(float lo, float hi) { return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16); }

template <typename T> struct ActT;
template <> struct ActT<bf16_t> {
  static DEVINL float ld(const bf16_t* p) { return bf2f(*p); }
  static DEVINL void st(bf16_t* p, float v) { *p = f2bf(v); }
};
template <> struct ActT<float> {
  static DEVINL float ld(const float* p) { return *p; }
  static DEVINL void st(float* p, float v) { *p = v; }
};

DEVINL float act_fwd(float u, int act) {
  switch (act) {
    case PLYOLO_ACT_SILU: return u / (1.0f + __expf(-u));
    case PLYOLO_ACT_RELU: return u > 0.f ? u : 0.f;
    case PLYOLO_ACT_LRELU: return u > 0.f ? u : 0.1f * u;
    default: return u;
  }
}
DEVINL float act_fwd_precise(float u, int act) {
  switch (act) {
    case PLYOLO_ACT_SILU: return u / (1.0f + expf(-u));
    case PLYOLO_ACT_RELU: return u > 0.f ? u : 0.f;
    case PLYOLO_ACT_LRELU: return u > 0.f ? u : 0.1f * u;
    default: return u;
  }
}
DEVINL float act_grad(float u, int act) {
  switch (act) {
    case PLYOLO_ACT_SILU: {
      float s = 1.0f / (1.0f + expf(-u));
      return s * (1.0f + u * (1.0f - s));
    }
    case PLYOLO_ACT_RELU: return u > 0.f ? 1.f : 0.f;
    case PLYOLO_ACT_LRELU: return u > 0.f ? 1.f : 0.1f;
    default: return 1.f;
  }
}

// 16-byte vector access to activation rows (8 x bf16 or 4 x fp32), fp32 math in registers
template <typename T> struct Vec;
template <> struct Vec<bf16_t> {
  static constexpr int N = 8;
  static DEVINL void load(const bf16_t* p, float* f) {
    const u32x4 v = *(const u32x4*)p;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = __uint_as_float(v[i] << 16);
      f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
    }
  }
  static DEVINL void store(bf16_t* p, const float* f) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = pack2bf(f[2 * i], f[2 * i + 1]);
    *(u32x4*)p = v;
  }
  static constexpr bool precise = false;
};
template <> struct Vec<float> {
  static constexpr int N = 4;
  static DEVINL void load(const float* p, float* f) {
    const f32x4 v = *(const f32x4*)p;
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = v[i];
  }
  static DEVINL void store(float* p, const float* f) {
    f32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = f[i];
    *(f32x4*)p = v;
  }
  static constexpr bool precise = true;
};

template <bool PRECISE> DEVINL float actf(float u, int act) { return PRECISE ? act_fwd_precise(u, act) : act_fwd(u, act); }


// ---- deterministic in-kernel reduction of per-workgroup partial rows -------------------
// Every workgroup publishes ONE partial row (two quantities x C channels) of
// rows[2][nrows][C] with hier_store().  Rows are grouped by 32: the last workgroup of a group
// to arrive sums the group's rows into gpart[2][ngroups][C]; the last group to finish sums
// the group rows and calls finish(c, sum0, sum1) for its columns.  Fixed summation order =>
// bit-reproducible.  Hand-off protocol (cdna_hip_programming.md Guideline 16, sc1 form): every
// handed-off value is stored write-through (`sc1`, relaxed agent-scope atomic store), each
// storing wave drains with s_waitcnt vmcnt(0), a workgroup barrier, then ONE lane takes the
// ticket; readers use sc1 loads.  No release fence: a `buffer_wbl2` would write back the
// whole XCD L2 (the conv output just stored) once per workgroup -- measured 2x on the step.
// Counters are zeroed once by the caller and reset themselves.  `cb` = column-block id
// (counter namespace), columns [c0, c0+ncols).
constexpr int HIER_GROUP = 32;
struct HierRed {
  float* rows;
  float* gpart;
  unsigned* gcnt;  // [ncb][ngroups]
  unsigned* fcnt;  // [ncb]
  int nrows, C;
};
DEVINL void hier_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEVINL float hier_load(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <typename F>
DEVINL void hier_finish(const HierRed& h, int row, int cb, int c0, int ncols, int* s_flag, F&& finish) {
  const int tid = threadIdx.x;
  const int ngroups = (h.nrows + HIER_GROUP - 1) / HIER_GROUP;
  const int group = row / HIER_GROUP;
  const int g0 = group * HIER_GROUP;
  const int gn = (h.nrows - g0 < HIER_GROUP) ? h.nrows - g0 : HIER_GROUP;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's row stores have left
  __syncthreads();
  if (tid == 0) {
    const unsigned t = __hip_atomic_fetch_add(h.gcnt + (size_t)cb * ngroups + group, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *s_flag = (t == (unsigned)gn - 1u) ? 1 : 0;
  }
  __syncthreads();
  if (!*s_flag) return;
  for (int c = tid; c < ncols; c += blockDim.x) {
    const int col = c0 + c;
    if (col >= h.C) continue;
    float s = 0.f, ss = 0.f;
    for (int k = 0; k < gn; ++k) {
      s += hier_load(h.rows + (size_t)(g0 + k) * h.C + col);
      ss += hier_load(h.rows + ((size_t)h.nrows + g0 + k) * h.C + col);
    }
    hier_store(h.gpart + (size_t)group * h.C + col, s);
    hier_store(h.gpart + ((size_t)ngroups + group) * h.C + col, ss);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    __hip_atomic_store(h.gcnt + (size_t)cb * ngroups + group, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned t = __hip_atomic_fetch_add(h.fcnt + cb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *s_flag = (t == (unsigned)ngroups - 1u) ? 1 : 0;
  }
  __syncthreads();
  if (!*s_flag) return;
  for (int c = tid; c < ncols; c += blockDim.x) {
    const int col = c0 + c;
    if (col >= h.C) continue;
    double s = 0.0, ss = 0.0;
    for (int k = 0; k < ngroups; ++k) {
      s += (double)hier_load(h.gpart + (size_t)k * h.C + col);
      ss += (double)hier_load(h.gpart + ((size_t)ngroups + k) * h.C + col);
    }
    finish(col, s, ss);
  }
  if (tid == 0) __hip_atomic_store(h.fcnt + cb, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------- host side
namespace plyolo {

void set_error(const char* fmt, ...);

struct PlanOp {
  std::function<hipError_t(hipStream_t)> fn;
  std::string label;   // kernel family / template instance, for the plan profiler
  double flops = 0.0;  // algorithmic FLOPs of this launch (0 = not a contraction)
  double bytes = 0.0;  // algorithmic HBM bytes of this launch
};
struct Plan {
  std::vector<PlanOp> ops;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
};
Plan* recording_plan();
// Describe the NEXT submitted launch (label + algorithmic work); consumed by submit().
void annotate(const char* label, double flops, double bytes);
void take_annotation(PlanOp* op);

// Either run `fn` now on `stream`, or append it to the plan being recorded.
template <typename F> int submit(void* stream, F&& fn) {
  Plan* p = recording_plan();
  if (p) {
    PlanOp op;
    op.fn = std::forward<F>(fn);
    take_annotation(&op);
    p->ops.emplace_back(std::move(op));
    return 0;
  }
  take_annotation(nullptr);
  hipError_t e = fn((hipStream_t)stream);
  if (e != hipSuccess) {
    set_error("HIP launch failed: %s", hipGetErrorString(e));
    return -2;
  }
  return 0;
}

inline hipError_t launch_status() { return hipGetLastError(); }

}  // namespace plyolo

#define PLY_CHECK_ARG(cond, ...)        \
  do {                                  \
    if (!(cond)) {                      \
      plyolo::set_error(__VA_ARGS__);   \
      return -1;                        \
    }                                   \
  } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t cdivz(size_t a, size_t b) { return (a + b - 1) / b; }
