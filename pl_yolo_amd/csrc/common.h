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

// The three cheap activations, for code that is inlined into the MFMA kernels (fused inference epilogue, lazy-input loaders).
// hswish / gelu stay out of those: the epilogue applies the activation to 64-128 accumulators per thread, fully unrolled, and
// the erff expansion of GELU in every one of them made the conv kernels 6-14 % slower even for launches that never take the
// branch (instruction footprint; measured against the round-1 build on one box).  Units with those activations use the
// BatchNorm-apply stream (bn_act_fwd) instead.
DEVINL float act_fwd_core(float u, int act) {
  switch (act) {
    case PLYOLO_ACT_SILU: return u * __builtin_amdgcn_rcpf(1.0f + __expf(-u));
    case PLYOLO_ACT_RELU: return u > 0.f ? u : 0.f;
    case PLYOLO_ACT_LRELU: return u > 0.f ? u : 0.1f * u;
    default: return u;
  }
}
// act(x * sc + sh) on the 8 bf16 channels of one staged 16-byte vector (lazy-input loaders).  The activation is dispatched ONCE
// per vector: a switch on a run-time activation inside the unrolled element loop compiles to a branch per element, which serialises
// the exp / rcp chains of the elements (round 4: that, not the arithmetic itself, was most of what the lazy loaders cost)
DEVINL u32x4 bn_act_vec8(u32x4 t, const float* sc, const float* sh, const int act) {
  if (act == PLYOLO_ACT_SILU) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float lo = act_fwd_core(fmaf(__uint_as_float(t[i] << 16), sc[2 * i], sh[2 * i]), PLYOLO_ACT_SILU);
      const float hi = act_fwd_core(fmaf(__uint_as_float(t[i] & 0xffff0000u), sc[2 * i + 1], sh[2 * i + 1]), PLYOLO_ACT_SILU);
      t[i] = pack2bf(lo, hi);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float lo = act_fwd_core(fmaf(__uint_as_float(t[i] << 16), sc[2 * i], sh[2 * i]), act);
      const float hi = act_fwd_core(fmaf(__uint_as_float(t[i] & 0xffff0000u), sc[2 * i + 1], sh[2 * i + 1]), act);
      t[i] = pack2bf(lo, hi);
    }
  }
  return t;
}
DEVINL float act_fwd(float u, int act) {
  switch (act) {
    case PLYOLO_ACT_SILU: return u * __builtin_amdgcn_rcpf(1.0f + __expf(-u));  // hardware exp2 / rcp (1 ulp): bf16 storage path
    case PLYOLO_ACT_RELU: return u > 0.f ? u : 0.f;
    case PLYOLO_ACT_LRELU: return u > 0.f ? u : 0.1f * u;
    case PLYOLO_ACT_HSWISH: return u * fminf(fmaxf(u + 3.0f, 0.f), 6.0f) * (1.0f / 6.0f);
    case PLYOLO_ACT_GELU: return 0.5f * u * (1.0f + erff(u * 0.70710678118654752f));
    default: return u;
  }
}
DEVINL float act_fwd_precise(float u, int act) {
  switch (act) {
    case PLYOLO_ACT_SILU: return u / (1.0f + expf(-u));
    case PLYOLO_ACT_RELU: return u > 0.f ? u : 0.f;
    case PLYOLO_ACT_LRELU: return u > 0.f ? u : 0.1f * u;
    case PLYOLO_ACT_HSWISH: return u * fminf(fmaxf(u + 3.0f, 0.f), 6.0f) / 6.0f;   // x * relu6(x + 3) / 6, activation.py:24
    case PLYOLO_ACT_GELU: return 0.5f * u * (1.0f + erff(u * 0.70710678118654752f));
    default: return u;
  }
}
// PRECISE = libm exp + IEEE divide (fp32 parity mode); otherwise the hardware exp2 / rcp instructions (1 ulp),
// whose error vanishes under the bf16 rounding of the stored gradient
template <bool PRECISE = true>
DEVINL float act_grad(float u, int act) {
  switch (act) {
    case PLYOLO_ACT_SILU: {
      const float s = PRECISE ? 1.0f / (1.0f + expf(-u)) : __builtin_amdgcn_rcpf(1.0f + __expf(-u));
      return s * (1.0f + u * (1.0f - s));
    }
    case PLYOLO_ACT_RELU: return u > 0.f ? 1.f : 0.f;
    case PLYOLO_ACT_LRELU: return u > 0.f ? 1.f : 0.1f;
    // d/du [u * relu6(u+3) / 6] = (relu6(u+3) + u * [0 < u+3 < 6]) / 6   (autograd's subgradient 0 at both kinks)
    case PLYOLO_ACT_HSWISH: return (fminf(fmaxf(u + 3.0f, 0.f), 6.0f) + ((u > -3.0f && u < 3.0f) ? u : 0.f)) * (1.0f / 6.0f);
    // d/du [u * Phi(u)] = Phi(u) + u * phi(u)
    case PLYOLO_ACT_GELU: return 0.5f * (1.0f + erff(u * 0.70710678118654752f)) + u * 0.3989422804014327f * (PRECISE ? expf(-0.5f * u * u) : __expf(-0.5f * u * u));
    default: return 1.f;
  }
}

// 16-byte vector access to activation rows (8 x bf16 or 4 x fp32), fp32 math in registers
template <typename T> struct Vec;
template <> struct Vec<bf16_t> {
  static constexpr int N = 8;
  typedef u32x4 raw_t;     // a requested vector that nothing has touched yet (the wait for it sits at cvt, not at the request)
  static DEVINL raw_t load_raw(const bf16_t* p) { return *(const u32x4*)p; }
  static DEVINL void cvt(const raw_t& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = __uint_as_float(v[i] << 16);
      f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
    }
  }
  static DEVINL void load(const bf16_t* p, float* f) { cvt(load_raw(p), f); }
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
  typedef f32x4 raw_t;
  static DEVINL raw_t load_raw(const float* p) { return *(const f32x4*)p; }
  static DEVINL void cvt(const raw_t& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = v[i];
  }
  static DEVINL void load(const float* p, float* f) { cvt(load_raw(p), f); }
  static DEVINL void store(float* p, const float* f) {
    f32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = f[i];
    *(f32x4*)p = v;
  }
  static constexpr bool precise = true;
};

template <bool PRECISE> DEVINL float actf(float u, int act) { return PRECISE ? act_fwd_precise(u, act) : act_fwd(u, act); }


namespace plyolo {

void set_error(const char* fmt, ...);

struct PlanOp {
  std::function<hipError_t(hipStream_t)> fn;  // empty for the ordering markers below
  std::string label;   // kernel family / template instance, for the plan profiler
  double flops = 0.0;  // algorithmic FLOPs of this launch (0 = not a contraction)
  double bytes = 0.0;  // algorithmic HBM bytes of this launch
  int lane = 0;        // launches of one lane are ordered; lanes run concurrently inside a hipGraph
  int kind = 0;        // 0 launch, 1 record event `ev` on `lane`, 2 make `lane` wait for event `ev`, 3 host hook `ev` on `lane`
  int ev = -1;
};
struct Plan {
  std::vector<PlanOp> ops;
  int cur_lane = 0;    // lane of the launches being recorded
  int nlanes = 1, nevents = 0;
  std::vector<hipStream_t> side;   // streams of lanes 1.. (created on first replay) [+ the plan's own main stream, PLYOLO_OWN_MAIN]
  std::vector<hipEvent_t> io_events;  // fork from / join into the caller's stream (PLYOLO_OWN_MAIN)
  std::vector<hipEvent_t> events;  // fork/join + recorded events
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  int (*hook)(int id, void* stream, void* user) = nullptr;   // plyolo_plan_set_hook
  void* hook_user = nullptr;
  int nhooks = 0;
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
    op.lane = p->cur_lane;
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

// hipFuncAttributeMaxDynamicSharedMemorySize, set ONCE per kernel and size: the attribute call costs host time on
// every launch of the eager replay (~1300 convolution launches per step) and never changes after the first.
hipError_t ensure_dynamic_lds(const void* kernel, size_t bytes);

// Byte fill as an ordinary kernel launch.  (hipMemsetAsync nodes inside a captured hipGraph were
// observed to misbehave on re-launch for small sizes on ROCm 7.2; a kernel node is always safe.)
__global__ void fill_bytes_kernel(unsigned char* p, unsigned value, size_t bytes);
hipError_t fill_async(void* p, int value, size_t bytes, hipStream_t s);

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
