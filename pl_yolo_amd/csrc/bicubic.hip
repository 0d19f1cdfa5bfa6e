// Bicubic x2 upsampling, forward and backward (e-yolox neck).
//
// Replaces nn.Upsample(scale_factor=2, mode="bicubic") (reference models/necks/pafpn_al.py:25,67,73) and its autograd.
// Semantics restated from ATen's upsample_bicubic2d (align_corners=False, scale 1/2 passed explicitly):
//   source coordinate  s = (o + 0.5) * 0.5 - 0.5   (NOT clamped for the cubic filter),  i = floor(s),  t = s - i
//   so output 2h reads around i = h-1 with t = 0.75 and output 2h+1 around i = h with t = 0.25;
//   taps i-1 .. i+2, each index clamped to [0, size-1]; cubic convolution coefficients with A = -0.75:
//   c0 = ((A(t+1) - 5A)(t+1) + 8A)(t+1) - 4A,  c1 = ((A+2)t - (A+3))t^2 + 1,  c2 = c1(1-t),  c3 = c0(2-t)... (c3 uses 2-t)
//   rows are interpolated first, then the four row results along y.
// The backward is the exact transpose, written as a GATHER (an input pixel sums the <= 10 x 10 output pixels whose
// clamped taps reach it): deterministic, no atomics.  HBM/L2 streams, 16-byte channel vectors, fp32 arithmetic.
#include "common.h"

namespace {

DEVINL float cc1(float x) { const float A = -0.75f; return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
DEVINL float cc2(float x) { const float A = -0.75f; return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }
DEVINL void cubic_coef(float t, float* c) {
  c[0] = cc2(t + 1.f);
  c[1] = cc1(t);
  c[2] = cc1(1.f - t);
  c[3] = cc2(2.f - t);
}
DEVINL int clampi(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }

template <typename T>
__global__ void bicubic2x_fwd_kernel(int N, int H, int W, int C, const T* __restrict__ in, int i_ld, T* __restrict__ out, int o_ld) {
  constexpr int V = Vec<T>::N;
  const int cvn = C / V;
  const size_t total = (size_t)N * (2 * H) * (2 * W) * cvn;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % cvn) * V;
    size_t t = idx / cvn;
    const int ox = (int)(t % (2 * W));
    t /= (2 * W);
    const int oy = (int)(t % (2 * H));
    const int n = (int)(t / (2 * H));
    const int iy = (oy >> 1) - 1 + (oy & 1), ix = (ox >> 1) - 1 + (ox & 1);
    float cy[4], cx[4];
    cubic_coef((oy & 1) ? 0.25f : 0.75f, cy);
    cubic_coef((ox & 1) ? 0.25f : 0.75f, cx);
    float acc[V];
#pragma unroll
    for (int i = 0; i < V; ++i) acc[i] = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int yy = clampi(iy - 1 + a, H - 1);
      float row[V];
#pragma unroll
      for (int i = 0; i < V; ++i) row[i] = 0.f;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int xx = clampi(ix - 1 + b, W - 1);
        float f[V];
        Vec<T>::load(in + ((size_t)(n * H + yy) * W + xx) * i_ld + c, f);
#pragma unroll
        for (int i = 0; i < V; ++i) row[i] = fmaf(f[i], cx[b], row[i]);
      }
#pragma unroll
      for (int i = 0; i < V; ++i) acc[i] = fmaf(row[i], cy[a], acc[i]);
    }
    Vec<T>::store(out + ((size_t)(n * 2 * H + oy) * (2 * W) + ox) * o_ld + c, acc);
  }
}

// weight with which output index o (along one axis of input size S) reads input index p: sum of the coefficients of its
// four taps whose clamped index equals p
DEVINL float tap_weight(int o, int p, int S) {
  const int i0 = (o >> 1) - 1 + (o & 1);
  float cf[4];
  cubic_coef((o & 1) ? 0.25f : 0.75f, cf);
  float w = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a)
    if (clampi(i0 - 1 + a, S - 1) == p) w += cf[a];
  return w;
}

template <typename T>
__global__ void bicubic2x_bwd_kernel(int N, int H, int W, int C, const T* __restrict__ dout, int d_ld, T* __restrict__ din, int i_ld,
                                     int accumulate) {
  constexpr int V = Vec<T>::N;
  const int cvn = C / V;
  const size_t total = (size_t)N * H * W * cvn;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % cvn) * V;
    size_t t = idx / cvn;
    const int x = (int)(t % W);
    t /= W;
    const int y = (int)(t % H);
    const int n = (int)(t / H);
    float wy[10], wx[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) {
      const int oy = 2 * y - 4 + k, ox = 2 * x - 4 + k;
      wy[k] = (oy >= 0 && oy < 2 * H) ? tap_weight(oy, y, H) : 0.f;
      wx[k] = (ox >= 0 && ox < 2 * W) ? tap_weight(ox, x, W) : 0.f;
    }
    float s[V];
#pragma unroll
    for (int i = 0; i < V; ++i) s[i] = 0.f;
    for (int a = 0; a < 10; ++a) {
      if (wy[a] == 0.f) continue;
      const int oy = 2 * y - 4 + a;
      float row[V];
#pragma unroll
      for (int i = 0; i < V; ++i) row[i] = 0.f;
      for (int b = 0; b < 10; ++b) {
        if (wx[b] == 0.f) continue;
        const int ox = 2 * x - 4 + b;
        float f[V];
        Vec<T>::load(dout + ((size_t)(n * 2 * H + oy) * (2 * W) + ox) * d_ld + c, f);
#pragma unroll
        for (int i = 0; i < V; ++i) row[i] = fmaf(f[i], wx[b], row[i]);
      }
#pragma unroll
      for (int i = 0; i < V; ++i) s[i] = fmaf(row[i], wy[a], s[i]);
    }
    T* dst = din + ((size_t)(n * H + y) * W + x) * i_ld + c;
    if (accumulate) {
      float o[V];
      Vec<T>::load(dst, o);
#pragma unroll
      for (int i = 0; i < V; ++i) s[i] += o[i];
    }
    Vec<T>::store(dst, s);
  }
}

inline int grid_for(size_t work) {
  size_t g = (work + 255) / 256;
  return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

}  // namespace

using plyolo::submit;

#define DISPATCH_T(dtype, ...)                       \
  if ((dtype) == PLYOLO_BF16) { typedef bf16_t T; __VA_ARGS__ } \
  else { typedef float T; __VA_ARGS__ }

extern "C" {

int plyolo_bicubic2x_fwd(int dtype, int N, int H, int W, int C, const void* in, int i_ld, void* out, int o_ld, void* stream) {
  const int V = dtype == PLYOLO_BF16 ? 8 : 4;
  PLY_CHECK_ARG(in && out && N > 0 && H > 0 && W > 0 && C % V == 0 && i_ld % V == 0 && o_ld % V == 0, "bicubic2x_fwd: C/ld must be multiples of %d", V);
  plyolo::annotate("bicubic2x_fwd", 0.0, (double)N * H * W * C * (dtype == PLYOLO_BF16 ? 2.0 : 4.0) * 5.0);
  const size_t work = (size_t)N * 4 * H * W * (C / V);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    DISPATCH_T(dtype, hipLaunchKernelGGL(bicubic2x_fwd_kernel<T>, dim3(grid_for(work)), dim3(256), 0, s, N, H, W, C, (const T*)in, i_ld, (T*)out, o_ld);)
    return hipGetLastError();
  });
}

int plyolo_bicubic2x_bwd(int dtype, int N, int H, int W, int C, const void* dout, int d_ld, void* din, int i_ld, int accumulate,
                         void* stream) {
  const int V = dtype == PLYOLO_BF16 ? 8 : 4;
  PLY_CHECK_ARG(dout && din && N > 0 && H > 0 && W > 0 && C % V == 0 && i_ld % V == 0 && d_ld % V == 0, "bicubic2x_bwd: C/ld must be multiples of %d", V);
  plyolo::annotate("bicubic2x_bwd", 0.0, (double)N * H * W * C * (dtype == PLYOLO_BF16 ? 2.0 : 4.0) * 5.0);
  const size_t work = (size_t)N * H * W * (C / V);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    DISPATCH_T(dtype, hipLaunchKernelGGL(bicubic2x_bwd_kernel<T>, dim3(grid_for(work)), dim3(256), 0, s, N, H, W, C, (const T*)dout, d_ld, (T*)din,
                                         i_ld, accumulate);)
    return hipGetLastError();
  });
}

}  // extern "C"
