// Depthwise 3x3 convolution (groups == channels, stride 1, pad 1) for gfx950: forward, data gradient, weight gradient.
//
// Replaces nn.Conv2d(C, C, 3, 1, 1, groups=C, bias=False) inside BaseConv as the e-yolox family uses it
// (reference models/backbones/ecmnet.py:157,160 and models/necks/pafpn_al.py:162,165: Bottleneck.conv0 / conv3) and what
// ATen's convolution_backward computes for it.
//
// One multiply-add per loaded element and tap: the op is a pure HBM stream (9 neighbour reads served by L1/L2, one
// write), no contraction over channels -- so no MFMA (north_star: "MFMA only for the tiles that are genuinely dense
// GEMMs").  Layout shared with the BatchNorm streams (bn.hip): a thread owns ONE 16-byte channel vector column for the
// whole launch -- its 9 x 8 weights live in registers -- and walks down the pixel rows; consecutive lanes touch
// consecutive vectors of a pixel (full 128-byte lines).
//   dwconv3x3 (fwd / dgrad)  y[p, c] (+)= sum_t x[p + t, c] * w[c, t]   (dgrad: the taps mirrored, x = dz, y = dx);
//                            forward also adds the BatchNorm sum / sum-of-squares partials to the fp64 stat slots
//   dwconv3x3_wgrad          dw[c, t] = sum_p dz[p, c] * x[p + t, c]: per-workgroup partials in a fixed layout, folded in a
//                            fixed order by a second tiny launch (deterministic, no atomics)
// bf16 mode rounds the weights to bf16 like the MFMA packs do; accumulation is fp32, statistics come from the fp32 sums.
#include "common.h"

namespace {

constexpr int NSLOT = PLYOLO_STAT_SLOTS;

DEVINL void slot_add(double* p, double v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct DwMap {   // column-fixed streaming layout: cols = min(C/V, 256) vector columns, rpb = 256/cols pixel rows per pass
  int cols, rpb, tcol, trow;
  DEVINL DwMap(int cvn) {
    cols = cvn < 256 ? cvn : 256;
    rpb = 256 / cols;
    tcol = threadIdx.x % cols;
    trow = threadIdx.x / cols;
  }
};

template <typename T> DEVINL float wround(float v) {
  if (sizeof(T) == 2) return bf2f(f2bf(v));
  return v;
}

template <typename T>
__global__ __launch_bounds__(256) void dwconv3x3_kernel(int N, int H, int W, int C, const T* __restrict__ x, int x_ld,
                                                        const float* __restrict__ w, int flip, T* __restrict__ y, int y_ld,
                                                        int accumulate, double* stats) {
  constexpr int V = Vec<T>::N;
  __shared__ float red[256 * 2 * V];
  const int cvn = C / V;
  const DwMap cm(cvn);
  const int M = N * H * W, step = gridDim.x * cm.rpb;
  double* slot = stats ? stats + (size_t)(blockIdx.x % NSLOT) * 2 * C : nullptr;
  for (int cv0 = 0; cv0 < cvn; cv0 += cm.cols) {
    const int cv = cv0 + cm.tcol;
    float s1[V], s2[V];
#pragma unroll
    for (int i = 0; i < V; ++i) s1[i] = s2[i] = 0.f;
    if (cm.trow < cm.rpb && cv < cvn) {
      const int c = cv * V;
      float wt[9][V];
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < V; ++i) wt[t][i] = wround<T>(w[(size_t)(c + i) * 9 + (flip ? 8 - t : t)]);
      for (int m = blockIdx.x * cm.rpb + cm.trow; m < M; m += step) {
        const int px = m % W, t2 = m / W, py = t2 % H;
        float acc[V];
#pragma unroll
        for (int i = 0; i < V; ++i) acc[i] = 0.f;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
          for (int dx = -1; dx <= 1; ++dx) {
            if (py + dy < 0 || py + dy >= H || px + dx < 0 || px + dx >= W) continue;
            float f[V];
            Vec<T>::load(x + (size_t)(m + dy * W + dx) * x_ld + c, f);
#pragma unroll
            for (int i = 0; i < V; ++i) acc[i] = fmaf(f[i], wt[(dy + 1) * 3 + dx + 1][i], acc[i]);
          }
        if (stats) {
#pragma unroll
          for (int i = 0; i < V; ++i) { s1[i] += acc[i]; s2[i] = fmaf(acc[i], acc[i], s2[i]); }
        }
        T* dst = y + (size_t)m * y_ld + c;
        if (accumulate) {
          float o[V];
          Vec<T>::load(dst, o);
#pragma unroll
          for (int i = 0; i < V; ++i) acc[i] += o[i];
        }
        Vec<T>::store(dst, acc);
      }
    }
    if (stats) {   // one fp64 add per workgroup and channel (the order cannot change the fp32 result)
      __syncthreads();
#pragma unroll
      for (int i = 0; i < V; ++i) { red[(threadIdx.x * 2 + 0) * V + i] = s1[i]; red[(threadIdx.x * 2 + 1) * V + i] = s2[i]; }
      __syncthreads();
      if (cm.trow == 0 && cv < cvn) {
        for (int k = 1; k < cm.rpb; ++k)
#pragma unroll
          for (int i = 0; i < V; ++i) {
            s1[i] += red[((k * cm.cols + cm.tcol) * 2 + 0) * V + i];
            s2[i] += red[((k * cm.cols + cm.tcol) * 2 + 1) * V + i];
          }
#pragma unroll
        for (int i = 0; i < V; ++i) {
          slot_add(slot + cv * V + i, (double)s1[i]);
          slot_add(slot + C + cv * V + i, (double)s2[i]);
        }
      }
      __syncthreads();
    }
  }
}

// partial[block][c][t] = sum over the block's pixel rows of dz[p, c] * x[p + t, c]
template <typename T>
__global__ __launch_bounds__(256) void dwconv3x3_wgrad_kernel(int N, int H, int W, int C, const T* __restrict__ x, int x_ld,
                                                              const T* __restrict__ dz, int dz_ld, float* __restrict__ partial) {
  constexpr int V = Vec<T>::N;
  extern __shared__ __align__(16) float wred[];   // [256][V]
  const int cvn = C / V;
  const DwMap cm(cvn);
  const int M = N * H * W, step = gridDim.x * cm.rpb;
  float* mine = partial + (size_t)blockIdx.x * C * 9;
  for (int cv0 = 0; cv0 < cvn; cv0 += cm.cols) {
    const int cv = cv0 + cm.tcol, c = cv * V;
    float acc[9][V];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int i = 0; i < V; ++i) acc[t][i] = 0.f;
    if (cm.trow < cm.rpb && cv < cvn)
      for (int m = blockIdx.x * cm.rpb + cm.trow; m < M; m += step) {
        const int px = m % W, t2 = m / W, py = t2 % H;
        float g[V];
        Vec<T>::load(dz + (size_t)m * dz_ld + c, g);
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
          for (int dx = -1; dx <= 1; ++dx) {
            if (py + dy < 0 || py + dy >= H || px + dx < 0 || px + dx >= W) continue;
            float f[V];
            Vec<T>::load(x + (size_t)(m + dy * W + dx) * x_ld + c, f);
#pragma unroll
            for (int i = 0; i < V; ++i) acc[(dy + 1) * 3 + dx + 1][i] = fmaf(g[i], f[i], acc[(dy + 1) * 3 + dx + 1][i]);
          }
      }
    // fold the rpb pixel rows of the block, tap by tap, in a fixed order
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < V; ++i) wred[threadIdx.x * V + i] = acc[t][i];
      __syncthreads();
      if (cm.trow == 0 && cv < cvn) {
#pragma unroll
        for (int i = 0; i < V; ++i) {
          float s = acc[t][i];
          for (int k = 1; k < cm.rpb; ++k) s += wred[(k * cm.cols + cm.tcol) * V + i];
          mine[(size_t)(c + i) * 9 + t] = s;
        }
      }
    }
    __syncthreads();
  }
}

__global__ void dwconv_wgrad_fold_kernel(const float* __restrict__ partial, int R, int n, float* __restrict__ dw, int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int r = 0; r < R; ++r) s += partial[(size_t)r * n + i];
  dw[i] = accumulate ? dw[i] + s : s;
}

inline int dw_grid(int M, int cvn) {
  const int cols = cvn < 256 ? cvn : 256, rpb = 256 / cols;
  int g = (M + rpb - 1) / rpb;
  if (g > 2048) g = 2048;
  return g < 1 ? 1 : g;
}

}  // namespace

using plyolo::submit;

#define DISPATCH_T(dtype, ...)                       \
  if ((dtype) == PLYOLO_BF16) { typedef bf16_t T; __VA_ARGS__ } \
  else { typedef float T; __VA_ARGS__ }

extern "C" {

static int dw_check(int dtype, int N, int H, int W, int C, int a_ld, int b_ld, const char* who) {
  const int V = dtype == PLYOLO_BF16 ? 8 : 4;
  PLY_CHECK_ARG(dtype == PLYOLO_BF16 || dtype == PLYOLO_F32, "%s: bad dtype %d", who, dtype);
  PLY_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && (double)N * H * W < 2147483000.0, "%s: bad dimensions", who);
  PLY_CHECK_ARG(C % V == 0 && a_ld % V == 0 && b_ld % V == 0 && a_ld >= C && b_ld >= C, "%s: C / pitches must be multiples of %d (C %d, %d, %d)", who, V, C, a_ld, b_ld);
  return 0;
}

int plyolo_dwconv3x3_fwd(int dtype, int N, int H, int W, int C, const void* x, int x_ld, const float* w, void* y, int y_ld,
                         double* stats, void* stream) {
  if (dw_check(dtype, N, H, W, C, x_ld, y_ld, "dwconv3x3_fwd")) return -1;
  PLY_CHECK_ARG(x && w && y, "dwconv3x3_fwd: null pointer");
  const int V = dtype == PLYOLO_BF16 ? 8 : 4, M = N * H * W;
  const int grid = dw_grid(M, C / V);
  plyolo::annotate("dwconv3x3_fwd", 2.0 * 9 * (double)M * C, (double)M * C * (dtype == PLYOLO_BF16 ? 2.0 : 4.0) * 2.0);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    DISPATCH_T(dtype, hipLaunchKernelGGL(dwconv3x3_kernel<T>, dim3(grid), dim3(256), 0, s, N, H, W, C, (const T*)x, x_ld, w, 0, (T*)y, y_ld, 0, stats);)
    return hipGetLastError();
  });
}

int plyolo_dwconv3x3_dgrad(int dtype, int N, int H, int W, int C, const void* dy, int dy_ld, const float* w, void* dx, int dx_ld,
                           int accumulate, void* stream) {
  if (dw_check(dtype, N, H, W, C, dy_ld, dx_ld, "dwconv3x3_dgrad")) return -1;
  PLY_CHECK_ARG(dy && w && dx, "dwconv3x3_dgrad: null pointer");
  const int V = dtype == PLYOLO_BF16 ? 8 : 4, M = N * H * W;
  const int grid = dw_grid(M, C / V);
  plyolo::annotate("dwconv3x3_dgrad", 2.0 * 9 * (double)M * C, (double)M * C * (dtype == PLYOLO_BF16 ? 2.0 : 4.0) * (accumulate ? 3.0 : 2.0));
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    DISPATCH_T(dtype, hipLaunchKernelGGL(dwconv3x3_kernel<T>, dim3(grid), dim3(256), 0, s, N, H, W, C, (const T*)dy, dy_ld, w, 1, (T*)dx, dx_ld,
                                         accumulate, (double*)nullptr);)
    return hipGetLastError();
  });
}

/* number of fp32 partial slabs ([C][9] each) plyolo_dwconv3x3_wgrad needs in `partial` */
int plyolo_dwconv3x3_wgrad_blocks(int dtype, int N, int H, int W, int C) {
  const int V = dtype == PLYOLO_BF16 ? 8 : 4;
  if (C <= 0 || C % V) return -1;
  int g = dw_grid(N * H * W, C / V);
  return g > 512 ? 512 : g;
}

int plyolo_dwconv3x3_wgrad(int dtype, int N, int H, int W, int C, const void* x, int x_ld, const void* dz, int dz_ld, float* partial,
                           float* dw, int accumulate, void* stream) {
  if (dw_check(dtype, N, H, W, C, x_ld, dz_ld, "dwconv3x3_wgrad")) return -1;
  PLY_CHECK_ARG(x && dz && partial && dw, "dwconv3x3_wgrad: null pointer");
  const int V = dtype == PLYOLO_BF16 ? 8 : 4, M = N * H * W;
  const int R = plyolo_dwconv3x3_wgrad_blocks(dtype, N, H, W, C);
  plyolo::annotate("dwconv3x3_wgrad", 2.0 * 9 * (double)M * C, (double)M * C * (dtype == PLYOLO_BF16 ? 2.0 : 4.0) * 2.0);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    DISPATCH_T(dtype, hipLaunchKernelGGL(dwconv3x3_wgrad_kernel<T>, dim3(R), dim3(256), 256 * V * sizeof(float), s, N, H, W, C, (const T*)x, x_ld,
                                         (const T*)dz, dz_ld, partial);)
    hipLaunchKernelGGL(dwconv_wgrad_fold_kernel, dim3(cdiv(C * 9, 256)), dim3(256), 0, s, partial, R, C * 9, dw, accumulate);
    return hipGetLastError();
  });
}

}  // extern "C"
