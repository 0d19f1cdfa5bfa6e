// Train-mode BatchNorm + activation around the convolutions, forward and backward.
//
// Replaces nn.BatchNorm2d(eps=1e-3, momentum=0.03) + SiLU inside BaseConv
// (reference models/layers/normalization.py:8, network_blocks.py:30-37) and what
// autograd does for them.
//
// All of these are HBM streams.  Layout trick shared by the streaming kernels: a
// thread owns ONE 16-byte channel vector column for the whole launch and walks down
// the pixel rows, so the per-channel coefficients live in registers, consecutive
// lanes touch consecutive 16-byte vectors of a pixel row (full 128-byte lines), and
// there is no integer division in the loop.
//   bn_finalize      conv-epilogue partials -> mean/var -> (scale, shift, mean, invstd),
//                    running statistics; chunked over many workgroups, the last one to
//                    arrive finishes (agent-scope release/acquire, counter self-resets)
//   bn_act_fwd       out = act(z*scale+shift) (+ residual)      strided output = concat
//   bn_act_bwd_reduce  per-channel  sum du, sum du*zhat  (du = dout*act'(u)) partials
//   bn_bwd_finalize  dgamma/dbeta + the three coefficients of  dz = A*du + B*z + Cc
//   bn_act_bwd_dz    dz
#include "common.h"

namespace {

constexpr int FIN_CHUNKS = 32;

struct FinWs {
  double* part;        // [FIN_CHUNKS][2][C]
  unsigned* counter;   // [ceil(C/32)]
};

__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* stats, int rows, int C, double count, const float* gamma,
                                                          const float* beta, float eps, float momentum, float* rmean, float* rvar,
                                                          int64_t* nbt, float* coef, FinWs ws, int nchunk) {
  // 32 channels x 8 row-slices per block; blockIdx.y = row chunk
  __shared__ double red[2][8][32];
  __shared__ int s_last;
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  const int per = (rows + nchunk - 1) / nchunk;
  const int r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
  double s = 0.0, ss = 0.0;
  if (c < C)
    for (int r = r0 + sl; r < r1; r += 8) {
      s += stats[(size_t)r * C + c];
      ss += stats[((size_t)rows + r) * C + c];
    }
  red[0][sl][cl] = s;
  red[1][sl][cl] = ss;
  __syncthreads();
  if (sl == 0 && c < C) {
    for (int k = 1; k < 8; ++k) { s += red[0][k][cl]; ss += red[1][k][cl]; }
    ws.part[((size_t)blockIdx.y * 2 + 0) * C + c] = s;
    ws.part[((size_t)blockIdx.y * 2 + 1) * C + c] = ss;
  }
  // publish this chunk, then take a ticket; the last arriver reduces all chunks
  __threadfence();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // keep the drain (ROCm 7.2 may drop the fence's own wait)
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = atomicAdd(&ws.counter[blockIdx.x], 1u);
    s_last = (t == (unsigned)nchunk - 1u) ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  if (sl == 0 && c < C) {
    s = 0.0; ss = 0.0;
    for (int k = 0; k < nchunk; ++k) {
      s += ws.part[((size_t)k * 2 + 0) * C + c];
      ss += ws.part[((size_t)k * 2 + 1) * C + c];
    }
    const double mean = s / count;
    double var = ss / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float scale = g * invstd;
    coef[c] = scale;
    coef[C + c] = b - (float)mean * scale;
    coef[2 * C + c] = (float)mean;
    coef[3 * C + c] = invstd;
    if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
    if (rvar) {
      const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
      rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
    }
  }
  if (threadIdx.x == 0) {
    ws.counter[blockIdx.x] = 0u;  // ready for the next launch
    if (nbt && blockIdx.x == 0) *nbt += 1;
  }
}

__global__ void bn_eval_coef_kernel(int C, const float* gamma, const float* beta, const float* rmean, const float* rvar,
                                    float eps, float* coef) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float invstd = 1.0f / sqrtf(rvar[c] + eps);
  const float scale = (gamma ? gamma[c] : 1.f) * invstd;
  coef[c] = scale;
  coef[C + c] = (beta ? beta[c] : 0.f) - rmean[c] * scale;
  coef[2 * C + c] = rmean[c];
  coef[3 * C + c] = invstd;
}

// column-fixed streaming layout: cols = min(C/V, 256) vector columns, rpb = 256/cols rows per block pass
struct ColMap {
  int cols, rpb, tcol, trow;
  DEVINL ColMap(int cvn) {
    cols = cvn < 256 ? cvn : 256;
    rpb = 256 / cols;
    tcol = threadIdx.x % cols;
    trow = threadIdx.x / cols;
  }
};

template <typename T>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(int M, int C, const T* __restrict__ z, int z_ld, const float* __restrict__ coef,
                                                         int act, const T* __restrict__ res, int r_ld, T* __restrict__ out, int o_ld) {
  constexpr int V = Vec<T>::N;
  const int cvn = C / V;
  const ColMap cm(cvn);
  if (cm.trow >= cm.rpb) return;
  const int step = gridDim.x * cm.rpb;
  for (int cv = cm.tcol; cv < cvn; cv += cm.cols) {
    const int c = cv * V;
    float sc[V], sh[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
      sc[i] = coef ? coef[c + i] : 1.f;
      sh[i] = coef ? coef[C + c + i] : 0.f;
    }
#pragma unroll 2
    for (int m = blockIdx.x * cm.rpb + cm.trow; m < M; m += step) {
      float f[V], r[V];
      Vec<T>::load(z + (size_t)m * z_ld + c, f);
      if (res) Vec<T>::load(res + (size_t)m * r_ld + c, r);
#pragma unroll
      for (int i = 0; i < V; ++i) {
        const float u = actf<Vec<T>::precise>(fmaf(f[i], sc[i], sh[i]), act);
        f[i] = res ? u + r[i] : u;
      }
      Vec<T>::store(out + (size_t)m * o_ld + c, f);
    }
  }
}

// partial[0][row][c] = sum du ; partial[1][row][c] = sum du * zhat ; one partial row per block
template <typename T>
__global__ __launch_bounds__(256) void bn_act_bwd_reduce_kernel(int M, int C, const T* __restrict__ dout, int d_ld, const T* __restrict__ z,
                                                                int z_ld, const float* __restrict__ coef, int act, float* partial, int rows) {
  constexpr int V = Vec<T>::N;
  __shared__ float red[256 * 2 * V];
  const int cvn = C / V;
  const ColMap cm(cvn);
  const int step = gridDim.x * cm.rpb;
  const int row = blockIdx.x;
  for (int cv0 = 0; cv0 < cvn; cv0 += cm.cols) {
    const int cv = cv0 + cm.tcol;
    float s1[V], s2[V];
#pragma unroll
    for (int i = 0; i < V; ++i) s1[i] = s2[i] = 0.f;
    if (cm.trow < cm.rpb && cv < cvn) {
      const int c = cv * V;
      float sc[V], sh[V], mu[V], is[V];
#pragma unroll
      for (int i = 0; i < V; ++i) { sc[i] = coef[c + i]; sh[i] = coef[C + c + i]; mu[i] = coef[2 * C + c + i]; is[i] = coef[3 * C + c + i]; }
#pragma unroll 2
      for (int m = blockIdx.x * cm.rpb + cm.trow; m < M; m += step) {
        float d[V], zz[V];
        Vec<T>::load(dout + (size_t)m * d_ld + c, d);
        Vec<T>::load(z + (size_t)m * z_ld + c, zz);
#pragma unroll
        for (int i = 0; i < V; ++i) {
          const float u = fmaf(zz[i], sc[i], sh[i]);
          const float du = d[i] * act_grad(u, act);
          s1[i] += du;
          s2[i] += du * ((zz[i] - mu[i]) * is[i]);
        }
      }
    }
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
        partial[(size_t)row * C + cv * V + i] = s1[i];
        partial[((size_t)rows + row) * C + cv * V + i] = s2[i];
      }
    }
  }
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* partial, int rows, int C, double count, const float* gamma,
                                                              const float* coef, float* dgamma, float* dbeta, int accumulate, float* bcoef,
                                                              FinWs ws, int nchunk) {
  // same chunked last-arriver reduction as bn_finalize_kernel
  __shared__ double red[2][8][32];
  __shared__ int s_last;
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  const int per = (rows + nchunk - 1) / nchunk;
  const int r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
  double s = 0.0, ss = 0.0;
  if (c < C)
    for (int r = r0 + sl; r < r1; r += 8) {
      s += partial[(size_t)r * C + c];
      ss += partial[((size_t)rows + r) * C + c];
    }
  red[0][sl][cl] = s;
  red[1][sl][cl] = ss;
  __syncthreads();
  if (sl == 0 && c < C) {
    for (int k = 1; k < 8; ++k) { s += red[0][k][cl]; ss += red[1][k][cl]; }
    ws.part[((size_t)blockIdx.y * 2 + 0) * C + c] = s;
    ws.part[((size_t)blockIdx.y * 2 + 1) * C + c] = ss;
  }
  __threadfence();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = atomicAdd(&ws.counter[blockIdx.x], 1u);
    s_last = (t == (unsigned)nchunk - 1u) ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  if (sl == 0 && c < C) {
    s = 0.0; ss = 0.0;
    for (int k = 0; k < nchunk; ++k) {
      s += ws.part[((size_t)k * 2 + 0) * C + c];
      ss += ws.part[((size_t)k * 2 + 1) * C + c];
    }
    const float db = (float)s, dg = (float)ss;
    if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + db;
    if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + dg;
    const float mean = coef[2 * C + c], invstd = coef[3 * C + c];
    const float A = (gamma ? gamma[c] : 1.f) * invstd;
    const float B = (float)(-(double)A * (ss / count) * (double)invstd);
    const float Cc = (float)(-(double)A * (s / count) - (double)B * (double)mean);
    bcoef[c] = A;
    bcoef[C + c] = B;
    bcoef[2 * C + c] = Cc;
  }
  if (threadIdx.x == 0) ws.counter[blockIdx.x] = 0u;
}

template <typename T>
__global__ __launch_bounds__(256) void bn_act_bwd_dz_kernel(int M, int C, const T* __restrict__ dout, int d_ld, const T* __restrict__ z,
                                                            int z_ld, const float* __restrict__ coef, const float* __restrict__ bcoef,
                                                            int act, T* __restrict__ dz, int dz_ld) {
  constexpr int V = Vec<T>::N;
  const int cvn = C / V;
  const ColMap cm(cvn);
  if (cm.trow >= cm.rpb) return;
  const int step = gridDim.x * cm.rpb;
  for (int cv = cm.tcol; cv < cvn; cv += cm.cols) {
    const int c = cv * V;
    float sc[V], sh[V], A[V], B[V], Cc[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
      sc[i] = coef[c + i]; sh[i] = coef[C + c + i];
      A[i] = bcoef[c + i]; B[i] = bcoef[C + c + i]; Cc[i] = bcoef[2 * C + c + i];
    }
#pragma unroll 2
    for (int m = blockIdx.x * cm.rpb + cm.trow; m < M; m += step) {
      float d[V], zz[V];
      Vec<T>::load(dout + (size_t)m * d_ld + c, d);
      Vec<T>::load(z + (size_t)m * z_ld + c, zz);
#pragma unroll
      for (int i = 0; i < V; ++i) {
        const float u = fmaf(zz[i], sc[i], sh[i]);
        const float du = d[i] * act_grad(u, act);
        d[i] = fmaf(A[i], du, fmaf(B[i], zz[i], Cc[i]));
      }
      Vec<T>::store(dz + (size_t)m * dz_ld + c, d);
    }
  }
}

inline int stream_grid(int M, int cvn) {
  const int cols = cvn < 256 ? cvn : 256, rpb = 256 / cols;
  int g = (M + rpb - 1) / rpb;
  if (g > 2048) g = 2048;  // 256 CUs x 8 workgroups, grid-stride the rest
  return g < 1 ? 1 : g;
}

}  // namespace

using plyolo::submit;

#define DISPATCH_T(dtype, ...)                       \
  if ((dtype) == PLYOLO_BF16) { typedef bf16_t T; __VA_ARGS__ } \
  else { typedef float T; __VA_ARGS__ }

extern "C" {

// layout: arrival counters FIRST (fixed offset: the workspace is shared by layers of different C),
// chunk partials after them
constexpr size_t FIN_COUNTER_BYTES = 1024;  // up to 256 column blocks = 8192 channels
size_t plyolo_bn_finalize_workspace(int C) { return FIN_COUNTER_BYTES + (size_t)FIN_CHUNKS * 2 * C * sizeof(double); }

int plyolo_bn_finalize(const float* stats, int rows, int C, double count, const float* gamma, const float* beta, float eps,
                       float momentum, float* running_mean, float* running_var, int64_t* nbt, float* coef, void* workspace,
                       size_t ws_bytes, void* stream) {
  PLY_CHECK_ARG(workspace != nullptr && ws_bytes >= plyolo_bn_finalize_workspace(C), "bn_finalize: workspace too small");
  FinWs ws;
  ws.counter = (unsigned*)workspace;
  ws.part = (double*)((unsigned char*)workspace + FIN_COUNTER_BYTES);
  int nchunk = rows / 32;
  if (nchunk < 1) nchunk = 1;
  if (nchunk > FIN_CHUNKS) nchunk = FIN_CHUNKS;
  plyolo::annotate("bn_finalize", 0.0, 8.0 * rows * C);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 32), nchunk), dim3(256), 0, s, stats, rows, C, count, gamma, beta, eps, momentum,
                       running_mean, running_var, nbt, coef, ws, nchunk);
    return hipGetLastError();
  });
}

int plyolo_bn_eval_coef(int C, const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                        float eps, float* coef, void* stream) {
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    hipLaunchKernelGGL(bn_eval_coef_kernel, dim3(cdiv(C, 256)), dim3(256), 0, s, C, gamma, beta, running_mean, running_var, eps, coef);
    return hipGetLastError();
  });
}

int plyolo_bn_act_fwd(int dtype, int M, int C, const void* z, int z_ld, const float* coef, int act, const void* res, int r_ld,
                      void* out, int o_ld, void* stream) {
  const int V = dtype == PLYOLO_BF16 ? 8 : 4;
  PLY_CHECK_ARG(C % V == 0 && z_ld % V == 0 && o_ld % V == 0 && (!res || r_ld % V == 0), "bn_act_fwd: C/ld must be multiples of %d", V);
  const int grid = stream_grid(M, C / V);
  plyolo::annotate("bn_act_fwd", 0.0, (double)M * C * (dtype == PLYOLO_BF16 ? 2.0 : 4.0) * (res ? 3.0 : 2.0));
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    DISPATCH_T(dtype, hipLaunchKernelGGL(bn_act_fwd_kernel<T>, dim3(grid), dim3(256), 0, s, M, C, (const T*)z, z_ld, coef, act,
                                         (const T*)res, r_ld, (T*)out, o_ld);)
    return hipGetLastError();
  });
}

int plyolo_bn_bwd_rows(int M) {
  int r = M / 64;
  if (r < 1) r = 1;
  if (r > 1024) r = 1024;
  return r;
}

int plyolo_bn_act_bwd_reduce(int dtype, int M, int C, const void* dout, int d_ld, const void* z, int z_ld, const float* coef,
                             int act, float* partial, void* stream) {
  const int V = dtype == PLYOLO_BF16 ? 8 : 4;
  PLY_CHECK_ARG(C % V == 0 && z_ld % V == 0 && d_ld % V == 0, "bn_act_bwd_reduce: C/ld must be multiples of %d", V);
  const int rows = plyolo_bn_bwd_rows(M);
  plyolo::annotate("bn_act_bwd_reduce", 0.0, (double)M * C * (dtype == PLYOLO_BF16 ? 2.0 : 4.0) * 2.0);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    DISPATCH_T(dtype, hipLaunchKernelGGL(bn_act_bwd_reduce_kernel<T>, dim3(rows), dim3(256), 0, s, M, C, (const T*)dout, d_ld,
                                         (const T*)z, z_ld, coef, act, partial, rows);)
    return hipGetLastError();
  });
}

int plyolo_bn_bwd_finalize(const float* partial, int rows, int C, double count, const float* gamma, const float* coef,
                           float* dgamma, float* dbeta, int accumulate, float* bcoef, void* workspace, size_t ws_bytes, void* stream) {
  PLY_CHECK_ARG(workspace != nullptr && ws_bytes >= plyolo_bn_finalize_workspace(C), "bn_bwd_finalize: workspace too small");
  FinWs ws;
  ws.counter = (unsigned*)workspace;
  ws.part = (double*)((unsigned char*)workspace + FIN_COUNTER_BYTES);
  int nchunk = rows / 32;
  if (nchunk < 1) nchunk = 1;
  if (nchunk > FIN_CHUNKS) nchunk = FIN_CHUNKS;
  plyolo::annotate("bn_bwd_finalize", 0.0, 8.0 * rows * C);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, 32), nchunk), dim3(256), 0, s, partial, rows, C, count, gamma, coef, dgamma,
                       dbeta, accumulate, bcoef, ws, nchunk);
    return hipGetLastError();
  });
}

int plyolo_bn_act_bwd_dz(int dtype, int M, int C, const void* dout, int d_ld, const void* z, int z_ld, const float* coef,
                         const float* bcoef, int act, void* dz, int dz_ld, void* stream) {
  const int V = dtype == PLYOLO_BF16 ? 8 : 4;
  PLY_CHECK_ARG(C % V == 0 && z_ld % V == 0 && d_ld % V == 0 && dz_ld % V == 0, "bn_act_bwd_dz: C/ld must be multiples of %d", V);
  const int grid = stream_grid(M, C / V);
  plyolo::annotate("bn_act_bwd_dz", 0.0, (double)M * C * (dtype == PLYOLO_BF16 ? 2.0 : 4.0) * 3.0);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    DISPATCH_T(dtype, hipLaunchKernelGGL(bn_act_bwd_dz_kernel<T>, dim3(grid), dim3(256), 0, s, M, C, (const T*)dout, d_ld, (const T*)z,
                                         z_ld, coef, bcoef, act, (T*)dz, dz_ld);)
    return hipGetLastError();
  });
}

}  // extern "C"
