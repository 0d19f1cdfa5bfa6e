// HBM-bound kernels around the convolutions: BatchNorm statistics/normalise +
// activation (fwd and bwd), Focus space-to-depth, concat-slice copies, nearest
// upsample, SPP max-pools, weight (re)packing, SGD/EMA.  All activations are NHWC
// with a row pitch (`ld`), every thread moves whole 16-byte vectors.
//
// Reference semantics: BaseConv = act(bn(conv(x))) models/layers/network_blocks.py:30-37,
// BatchNorm2d(eps 1e-3, momentum 0.03) models/layers/normalization.py:8, Focus
// network_blocks.py:50-65, SPP pools :144, nn.Upsample(nearest x2) necks/pafpn_csp.py:22.
#include <type_traits>

#include "common.h"

namespace {

// ------------------------------------------------------------- data movement
template <typename T>
__global__ void focus_kernel(const float* img, int N, int H, int W, T* out, int Cp) {
  const int OH = H / 2, OW = W / 2;
  const size_t total = (size_t)N * OH * OW;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int ox = (int)(idx % OW);
    const size_t t = idx / OW;
    const int oy = (int)(t % OH);
    const int n = (int)(t / OH);
    T* o = out + idx * Cp;
    // blocks: TL (0,0), BL (1,0), TR (0,1), BR (1,1) -- network_blocks.py:52-63
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int dy = b & 1, dx = b >> 1;
#pragma unroll
      for (int ch = 0; ch < 3; ++ch)
        ActT<T>::st(o + b * 3 + ch, img[((size_t)(n * 3 + ch) * H + 2 * oy + dy) * W + 2 * ox + dx]);
    }
    for (int c = 12; c < Cp; ++c) ActT<T>::st(o + c, 0.f);
  }
}

template <typename T>
__global__ void copy_add_kernel(size_t M, int C, const T* in, int i_ld, T* out, int o_ld, int accumulate) {
  constexpr int V = Vec<T>::N;
  const int cvn = C / V;
  const size_t total = M * cvn;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const size_t m = idx / cvn;
    const int c = (int)(idx - m * cvn) * V;
    float a[V], b[V];
    if (in) Vec<T>::load(in + m * i_ld + c, a);
    else {
#pragma unroll
      for (int i = 0; i < V; ++i) a[i] = 0.f;
    }
    if (accumulate) {
      Vec<T>::load(out + m * o_ld + c, b);
#pragma unroll
      for (int i = 0; i < V; ++i) a[i] += b[i];
    }
    Vec<T>::store(out + m * o_ld + c, a);
  }
}

template <typename T>
__global__ void upsample2x_fwd_kernel(int N, int H, int W, int C, const T* in, int i_ld, T* out, int o_ld) {
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
    float a[V];
    Vec<T>::load(in + ((size_t)(n * H + oy / 2) * W + ox / 2) * i_ld + c, a);
    Vec<T>::store(out + ((size_t)(n * 2 * H + oy) * (2 * W) + ox) * o_ld + c, a);
  }
}

template <typename T>
__global__ void upsample2x_bwd_kernel(int N, int H, int W, int C, const T* dout, int d_ld, T* din, int i_ld, int accumulate) {
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
    float s[V], a[V];
#pragma unroll
    for (int i = 0; i < V; ++i) s[i] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      Vec<T>::load(dout + ((size_t)(n * 2 * H + 2 * y + (k >> 1)) * (2 * W) + 2 * x + (k & 1)) * d_ld + c, a);
#pragma unroll
      for (int i = 0; i < V; ++i) s[i] += a[i];
    }
    T* dst = din + ((size_t)(n * H + y) * W + x) * i_ld + c;
    if (accumulate) {
      Vec<T>::load(dst, a);
#pragma unroll
      for (int i = 0; i < V; ++i) s[i] += a[i];
    }
    Vec<T>::store(dst, s);
  }
}

template <typename T>
__global__ void maxpool_fwd_kernel(int N, int H, int W, int C, int k, const T* in, int i_ld, T* out, int o_ld) {
  constexpr int V = Vec<T>::N;
  const int cvn = C / V, rad = k / 2;
  const size_t total = (size_t)N * H * W * cvn;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % cvn) * V;
    size_t t = idx / cvn;
    const int x = (int)(t % W);
    t /= W;
    const int y = (int)(t % H);
    const int n = (int)(t / H);
    float best[V], a[V];
#pragma unroll
    for (int i = 0; i < V; ++i) best[i] = -INFINITY;
    for (int yy = max(0, y - rad); yy <= min(H - 1, y + rad); ++yy)
      for (int xx = max(0, x - rad); xx <= min(W - 1, x + rad); ++xx) {
        Vec<T>::load(in + ((size_t)(n * H + yy) * W + xx) * i_ld + c, a);
#pragma unroll
        for (int i = 0; i < V; ++i) best[i] = a[i] > best[i] ? a[i] : best[i];
      }
    Vec<T>::store(out + ((size_t)(n * H + y) * W + x) * o_ld + c, best);
  }
}

// scatter dout to the FIRST maximum of each window (row-major scan, strict '>'), like ATen
template <typename T>
__global__ void maxpool_bwd_kernel(int N, int H, int W, int C, int k, const T* in, int i_ld, const T* dout, int d_ld,
                                   float* din) {
  constexpr int V = Vec<T>::N;
  const int cvn = C / V, rad = k / 2;
  const size_t total = (size_t)N * H * W * cvn;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % cvn) * V;
    size_t t = idx / cvn;
    const int x = (int)(t % W);
    t /= W;
    const int y = (int)(t % H);
    const int n = (int)(t / H);
    float best[V], a[V], g[V];
    int arg[V];
#pragma unroll
    for (int i = 0; i < V; ++i) { best[i] = -INFINITY; arg[i] = -1; }
    for (int yy = max(0, y - rad); yy <= min(H - 1, y + rad); ++yy)
      for (int xx = max(0, x - rad); xx <= min(W - 1, x + rad); ++xx) {
        Vec<T>::load(in + ((size_t)(n * H + yy) * W + xx) * i_ld + c, a);
#pragma unroll
        for (int i = 0; i < V; ++i)
          if (a[i] > best[i] || arg[i] < 0) { best[i] = a[i]; arg[i] = yy * W + xx; }
      }
    Vec<T>::load(dout + ((size_t)(n * H + y) * W + x) * d_ld + c, g);
#pragma unroll
    for (int i = 0; i < V; ++i) atomicAdd(din + ((size_t)n * H * W + arg[i]) * C + c + i, g[i]);
  }
}

// Backward of ALL the stride-1 pools of one SPP block in one launch.  One workgroup owns an
// (image, 8-channel group) plane held in LDS: the first-maximum argmax of a k x k window is
// separable (first row holding the window maximum, first column of that row's maximum), so each
// pool costs 2k LDS reads per element instead of k*k global reads; the routed gradients are
// accumulated in LDS and written once (no fp32 scratch image, no global atomics).
// Determinism: several source pixels route their gradient to the same maximum; the sums are kept in FP64 in LDS
// (ds_add_f64), so the order in which the waves arrive cannot change the rounded result (bf16 / fp32 addends are
// summed exactly unless they span more than 2^29 in magnitude) -- with fp32 LDS atomics two runs of the same step
// differed in the last bit here and the backbone upstream amplified that to percents (YOLOX-x at 1280x1280).
// CG = channels per workgroup plane: 8 (16-byte vectors) while the plane fits in LDS, 4 for the larger maps.
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
template <typename T, int CG> DEVINL void spp_ld(const T* q, float* f) {
  if constexpr (sizeof(T) == 2 && CG == 8) { Vec<T>::load(q, f); }
  else if constexpr (sizeof(T) == 2 && CG == 4) {
    const u32x2 v = *(const u32x2*)q;
    f[0] = __uint_as_float(v[0] << 16); f[1] = __uint_as_float(v[0] & 0xffff0000u);
    f[2] = __uint_as_float(v[1] << 16); f[3] = __uint_as_float(v[1] & 0xffff0000u);
  } else {
#pragma unroll
    for (int i = 0; i < CG; i += 4) Vec<T>::load(q + i, f + i);
  }
}
template <typename T, int CG> DEVINL void spp_st(T* q, const float* f) {
  if constexpr (sizeof(T) == 2 && CG == 8) { Vec<T>::store(q, f); }
  else if constexpr (sizeof(T) == 2 && CG == 4) {
    u32x2 v;
    v[0] = pack2bf(f[0], f[1]); v[1] = pack2bf(f[2], f[3]);
    *(u32x2*)q = v;
  } else {
#pragma unroll
    for (int i = 0; i < CG; i += 4) Vec<T>::store(q + i, f + i);
  }
}

// Forward of the same block in ONE launch: the exact cascade pool_k = pool_(k - k_prev + 1)(pool_k_prev) (5 -> 9 -> 13: three 5x5
// pools) on an (image, CG-channel) plane in LDS, every pool separable (row maximum, then column maximum); one thread = one pixel
// with its CG channels as a 16- / 8-byte vector.  Replaces three plyolo_maxpool_s1_fwd launches (~20 us each on 20x20 maps, each
// re-reading the previous pool's output from memory).
template <typename T, int SPP_CG>
__global__ __launch_bounds__(256) void spp_pools_fwd_kernel(int H, int W, int C, int nk, int w0, int w1, int w2, const T* __restrict__ in,
                                                            int i_ld, T* o0, T* o1, T* o2, int ol0, int ol1, int ol2) {
  extern __shared__ __align__(16) unsigned char spp_smem[];
  const int HW = H * W, n = blockIdx.x, c0 = blockIdx.y * SPP_CG;
  T* cur = (T*)spp_smem;                 // [HW][CG] the plane being pooled
  T* row = cur + (size_t)HW * SPP_CG;    // [HW][CG] its row-pass maximum
  for (int p = threadIdx.x; p < HW; p += 256) {
    float v[SPP_CG];
    spp_ld<T, SPP_CG>(in + ((size_t)n * HW + p) * i_ld + c0, v);
    spp_st<T, SPP_CG>(cur + p * SPP_CG, v);
  }
  __syncthreads();
  for (int j = 0; j < nk; ++j) {
    const int rad = (j == 0 ? w0 : (j == 1 ? w1 : w2)) / 2;
    T* out = j == 0 ? o0 : (j == 1 ? o1 : o2);
    const int ol = j == 0 ? ol0 : (j == 1 ? ol1 : ol2);
    for (int p = threadIdx.x; p < HW; p += 256) {
      const int y = p / W, x = p - y * W;
      const int xa = max(x - rad, 0), xb = min(x + rad, W - 1);
      float best[SPP_CG];
      spp_ld<T, SPP_CG>(cur + (y * W + xa) * SPP_CG, best);
      for (int xi = xa + 1; xi <= xb; ++xi) {
        float v[SPP_CG];
        spp_ld<T, SPP_CG>(cur + (y * W + xi) * SPP_CG, v);
#pragma unroll
        for (int ch = 0; ch < SPP_CG; ++ch) best[ch] = fmaxf(best[ch], v[ch]);
      }
      spp_st<T, SPP_CG>(row + p * SPP_CG, best);
    }
    __syncthreads();
    for (int p = threadIdx.x; p < HW; p += 256) {
      const int y = p / W, x = p - y * W;
      const int ya = max(y - rad, 0), yb = min(y + rad, H - 1);
      float best[SPP_CG];
      spp_ld<T, SPP_CG>(row + (ya * W + x) * SPP_CG, best);
      for (int yi = ya + 1; yi <= yb; ++yi) {
        float v[SPP_CG];
        spp_ld<T, SPP_CG>(row + (yi * W + x) * SPP_CG, v);
#pragma unroll
        for (int ch = 0; ch < SPP_CG; ++ch) best[ch] = fmaxf(best[ch], v[ch]);
      }
      spp_st<T, SPP_CG>(cur + p * SPP_CG, best);      // the next pool of the cascade reads it (nobody reads `cur` in this phase)
      spp_st<T, SPP_CG>(out + ((size_t)n * HW + p) * ol + c0, best);
    }
    __syncthreads();
  }
}

template <typename T, int SPP_CG>
__global__ __launch_bounds__(256) void spp_pools_bwd_kernel(int H, int W, int C, int nk, int k0, int k1, int k2, const T* __restrict__ in,
                                                            int i_ld, const T* d0, const T* d1, const T* d2, int dl0, int dl1, int dl2,
                                                            T* din, int di_ld, int accumulate) {
  extern __shared__ __align__(16) unsigned char spp_smem[];
  const int HW = H * W, n = blockIdx.x, c0 = blockIdx.y * SPP_CG;
  double* gs = (double*)spp_smem;                            // [HW][CG] routed gradient (fp64: order-independent sums)
  T* xs = (T*)(gs + (size_t)HW * SPP_CG);                    // [HW][CG] input plane
  T* rv = xs + (size_t)HW * SPP_CG;                          // [HW][CG] row-pass maximum
  unsigned short* ra = (unsigned short*)(rv + (size_t)HW * SPP_CG);  // [HW][CG] its column
  const int items = HW * SPP_CG;
  for (int it = threadIdx.x; it < items; it += 256) {
    const int p = it / SPP_CG, ch = it % SPP_CG;
    xs[it] = (c0 + ch < C) ? in[((size_t)n * HW + p) * i_ld + c0 + ch] : (T)0;
    gs[it] = 0.0;
  }
  __syncthreads();
  for (int j = 0; j < nk; ++j) {
    const int k = j == 0 ? k0 : (j == 1 ? k1 : k2);
    const T* dout = j == 0 ? d0 : (j == 1 ? d1 : d2);
    const int dl = j == 0 ? dl0 : (j == 1 ? dl1 : dl2);
    if (dout == nullptr) continue;
    const int rad = k / 2;
    // Window scans with a compile-time trip count (5 / 9 / 13, the SPP kernels of every shipped config): all LDS
    // reads of a window are issued before the first compare (a data-dependent loop serialised read -> compare ->
    // read: ~100 cycles of LDS latency x 54 reads per element).  Out-of-range taps are clamped onto the border
    // element: a duplicate of an element already seen never wins a strict '>' -- the first maximum is unchanged.
    // One thread = one pixel, all CG channels of the plane in registers: 16- / 8-byte LDS vectors instead of one
    // 2-byte read per (pixel, channel, tap) -- the element-wise form issued ~1000 LDS instructions per thread and
    // was bound by the LDS instruction rate.
    auto pool = [&](auto KC) {
      constexpr int K = decltype(KC)::value;
      for (int p = threadIdx.x; p < HW; p += 256) {
        const int y = p / W, x = p - y * W;
        float best[SPP_CG];
        int ax[SPP_CG];
#pragma unroll
        for (int d = 0; d < K; ++d) {
          const int xi = min(max(x + d - K / 2, 0), W - 1);
          float v[SPP_CG];
          spp_ld<T, SPP_CG>(xs + (y * W + xi) * SPP_CG, v);
#pragma unroll
          for (int ch = 0; ch < SPP_CG; ++ch)
            if (d == 0 || v[ch] > best[ch]) { best[ch] = v[ch]; ax[ch] = xi; }
        }
        spp_st<T, SPP_CG>(rv + p * SPP_CG, best);
#pragma unroll
        for (int ch = 0; ch < SPP_CG; ++ch) ra[p * SPP_CG + ch] = (unsigned short)ax[ch];
      }
      __syncthreads();
      for (int p = threadIdx.x; p < HW; p += 256) {
        const int y = p / W, x = p - y * W;
        float best[SPP_CG];
        int ay[SPP_CG];
#pragma unroll
        for (int d = 0; d < K; ++d) {
          const int yi = min(max(y + d - K / 2, 0), H - 1);
          float v[SPP_CG];
          spp_ld<T, SPP_CG>(rv + (yi * W + x) * SPP_CG, v);
#pragma unroll
          for (int ch = 0; ch < SPP_CG; ++ch)
            if (d == 0 || v[ch] > best[ch]) { best[ch] = v[ch]; ay[ch] = yi; }
        }
        float g[SPP_CG];
        spp_ld<T, SPP_CG>(dout + ((size_t)n * HW + p) * dl + c0, g);
#pragma unroll
        for (int ch = 0; ch < SPP_CG; ++ch) {
          const int arg = ay[ch] * W + ra[(ay[ch] * W + x) * SPP_CG + ch];
          atomicAdd(gs + arg * SPP_CG + ch, (double)g[ch]);
        }
      }
      __syncthreads();
    };
    const bool vec_ok = c0 + SPP_CG <= C && dl % SPP_CG == 0 && ((size_t)dout & 15) == 0;
    if (vec_ok && k == 5) { pool(std::integral_constant<int, 5>{}); continue; }
    if (vec_ok && k == 9) { pool(std::integral_constant<int, 9>{}); continue; }
    if (vec_ok && k == 13) { pool(std::integral_constant<int, 13>{}); continue; }
    for (int it = threadIdx.x; it < items; it += 256) {
      const int p = it / SPP_CG, ch = it % SPP_CG;
      const int y = p / W, x = p - y * W;
      const int xa = max(0, x - rad), xb = min(W - 1, x + rad);
      float best = ActT<T>::ld(xs + (y * W + xa) * SPP_CG + ch);
      int ax = xa;
      for (int xx = xa + 1; xx <= xb; ++xx) {
        const float v = ActT<T>::ld(xs + (y * W + xx) * SPP_CG + ch);
        if (v > best) { best = v; ax = xx; }
      }
      ActT<T>::st(rv + it, best);
      ra[it] = (unsigned short)ax;
    }
    __syncthreads();
    for (int it = threadIdx.x; it < items; it += 256) {
      const int p = it / SPP_CG, ch = it % SPP_CG;
      if (c0 + ch >= C) continue;
      const int y = p / W, x = p - y * W;
      const int ya = max(0, y - rad), yb = min(H - 1, y + rad);
      float best = ActT<T>::ld(rv + (ya * W + x) * SPP_CG + ch);
      int ay = ya;
      for (int yy = ya + 1; yy <= yb; ++yy) {
        const float v = ActT<T>::ld(rv + (yy * W + x) * SPP_CG + ch);
        if (v > best) { best = v; ay = yy; }
      }
      const int arg = ay * W + ra[(ay * W + x) * SPP_CG + ch];
      const float g = ActT<T>::ld(dout + ((size_t)n * HW + p) * dl + c0 + ch);
      atomicAdd(gs + arg * SPP_CG + ch, (double)g);
    }
    __syncthreads();
  }
  for (int it = threadIdx.x; it < items; it += 256) {
    const int p = it / SPP_CG, ch = it % SPP_CG;
    if (c0 + ch >= C) continue;
    T* dst = din + ((size_t)n * HW + p) * di_ld + c0 + ch;
    float v = (float)gs[it];
    if (accumulate) v += ActT<T>::ld(dst);
    ActT<T>::st(dst, v);
  }
}

// MaxPool2d(kernel 2, stride 2) of the YOLOv7 Transition blocks (reference
// models/backbones/eelan.py:129, models/necks/yolov7_neck.py:152): forward, and the backward
// that routes dout to the FIRST maximum of each 2x2 window in row-major order (ATen rule).
template <typename T>
__global__ void maxpool2x2_fwd_kernel(int N, int H, int W, int C, const T* in, int i_ld, T* out, int o_ld) {
  constexpr int V = Vec<T>::N;
  const int cvn = C / V, OH = H / 2, OW = W / 2;
  const size_t total = (size_t)N * OH * OW * cvn;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % cvn) * V;
    size_t t = idx / cvn;
    const int x = (int)(t % OW);
    t /= OW;
    const int y = (int)(t % OH);
    const int n = (int)(t / OH);
    float best[V], a[V];
#pragma unroll
    for (int i = 0; i < V; ++i) best[i] = -INFINITY;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      Vec<T>::load(in + ((size_t)(n * H + 2 * y + (k >> 1)) * W + 2 * x + (k & 1)) * i_ld + c, a);
#pragma unroll
      for (int i = 0; i < V; ++i) best[i] = a[i] > best[i] ? a[i] : best[i];
    }
    Vec<T>::store(out + ((size_t)(n * OH + y) * OW + x) * o_ld + c, best);
  }
}

template <typename T>
__global__ void maxpool2x2_bwd_kernel(int N, int H, int W, int C, const T* in, int i_ld, const T* dout, int d_ld, T* din, int di_ld,
                                      int accumulate) {
  constexpr int V = Vec<T>::N;
  const int cvn = C / V, OH = H / 2, OW = W / 2;
  const size_t total = (size_t)N * OH * OW * cvn;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % cvn) * V;
    size_t t = idx / cvn;
    const int x = (int)(t % OW);
    t /= OW;
    const int y = (int)(t % OH);
    const int n = (int)(t / OH);
    float a[4][V], g[V];
#pragma unroll
    for (int k = 0; k < 4; ++k) Vec<T>::load(in + ((size_t)(n * H + 2 * y + (k >> 1)) * W + 2 * x + (k & 1)) * i_ld + c, a[k]);
    Vec<T>::load(dout + ((size_t)(n * OH + y) * OW + x) * d_ld + c, g);
    int arg[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
      float best = a[0][i];
      arg[i] = 0;
#pragma unroll
      for (int k = 1; k < 4; ++k)
        if (a[k][i] > best) { best = a[k][i]; arg[i] = k; }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      T* dst = din + ((size_t)(n * H + 2 * y + (k >> 1)) * W + 2 * x + (k & 1)) * di_ld + c;
      float o[V];
      if (accumulate) Vec<T>::load(dst, o);
#pragma unroll
      for (int i = 0; i < V; ++i) o[i] = (accumulate ? o[i] : 0.f) + (arg[i] == k ? g[i] : 0.f);
      Vec<T>::store(dst, o);
    }
  }
}

// ---- ImplicitHead (reference models/heads/implicit_head.py:5-62): y = m * (W (x + a) + b)
// bias_eff[co] = b[co] + sum_ci W[co][ci] * a[ci]
__global__ void implicit_bias_kernel(const float* W, const float* a, const float* b, float* out, int Cout, int Cin) {
  const int co = blockIdx.x;
  __shared__ float red[256];
  float s = 0.f;
  for (int ci = threadIdx.x; ci < Cin; ci += 256) s += W[(size_t)co * Cin + ci] * a[ci];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[co] = b[co] + red[0];
}
// y[r][c] = m[c] * u[r][c]
__global__ void scale_channels_kernel(const float* u, const float* m, float* y, size_t rows, int C) {
  const size_t total = rows * C;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x)
    y[idx] = u[idx] * m[idx % C];
}
// du = m * dy  (fp32 [rows][C] -> activation dtype [rows][du_ld], pad columns untouched);
// dm partial[blk][c] = sum over the block's rows of dy*u
template <typename T>
__global__ void implicit_bwd_kernel(const float* dy, const float* u, const float* m, T* du, int du_ld, float* partial, size_t rows, int C) {
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const float mc = m[c];
    float s = 0.f;
    for (size_t r = blockIdx.x; r < rows; r += gridDim.x) {
      const float g = dy[r * C + c];
      s += g * u[r * C + c];
      ActT<T>::st(du + r * du_ld + c, g * mc);
    }
    partial[(size_t)blockIdx.x * C + c] = s;
  }
}
// dm[c] = sum_blk partial ; da[ci] = sum_co W[co][ci]*sdu[co] ; dW[co][ci] += sdu[co]*a[ci] ; db[co] = sdu[co]
__global__ void implicit_param_grads_kernel(const float* partial, int nblk, const float* W, const float* a, const float* sdu, float* dm,
                                            float* da, float* dW, float* db, int Cout, int Cin) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  {
    // dm: 8 channels x 32 partial groups per workgroup pass, folded through LDS in a fixed order (one thread per
    // channel walking all nblk partials was a 2048-deep chain of dependent loads: 0.3 ms per head level)
    __shared__ float red[256];
    const int cl = threadIdx.x & 7, part = threadIdx.x >> 3;
    for (int c0 = blockIdx.x * 8; c0 < Cout; c0 += gridDim.x * 8) {
      const int c = c0 + cl;
      float s = 0.f;
      if (c < Cout)
        for (int k = part; k < nblk; k += 32) s += partial[(size_t)k * Cout + c];
      red[threadIdx.x] = s;
      __syncthreads();
      if (part == 0 && c < Cout) {
        for (int q = 1; q < 32; ++q) s += red[q * 8 + cl];
        dm[c] = s;
        db[c] = sdu[c];
      }
      __syncthreads();
    }
  }
  {
    // da[ci] = sum_co W[co][ci] * sdu[co]: 32 input channels x 8 co-groups per workgroup pass (same reason)
    __shared__ float red2[256];
    const int il = threadIdx.x & 31, part = threadIdx.x >> 5;
    for (int i0 = blockIdx.x * 32; i0 < Cin; i0 += gridDim.x * 32) {
      const int ci = i0 + il;
      float s = 0.f;
      if (ci < Cin)
        for (int co = part; co < Cout; co += 8) s += W[(size_t)co * Cin + ci] * sdu[co];
      red2[threadIdx.x] = s;
      __syncthreads();
      if (part == 0 && ci < Cin) {
        for (int q = 1; q < 8; ++q) s += red2[q * 32 + il];
        da[ci] = s;
      }
      __syncthreads();
    }
  }
  for (int idx = tid; idx < Cout * Cin; idx += gridDim.x * blockDim.x) dW[idx] += sdu[idx / Cin] * a[idx % Cin];
}

template <typename T>
__global__ void f32_to_act_kernel(size_t M, int C, const float* in, T* out, int o_ld, int accumulate) {
  const size_t total = M * C;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const size_t m = idx / C;
    const int c = (int)(idx - m * C);
    float v = in[idx];
    T* dst = out + m * o_ld + c;
    if (accumulate) v += ActT<T>::ld(dst);
    ActT<T>::st(dst, v);
  }
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(int N, int H, int W, int C, const T* in, int i_ld, float* out) {
  const size_t total = (size_t)N * C * H * W;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(idx % W);
    size_t t = idx / W;
    const int y = (int)(t % H);
    t /= H;
    const int c = (int)(t % C);
    const int n = (int)(t / C);
    out[idx] = ActT<T>::ld(in + ((size_t)(n * H + y) * W + x) * i_ld + c);
  }
}
template <typename T>
__global__ void nchw_to_nhwc_kernel(int N, int H, int W, int C, const float* in, T* out, int o_ld) {
  const size_t total = (size_t)N * C * H * W;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C);
    size_t t = idx / C;
    const int x = (int)(t % W);
    t /= W;
    const int y = (int)(t % H);
    const int n = (int)(t / H);
    ActT<T>::st(out + ((size_t)(n * H + y) * W + x) * o_ld + c, in[((size_t)(n * C + c) * H + y) * W + x]);
  }
}

// ------------------------------------------------------------ weight packing
// fp32 (parity path): wp [tap][Cout_total][Cin_p], wpd [tap][Cin_p][Cout_p8].
// bf16 (MFMA path): both packs are in MFMA B-fragment order (conv_mfma.hip, conv_mfma_pack_elems):
//   wp  fragment (tap, nb, kb) = 64 lanes x 8 values, lane (r = l&31, h = l>>5), j  <-  W[nb*32+r][kb*16+h*8+j][tap]
//   wpd fragment (tap, nb, kb)                                                      <-  W[kb*16+h*8+j][nb*32+r][tap]
// Pad positions are never written (the arenas are zero-initialised once).
// workgroup -> (table entry, slice, slices of that entry): the legacy 2-D launch (entry = blockIdx.x, gridDim.y slices for everyone)
// or a flat launch planned by plyolo_pack_plan (entry i owns workgroups [blk0, blk0 + nblk): slices in proportion to its size)
DEVINL void pack_block(const plyolo_pack_entry* table, int n_flat, int* ent, int* slice, int* nslice) {
  if (n_flat <= 0) { *ent = blockIdx.x; *slice = blockIdx.y; *nslice = gridDim.y; return; }
  // last entry with blk0 <= blockIdx.x.  One thread bisecting the table in global memory was ~7 DEPENDENT round trips in front of
  // every workgroup's work (4400 workgroups in two rounds: a third of the 35 us launch); here the starts are fetched by all
  // threads at once -- one round trip -- and counted
  __shared__ int s_cnt;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  const int b = (int)blockIdx.x;
  int below = 0;
  for (int i = threadIdx.x; i < n_flat; i += blockDim.x) below += table[i].blk0 <= b ? 1 : 0;
  if (below) atomicAdd(&s_cnt, below);
  __syncthreads();
  *ent = s_cnt - 1;            // blk0 is increasing and table[0].blk0 == 0
  *slice = *nslice = -1;       // flat launch: the caller derives them from the entry it loads anyway (no extra round trip)
}

template <typename T>
__global__ void pack_weights_kernel(const plyolo_pack_entry* table, int n_flat) {
  int ent, slice, nslice;
  pack_block(table, n_flat, &ent, &slice, &nslice);
  const plyolo_pack_entry e = table[ent];
  if (nslice < 0) { slice = (int)blockIdx.x - e.blk0; nslice = e.nblk; }
  const int taps = e.ksize * e.ksize;
  const int nf = taps * e.Cout * e.Cin_p;
  T* wp = (T*)e.wp;
  T* wpd = (T*)e.wpd;
  constexpr bool FRAG = sizeof(T) == 2;
  const int nkb_f = (e.Cin_p + 15) / 16, nnb_f = (e.Cout_total + 31) / 32;
  const int nkb_d = (e.Cout_total + 15) / 16, nnb_d = (e.Cin_p + 31) / 32;
  if (FRAG && (e.co_off & 7) == 0) {
    // one 16-byte fragment chunk (8 consecutive j) per thread.  fwd chunk (co, kb, h, t): ci = kb*16 + h*8 + j
    const int nck = nkb_f * 2;
    const int nfw = e.Cout * nck * taps;
    const int c8 = (e.Cout + 7) / 8;
    const int ndg = wpd ? e.Cin_p * c8 * taps : 0;
    // A workgroup owns ~256 chunks of either pack (plyolo_pack_plan: 2048 weights), i.e. ONE chunk of each per thread: the eight
    // values of the data-gradient chunk are requested before the forward chunk is converted and stored, so the two chunks cost
    // one round trip instead of two (the launch is a chain of round trips: table, entry, forward chunk, data-gradient chunk)
    const int idx_d0 = slice * blockDim.x + threadIdx.x;
    float fd0[8];
    {
      const int t = idx_d0 % taps, ci = (idx_d0 / taps) % e.Cin_p, co0 = (idx_d0 / (taps * e.Cin_p)) * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) fd0[j] = (idx_d0 < ndg && ci < e.Cin && co0 + j < e.Cout) ? e.w[((size_t)(co0 + j) * e.Cin + ci) * taps + t] : 0.f;
    }
    for (int idx = slice * blockDim.x + threadIdx.x; idx < nfw; idx += nslice * blockDim.x) {
      const int t = idx % taps, ck = (idx / taps) % nck, co = idx / (taps * nck);
      const int kb = ck >> 1, h = ck & 1, ci0 = kb * 16 + h * 8, cot = e.co_off + co;
      const float* src = e.w + ((size_t)co * e.Cin + ci0) * taps + t;
      float f[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = ci0 + j < e.Cin ? src[(size_t)j * taps] : 0.f;
      u32x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = pack2bf(f[2 * j], f[2 * j + 1]);
      *(u32x4*)((bf16_t*)e.wp + (((size_t)t * nnb_f + (cot >> 5)) * nkb_f + kb) * 512 + (h * 32 + (cot & 31)) * 8) = v;
    }
    if (wpd) {
      // dgrad chunk (ci, kbd, h, t): cot = kbd*16 + h*8 + j, restricted to this entry's rows
      for (int idx = idx_d0; idx < ndg; idx += nslice * blockDim.x) {
        const int t = idx % taps, ci = (idx / taps) % e.Cin_p, g8 = idx / (taps * e.Cin_p);
        const int co0 = g8 * 8, cot0 = e.co_off + co0;
        float f[8];
        if (idx == idx_d0) {
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] = fd0[j];
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] = (ci < e.Cin && co0 + j < e.Cout) ? e.w[((size_t)(co0 + j) * e.Cin + ci) * taps + t] : 0.f;
        }
        u32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = pack2bf(f[2 * j], f[2 * j + 1]);
        const int kb = cot0 >> 4, h = (cot0 >> 3) & 1;
        bf16_t* dst = (bf16_t*)e.wpd + (((size_t)t * nnb_d + (ci >> 5)) * nkb_d + kb) * 512 + (h * 32 + (ci & 31)) * 8;
        if (co0 + 8 <= e.Cout) {
          *(u32x4*)dst = v;
        } else {  // the next entry of a merged conv (or the zero pad) owns the rest of this chunk
          for (int j = 0; co0 + j < e.Cout; ++j) dst[j] = f2bf(f[j]);
        }
      }
    }
  } else
  for (int idx = slice * blockDim.x + threadIdx.x; idx < nf; idx += nslice * blockDim.x) {
    const int ci = idx % e.Cin_p;
    const int co = (idx / e.Cin_p) % e.Cout;
    const int t = idx / (e.Cin_p * e.Cout);
    const float v = ci < e.Cin ? e.w[((size_t)co * e.Cin + ci) * taps + t] : 0.f;
    const int cot = e.co_off + co;
    if (FRAG) {
      {
        const int nb = cot >> 5, r = cot & 31, kb = ci >> 4, h = (ci >> 3) & 1, j = ci & 7;
        ActT<T>::st(wp + (((size_t)t * nnb_f + nb) * nkb_f + kb) * 512 + (h * 32 + r) * 8 + j, v);
      }
      if (wpd) {
        const int nb = ci >> 5, r = ci & 31, kb = cot >> 4, h = (cot >> 3) & 1, j = cot & 7;
        ActT<T>::st(wpd + (((size_t)t * nnb_d + nb) * nkb_d + kb) * 512 + (h * 32 + r) * 8 + j, v);
      }
    } else {
      ActT<T>::st(wp + ((size_t)t * e.Cout_total + cot) * e.Cin_p + ci, v);
      if (wpd) ActT<T>::st(wpd + ((size_t)t * e.Cin_p + ci) * e.Cout_p8 + cot, v);
    }
  }
  if (e.b && slice == 0)
    for (int i = threadIdx.x; i < e.Cout; i += blockDim.x) e.bp[e.co_off + i] = e.b[i];
}

// slab 0 += slabs 1..n-1 (fixed order).  Runs right after a wgrad launch (on the weight-gradient lane), so
// that the final unpack only permutes one slab per convolution.
// blockIdx.y = slab group: group g folds slabs [g*per, min((g+1)*per, nslab)) into slab g*per (in place).  A small
// weight tensor with up to 1024 slabs has only a handful of column blocks, so the slab range is first folded by
// many groups in parallel and the few group heads are folded by a second launch (stride = per slabs); the
// summation order is fixed by (per, nslab), never by timing.
__global__ void reduce_slabs_kernel(float* dwp, int nslab, size_t elems, int per, size_t stride) {
  const int first = blockIdx.y * per;
  const int n = nslab - first < per ? nslab - first : per;
  float* base = dwp + (size_t)first * stride;
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < elems; i += (size_t)gridDim.x * blockDim.x * 4) {
    if (i + 4 <= elems) {
      f32x4 a = *(const f32x4*)(base + i);
#pragma unroll 4
      for (int sl = 1; sl < n; ++sl) a += *(const f32x4*)(base + (size_t)sl * stride + i);
      *(f32x4*)(base + i) = a;
    } else {
      for (size_t k = i; k < elems; ++k) {
        float a = base[k];
        for (int sl = 1; sl < n; ++sl) a += base[(size_t)sl * stride + k];
        base[k] = a;
      }
    }
  }
}

// the same fold for SEVERAL weight tensors in one launch (blockIdx.z = job): the per-layer reductions are tiny (a few MB each,
// 69 of them per YOLOX-s step at ~12 us apiece, mostly launch floor); batched over the layers whose weight gradients finished
// since the last flush they run as two fat launches.  stage 0: every job's slab groups; stage 1: the group heads.
__global__ void reduce_slabs_multi_kernel(const plyolo_reduce_job* jobs, int stage) {
  const plyolo_reduce_job j = jobs[blockIdx.z];
  const int per = stage == 0 ? j.per : j.groups;
  const int nsl = stage == 0 ? j.nslab : j.groups;
  const size_t stride = stage == 0 ? j.elems : j.elems * (size_t)j.per;
  if (stage == 1 && j.groups <= 1) return;
  const int first = blockIdx.y * per;
  if (first >= nsl) return;
  const int n = nsl - first < per ? nsl - first : per;
  if (n <= 1 && stage == 0) return;
  float* base = j.dwp + (size_t)first * stride;
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < j.elems; i += (size_t)gridDim.x * blockDim.x * 4) {
    f32x4 a = *(const f32x4*)(base + i);
#pragma unroll 4
    for (int sl = 1; sl < n; ++sl) a += *(const f32x4*)(base + (size_t)sl * stride + i);
    *(f32x4*)(base + i) = a;
  }
}

__global__ void unpack_wgrads_kernel(const plyolo_pack_entry* table, int accumulate, int n_flat) {
  int ent, slice, nslice;
  pack_block(table, n_flat, &ent, &slice, &nslice);
  const plyolo_pack_entry e = table[ent];
  if (nslice < 0) { slice = (int)blockIdx.x - e.blk0; nslice = e.nblk; }
  if (!e.dw) return;
  const int taps = e.ksize * e.ksize;
  // one thread per (co, ci): the slab reads are coalesced along ci for every tap, and the thread's taps are
  // CONSECUTIVE floats of the OIHW gradient, so neighbouring threads store one contiguous run (a thread per
  // packed element stored 4 bytes every 36)
  const size_t plane = (size_t)e.Cout_total * e.Cin_p;
  const size_t slab = (size_t)taps * plane;
  const int n2 = e.Cout * e.Cin;
  for (int idx = slice * blockDim.x + threadIdx.x; idx < n2; idx += nslice * blockDim.x) {
    const int ci = idx % e.Cin, co = idx / e.Cin;
    const float* src = e.dwp + (size_t)(e.co_off + co) * e.Cin_p + ci;
    float* dst = e.dw + (size_t)idx * taps;
    if (e.nslab == 1 && taps == 9) {
      // the common case (slabs already folded, 3x3): the nine taps -- and the nine old values when accumulating -- are requested
      // together; the rolled loop below waited for every tap's round trip in turn (9 of them per thread: most of this launch)
      float g[9], o[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) g[t] = src[(size_t)t * plane];
#pragma unroll
      for (int t = 0; t < 9; ++t) o[t] = accumulate ? dst[t] : 0.f;
#pragma unroll
      for (int t = 0; t < 9; ++t) dst[t] = o[t] + (0.f + g[t]);
      continue;
    }
    for (int t = 0; t < taps; ++t) {
      float g = 0.f;
      for (int sl = 0; sl < e.nslab; ++sl) g += src[(size_t)sl * slab + (size_t)t * plane];  // fixed order: deterministic
      dst[t] = (accumulate ? dst[t] : 0.f) + g;
    }
  }
  if (e.db && slice == 0)
    for (int i = threadIdx.x; i < e.Cout; i += blockDim.x) e.db[i] = (accumulate ? e.db[i] : 0.f) + e.dbp[e.co_off + i];
}

// dbias[c] += sum over rows; block = (row slices) x (channels), one atomic per channel per block
template <typename T>
__global__ void bias_grad_kernel(const T* dy, int M, int C, int ld, float* db) {
  __shared__ float red[256];
  const int cols = C < 256 ? C : 256, rg = 256 / cols;
  const int tcol = threadIdx.x % cols, trow = threadIdx.x / cols;
  for (int c0 = 0; c0 < C; c0 += cols) {
    const int c = c0 + tcol;
    float s = 0.f;
    if (trow < rg && c < C)
      for (int m = blockIdx.x * rg + trow; m < M; m += gridDim.x * rg) s += ActT<T>::ld(dy + (size_t)m * ld + c);
    __syncthreads();
    red[threadIdx.x] = s;
    __syncthreads();
    if (trow == 0 && c < C) {
      for (int k = 1; k < rg; ++k) s += red[k * cols + tcol];
      atomicAdd(db + c, s);
    }
  }
}

// same sums with 16-byte row vectors (column-fixed threads); needs ld % V == 0 and the padded columns inside the row
template <typename T>
__global__ __launch_bounds__(256) void bias_grad_vec_kernel(const T* __restrict__ dy, int M, int C, int ld, float* db) {
  constexpr int V = Vec<T>::N;
  __shared__ float red[256 * V];
  const int cvn = (C + V - 1) / V;
  const int cols = cvn < 256 ? cvn : 256, rg = 256 / cols;
  const int tcol = threadIdx.x % cols, trow = threadIdx.x / cols;
  for (int cv0 = 0; cv0 < cvn; cv0 += cols) {
    const int cv = cv0 + tcol;
    float s[V];
#pragma unroll
    for (int i = 0; i < V; ++i) s[i] = 0.f;
    if (trow < rg && cv < cvn) {
#pragma unroll 4
      for (int m = blockIdx.x * rg + trow; m < M; m += gridDim.x * rg) {
        float f[V];
        Vec<T>::load(dy + (size_t)m * ld + cv * V, f);
#pragma unroll
        for (int i = 0; i < V; ++i) s[i] += f[i];
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < V; ++i) red[threadIdx.x * V + i] = s[i];
    __syncthreads();
    // column sums by cols*V threads in parallel (one thread summing all row groups serialised up to 2040 LDS reads)
    for (int j = threadIdx.x; j < cols * V; j += 256) {
      const int col = j / V, i = j - col * V, c = (cv0 + col) * V + i;
      if (cv0 + col < cvn && c < C) {
        float t = 0.f;
        for (int k = 0; k < rg; ++k) t += red[(k * cols + col) * V + i];
        atomicAdd(db + c, t);
      }
    }
  }
}

// several bias gradients in one launch (the six prediction convs of a DecoupledHead: 2 launches instead of 12)
struct BiasJobs {
  plyolo_bias_job j[PLYOLO_BIAS_JOBS_MAX];
  int n;
};
__global__ void bias_zero_multi_kernel(const BiasJobs jobs) {
  const plyolo_bias_job j = jobs.j[blockIdx.x];
  for (int i = threadIdx.x; i < j.C; i += blockDim.x) j.db[i] = 0.f;
}
template <typename T>
__global__ __launch_bounds__(256) void bias_grad_vec_multi_kernel(const BiasJobs jobs) {
  constexpr int V = Vec<T>::N;
  const plyolo_bias_job jb = jobs.j[blockIdx.y];
  const T* __restrict__ dy = (const T*)jb.dy;
  const int M = jb.M, C = jb.C, ld = jb.ld;
  float* db = jb.db;
  const int nblk = jb.nblk;                 // workgroups of THIS job (the launch has the maximum over the jobs)
  if ((int)blockIdx.x >= nblk) return;
  __shared__ float red[256 * V];
  const int cvn = (C + V - 1) / V;
  const int cols = cvn < 256 ? cvn : 256, rg = 256 / cols;
  const int tcol = threadIdx.x % cols, trow = threadIdx.x / cols;
  for (int cv0 = 0; cv0 < cvn; cv0 += cols) {
    const int cv = cv0 + tcol;
    float s[V];
#pragma unroll
    for (int i = 0; i < V; ++i) s[i] = 0.f;
    if (trow < rg && cv < cvn) {
#pragma unroll 4
      for (int m = blockIdx.x * rg + trow; m < M; m += nblk * rg) {
        float f[V];
        Vec<T>::load(dy + (size_t)m * ld + cv * V, f);
#pragma unroll
        for (int i = 0; i < V; ++i) s[i] += f[i];
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < V; ++i) red[threadIdx.x * V + i] = s[i];
    __syncthreads();
    for (int j = threadIdx.x; j < cols * V; j += 256) {
      const int col = j / V, i = j - col * V, c = (cv0 + col) * V + i;
      if (cv0 + col < cvn && c < C) {
        float t = 0.f;
        for (int k = 0; k < rg; ++k) t += red[(k * cols + col) * V + i];
        atomicAdd(db + c, t);
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------ optimizer
__global__ void sgd_kernel(float* p, const float* g, float* mom, size_t n, const float* lr_dev, float lr, float momentum,
                           int first) {
  const float rate = lr_dev ? *lr_dev : lr;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float b = g[i];
    if (momentum != 0.f) {
      if (!first) b = momentum * mom[i] + b;
      mom[i] = b;
    }
    p[i] -= rate * b;
  }
}
__global__ void ema_kernel(float* ema, const float* model, size_t n, float d) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float v = ema[i] * d;
    v += (1.0f - d) * model[i];
    ema[i] = v;
  }
}

inline unsigned grid_for(size_t work) {
  size_t b = (work + 255) / 256;
  if (b > 2048) b = 2048;  // 256 CUs x 8 blocks, grid-stride the rest
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

using plyolo::submit;

#define DISPATCH_T(dtype, ...)                       \
  if ((dtype) == PLYOLO_BF16) { typedef bf16_t T; __VA_ARGS__ } \
  else { typedef float T; __VA_ARGS__ }

extern "C" {

int plyolo_focus_s2d(int dtype, const float* img, int N, int H, int W, void* out, int Cp, void* stream) {
  plyolo::annotate("focus_s2d", 0.0, (double)N * H * W * 3 * 4.0 + (double)N * H * W / 4 * Cp * 2.0);
  PLY_CHECK_ARG(H % 2 == 0 && W % 2 == 0 && Cp >= 12, "focus: H,W must be even and Cp >= 12");
  const size_t work = (size_t)N * (H / 2) * (W / 2);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    DISPATCH_T(dtype, hipLaunchKernelGGL(focus_kernel<T>, dim3(grid_for(work)), dim3(256), 0, s, img, N, H, W, (T*)out, Cp);)
    return hipGetLastError();
  });
}

int plyolo_copy_add(int dtype, int M, int C, const void* in, int i_ld, void* out, int o_ld, int accumulate, void* stream) {
  plyolo::annotate("copy_add", 0.0, (double)M * C * (dtype == PLYOLO_BF16 ? 2.0 : 4.0) * (accumulate ? 3.0 : 2.0));
  const int V = dtype == PLYOLO_BF16 ? 8 : 4;
  PLY_CHECK_ARG(C % V == 0 && i_ld % V == 0 && o_ld % V == 0, "copy_add: C/ld must be multiples of %d", V);
  const size_t work = (size_t)M * (C / V);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    DISPATCH_T(dtype, hipLaunchKernelGGL(copy_add_kernel<T>, dim3(grid_for(work)), dim3(256), 0, s, (size_t)M, C, (const T*)in, i_ld,
                                         (T*)out, o_ld, accumulate);)
    return hipGetLastError();
  });
}

int plyolo_upsample2x_fwd(int dtype, int N, int H, int W, int C, const void* in, int i_ld, void* out, int o_ld, void* stream) {
  plyolo::annotate("upsample2x_fwd", 0.0, (double)N * H * W * C * (dtype == PLYOLO_BF16 ? 2.0 : 4.0) * 5.0);
  const int V = dtype == PLYOLO_BF16 ? 8 : 4;
  PLY_CHECK_ARG(C % V == 0 && i_ld % V == 0 && o_ld % V == 0, "upsample: C/ld must be multiples of %d", V);
  const size_t work = (size_t)N * 4 * H * W * (C / V);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    DISPATCH_T(dtype, hipLaunchKernelGGL(upsample2x_fwd_kernel<T>, dim3(grid_for(work)), dim3(256), 0, s, N, H, W, C, (const T*)in,
                                         i_ld, (T*)out, o_ld);)
    return hipGetLastError();
  });
}

int plyolo_upsample2x_bwd(int dtype, int N, int H, int W, int C, const void* dout, int d_ld, void* din, int i_ld, int accumulate,
                          void* stream) {
  plyolo::annotate("upsample2x_bwd", 0.0, (double)N * H * W * C * (dtype == PLYOLO_BF16 ? 2.0 : 4.0) * 5.0);
  const int V = dtype == PLYOLO_BF16 ? 8 : 4;
  PLY_CHECK_ARG(C % V == 0 && i_ld % V == 0 && d_ld % V == 0, "upsample: C/ld must be multiples of %d", V);
  const size_t work = (size_t)N * H * W * (C / V);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    DISPATCH_T(dtype, hipLaunchKernelGGL(upsample2x_bwd_kernel<T>, dim3(grid_for(work)), dim3(256), 0, s, N, H, W, C, (const T*)dout,
                                         d_ld, (T*)din, i_ld, accumulate);)
    return hipGetLastError();
  });
}

int plyolo_maxpool_s1_fwd(int dtype, int N, int H, int W, int C, int k, const void* in, int i_ld, void* out, int o_ld, void* stream) {
  plyolo::annotate("maxpool_s1_fwd", 0.0, (double)N * H * W * C * (dtype == PLYOLO_BF16 ? 2.0 : 4.0) * 2.0);
  const int V = dtype == PLYOLO_BF16 ? 8 : 4;
  PLY_CHECK_ARG(C % V == 0 && i_ld % V == 0 && o_ld % V == 0 && (k & 1), "maxpool: C/ld must be multiples of %d, k odd", V);
  const size_t work = (size_t)N * H * W * (C / V);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    DISPATCH_T(dtype, hipLaunchKernelGGL(maxpool_fwd_kernel<T>, dim3(grid_for(work)), dim3(256), 0, s, N, H, W, C, k, (const T*)in, i_ld,
                                         (T*)out, o_ld);)
    return hipGetLastError();
  });
}

int plyolo_maxpool_s1_bwd(int dtype, int N, int H, int W, int C, int k, const void* in, int i_ld, const void* dout, int d_ld,
                          float* din_f32, void* stream) {
  plyolo::annotate("maxpool_s1_bwd", 0.0, (double)N * H * W * C * (dtype == PLYOLO_BF16 ? 2.0 : 4.0) * 3.0);
  const int V = dtype == PLYOLO_BF16 ? 8 : 4;
  PLY_CHECK_ARG(C % V == 0 && i_ld % V == 0 && d_ld % V == 0 && (k & 1), "maxpool: C/ld must be multiples of %d, k odd", V);
  const size_t work = (size_t)N * H * W * (C / V);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    DISPATCH_T(dtype, hipLaunchKernelGGL(maxpool_bwd_kernel<T>, dim3(grid_for(work)), dim3(256), 0, s, N, H, W, C, k, (const T*)in, i_ld,
                                         (const T*)dout, d_ld, din_f32);)
    return hipGetLastError();
  });
}

static size_t spp_bwd_lds(int dtype, int H, int W, int cg) {
  const size_t es = dtype == PLYOLO_BF16 ? 2 : 4;
  return (size_t)H * W * cg * (8 + 2 * es + 2);
}
// channels per workgroup plane: 8 while the plane fits in LDS, else 4; 0 = does not fit at all
static int spp_bwd_cg(int dtype, int H, int W) {
  if (H * W > 65535) return 0;
  if (spp_bwd_lds(dtype, H, W, 8) <= 150 * 1024) return 8;
  if (spp_bwd_lds(dtype, H, W, 4) <= 150 * 1024) return 4;
  return 0;
}

int plyolo_spp_pools_bwd_fits(int dtype, int H, int W) { return spp_bwd_cg(dtype, H, W) ? 1 : 0; }

// forward cascade in one launch: two planes of [H*W][CG] elements in LDS (CG = one 16-byte vector of channels)
int plyolo_spp_pools_fwd_fits(int dtype, int H, int W, int C, int nk, const int* ks) {
  if (nk < 1 || nk > 3 || !ks) return 0;
  const int cg = dtype == PLYOLO_BF16 ? 8 : 4, esz = dtype == PLYOLO_BF16 ? 2 : 4;
  if (C % cg != 0 || (size_t)2 * H * W * cg * esz > 64 * 1024) return 0;
  int prev = 1;
  for (int j = 0; j < nk; ++j) {     // a cascade: odd, increasing windows
    if (!(ks[j] & 1) || ks[j] < prev) return 0;
    prev = ks[j];
  }
  return 1;
}
int plyolo_spp_pools_fwd(int dtype, int N, int H, int W, int C, int nk, const int* ks, const void* in, int i_ld, void* const* outs,
                         const int* o_lds, void* stream) {
  PLY_CHECK_ARG(in && outs && o_lds && plyolo_spp_pools_fwd_fits(dtype, H, W, C, nk, ks), "spp_pools_fwd: not a cascade of 1..3 odd windows on a plane that fits in LDS (use plyolo_maxpool_s1_fwd)");
  const int V = dtype == PLYOLO_BF16 ? 8 : 4;
  int w[3] = {1, 1, 1}, ol[3] = {0, 0, 0};
  void* o[3] = {nullptr, nullptr, nullptr};
  int prev = 1;
  for (int j = 0; j < nk; ++j) {
    PLY_CHECK_ARG(outs[j] && o_lds[j] % V == 0 && i_ld % V == 0, "spp_pools_fwd: pitches must be multiples of %d", V);
    w[j] = ks[j] - prev + 1;       // pool_k(x) == pool_(k - prev + 1)(pool_prev(x)) for stride-1 max pools
    prev = ks[j];
    o[j] = outs[j]; ol[j] = o_lds[j];
  }
  const size_t lds = (size_t)2 * H * W * V * (dtype == PLYOLO_BF16 ? 2 : 4);
  plyolo::annotate("spp_pools_fwd", 0.0, (double)N * H * W * C * (dtype == PLYOLO_BF16 ? 2.0 : 4.0) * (1.0 + nk));
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    DISPATCH_T(dtype, {
      auto kern = spp_pools_fwd_kernel<T, Vec<T>::N>;
      if (hipError_t e = plyolo::ensure_dynamic_lds((const void*)kern, lds); e != hipSuccess) return e;
      hipLaunchKernelGGL(kern, dim3(N, C / V), dim3(256), lds, s, H, W, C, nk, w[0], w[1], w[2], (const T*)in, i_ld, (T*)o[0], (T*)o[1], (T*)o[2],
                         ol[0], ol[1], ol[2]);
    })
    return hipGetLastError();
  });
}

int plyolo_spp_pools_bwd(int dtype, int N, int H, int W, int C, int nk, const int* ks, const void* in, int i_ld,
                         const void* const* douts, const int* d_lds, void* din, int di_ld, int accumulate, void* stream) {
  PLY_CHECK_ARG(nk >= 1 && nk <= 3 && ks && douts && d_lds, "spp_pools_bwd: 1..3 pools");
  PLY_CHECK_ARG(plyolo_spp_pools_bwd_fits(dtype, H, W), "spp_pools_bwd: %dx%d plane does not fit in LDS (use plyolo_maxpool_s1_bwd)", H, W);
  int k[3] = {1, 1, 1}, dl[3] = {0, 0, 0};
  const void* d[3] = {nullptr, nullptr, nullptr};
  for (int j = 0; j < nk; ++j) {
    PLY_CHECK_ARG(ks[j] & 1, "spp_pools_bwd: k must be odd");
    k[j] = ks[j]; d[j] = douts[j]; dl[j] = d_lds[j];
  }
  const int cg = spp_bwd_cg(dtype, H, W);
  const size_t lds = spp_bwd_lds(dtype, H, W, cg);
  PLY_CHECK_ARG(cg == 8 || C % 4 == 0, "spp_pools_bwd: C must be a multiple of 4");
  plyolo::annotate("spp_pools_bwd", 0.0, (double)N * H * W * C * (dtype == PLYOLO_BF16 ? 2.0 : 4.0) * (2.0 + nk));
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    DISPATCH_T(dtype, {
      auto kern = cg == 8 ? spp_pools_bwd_kernel<T, 8> : spp_pools_bwd_kernel<T, 4>;
      if (hipError_t e = plyolo::ensure_dynamic_lds((const void*)kern, lds); e != hipSuccess) return e;
      hipLaunchKernelGGL(kern, dim3(N, cdiv(C, cg)), dim3(256), lds, s, H, W, C, nk, k[0], k[1], k[2], (const T*)in, i_ld,
                         (const T*)d[0], (const T*)d[1], (const T*)d[2], dl[0], dl[1], dl[2], (T*)din, di_ld, accumulate);
    })
    return hipGetLastError();
  });
}

int plyolo_maxpool2x2_fwd(int dtype, int N, int H, int W, int C, const void* in, int i_ld, void* out, int o_ld, void* stream) {
  const int V = dtype == PLYOLO_BF16 ? 8 : 4;
  PLY_CHECK_ARG(C % V == 0 && i_ld % V == 0 && o_ld % V == 0 && H % 2 == 0 && W % 2 == 0, "maxpool2x2: C/ld multiples of %d, even H/W", V);
  const size_t work = (size_t)N * (H / 2) * (W / 2) * (C / V);
  plyolo::annotate("maxpool2x2_fwd", 0.0, (double)N * H * W * C * (dtype == PLYOLO_BF16 ? 2.0 : 4.0) * 1.25);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    DISPATCH_T(dtype, hipLaunchKernelGGL(maxpool2x2_fwd_kernel<T>, dim3(grid_for(work)), dim3(256), 0, s, N, H, W, C, (const T*)in, i_ld,
                                         (T*)out, o_ld);)
    return hipGetLastError();
  });
}

int plyolo_maxpool2x2_bwd(int dtype, int N, int H, int W, int C, const void* in, int i_ld, const void* dout, int d_ld, void* din,
                          int di_ld, int accumulate, void* stream) {
  const int V = dtype == PLYOLO_BF16 ? 8 : 4;
  PLY_CHECK_ARG(C % V == 0 && i_ld % V == 0 && d_ld % V == 0 && di_ld % V == 0 && H % 2 == 0 && W % 2 == 0, "maxpool2x2: C/ld multiples of %d, even H/W", V);
  const size_t work = (size_t)N * (H / 2) * (W / 2) * (C / V);
  plyolo::annotate("maxpool2x2_bwd", 0.0, (double)N * H * W * C * (dtype == PLYOLO_BF16 ? 2.0 : 4.0) * 2.25);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    DISPATCH_T(dtype, hipLaunchKernelGGL(maxpool2x2_bwd_kernel<T>, dim3(grid_for(work)), dim3(256), 0, s, N, H, W, C, (const T*)in, i_ld,
                                         (const T*)dout, d_ld, (T*)din, di_ld, accumulate);)
    return hipGetLastError();
  });
}

int plyolo_implicit_bias(const float* W, const float* a, const float* b, float* out, int Cout, int Cin, void* stream) {
  plyolo::annotate("implicit_bias", 0.0, 4.0 * Cout * Cin);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    hipLaunchKernelGGL(implicit_bias_kernel, dim3(Cout), dim3(256), 0, s, W, a, b, out, Cout, Cin);
    return hipGetLastError();
  });
}

int plyolo_scale_channels(const float* u, const float* m, float* y, size_t rows, int C, void* stream) {
  plyolo::annotate("scale_channels", 0.0, 8.0 * rows * C);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    hipLaunchKernelGGL(scale_channels_kernel, dim3(grid_for(rows * C)), dim3(256), 0, s, u, m, y, rows, C);
    return hipGetLastError();
  });
}

int plyolo_implicit_bwd_blocks(size_t rows) { return rows < 512 ? (int)(rows < 1 ? 1 : rows) : 512; }

int plyolo_implicit_bwd(int dtype, const float* dy, const float* u, const float* m, void* du, int du_ld, float* partial, size_t rows,
                        int C, void* stream) {
  const int nblk = plyolo_implicit_bwd_blocks(rows);
  plyolo::annotate("implicit_bwd", 0.0, 10.0 * rows * C);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    DISPATCH_T(dtype, hipLaunchKernelGGL(implicit_bwd_kernel<T>, dim3(nblk), dim3(256), 0, s, dy, u, m, (T*)du, du_ld, partial, rows, C);)
    return hipGetLastError();
  });
}

int plyolo_implicit_param_grads(const float* partial, int nblk, const float* W, const float* a, const float* sdu, float* dm, float* da,
                                float* dW, float* db, int Cout, int Cin, void* stream) {
  plyolo::annotate("implicit_param_grads", 0.0, 8.0 * Cout * Cin);
  const int n = Cout > Cin ? Cout : Cin;
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    hipLaunchKernelGGL(implicit_param_grads_kernel, dim3(cdiv(n, 256) > 64 ? cdiv(n, 256) : 64), dim3(256), 0, s, partial, nblk, W, a, sdu, dm, da,
                       dW, db, Cout, Cin);
    return hipGetLastError();
  });
}

int plyolo_f32_to_act(int dtype, int M, int C, const float* in, void* out, int o_ld, int accumulate, void* stream) {
  plyolo::annotate("f32_to_act", 0.0, (double)M * C * 6.0);
  const size_t work = (size_t)M * C;
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    DISPATCH_T(dtype, hipLaunchKernelGGL(f32_to_act_kernel<T>, dim3(grid_for(work)), dim3(256), 0, s, (size_t)M, C, in, (T*)out, o_ld,
                                         accumulate);)
    return hipGetLastError();
  });
}

int plyolo_memset_async(void* p, int value, size_t bytes, void* stream) {
  plyolo::annotate("memset", 0.0, (double)bytes);
  return submit(stream, [=](hipStream_t s) -> hipError_t { return plyolo::fill_async(p, value, bytes, s); });
}

int plyolo_nhwc_to_nchw_f32(int dtype, int N, int H, int W, int C, const void* in, int i_ld, float* out, void* stream) {
  const size_t work = (size_t)N * H * W * C;
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    DISPATCH_T(dtype, hipLaunchKernelGGL(nhwc_to_nchw_kernel<T>, dim3(grid_for(work)), dim3(256), 0, s, N, H, W, C, (const T*)in, i_ld, out);)
    return hipGetLastError();
  });
}

int plyolo_nchw_f32_to_nhwc(int dtype, int N, int H, int W, int C, const float* in, void* out, int o_ld, void* stream) {
  const size_t work = (size_t)N * H * W * C;
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    DISPATCH_T(dtype, hipLaunchKernelGGL(nchw_to_nhwc_kernel<T>, dim3(grid_for(work)), dim3(256), 0, s, N, H, W, C, in, (T*)out, o_ld);)
    return hipGetLastError();
  });
}

int plyolo_pack_weights(const plyolo_pack_entry* table_dev, int n, int dtype, int max_elems, void* stream) {
  plyolo::annotate("pack_weights", 0.0, 0.0);
  int gy = cdiv(max_elems, 256 * 8);
  if (gy < 1) gy = 1;
  if (gy > 64) gy = 64;
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    DISPATCH_T(dtype, hipLaunchKernelGGL(pack_weights_kernel<T>, dim3(n, gy), dim3(256), 0, s, table_dev, 0);)
    return hipGetLastError();
  });
}

int plyolo_reduce_slabs(float* dwp, int nslab, size_t elems, void* stream) {
  PLY_CHECK_ARG(dwp && nslab >= 1 && elems % 4 == 0, "reduce_slabs: slab length must be a multiple of 4 floats");
  if (nslab == 1) return 0;
  plyolo::annotate("reduce_slabs", 0.0, 4.0 * (double)elems * nslab);
  // column blocks alone fill the chip only for the largest weight tensors: split the slab range into groups until
  // about 1024 workgroups exist, at least 4 slabs per group
  const int cols = grid_for(elems / 4);
  int groups = 1;
  if (cols < 512 && nslab >= 16) {
    groups = (1024 + cols - 1) / cols;
    if (groups > nslab / 4) groups = nslab / 4;
  }
  const int per = (nslab + groups - 1) / groups;
  groups = (nslab + per - 1) / per;
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(cols, groups), dim3(256), 0, s, dwp, nslab, elems, per, elems);
    if (groups > 1)  // fold the group heads (slabs 0, per, 2*per, ...)
      hipLaunchKernelGGL(reduce_slabs_kernel, dim3(cols, 1), dim3(256), 0, s, dwp, groups, elems, groups, elems * (size_t)per);
    return hipGetLastError();
  });
}

/* grouping of one weight tensor's slab fold, identical to plyolo_reduce_slabs (so both give bit-identical sums) */
int plyolo_reduce_slabs_plan(int nslab, size_t elems, int* per, int* groups) {
  PLY_CHECK_ARG(nslab >= 1 && elems % 4 == 0 && per && groups, "reduce_slabs_plan: slab length must be a multiple of 4 floats");
  const int cols = grid_for(elems / 4);
  int g = 1;
  if (cols < 512 && nslab >= 16) {
    g = (1024 + cols - 1) / cols;
    if (g > nslab / 4) g = nslab / 4;
  }
  const int p = (nslab + g - 1) / g;
  *per = p;
  *groups = (nslab + p - 1) / p;
  return 0;
}

int plyolo_reduce_slabs_multi(const plyolo_reduce_job* jobs_dev, int njobs, int max_cols, int max_groups, double total_bytes, void* stream) {
  PLY_CHECK_ARG(jobs_dev && njobs > 0 && max_cols > 0 && max_groups > 0, "reduce_slabs_multi: bad arguments");
  plyolo::annotate("reduce_slabs", 0.0, total_bytes);
  const int cols = max_cols > 256 ? 256 : max_cols;
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    hipLaunchKernelGGL(reduce_slabs_multi_kernel, dim3(cols, max_groups, njobs), dim3(256), 0, s, jobs_dev, 0);
    if (max_groups > 1) hipLaunchKernelGGL(reduce_slabs_multi_kernel, dim3(cols, 1, njobs), dim3(256), 0, s, jobs_dev, 1);
    return hipGetLastError();
  });
}

int plyolo_unpack_wgrads(const plyolo_pack_entry* table_dev, int n, int max_elems, int accumulate, void* stream) {
  plyolo::annotate("unpack_wgrads", 0.0, 0.0);
  int gy = cdiv(max_elems, 256 * 8);
  if (gy < 1) gy = 1;
  if (gy > 64) gy = 64;
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    hipLaunchKernelGGL(unpack_wgrads_kernel, dim3(n, gy), dim3(256), 0, s, table_dev, accumulate, 0);
    return hipGetLastError();
  });
}

// Balanced launches: an entry gets one workgroup per 2048 weights (pack: 8 per thread and pass; a 512x512x3x3 layer gets 1152
// workgroups, a 32-channel one a single one) instead of the same gridDim.y for every entry.
int plyolo_pack_plan(plyolo_pack_entry* t, int n) {
  PLY_CHECK_ARG(t && n > 0, "pack_plan: empty table");
  int total = 0;
  for (int i = 0; i < n; ++i) {
    const long long elems = (long long)t[i].ksize * t[i].ksize * t[i].Cout * t[i].Cin_p;
    long long nb = (elems + 2047) / 2048;
    if (nb < 1) nb = 1;
    if (nb > 4096) nb = 4096;
    t[i].blk0 = total;
    t[i].nblk = (int)nb;
    total += (int)nb;
  }
  return total;
}
int plyolo_pack_weights_flat(const plyolo_pack_entry* table_dev, int n, int dtype, int total_blocks, void* stream) {
  PLY_CHECK_ARG(table_dev && n > 0 && total_blocks >= n, "pack_weights_flat: run plyolo_pack_plan on the host table first");
  plyolo::annotate("pack_weights", 0.0, 0.0);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    DISPATCH_T(dtype, hipLaunchKernelGGL(pack_weights_kernel<T>, dim3(total_blocks), dim3(256), 0, s, table_dev, n);)
    return hipGetLastError();
  });
}
int plyolo_unpack_wgrads_flat(const plyolo_pack_entry* table_dev, int n, int total_blocks, int accumulate, void* stream) {
  PLY_CHECK_ARG(table_dev && n > 0 && total_blocks >= n, "unpack_wgrads_flat: run plyolo_pack_plan on the host table first");
  plyolo::annotate("unpack_wgrads", 0.0, 0.0);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    hipLaunchKernelGGL(unpack_wgrads_kernel, dim3(total_blocks), dim3(256), 0, s, table_dev, accumulate, n);
    return hipGetLastError();
  });
}

int plyolo_bias_grad(int dtype, const void* dy, int M, int C, int ld, float* dbias, void* stream) {
  plyolo::annotate("bias_grad", 0.0, (double)M * C * 2.0);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    hipError_t e = plyolo::fill_async(dbias, 0, (size_t)C * 4, s);
    if (e != hipSuccess) return e;
    const int V = dtype == PLYOLO_BF16 ? 8 : 4;
    const int cvn = (C + V - 1) / V;
    if (ld % V == 0 && cvn * V <= ld && ((uintptr_t)dy & 15) == 0) {
      // few, fat workgroups: every workgroup ends with C same-address atomics
      long nb = (long)M * cvn / 2048;
      nb = nb < 1 ? 1 : (nb > 512 ? 512 : nb);
      DISPATCH_T(dtype, hipLaunchKernelGGL(bias_grad_vec_kernel<T>, dim3((unsigned)nb), dim3(256), 0, s, (const T*)dy, M, C, ld, dbias);)
      return hipGetLastError();
    }
    int nb = M / 64;
    if (nb < 1) nb = 1;
    if (nb > 1024) nb = 1024;
    DISPATCH_T(dtype, hipLaunchKernelGGL(bias_grad_kernel<T>, dim3(nb), dim3(256), 0, s, (const T*)dy, M, C, ld, dbias);)
    return hipGetLastError();
  });
}

int plyolo_bias_grad_multi(int dtype, const plyolo_bias_job* jobs, int njobs, void* stream) {
  PLY_CHECK_ARG(jobs && njobs >= 1 && njobs <= PLYOLO_BIAS_JOBS_MAX, "bias_grad_multi: 1..%d jobs", PLYOLO_BIAS_JOBS_MAX);
  const int V = dtype == PLYOLO_BF16 ? 8 : 4;
  BiasJobs bj{};
  bj.n = njobs;
  int nbmax = 1;
  double bytes = 0.0;
  for (int i = 0; i < njobs; ++i) {
    plyolo_bias_job j = jobs[i];
    const int cvn = (j.C + V - 1) / V;
    PLY_CHECK_ARG(j.dy && j.db && j.M > 0 && j.C > 0 && j.ld % V == 0 && cvn * V <= j.ld && ((uintptr_t)j.dy & 15) == 0,
                  "bias_grad_multi: job %d needs 16-byte aligned rows that hold C rounded up to %d channels (use plyolo_bias_grad)", i, V);
    long nb = (long)j.M * cvn / 2048;
    nb = nb < 1 ? 1 : (nb > 512 ? 512 : nb);     // the grid plyolo_bias_grad gives the same matrix
    j.nblk = (int)nb;
    nbmax = nb > nbmax ? (int)nb : nbmax;
    bytes += (double)j.M * j.C * (dtype == PLYOLO_BF16 ? 2.0 : 4.0);
    bj.j[i] = j;
  }
  plyolo::annotate("bias_grad", 0.0, bytes);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    hipLaunchKernelGGL(bias_zero_multi_kernel, dim3(njobs), dim3(128), 0, s, bj);
    DISPATCH_T(dtype, hipLaunchKernelGGL(bias_grad_vec_multi_kernel<T>, dim3(nbmax, njobs), dim3(256), 0, s, bj);)
    return hipGetLastError();
  });
}

int plyolo_sgd_momentum(float* p, const float* g, float* mom, size_t n, const float* lr_dev, float lr, float momentum, int first_step,
                        void* stream) {
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n)), dim3(256), 0, s, p, g, mom, n, lr_dev, lr, momentum, first_step);
    return hipGetLastError();
  });
}

int plyolo_ema_update(float* ema, const float* model, size_t n, float decay, void* stream) {
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    hipLaunchKernelGGL(ema_kernel, dim3(grid_for(n)), dim3(256), 0, s, ema, model, n, decay);
    return hipGetLastError();
  });
}

}  // extern "C"
