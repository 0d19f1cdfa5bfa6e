// HBM-bound kernels around the convolutions: BatchNorm statistics/normalise +
// activation (fwd and bwd), Focus space-to-depth, concat-slice copies, nearest
// upsample, SPP max-pools, weight (re)packing, SGD/EMA.  All activations are NHWC
// with a row pitch (`ld`), every thread moves whole 16-byte vectors.
//
// Reference semantics: BaseConv = act(bn(conv(x))) models/layers/network_blocks.py:30-37,
// BatchNorm2d(eps 1e-3, momentum 0.03) models/layers/normalization.py:8, Focus
// network_blocks.py:50-65, SPP pools :144, nn.Upsample(nearest x2) necks/pafpn_csp.py:22.
#include "common.h"

namespace {

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

// ------------------------------------------------------------------ BatchNorm
__global__ void bn_finalize_kernel(const float* stats, int rows, int C, double count, const float* gamma,
                                   const float* beta, float eps, float momentum, float* rmean, float* rvar,
                                   int64_t* nbt, float* coef) {
  // 32 channels x 8 row-slices per block
  __shared__ double red[2][8][32];
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  double s = 0.0, ss = 0.0;
  if (c < C)
    for (int r = sl; r < rows; r += 8) {
      s += stats[(size_t)r * C + c];
      ss += stats[((size_t)rows + r) * C + c];
    }
  red[0][sl][cl] = s;
  red[1][sl][cl] = ss;
  __syncthreads();
  if (sl == 0 && c < C) {
    for (int k = 1; k < 8; ++k) { s += red[0][k][cl]; ss += red[1][k][cl]; }
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
  if (nbt && blockIdx.x == 0 && threadIdx.x == 0) *nbt += 1;
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

template <typename T>
__global__ void bn_act_fwd_kernel(size_t M, int C, const T* z, int z_ld, const float* coef, int act, const T* res,
                                  int r_ld, T* out, int o_ld) {
  constexpr int V = Vec<T>::N;
  const int cvn = C / V;
  const size_t total = M * cvn;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const size_t m = idx / cvn;
    const int c = (int)(idx - m * cvn) * V;
    float f[V], r[V];
    Vec<T>::load(z + m * z_ld + c, f);
    if (res) Vec<T>::load(res + m * r_ld + c, r);
#pragma unroll
    for (int i = 0; i < V; ++i) {
      float u = coef ? fmaf(f[i], coef[c + i], coef[C + c + i]) : f[i];
      u = actf<Vec<T>::precise>(u, act);
      f[i] = res ? u + r[i] : u;
    }
    Vec<T>::store(out + m * o_ld + c, f);
  }
}

// partial[0][row][c] = sum du ; partial[1][row][c] = sum du * zhat
template <typename T>
__global__ void bn_act_bwd_reduce_kernel(size_t M, int C, const T* dout, int d_ld, const T* z, int z_ld, const float* coef,
                                         int act, float* partial, int rows) {
  constexpr int V = Vec<T>::N;
  __shared__ float red[256 * 2 * V];
  const int cvn = C / V;
  const int cols = cvn < 256 ? cvn : 256;
  const int rg = 256 / cols;
  const int tcol = threadIdx.x % cols, trow = threadIdx.x / cols;
  const int row = blockIdx.x;
  const size_t chunk = (M + rows - 1) / rows;
  const size_t m0 = (size_t)row * chunk, m1 = m0 + chunk < M ? m0 + chunk : M;
  for (int cv0 = 0; cv0 < cvn; cv0 += cols) {
    const int cv = cv0 + tcol;
    float s1[V], s2[V];
#pragma unroll
    for (int i = 0; i < V; ++i) s1[i] = s2[i] = 0.f;
    if (trow < rg && cv < cvn) {
      const int c = cv * V;
      float sc[V], sh[V], mu[V], is[V];
#pragma unroll
      for (int i = 0; i < V; ++i) { sc[i] = coef[c + i]; sh[i] = coef[C + c + i]; mu[i] = coef[2 * C + c + i]; is[i] = coef[3 * C + c + i]; }
      for (size_t m = m0 + trow; m < m1; m += rg) {
        float d[V], zz[V];
        Vec<T>::load(dout + m * d_ld + c, d);
        Vec<T>::load(z + m * z_ld + c, zz);
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
    if (trow == 0 && cv < cvn) {
      for (int k = 1; k < rg; ++k)
#pragma unroll
        for (int i = 0; i < V; ++i) {
          s1[i] += red[((k * cols + tcol) * 2 + 0) * V + i];
          s2[i] += red[((k * cols + tcol) * 2 + 1) * V + i];
        }
#pragma unroll
      for (int i = 0; i < V; ++i) {
        partial[(size_t)row * C + cv * V + i] = s1[i];
        partial[((size_t)rows + row) * C + cv * V + i] = s2[i];
      }
    }
  }
}

__global__ void bn_bwd_finalize_kernel(const float* partial, int rows, int C, double count, const float* gamma,
                                       const float* coef, float* dgamma, float* dbeta, int accumulate, float* bcoef) {
  __shared__ double red[2][8][32];
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  double s = 0.0, ss = 0.0;
  if (c < C)
    for (int r = sl; r < rows; r += 8) {
      s += partial[(size_t)r * C + c];
      ss += partial[((size_t)rows + r) * C + c];
    }
  red[0][sl][cl] = s;
  red[1][sl][cl] = ss;
  __syncthreads();
  if (sl == 0 && c < C) {
    for (int k = 1; k < 8; ++k) { s += red[0][k][cl]; ss += red[1][k][cl]; }
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
}

template <typename T>
__global__ void bn_act_bwd_dz_kernel(size_t M, int C, const T* dout, int d_ld, const T* z, int z_ld, const float* coef,
                                     const float* bcoef, int act, T* dz, int dz_ld) {
  constexpr int V = Vec<T>::N;
  const int cvn = C / V;
  const size_t total = M * cvn;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const size_t m = idx / cvn;
    const int c = (int)(idx - m * cvn) * V;
    float d[V], zz[V];
    Vec<T>::load(dout + m * d_ld + c, d);
    Vec<T>::load(z + m * z_ld + c, zz);
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const float u = fmaf(zz[i], coef[c + i], coef[C + c + i]);
      const float du = d[i] * act_grad(u, act);
      d[i] = fmaf(bcoef[c + i], du, fmaf(bcoef[C + c + i], zz[i], bcoef[2 * C + c + i]));
    }
    Vec<T>::store(dz + m * dz_ld + c, d);
  }
}

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
template <typename T>
__global__ void pack_weights_kernel(const plyolo_pack_entry* table) {
  const plyolo_pack_entry e = table[blockIdx.x];
  const int taps = e.ksize * e.ksize;
  const int nf = taps * e.Cout * e.Cin_p;
  T* wp = (T*)e.wp;
  T* wpd = (T*)e.wpd;
  for (int idx = blockIdx.y * blockDim.x + threadIdx.x; idx < nf; idx += gridDim.y * blockDim.x) {
    const int ci = idx % e.Cin_p;
    const int co = (idx / e.Cin_p) % e.Cout;
    const int t = idx / (e.Cin_p * e.Cout);
    const float v = ci < e.Cin ? e.w[((size_t)co * e.Cin + ci) * taps + t] : 0.f;
    ActT<T>::st(wp + ((size_t)t * e.Cout_total + e.co_off + co) * e.Cin_p + ci, v);
    if (wpd) ActT<T>::st(wpd + ((size_t)t * e.Cin_p + ci) * e.Cout_p8 + e.co_off + co, v);
  }
  if (e.b && blockIdx.y == 0)
    for (int i = threadIdx.x; i < e.Cout; i += blockDim.x) e.bp[e.co_off + i] = e.b[i];
}

__global__ void unpack_wgrads_kernel(const plyolo_pack_entry* table, int accumulate) {
  const plyolo_pack_entry e = table[blockIdx.x];
  if (!e.dw) return;
  const int taps = e.ksize * e.ksize;
  const int n = e.Cout * e.Cin * taps;
  for (int idx = blockIdx.y * blockDim.x + threadIdx.x; idx < n; idx += gridDim.y * blockDim.x) {
    const int t = idx % taps;
    const int ci = (idx / taps) % e.Cin;
    const int co = idx / (taps * e.Cin);
    const float g = e.dwp[((size_t)t * e.Cout_total + e.co_off + co) * e.Cin_p + ci];
    e.dw[idx] = (accumulate ? e.dw[idx] : 0.f) + g;
  }
  if (e.db && blockIdx.y == 0)
    for (int i = threadIdx.x; i < e.Cout; i += blockDim.x) e.db[i] = (accumulate ? e.db[i] : 0.f) + e.dbp[e.co_off + i];
}

template <typename T>
__global__ void bias_grad_kernel(const T* dy, int M, int C, int ld, float* db) {
  const int c = blockIdx.x;
  __shared__ double red[256];
  double s = 0.0;
  for (int m = threadIdx.x; m < M; m += blockDim.x) s += (double)ActT<T>::ld(dy + (size_t)m * ld + c);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) db[c] = (float)red[0];
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

int plyolo_bn_finalize(const float* stats, int rows, int C, double count, const float* gamma, const float* beta, float eps,
                       float momentum, float* running_mean, float* running_var, int64_t* nbt, float* coef, void* stream) {
  plyolo::annotate("bn_finalize", 0.0, 8.0 * rows * C);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 32)), dim3(256), 0, s, stats, rows, C, count, gamma, beta, eps, momentum,
                       running_mean, running_var, nbt, coef);
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
  plyolo::annotate("bn_act_fwd", 0.0, (double)M * C * (dtype == PLYOLO_BF16 ? 2.0 : 4.0) * (res ? 3.0 : 2.0));
  const int V = dtype == PLYOLO_BF16 ? 8 : 4;
  PLY_CHECK_ARG(C % V == 0 && z_ld % V == 0 && o_ld % V == 0 && (!res || r_ld % V == 0), "bn_act_fwd: C/ld must be multiples of %d", V);
  const size_t work = (size_t)M * (C / V);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    DISPATCH_T(dtype, hipLaunchKernelGGL(bn_act_fwd_kernel<T>, dim3(grid_for(work)), dim3(256), 0, s, (size_t)M, C, (const T*)z, z_ld,
                                         coef, act, (const T*)res, r_ld, (T*)out, o_ld);)
    return hipGetLastError();
  });
}

int plyolo_bn_bwd_rows(int M) {
  int r = M / 128;
  if (r < 1) r = 1;
  if (r > 512) r = 512;
  return r;
}

int plyolo_bn_act_bwd_reduce(int dtype, int M, int C, const void* dout, int d_ld, const void* z, int z_ld, const float* coef,
                             int act, float* partial, void* stream) {
  plyolo::annotate("bn_act_bwd_reduce", 0.0, (double)M * C * (dtype == PLYOLO_BF16 ? 2.0 : 4.0) * 2.0);
  const int V = dtype == PLYOLO_BF16 ? 8 : 4;
  PLY_CHECK_ARG(C % V == 0 && z_ld % V == 0 && d_ld % V == 0, "bn_act_bwd_reduce: C/ld must be multiples of %d", V);
  const int rows = plyolo_bn_bwd_rows(M);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    DISPATCH_T(dtype, hipLaunchKernelGGL(bn_act_bwd_reduce_kernel<T>, dim3(rows), dim3(256), 0, s, (size_t)M, C, (const T*)dout, d_ld,
                                         (const T*)z, z_ld, coef, act, partial, rows);)
    return hipGetLastError();
  });
}

int plyolo_bn_bwd_finalize(const float* partial, int rows, int C, double count, const float* gamma, const float* coef,
                           float* dgamma, float* dbeta, int accumulate, float* bcoef, void* stream) {
  plyolo::annotate("bn_bwd_finalize", 0.0, 8.0 * rows * C);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, 32)), dim3(256), 0, s, partial, rows, C, count, gamma, coef, dgamma, dbeta,
                       accumulate, bcoef);
    return hipGetLastError();
  });
}

int plyolo_bn_act_bwd_dz(int dtype, int M, int C, const void* dout, int d_ld, const void* z, int z_ld, const float* coef,
                         const float* bcoef, int act, void* dz, int dz_ld, void* stream) {
  plyolo::annotate("bn_act_bwd_dz", 0.0, (double)M * C * (dtype == PLYOLO_BF16 ? 2.0 : 4.0) * 3.0);
  const int V = dtype == PLYOLO_BF16 ? 8 : 4;
  PLY_CHECK_ARG(C % V == 0 && z_ld % V == 0 && d_ld % V == 0 && dz_ld % V == 0, "bn_act_bwd_dz: C/ld must be multiples of %d", V);
  const size_t work = (size_t)M * (C / V);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    DISPATCH_T(dtype, hipLaunchKernelGGL(bn_act_bwd_dz_kernel<T>, dim3(grid_for(work)), dim3(256), 0, s, (size_t)M, C, (const T*)dout,
                                         d_ld, (const T*)z, z_ld, coef, bcoef, act, (T*)dz, dz_ld);)
    return hipGetLastError();
  });
}

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
  return submit(stream, [=](hipStream_t s) -> hipError_t { return hipMemsetAsync(p, value, bytes, s); });
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
    DISPATCH_T(dtype, hipLaunchKernelGGL(pack_weights_kernel<T>, dim3(n, gy), dim3(256), 0, s, table_dev);)
    return hipGetLastError();
  });
}

int plyolo_unpack_wgrads(const plyolo_pack_entry* table_dev, int n, int max_elems, int accumulate, void* stream) {
  plyolo::annotate("unpack_wgrads", 0.0, 0.0);
  int gy = cdiv(max_elems, 256 * 8);
  if (gy < 1) gy = 1;
  if (gy > 64) gy = 64;
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    hipLaunchKernelGGL(unpack_wgrads_kernel, dim3(n, gy), dim3(256), 0, s, table_dev, accumulate);
    return hipGetLastError();
  });
}

int plyolo_bias_grad(int dtype, const void* dy, int M, int C, int ld, float* dbias, void* stream) {
  plyolo::annotate("bias_grad", 0.0, (double)M * C * 2.0);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    DISPATCH_T(dtype, hipLaunchKernelGGL(bias_grad_kernel<T>, dim3(C), dim3(256), 0, s, (const T*)dy, M, C, ld, dbias);)
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
