// fp32 convolution kernels (PLYOLO_F32 "parity mode"): plain FMA arithmetic in the
// same NHWC / packed-weight layouts as the MFMA path.  They exist so that the HIP
// graph can be checked against the oracle at fp32 tolerances (losses within 1e-4,
// SURVEY.md section 7 hard part 2) and double as an on-device cross-check of the
// bf16 MFMA kernels.  Reference semantics: nn.Conv2d fwd/bwd as used by BaseConv
// (models/layers/network_blocks.py:18-26) and DecoupledHead (decoupled_head.py:43-62).
#include "common.h"

namespace {

struct RefP {
  const float* x;
  const float* w;
  float* y;
  const float* bias;
  int N, H, W, OH, OW, Cin, Cout, x_ld, y_ld, ks, stride, pad, accumulate, Kc;
  const float* pre;   // lazy input (plyolo_conv_desc::x_coef): x is read as act(x * pre[c] + pre[pre_ld + c])
  int pre_ld, pre_act;
};

DEVINL float ref_x(const float* xr, int ci, const float* pre, int pre_ld, int pre_act) {
  const float v = xr[ci];
  return pre ? act_fwd_precise(fmaf(v, pre[ci], pre[pre_ld + ci]), pre_act) : v;
}

// one thread per (output pixel, co); co fastest so a wave shares the input pixel
__global__ void k_conv_ref_fwd(const RefP p) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)p.N * p.OH * p.OW * p.Cout;
  if (idx >= total) return;
  const int co = (int)(idx % p.Cout);
  size_t m = idx / p.Cout;
  const int ox = (int)(m % p.OW);
  m /= p.OW;
  const int oy = (int)(m % p.OH);
  const int n = (int)(m / p.OH);
  float acc = 0.f;
  for (int kh = 0; kh < p.ks; ++kh) {
    const int iy = oy * p.stride + kh - p.pad;
    if (iy < 0 || iy >= p.H) continue;
    for (int kw = 0; kw < p.ks; ++kw) {
      const int ix = ox * p.stride + kw - p.pad;
      if (ix < 0 || ix >= p.W) continue;
      const float* xr = p.x + ((size_t)(n * p.H + iy) * p.W + ix) * p.x_ld;
      const float* wr = p.w + ((size_t)(kh * p.ks + kw) * p.Cout + co) * p.Cin;
      for (int ci = 0; ci < p.Cin; ++ci) acc = fmaf(ref_x(xr, ci, p.pre, p.pre_ld, p.pre_act), wr[ci], acc);
    }
  }
  if (p.bias) acc += p.bias[co];
  p.y[((size_t)(n * p.OH + oy) * p.OW + ox) * p.y_ld + co] = acc;
}

// dx[n,y,x,ci] (+)= sum dy[n,oh,ow,co] * wd[tap][ci][co] ; one thread per (input pixel, ci)
__global__ void k_conv_ref_dgrad(const RefP p) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)p.N * p.H * p.W * p.Cin;
  if (idx >= total) return;
  const int ci = (int)(idx % p.Cin);
  size_t m = idx / p.Cin;
  const int ix = (int)(m % p.W);
  m /= p.W;
  const int iy = (int)(m % p.H);
  const int n = (int)(m / p.H);
  float acc = 0.f;
  for (int kh = 0; kh < p.ks; ++kh) {
    const int ty = iy + p.pad - kh;
    if (ty < 0 || (ty % p.stride) != 0) continue;
    const int oy = ty / p.stride;
    if (oy >= p.OH) continue;
    for (int kw = 0; kw < p.ks; ++kw) {
      const int tx = ix + p.pad - kw;
      if (tx < 0 || (tx % p.stride) != 0) continue;
      const int ox = tx / p.stride;
      if (ox >= p.OW) continue;
      const float* dyr = p.x + ((size_t)(n * p.OH + oy) * p.OW + ox) * p.x_ld;  // p.x = dy here
      const float* wr = p.w + ((size_t)(kh * p.ks + kw) * p.Cin + ci) * p.Kc;
      for (int co = 0; co < p.Cout; ++co) acc = fmaf(dyr[co], wr[co], acc);
    }
  }
  float* dst = p.y + ((size_t)(n * p.H + iy) * p.W + ix) * p.y_ld + ci;  // p.y = dx here
  if (p.accumulate) acc += *dst;
  *dst = acc;
}

// dwp[tap][co][ci] += sum_pixels dy*x ; thread per (tap,co,ci), pixel range split over gridDim.y
__global__ void k_conv_ref_wgrad(const float* x, const float* dy, float* dwp, int N, int H, int W, int OH, int OW,
                               int Cin, int Cout, int x_ld, int dy_ld, int ks, int stride, int pad, const float* pre, int pre_ld,
                               int pre_act) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = ks * ks * Cout * Cin;
  if (idx >= total) return;
  const int ci = idx % Cin;
  const int co = (idx / Cin) % Cout;
  const int t = idx / (Cin * Cout);
  const int kh = t / ks, kw = t % ks;
  const size_t M = (size_t)N * OH * OW;
  const size_t chunk = (M + gridDim.y - 1) / gridDim.y;
  const size_t m0 = (size_t)blockIdx.y * chunk;
  const size_t m1 = m0 + chunk < M ? m0 + chunk : M;
  double acc = 0.0;
  for (size_t m = m0; m < m1; ++m) {
    const int ox = (int)(m % OW);
    const size_t m2 = m / OW;
    const int oy = (int)(m2 % OH);
    const int n = (int)(m2 / OH);
    const int iy = oy * stride + kh - pad, ix = ox * stride + kw - pad;
    if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
    acc += (double)dy[m * dy_ld + co] * (double)ref_x(x + ((size_t)(n * H + iy) * W + ix) * x_ld, ci, pre, pre_ld, pre_act);
  }
  atomicAdd(dwp + idx, (float)acc);
}

// per-channel sum / sum of squares of y [M][C] (pitch ld), added to the fp64 stat slots (block b -> slot b % NSLOTS)
__global__ void channel_stats_f32(const float* y, size_t M, int C, int ld, double* stats, int rows) {
  const int row = blockIdx.x;
  const size_t chunk = (M + rows - 1) / rows;
  const size_t m0 = (size_t)row * chunk, m1 = m0 + chunk < M ? m0 + chunk : M;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    double s = 0.0, ss = 0.0;
    for (size_t m = m0; m < m1; ++m) {
      const double v = y[m * ld + c];
      s += v;
      ss += v * v;
    }
    double* slot = stats + (size_t)(row % PLYOLO_STAT_SLOTS) * 2 * C;
    __hip_atomic_fetch_add(slot + c, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(slot + C + c, ss, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

}  // namespace

namespace plyolo {

constexpr int REF_STAT_ROWS = 64;

static RefP make(const plyolo_conv_desc* d) {
  RefP p{};
  p.N = d->N; p.H = d->H; p.W = d->W;
  p.ks = d->ksize; p.stride = d->stride; p.pad = (d->ksize - 1) / 2;
  p.OH = (d->H + 2 * p.pad - d->ksize) / d->stride + 1;
  p.OW = (d->W + 2 * p.pad - d->ksize) / d->stride + 1;
  p.Cin = d->Cin; p.Cout = d->Cout; p.x_ld = d->x_ld; p.y_ld = d->y_ld;
  p.Kc = (d->Cout + 7) & ~7;
  p.pre = d->x_coef; p.pre_ld = d->x_coef_ld; p.pre_act = d->x_act;
  return p;
}

int conv_ref_fwd(const plyolo_conv_desc* d, const void* x, const void* wp, const float* bias, void* y, double* stats,
                 void* stream) {
  RefP p = make(d);
  p.x = (const float*)x; p.w = (const float*)wp; p.y = (float*)y; p.bias = bias;
  const size_t total = (size_t)p.N * p.OH * p.OW * p.Cout;
  const size_t M = (size_t)p.N * p.OH * p.OW;
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    hipLaunchKernelGGL(k_conv_ref_fwd, dim3((unsigned)cdivz(total, 256)), dim3(256), 0, s, p);
    if (stats) hipLaunchKernelGGL(channel_stats_f32, dim3(REF_STAT_ROWS), dim3(256), 0, s, p.y, M, p.Cout, p.y_ld, stats, REF_STAT_ROWS);
    return hipGetLastError();
  });
}

int conv_ref_dgrad(const plyolo_conv_desc* d, const void* dy, const void* wpd, void* dx, int accumulate, void* stream) {
  RefP p = make(d);
  p.x = (const float*)dy; p.w = (const float*)wpd; p.y = (float*)dx; p.accumulate = accumulate;
  p.pre = nullptr;   // the data gradient never reads x
  // roles: p.x_ld must be dy's pitch, p.y_ld dx's pitch
  p.x_ld = d->y_ld; p.y_ld = d->x_ld;
  const size_t total = (size_t)p.N * p.H * p.W * p.Cin;
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    hipLaunchKernelGGL(k_conv_ref_dgrad, dim3((unsigned)cdivz(total, 256)), dim3(256), 0, s, p);
    return hipGetLastError();
  });
}

int conv_ref_wgrad(const plyolo_conv_desc* d, const void* x, const void* dy, float* dwp, void* stream) {
  RefP p = make(d);
  const int total = p.ks * p.ks * p.Cout * p.Cin;
  const size_t M = (size_t)p.N * p.OH * p.OW;
  int split = (int)(M / 512);
  if (split < 1) split = 1;
  if (split > 64) split = 64;
  const float* xf = (const float*)x;
  const float* dyf = (const float*)dy;
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    hipLaunchKernelGGL(k_conv_ref_wgrad, dim3(cdiv(total, 128), split), dim3(128), 0, s, xf, dyf, dwp, p.N, p.H, p.W, p.OH, p.OW,
                       p.Cin, p.Cout, p.x_ld, p.y_ld, p.ks, p.stride, p.pad, p.pre, p.pre_ld, p.pre_act);
    return hipGetLastError();
  });
}

}  // namespace plyolo

