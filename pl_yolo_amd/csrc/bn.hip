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
//
// Batch statistics arrive in fp64 STAT SLOTS (include/plyolo.h): the producer (conv epilogue or
// bn_act_bwd_reduce) adds one partial per workgroup and channel with agent-scope fp64 atomics;
// the consumer's workgroups each sum the 8 slots in their prologue (a few KB from L2) and derive
// the per-channel coefficients themselves -- no finalize launch between producer and consumer.
//   bn_act_fwd         out = act(z*scale+shift) (+ residual); strided output = concat;
//                      workgroup 0 also publishes coef (for the backward) + running statistics
//   bn_act_bwd_reduce  per-channel  sum du, sum du*zhat  (du = dout*act'(u))  -> slots
//   bn_act_bwd_dz      dz = A*du + B*z + Cc ; workgroup 0 publishes dgamma / dbeta
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int NSLOT = PLYOLO_STAT_SLOTS;
constexpr int BN_MAXC = 2048;  // channels per launch (LDS: 3 floats per channel)

DEVINL void slot_add(double* p, double v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// sums of the stat slots for channel c
DEVINL void slot_sums(const double* slots, int C, int c, double* s, double* ss) {
  double a = 0.0, b = 0.0;
#pragma unroll
  for (int r = 0; r < NSLOT; ++r) {
    a += slots[((size_t)r * 2 + 0) * C + c];
    b += slots[((size_t)r * 2 + 1) * C + c];
  }
  *s = a;
  *ss = b;
}

// (scale, shift, mean, invstd) of channel c from the slot sums; optionally the running statistics
DEVINL void bn_coef(const plyolo_bn_stats& st, int C, int c, bool publish, float* coef, float* scale_o, float* shift_o) {
  double s, ss;
  slot_sums(st.slots, C, c, &s, &ss);
  const double mean = s / st.count;
  double var = ss / st.count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)st.eps));
  const bool second = st.split > 0 && c >= st.split;
  const int cp = second ? c - st.split : c;
  const float* gam = second ? st.gamma2 : st.gamma;
  const float* bet = second ? st.beta2 : st.beta;
  const float g = gam ? gam[cp] : 1.f, b = bet ? bet[cp] : 0.f;
  const float scale = g * invstd, shift = b - (float)mean * scale;
  *scale_o = scale;
  *shift_o = shift;
  if (publish) {
    coef[c] = scale;
    coef[C + c] = shift;
    coef[2 * C + c] = (float)mean;
    coef[3 * C + c] = invstd;
    const float mom = st.momentum;
    float* rm = second ? st.running_mean2 : st.running_mean;
    float* rv = second ? st.running_var2 : st.running_var;
    if (rm) rm[cp] = (1.f - mom) * rm[cp] + mom * (float)mean;
    if (rv) {
      const double unb = st.count > 1.0 ? var * st.count / (st.count - 1.0) : var;
      rv[cp] = (1.f - mom) * rv[cp] + mom * (float)unb;
    }
    if (c == 0 && st.num_batches_tracked) *st.num_batches_tracked += 1;
    if (second && cp == 0 && st.num_batches_tracked2) *st.num_batches_tracked2 += 1;
  }
}

__global__ void bn_finalize_kernel(plyolo_bn_stats st, int C, float* coef) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float sc, sh;
  bn_coef(st, C, c, true, coef, &sc, &sh);
}

// coef rows have Ct channels; this module's C channels start at column c_off (merged convolutions)
__global__ void bn_eval_coef_kernel(int C, const float* gamma, const float* beta, const float* rmean, const float* rvar,
                                    float eps, float* coef, int Ct, int c_off) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float invstd = 1.0f / sqrtf(rvar[c] + eps);
  const float scale = (gamma ? gamma[c] : 1.f) * invstd;
  coef[c_off + c] = scale;
  coef[Ct + c_off + c] = (beta ? beta[c] : 0.f) - rmean[c] * scale;
  coef[2 * Ct + c_off + c] = rmean[c];
  coef[3 * Ct + c_off + c] = invstd;
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

// ACTC: the activation as a compile-time constant (SiLU, the one every shipped config uses) or -1 = the runtime `act`.
// The runtime switch makes the compiler budget registers for its heaviest branch (GELU: erff + expf) in EVERY launch:
// with hswish / gelu added, bn_act_bwd_dz went from 79 to 85 VGPRs and lost a wave per SIMD until SiLU got its own instance.
template <typename T, bool FUSED, int ACTC = -1, int UNR = 2>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(int M, int C, const T* __restrict__ z, int z_ld, float* coef, int act_rt,
                                                         const T* __restrict__ res, int r_ld, T* __restrict__ out, int o_ld,
                                                         plyolo_bn_stats st, plyolo_split sp) {
  constexpr int V = Vec<T>::N;
  const int act = ACTC >= 0 ? ACTC : act_rt;
  __shared__ float s_co[FUSED ? 2 * BN_MAXC : 2];
  const int cvn = C / V;
  const ColMap cm(cvn);
  // the first row vector of this thread is requested BEFORE the coefficient prologue (slot sums, fp64 finalisation): on the 20x20 and 40x40
  // maps a workgroup streams one or two vectors per thread and the launch is two dependent round trips long -- this makes it one
  const int m_first = blockIdx.x * cm.rpb + cm.trow;
  const bool pre = FUSED && cm.trow < cm.rpb && m_first < M;
  // (unconditional, on a clamped row: a request behind a branch is waited for at the join)
  typename Vec<T>::raw_t pr;
  if (FUSED) pr = Vec<T>::load_raw(z + (size_t)(m_first < M ? m_first : M - 1) * z_ld + cm.tcol * V);
  if (FUSED) {
    for (int c = threadIdx.x; c < C; c += 256) bn_coef(st, C, c, blockIdx.x == 0, coef, &s_co[c], &s_co[C + c]);
    __syncthreads();
  }
  if (cm.trow >= cm.rpb) return;
  const int step = gridDim.x * cm.rpb;
  for (int cv = cm.tcol; cv < cvn; cv += cm.cols) {
    const int c = cv * V;
    // column-fixed: the destination matrix of this thread's channel vector is chosen once
    const bool second = sp.split > 0 && c >= sp.split;
    T* ob = second ? (T*)sp.p2 + (c - sp.split) : out + c;
    const int ol = second ? sp.ld2 : o_ld;
    float sc[V], sh[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
      if (FUSED) { sc[i] = s_co[c + i]; sh[i] = s_co[C + c + i]; }
      else { sc[i] = coef ? coef[c + i] : 1.f; sh[i] = coef ? coef[C + c + i] : 0.f; }
    }
    auto row = [&](const int m, float* f) {
      float r[V];
      if (res) Vec<T>::load(res + (size_t)m * r_ld + c, r);
#pragma unroll
      for (int i = 0; i < V; ++i) {
        const float u = actf<Vec<T>::precise>(fmaf(f[i], sc[i], sh[i]), act);
        f[i] = res ? u + r[i] : u;
      }
      Vec<T>::store(ob + (size_t)m * ol, f);
    };
    int m = m_first;
    if (pre && cv == cm.tcol) {
      float pf[V];
      Vec<T>::cvt(pr, pf);
      row(m, pf);
      m += step;
    }
#pragma unroll UNR
    for (; m < M; m += step) {
      float f[V];
      Vec<T>::load(z + (size_t)m * z_ld + c, f);
      row(m, f);
    }
  }
}

// bslots[slot][0][c] += sum du ; bslots[slot][1][c] += sum du * zhat ; one add per block and channel
template <typename T, int ACTC = -1, int UNR = 2>
__global__ __launch_bounds__(256) void bn_act_bwd_reduce_kernel(int M, int C, const T* __restrict__ dout, int d_ld, const T* __restrict__ z,
                                                                int z_ld, const float* __restrict__ coef, int act_rt, double* bslots,
                                                                plyolo_split sp) {
  constexpr int V = Vec<T>::N;
  const int act = ACTC >= 0 ? ACTC : act_rt;
  __shared__ float red[256 * 2 * V];
  const int cvn = C / V;
  const ColMap cm(cvn);
  const int step = gridDim.x * cm.rpb;
  double* slot = bslots + (size_t)(blockIdx.x % NSLOT) * 2 * C;
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
      const bool second = sp.split > 0 && c >= sp.split;
      const T* db = second ? (const T*)sp.p2 + (c - sp.split) : dout + c;
      const int dl = second ? sp.ld2 : d_ld;
#pragma unroll UNR
      for (int m = blockIdx.x * cm.rpb + cm.trow; m < M; m += step) {
        float d[V], zz[V];
        Vec<T>::load(db + (size_t)m * dl, d);
        Vec<T>::load(z + (size_t)m * z_ld + c, zz);
#pragma unroll
        for (int i = 0; i < V; ++i) {
          const float u = fmaf(zz[i], sc[i], sh[i]);
          const float du = d[i] * act_grad<Vec<T>::precise>(u, act);
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
    }
    __syncthreads();
    // transpose through LDS so that consecutive lanes add to consecutive channels (coalesced atomics)
    float* tot = red;  // [2][cols*V]
    const int nch = cm.cols * V;
    if (cm.trow == 0 && cv < cvn) {
#pragma unroll
      for (int i = 0; i < V; ++i) { tot[cm.tcol * V + i] = s1[i]; tot[nch + cm.tcol * V + i] = s2[i]; }
    }
    __syncthreads();
    const int cbase = cv0 * V, cend = C - cbase < nch ? C - cbase : nch;
    for (int j = threadIdx.x; j < 2 * cend; j += 256) {
      const int which = j >= cend, cc = j - which * cend;
      slot_add(slot + (size_t)which * C + cbase + cc, (double)tot[which * nch + cc]);
    }
    __syncthreads();
  }
}

template <typename T, int ACTC = -1, int UNR = 2>
__global__ __launch_bounds__(256) void bn_act_bwd_dz_kernel(int M, int C, const T* __restrict__ dout, int d_ld, const T* __restrict__ z,
                                                            int z_ld, const float* __restrict__ coef, const double* __restrict__ bslots,
                                                            double count, const float* gamma, float* dgamma, float* dbeta,
                                                            int accumulate, int act_rt, T* __restrict__ dz, int dz_ld, plyolo_split sp,
                                                            plyolo_bn_bwd_split p2) {
  constexpr int V = Vec<T>::N;
  const int act = ACTC >= 0 ? ACTC : act_rt;
  __shared__ float s_b[3 * BN_MAXC];
  const int cvn = C / V;
  const ColMap cm(cvn);
  // first row vectors (gradient and z) requested before the coefficient prologue, as in bn_act_fwd_kernel
  const int m_first = blockIdx.x * cm.rpb + cm.trow;
  const bool pre = cm.trow < cm.rpb && m_first < M;
  typename Vec<T>::raw_t prd, prz;
  {
    const int c = cm.tcol * V, mp = m_first < M ? m_first : M - 1;
    const bool second = sp.split > 0 && c >= sp.split;
    prd = Vec<T>::load_raw((second ? (const T*)sp.p2 + (c - sp.split) : dout + c) + (size_t)mp * (second ? sp.ld2 : d_ld));
    prz = Vec<T>::load_raw(z + (size_t)mp * z_ld + c);
  }
  // dz = A*du + B*z + Cc with A = gamma*invstd, B = -A*invstd*mean(du*zhat), Cc = -A*mean(du) - B*mean
  for (int c = threadIdx.x; c < C; c += 256) {
    double s, ss;
    slot_sums(bslots, C, c, &s, &ss);
    const float mean = coef[2 * C + c], invstd = coef[3 * C + c];
    const bool sec = p2.split > 0 && c >= p2.split;
    const int cp = sec ? c - p2.split : c;
    const float* gam = sec ? p2.gamma2 : gamma;
    const float A = (gam ? gam[cp] : 1.f) * invstd;
    const float B = (float)(-(double)A * (ss / count) * (double)invstd);
    s_b[c] = A;
    s_b[C + c] = B;
    s_b[2 * C + c] = (float)(-(double)A * (s / count) - (double)B * (double)mean);
    if (blockIdx.x == 0) {
      float* db_ = sec ? p2.dbeta2 : dbeta;
      float* dg_ = sec ? p2.dgamma2 : dgamma;
      if (db_) db_[cp] = (accumulate ? db_[cp] : 0.f) + (float)s;
      if (dg_) dg_[cp] = (accumulate ? dg_[cp] : 0.f) + (float)ss;
    }
  }
  __syncthreads();
  if (cm.trow >= cm.rpb) return;
  const int step = gridDim.x * cm.rpb;
  for (int cv = cm.tcol; cv < cvn; cv += cm.cols) {
    const int c = cv * V;
    float sc[V], sh[V], A[V], B[V], Cc[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
      sc[i] = coef[c + i]; sh[i] = coef[C + c + i];
      A[i] = s_b[c + i]; B[i] = s_b[C + c + i]; Cc[i] = s_b[2 * C + c + i];
    }
    const bool second = sp.split > 0 && c >= sp.split;
    const T* db = second ? (const T*)sp.p2 + (c - sp.split) : dout + c;
    const int dl = second ? sp.ld2 : d_ld;
    T* fwd = (T*)sp.fwd_to;
    auto row = [&](const int m, float* d, const float* zz) {
      float e[V];
      if (fwd) {   // the shortcut's share of dout: copied / added by the pass that read it (no plyolo_copy_add launch)
#pragma unroll
        for (int i = 0; i < V; ++i) e[i] = d[i];
      }
#pragma unroll
      for (int i = 0; i < V; ++i) {
        const float u = fmaf(zz[i], sc[i], sh[i]);
        const float du = d[i] * act_grad<Vec<T>::precise>(u, act);
        d[i] = fmaf(A[i], du, fmaf(B[i], zz[i], Cc[i]));
      }
      Vec<T>::store(dz + (size_t)m * dz_ld + c, d);
      if (fwd) {
        if (sp.fwd_acc) {
          float o[V];
          Vec<T>::load(fwd + (size_t)m * sp.fwd_ld + c, o);
#pragma unroll
          for (int i = 0; i < V; ++i) e[i] += o[i];
        }
        Vec<T>::store(fwd + (size_t)m * sp.fwd_ld + c, e);
      }
    };
    int m = m_first;
    if (pre && cv == cm.tcol) {
      float pd[V], pz[V];
      Vec<T>::cvt(prd, pd);
      Vec<T>::cvt(prz, pz);
      row(m, pd, pz);
      m += step;
    }
#pragma unroll UNR
    for (; m < M; m += step) {
      float d[V], zz[V];
      Vec<T>::load(db + (size_t)m * dl, d);
      Vec<T>::load(z + (size_t)m * z_ld + c, zz);
      row(m, d, zz);
    }
  }
}

// slots += per-channel sum / sum of squares of an activation matrix (BatchNorm applied directly to a tensor:
// the identity branch of RepConv, yolov7_neck.py:191,204-209)
template <typename T>
__global__ __launch_bounds__(256) void channel_stats_kernel(int M, int C, const T* __restrict__ x, int x_ld, double* slots) {
  constexpr int V = Vec<T>::N;
  __shared__ float red[256 * 2 * V];
  const int cvn = C / V;
  const ColMap cm(cvn);
  const int step = gridDim.x * cm.rpb;
  double* slot = slots + (size_t)(blockIdx.x % NSLOT) * 2 * C;
  for (int cv0 = 0; cv0 < cvn; cv0 += cm.cols) {
    const int cv = cv0 + cm.tcol;
    float s1[V], s2[V];
#pragma unroll
    for (int i = 0; i < V; ++i) s1[i] = s2[i] = 0.f;
    if (cm.trow < cm.rpb && cv < cvn)
      for (int m = blockIdx.x * cm.rpb + cm.trow; m < M; m += step) {
        float f[V];
        Vec<T>::load(x + (size_t)m * x_ld + cv * V, f);
#pragma unroll
        for (int i = 0; i < V; ++i) { s1[i] += f[i]; s2[i] += f[i] * f[i]; }
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
        slot_add(slot + cv * V + i, (double)s1[i]);
        slot_add(slot + C + cv * V + i, (double)s2[i]);
      }
    }
    __syncthreads();
  }
}

// din (+)= dout * act'(z): backward of a bare activation (RepConv applies SiLU to a SUM of BatchNorm outputs)
template <typename T>
__global__ __launch_bounds__(256) void act_bwd_kernel(int M, int C, const T* __restrict__ dout, int d_ld, const T* __restrict__ z, int z_ld,
                                                      int act, T* __restrict__ din, int di_ld, int accumulate) {
  constexpr int V = Vec<T>::N;
  const int cvn = C / V;
  const ColMap cm(cvn);
  if (cm.trow >= cm.rpb) return;
  const int step = gridDim.x * cm.rpb;
  for (int cv = cm.tcol; cv < cvn; cv += cm.cols) {
    const int c = cv * V;
    for (int m = blockIdx.x * cm.rpb + cm.trow; m < M; m += step) {
      float d[V], zz[V], o[V];
      Vec<T>::load(dout + (size_t)m * d_ld + c, d);
      Vec<T>::load(z + (size_t)m * z_ld + c, zz);
      if (accumulate) Vec<T>::load(din + (size_t)m * di_ld + c, o);
#pragma unroll
      for (int i = 0; i < V; ++i) d[i] = d[i] * act_grad<Vec<T>::precise>(zz[i], act) + (accumulate ? o[i] : 0.f);
      Vec<T>::store(din + (size_t)m * di_ld + c, d);
    }
  }
}

// ---- deploy-time folding (inference export): BatchNorm folded into the convolution it follows, and the three
// branches of a RepConv collapsed into one 3x3 convolution with bias (reference models/necks/yolov7_neck.py:213-348:
// _fuse_bn_tensor / get_equivalent_kernel_bias / fuse_conv_bn / fuse_repvgg_block; network_blocks.py:39-40 fuseforward).
// fp32 master weights in, fp32 out; one thread per output weight, same operation order as the reference
// (std = sqrt(var + eps); t = gamma / std; w * t; beta - mean * gamma / std).
struct FoldBn {
  const float *gamma, *beta, *mean, *var;
  float eps;
};
DEVINL void fold_coef(const FoldBn& bn, int co, float* t, float* bias) {
  const float sd = sqrtf(bn.var[co] + bn.eps);
  const float g = bn.gamma ? bn.gamma[co] : 1.f, be = bn.beta ? bn.beta[co] : 0.f;
  *t = g / sd;
  *bias = be - bn.mean[co] * g / sd;
}
// conv [Cout][K] (K = Cin*k*k) + optional conv bias -> folded weights / bias
__global__ void fold_conv_bn_kernel(const float* w, const float* cb, FoldBn bn, int Cout, int K, float* wo, float* bo) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)Cout * K) return;
  const int co = (int)(idx / K);
  float t, bias;
  fold_coef(bn, co, &t, &bias);
  wo[idx] = w[idx] * t;
  if (idx % K == 0) bo[co] = cb ? bias + cb[co] * t : bias;   // fuse_conv_bn is only defined for bias-free convs; a bias scales with t
}
// RepConv: 3x3 branch + 1x1 branch padded to the centre tap + identity BatchNorm (an identity 1x1 kernel), each folded with
// its own BatchNorm, summed.  w3 [Cout][Cin][3][3], w1 [Cout][Cin]; has_id needs Cout == Cin.
__global__ void repconv_fuse_kernel(const float* w3, FoldBn bn3, const float* w1, FoldBn bn1, FoldBn bnid, int has_id, int Cout, int Cin,
                                    float* wo, float* bo) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)Cout * Cin * 9) return;
  const int tap = (int)(idx % 9), ci = (int)((idx / 9) % Cin), co = (int)(idx / (9 * (size_t)Cin));
  float t3, b3, t1, b1, ti = 0.f, bi = 0.f;
  fold_coef(bn3, co, &t3, &b3);
  fold_coef(bn1, co, &t1, &b1);
  if (has_id) fold_coef(bnid, co, &ti, &bi);
  float v = w3[idx] * t3;
  if (tap == 4) {
    float c = w1[(size_t)co * Cin + ci] * t1;
    if (has_id) c = c + (ci == co ? 1.f : 0.f) * ti;     // kernel3x3 + pad(kernel1x1) + kernelid, in that order (:217)
    v = v + c;
  }
  wo[idx] = v;
  if (tap == 0 && ci == 0) bo[co] = has_id ? (b3 + b1) + bi : b3 + b1;
}
// coef (scale | shift) of a BatchNorm-free conv unit with bias: scale 1, shift = bias (fused inference epilogue)
__global__ void bias_coef_kernel(int C, const float* bias, float* coef) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  coef[c] = 1.f;
  coef[C + c] = bias ? bias[c] : 0.f;
  coef[2 * C + c] = 0.f;
  coef[3 * C + c] = 1.f;
}

// 256 CUs x 4 workgroups; every workgroup re-reads the stat slots, so keep the grid bounded
inline int bn_unr() {
  static const int u = getenv("PLYOLO_BN_UNR") ? atoi(getenv("PLYOLO_BN_UNR")) : 2;
  return u;
}

inline unsigned grid_lines(size_t lines) {
  size_t g = (lines + 255) / 256;
  return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

inline int stream_grid(int M, int cvn) {
  const int cols = cvn < 256 ? cvn : 256, rpb = 256 / cols;
  int g = (M + rpb - 1) / rpb;
  // measured (tools/bench_bn.py): 2048 workgroups stream 5-8 % faster on tensors of 50 MB and more, 1024 is better
  // below (each workgroup pays the slot-sum prologue)
  static const int cap_env = getenv("PLYOLO_BN_GRID") ? atoi(getenv("PLYOLO_BN_GRID")) : 0;
  const int cap = cap_env > 0 ? cap_env : ((double)M * cvn * 16.0 >= 48.0e6 ? 2048 : 1024);
  if (g > cap) g = cap;
  return g < 1 ? 1 : g;
}

// ---- norm = "ln" of BaseConv: nn.LayerNorm(out_channels) applied to an NCHW tensor (reference models/layers/normalization.py:9-10)
// normalises the LAST axis -- the image WIDTH -- with affine parameters indexed by the column x (torch requires
// W == out_channels).  In the NHWC storage of this library: for every (image, row, channel) the W values x[n, y, :, c].
// One thread per (n, y, c) line (adjacent threads = adjacent channels: coalesced), fp32 statistics kept for the backward.
template <typename T>
__global__ void lnw_act_fwd_kernel(int N, int H, int W, int C, const T* __restrict__ x, int x_ld, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float eps, int act, T* __restrict__ out, int o_ld, float* __restrict__ stats) {
  const size_t lines = (size_t)N * H * C;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < lines; idx += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C);
    const size_t row = idx / C;               // n * H + y
    const T* xp = x + row * W * (size_t)x_ld + c;
    float s = 0.f;
    for (int w = 0; w < W; ++w) s += ActT<T>::ld(xp + (size_t)w * x_ld);
    const float mean = s / (float)W;
    float v = 0.f;
    for (int w = 0; w < W; ++w) { const float d = ActT<T>::ld(xp + (size_t)w * x_ld) - mean; v = fmaf(d, d, v); }
    const float rstd = 1.0f / sqrtf(v / (float)W + eps);
    stats[2 * idx] = mean;
    stats[2 * idx + 1] = rstd;
    T* op = out + row * W * (size_t)o_ld + c;
    for (int w = 0; w < W; ++w) {
      const float u = fmaf((ActT<T>::ld(xp + (size_t)w * x_ld) - mean) * rstd, gamma[w], beta[w]);
      ActT<T>::st(op + (size_t)w * o_ld, act_fwd_precise(u, act));
    }
  }
}

// dx of the same lines: du = dout * act'(u); g = du * gamma[w]; dx = rstd * (g - mean_w(g) - xhat * mean_w(g * xhat))
template <typename T>
__global__ void lnw_act_bwd_dx_kernel(int N, int H, int W, int C, const T* __restrict__ dout, int d_ld, const T* __restrict__ x, int x_ld,
                                      const float* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta, int act,
                                      T* __restrict__ dx, int dx_ld, int accumulate) {
  const size_t lines = (size_t)N * H * C;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < lines; idx += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C);
    const size_t row = idx / C;
    const T* xp = x + row * W * (size_t)x_ld + c;
    const T* dp = dout + row * W * (size_t)d_ld + c;
    const float mean = stats[2 * idx], rstd = stats[2 * idx + 1];
    float s1 = 0.f, s2 = 0.f;
    for (int w = 0; w < W; ++w) {
      const float xh = (ActT<T>::ld(xp + (size_t)w * x_ld) - mean) * rstd;
      const float g = ActT<T>::ld(dp + (size_t)w * d_ld) * act_grad<true>(fmaf(xh, gamma[w], beta[w]), act) * gamma[w];
      s1 += g;
      s2 = fmaf(g, xh, s2);
    }
    s1 /= (float)W;
    s2 /= (float)W;
    T* op = dx + row * W * (size_t)dx_ld + c;
    for (int w = 0; w < W; ++w) {
      const float xh = (ActT<T>::ld(xp + (size_t)w * x_ld) - mean) * rstd;
      const float g = ActT<T>::ld(dp + (size_t)w * d_ld) * act_grad<true>(fmaf(xh, gamma[w], beta[w]), act) * gamma[w];
      float r = rstd * (g - s1 - xh * s2);
      if (accumulate) r += ActT<T>::ld(op + (size_t)w * dx_ld);
      ActT<T>::st(op + (size_t)w * dx_ld, r);
    }
  }
}

// dgamma[w] = sum over lines of du * xhat, dbeta[w] = sum of du: one workgroup per column w, fixed summation order (deterministic)
template <typename T>
__global__ __launch_bounds__(256) void lnw_act_bwd_params_kernel(int N, int H, int W, int C, const T* __restrict__ dout, int d_ld,
                                                                const T* __restrict__ x, int x_ld, const float* __restrict__ stats,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta, int act,
                                                                float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate) {
  const int w = blockIdx.x;
  const size_t lines = (size_t)N * H * C;
  const float gw = gamma[w], bw = beta[w];
  double a = 0.0, b = 0.0;
  for (size_t idx = threadIdx.x; idx < lines; idx += 256) {
    const int c = (int)(idx % C);
    const size_t row = idx / C;
    const float xh = (ActT<T>::ld(x + (row * W + w) * (size_t)x_ld + c) - stats[2 * idx]) * stats[2 * idx + 1];
    const float du = ActT<T>::ld(dout + (row * W + w) * (size_t)d_ld + c) * act_grad<true>(fmaf(xh, gw, bw), act);
    a += (double)du * xh;
    b += (double)du;
  }
  __shared__ double ra[256], rb[256];
  ra[threadIdx.x] = a; rb[threadIdx.x] = b;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) { ra[threadIdx.x] += ra[threadIdx.x + st]; rb[threadIdx.x] += rb[threadIdx.x + st]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    dgamma[w] = (accumulate ? dgamma[w] : 0.f) + (float)ra[0];
    dbeta[w] = (accumulate ? dbeta[w] : 0.f) + (float)rb[0];
  }
}

}  // namespace

using plyolo::submit;

#define DISPATCH_T(dtype, ...)                       \
  if ((dtype) == PLYOLO_BF16) { typedef bf16_t T; __VA_ARGS__ } \
  else { typedef float T; __VA_ARGS__ }

extern "C" {

int plyolo_bn_finalize(const plyolo_bn_stats* stp, int C, float* coef, void* stream) {
  PLY_CHECK_ARG(stp && stp->slots && stp->count > 0 && coef, "bn_finalize: incomplete plyolo_bn_stats");
  const plyolo_bn_stats st = *stp;
  plyolo::annotate("bn_finalize", 0.0, 16.0 * NSLOT * C);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 256)), dim3(256), 0, s, st, C, coef);
    return hipGetLastError();
  });
}

int plyolo_bn_eval_coef(int C, const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                        float eps, float* coef, void* stream) {
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    hipLaunchKernelGGL(bn_eval_coef_kernel, dim3(cdiv(C, 256)), dim3(256), 0, s, C, gamma, beta, running_mean, running_var, eps, coef, C, 0);
    return hipGetLastError();
  });
}

int plyolo_bias_coef(int C, const float* bias, float* coef, void* stream) {
  PLY_CHECK_ARG(C > 0 && coef, "bias_coef: bad arguments");
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    hipLaunchKernelGGL(bias_coef_kernel, dim3(cdiv(C, 256)), dim3(256), 0, s, C, bias, coef);
    return hipGetLastError();
  });
}

static FoldBn fold_of(const plyolo_bn_params* b) {
  FoldBn f{};
  if (b) { f.gamma = b->gamma; f.beta = b->beta; f.mean = b->running_mean; f.var = b->running_var; f.eps = b->eps; }
  return f;
}

int plyolo_fold_conv_bn(const float* w, const float* conv_bias, const plyolo_bn_params* bn, int Cout, int K, float* w_out, float* b_out,
                        void* stream) {
  PLY_CHECK_ARG(w && bn && bn->running_mean && bn->running_var && w_out && b_out && Cout > 0 && K > 0, "fold_conv_bn: incomplete arguments");
  const FoldBn f = fold_of(bn);
  const size_t total = (size_t)Cout * K;
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    hipLaunchKernelGGL(fold_conv_bn_kernel, dim3((unsigned)cdivz(total, 256)), dim3(256), 0, s, w, conv_bias, f, Cout, K, w_out, b_out);
    return hipGetLastError();
  });
}

int plyolo_repconv_fuse(const float* w3, const plyolo_bn_params* bn3, const float* w1, const plyolo_bn_params* bn1,
                        const plyolo_bn_params* bn_id, int Cout, int Cin, float* w_out, float* b_out, void* stream) {
  PLY_CHECK_ARG(w3 && w1 && bn3 && bn1 && bn3->running_var && bn1->running_var && w_out && b_out && Cout > 0 && Cin > 0,
                "repconv_fuse: incomplete arguments");
  PLY_CHECK_ARG(!bn_id || (Cout == Cin && bn_id->running_var), "repconv_fuse: the identity branch needs Cout == Cin");
  const FoldBn f3 = fold_of(bn3), f1 = fold_of(bn1), fi = fold_of(bn_id);
  const int has_id = bn_id != nullptr;
  const size_t total = (size_t)Cout * Cin * 9;
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    hipLaunchKernelGGL(repconv_fuse_kernel, dim3((unsigned)cdivz(total, 256)), dim3(256), 0, s, w3, f3, w1, f1, fi, has_id, Cout, Cin, w_out, b_out);
    return hipGetLastError();
  });
}

int plyolo_bn_eval_coef_at(int C, const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                           float eps, float* coef, int C_total, int c_off, void* stream) {
  PLY_CHECK_ARG(C > 0 && c_off >= 0 && c_off + C <= C_total, "bn_eval_coef_at: bad channel window");
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    hipLaunchKernelGGL(bn_eval_coef_kernel, dim3(cdiv(C, 256)), dim3(256), 0, s, C, gamma, beta, running_mean, running_var, eps, coef,
                       C_total, c_off);
    return hipGetLastError();
  });
}

static int check_split(const plyolo_split* sp, int C, int V, const char* who) {
  if (!sp || sp->split == 0) return 0;
  PLY_CHECK_ARG(sp->split > 0 && sp->split < C && sp->split % V == 0 && sp->p2 && sp->ld2 % V == 0 && sp->ld2 >= C - sp->split,
                "%s: bad channel split", who);
  return 0;
}

int plyolo_bn_act_fwd(int dtype, int M, int C, const void* z, int z_ld, float* coef, int act, const void* res, int r_ld,
                      void* out, int o_ld, const plyolo_bn_stats* stp, const plyolo_split* osp, void* stream) {
  const int V = dtype == PLYOLO_BF16 ? 8 : 4;
  PLY_CHECK_ARG(C % V == 0 && z_ld % V == 0 && o_ld % V == 0 && (!res || r_ld % V == 0), "bn_act_fwd: C/ld must be multiples of %d", V);
  PLY_CHECK_ARG(!stp || (stp->slots && stp->count > 0 && coef && C <= BN_MAXC), "bn_act_fwd: incomplete plyolo_bn_stats (or C > %d)", BN_MAXC);
  if (check_split(osp, C, V, "bn_act_fwd")) return -1;
  PLY_CHECK_ARG(!(osp && osp->split) || !res, "bn_act_fwd: a split output cannot take a residual");
  const int grid = stream_grid(M, C / V);
  plyolo_bn_stats st{};
  if (stp) st = *stp;
  plyolo_split sp{};
  if (osp) sp = *osp;
  const bool fused = stp != nullptr;
  plyolo::annotate("bn_act_fwd", 0.0, (double)M * C * (dtype == PLYOLO_BF16 ? 2.0 : 4.0) * (res ? 3.0 : 2.0));
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    DISPATCH_T(dtype, {
      auto kern = fused ? (act == PLYOLO_ACT_SILU ? (bn_unr() == 4 ? bn_act_fwd_kernel<T, true, PLYOLO_ACT_SILU, 4> : bn_act_fwd_kernel<T, true, PLYOLO_ACT_SILU>)
                                                  : bn_act_fwd_kernel<T, true, -1>)
                        : (act == PLYOLO_ACT_SILU ? bn_act_fwd_kernel<T, false, PLYOLO_ACT_SILU> : bn_act_fwd_kernel<T, false, -1>);
      hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, s, M, C, (const T*)z, z_ld, coef, act, (const T*)res, r_ld, (T*)out, o_ld, st, sp);
    })
    return hipGetLastError();
  });
}

int plyolo_channel_stats(int dtype, int M, int C, const void* x, int x_ld, double* slots, void* stream) {
  const int V = dtype == PLYOLO_BF16 ? 8 : 4;
  PLY_CHECK_ARG(C % V == 0 && x_ld % V == 0 && slots, "channel_stats: C/ld must be multiples of %d", V);
  int rows = M / 32;
  if (rows < 1) rows = 1;
  if (rows > 1024) rows = 1024;
  plyolo::annotate("channel_stats", 0.0, (double)M * C * (dtype == PLYOLO_BF16 ? 2.0 : 4.0));
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    DISPATCH_T(dtype, hipLaunchKernelGGL(channel_stats_kernel<T>, dim3(rows), dim3(256), 0, s, M, C, (const T*)x, x_ld, slots);)
    return hipGetLastError();
  });
}

int plyolo_act_bwd(int dtype, int M, int C, const void* dout, int d_ld, const void* z, int z_ld, int act, void* din, int di_ld,
                   int accumulate, void* stream) {
  const int V = dtype == PLYOLO_BF16 ? 8 : 4;
  PLY_CHECK_ARG(C % V == 0 && z_ld % V == 0 && d_ld % V == 0 && di_ld % V == 0, "act_bwd: C/ld must be multiples of %d", V);
  const int grid = stream_grid(M, C / V);
  plyolo::annotate("act_bwd", 0.0, (double)M * C * (dtype == PLYOLO_BF16 ? 2.0 : 4.0) * 3.0);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    DISPATCH_T(dtype, hipLaunchKernelGGL(act_bwd_kernel<T>, dim3(grid), dim3(256), 0, s, M, C, (const T*)dout, d_ld, (const T*)z, z_ld,
                                         act, (T*)din, di_ld, accumulate);)
    return hipGetLastError();
  });
}

int plyolo_bn_act_bwd_reduce(int dtype, int M, int C, const void* dout, int d_ld, const void* z, int z_ld, const float* coef,
                             int act, double* bslots, const plyolo_split* dsp, void* stream) {
  const int V = dtype == PLYOLO_BF16 ? 8 : 4;
  PLY_CHECK_ARG(C % V == 0 && z_ld % V == 0 && d_ld % V == 0, "bn_act_bwd_reduce: C/ld must be multiples of %d", V);
  if (check_split(dsp, C, V, "bn_act_bwd_reduce")) return -1;
  plyolo_split sp{};
  if (dsp) sp = *dsp;
  int div = 32;   // pixel rows per workgroup (measured: 32 beats 64 on the 40x40 / 20x20 maps, equal on the large ones) (every workgroup ends with 2*C fp64 slot atomics)
  if (const char* e = getenv("PLYOLO_BN_RED_DIV")) { const int v = atoi(e); if (v >= 1) div = v; }
  int rows = M / div;
  if (rows < 1) rows = 1;
  static const int cap_env = getenv("PLYOLO_BN_RED_CAP") ? atoi(getenv("PLYOLO_BN_RED_CAP")) : 0;
  // 512 workgroups x 4 rows in flight per thread instead of 1024 x 2: the same stand-alone time (faster on the 20x20 maps: half
  // the fp64 slot atomics) and the lighter co-runner for the weight-gradient lane -- step 9.83 vs 9.94 ms, four alternations
  static const int unr = getenv("PLYOLO_BN_RED_UNR") ? atoi(getenv("PLYOLO_BN_RED_UNR")) : 4;
  const int cap = cap_env > 0 ? cap_env : 512;
  if (rows > cap) rows = cap;
  plyolo::annotate("bn_act_bwd_reduce", 0.0, (double)M * C * (dtype == PLYOLO_BF16 ? 2.0 : 4.0) * 2.0);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    DISPATCH_T(dtype, {
      auto kern = act == PLYOLO_ACT_SILU ? (unr == 4 ? bn_act_bwd_reduce_kernel<T, PLYOLO_ACT_SILU, 4> : unr == 1 ? bn_act_bwd_reduce_kernel<T, PLYOLO_ACT_SILU, 1> : bn_act_bwd_reduce_kernel<T, PLYOLO_ACT_SILU, 2>)
                                         : bn_act_bwd_reduce_kernel<T, -1>;
      hipLaunchKernelGGL(kern, dim3(rows), dim3(256), 0, s, M, C, (const T*)dout, d_ld, (const T*)z, z_ld, coef, act, bslots, sp);
    })
    return hipGetLastError();
  });
}

int plyolo_bn_act_bwd_dz(int dtype, int M, int C, const void* dout, int d_ld, const void* z, int z_ld, const float* coef,
                         const double* bslots, const float* gamma, float* dgamma, float* dbeta, int accumulate, int act, void* dz,
                         int dz_ld, const plyolo_split* dsp, const plyolo_bn_bwd_split* par2, void* stream) {
  const int V = dtype == PLYOLO_BF16 ? 8 : 4;
  PLY_CHECK_ARG(C % V == 0 && z_ld % V == 0 && d_ld % V == 0 && dz_ld % V == 0, "bn_act_bwd_dz: C/ld must be multiples of %d", V);
  PLY_CHECK_ARG(C <= BN_MAXC && bslots && coef, "bn_act_bwd_dz: C > %d or missing slots/coef", BN_MAXC);
  if (check_split(dsp, C, V, "bn_act_bwd_dz")) return -1;
  plyolo_split sp{};
  if (dsp) sp = *dsp;
  PLY_CHECK_ARG(!sp.fwd_to || (sp.split == 0 && sp.fwd_ld % V == 0 && sp.fwd_ld >= C), "bn_act_bwd_dz: the forwarded gradient needs fwd_ld %% %d == 0, fwd_ld >= C and no channel split", V);
  plyolo_bn_bwd_split p2{};
  if (par2) p2 = *par2;
  const int grid = stream_grid(M, C / V);
  const double count = (double)M;
  plyolo::annotate("bn_act_bwd_dz", 0.0, (double)M * C * (dtype == PLYOLO_BF16 ? 2.0 : 4.0) * 3.0);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    DISPATCH_T(dtype, {
      auto kern = act == PLYOLO_ACT_SILU ? (bn_unr() == 4 ? bn_act_bwd_dz_kernel<T, PLYOLO_ACT_SILU, 4> : bn_act_bwd_dz_kernel<T, PLYOLO_ACT_SILU>)
                                         : bn_act_bwd_dz_kernel<T, -1>;
      hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, s, M, C, (const T*)dout, d_ld, (const T*)z, z_ld, coef, bslots, count, gamma, dgamma,
                         dbeta, accumulate, act, (T*)dz, dz_ld, sp, p2);
    })
    return hipGetLastError();
  });
}

int plyolo_lnw_act_fwd(int dtype, int N, int H, int W, int C, const void* x, int x_ld, const float* gamma, const float* beta, float eps, int act,
                       void* out, int o_ld, float* stats, void* stream) {
  PLY_CHECK_ARG(x && out && gamma && beta && stats && N > 0 && H > 0 && W > 0 && C > 0 && x_ld >= C && o_ld >= C, "lnw_act_fwd: bad arguments");
  const size_t lines = (size_t)N * H * C;
  plyolo::annotate("lnw_act_fwd", 0.0, (double)lines * W * (dtype == PLYOLO_BF16 ? 2.0 : 4.0) * 2.0);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    DISPATCH_T(dtype, hipLaunchKernelGGL(lnw_act_fwd_kernel<T>, dim3(grid_lines(lines)), dim3(256), 0, s, N, H, W, C, (const T*)x, x_ld, gamma, beta, eps,
                                         act, (T*)out, o_ld, stats);)
    return hipGetLastError();
  });
}

int plyolo_lnw_act_bwd(int dtype, int N, int H, int W, int C, const void* dout, int d_ld, const void* x, int x_ld, const float* stats,
                       const float* gamma, const float* beta, int act, void* dx, int dx_ld, int accumulate_dx, float* dgamma, float* dbeta,
                       int accumulate_params, void* stream) {
  PLY_CHECK_ARG(dout && x && stats && gamma && beta && dx && dgamma && dbeta && N > 0 && H > 0 && W > 0 && C > 0, "lnw_act_bwd: bad arguments");
  const size_t lines = (size_t)N * H * C;
  plyolo::annotate("lnw_act_bwd", 0.0, (double)lines * W * (dtype == PLYOLO_BF16 ? 2.0 : 4.0) * 5.0);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    DISPATCH_T(dtype, {
      hipLaunchKernelGGL(lnw_act_bwd_params_kernel<T>, dim3(W), dim3(256), 0, s, N, H, W, C, (const T*)dout, d_ld, (const T*)x, x_ld, stats, gamma, beta,
                         act, dgamma, dbeta, accumulate_params);
      hipLaunchKernelGGL(lnw_act_bwd_dx_kernel<T>, dim3(grid_lines(lines)), dim3(256), 0, s, N, H, W, C, (const T*)dout, d_ld, (const T*)x, x_ld, stats,
                         gamma, beta, act, (T*)dx, dx_ld, accumulate_dx);
    })
    return hipGetLastError();
  });
}

}  // extern "C"
