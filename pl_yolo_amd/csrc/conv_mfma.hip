// bf16 MFMA direct convolution for gfx950 (forward and data-gradient).
//
// Replaces nn.Conv2d inside BaseConv (reference models/layers/network_blocks.py:18-26)
// and what ATen's convolution_backward computes for its input.
//
// Formulation (im2col-free): one workgroup (4 waves) owns an 8x16 tile of output
// positions x BN output channels.  Per Cin-chunk (CK channels) the input HALO tile
// ((8-1)*si+ext_y) x ((16-1)*si+ext_x) pixels is staged ONCE into LDS from coalesced
// NHWC rows; the packed weights of one filter tap ([BN][CK]) are streamed through a
// double-buffered LDS tile with a register prefetch one tap ahead; each tap is then a
// dense [128 x CK] x [CK x BN] product on v_mfma_f32_32x32x16_bf16 whose A fragments
// are the tap-shifted rows of the halo tile (no data movement per tap).
// Epilogue (fused): per-channel sum / sum-of-squares partials for train-mode
// BatchNorm, optional bias, bf16 (or fp32) store through an LDS transpose so that
// global stores are whole 16-byte channel vectors; concat = strided store (y_ld).
//
// The same kernel serves dgrad: stride-1 dgrad is a forward conv with the tap table
// mirrored; stride-2 dgrad is four launches, one per output parity class, each with
// the 1/2/2/4 taps that reach that class (so no zero-stuffing and no atomics).
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int TH = 8, TW = 16, BM = 128;

struct ConvP {
  const bf16_t* x;
  const bf16_t* w;
  void* y;
  const float* bias;
  float* stats;
  int N, H, W, Cin, Cout, x_ld, y_ld;
  int OHt, OWt;  // extent of the output position grid handled by this launch
  int OHf, OWf;  // full output tensor spatial dims
  int so, oy_off, ox_off;
  int si, iy_off, ix_off;
  int ITH, ITW;
  int ntaps;
  int tiles_y, tiles_x, nmb;
  int accumulate;
  int w_ld, wtap_stride;
  signed char tap_dy[9], tap_dx[9], tap_w[9];  // host-side table
  // the same table packed 8 bits per tap (dy | dx<<2 | w<<4): decoded with scalar shifts in the
  // kernel -- indexing a kernarg ARRAY with a runtime tap index makes hipcc emit VMEM byte loads,
  // whose s_waitcnt vmcnt(0) would also drain the in-flight weight prefetch every tap
  unsigned long long taps_lo;
  unsigned int taps_hi;
  plyolo_bn_fuse fin;  // fin.coef != NULL: BatchNorm statistics are finished inside this launch
  int ablate;  // diagnostic builds only (PLYOLO_ABLATE): 1 skip stores, 2 skip stats, 4 skip halo loads, 8 skip MFMA, 16 skip weight loads
};

DEVINL unsigned tap_code(const ConvP& p, int t) {
  return t < 8 ? (unsigned)((p.taps_lo >> (8 * t)) & 0xffull) : (p.taps_hi & 0xffu);
}

DEVINL u32x4 add_bf16x8(u32x4 a, u32x4 b) {
  u32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float lo = __uint_as_float(a[i] << 16) + __uint_as_float(b[i] << 16);
    float hi = __uint_as_float(a[i] & 0xffff0000u) + __uint_as_float(b[i] & 0xffff0000u);
    r[i] = pack2bf(lo, hi);
  }
  return r;
}

template <int BN, int CK, int WM, int WN, bool OUT_F32>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvP p) {
  constexpr int MT = BM / WM / 32, NT = BN / WN / 32;
  constexpr int ROWB = CK * 2 + 16;  // LDS row pitch in bytes (pad: 16 B)
  constexpr int CV = CK / 8;         // 16-byte vectors per row
  constexpr int WVEC = BN * CV;      // weight vectors per tap tile
  constexpr int WV = (WVEC + 255) / 256;
  extern __shared__ __align__(16) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;

  // XCD-aware bijective remap: workgroups with equal (blockIdx.x % 8) share an XCD's
  // L2, give each such group a contiguous run of tiles so halo re-reads hit L2.
  int tile;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    tile = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
  }
  const int txi = tile % p.tiles_x;
  const int t2 = tile / p.tiles_x;
  const int tyi = t2 % p.tiles_y;
  const int n = t2 / p.tiles_y;
  const int oy0 = tyi * TH, ox0 = txi * TW;
  const int iy0 = oy0 * p.si + p.iy_off, ix0 = ox0 * p.si + p.ix_off;
  const int cout0 = blockIdx.y * BN;

  const int in_bytes = p.ITH * p.ITW * ROWB;
  unsigned char* wbase = smem + in_bytes;

  int arow[MT], brow[NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = (wm * MT + mt) * 32 + r;
    arow[mt] = (((m >> 4) * p.si) * p.ITW + (m & 15) * p.si) * ROWB + h * 16;
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) brow[nt] = (wn * (BN / WN) + nt * 32 + r) * ROWB + h * 16;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;

  const int nchunks = (p.Cin + CK - 1) / CK;
  const int total = nchunks * p.ntaps;
  u32x4 wreg[WV];

  auto load_w = [&](int phase) {
    const int chunk = phase / p.ntaps, t = phase - chunk * p.ntaps;
    const int c0 = chunk * CK;
    const bf16_t* wt = p.w + (size_t)(tap_code(p, t) >> 4) * p.wtap_stride;
#pragma unroll
    for (int v = 0; v < WV; ++v) {
      const int idx = tid + v * 256;
      u32x4 val = {0u, 0u, 0u, 0u};
      if (idx < WVEC) {
        const int row = idx / CV, cv = idx % CV;
        const int co = cout0 + row, c = c0 + cv * 8;
        if (co < p.Cout && c < p.Cin) val = *(const u32x4*)(wt + (size_t)co * p.w_ld + c);
      }
      wreg[v] = val;
    }
  };
  auto store_w = [&](int buf) {
    unsigned char* wb = wbase + buf * (BN * ROWB);
#pragma unroll
    for (int v = 0; v < WV; ++v) {
      const int idx = tid + v * 256;
      if (idx < WVEC) {
        const int row = idx / CV, cv = idx % CV;
        *(u32x4*)(wb + row * ROWB + cv * 16) = wreg[v];
      }
    }
  };

  const int abl = p.ablate;
  load_w(0);
  int phase = 0;
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const int c0 = chunk * CK;
    __syncthreads();  // every wave is done reading the previous chunk's halo tile
    if (!(abl & 4) || chunk == 0) {
      // halo tile: issue a whole batch of 16-byte loads before the first LDS write so that
      // HV loads per thread are in flight at once (a load->wait->write loop serialises them)
      constexpr int HV = 6;
      const int nvec = p.ITH * p.ITW * CV;
      const bf16_t* xn = p.x + (size_t)n * p.H * p.W * p.x_ld;
      for (int base = 0; base < nvec; base += HV * 256) {
        u32x4 hv[HV];
#pragma unroll
        for (int v = 0; v < HV; ++v) {
          const int idx = base + tid + v * 256;
          u32x4 val = {0u, 0u, 0u, 0u};
          if (idx < nvec) {
            const int pix = idx / CV, cv = idx - pix * CV;
            const int iy = pix / p.ITW, ix = pix - iy * p.ITW;
            const int gy = iy0 + iy, gx = ix0 + ix, c = c0 + cv * 8;
            if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W && c < p.Cin)
              val = *(const u32x4*)(xn + ((size_t)gy * p.W + gx) * p.x_ld + c);
          }
          hv[v] = val;
        }
#pragma unroll
        for (int v = 0; v < HV; ++v) {
          const int idx = base + tid + v * 256;
          if (idx < nvec) {
            const int pix = idx / CV, cv = idx - pix * CV;
            *(u32x4*)(smem + pix * ROWB + cv * 16) = hv[v];
          }
        }
      }
    }
    // the chunk tail (Cin % CK) is zero-filled in LDS, so every chunk runs all CK/16 k-steps
    // and the loop fully unrolls (fragment loads of step k+1 overlap the MFMAs of step k)
    constexpr int ksteps = CK / 16;
    for (int t = 0; t < p.ntaps; ++t, ++phase) {
      const int buf = phase & 1;
      store_w(buf);
      __syncthreads();  // halo tile + this tap's weights visible
      if (phase + 1 < total && !(abl & 16)) load_w(phase + 1);
      const unsigned tc = tap_code(p, t);
      const int toff = ((int)(tc & 3u) * p.ITW + (int)((tc >> 2) & 3u)) * ROWB;
      const unsigned char* wb = wbase + buf * (BN * ROWB);
      if (!(abl & 8))
#pragma unroll
      for (int kk = 0; kk < ksteps; ++kk) {
        bf16x8 a[MT], b[NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[mt] = *(const bf16x8*)(smem + arow[mt] + toff + kk * 32);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[nt] = *(const bf16x8*)(wb + brow[nt] + kk * 32);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
      }
    }
  }
  __syncthreads();  // all LDS operand reads retired; LDS is reused for the epilogue

  // ---- epilogue ---------------------------------------------------------------
  constexpr int SROW = OUT_F32 ? (BN + 4) * 4 : (BN * 2 + 16);  // staging row pitch (bytes)
  float* red = (float*)(smem + BM * SROW);                       // [WM][2][BN]

  if (p.stats != nullptr && !(abl & 2)) {
    float s1[NT], s2[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) s1[nt] = s2[nt] = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int m = (wm * MT + mt) * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        const bool valid = (oy0 + (m >> 4) < p.OHt) && (ox0 + (m & 15) < p.OWt);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const float v = valid ? acc[mt][nt][i] : 0.f;
          s1[nt] += v;
          s2[nt] += v * v;
        }
      }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      s1[nt] += __shfl_xor(s1[nt], 32);
      s2[nt] += __shfl_xor(s2[nt], 32);
      if (h == 0) {
        const int col = wn * (BN / WN) + nt * 32 + r;
        red[(wm * 2 + 0) * BN + col] = s1[nt];
        red[(wm * 2 + 1) * BN + col] = s2[nt];
      }
    }
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int m = (wm * MT + mt) * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        const int col = wn * (BN / WN) + nt * 32 + r;
        if (OUT_F32)
          *(float*)(smem + m * SROW + col * 4) = acc[mt][nt][i];
        else
          *(bf16_t*)(smem + m * SROW + col * 2) = f2bf(acc[mt][nt][i]);
      }
  __syncthreads();

  if (abl & 1) return;
  if (p.stats != nullptr && !(abl & 2) && tid < BN) {
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int w = 0; w < WM; ++w) {
      s += red[(w * 2 + 0) * BN + tid];
      ss += red[(w * 2 + 1) * BN + tid];
    }
    const int co = cout0 + tid;
    if (co < p.Cout) {
      if (p.fin.coef != nullptr) {  // handed to another workgroup inside this launch: write-through
        hier_store(p.stats + (size_t)tile * p.Cout + co, s);
        hier_store(p.stats + ((size_t)p.nmb + tile) * p.Cout + co, ss);
      } else {
        p.stats[(size_t)tile * p.Cout + co] = s;
        p.stats[((size_t)p.nmb + tile) * p.Cout + co] = ss;
      }
    }
  }

  if (p.fin.coef != nullptr) {
    // ---- BatchNorm finish inside the launch (before the bulk output stores, so the drain in
    // hier_finish only waits for the two row stores): hierarchical last-arriver reduction of
    // the per-workgroup partial rows, then coef / running statistics
    __shared__ int s_flag;
    HierRed h;
    h.rows = p.stats; h.gpart = p.fin.gpart; h.gcnt = p.fin.gcnt; h.fcnt = p.fin.fcnt; h.nrows = p.nmb; h.C = p.Cout;
    hier_finish(h, tile, (int)blockIdx.y, cout0, BN, &s_flag, [&](int co, double s, double ss) {
      const double count = p.fin.count;
      const double mean = s / count;
      double var = ss / count - mean * mean;
      if (var < 0.0) var = 0.0;
      const float invstd = (float)(1.0 / sqrt(var + (double)p.fin.eps));
      const float g = p.fin.gamma ? p.fin.gamma[co] : 1.f, b = p.fin.beta ? p.fin.beta[co] : 0.f;
      const float scale = g * invstd;
      float* coef = p.fin.coef;
      coef[co] = scale;
      coef[p.Cout + co] = b - (float)mean * scale;
      coef[2 * p.Cout + co] = (float)mean;
      coef[3 * p.Cout + co] = invstd;
      const float mom = p.fin.momentum;
      if (p.fin.running_mean) p.fin.running_mean[co] = (1.f - mom) * p.fin.running_mean[co] + mom * (float)mean;
      if (p.fin.running_var) {
        const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
        p.fin.running_var[co] = (1.f - mom) * p.fin.running_var[co] + mom * (float)unb;
      }
      if (co == 0 && p.fin.num_batches_tracked) *p.fin.num_batches_tracked += 1;
    });
    __syncthreads();  // s_flag readers are done before the staging tile is consumed below
  }

  if (OUT_F32) {
    float* y = (float*)p.y;
    for (int idx = tid; idx < BM * BN; idx += 256) {
      const int m = idx / BN, c = idx - m * BN;
      const int a = oy0 + (m >> 4), b = ox0 + (m & 15), co = cout0 + c;
      if (a < p.OHt && b < p.OWt && co < p.Cout) {
        const int oy = a * p.so + p.oy_off, ox = b * p.so + p.ox_off;
        float v = *(const float*)(smem + m * SROW + c * 4);
        if (p.bias) v += p.bias[co];
        float* dst = y + ((size_t)(n * p.OHf + oy) * p.OWf + ox) * p.y_ld + co;
        if (p.accumulate) v += *dst;
        *dst = v;
      }
    }
  } else {
    bf16_t* y = (bf16_t*)p.y;
    constexpr int VPR = BN / 8;
    for (int idx = tid; idx < BM * VPR; idx += 256) {
      const int m = idx / VPR, v = idx - m * VPR;
      const int a = oy0 + (m >> 4), b = ox0 + (m & 15), co = cout0 + v * 8;
      if (a < p.OHt && b < p.OWt && co < p.Cout) {
        const int oy = a * p.so + p.oy_off, ox = b * p.so + p.ox_off;
        u32x4 val = *(const u32x4*)(smem + m * SROW + v * 16);
        bf16_t* dst = y + ((size_t)(n * p.OHf + oy) * p.OWf + ox) * p.y_ld + co;
        if (p.accumulate) val = add_bf16x8(*(const u32x4*)dst, val);
        *(u32x4*)dst = val;
      }
    }
  }

}

template <int BN, int CK, bool OUT_F32>
hipError_t launch_inst(const ConvP& p, hipStream_t s) {
  constexpr int WM = (BN == 32) ? 4 : 2, WN = (BN == 32) ? 1 : 2;
  constexpr int ROWB = CK * 2 + 16;
  constexpr int SROW = OUT_F32 ? (BN + 4) * 4 : (BN * 2 + 16);
  size_t lds_main = (size_t)p.ITH * p.ITW * ROWB + 2 * BN * ROWB;
  size_t lds_epi = (size_t)BM * SROW + WM * 2 * BN * 4;
  size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
  auto kern = conv_mfma_kernel<BN, CK, WM, WN, OUT_F32>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  dim3 grid(p.nmb, (p.Cout + BN - 1) / BN);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, p);
  return hipGetLastError();
}

template <bool OUT_F32>
hipError_t launch_bn(const ConvP& p, int BN, int CK, hipStream_t s) {
#define PLY_CASE(bn, ck) \
  if (BN == bn && CK == ck) return launch_inst<bn, ck, OUT_F32>(p, s);
  PLY_CASE(32, 16) PLY_CASE(32, 32) PLY_CASE(32, 64)
  PLY_CASE(64, 16) PLY_CASE(64, 32) PLY_CASE(64, 64)
  if (!OUT_F32) {
    if (BN == 128 && CK == 16) return launch_inst<128, 16, false>(p, s);
    if (BN == 128 && CK == 32) return launch_inst<128, 32, false>(p, s);
    if (BN == 128 && CK == 64) return launch_inst<128, 64, false>(p, s);
  }
#undef PLY_CASE
  return hipErrorInvalidValue;
}

void pick_tiles(int Cin, int Cout, int si, int ext, bool out_f32, int* BN, int* CK) {
  int bn = Cout >= 128 ? 128 : (Cout > 32 ? 64 : 32);
  if (out_f32 && bn > 64) bn = 64;
  int ck = Cin >= 64 ? 64 : (Cin >= 32 ? 32 : 16);
  if (si == 2 && ext > 1 && ck > 32) ck = 32;  // stride-2 halo tile is 17x33 pixels
  if (const char* e = getenv("PLYOLO_FORCE_CK")) { const int v = atoi(e); if (v == 16 || v == 32 || v == 64) ck = v < ck ? v : ck; }
  if (const char* e = getenv("PLYOLO_FORCE_BN")) { const int v = atoi(e); if (v == 32 || v == 64 || v == 128) bn = v < bn ? v : bn; }
  *BN = bn;
  *CK = ck;
}

void set_grid(ConvP& p) {
  if (const char* e = getenv("PLYOLO_ABLATE")) p.ablate = atoi(e);
  p.taps_lo = 0ull;
  p.taps_hi = 0u;
  for (int t = 0; t < p.ntaps; ++t) {
    const unsigned code = (unsigned)p.tap_dy[t] | ((unsigned)p.tap_dx[t] << 2) | ((unsigned)p.tap_w[t] << 4);
    if (t < 8) p.taps_lo |= (unsigned long long)code << (8 * t);
    else p.taps_hi = code;
  }
  p.tiles_y = (p.OHt + TH - 1) / TH;
  p.tiles_x = (p.OWt + TW - 1) / TW;
  p.nmb = p.N * p.tiles_y * p.tiles_x;
}

}  // namespace

namespace plyolo {

// number of per-block stat rows a forward launch writes
int conv_mfma_stat_rows(const plyolo_conv_desc* d) {
  const int OH = (d->H + 2 * ((d->ksize - 1) / 2) - d->ksize) / d->stride + 1;
  const int OW = (d->W + 2 * ((d->ksize - 1) / 2) - d->ksize) / d->stride + 1;
  if (d->ksize == 1 && d->stride == 1 && ((size_t)d->N * d->H * d->W) % TW == 0) {
    const size_t rows = (size_t)d->N * d->H * d->W / TW;
    return (int)((rows + TH - 1) / TH);
  }
  return d->N * ((OH + TH - 1) / TH) * ((OW + TW - 1) / TW);
}

int conv_mfma_fwd(const plyolo_conv_desc* d, const void* x, const void* wp, const float* bias, void* y,
                  float* stats, const plyolo_bn_fuse* fin, void* stream) {
  const int pad = (d->ksize - 1) / 2;
  ConvP p{};
  p.x = (const bf16_t*)x;
  p.w = (const bf16_t*)wp;
  p.y = y;
  p.bias = bias;
  p.stats = stats;
  if (fin) { p.fin = *fin; p.stats = fin->rows; }
  p.N = d->N; p.H = d->H; p.W = d->W;
  p.Cin = d->Cin; p.Cout = d->Cout; p.x_ld = d->x_ld; p.y_ld = d->y_ld;
  p.OHf = (d->H + 2 * pad - d->ksize) / d->stride + 1;
  p.OWf = (d->W + 2 * pad - d->ksize) / d->stride + 1;
  if (d->ksize == 1 && d->stride == 1 && ((size_t)d->N * d->H * d->W) % TW == 0) {
    // pointwise conv: pixels are independent -> view the tensor as one [rows x 16] image
    const int rows = (int)((size_t)d->N * d->H * d->W / TW);
    p.N = 1; p.H = rows; p.W = TW; p.OHf = rows; p.OWf = TW;
  }
  p.OHt = p.OHf; p.OWt = p.OWf;
  p.so = 1; p.oy_off = 0; p.ox_off = 0;
  p.si = d->stride; p.iy_off = -pad; p.ix_off = -pad;
  p.ITH = (TH - 1) * p.si + d->ksize;
  p.ITW = (TW - 1) * p.si + d->ksize;
  p.ntaps = d->ksize * d->ksize;
  for (int kh = 0; kh < d->ksize; ++kh)
    for (int kw = 0; kw < d->ksize; ++kw) {
      const int t = kh * d->ksize + kw;
      p.tap_dy[t] = (signed char)kh; p.tap_dx[t] = (signed char)kw; p.tap_w[t] = (signed char)t;
    }
  p.accumulate = 0;
  p.w_ld = d->Cin;
  p.wtap_stride = d->Cout * d->Cin;
  set_grid(p);
  int BN, CK;
  pick_tiles(p.Cin, p.Cout, p.si, d->ksize, d->y_f32 != 0, &BN, &CK);
  const bool f32 = d->y_f32 != 0;
  {
    char lab[64];
    snprintf(lab, sizeof(lab), "conv_mfma_fwd<BN%d,CK%d>%s", BN, CK, f32 ? "f32out" : "");
    const double M = (double)p.N * p.OHf * p.OWf;
    annotate(lab, 2.0 * M * d->Cout * d->Cin * d->ksize * d->ksize, M * (d->Cout * (f32 ? 4.0 : 2.0)) + (double)d->N * d->H * d->W * d->Cin * 2.0);
  }
  return submit(stream, [=](hipStream_t s) { return f32 ? launch_bn<true>(p, BN, CK, s) : launch_bn<false>(p, BN, CK, s); });
}

// dx[N,H,W,Cin] = sum_taps dy[...] * w ; weights packed [tap][Cin][Cout_p8]
int conv_mfma_dgrad(const plyolo_conv_desc* d, const void* dy, const void* wpd, void* dx, int accumulate,
                    void* stream) {
  const int pad = (d->ksize - 1) / 2;
  const int OH = (d->H + 2 * pad - d->ksize) / d->stride + 1;
  const int OW = (d->W + 2 * pad - d->ksize) / d->stride + 1;
  const int Kc = (d->Cout + 7) & ~7;  // contraction length (= packed Cout)
  ConvP b{};
  b.x = (const bf16_t*)dy;
  b.w = (const bf16_t*)wpd;
  b.y = dx;
  b.bias = nullptr; b.stats = nullptr;
  b.N = d->N; b.H = OH; b.W = OW;
  b.Cin = Kc; b.Cout = d->Cin; b.x_ld = d->y_ld; b.y_ld = d->x_ld;
  b.OHf = d->H; b.OWf = d->W;
  b.accumulate = accumulate;
  b.w_ld = Kc;
  b.wtap_stride = d->Cin * Kc;
  b.si = 1;
  int rc = 0;
  if (d->stride == 1) {
    ConvP p = b;
    if (d->ksize == 1 && ((size_t)d->N * d->H * d->W) % TW == 0) {
      const int rows = (int)((size_t)d->N * d->H * d->W / TW);
      p.N = 1; p.H = rows; p.W = TW; p.OHf = rows; p.OWf = TW;
    }
    p.OHt = p.OHf; p.OWt = p.OWf;
    p.so = 1; p.oy_off = 0; p.ox_off = 0;
    p.iy_off = -pad; p.ix_off = -pad;
    p.ITH = (TH - 1) + d->ksize; p.ITW = (TW - 1) + d->ksize;
    p.ntaps = d->ksize * d->ksize;
    for (int dy_ = 0; dy_ < d->ksize; ++dy_)
      for (int dx_ = 0; dx_ < d->ksize; ++dx_) {
        const int t = dy_ * d->ksize + dx_;
        // dX[y,x] += dZ[y-pad+dy, x-pad+dx] * W[kh = k-1-dy][kw = k-1-dx]
        p.tap_dy[t] = (signed char)dy_; p.tap_dx[t] = (signed char)dx_;
        p.tap_w[t] = (signed char)((d->ksize - 1 - dy_) * d->ksize + (d->ksize - 1 - dx_));
      }
    set_grid(p);
    int BN, CK;
    pick_tiles(p.Cin, p.Cout, 1, d->ksize, false, &BN, &CK);
    {
      char lab[64];
      snprintf(lab, sizeof(lab), "conv_mfma_dgrad<BN%d,CK%d>", BN, CK);
      const double Mo = (double)d->N * OH * OW, Mi = (double)d->N * d->H * d->W;
      annotate(lab, 2.0 * Mo * d->Cout * d->Cin * d->ksize * d->ksize, (Mo * Kc + Mi * d->Cin * (accumulate ? 2.0 : 1.0)) * 2.0);
    }
    rc = submit(stream, [=](hipStream_t s) { return launch_bn<false>(p, BN, CK, s); });
    return rc;
  }
  // stride 2 (ksize 3 pad 1, or ksize 1): one launch per output parity class
  for (int py = 0; py < 2 && rc == 0; ++py)
    for (int px = 0; px < 2 && rc == 0; ++px) {
      ConvP p = b;
      p.OHt = (d->H - py + 1) / 2; p.OWt = (d->W - px + 1) / 2;
      if (p.OHt <= 0 || p.OWt <= 0) continue;
      p.so = 2; p.oy_off = py; p.ox_off = px;
      p.iy_off = 0; p.ix_off = 0;
      // taps reaching rows y = 2a+py:  (y + pad - kh) even  ->  oh = (y + pad - kh)/2 = a + dy
      int ky[2], dyv[2], nky = 0, kx[2], dxv[2], nkx = 0;
      for (int kh = 0; kh < d->ksize; ++kh)
        if (((py + pad - kh) & 1) == 0) { ky[nky] = kh; dyv[nky] = (py + pad - kh) / 2; ++nky; }
      for (int kw = 0; kw < d->ksize; ++kw)
        if (((px + pad - kw) & 1) == 0) { kx[nkx] = kw; dxv[nkx] = (px + pad - kw) / 2; ++nkx; }
      if (nky == 0 || nkx == 0) {
        // (ksize 1, odd class): no tap reaches this class -> gradient is zero there
        if (!accumulate) {
          // handled by the caller zero-filling dx for ksize-1 stride-2 convs (not used by the YOLOX graphs)
        }
        continue;
      }
      int miny = 9, maxy = -9, minx = 9, maxx = -9;
      for (int i = 0; i < nky; ++i) { miny = dyv[i] < miny ? dyv[i] : miny; maxy = dyv[i] > maxy ? dyv[i] : maxy; }
      for (int i = 0; i < nkx; ++i) { minx = dxv[i] < minx ? dxv[i] : minx; maxx = dxv[i] > maxx ? dxv[i] : maxx; }
      p.iy_off = miny; p.ix_off = minx;
      p.ITH = (TH - 1) + (maxy - miny + 1); p.ITW = (TW - 1) + (maxx - minx + 1);
      p.ntaps = 0;
      for (int i = 0; i < nky; ++i)
        for (int j = 0; j < nkx; ++j) {
          p.tap_dy[p.ntaps] = (signed char)(dyv[i] - miny);
          p.tap_dx[p.ntaps] = (signed char)(dxv[j] - minx);
          p.tap_w[p.ntaps] = (signed char)(ky[i] * d->ksize + kx[j]);
          ++p.ntaps;
        }
      set_grid(p);
      int BN, CK;
      pick_tiles(p.Cin, p.Cout, 1, 2, false, &BN, &CK);
      {
        char lab[64];
        snprintf(lab, sizeof(lab), "conv_mfma_dgrad_s2<BN%d,CK%d>", BN, CK);
        const double Mo = (double)d->N * OH * OW, Mi = (double)d->N * d->H * d->W;
        // one quarter of the layer's algorithmic work per parity-class launch
        annotate(lab, 0.25 * 2.0 * Mo * d->Cout * d->Cin * d->ksize * d->ksize, 0.25 * (Mo * Kc + Mi * d->Cin * (accumulate ? 2.0 : 1.0)) * 2.0);
      }
      rc = submit(stream, [=](hipStream_t s) { return launch_bn<false>(p, BN, CK, s); });
    }
  return rc;
}

}  // namespace plyolo
