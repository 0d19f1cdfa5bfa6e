// RED instances of the bf16 MFMA convolution (conv_mfma_body.h): data gradients whose store loop also folds the BatchNorm-backward
// reduction of the upstream unit(s) (plyolo_bn_red, bnred.h) -- the shapes the 3x3 data gradients of a CSPDarknet / PAFPN /
// decoupled head run: 8-row tiles with 32-channel double-buffered chunks on v_mfma_f32_16x16x32_bf16 (64 / 128 output channels),
// the 16-row tile of the 32-channel layers, and the four-parity-class launch of the stride-2 layers.  Own translation unit:
// co-compiled template instances perturb each other's code (see conv_mfma_body.h).
#include "conv_mfma_body.h"

namespace {

constexpr size_t RED_LDS = 16384;   // bnred_flush scratch (aliases the epilogue staging)

template <int BN>
hipError_t launch_red_mf16(ConvP p, hipStream_t s) {
  constexpr int CK = 32, TH = 8, BM = TH * TW, WN = BN / 32, WM = 4 / WN;
  constexpr int ROWB = CK * 2 + 32, SROW = BN * 2 + 16;
  p.rowp = (p.ITW * ROWB + 255) & ~255;
  p.bufsz = p.ITH * p.rowp;
  const size_t lds_main = 2 * (size_t)p.bufsz, lds_epi = (size_t)BM * SROW + WM * 2 * BN * 4;
  size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
  lds = lds > RED_LDS ? lds : RED_LDS;
  auto kern = conv_mfma_kernel<BN, CK, TH, false, 0, true, false, true, false, true>;
  if (hipError_t e = plyolo::ensure_dynamic_lds((const void*)kern, lds); e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(p.nmb, (p.Cout + BN - 1) / BN), dim3(256), lds, s, p);
  return hipGetLastError();
}

// single-buffered / double-buffered 32x32x16 instances (the 32-output-channel layers)
template <int BN, int CK, int TH>
hipError_t launch_red_plain(ConvP p, hipStream_t s) {
  constexpr int BM = TH * TW, WN = BN / 32, WM = 4 / WN;
  constexpr int ROWB = CK * 2 + 16, SROW = BN * 2 + 16;
  p.rowp = p.ITW * ROWB;
  if (p.si == 1) p.rowp = (p.rowp + 255) & ~255;
  size_t lds_main = (size_t)p.ITH * p.rowp;
  const size_t lds_epi = (size_t)BM * SROW + WM * 2 * BN * 4;
  auto kern = conv_mfma_kernel<BN, CK, TH, false, 0, false, false, false, false, true>;
  if (p.db && p.si == 1 && p.Cin > CK) {
    p.bufsz = p.ITH * p.rowp;
    lds_main = 2 * (size_t)p.bufsz;
    kern = conv_mfma_kernel<BN, CK, TH, false, 0, true, false, false, false, true>;
  }
  size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
  lds = lds > RED_LDS ? lds : RED_LDS;
  if (hipError_t e = plyolo::ensure_dynamic_lds((const void*)kern, lds); e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(p.nmb, (p.Cout + BN - 1) / BN), dim3(256), lds, s, p);
  return hipGetLastError();
}

template <int BN, int CK, int TH>
hipError_t launch_red_jobs(ConvJobs jobs, hipStream_t s) {
  constexpr int BM = TH * TW, WN = BN / 32, WM = 4 / WN;
  constexpr int ROWB = CK * 2 + 16, SROW = BN * 2 + 16;
  size_t lds = (size_t)BM * SROW + WM * 2 * BN * 4;
  lds = lds > RED_LDS ? lds : RED_LDS;
  int total = 0, ny = 1;
  bool db = true;
  for (int j = 0; j < jobs.n; ++j) {
    ConvP& p = jobs.c[j];
    p.rowp = (p.ITW * ROWB + 255) & ~255;
    db = db && p.db && p.si == 1 && p.Cin > CK;
    jobs.start[j] = total;
    total += p.nmb;
    ny = (p.Cout + BN - 1) / BN;
  }
  jobs.start[jobs.n] = total;
  for (int j = 0; j < jobs.n; ++j) {
    ConvP& p = jobs.c[j];
    p.bufsz = p.ITH * p.rowp;
    const size_t m = (db ? 2 : 1) * (size_t)p.bufsz;
    lds = m > lds ? m : lds;
  }
  auto kern = db ? conv_mfma_jobs_kernel<BN, CK, TH, true, true> : conv_mfma_jobs_kernel<BN, CK, TH, false, true>;
  if (hipError_t e = plyolo::ensure_dynamic_lds((const void*)kern, lds); e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(total, ny), dim3(256), lds, s, jobs);
  return hipGetLastError();
}

}  // namespace

namespace plyolo {

// which (BN, CK, TH) tile configurations have a RED instance
int conv_mfma_red_has(int BN, int CK, int TH, int jobs) {
  if (CK != 32) return 0;
  if (jobs) return (TH == 8 && (BN == 128 || BN == 64)) || (TH == 16 && BN == 32);
  return (TH == 8 && (BN == 128 || BN == 64)) || (TH == 16 && BN == 32);
}

hipError_t conv_mfma_launch_red(const void* convp, int BN, int CK, int TH, hipStream_t s) {
  const ConvP& p = *(const ConvP*)convp;
  if (TH == 8 && CK == 32 && (BN == 128 || BN == 64)) {
    if (p.db && p.si == 1 && p.Cin > CK) return BN == 128 ? launch_red_mf16<128>(p, s) : launch_red_mf16<64>(p, s);
    return BN == 128 ? launch_red_plain<128, 32, 8>(p, s) : launch_red_plain<64, 32, 8>(p, s);
  }
  if (TH == 16 && CK == 32 && BN == 32) return launch_red_plain<32, 32, 16>(p, s);
  return hipErrorInvalidValue;
}

hipError_t conv_mfma_launch_jobs_red(const void* jobsp, int BN, int CK, int TH, hipStream_t s) {
  const ConvJobs& jobs = *(const ConvJobs*)jobsp;
  if (TH == 8 && BN == 128) return launch_red_jobs<128, 32, 8>(jobs, s);
  if (TH == 8 && BN == 64) return launch_red_jobs<64, 32, 8>(jobs, s);
  if (TH == 16 && BN == 32) return launch_red_jobs<32, 32, 16>(jobs, s);
  return hipErrorInvalidValue;
}

}  // namespace plyolo
