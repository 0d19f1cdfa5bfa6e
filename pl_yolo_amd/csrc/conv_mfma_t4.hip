// 4-row-tile instances of the 3x3 stride-1 bf16 MFMA convolution (conv_mfma_body.h, v_mfma_f32_16x16x32_bf16 path, 32-channel
// double-buffered chunks) for the SMALL maps: a 40x40 / 20x20 layer at batch 32 is 200-500 workgroups of 8x16 positions on 256 CUs
// (one to two per CU, the last ones alone on theirs); 4x16 tiles double the count for the same work.  Forward and data gradient,
// the latter also as RED instance (bnred.h).  Own translation unit: see conv_mfma_body.h.
#include "conv_mfma_body.h"

namespace {

template <int BN, bool RED>
hipError_t launch_t4_inst(ConvP p, hipStream_t s) {
  constexpr int CK = 32, TH = 4, BM = TH * TW, WN = BN / 32, WM = 4 / WN;
  constexpr int ROWB = CK * 2 + 32, SROW = BN * 2 + 16;
  p.rowp = (p.ITW * ROWB + 255) & ~255;
  p.bufsz = p.ITH * p.rowp;
  const size_t lds_main = 2 * (size_t)p.bufsz, lds_epi = (size_t)BM * SROW + WM * 2 * BN * 4;
  size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
  if (RED) lds = lds > 16384 ? lds : 16384;
  auto kern = conv_mfma_kernel<BN, CK, TH, false, 0, true, false, true, false, RED>;
  if (hipError_t e = plyolo::ensure_dynamic_lds((const void*)kern, lds); e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(p.nmb, (p.Cout + BN - 1) / BN), dim3(256), lds, s, p);
  return hipGetLastError();
}

}  // namespace

namespace plyolo {

// `convp`: a ConvP prepared for 4-row tiles (apply_tiles(p, 3, 3, 4)), 3x3 stride 1, Cin > 32, bf16 output
hipError_t conv_mfma_launch_t4(const void* convp, int BN, int red, hipStream_t s) {
  const ConvP& p = *(const ConvP*)convp;
  if (BN == 128) return red ? launch_t4_inst<128, true>(p, s) : launch_t4_inst<128, false>(p, s);
  if (BN == 64) return red ? launch_t4_inst<64, true>(p, s) : launch_t4_inst<64, false>(p, s);
  return hipErrorInvalidValue;
}

}  // namespace plyolo
