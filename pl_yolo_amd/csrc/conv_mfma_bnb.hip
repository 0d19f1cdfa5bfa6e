// BNB instances of the bf16 MFMA convolution (conv_mfma_body.h): the 3x3 stride-1 data gradient of a BaseConv unit with the unit's
// own BatchNorm + SiLU backward in its halo loader (plyolo_conv2d_dgrad_bn on a 3x3 unit).  What autograd does for
// act(bn(conv3x3(x))) between the gradient of the activated output and dx (reference models/layers/network_blocks.py:18-40): the
// halo tile is requested as (dout, z) vector pairs and staged as dz = A*du + B*z + Cc -- the arithmetic of bn_act_bwd_dz (bn.hip),
// bit for bit; padding pixels stay zero -- so the bn_act_bwd_dz pass of the unit (read dout, read z, write dz) and its launch leave
// the data-gradient chain.  The workgroups of output block 0 write the dz of their own tile's pixels once, for the weight gradient.
// Every instance also carries the RED store loop (bnred.h; red.n == 0: nothing folded).  Own translation unit: co-compiled template
// instances perturb each other's code (see conv_mfma_body.h).
#include "conv_mfma_body.h"

namespace {

constexpr size_t RED_LDS = 16384;   // bnred_flush scratch (aliases the epilogue staging)

// OCC = waves per SIMD the register allocation is held to: the 8-row 128-channel tile needs ~190 VGPRs with the second operand of the
// loader in flight (two waves per SIMD); held to the three waves of the plain data gradient (168) it spills its loader offsets
// (PLYOLO_BNB_OCC=3, A/B)
template <int BN, int TH, int OCC>
__global__ __launch_bounds__(256, OCC) void conv_mfma_bnb16_kernel(const ConvP p) {
  conv_mfma_body<BN, 32, TH, false, 0, true, false, true, false, true, true>(p, (int)blockIdx.x, (int)gridDim.x);
}

template <int BN, int TH>
hipError_t launch_bnb_mf16(ConvP p, hipStream_t s) {
  constexpr int CK = 32, BM = TH * TW, WN = BN / 32, WM = 4 / WN;
  constexpr int ROWB = CK * 2 + 32, SROW = BN * 2 + 16;
  p.rowp = (p.ITW * ROWB + 255) & ~255;
  p.bufsz = p.ITH * p.rowp;
  p.btab = 2 * p.bufsz;
  const size_t lds_main = 2 * (size_t)p.bufsz + (size_t)p.Cin * 20, lds_epi = (size_t)BM * SROW + WM * 2 * BN * 4;
  size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
  lds = lds > RED_LDS ? lds : RED_LDS;
  static const int occ3 = getenv("PLYOLO_BNB_OCC") ? atoi(getenv("PLYOLO_BNB_OCC")) == 3 : 0;
  auto kern = conv_mfma_bnb16_kernel<BN, TH, 2>;
  if constexpr (BN == 128 && TH == 8) {
    if (occ3) kern = conv_mfma_bnb16_kernel<BN, TH, 3>;
  }
  if (hipError_t e = plyolo::ensure_dynamic_lds((const void*)kern, lds); e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(p.nmb, (p.Cout + BN - 1) / BN), dim3(256), lds, s, p);
  return hipGetLastError();
}

// 32x32x16 instances (the 32-channel data gradients of the 160x160 / 320x320 maps): one chunk, or double-buffered chunks
template <int BN, int CK, int TH>
hipError_t launch_bnb_plain(ConvP p, hipStream_t s) {
  constexpr int BM = TH * TW, WN = BN / 32, WM = 4 / WN;
  constexpr int ROWB = CK * 2 + 16, SROW = BN * 2 + 16;
  p.rowp = (p.ITW * ROWB + 255) & ~255;
  size_t lds_main = (size_t)p.ITH * p.rowp;
  const size_t lds_epi = (size_t)BM * SROW + WM * 2 * BN * 4;
  auto kern = conv_mfma_kernel<BN, CK, TH, false, 0, false, false, false, false, true, true>;
  if (p.db && p.Cin > CK) {
    p.bufsz = p.ITH * p.rowp;
    lds_main = 2 * (size_t)p.bufsz;
    kern = conv_mfma_kernel<BN, CK, TH, false, 0, true, false, false, false, true, true>;
  }
  p.btab = (int)lds_main;
  lds_main += (size_t)p.Cin * 20;
  size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
  lds = lds > RED_LDS ? lds : RED_LDS;
  if (hipError_t e = plyolo::ensure_dynamic_lds((const void*)kern, lds); e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(p.nmb, (p.Cout + BN - 1) / BN), dim3(256), lds, s, p);
  return hipGetLastError();
}

}  // namespace

namespace plyolo {

// which (BN, CK, TH, chunk count) tile configurations have a BNB instance (3x3 stride 1)
int conv_mfma_bnb_has(int BN, int CK, int TH, int multi_chunk) {
  if (CK != 32) return 0;
  if ((TH == 8 || TH == 4) && (BN == 128 || BN == 64)) return multi_chunk ? 1 : 0;   // v_mfma_f32_16x16x32_bf16 tiles: double-buffered chunks
  return (TH == 16 && BN == 32) ? 1 : 0;
}

// `convp`: a ConvP of a 3x3 stride-1 data gradient with its BNB fields filled and its tiles applied
hipError_t conv_mfma_launch_bnb(const void* convp, int BN, int CK, int TH, hipStream_t s) {
  const ConvP& p = *(const ConvP*)convp;
  if (CK == 32 && TH == 8 && BN == 128) return launch_bnb_mf16<128, 8>(p, s);
  if (CK == 32 && TH == 8 && BN == 64) return launch_bnb_mf16<64, 8>(p, s);
  if (CK == 32 && TH == 4 && BN == 128) return launch_bnb_mf16<128, 4>(p, s);
  if (CK == 32 && TH == 4 && BN == 64) return launch_bnb_mf16<64, 4>(p, s);
  if (CK == 32 && TH == 16 && BN == 32) return launch_bnb_plain<32, 32, 16>(p, s);
  return hipErrorInvalidValue;
}

}  // namespace plyolo
