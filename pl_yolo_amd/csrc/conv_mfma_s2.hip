// 3x3 stride-2 forward instances of the bf16 MFMA convolution (conv_mfma_body.h, S2): the five stride-2 layers of a CSPDarknet /
// PAFPN (reference models/backbones/darknet_csp.py: dark2..dark5 BaseConv(…, 3, 2); models/necks/pafpn_csp.py: bu_conv1 / bu_conv2).
// Through the generic path (si = 2 at run time) those layers ran at 280-440 TFLOP/s against 850-1070 for the stride-1 3x3 layers
// of the same maps: their 17 x 33 halo tile took the index-arithmetic loader (two exposed memory round trips per 32-channel chunk,
// no double buffer) and every fragment read hit LDS 2-way conflicted (pixels two columns apart).  Own translation unit: co-compiled
// template instances perturb each other's code (see conv_mfma_body.h).
#define PLYOLO_CONV_PD 2   // forward instances: weight fragments two taps ahead (see the A/B in profiles/r04_ab_fusions.txt)
#include "conv_mfma_body.h"

namespace {

template <int BN>
hipError_t launch_s2_inst(ConvP p, hipStream_t s) {
  constexpr int CK = 32, TH = 4, BM = TH * TW, WN = BN / 32, WM = 4 / WN;
  constexpr int ROWB = CK * 2 + 16, SROW = BN * 2 + 16;
  p.rowp = (33 * ROWB + 255) & ~255;
  p.bufsz = p.ITH * p.rowp;
  p.db = 1;
  const size_t lds_main = 2 * (size_t)p.bufsz, lds_epi = (size_t)BM * SROW + WM * 2 * BN * 4;
  const size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
  auto kern = conv_mfma_kernel<BN, CK, TH, false, 0, true, false, false, true>;
  if (hipError_t e = plyolo::ensure_dynamic_lds((const void*)kern, lds); e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(p.nmb, (p.Cout + BN - 1) / BN), dim3(256), lds, s, p);
  return hipGetLastError();
}

// ragged column blocks (see conv_mfma_rag.hip): full 128-channel blocks + a 64-channel block for the last 32 / 64 channels of a 160- / 320-channel
// layer (the 4 x 16 tile has two 32-pixel fragments: 64 channels, two waves by two, is its narrowest instance)
__global__ __launch_bounds__(256, 2) void conv_mfma_s2_rag_kernel(const ConvP p, const int nfull) {
  if ((int)blockIdx.y < nfull)
    conv_mfma_body<128, 32, 4, false, 0, true, false, false, true>(p, (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.y * 4);
  else
    conv_mfma_body<64, 32, 4, false, 0, true, false, false, true>(p, (int)blockIdx.x, (int)gridDim.x, nfull * 4);
}

hipError_t launch_s2_rag(ConvP p, hipStream_t s) {
  constexpr int CK = 32, TH = 4, BM = TH * TW;
  constexpr int ROWB = CK * 2 + 16;
  p.rowp = (33 * ROWB + 255) & ~255;
  p.bufsz = p.ITH * p.rowp;
  p.db = 1;
  const size_t lds_main = 2 * (size_t)p.bufsz, lds_epi = (size_t)BM * (128 * 2 + 16) + 1 * 2 * 128 * 4;   // (the 128-channel block's staging is the larger one)
  const size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
  auto kern = conv_mfma_s2_rag_kernel;
  if (hipError_t e = plyolo::ensure_dynamic_lds((const void*)kern, lds); e != hipSuccess) return e;
  const int nfull = p.Cout / 128;
  hipLaunchKernelGGL(kern, dim3(p.nmb, nfull + 1), dim3(256), lds, s, p, nfull);
  return hipGetLastError();
}

}  // namespace

namespace plyolo {

// 1: conv_mfma_launch_s2 runs this layer with a ragged last block (Cout = 128 * n + rem, n >= 1, 0 < rem <= 64; PLYOLO_RAG=0: never)
int conv_mfma_s2_ragged(int Cout) {
  const int on = getenv("PLYOLO_RAG") ? atoi(getenv("PLYOLO_RAG")) : 1;
  const int rem = Cout % 128;
  return on && Cout > 128 && rem > 0 && rem <= 64;
}

// `convp`: a ConvP prepared for 4-row tiles (ITH 9, ITW 33, si 2); `ragged`: conv_mfma_s2_ragged(Cout), evaluated by the caller where
// the launch is recorded
hipError_t conv_mfma_launch_s2(const void* convp, int BN, int ragged, hipStream_t s) {
  const ConvP& p = *(const ConvP*)convp;
  if (BN == 128 && ragged) return launch_s2_rag(p, s);
  if (BN == 128) return launch_s2_inst<128>(p, s);
  if (BN == 64) return launch_s2_inst<64>(p, s);
  return hipErrorInvalidValue;
}

}  // namespace plyolo
