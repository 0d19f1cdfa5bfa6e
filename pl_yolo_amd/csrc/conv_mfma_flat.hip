// FLAT instances of the 3x3 stride-1 bf16 MFMA convolution (conv_mfma_body.h, v_mfma_f32_16x16x32_bf16 path, 32-channel double-buffered
// chunks) for the 20- and 40-wide maps: a tile is four WHOLE image rows (80 / 160 pixels), its 16-pixel fragments run over the row-major
// pixel list.  The rectangular tiles cover a 20-wide map at 62 % (4 x 16 positions per tile, two tile columns of which the second holds 4
// real ones) and a 40-wide one at 83 % (8 x 16, three tile columns); here every staged halo pixel and every MFMA row is a real pixel.
// Forward and data gradient, the latter also as RED instance (bnred.h).  Own translation unit: see conv_mfma_body.h.
#include "conv_mfma_body.h"

namespace {

// TH = pixels per tile / 16: 10 for the 40-wide maps (4 rows x 40), 5 for the 20-wide ones (4 rows x 20)
template <int TH, bool RED>
hipError_t launch_flat_inst(ConvP p, hipStream_t s) {
  constexpr int BN = 128, CK = 32, BM = TH * TW, WN = BN / 32, WM = 4 / WN;
  constexpr int ROWB = CK * 2 + 32, SROW = BN * 2 + 16;
  p.rowp = (p.ITW * ROWB + 255) & ~255;
  p.bufsz = p.ITH * p.rowp;
  const size_t lds_main = 2 * (size_t)p.bufsz, lds_epi = (size_t)BM * SROW + WM * 2 * BN * 4;
  size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
  if (RED) lds = lds > 16384 ? lds : 16384;
  auto kern = conv_mfma_kernel<BN, CK, TH, false, 0, true, false, true, false, RED, false, true>;
  if (hipError_t e = plyolo::ensure_dynamic_lds((const void*)kern, lds); e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(p.nmb, (p.Cout + BN - 1) / BN), dim3(256), lds, s, p);
  return hipGetLastError();
}

}  // namespace

namespace plyolo {

// `convp`: a ConvP prepared for FLAT tiles (tw = OWt in {20, 40}, trows = 4, ITH = 6, ITW = tw + 2, tiles_x = 1), 3x3 stride 1,
// Cin > 32, more than 64 output channels, bf16 output
hipError_t conv_mfma_launch_flat(const void* convp, int red, hipStream_t s) {
  const ConvP& p = *(const ConvP*)convp;
  const int px = p.tw * p.trows;      // pixels per tile: 160 (4 x 40) or 80 (4 x 20, 2 x 40)
  if (px == 160) return red ? launch_flat_inst<10, true>(p, s) : launch_flat_inst<10, false>(p, s);
  if (px == 80) return red ? launch_flat_inst<5, true>(p, s) : launch_flat_inst<5, false>(p, s);
  return hipErrorInvalidValue;
}

}  // namespace plyolo
