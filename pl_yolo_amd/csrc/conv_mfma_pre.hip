// Lazy-input (PRE) instantiations of the bf16 MFMA convolution: the halo tile is staged as act(x * scale + shift) of the
// producing layer's BatchNorm (plyolo_conv_desc::x_coef).  Own translation unit -- see conv_mfma_body.h.
#include "conv_mfma_body.h"

namespace {

// lazy-input instances: 16- and 32-channel chunks only (pick_tiles never chooses wider chunks for them)
template <bool OUT_F32>
hipError_t launch_bn_pre(const ConvP& p, int BN, int CK, int TH, hipStream_t s) {
#define PLY_PCASE(bn, ck)                                                         \
  if (BN == bn && CK == ck) {                                                     \
    if (TH == 16) return launch_inst<bn, ck, 16, OUT_F32, true>(p, s);            \
    return launch_inst<bn, ck, 8, OUT_F32, true>(p, s);                           \
  }
  PLY_PCASE(32, 16) PLY_PCASE(32, 32) PLY_PCASE(64, 16) PLY_PCASE(64, 32)
  if (!OUT_F32) { PLY_PCASE(128, 16) PLY_PCASE(128, 32) }
#undef PLY_PCASE
  return hipErrorInvalidValue;
}

}  // namespace

namespace plyolo {

hipError_t conv_mfma_launch_pre(const void* convp, int BN, int CK, int TH, bool out_f32, hipStream_t s) {
  const ConvP& p = *(const ConvP*)convp;
  return out_f32 ? launch_bn_pre<true>(p, BN, CK, TH, s) : launch_bn_pre<false>(p, BN, CK, TH, s);
}

}  // namespace plyolo
