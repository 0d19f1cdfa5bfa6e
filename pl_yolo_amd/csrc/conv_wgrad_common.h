// Shared declarations of the bf16 MFMA weight-gradient kernels (conv_wgrad_mfma.hip, conv_wgrad3r.hip).
#pragma once
#include "common.h"

namespace {

constexpr int TH = 8, TW = 16;

struct WgP {
  const bf16_t* x;
  const bf16_t* dy;
  float* dw;
  int N, H, W, OH, OW, Cin, Cout, x_ld, dy_ld;
  int si, pad, ITH, ITW;
  int tiles_y, tiles_x, ntiles;
  const float* pre;  // lazy input (plyolo_conv_desc::x_coef): X tiles are staged as act(x * pre[c] + pre[pre_ld + c]), padding stays zero
  int pre_ld, pre_act;
  int nci;  // number of ci tiles
  int nslabt;  // slab tiles per spatial split = nco * nci
  int S;       // spatial splits; the grid is 1-D: S * nslabt workgroups
  int xcd;     // 1: slab tiles of one split adjacent in an XCD-contiguous order (PLYOLO_WG_XCD, default), 0: the round-1 order
  int ablate;  // diagnostics (PLYOLO_ABLATE_WG): 1 skip atomics, 2 skip tile loads after the first, 4 skip MFMA, 8 force S
  // BNB instances (plyolo_conv2d_wgrad_bn): `dy` is the gradient of the unit's ACTIVATED output; the loader forms
  // dz = A*du + B*z + Cc (bn.hip: bn_act_bwd_dz, bit for bit) from it and the unit's raw conv output z on the way into LDS
  const bf16_t* bz;
  int bz_ld;
  const float* bcoef;     // (scale | shift | mean | invstd) [4][Cout]
  const double* bslots;   // [PLYOLO_STAT_SLOTS][2][Cout]
  const float* bgamma;
  float *bdgamma, *bdbeta;
  double bcount;
};

DEVINL float wg_silu_grad(float u) {      // == act_grad<false>(u, PLYOLO_ACT_SILU) (bn.hip) == pw_act_grad (conv_pw.hip)
  const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-u));
  return s * (1.0f + u * (1.0f - s));
}

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

DEVINL s16x4 tr_read(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
}

// LDS row pitch of a pixel row of `ch` channels.  Rows of a multiple of 128 bytes need a pad that keeps the four pixel rows of a transposed
// read on distinct banks: 64 bytes did (rounds 1-4); 16 bytes do as well (4 * pitch = 64 mod 256, the rule of conv_pw_bwd.hip) and shrink
// the two tiles of the 64 x 64 3x3 variant from 126 KB to 97 KB -- room for a data-gradient workgroup of the main lane on the same CU
// (PLYOLO_WG_PAD64 at build time restores the old pitch)
#ifdef PLYOLO_WG_PAD64
constexpr int pitch_for(int ch) { return ch * 2 + ((ch * 2) % 128 == 0 ? 64 : 0); }
#else
constexpr int pitch_for(int ch) { return ch * 2 + ((ch * 2) % 128 == 0 ? 16 : 0); }
#endif

}  // namespace
