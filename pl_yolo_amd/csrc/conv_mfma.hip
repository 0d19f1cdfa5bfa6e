// bf16 MFMA direct convolution for gfx950 (forward and data-gradient).
//
// Replaces nn.Conv2d inside BaseConv (reference models/layers/network_blocks.py:18-26)
// and what ATen's convolution_backward computes for its input.
//
// Formulation (im2col-free): one workgroup (4 waves) owns a TH x 16 tile (TH = 8, or 16 for the
// 32-output-channel layers) of output positions x BN output channels.  Per Cin-chunk (CK = 32 channels;
// stride-1 tiles double-buffer the chunk in LDS) the input HALO tile
// ((TH-1)*si+ext_y) x ((16-1)*si+ext_x) pixels is staged ONCE into LDS from coalesced
// NHWC rows; each filter tap is then a dense [TH*16 x CK] x [CK x BN] product on
// v_mfma_f32_32x32x16_bf16 whose A fragments are the tap-shifted rows of the halo tile (no
// data movement per tap) and whose B fragments come straight from the fragment-ordered
// weight pack in global memory (one coalesced 1-KiB load per fragment, prefetched a tap ahead).
// Epilogue (fused): per-channel sum / sum-of-squares partials for train-mode
// BatchNorm, optional bias, bf16 (or fp32) store through an LDS transpose so that
// global stores are whole 16-byte channel vectors; concat = strided store (y_ld).
//
// The same kernel serves dgrad: stride-1 dgrad is a forward conv with the tap table
// mirrored; stride-2 dgrad is four jobs of ONE launch, one per output parity class, each with
// the 1/2/2/4 taps that reach that class (so no zero-stuffing and no atomics).


// Every instance of THIS translation unit fetches its weight fragments two taps ahead: the forward launches and -- the same template
// instances -- the plain data gradients launched from here (conv_mfma_dgrad without a folded reduction: launch_bn<false>, launch_jobs).
// The RED / 4-row / stride-2-dgrad instances (conv_mfma_red.hip, conv_mfma_t4.hip, conv_s2d.hip), i.e. almost every data gradient of a
// training plan, keep one tap (measured in profiles/r04_ab_fusions.txt: two taps everywhere made the launches shorter and the step no shorter)
#define PLYOLO_CONV_PD 2
#include "conv_mfma_body.h"

namespace plyolo {
#ifdef PLYOLO_OPTIN   // opt-in paths (make OPTIN=1)
// lazy-input instances (conv_mfma_pre.hip); `convp` is a ConvP
hipError_t conv_mfma_launch_pre(const void* convp, int BN, int CK, int TH, bool out_f32, hipStream_t s);
// weights-stationary 3x3 stride-1 kernel (conv3ws.hip)
int conv3ws_accepts(int N, int H, int W, int Cin, int Cout, int x_ld, int y_ld, const void* y);
hipError_t conv3ws_launch(const void* x, const void* w, void* y, double* stats, int N, int H, int W, int Cin, int Cout, int x_ld, int y_ld,
                          int nkb, int nnb, int accumulate, unsigned long long taps_lo, unsigned taps_hi, hipStream_t s);
#else
static inline hipError_t conv_mfma_launch_pre(const void*, int, int, int, bool, hipStream_t) { return hipErrorNotSupported; }   // refused at the C ABI (api.hip: check_conv)
static inline int conv3ws_accepts(int, int, int, int, int, int, int, const void*) { return 0; }
static inline hipError_t conv3ws_launch(const void*, const void*, void*, double*, int, int, int, int, int, int, int, int, int, int, unsigned long long, unsigned,
                                        hipStream_t) { return hipErrorNotSupported; }
#endif
}

namespace plyolo {
hipError_t conv_mfma_launch_s2(const void* convp, int BN, int ragged, hipStream_t s);   // conv_mfma_s2.hip
int conv_mfma_s2_ragged(int Cout);
hipError_t conv_mfma_launch_t4(const void* convp, int BN, int red, hipStream_t s);   // conv_mfma_t4.hip
hipError_t conv_mfma_launch_flat(const void* convp, int red, hipStream_t s);         // conv_mfma_flat.hip
hipError_t conv_mfma_launch_rag(const void* convp, hipStream_t s);                   // conv_mfma_rag.hip
hipError_t conv_mfma_launch_n96(const void* convp, int CK, hipStream_t s);           // conv_mfma_rag.hip: one 96-channel block of three waves
hipError_t conv_mfma_launch_jobs_n96(const void* jobsp, hipStream_t s);
// conv_mfma_red.hip: data-gradient instances that fold the upstream BatchNorm-backward reduction into their store loop
// conv_s2d.hip: 3x3 stride-2 data gradient, the four parity classes on one staged tile
int conv_s2d_th(int BN);
hipError_t conv_s2d_launch(const void* convp, int BN, int red, hipStream_t s);
int conv_mfma_red_has(int BN, int CK, int TH, int jobs);
// conv_mfma_bnb.hip: 3x3 stride-1 data gradients with the unit's BatchNorm + SiLU backward in the halo loader
int conv_mfma_bnb_has(int BN, int CK, int TH, int multi_chunk);
hipError_t conv_mfma_launch_bnb(const void* convp, int BN, int CK, int TH, hipStream_t s);
hipError_t conv_mfma_launch_red(const void* convp, int BN, int CK, int TH, hipStream_t s);
hipError_t conv_mfma_launch_jobs_red(const void* jobsp, int BN, int CK, int TH, hipStream_t s);
}

namespace {

template <bool OUT_F32>
hipError_t launch_bn(const ConvP& p, int BN, int CK, int TH, hipStream_t s) {
  if (p.pre) return plyolo::conv_mfma_launch_pre(&p, BN, CK, TH, OUT_F32, s);
#define PLY_CASE(bn, ck)                                                      \
  if (BN == bn && CK == ck) {                                                 \
    if (TH == 16) return launch_inst<bn, ck, 16, OUT_F32>(p, s);              \
    return launch_inst<bn, ck, 8, OUT_F32>(p, s);                             \
  }
  PLY_CASE(32, 16) PLY_CASE(32, 32) PLY_CASE(32, 64)
  PLY_CASE(64, 16) PLY_CASE(64, 32) PLY_CASE(64, 64)
  if (!OUT_F32) {
    PLY_CASE(128, 16) PLY_CASE(128, 32) PLY_CASE(128, 64)
    // whole-K chunks for Cin = 128 on the small tile: no chunk boundary (barrier + exposed halo reload) at all
    if (BN == 128 && CK == 128 && TH == 8) return launch_inst<128, 128, 8, false>(p, s);
    if (BN == 64 && CK == 128 && TH == 8) return launch_inst<64, 128, 8, false>(p, s);
  }
#undef PLY_CASE
  return hipErrorInvalidValue;
}

template <int BN, int CK, int TH>
hipError_t launch_jobs_inst(ConvJobs jobs, hipStream_t s) {
  constexpr int BM = TH * TW, WN = BN / 32, WM = 4 / WN;
  constexpr int ROWB = CK * 2 + 16, SROW = BN * 2 + 16;
  size_t lds = (size_t)BM * SROW + WM * 2 * BN * 4;
  int total = 0, ny = 1;
  for (int j = 0; j < jobs.n; ++j) {
    ConvP& p = jobs.c[j];
    p.rowp = p.ITW * ROWB;
    if (p.si == 1) p.rowp = (p.rowp + 255) & ~255;
    const size_t m = (size_t)p.ITH * p.rowp;
    lds = m > lds ? m : lds;
    jobs.start[j] = total;
    total += p.nmb;
    ny = (p.Cout + BN - 1) / BN;
  }
  jobs.start[jobs.n] = total;
  auto kern = conv_mfma_jobs_kernel<BN, CK, TH>;
  // the parity classes of a stride-2 data gradient are stride-1 tiles: same double-buffered halo as the plain launches
  constexpr bool HAS_DB = (TH == 8 && (CK == 32 || CK == 64)) || (TH == 16 && CK == 32);
  if constexpr (HAS_DB) {
    bool db = true;
    for (int j = 0; j < jobs.n; ++j) db = db && jobs.c[j].db && jobs.c[j].si == 1 && jobs.c[j].Cin > CK && !jobs.c[j].ablate;
    if (db) {
      for (int j = 0; j < jobs.n; ++j) {
        ConvP& p = jobs.c[j];
        p.bufsz = p.ITH * p.rowp;
        const size_t m2 = 2 * (size_t)p.bufsz;
        lds = m2 > lds ? m2 : lds;
      }
      kern = conv_mfma_jobs_kernel<BN, CK, TH, true>;
    }
  }
  if (hipError_t e = plyolo::ensure_dynamic_lds((const void*)kern, lds); e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(total, ny), dim3(256), lds, s, jobs);
  return hipGetLastError();
}

// ragged column blocks for the four-job stride-2 data gradient (see conv_mfma_rag.hip): full 128-channel blocks + a 32- / 64-channel block
template <int REM>
__global__ __launch_bounds__(256, 2) void conv_mfma_jobs_rag_kernel(const ConvJobs jobs, const int nfull) {
  int j = 0;
  for (int k = 1; k < 4; ++k)
    if (k < jobs.n && (int)blockIdx.x >= jobs.start[k]) j = k;
  if ((int)blockIdx.y < nfull)
    conv_mfma_body<128, 32, 8, false, 0, true>(jobs.c[j], (int)blockIdx.x - jobs.start[j], jobs.start[j + 1] - jobs.start[j], (int)blockIdx.y * 4);
  else
    conv_mfma_body<REM, 32, 8, false, 0, true>(jobs.c[j], (int)blockIdx.x - jobs.start[j], jobs.start[j + 1] - jobs.start[j], nfull * 4);
}

// 1: launch_jobs runs these jobs with a ragged last block (128-channel blocks of 8-row tiles, 32-channel double-buffered chunks, every job
// double-buffer capable, Cout = 128 * n + rem with n >= 1 and 0 < rem <= 64; PLYOLO_RAG=0: never)
bool jobs_ragged(const ConvJobs& jobs, int BN, int CK, int TH) {
  const int on = getenv("PLYOLO_RAG") ? atoi(getenv("PLYOLO_RAG")) : 1;
  if (!on || BN != 128 || CK != 32 || TH != 8 || jobs.n < 1) return false;
  const int Cout = jobs.c[0].Cout, rem = Cout % 128;
  if (Cout <= 128 || rem == 0 || rem > 64) return false;
  for (int j = 0; j < jobs.n; ++j)
    if (!(jobs.c[j].db && jobs.c[j].si == 1 && jobs.c[j].Cin > CK && !jobs.c[j].ablate) || jobs.c[j].Cout != Cout) return false;
  return true;
}

template <int REM>
hipError_t launch_jobs_rag_inst(ConvJobs jobs, hipStream_t s) {
  constexpr int CK = 32, TH = 8, BM = TH * TW, ROWB = CK * 2 + 16;
  size_t lds = (size_t)BM * (128 * 2 + 16) + 1 * 2 * 128 * 4;       // epilogue staging of the 128-channel block (the larger one)
  int total = 0;
  for (int j = 0; j < jobs.n; ++j) {
    ConvP& p = jobs.c[j];
    p.rowp = (p.ITW * ROWB + 255) & ~255;
    p.bufsz = p.ITH * p.rowp;
    const size_t m2 = 2 * (size_t)p.bufsz;
    lds = m2 > lds ? m2 : lds;
    jobs.start[j] = total;
    total += p.nmb;
  }
  jobs.start[jobs.n] = total;
  auto kern = conv_mfma_jobs_rag_kernel<REM>;
  if (hipError_t e = plyolo::ensure_dynamic_lds((const void*)kern, lds); e != hipSuccess) return e;
  const int nfull = jobs.c[0].Cout / 128;
  hipLaunchKernelGGL(kern, dim3(total, nfull + 1), dim3(256), lds, s, jobs, nfull);
  return hipGetLastError();
}

// 1: launch_jobs runs these jobs as one 96-channel block of three waves (65 .. 96 output channels; conditions as jobs_ragged)
bool jobs_n96(const ConvJobs& jobs, int BN, int CK, int TH) {
  const int on = getenv("PLYOLO_RAG") ? atoi(getenv("PLYOLO_RAG")) : 1;
  if (!on || BN != 128 || CK != 32 || TH != 8 || jobs.n < 1) return false;
  const int Cout = jobs.c[0].Cout;
  if (Cout <= 64 || Cout > 96) return false;
  for (int j = 0; j < jobs.n; ++j)
    if (!(jobs.c[j].db && jobs.c[j].si == 1 && jobs.c[j].Cin > CK && !jobs.c[j].ablate) || jobs.c[j].Cout != Cout) return false;
  return true;
}

// `mode`: decided ONCE where the launch is recorded (PLYOLO_RAG is read there; a recorded launch replays what its label says):
// 0 whole 128-channel blocks, 1 ragged last block (jobs_ragged), 2 one 96-channel block (jobs_n96)
hipError_t launch_jobs(const ConvJobs& jobs, int BN, int CK, int TH, int mode, hipStream_t s) {
  if (mode == 2) return plyolo::conv_mfma_launch_jobs_n96(&jobs, s);
  if (mode == 1) return jobs.c[0].Cout % 128 <= 32 ? launch_jobs_rag_inst<32>(jobs, s) : launch_jobs_rag_inst<64>(jobs, s);
#define PLY_JCASE(bn, ck)                                                \
  if (BN == bn && CK == ck) {                                            \
    if (TH == 16) return launch_jobs_inst<bn, ck, 16>(jobs, s);          \
    return launch_jobs_inst<bn, ck, 8>(jobs, s);                         \
  }
  PLY_JCASE(32, 16) PLY_JCASE(32, 32) PLY_JCASE(32, 64)
  PLY_JCASE(64, 16) PLY_JCASE(64, 32) PLY_JCASE(64, 64)
  PLY_JCASE(128, 16) PLY_JCASE(128, 32) PLY_JCASE(128, 64)
  if (BN == 128 && CK == 128 && TH == 8) return launch_jobs_inst<128, 128, 8>(jobs, s);
  if (BN == 64 && CK == 128 && TH == 8) return launch_jobs_inst<64, 128, 8>(jobs, s);
#undef PLY_JCASE
  return hipErrorInvalidValue;
}

// tile choice: BN output channels x CK-channel chunks x TH output rows (x 16 columns)
void pick_tiles(const ConvP& p, int ext_y, bool out_f32, int* BN, int* CK, int* TH) {
  int bn = p.Cout > 64 ? 128 : (p.Cout > 32 ? 64 : 32);
  if (out_f32 && bn > 64) bn = 64;
  int ck = p.Cin >= 64 ? 64 : (p.Cin >= 32 ? 32 : 16);
  // Measured per layer INSIDE the training step (bench.py --profile-out under PLYOLO_FORCE_*; the stand-alone
  // microbenchmark is misleading here: its operands sit in the 256 MB Infinity Cache): the 8x16 tile beats the 16x16
  // tile on every layer with 64+ output channels -- 3 instead of 2 workgroups per CU, double-buffered halo, finer
  // tail -- and on the large maps (M >= 150k positions: HBM-bound 1x1 layers, the 80x80 head) 32-channel chunks
  // beat 64 (3x3 128->128 @80x80: 82.7 -> 79.2 us, 1x1 128->128 @80x80: 43.5 -> 37.0 us, 1x1 64->64 @160x160:
  // 76 -> 56 us).  Only the 32-output-channel layers (160x160 / 320x320 maps) keep the 16-row tile.
  const double Mpos = (double)p.N * p.OHt * p.OWt;
  int th = (p.OHt > 8 && bn == 32) ? 16 : 8;
  if (p.si == 2) th = 8;  // stride-2 halo tile: 17 x 33 pixels
  if (p.si == 2 && ext_y > 1 && ck > 32) ck = 32;
  // ... and 32-channel chunks everywhere: on the small maps 64/128-channel chunks are 1-3 us faster per launch when
  // timed alone, but the whole step is 2-3 % faster with 32 -- the backward runs the weight-gradient lane beside
  // this kernel, and the lighter workgroup (148 VGPRs, 20 KB of halo buffers) leaves it more of each CU
  static const int ck_mode = getenv("PLYOLO_CK_MODE") ? atoi(getenv("PLYOLO_CK_MODE")) : 0;  // 1: 64/128 on small maps
  if (p.si == 1 && ck > 32 && (ck_mode == 0 || (bn >= 64 && Mpos >= 150.0e3))) ck = 32;
  if (ck_mode == 1 && th == 8 && p.si == 1 && p.Cin == 128 && !out_f32 && bn >= 64 && ck == 64) ck = 128;
  // the 16-row tile double-buffers its halo only with 32-channel chunks (two 64-channel buffers leave no LDS for a second workgroup)
  if (th == 16 && p.si == 1 && ck == 64 && !out_f32 && p.db >= 2) ck = 32;
  if (p.pre && ck > 32) ck = 32;   // lazy-input instances exist for 16- / 32-channel chunks
  if (const char* e = getenv("PLYOLO_FORCE_CK")) { const int v = atoi(e); if (v == 16 || v == 32 || v == 64) ck = v < ck ? v : ck; }
  if (const char* e = getenv("PLYOLO_FORCE_BN")) { const int v = atoi(e); if (v == 32 || v == 64 || v == 128) bn = v < bn ? v : bn; }
  if (const char* e = getenv("PLYOLO_FORCE_TH")) { const int v = atoi(e); if (v == 8 || v == 16) th = v; }
  *BN = bn;
  *CK = ck;
  *TH = th;
}

// 4-row tiles (conv_mfma_t4.hip) for a 3x3 stride-1 launch whose 8-row tiling gives at most PLYOLO_TH4_MAX_WG workgroups
bool use_t4(const ConvP& p, int ksize, int BN, int CK, int TH, bool plain_bf16) {
  // stand-alone, batch 32: 128 -> 128 @20x20 (192 workgroups of 8 rows) 16.6 -> 13.8 us, 256 -> 256 @20x20 (384) 32.6 -> 28.6,
  // 128 -> 128 @40x40 (480) 23.2 -> 23.7: the 20x20 maps take the 4-row tiles
  const int max_wg = getenv("PLYOLO_TH4_MAX_WG") ? atoi(getenv("PLYOLO_TH4_MAX_WG")) : 400;
  if (max_wg <= 0 || !plain_bf16 || ksize != 3 || p.si != 1 || p.so != 1 || TH != 8 || CK != 32 || (BN != 128 && BN != 64)) return false;
  if (!(p.db && p.Cin > CK) || p.ablate || p.OHt <= 4) return false;
  return p.nmb * ((p.Cout + BN - 1) / BN) <= max_wg;
}

// row-flattened tiles (conv_mfma_flat.hip) for a 3x3 stride-1 launch on a 20- or 40-wide map with 128-channel output blocks;
// prepares the ConvP when taken.  PLYOLO_FLAT=0: the rectangular tiles (8 x 16, or 4 x 16 below PLYOLO_TH4_MAX_WG workgroups); 1 (default):
// the 20-wide maps (stand-alone, batch 32: 128 -> 128 13.8 -> 13.4 us, 256 -> 256 29.4 -> 27.0 forward, 28.6 -> 25.9 backward); 2: the 40-wide
// maps too -- measured SLOWER there (128 -> 128 @40x40 23.6 -> 27.2 us forward, 22.7 -> 25.8 backward): ten fragments per wave leave one
// fragment set and no room for the software pipeline of the 8 x 16 tile, which the 17 % of saved rows do not pay for
bool use_flat(ConvP& p, int ksize, int BN, int CK, bool plain_bf16) {
  const int mode = getenv("PLYOLO_FLAT") ? atoi(getenv("PLYOLO_FLAT")) : 1;
  if (mode == 0 || (mode == 1 && p.OWt != 20)) return false;
  const int trows = (mode == 3 && p.OWt == 40) ? 2 : 4;      // 3: 40-wide maps as tiles of TWO rows (80 pixels, the 20-wide maps' instance)
  if (!plain_bf16 || ksize != 3 || p.si != 1 || p.so != 1 || CK != 32 || BN != 128 || !(p.db && p.Cin > CK) || p.ablate) return false;
  if ((p.OWt != 20 && p.OWt != 40) || p.OHt % 4 != 0 || p.OWt != p.W || p.OHt != p.H) return false;
  p.tw = p.OWt; p.trows = trows; p.twinv = (65536 + p.tw - 1) / p.tw;
  p.ITH = p.trows + 2; p.ITW = p.tw + 2;
  p.tiles_x = 1; p.tiles_y = p.OHt / p.trows;
  p.nmb = p.N * p.tiles_y;
  return true;
}

// ragged column blocks (conv_mfma_rag.hip) for a 3x3 stride-1 launch of the 8 x 16 MF16 shape whose Cout leaves at most 64 channels behind
// its last full 128-channel block (160 = 128 + 32, 320 = 256 + 64: YOLOX-x); PLYOLO_RAG=0: whole 128-channel blocks everywhere
bool use_rag(const ConvP& p, int ksize, int BN, int CK, int TH, bool plain_bf16) {
  const int on = getenv("PLYOLO_RAG") ? atoi(getenv("PLYOLO_RAG")) : 1;      // (read per call: the tests switch it)
  if (!on || !plain_bf16 || ksize != 3 || p.si != 1 || p.so != 1 || BN != 128 || CK != 32 || TH != 8 || !(p.db && p.Cin > CK) || p.ablate) return false;
  const int rem = p.Cout % 128;
  return p.Cout > 128 && rem > 0 && rem <= 64;
}

// one 96-channel block of three waves (conv_mfma_rag.hip) for a 3x3 stride-1 launch with 65 .. 96 output channels (YOLOX-x: 80): the MF16
// instance (32-channel double-buffered chunks, Cin > 32) or the single-chunk one of a first convolution (Cin <= 16); PLYOLO_RAG=0: off
bool use_n96(const ConvP& p, int ksize, int BN, int CK, int TH, bool plain_bf16) {
  const int on = getenv("PLYOLO_RAG") ? atoi(getenv("PLYOLO_RAG")) : 1;
  if (!on || !plain_bf16 || ksize != 3 || p.si != 1 || p.so != 1 || BN != 128 || TH != 8 || p.ablate || p.Cout <= 64 || p.Cout > 96) return false;
  return (CK == 32 && p.db && p.Cin > CK) || (CK == 16 && p.Cin <= 16);
}

void set_taps(ConvP& p) {
  p.taps_lo = 0ull;
  p.taps_hi = 0u;
  for (int t = 0; t < p.ntaps; ++t) {
    const unsigned code = (unsigned)p.tap_dy[t] | ((unsigned)p.tap_dx[t] << 2) | ((unsigned)p.tap_w[t] << 4);
    if (t < 8) p.taps_lo |= (unsigned long long)code << (8 * t);
    else p.taps_hi = code;
  }
}

// choose the tiles, then derive the grid; ext = halo extent beyond (T-1)*si (per axis); the 32-bit offset limits of the
// loaders are checked at the C ABI (api.hip: check_conv)
void apply_tiles(ConvP& p, int ext_y, int ext_x, int TH) {
  p.ITH = (TH - 1) * p.si + ext_y;
  p.ITW = (TW - 1) * p.si + ext_x;
  p.tiles_y = (p.OHt + TH - 1) / TH;
  p.tiles_x = (p.OWt + TW - 1) / TW;
  p.nmb = p.N * p.tiles_y * p.tiles_x;
}

void finish(ConvP& p, int ext_y, int ext_x, bool out_f32, int* BN, int* CK, int* TH) {
  set_taps(p);
  if (const char* e = getenv("PLYOLO_ABLATE")) p.ablate = atoi(e);
  p.db = 1;
  if (const char* e = getenv("PLYOLO_DB")) p.db = atoi(e);
  pick_tiles(p, ext_y, out_f32, BN, CK, TH);
  apply_tiles(p, ext_y, ext_x, *TH);
}

void setup_fwd(const plyolo_conv_desc* d, ConvP& p, int* BN, int* CK, int* TH) {
  const int pad = (d->ksize - 1) / 2;
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
  p.ntaps = d->ksize * d->ksize;
  for (int kh = 0; kh < d->ksize; ++kh)
    for (int kw = 0; kw < d->ksize; ++kw) {
      const int t = kh * d->ksize + kw;
      p.tap_dy[t] = (signed char)kh; p.tap_dx[t] = (signed char)kw; p.tap_w[t] = (signed char)t;
    }
  p.accumulate = 0;
  p.nkb = (d->Cin + 15) / 16;
  p.nnb = (d->Cout + 31) / 32;
  finish(p, d->ksize, d->ksize, d->y_f32 != 0, BN, CK, TH);
}

}  // namespace

namespace plyolo {

// element counts of the fragment-native weight packs of one convolution (bf16 MFMA path):
//   forward  wp [tap][ceil(Cout/32)][ceil(Cin_p/16)][64 lanes][8]: lane (r = l&31, h = l>>5) holds
//            W[co = nb*32 + r][ci = kb*16 + h*8 + j][tap]   (zero outside the real extents)
//   dgrad    wpd[tap][ceil(Cin_p/32)][ceil(Cout/16)][64][8]:  W[co = kb*16 + h*8 + j][ci = nb*32 + r][tap]
void conv_mfma_pack_elems(int Cout_total, int Cin_p, int ksize, size_t* wp, size_t* wpd) {
  const size_t taps = (size_t)ksize * ksize;
  *wp = taps * ((Cout_total + 31) / 32) * ((Cin_p + 15) / 16) * 512;
  *wpd = taps * ((Cin_p + 31) / 32) * ((Cout_total + 15) / 16) * 512;
}

int conv_mfma_fwd(const plyolo_conv_desc* d, const void* x, const void* wp, const float* bias, void* y,
                  double* stats, const float* ep_coef, int ep_act, const void* ep_res, int ep_res_ld, void* stream) {
  ConvP p{};
  p.x = (const bf16_t*)x;
  p.w = (const bf16_t*)wp;
  p.y = y;
  p.bias = bias;
  p.stats = stats;
  p.ep_coef = ep_coef;
  p.ep_act = ep_act;
  p.ep_res = (const bf16_t*)ep_res;
  p.ep_res_ld = ep_res_ld;
  p.pre = d->x_coef; p.pre_ld = d->x_coef_ld; p.pre_act = d->x_act;
  int BN, CK, TH;
  setup_fwd(d, p, &BN, &CK, &TH);
  const bool f32 = d->y_f32 != 0;
  if (d->ksize == 3 && d->stride == 1 && !f32 && !p.pre && !bias && !ep_coef && !ep_res && !p.ablate &&
      conv3ws_accepts(p.N, p.H, p.W, p.Cin, p.Cout, p.x_ld, p.y_ld, y)) {
    char lab[64];
    snprintf(lab, sizeof(lab), "conv3ws_fwd<Cin%d,Cout%d>", p.Cin, p.Cout);
    const double M = (double)p.N * p.OHf * p.OWf;
    annotate(lab, 2.0 * M * d->Cout * d->Cin * 9.0, M * d->Cout * 2.0 + (double)d->N * d->H * d->W * d->Cin * 2.0);
    return submit(stream, [=](hipStream_t s) {
      return conv3ws_launch(p.x, p.w, p.y, p.stats, p.N, p.H, p.W, p.Cin, p.Cout, p.x_ld, p.y_ld, p.nkb, p.nnb, 0, p.taps_lo, p.taps_hi, s);
    });
  }
  // 3x3 stride 2: the dedicated instances (4-row tiles, double-buffered de-interleaved halo); PLYOLO_S2F=0: the generic path
  const bool s2f_on = !(getenv("PLYOLO_S2F") && atoi(getenv("PLYOLO_S2F")) == 0);
  // (Cin >= 64: the 32 -> 64 layer of the 320x320 map is a single chunk of tiny workgroups, 106 us generic against 131 here -- its
  // weight fragments, re-read from L2 by every workgroup, outweigh its tiles)
  const int s2_min_cin = getenv("PLYOLO_S2F_MINCIN") ? atoi(getenv("PLYOLO_S2F_MINCIN")) : 64;
  if (s2f_on && d->ksize == 3 && d->stride == 2 && !f32 && !p.pre && !p.ablate && d->Cin >= s2_min_cin && d->Cin >= 32 && d->Cout >= 64) {
    const int bn = d->Cout > 64 ? 128 : 64;
    apply_tiles(p, 3, 3, 4);
    char lab[64];
    const int rag = bn == 128 && conv_mfma_s2_ragged(d->Cout);      // decided here, once: the recorded launch replays what its label says
    if (rag) snprintf(lab, sizeof(lab), "conv_mfma_fwd_s2<BN128+64,CK32,TH4>");
    else snprintf(lab, sizeof(lab), "conv_mfma_fwd_s2<BN%d,CK32,TH4>", bn);
    const double M = (double)p.N * p.OHf * p.OWf;
    annotate(lab, 2.0 * M * d->Cout * d->Cin * 9.0, M * d->Cout * 2.0 + (double)d->N * d->H * d->W * d->Cin * 2.0);
    return submit(stream, [=](hipStream_t s) { return conv_mfma_launch_s2(&p, bn, rag, s); });
  }
  if (use_n96(p, d->ksize, BN, CK, TH, !f32 && !p.pre)) {
    char lab[64];
    snprintf(lab, sizeof(lab), "conv_mfma_fwd<BN96,CK%d,TH8>", CK);
    const double M = (double)p.N * p.OHf * p.OWf;
    annotate(lab, 2.0 * M * d->Cout * d->Cin * 9.0, M * d->Cout * 2.0 + (double)d->N * d->H * d->W * d->Cin * 2.0);
    const int ck = CK;
    return submit(stream, [=](hipStream_t s) { return conv_mfma_launch_n96(&p, ck, s); });
  }
  if (use_rag(p, d->ksize, BN, CK, TH, !f32 && !p.pre)) {
    char lab[64];
    snprintf(lab, sizeof(lab), "conv_mfma_fwd<BN128+%d,CK32,TH8>", p.Cout % 128 <= 32 ? 32 : 64);
    const double M = (double)p.N * p.OHf * p.OWf;
    annotate(lab, 2.0 * M * d->Cout * d->Cin * 9.0, M * d->Cout * 2.0 + (double)d->N * d->H * d->W * d->Cin * 2.0);
    return submit(stream, [=](hipStream_t s) { return conv_mfma_launch_rag(&p, s); });
  }
  if (use_flat(p, d->ksize, BN, CK, !f32 && !p.pre && !bias && !ep_coef && !ep_res)) {
    char lab[64];
    snprintf(lab, sizeof(lab), "conv_mfma_fwd<BN128,CK32,FLAT%d>", p.tw);
    const double M = (double)p.N * p.OHf * p.OWf;
    annotate(lab, 2.0 * M * d->Cout * d->Cin * 9.0, M * d->Cout * 2.0 + (double)d->N * d->H * d->W * d->Cin * 2.0);
    return submit(stream, [=](hipStream_t s) { return conv_mfma_launch_flat(&p, 0, s); });
  }
  if (use_t4(p, d->ksize, BN, CK, TH, !f32 && !p.pre)) {
    apply_tiles(p, 3, 3, 4);
    char lab[64];
    snprintf(lab, sizeof(lab), "conv_mfma_fwd<BN%d,CK32,TH4>", BN);
    const double M = (double)p.N * p.OHf * p.OWf;
    annotate(lab, 2.0 * M * d->Cout * d->Cin * 9.0, M * d->Cout * 2.0 + (double)d->N * d->H * d->W * d->Cin * 2.0);
    const int bn = BN;
    return submit(stream, [=](hipStream_t s) { return conv_mfma_launch_t4(&p, bn, 0, s); });
  }
  {
    char lab[64];
    snprintf(lab, sizeof(lab), "conv_mfma_fwd<BN%d,CK%d,TH%d>%s%s", BN, CK, TH, f32 ? "f32out" : "", p.pre ? "+bnact" : "");
    const double M = (double)p.N * p.OHf * p.OWf;
    annotate(lab, 2.0 * M * d->Cout * d->Cin * d->ksize * d->ksize, M * (d->Cout * (f32 ? 4.0 : 2.0)) + (double)d->N * d->H * d->W * d->Cin * 2.0);
  }
  return submit(stream, [=](hipStream_t s) { return f32 ? launch_bn<true>(p, BN, CK, TH, s) : launch_bn<false>(p, BN, CK, TH, s); });
}

// dx[N,H,W,Cin] = sum_taps dy[...] * w ; weights in the dgrad fragment pack
// red (optional): the BatchNorm-backward reduction of the unit(s) behind dx rides this launch's store loop (plyolo_bn_red); the
// caller has asked conv_mfma_dgrad_red_fits first
int conv_mfma_dgrad(const plyolo_conv_desc* d, const void* dy, const void* wpd, void* dx, int accumulate,
                    const plyolo_bn_red* red, void* stream, int* red_fits) {
  const int pad = (d->ksize - 1) / 2;
  const int OH = (d->H + 2 * pad - d->ksize) / d->stride + 1;
  const int OW = (d->W + 2 * pad - d->ksize) / d->stride + 1;
  const int Kc = (d->Cout + 7) & ~7;  // contraction length: dy rows are read in 16-byte vectors
  ConvP b{};
  b.x = (const bf16_t*)dy;
  b.w = (const bf16_t*)wpd;
  b.y = dx;
  b.bias = nullptr; b.stats = nullptr;
  b.N = d->N; b.H = OH; b.W = OW;
  b.Cin = Kc; b.Cout = d->Cin; b.x_ld = d->y_ld; b.y_ld = d->x_ld;
  b.OHf = d->H; b.OWf = d->W;
  b.accumulate = accumulate;
  b.nkb = (d->Cout + 15) / 16;
  b.nnb = (d->Cin + 31) / 32;
  b.si = 1;
  int rc = 0;
  int BN, CK, TH;
  if (d->stride == 1) {
    ConvP p = b;
    if (d->ksize == 1 && ((size_t)d->N * d->H * d->W) % TW == 0) {
      const int rows = (int)((size_t)d->N * d->H * d->W / TW);
      p.N = 1; p.H = rows; p.W = TW; p.OHf = rows; p.OWf = TW;
    }
    p.OHt = p.OHf; p.OWt = p.OWf;
    p.so = 1; p.oy_off = 0; p.ox_off = 0;
    p.iy_off = -pad; p.ix_off = -pad;
    p.ntaps = d->ksize * d->ksize;
    for (int dy_ = 0; dy_ < d->ksize; ++dy_)
      for (int dx_ = 0; dx_ < d->ksize; ++dx_) {
        const int t = dy_ * d->ksize + dx_;
        // dX[y,x] += dZ[y-pad+dy, x-pad+dx] * W[kh = k-1-dy][kw = k-1-dx]
        p.tap_dy[t] = (signed char)dy_; p.tap_dx[t] = (signed char)dx_;
        p.tap_w[t] = (signed char)((d->ksize - 1 - dy_) * d->ksize + (d->ksize - 1 - dx_));
      }
    finish(p, d->ksize, d->ksize, false, &BN, &CK, &TH);
    if (!red_fits && !(red && red->n > 0) && d->ksize == 3 && !p.ablate && conv3ws_accepts(p.N, p.H, p.W, p.Cin, p.Cout, p.x_ld, p.y_ld, dx)) {
      char lab[64];
      snprintf(lab, sizeof(lab), "conv3ws_dgrad<Cin%d,Cout%d>", p.Cin, p.Cout);
      const double Mo = (double)d->N * OH * OW, Mi = (double)d->N * d->H * d->W;
      annotate(lab, 2.0 * Mo * d->Cout * d->Cin * 9.0, (Mo * Kc + Mi * d->Cin * (accumulate ? 2.0 : 1.0)) * 2.0);
      return submit(stream, [=](hipStream_t s) {
        return conv3ws_launch(p.x, p.w, p.y, nullptr, p.N, p.H, p.W, p.Cin, p.Cout, p.x_ld, p.y_ld, p.nkb, p.nnb, p.accumulate, p.taps_lo, p.taps_hi, s);
      });
    }
    if (red_fits) { *red_fits = (d->ksize == 3 && !p.ablate) ? conv_mfma_red_has(BN, CK, TH, 0) : 0; return 0; }
    if (!(red && red->n > 0) && use_n96(p, d->ksize, BN, CK, TH, true)) {
      char lab[64];
      snprintf(lab, sizeof(lab), "conv_mfma_dgrad<BN96,CK%d,TH8>", CK);
      const double Mo = (double)d->N * OH * OW, Mi = (double)d->N * d->H * d->W;
      annotate(lab, 2.0 * Mo * d->Cout * d->Cin * 9.0, (Mo * Kc + Mi * d->Cin * (accumulate ? 2.0 : 1.0)) * 2.0);
      const int ck = CK;
      return submit(stream, [=](hipStream_t s) { return conv_mfma_launch_n96(&p, ck, s); });
    }
    if (!(red && red->n > 0) && use_rag(p, d->ksize, BN, CK, TH, true)) {   // (a folded reduction keeps the whole-block RED instance)
      char lab[64];
      snprintf(lab, sizeof(lab), "conv_mfma_dgrad<BN128+%d,CK32,TH8>", p.Cout % 128 <= 32 ? 32 : 64);
      const double Mo = (double)d->N * OH * OW, Mi = (double)d->N * d->H * d->W;
      annotate(lab, 2.0 * Mo * d->Cout * d->Cin * 9.0, (Mo * Kc + Mi * d->Cin * (accumulate ? 2.0 : 1.0)) * 2.0);
      return submit(stream, [=](hipStream_t s) { return conv_mfma_launch_rag(&p, s); });
    }
    if (use_flat(p, d->ksize, BN, CK, true)) {
      const bool rf = red && red->n > 0;
      if (rf) p.red = *red;
      char lab[64];
      snprintf(lab, sizeof(lab), "conv_mfma_dgrad<BN128,CK32,FLAT%d>%s", p.tw, rf ? "+bnred" : "");
      const double Mo = (double)d->N * OH * OW, Mi = (double)d->N * d->H * d->W;
      annotate(lab, 2.0 * Mo * d->Cout * d->Cin * 9.0, (Mo * Kc + Mi * d->Cin * ((accumulate ? 2.0 : 1.0) + (rf ? 1.0 : 0.0))) * 2.0);
      return submit(stream, [=](hipStream_t s) { return conv_mfma_launch_flat(&p, rf ? 1 : 0, s); });
    }
    if (use_t4(p, d->ksize, BN, CK, TH, true)) {
      const bool r4 = red && red->n > 0;
      if (r4) p.red = *red;
      apply_tiles(p, 3, 3, 4);
      char lab[64];
      snprintf(lab, sizeof(lab), "conv_mfma_dgrad<BN%d,CK32,TH4>%s", BN, r4 ? "+bnred" : "");
      const double Mo = (double)d->N * OH * OW, Mi = (double)d->N * d->H * d->W;
      annotate(lab, 2.0 * Mo * d->Cout * d->Cin * 9.0, (Mo * Kc + Mi * d->Cin * ((accumulate ? 2.0 : 1.0) + (r4 ? 1.0 : 0.0))) * 2.0);
      const int bn = BN;
      return submit(stream, [=](hipStream_t s) { return conv_mfma_launch_t4(&p, bn, r4 ? 1 : 0, s); });
    }
    const bool use_red = red && red->n > 0 && !p.ablate && conv_mfma_red_has(BN, CK, TH, 0);
    if (red && red->n > 0 && !use_red) { set_error("conv_mfma_dgrad: no RED instance for this tile configuration (ask plyolo_conv2d_dgrad_red_fits)"); return -1; }
    if (use_red) p.red = *red;
    {
      char lab[64];
      snprintf(lab, sizeof(lab), "conv_mfma_dgrad<BN%d,CK%d,TH%d>%s", BN, CK, TH, use_red ? "+bnred" : "");
      const double Mo = (double)d->N * OH * OW, Mi = (double)d->N * d->H * d->W;
      annotate(lab, 2.0 * Mo * d->Cout * d->Cin * d->ksize * d->ksize, (Mo * Kc + Mi * d->Cin * (accumulate ? 2.0 : 1.0) + (use_red ? Mi * d->Cin : 0.0)) * 2.0);
    }
    if (use_red) return submit(stream, [=](hipStream_t s) { return conv_mfma_launch_red(&p, BN, CK, TH, s); });
    return submit(stream, [=](hipStream_t s) { return launch_bn<false>(p, BN, CK, TH, s); });
  }
  // 3x3 stride 2 into at most 64 channels: all four parity classes on ONE staged dZ tile per workgroup (conv_s2d.hip; PLYOLO_S2D=0: the
  // four-job launch below).  Same box, batch 32: 64 -> 32 channels @320x320 147 -> 92 us, 128 -> 64 @160x160 74 -> 65 us.  The wider
  // layers keep the four jobs (4 launches of YOLOX-s, 0.33 ms): one workgroup holds 4 classes x MT fragments x 16 accumulators, and with
  // two workgroups per CU that leaves MT = 2 -- every weight fragment feeds two MFMAs and the launch is bound by the vector L1 (4-row
  // tiles 0.37 ms, 64-channel blocks 0.39 ms) -- while MT = 4 with one wave per SIMD exposes the LDS latency (8-row tiles 0.50 ms)
  {
    const int s2d = getenv("PLYOLO_S2D") ? atoi(getenv("PLYOLO_S2D")) : 1;            // (read per call: the tests switch them)
    const int s2d_maxc = getenv("PLYOLO_S2D_MAXC") ? atoi(getenv("PLYOLO_S2D_MAXC")) : 64;
    const int bn = d->Cin > 64 ? 128 : (d->Cin > 32 ? 64 : 32);
    if (s2d && d->Cin <= s2d_maxc && d->ksize == 3 && d->stride == 2 && !getenv("PLYOLO_ABLATE")) {
      if (red_fits) { *red_fits = 1; return 0; }
      ConvP p = b;
      const int th = conv_s2d_th(bn);
      p.OHt = (d->H + 1) / 2; p.OWt = (d->W + 1) / 2;
      p.so = 2; p.oy_off = 0; p.ox_off = 0; p.iy_off = 0; p.ix_off = 0;
      p.ntaps = 9;
      p.tiles_y = (p.OHt + th - 1) / th;
      p.tiles_x = (p.OWt + TW - 1) / TW;
      p.nmb = p.N * p.tiles_y * p.tiles_x;
      const bool use_red = red && red->n > 0;
      if (use_red) p.red = *red;
      char lab[64];
      snprintf(lab, sizeof(lab), "conv_s2d_dgrad<BN%d,TH%d>%s", bn, th, use_red ? "+bnred" : "");
      const double Mo = (double)d->N * OH * OW, Mi = (double)d->N * d->H * d->W;
      annotate(lab, 2.0 * Mo * d->Cout * d->Cin * 9.0, (Mo * Kc + Mi * d->Cin * ((accumulate ? 2.0 : 1.0) + (use_red ? 1.0 : 0.0))) * 2.0);
      return submit(stream, [=](hipStream_t s) { return conv_s2d_launch(&p, bn, use_red ? 1 : 0, s); });
    }
  }
  // stride 2 (ksize 3 pad 1, or ksize 1): one job per output parity class (1/2/2/4 taps), all four in ONE launch
  ConvJobs jobs{};
  int jBN = 0, jCK = 0, jTH = 0, ext[4][2] = {};
  double fl = 0.0, by = 0.0;
  for (int py = 0; py < 2; ++py)
    for (int px = 0; px < 2; ++px) {
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
      if (nky == 0 || nkx == 0) continue;  // (ksize 1, odd class): no tap reaches it; the caller zero-fills dx
      int miny = 9, maxy = -9, minx = 9, maxx = -9;
      for (int i = 0; i < nky; ++i) { miny = dyv[i] < miny ? dyv[i] : miny; maxy = dyv[i] > maxy ? dyv[i] : maxy; }
      for (int i = 0; i < nkx; ++i) { minx = dxv[i] < minx ? dxv[i] : minx; maxx = dxv[i] > maxx ? dxv[i] : maxx; }
      p.iy_off = miny; p.ix_off = minx;
      p.ntaps = 0;
      for (int i = 0; i < nky; ++i)
        for (int j = 0; j < nkx; ++j) {
          p.tap_dy[p.ntaps] = (signed char)(dyv[i] - miny);
          p.tap_dx[p.ntaps] = (signed char)(dxv[j] - minx);
          p.tap_w[p.ntaps] = (signed char)(ky[i] * d->ksize + kx[j]);
          ++p.ntaps;
        }
      int bn_, ck_, th_;
      finish(p, maxy - miny + 1, maxx - minx + 1, false, &bn_, &ck_, &th_);
      if (jobs.n == 0) { jBN = bn_; jCK = ck_; jTH = th_; }
      jTH = th_ < jTH ? th_ : jTH;   // the classes share one tile configuration (BN / CK agree by construction)
      ext[jobs.n][0] = maxy - miny + 1; ext[jobs.n][1] = maxx - minx + 1;
      jobs.c[jobs.n++] = p;
      const double Mo = (double)d->N * OH * OW, Mi = (double)d->N * d->H * d->W;
      fl += 0.25 * 2.0 * Mo * d->Cout * d->Cin * d->ksize * d->ksize;
      by += 0.25 * (Mo * Kc + Mi * d->Cin * (accumulate ? 2.0 : 1.0)) * 2.0;
    }
  if (jobs.n == 0) return 0;
  for (int j = 0; j < jobs.n; ++j) apply_tiles(jobs.c[j], ext[j][0], ext[j][1], jTH);
  if (red_fits) { *red_fits = (d->ksize == 3 && jobs.n == 4) ? conv_mfma_red_has(jBN, jCK, jTH, 1) : 0; return 0; }
  const bool use_red = red && red->n > 0 && conv_mfma_red_has(jBN, jCK, jTH, 1);
  if (red && red->n > 0 && !use_red) { set_error("conv_mfma_dgrad: no RED instance for this stride-2 tile configuration"); return -1; }
  if (use_red) {
    for (int j = 0; j < jobs.n; ++j) jobs.c[j].red = *red;
    by += (double)d->N * d->H * d->W * d->Cin * 2.0;
  }
  const int jmode = use_red ? 0 : (jobs_n96(jobs, jBN, jCK, jTH) ? 2 : (jobs_ragged(jobs, jBN, jCK, jTH) ? 1 : 0));
  {
    char lab[64];
    if (jmode == 2) snprintf(lab, sizeof(lab), "conv_mfma_dgrad_s2<BN96,CK32,TH8>x%d", jobs.n);
    else if (jmode == 1) snprintf(lab, sizeof(lab), "conv_mfma_dgrad_s2<BN128+%d,CK32,TH8>x%d", jobs.c[0].Cout % 128 <= 32 ? 32 : 64, jobs.n);
    else snprintf(lab, sizeof(lab), "conv_mfma_dgrad_s2<BN%d,CK%d,TH%d>x%d%s", jBN, jCK, jTH, jobs.n, use_red ? "+bnred" : "");
    annotate(lab, fl, by);
  }
  const int bn = jBN, ck = jCK, th = jTH;
  if (use_red) return submit(stream, [=](hipStream_t s) { return conv_mfma_launch_jobs_red(&jobs, bn, ck, th, s); });
  return submit(stream, [=](hipStream_t s) { return launch_jobs(jobs, bn, ck, th, jmode, s); });
}


// ---- 3x3 stride-1 data gradient with the unit's BatchNorm + SiLU backward in the halo loader (plyolo_conv2d_dgrad_bn, ksize 3)
namespace {
// the ConvP of the stride-1 data gradient (as conv_mfma_dgrad builds it) with its tiles chosen; false: no BNB instance runs these tiles
bool setup_dgrad_bnb(const plyolo_conv_desc* d, ConvP& p, int* BN, int* CK, int* TH) {
  const int Kc = (d->Cout + 7) & ~7;
  p.N = d->N; p.H = d->H; p.W = d->W;      // (3x3, pad 1, stride 1: the output gradient has the input's extent)
  p.Cin = Kc; p.Cout = d->Cin; p.x_ld = d->y_ld; p.y_ld = d->x_ld;
  p.OHf = d->H; p.OWf = d->W;
  p.nkb = (d->Cout + 15) / 16;
  p.nnb = (d->Cin + 31) / 32;
  p.si = 1;
  p.OHt = p.OHf; p.OWt = p.OWf;
  p.so = 1; p.oy_off = 0; p.ox_off = 0;
  p.iy_off = -1; p.ix_off = -1;
  p.ntaps = 9;
  for (int dy_ = 0; dy_ < 3; ++dy_)
    for (int dx_ = 0; dx_ < 3; ++dx_) {
      const int t = dy_ * 3 + dx_;
      p.tap_dy[t] = (signed char)dy_; p.tap_dx[t] = (signed char)dx_;
      p.tap_w[t] = (signed char)((2 - dy_) * 3 + (2 - dx_));
    }
  finish(p, 3, 3, false, BN, CK, TH);
  if (p.ablate || !p.db) return false;
  if (use_t4(p, 3, *BN, *CK, *TH, true)) {
    apply_tiles(p, 3, 3, 4);
    *TH = 4;
  }
  return plyolo::conv_mfma_bnb_has(*BN, *CK, *TH, p.Cin > *CK ? 1 : 0) == 1;
}
}  // namespace

// 1 when plyolo_conv2d_dgrad_bn covers this 3x3 unit: bf16, stride 1, SiLU, whole channel vectors, the per-channel table fits beside the
// halo buffers (<= 512 channels), the tiles it runs have a BNB instance, dx spans at most PLYOLO_FUSE_BNBWD3_BLK (default 2) output
// blocks (every block re-derives dz for the whole contraction length) and the output gradient is at most PLYOLO_FUSE_BNBWD3_MB
int conv_mfma_dgrad_bn_fits(const plyolo_conv_desc* d, int act) {
  if (d->dtype != PLYOLO_BF16 || d->ksize != 3 || d->stride != 1 || act != PLYOLO_ACT_SILU) return 0;
  if (d->Cout % 8 != 0 || d->Cout > 512 || d->x_coef) return 0;
  const double max_mb = getenv("PLYOLO_FUSE_BNBWD3_MB") ? atof(getenv("PLYOLO_FUSE_BNBWD3_MB")) : 1.0e9;
  if ((double)d->N * d->H * d->W * d->Cout * 2.0 > max_mb * 1.0e6) return 0;
  ConvP p{};
  int BN, CK, TH;
  if (!setup_dgrad_bnb(d, p, &BN, &CK, &TH)) return 0;
  const int maxblk = getenv("PLYOLO_FUSE_BNBWD3_BLK") ? atoi(getenv("PLYOLO_FUSE_BNBWD3_BLK")) : 2;
  return (d->Cin + BN - 1) / BN <= maxblk ? 1 : 0;
}

int conv_mfma_dgrad_bn(const plyolo_conv_desc* d, const plyolo_bn_bwd_fuse* f, const void* wpd, void* dx, int accumulate, const plyolo_bn_red* red,
                       void* stream) {
  ConvP p{};
  p.x = (const bf16_t*)f->dout;
  p.w = (const bf16_t*)wpd;
  p.y = dx;
  p.accumulate = accumulate;
  int BN, CK, TH;
  if (!setup_dgrad_bnb(d, p, &BN, &CK, &TH)) { set_error("conv_mfma_dgrad_bn: no BNB instance for this unit (ask plyolo_conv2d_dgrad_bn_fits)"); return -1; }
  p.x_ld = f->dout_ld;
  if (red && red->n > 0) p.red = *red;
  p.bz = (const bf16_t*)f->z; p.bz_ld = f->z_ld;
  p.bx2 = (const bf16_t*)f->dout2; p.bx2_ld = f->dout2_ld; p.bsplit = f->dout2 ? f->dout_split : 0;
  p.bcoef = f->coef; p.bslots = f->bslots;
  p.bgamma = f->gamma; p.bdgamma = f->dgamma; p.bdbeta = f->dbeta;
  p.bpsplit = f->par_split; p.bgamma2 = f->gamma2; p.bdgamma2 = f->dgamma2; p.bdbeta2 = f->dbeta2;
  p.bdz = (bf16_t*)f->dz; p.bdz_ld = f->dz_ld;
  p.bfwd = (bf16_t*)f->fwd_to; p.bfwd_ld = f->fwd_ld;
  {
    char lab[64];
    snprintf(lab, sizeof(lab), "conv_mfma_dgrad_bn<BN%d,CK%d,TH%d>%s", BN, CK, TH, p.red.n > 0 ? "+bnred" : "");
    const double M = (double)d->N * d->H * d->W;
    annotate(lab, 2.0 * M * d->Cout * d->Cin * 9.0,
             (M * d->Cout * (2.0 + (f->dz ? 1.0 : 0.0) + (f->fwd_to ? 1.0 : 0.0)) + M * d->Cin * ((accumulate ? 2.0 : 1.0) + (p.red.n > 0 ? 1.0 : 0.0))) * 2.0);
  }
  return submit(stream, [=](hipStream_t s) { return conv_mfma_launch_bnb(&p, BN, CK, TH, s); });
}

// 1 when conv_mfma_dgrad(d, ..., red) has a RED instance for the tiles this data gradient runs on (asked with the launch's own
// tile selection, nothing is submitted)
int conv_mfma_dgrad_red_fits(const plyolo_conv_desc* d) {
  if (d->dtype != PLYOLO_BF16) return 0;
  int fits = 0;
  conv_mfma_dgrad(d, nullptr, nullptr, nullptr, 0, nullptr, nullptr, &fits);
  return fits;
}

}  // namespace plyolo
