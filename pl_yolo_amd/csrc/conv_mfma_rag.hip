// RAGGED output-channel blocks for the 3x3 stride-1 bf16 MFMA convolution (conv_mfma_body.h, v_mfma_f32_16x16x32_bf16 path, 8 x 16 tiles,
// 32-channel double-buffered chunks).  The plain launch tiles Cout with 128-channel blocks: a 160-channel layer (YOLOX-x, 1.25 blocks)
// runs two blocks of MFMAs for 160 / 256 = 62 % of useful columns, a 320-channel layer three for 83 % -- the waves beyond Cout multiply
// clamped weights and store nothing.  Here the grid's last column block is a NARROWER instance of the same body (32 channels: four waves
// share the tile's 128 pixels; 64: two by two), so only real columns are computed; both widths live in ONE kernel (blockIdx.y picks the
// path) so that the blocks of one pixel tile still share their halo reads in L2 and the short blocks fill the tail of the launch.
// Measured stand-alone, B = 16 (profiles/r05_ab_ragged.txt): 160 -> 160 @160x160 260 -> 204 us, 320 -> 320 @80x80 192 -> 169, @160x160 752 -> 680;
// outputs bit-identical to whole blocks.  Forward and plain data gradient (a folded BatchNorm reduction keeps the whole-block RED instance: the
// layers this serves lie above the fold's size limit).  Below: one 96-channel block of three waves for layers with 65 .. 96 output channels.
// Own translation unit: see conv_mfma_body.h.
#define PLYOLO_CONV_PD 2
#include "conv_mfma_body.h"

namespace {

template <int REM>
__global__ __launch_bounds__(256, 2) void conv_mfma_rag_kernel(const ConvP p, const int nfull) {
  if ((int)blockIdx.y < nfull)
    conv_mfma_body<128, 32, 8, false, 0, true, false, true>(p, (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.y * 4);
  else
    conv_mfma_body<REM, 32, 8, false, 0, true, false, true>(p, (int)blockIdx.x, (int)gridDim.x, nfull * 4);
}

template <int REM>
hipError_t launch_rag_inst(ConvP p, hipStream_t s) {
  constexpr int CK = 32, TH = 8, BM = TH * TW;
  constexpr int ROWB = CK * 2 + 32;
  p.rowp = (p.ITW * ROWB + 255) & ~255;
  p.bufsz = p.ITH * p.rowp;
  auto epi = [](int BN) { return (size_t)BM * (BN * 2 + 16) + (size_t)(4 / (BN / 32)) * 2 * BN * 4; };
  const size_t lds_main = 2 * (size_t)p.bufsz, lds_epi = epi(128) > epi(REM) ? epi(128) : epi(REM);
  const size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
  auto kern = conv_mfma_rag_kernel<REM>;
  if (hipError_t e = plyolo::ensure_dynamic_lds((const void*)kern, lds); e != hipSuccess) return e;
  const int nfull = p.Cout / 128;
  hipLaunchKernelGGL(kern, dim3(p.nmb, nfull + 1), dim3(256), lds, s, p, nfull);
  return hipGetLastError();
}

// ---- 96-channel blocks: a layer with 65 .. 96 output channels (YOLOX-x: 80; YOLOX-m: 96) as ONE block of three waves (192 threads, a 32-channel
// block each) instead of a 128-channel block whose fourth wave -- and for 80 channels half of the third -- multiplies clamped weights
// (three waves per SIMD: the four halo vectors per thread of the 192-thread loader put the free allocation at 169 registers, one over the limit)
__global__ __launch_bounds__(192, 3) void conv_mfma_n96_kernel(const ConvP p) {
  conv_mfma_body<96, 32, 8, false, 0, true, false, true>(p, (int)blockIdx.x, (int)gridDim.x);
}
// (the first convolution of a CSPDarknet: 12 real input channels, one 16-channel chunk, no double buffer)
__global__ __launch_bounds__(192, 2) void conv_mfma_n96_ck16_kernel(const ConvP p) {
  conv_mfma_body<96, 16, 8, false, 0, false>(p, (int)blockIdx.x, (int)gridDim.x);
}
// the four parity jobs of a stride-2 data gradient into 65 .. 96 channels
__global__ __launch_bounds__(192, 2) void conv_mfma_jobs_n96_kernel(const ConvJobs jobs) {
  int j = 0;
  for (int k = 1; k < 4; ++k)
    if (k < jobs.n && (int)blockIdx.x >= jobs.start[k]) j = k;
  conv_mfma_body<96, 32, 8, false, 0, true>(jobs.c[j], (int)blockIdx.x - jobs.start[j], jobs.start[j + 1] - jobs.start[j]);
}

constexpr size_t n96_epi() { return (size_t)(8 * TW) * (96 * 2 + 16) + 1 * 2 * 96 * 4; }

}  // namespace

namespace plyolo {

// `convp`: a ConvP prepared for the 8 x 16 stride-1 tiles, 3x3, bf16 output, 65 .. 96 output channels; CK = 32: Cin > 32, double-buffered
// (the MF16 instance); CK = 16: a single 16-channel chunk
hipError_t conv_mfma_launch_n96(const void* convp, int CK, hipStream_t s) {
  ConvP p = *(const ConvP*)convp;
  if (p.Cout <= 64 || p.Cout > 96) return hipErrorInvalidValue;
  if (CK == 32) {
    constexpr int ROWB = 32 * 2 + 32;
    p.rowp = (p.ITW * ROWB + 255) & ~255;
    p.bufsz = p.ITH * p.rowp;
    const size_t lds_main = 2 * (size_t)p.bufsz, lds = lds_main > n96_epi() ? lds_main : n96_epi();
    auto kern = conv_mfma_n96_kernel;
    if (hipError_t e = plyolo::ensure_dynamic_lds((const void*)kern, lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(p.nmb, 1), dim3(192), lds, s, p);
    return hipGetLastError();
  }
  if (CK == 16) {
    constexpr int ROWB = 16 * 2 + 16;
    p.rowp = p.ITW * ROWB;
    if (p.si == 1) p.rowp = (p.rowp + 255) & ~255;
    const size_t lds_main = (size_t)p.ITH * p.rowp, lds = lds_main > n96_epi() ? lds_main : n96_epi();
    auto kern = conv_mfma_n96_ck16_kernel;
    if (hipError_t e = plyolo::ensure_dynamic_lds((const void*)kern, lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(p.nmb, 1), dim3(192), lds, s, p);
    return hipGetLastError();
  }
  return hipErrorInvalidValue;
}

// `jobsp`: ConvJobs of stride-1 parity classes (8 x 16 tiles, 32-channel double-buffered chunks, Cin > 32), 65 .. 96 output channels
hipError_t conv_mfma_launch_jobs_n96(const void* jobsp, hipStream_t s) {
  ConvJobs jobs = *(const ConvJobs*)jobsp;
  constexpr int ROWB = 32 * 2 + 16;
  size_t lds = n96_epi();
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
  auto kern = conv_mfma_jobs_n96_kernel;
  if (hipError_t e = plyolo::ensure_dynamic_lds((const void*)kern, lds); e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(total, 1), dim3(192), lds, s, jobs);
  return hipGetLastError();
}

// `convp`: a ConvP prepared for the 8 x 16 stride-1 tiles (apply_tiles), 3x3, Cin > 32, double-buffered, bf16 output, no lazy input;
// Cout = 128 * nfull + rem with nfull >= 1 and rem in (0, 64]
hipError_t conv_mfma_launch_rag(const void* convp, hipStream_t s) {
  const ConvP& p = *(const ConvP*)convp;
  const int rem = p.Cout % 128;
  if (p.Cout < 128 || rem == 0 || rem > 64) return hipErrorInvalidValue;
  return rem <= 32 ? launch_rag_inst<32>(p, s) : launch_rag_inst<64>(p, s);
}

}  // namespace plyolo
