// Microbenchmark: per-channel statistic accumulation with fp64 atomics into replicated slots.
// Variants: slot = XCC_ID (workgroup-scope atomics, L2-local) / blockIdx%R with agent-scope atomics.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__device__ inline int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xf; }
template <int MODE>
__global__ __launch_bounds__(256) void k(double* slots, int C, int R, int spin, int* xcc_hist) {
  // pretend work
  float v = threadIdx.x;
  for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
  const int c = threadIdx.x;
  if (c < C) {
    double val = 1.0 + (v == 12345.f ? 1.0 : 0.0);
    int slot;
    if (MODE == 0) slot = xcc_id(); else slot = blockIdx.x % R;
    double* p = slots + ((size_t)slot * 2) * C + c;
    if (MODE == 0) {
      __hip_atomic_fetch_add(p, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(p + C, 2.0 * val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
      __hip_atomic_fetch_add(p, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(p + C, 2.0 * val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (threadIdx.x == 0 && xcc_hist) atomicAdd(&xcc_hist[(blockIdx.x % 8) * 16 + xcc_id()], 1);
}
__global__ void check(const double* slots, int C, int R, double* out) {
  const int c = threadIdx.x + blockIdx.x * blockDim.x;
  if (c >= C) return;
  double s = 0, s2 = 0;
  for (int r = 0; r < R; ++r) { s += slots[(size_t)r * 2 * C + c]; s2 += slots[(size_t)r * 2 * C + C + c]; }
  out[c] = s; out[C + c] = s2;
}
int main() {
  const int C = 128;
  double *slots, *out; int* hist;
  CK(hipMalloc(&slots, 64 * 2 * C * 8)); CK(hipMalloc(&out, 2 * C * 8)); CK(hipMalloc(&hist, 128 * 4));
  CK(hipMemset(hist, 0, 128 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<double> h(2 * C);
  for (int nblk : {800, 3200}) for (int spin : {0, 20000}) for (int mode = 0; mode < 4; ++mode) {
    const int R = mode == 0 ? 16 : (mode == 1 ? 8 : (mode == 2 ? 32 : 64));
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipMemset(slots, 0, 64 * 2 * C * 8));
      CK(hipEventRecord(e0));
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(nblk), dim3(256), 0, 0, slots, C, R, spin, rep == 0 ? hist : nullptr);
      else hipLaunchKernelGGL(k<1>, dim3(nblk), dim3(256), 0, 0, slots, C, R, spin, (int*)nullptr);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
    }
    hipLaunchKernelGGL(check, dim3(1), dim3(128), 0, 0, slots, C, R, out);
    CK(hipMemcpy(h.data(), out, 2 * C * 8, hipMemcpyDeviceToHost));
    bool ok = true; for (int c = 0; c < C; ++c) ok = ok && h[c] == nblk && h[C + c] == 2.0 * nblk;
    printf("nblk %4d spin %5d mode %s R %2d: %.1f us  sums %s (%.0f)\n", nblk, spin, mode == 0 ? "xcc/wg-scope " : "blk%R/agent  ", R, best * 1e3, ok ? "OK" : "WRONG", h[0]);
  }
  std::vector<int> hh(128); CK(hipMemcpy(hh.data(), hist, 128 * 4, hipMemcpyDeviceToHost));
  for (int b = 0; b < 8; ++b) { printf("blockIdx%%8=%d -> xcc:", b); for (int x = 0; x < 16; ++x) if (hh[b * 16 + x]) printf(" [%d]=%d", x, hh[b * 16 + x]); printf("\n"); }
  return 0;
}
