// Microbenchmark: per-node cost of dependent kernel chains in a hipGraph, single lane vs forked lanes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void tiny(float* p, int iters) {
  float v = p[threadIdx.x + blockIdx.x * blockDim.x];
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
  p[threadIdx.x + blockIdx.x * blockDim.x] = v;
}
int main() {
  float* buf; CK(hipMalloc(&buf, 64 << 20)); CK(hipMemset(buf, 0, 64 << 20));
  hipStream_t s[8]; for (auto& x : s) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int N = 640;
  for (int blocks : {1, 100, 1024}) for (int iters : {0, 2000}) for (int lanes : {1, 2, 4}) {
    hipGraph_t g; hipGraphExec_t ge;
    std::vector<hipEvent_t> ev(lanes * 2); for (auto& evx : ev) CK(hipEventCreateWithFlags(&evx, hipEventDisableTiming));
    CK(hipStreamBeginCapture(s[0], hipStreamCaptureModeThreadLocal));
    CK(hipEventRecord(ev[0], s[0]));
    for (int l = 1; l < lanes; ++l) CK(hipStreamWaitEvent(s[l], ev[0], 0));
    for (int i = 0; i < N / lanes; ++i)
      for (int l = 0; l < lanes; ++l) hipLaunchKernelGGL(tiny, dim3(blocks), dim3(256), 0, s[l], buf + l * (1 << 20), iters);
    for (int l = 1; l < lanes; ++l) { CK(hipEventRecord(ev[l], s[l])); CK(hipStreamWaitEvent(s[0], ev[l], 0)); }
    CK(hipStreamEndCapture(s[0], &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, s[0]));
    CK(hipStreamSynchronize(s[0]));
    CK(hipEventRecord(e0, s[0]));
    for (int w = 0; w < 5; ++w) CK(hipGraphLaunch(ge, s[0]));
    CK(hipEventRecord(e1, s[0])); CK(hipStreamSynchronize(s[0]));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("graph blocks %4d iters %4d lanes %d: %.2f us per kernel (%.2f ms per %d-node graph)\n", blocks, iters, lanes, ms / 5 / N * 1e3, ms / 5, N);
    // plain stream
    if (lanes == 1) {
      for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(blocks), dim3(256), 0, s[0], buf, iters);
      CK(hipStreamSynchronize(s[0]));
      CK(hipEventRecord(e0, s[0]));
      for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(blocks), dim3(256), 0, s[0], buf, iters);
      CK(hipEventRecord(e1, s[0])); CK(hipStreamSynchronize(s[0]));
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("stream blocks %4d iters %4d        : %.2f us per kernel\n", blocks, iters, ms / N * 1e3);
    }
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
