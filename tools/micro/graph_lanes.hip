// Microbenchmark: hipGraph with a main chain (lane 0) and a dependent side lane (one edge per step),
// the shape of the "weight gradients on their own lane" schedule.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void spin(float* p, int iters) {
  float v = p[threadIdx.x + blockIdx.x * blockDim.x];
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
  p[threadIdx.x + blockIdx.x * blockDim.x] = v;
}
int main() {
  float* buf; CK(hipMalloc(&buf, 256 << 20)); CK(hipMemset(buf, 0, 256 << 20));
  hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int N = 80;
  for (int blocks : {64, 256, 1024}) for (int iters : {1000, 4000}) for (int mode = 0; mode < 4; ++mode) {
    // mode 0: one lane, 3N kernels (a, b, w per step)   mode 1: w on lane 1 with an edge per step
    // mode 2: like 1 plus a back edge every 4 steps      mode 3: one lane, only 2N kernels (a, b) = lower bound
    std::vector<hipEvent_t> ev(3 * N + 4); for (auto& x : ev) CK(hipEventCreateWithFlags(&x, hipEventDisableTiming));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
    int ne = 0;
    if (mode == 1 || mode == 2) { CK(hipEventRecord(ev[ne], s0)); CK(hipStreamWaitEvent(s1, ev[ne], 0)); ++ne; }
    std::vector<int> back(N, -1);
    for (int i = 0; i < N; ++i) {
      if (mode == 2 && i >= 4 && back[i - 4] >= 0) CK(hipStreamWaitEvent(s0, ev[back[i - 4]], 0));
      hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, s0, buf, iters);
      hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, s0, buf + (16 << 20), iters);
      if (mode == 0) hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, s0, buf + (32 << 20), iters);
      if (mode == 1 || mode == 2) {
        CK(hipEventRecord(ev[ne], s0)); CK(hipStreamWaitEvent(s1, ev[ne], 0)); ++ne;
        hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, s1, buf + (32 << 20), iters);
        if (mode == 2) { CK(hipEventRecord(ev[ne], s1)); back[i] = ne; ++ne; }
      }
    }
    if (mode == 1 || mode == 2) { CK(hipEventRecord(ev[ne], s1)); CK(hipStreamWaitEvent(s0, ev[ne], 0)); ++ne; }
    CK(hipStreamEndCapture(s0, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, s0));
    CK(hipStreamSynchronize(s0));
    CK(hipEventRecord(e0, s0));
    for (int w = 0; w < 5; ++w) CK(hipGraphLaunch(ge, s0));
    CK(hipEventRecord(e1, s0)); CK(hipStreamSynchronize(s0));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("blocks %4d iters %4d mode %d: %.3f ms per graph (%.1f us per step)\n", blocks, iters, mode, ms / 5, ms / 5 / N * 1e3);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
