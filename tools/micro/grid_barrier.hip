// Microbenchmark (round 5, review item 6): what does a grid-wide barrier cost on the 256 CUs of an MI355X, against the kernel
// boundary it would replace?  G co-resident workgroups of 256 threads; per barrier: __syncthreads, one agent-scope atomic add
// (release) by thread 0, a bounded spin on an agent-scope atomic load (acquire), __syncthreads.  Variants: K barriers back to
// back inside one launch (per-barrier cost = slope), and the pattern a conv + BatchNorm fusion would run -- every workgroup adds
// 2*C fp64 partials to stat slots (agent scope), barrier, every workgroup reads the 8 slots x 2 x C back with agent-scope loads.
// Compared with: the same work split over two launches on one stream.   hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ int g_sleep = 2;     // round 6: 0 = spin without s_sleep (set from argv[1])

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target, unsigned* err) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    const int sl = g_sleep;
    while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (sl) __builtin_amdgcn_s_sleep(2);
      if (++spins > (1 << 22)) { *err = 1u; ok = false; break; }   // bounded: a hung GPU costs the lease
    }
  }
  __syncthreads();
  return ok;
}

// round 6: TWO-LEVEL barrier.  The flat barrier's cost is the G same-address atomics, ~30 ns each (7.9 us at G = 256).  Here the
// workgroups of one XCD (blockIdx % 8) add to their group's counter (own 128-byte line: 8 counters take G/8 atomics each, in
// parallel), the LAST arriver of a group adds to the global counter (8 atomics), everybody polls the global counter.
__device__ __forceinline__ bool grid_barrier2(unsigned* cnt, unsigned epoch, unsigned* err) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    const unsigned G = gridDim.x, grp = blockIdx.x & 7u, ngrp = G < 8u ? G : 8u;
    const unsigned gsize = (G - grp + 7u) / 8u;
    const unsigned old = __hip_atomic_fetch_add(cnt + 32u * (1u + grp), 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1u == epoch * gsize) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    const int sl = g_sleep;
    while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < epoch * ngrp) {
      if (sl) __builtin_amdgcn_s_sleep(2);
      if (++spins > (1 << 22)) { *err = 1u; ok = false; break; }
    }
  }
  __syncthreads();
  return ok;
}

__global__ __launch_bounds__(256) void k_barriers2(unsigned* cnt, int K, unsigned* err, float* sink) {
  float v = threadIdx.x;
  for (int k = 0; k < K; ++k) {
    v = v * 1.0001f + 0.5f;
    grid_barrier2(cnt, (unsigned)(k + 1), err);
  }
  if (v == 12345.f) sink[0] = v;
}

__global__ __launch_bounds__(256) void k_barriers(unsigned* counter, int K, unsigned* err, float* sink) {
  float v = threadIdx.x;
  for (int k = 0; k < K; ++k) {
    v = v * 1.0001f + 0.5f;
    grid_barrier(counter, (unsigned)(k + 1) * gridDim.x, err);
  }
  if (v == 12345.f) sink[0] = v;
}

// conv + BatchNorm pattern: partial statistics -> barrier -> every workgroup reads the slot sums
template <bool SPLIT_A, bool SPLIT_B>
__global__ __launch_bounds__(256) void k_stats(double* slots, int C, unsigned* counter, unsigned* err, float* out) {
  const int tid = threadIdx.x;
  if (!SPLIT_B) {
    if (tid < C) {
      double* s = slots + (size_t)(blockIdx.x % 8) * 2 * C;
      __hip_atomic_fetch_add(s + tid, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(s + C + tid, 2.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (SPLIT_A) return;
  if (!SPLIT_B) grid_barrier(counter, gridDim.x, err);
  if (tid < C) {
    double a = 0.0, b = 0.0;
    for (int sl = 0; sl < 8; ++sl) {
      a += __hip_atomic_load(slots + (size_t)sl * 2 * C + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      b += __hip_atomic_load(slots + (size_t)sl * 2 * C + C + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (blockIdx.x == 0) { out[tid] = (float)a; out[C + tid] = (float)b; }
    else if (a == 12345.0) out[tid] = (float)b;
  }
}

int main(int argc, char** argv) {
  const int C = 128;
  // round 6 (review item 4): the workgroup counts of the 20x20 layers (64 ... 200) beside the 256 / 512 / 768 of round 5;
  // argv[1] = 0: spin without s_sleep
  const int sleep_on = argc > 1 ? atoi(argv[1]) : 2;
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_sleep), &sleep_on, sizeof(int)));
  printf("spin %s s_sleep\n", sleep_on ? "with" : "without");
  unsigned *counter, *err; double* slots; float *sink, *out;
  unsigned* cnt2; CK(hipMalloc(&cnt2, 9 * 32 * 4));
  CK(hipMalloc(&counter, 4)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&slots, 8 * 2 * C * 8)); CK(hipMalloc(&sink, 4)); CK(hipMalloc(&out, 2 * C * 4));
  CK(hipMemset(err, 0, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int nb = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_barriers, 256, 0));
  printf("occupancy: %d workgroups of 256 threads per CU\n", nb);
  for (int G : {32, 64, 100, 128, 200, 256, 512, 768}) {
    float t0 = 0.f;
    for (int K : {0, 1, 2, 4, 8}) {
      float best = 1e9;
      for (int rep = 0; rep < 7; ++rep) {
        CK(hipMemsetAsync(counter, 0, 4, 0));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_barriers, dim3(G), dim3(256), 0, 0, counter, K, err, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
      }
      if (K == 0) t0 = best;
      printf("G %3d  K %d barriers: %.1f us%s\n", G, K, best * 1e3, K ? "" : "  (empty launch)");
      if (K) printf("        per barrier %.2f us\n", (best - t0) * 1e3 / K);
    }
    {   // two-level barrier, same protocol
      float t02 = 0.f;
      for (int K : {0, 1, 2, 4, 8}) {
        float best = 1e9;
        for (int rep = 0; rep < 7; ++rep) {
          CK(hipMemsetAsync(cnt2, 0, 9 * 32 * 4, 0));
          CK(hipEventRecord(e0));
          hipLaunchKernelGGL(k_barriers2, dim3(G), dim3(256), 0, 0, cnt2, K, err, sink);
          CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
        }
        if (K == 0) t02 = best;
        if (K) printf("G %3d  two-level, K %d barriers: %.1f us, per barrier %.2f us\n", G, K, best * 1e3, (best - t02) * 1e3 / K);
      }
    }
    // statistics pattern, fused vs two launches
    float bf = 1e9, bs = 1e9;
    std::vector<float> h(2 * C);
    for (int rep = 0; rep < 7; ++rep) {
      CK(hipMemsetAsync(counter, 0, 4, 0)); CK(hipMemsetAsync(slots, 0, 8 * 2 * C * 8, 0));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL((k_stats<false, false>), dim3(G), dim3(256), 0, 0, slots, C, counter, err, out);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); bf = ms < bf ? ms : bf;
      CK(hipMemcpy(h.data(), out, 2 * C * 4, hipMemcpyDeviceToHost));
      if (h[0] != (float)G || h[C] != 2.f * G) printf("  fused: WRONG sums %.0f %.0f (want %d %d)\n", h[0], h[C], G, 2 * G);
      CK(hipMemsetAsync(slots, 0, 8 * 2 * C * 8, 0));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL((k_stats<true, false>), dim3(G), dim3(256), 0, 0, slots, C, counter, err, out);
      hipLaunchKernelGGL((k_stats<false, true>), dim3(G), dim3(256), 0, 0, slots, C, counter, err, out);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1)); bs = ms < bs ? ms : bs;
    }
    printf("G %3d  statistics -> barrier -> read back: one launch %.1f us, two launches %.1f us\n", G, bf * 1e3, bs * 1e3);
  }
  unsigned herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
  printf("spin cap hit: %u\n", herr);
  return 0;
}
