// Stand-alone probe (hipcc --offload-arch=gfx950 -O3): what does a grid-wide barrier between persistent workgroups cost on this chip, against the
// boundary between two dependent launches?  Decides whether "rows phase -> grid barrier -> columns phase" in one persistent kernel can beat the
// two-kernel iteration of a few-plane launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// sense-reversing barrier on one device-scope counter: last arriver bumps the generation
template <bool SLEEP>
__global__ void k_barrier(unsigned* count, volatile unsigned* gen, int iters, int work, float* sink) {
  float acc = threadIdx.x;
  for (int i = 0; i < iters; ++i) {
    for (int w = 0; w < work; ++w) acc = acc * 1.0001f + 0.5f;            // (a little dependent work between barriers)
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      const unsigned g = __hip_atomic_load((unsigned*)gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__hip_atomic_fetch_add(count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
        __hip_atomic_store(count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add((unsigned*)gen, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        while (__hip_atomic_load((unsigned*)gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == g) {
          if (SLEEP) __builtin_amdgcn_s_sleep(1);
        }
      }
      __threadfence();
    }
    __syncthreads();
  }
  if (acc == 1.2345f) sink[0] = acc;
}
// two levels: the workgroups of one XCD (blockIdx % 8: the dispatcher deals workgroups round-robin over the 8 XCDs) meet on their own counter (its own
// 128-byte line), the last of each XCD goes to the top counter, the last of those bumps the generation everybody spins on
__global__ void k_barrier2(unsigned* base, int iters, int work, float* sink) {
  unsigned* top = base;
  unsigned* gen = base + 32;
  unsigned* mine = base + 64 + 32 * (blockIdx.x & 7);
  const unsigned per_xcd = (gridDim.x >> 3) + ((blockIdx.x & 7) < (gridDim.x & 7) ? 1u : 0u);
  float acc = threadIdx.x;
  for (int i = 0; i < iters; ++i) {
    for (int w = 0; w < work; ++w) acc = acc * 1.0001f + 0.5f;
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned g = __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      bool spin = true;
      if (__hip_atomic_fetch_add(mine, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == per_xcd - 1) {
        __hip_atomic_store(mine, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_fetch_add(top, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == 7) {
          __hip_atomic_store(top, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_fetch_add(gen, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
          spin = false;
        }
      }
      if (spin) while (__hip_atomic_load(gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == g) {}
    }
    __syncthreads();
  }
  if (acc == 1.2345f) sink[0] = acc;
}
// the leanest form: ONE release fence in front of a relaxed arrive, a relaxed spin (no cache invalidation per poll), ONE acquire fence behind it;
// per-XCD counters + a top counter as above.  dirty != nullptr: every workgroup also writes 48 KB of its own between barriers (a phase of a
// 3-plane iteration leaves ~12 MB of dirty lines in the L2s that the release has to write back)
__global__ void k_barrier3(unsigned* base, int iters, int work, float* sink, float* dirty) {
  unsigned* top = base;
  unsigned* gen = base + 32;
  unsigned* mine = base + 64 + 32 * (blockIdx.x & 7);
  const unsigned per_xcd = (gridDim.x >> 3) + ((blockIdx.x & 7) < (gridDim.x & 7) ? 1u : 0u);
  float acc = threadIdx.x;
  for (int i = 0; i < iters; ++i) {
    for (int w = 0; w < work; ++w) acc = acc * 1.0001f + 0.5f;
    if (dirty)
      for (int k = threadIdx.x; k < 12288; k += blockDim.x) dirty[(size_t)blockIdx.x * 12288 + k] = acc + i;
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned g = __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      bool spin = true;
      if (__hip_atomic_fetch_add(mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == per_xcd - 1) {
        __hip_atomic_store(mine, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 7) {
          __hip_atomic_store(top, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_fetch_add(gen, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          spin = false;
        }
      }
      if (spin) while (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == g) {}
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }
  if (acc == 1.2345f) sink[0] = acc;
}
__global__ void k_empty(float* sink, int work) {
  float acc = threadIdx.x;
  for (int w = 0; w < work; ++w) acc = acc * 1.0001f + 0.5f;
  if (acc == 1.2345f) sink[0] = acc;
}

int main() {
  unsigned* cnt;
  float* sink;
  CK(hipMalloc(&cnt, 8));
  CK(hipMalloc(&sink, 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int grid : {192, 256, 512}) {
    for (int threads : {512, 1024}) {
      if (grid == 512 && threads == 1024) continue;                      // (not co-resident)
      CK(hipMemset(cnt, 0, 8));
      const int iters = 2000;
      for (int sl = 0; sl < 2; ++sl)
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemset(cnt, 0, 8));
        CK(hipEventRecord(e0));
        if (sl) hipLaunchKernelGGL(k_barrier<true>, dim3(grid), dim3(threads), 0, 0, cnt, cnt + 1, iters, 64, sink);
        else hipLaunchKernelGGL(k_barrier<false>, dim3(grid), dim3(threads), 0, 0, cnt, cnt + 1, iters, 64, sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("grid barrier (agent-scope atomics%s), %3d workgroups x %4d threads: %6.2f us per barrier (2000 in one launch)\n", sl ? ", s_sleep in the spin" : "", grid, threads, ms * 1e3 / iters);
      }
    }
  }
  {
    unsigned* b2;
    CK(hipMalloc(&b2, 4096));
    for (int grid : {192, 256}) {
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemset(b2, 0, 4096));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_barrier2, dim3(grid), dim3(512), 0, 0, b2, 2000, 64, sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("two-level grid barrier (one counter per XCD + a top counter), %3d workgroups x 512 threads: %6.2f us per barrier\n", grid, ms * 1e3 / 2000);
      }
    }
  }
  {
    unsigned* b3;
    float* dirty;
    CK(hipMalloc(&b3, 4096));
    CK(hipMalloc(&dirty, (size_t)256 * 12288 * 4));
    for (int grid : {192, 256})
      for (int d = 0; d < 2; ++d)
        for (int rep = 0; rep < 2; ++rep) {
          CK(hipMemset(b3, 0, 4096));
          CK(hipEventRecord(e0));
          hipLaunchKernelGGL(k_barrier3, dim3(grid), dim3(512), 0, 0, b3, 2000, 64, sink, d ? dirty : (float*)nullptr);
          CK(hipEventRecord(e1));
          CK(hipEventSynchronize(e1));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          if (rep)
            printf("two-level grid barrier, relaxed atomics + one release / one acquire fence%s, %3d workgroups x 512 threads: %6.2f us per barrier\n",
                   d ? ", 48 KB written per workgroup in between" : "", grid, ms * 1e3 / 2000);
        }
  }
  for (int grid : {192, 256}) {
    const int n = 2000;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(512), 0, 0, sink, 64);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) printf("back-to-back launches, %3d workgroups x 512 threads:   %6.2f us per launch (2000 dependent launches on one stream)\n", grid, ms * 1e3 / n);
    }
  }
  return 0;
}
