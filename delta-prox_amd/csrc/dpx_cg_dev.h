// Device-side pieces of the conjugate-gradient control shared by the kernels that carry it in their epilogues (dpx_cg.hip,
// dpx_elementwise.hip, dpx_fft.hip): the state block, the stop rule, and the "last workgroup to arrive finishes the reduction"
// hand-over (no spinning: a workgroup never waits for another one).
#pragma once
#include "dpx_common.h"

namespace dpx {

// state block: floats gamma[B], gamma_prev[B], beta[B], pAp[B], tol2[B]; then ints done, n_done, it, pad
struct CgState {
  float* f;
  int B;
  __host__ __device__ float* gamma() const { return f; }
  __host__ __device__ float* gamma_prev() const { return f + B; }
  __host__ __device__ float* beta() const { return f + 2 * B; }
  __host__ __device__ float* pAp() const { return f + 3 * B; }
  __host__ __device__ float* tol2() const { return f + 4 * B; }
  __host__ __device__ int* flags() const { return (int*)(f + 5 * B); }     // done, n_done, it
};

// a value another workgroup (possibly behind another XCD's L2) wrote before it took its ticket.  dpx_last_block's acquire fence
// (agent scope: the L2's non-coherent lines are invalidated) makes plain loads safe; atomic (sc1) loads here serialised into one
// memory round trip each -- 13 .. 200 dependent round trips per wave: the fused iteration ran 20 % SLOWER than the unfused one.
__device__ __forceinline__ float dpx_ld_agent(const float* p) { return *p; }

// Every workgroup calls this after its own results are written: release them (agent scope), take a ticket, and learn -- uniformly --
// whether it is the last of `nblocks` to arrive; the last one acquires and resets the counter for the next launch.
//
// The results a workgroup hands over must have been written with dpx_st_agent (write-through stores): the release side is then just
// "my stores have completed" (s_waitcnt) in front of an agent-scope ticket -- a full release fence writes the XCD's whole L2 back,
// once per workgroup: measured 47 us on an 800-workgroup launch whose own work takes 9 us.  Only the last workgroup pays an acquire
// (L2 invalidate) before it reads the others' results with plain loads.
#ifndef DPX_LAST_BLOCK_FULL_FENCE
#define DPX_LAST_BLOCK_FULL_FENCE 0
#endif
__device__ __forceinline__ void dpx_st_agent(float* p, float v) {
#ifdef DPX_EMULATED
  *p = v;
#else
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ bool dpx_last_block(unsigned* counter, unsigned nblocks, int* sh_flag) {
#ifdef DPX_EMULATED
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned ticket = atomicAdd(counter, 1u);
    *sh_flag = (ticket == nblocks - 1);
    if (ticket == nblocks - 1) *counter = 0u;
  }
  __syncthreads();
  return *sh_flag != 0;
#else
  if (DPX_LAST_BLOCK_FULL_FENCE) __threadfence();
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *sh_flag = (ticket == nblocks - 1);
    if (ticket == nblocks - 1) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  const bool last = *sh_flag != 0;
  if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return last;
#endif
}

// The stop rule of cg() (solver_cg.py:103-104) and, if the solve goes on, beta / gamma of the next iteration (:109-115), from the
// B x B Gram matrix G of the residuals.  M: 64 * 65 doubles and sh: 2 ints of shared memory; every thread of the workgroup calls it
// (block-uniform control flow; blockDim.x >= 64).  init_rtol >= 0 on the FIRST test of a solve: the tolerances rtol ||b_i|| come
// from G's diagonal (r = b then) and the rest of the state is initialised (what k_cg_init does for the step-by-step interface).
//   lambda_max(R R^T) <= tau^2  <=>  tau^2 I - sym(G) positive semidefinite: right-looking LDL^T without pivoting in float64,
//   PSD <=> every pivot >= 0 (a zero pivot must come with a zero column); a NaN anywhere fails the test.
__device__ __forceinline__ void cg_test_block(CgState S, const float* __restrict__ G, double* M, int* sh, float init_rtol) {
  const int B = S.B, t = threadIdx.x, nthr = blockDim.x;
  int* fl = S.flags();
  // (everything the test needs from the state is requested at once -- the two control words, the tolerances, gamma of the previous
  //  iteration: one memory round trip in front of the factorisation instead of three or four; this block is the tail of a launch that is
  //  a few microseconds long)
  const int f0 = fl[0], f2 = fl[2];
  const float gprev = t < B ? S.gamma_prev()[t] : 1.f;
  float tau2 = INFINITY;
  for (int i = 0; i < B; ++i) tau2 = fminf(tau2, S.tol2()[i]);
  if (f0) return;                                           // converged earlier: the solve is frozen (uniform)
  if (init_rtol >= 0.f && f2 == 0) {
    if (t < B) {
      const float nb = sqrtf(fmaxf(G[t * B + t], 0.f));
      const float tl = init_rtol * nb;
      S.tol2()[t] = tl * tl;
      S.gamma()[t] = 0.f;
      S.gamma_prev()[t] = 1.f;
      S.beta()[t] = 0.f;
      S.pAp()[t] = 1.f;
    }
    __syncthreads();
    tau2 = INFINITY;                                        // (the tolerances just written, not what was there before)
    for (int i = 0; i < B; ++i) tau2 = fminf(tau2, S.tol2()[i]);
  }
  for (int e = t; e < B * B; e += nthr) {
    const int i = e / B, j = e - i * B;
    const double g = 0.5 * ((double)G[i * B + j] + (double)G[j * B + i]);
    M[i * 65 + j] = (i == j ? (double)tau2 : 0.0) - g;
  }
  if (t == 0) sh[0] = 1;
  __syncthreads();
  double scale = 0.0;                                       // scale-aware zero threshold: round-off of the fp32 Gram entries
  for (int i = 0; i < B; ++i) scale = fmax(scale, fabs((double)G[i * B + i]));
  const double tiny = 1e-12 * fmax(scale, (double)tau2);
  for (int k = 0; k < B; ++k) {
    const double d = M[k * 65 + k];
    if (!(d >= 0.0)) {                                      // negative or NaN pivot (every thread reads the same value)
      if (t == 0) sh[0] = 0;
      break;
    }
    if (d <= tiny) {                                        // zero pivot: PSD only if the rest of the column vanishes
      if (t == 0) sh[1] = 0;
      __syncthreads();
      for (int i = k + 1 + t; i < B; i += nthr)
        if (fabs(M[i * 65 + k]) > tiny) sh[1] = 1;
      __syncthreads();
      if (sh[1]) {
        if (t == 0) sh[0] = 0;
        break;
      }
      continue;
    }
    const double inv = 1.0 / d;
    for (int i = k + 1 + t; i < B; i += nthr) {             // thread i owns row i of the trailing block
      const double l = M[i * 65 + k] * inv;
      for (int j = k + 1; j <= i; ++j) M[i * 65 + j] -= l * M[j * 65 + k];
    }
    __syncthreads();
  }
  __syncthreads();
  if (sh[0]) {
    if (t == 0) {
      fl[0] = 1;
      fl[1] = f2;
    }
    return;
  }
  const bool first = f2 == 0;
  __syncthreads();
  if (t < B) {
    const float g = G[t * B + t];                           // gamma_i = <r_i, r_i>
    S.beta()[t] = first ? 0.f : g / gprev;                  // beta = gamma / gamma_1   (solver_cg.py:112)
    S.gamma()[t] = g;
    S.gamma_prev()[t] = g;
  }
  if (t == 0) fl[2] = f2 + 1;
}

}  // namespace dpx
