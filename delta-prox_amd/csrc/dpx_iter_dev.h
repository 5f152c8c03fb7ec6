// Device-side declarations shared by the row kernels of the two-kernel ADMM iteration (dpx_iter.hip: the streaming and the lock-step
// kernel; dpx_iter_par.hip: the row-parallel kernel of launches with few planes).  Not part of the C ABI.
#pragma once
#include "dpx_fft_reg.h"

namespace dpx {

struct IterTerm {
  int linop, prox;
  float alpha;
  const float* lam;
  const float* u_in;
  float* u_out;
  float* v_out;
};
struct IterTerms {
  IterTerm t[DPX_MAX_TERMS];
  int n;
  float dual;          // 1: ADMM.  0: half-quadratic splitting (DPX_TERM_NO_DUAL) -- the incoming duals count as zero and the right-hand side
                       // sees v_i alone; applied as fma(dual, u, K x) / fma(-dual, u', v): the same instructions, and exact for dual = 1
  int emit_bf16;       // x_out / v_out / rhs_out are bf16 planes (the bf16 history of the unrolled forward pass) instead of fp32
  int vxu;             // 1: ADMM in the order v, x, u (DPX_TERM_VXU, algo/admm.py:103-120) -- the stream in `u` carries q = u' - v (u' = -u):
                       // t = q + x, d = K x + t, v = prox(d), q' = t - v goes out, the next right-hand side sees v - t
  int u_live;          // 0: the incoming duals are all zero (DPX_TERM_U_ZERO, first iteration after ADMM.initialize) -- the streaming kernel then
                       // fetches every u row from row 0 of plane 0 (cache hits) instead of streaming the planes from HBM, the lock-step
                       // kernel does not load them at all
  float* rhs_out;      // nullable: the next x-update's right-hand-side increment rho' sum K_i^T (v_i - u_i) as an image (the unrolled
                       // forward pass keeps it for the backward pass); like x_out / v_out an emit store, never counted in the waits
};

__device__ __forceinline__ float prox1(int kind, float d, float lam) {
  if (kind == DPX_PROX_NORM1) {
    const float m = fmaxf(fabsf(d) - lam, 0.f);
    return d > 0.f ? m : (d < 0.f ? -m : 0.f * m);
  }
  if (kind == DPX_PROX_NONNEG) return fmaxf(d, 0.f);
  return d / (1.f + 2.f * lam);
}

// Soft threshold of V pixel pairs, v = sign(d) max(|d| - lam, 0) (proxfn/norm.py), as  d - median(d, -lam, lam): one v_med3_f32 per
// value and a packed subtraction per pair instead of the compare / select form's six vector instructions per value -- the same
// values (d - lam and d + lam are the same single roundings as |d| - lam with the sign put back; inside the band d - d = 0).  The
// median form needs lam >= 0: a wave that holds a negative threshold anywhere (one vote per term and row) takes the general form.
template <int V> __device__ __forceinline__ void soft_threshold_pairs(const float2 (&d)[V], float2 (&v)[V], float lam) {
  if (__ballot(lam < 0.f) == 0) {
#pragma unroll
    for (int m = 0; m < V; ++m) {
#ifdef DPX_EMULATED
      const float cx = fminf(fmaxf(d[m].x, -lam), lam), cy = fminf(fmaxf(d[m].y, -lam), lam);
#else
      const float cx = __builtin_amdgcn_fmed3f(d[m].x, -lam, lam), cy = __builtin_amdgcn_fmed3f(d[m].y, -lam, lam);
#endif
      v[m] = make_float2(d[m].x - cx, d[m].y - cy);
    }
  } else {
#pragma unroll
    for (int m = 0; m < V; ++m) v[m] = make_float2(prox1(DPX_PROX_NORM1, d[m].x, lam), prox1(DPX_PROX_NORM1, d[m].y, lam));
  }
}

// Cache policy of the row kernels' four streams (dpx_common.h): spectrum in / u in (loads, 1 = nt), u out / spectrum out
// (stores, 2 = nt); see the comment in front of k_iter_rows_seq (dpx_iter.hip).
#ifndef DPX_R_LDX
#define DPX_R_LDX 0
#endif
#ifndef DPX_R_LDU
#define DPX_R_LDU 1
#endif
#ifndef DPX_R_STU
#define DPX_R_STU 2
#endif
#ifndef DPX_R_STX
#define DPX_R_STX 1       // the spectrum handed to the next kernel: write-through (`sc1`), nothing left dirty at the kernel boundary (+0.5 ... 1 %)
#endif
constexpr int R_LDX = DPX_R_LDX, R_LDU = DPX_R_LDU, R_STU = DPX_R_STU, R_STX = DPX_R_STX;

// dpx_iter_par.hip: the row pass for launches of a few planes (false: not applicable -- the caller keeps the streaming kernel)
bool launch_iter_rows_par(const float2* sin, float2* sout, const IterTerms& TT, const float* rho_next, float* x_out, int emit_v, int C, int H, int W,
                          int P, const float2* twW, hipStream_t s, bool forced);

}  // namespace dpx
