// Device-side pieces shared by the kernels that evaluate the closed-form proximal operators of the terms (dpx_elementwise.hip: the staged z /
// dual / rhs passes; dpx_conv_bf16.hip: the head pass of dpx_admm_cg_pnp_iter; dpx_fft.hip: the fused row pass of size-generic planes): one
// definition, so that every path rounds alike.
#pragma once
#include "dpx_common.h"

namespace dpx {

// soft threshold / nonneg / v / (1 + 2 lam)          dprox/proxfn/norm.py:6-27, nonneg.py:10-11, sum_square.py:26-27
__device__ __forceinline__ float prox_eval(int kind, float d, float lam) {
  switch (kind) {
    case DPX_PROX_NORM1: {                                   // sign(d) * max(|d| - lam, 0)
      const float m = fmaxf(fabsf(d) - lam, 0.f);
      return d > 0.f ? m : (d < 0.f ? -m : 0.f * m);
    }
    case DPX_PROX_NONNEG: return fmaxf(d, 0.f);
    case DPX_PROX_SUMSQ: return d / (1.f + 2.f * lam);
    default: return d;
  }
}

struct TermPack {
  dpx_term t[DPX_MAX_TERMS];
  int n;
};

}  // namespace dpx
