// Device-side declarations shared by the row kernels of the unrolled backward iteration (dpx_bwd_rows.hip: lock-step bands;
// dpx_bwd_rows_par.hip: the rows of a band side by side).  Not part of the C ABI.
#pragma once
#include "dpx_fft_reg.h"

namespace dpx {

struct BwdRowTerm {
  int linop, prox;
  float alpha;
  const float* lam;       // [B], iteration t - 1
  const float* v;         // saved prox output of iteration t - 1 (fp32 or bf16 history plane)
  const float* a_in;      // the z stage's share of d/du from the previous backward step (nullable = 0)
  float* a_out;           // g_d of this step
};
struct BwdRowTerms {
  BwdRowTerm t[DPX_MAX_TERMS];
  int n;
  int hist_bf16;
  const float* x;         // history planes of iteration t
  const float* rhs;
  float* g_out;           // nullable: g_rhs as an image ...
  int g_acc;              // ... stored (0) or added to what the plane holds (1: the sum over the iterations, for the offsets' gradient)
};

template <bool BF16> __device__ __forceinline__ float2 hist_pair(const float* plane, size_t pair) {
  if constexpr (BF16) {
    const unsigned u = ((const unsigned*)plane)[pair];
    return make_float2(__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u));
  }
  return ((const float2*)plane)[pair];
}

// one component of the z stage: kg = g_v, a = the dual gradient's other share, v = saved prox output; returns g_d, adds to lt.
// KIND: 1 soft threshold, 2 clipping at zero, 0 v / (1 + 2 lam) with sq = 1 / (1 + 2 lam) -- chosen once per term, outside the element loop
template <int KIND> __device__ __forceinline__ float bwd_gd(float sq, float kg, float a, float v, float& lt) {
  const float gu = a - kg, diff = kg - gu;
  float J, dl;
  if constexpr (KIND == 1) {
    J = v != 0.f ? 1.f : 0.f;
    dl = v > 0.f ? -1.f : (v < 0.f ? 1.f : 0.f);
  } else if constexpr (KIND == 2) {
    J = v > 0.f ? 1.f : 0.f;
    dl = 0.f;
  } else {
    J = sq;
    dl = -2.f * v * sq;
  }
  lt = fmaf(diff, dl, lt);
  return fmaf(J, diff, gu);
}
template <int KIND, int V> __device__ __forceinline__ float bwd_gd_row(float sq, float rho, float2 (&w)[V], const float2 (&av)[V], const float2 (&vv)[V]) {
  float lt = 0.f;
#pragma unroll
  for (int m = 0; m < V; ++m) {
    w[m].x = bwd_gd<KIND>(sq, rho * w[m].x, av[m].x, vv[m].x, lt);
    w[m].y = bwd_gd<KIND>(sq, rho * w[m].y, av[m].y, vv[m].y, lt);
  }
  return lt;
}

__device__ __forceinline__ float bwd_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// dpx_bwd_rows_par.hip: own rows per workgroup of the row-parallel kernel (0: the plane is not taken) and its launch
int bwd_rows_par_own(int P, int H, int W);
int bwd_rows_par_launch(const float2* sin, float2* sout, const BwdRowTerms& TT, const float* rho, float* part_a, float* part_b, float* part_lam, int B,
                        int C, int H, int W, int bands, const float2* twW, hipStream_t s);

}  // namespace dpx
