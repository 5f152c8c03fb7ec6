// Fused stages of the two splitting methods whose right-hand side nests the Psi operators twice:
//
//   LinearizedADMM (reference dprox/algo/admm.py:78-100):  b_i = x - K_i^T (K_i x - v_i + u_i)
//   PockChambolle  (reference dprox/algo/pc.py:6-40)    :  b_i = x - K_i^T z_i ,   z_i <- z_i + r K_i xbar ;  z_i <- z_i - r prox_i(z_i, r)
//
// and in both the x-update's right-hand side is  K^T b_Omega + rho sum_i K_i^T b_i  (least_squares.rhs, sum_square.py:126-135):
// with K_i in {identity, grad_H, grad_W} that is a radius-2 circular stencil of x, v_i, u_i (or z_i).  One gather kernel evaluates
// it per pixel (the neighbours come from L2 / the Infinity Cache: the planes were just streamed) instead of the 10-15 image passes
// of the op-by-op formulation (forward stencil, AXPY, adjoint stencil, AXPY, adjoint stencil, AXPY per term).
#include "dpx_common.h"

namespace dpx {

struct SplitPack {
  dpx_term t[DPX_MAX_TERMS];
  int n;
};

__device__ __forceinline__ float split_prox(int kind, float d, float lam) {
  if (kind == DPX_PROX_NORM1) {
    const float m = fmaxf(fabsf(d) - lam, 0.f);
    return d > 0.f ? m : (d < 0.f ? -m : 0.f * m);
  }
  if (kind == DPX_PROX_NONNEG) return fmaxf(d, 0.f);
  if (kind == DPX_PROX_SUMSQ) return d / (1.f + 2.f * lam);
  return d;
}

// MODE 0 (Pock-Chambolle): q_i = z_i (terms[i].v).   MODE 1 (linearised ADMM): q_i = (K_i x - v_i) + u_i.
template <int MODE>
__global__ void k_split_rhs(float* __restrict__ rhs, const float* __restrict__ ktb, const float* __restrict__ x, const float* __restrict__ rho,
                            SplitPack T, int B, int C, int H, int W) {
  const long total = (long)B * C * H * W;
  for (long i = (long)xcd_block() * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {      // (XCD-contiguous rows: dpx_common.h)
    const int w = (int)(i % W);
    long r = i / W;
    const int h = (int)(r % H);
    const long pl = r / H;
    const int b = (int)(pl / C);
    const float* xp = x + pl * (long)H * W;
    auto at = [&](const float* p, int hh, int ww) {          // circular indexing, |offset| <= 2
      hh = hh < 0 ? hh + H : (hh >= H ? hh - H : hh);
      ww = ww < 0 ? ww + W : (ww >= W ? ww - W : ww);
      return p[(long)hh * W + ww];
    };
    float acc = 0.f;
    for (int t = 0; t < T.n; ++t) {
      const dpx_term tm = T.t[t];
      const float* vp = tm.v + pl * (long)H * W;
      const float* up = MODE == 1 ? tm.u + pl * (long)H * W : nullptr;
      auto q = [&](int hh, int ww) -> float {
        if constexpr (MODE == 0) {
          return at(vp, hh, ww);
        } else {
          float kx;
          if (tm.linop == DPX_LIN_IDENTITY) kx = at(xp, hh, ww);
          else if (tm.linop == DPX_LIN_GRAD_H) kx = at(xp, hh + 1, ww) - at(xp, hh, ww);
          else kx = at(xp, hh, ww + 1) - at(xp, hh, ww);
          return (kx - at(vp, hh, ww)) + at(up, hh, ww);
        }
      };
      auto bval = [&](int hh, int ww) -> float {             // b_i = x - K_i^T q_i
        float kt;
        if (tm.linop == DPX_LIN_IDENTITY) kt = q(hh, ww);
        else if (tm.linop == DPX_LIN_GRAD_H) kt = q(hh - 1, ww) - q(hh, ww);
        else kt = q(hh, ww - 1) - q(hh, ww);
        return at(xp, hh, ww) - kt;
      };
      float c;                                              // K_i^T b_i
      if (tm.linop == DPX_LIN_IDENTITY) c = bval(h, w);
      else if (tm.linop == DPX_LIN_GRAD_H) c = bval(h - 1, w) - bval(h, w);
      else c = bval(h, w - 1) - bval(h, w);
      acc += c;
    }
    rhs[i] = (ktb ? ktb[i] : 0.f) + rho[b] * acc;
  }
}

// Pock-Chambolle dual step, in place on z_i (terms[i].v):  z += r K_i xbar ;  z -= r prox_i(z, r * alpha)   with r = lam_i[b]
__global__ void k_pc_dual(const float* __restrict__ xbar, SplitPack T, int B, int C, int H, int W) {
  const long total = (long)B * C * H * W;
  for (long i = (long)xcd_block() * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {      // (XCD-contiguous rows: dpx_common.h)
    const int w = (int)(i % W);
    long r = i / W;
    const int h = (int)(r % H);
    const long pl = r / H;
    const int b = (int)(pl / C);
    const float* xp = xbar + pl * (long)H * W;
    const float x0 = xp[(long)h * W + w];
    for (int t = 0; t < T.n; ++t) {
      const dpx_term tm = T.t[t];
      const float step = tm.lam ? tm.lam[b] : 0.f;
      float kx;
      if (tm.linop == DPX_LIN_IDENTITY) kx = x0;
      else if (tm.linop == DPX_LIN_GRAD_H) kx = xp[(long)(h + 1 == H ? 0 : h + 1) * W + w] - x0;
      else kx = xp[(long)h * W + (w + 1 == W ? 0 : w + 1)] - x0;
      float z = tm.v[i] + step * kx;
      z = z - step * split_prox(tm.prox, z, step * tm.alpha);
      tm.v[i] = z;
    }
  }
}

}  // namespace dpx

using namespace dpx;

static int split_pack(SplitPack& P, const dpx_term* terms, int nterms, int need_u, const char* who) {
  DPX_REQUIRE(terms && nterms >= 1 && nterms <= DPX_MAX_TERMS, "%s: 1..%d terms", who, DPX_MAX_TERMS);
  P.n = nterms;
  for (int i = 0; i < nterms; ++i) {
    DPX_REQUIRE(terms[i].linop >= DPX_LIN_IDENTITY && terms[i].linop <= DPX_LIN_GRAD_W, "%s: term %d: unknown linop", who, i);
    DPX_REQUIRE(terms[i].v && (!need_u || terms[i].u), "%s: term %d lacks v / u", who, i);
    P.t[i] = terms[i];
  }
  return DPX_OK;
}

// mode 0: Pock-Chambolle (terms[i].v = z_i), 1: linearised ADMM (terms[i].v / .u = v_i / u_i)
extern "C" int dpx_split_rhs(float* rhs, const float* ktb, const float* x, const float* rho, const dpx_term* terms, int nterms, int mode,
                             int B, int C, int H, int W, dpx_stream_t stream) {
  DPX_REQUIRE(rhs && x && rho && rhs != x && B > 0 && C > 0 && H >= 3 && W >= 3 && (mode == 0 || mode == 1), "dpx_split_rhs: bad arguments");
  SplitPack P;
  const int rc = split_pack(P, terms, nterms, mode == 1, "dpx_split_rhs");
  if (rc) return rc;
  const long n = (long)B * C * H * W;
  if (mode == 0)
    DPX_LAUNCH("k_split_rhs", (k_split_rhs<0>), dim3(grid_for8(n, 256, 8192)), dim3(256), 0, (hipStream_t)stream, rhs, ktb, x, rho, P, B, C, H, W);
  else
    DPX_LAUNCH("k_split_rhs", (k_split_rhs<1>), dim3(grid_for8(n, 256, 8192)), dim3(256), 0, (hipStream_t)stream, rhs, ktb, x, rho, P, B, C, H, W);
  return launch_status("dpx_split_rhs");
}

extern "C" int dpx_pc_dual(const float* xbar, const dpx_term* terms, int nterms, int B, int C, int H, int W, dpx_stream_t stream) {
  DPX_REQUIRE(xbar && B > 0 && C > 0 && H > 0 && W > 0, "dpx_pc_dual: bad arguments");
  SplitPack P;
  const int rc = split_pack(P, terms, nterms, 0, "dpx_pc_dual");
  if (rc) return rc;
  for (int i = 0; i < nterms; ++i) DPX_REQUIRE(terms[i].prox >= DPX_PROX_NORM1 && terms[i].prox <= DPX_PROX_SUMSQ, "dpx_pc_dual: closed-form proxes only");
  DPX_LAUNCH("k_pc_dual", k_pc_dual, dim3(grid_for8((long)B * C * H * W, 256, 8192)), dim3(256), 0, (hipStream_t)stream, xbar, P, B, C, H, W);
  return launch_status("dpx_pc_dual");
}
