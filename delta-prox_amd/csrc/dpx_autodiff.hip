// Fused backward stages of the ADMM iteration (config 5: unrolled training).  The reference differentiates
// dprox/algo/admm.py:49-59 with PyTorch autograd through ~80 eager ops per iteration; here each of the three forward
// stages (dpx_admm_rhs, dpx_fourier_solve, dpx_admm_zupdate) has one backward kernel (+ the transforms of
// dpx_fourier_apply_inv), with the per-image scalar gradients (d/d rho, d/d lambda) reduced deterministically:
// per-block partial sums (wave shuffles + one LDS hop) and a fixed-order finishing pass, no atomics.
//
//   z/dual stage  d = K x + u, v = prox(d, lam), u' = d - v:
//       g_d = J^T (g_v - g_u') + g_u',  g_u = g_d,  g_x = sum_i K_i^T g_d,  g_lam = alpha <g_v - g_u', dprox/dlam>
//       (J and dprox/dlam are recovered from the saved OUTPUT v: soft-threshold passes where v != 0, nonneg where v > 0)
//   x stage       g_rho = -<g_rhs, sum_i K_i^T K_i x>            (g_rhs = M g_x comes from dpx_fourier_apply_inv)
//   rhs stage     g_v_i = rho K_i g, g_u_i = -g_v_i, g_rho = <g, rhs> / rho
#include "dpx_cg_dev.h"

namespace dpx {

__device__ __forceinline__ float ad_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float ad_block_sum(float v, float* sh) {
  v = ad_wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wid] = v;
  __syncthreads();
  float r = 0.f;
  if (wid == 0) {
    r = lane < (int)(blockDim.x >> 6) ? sh[lane] : 0.f;
    r = ad_wave_sum(r);
  }
  return r;                                              // valid in thread 0
}

struct BwdTerm {
  int linop, prox;
  float alpha;
  const float* lam;
  const float* v;       // saved forward output of the prox
  const float* gv;      // incoming gradients (nullable = 0)
  const float* gun;     //   ... w.r.t. the updated dual u'
  float* gu;            // outgoing gradient w.r.t. the incoming dual u
};
struct BwdPack {
  BwdTerm t[DPX_MAX_TERMS];
  int n;
  int hist_bf16;        // the saved planes (v here, x / rhs in the other two stages) are bf16 history slots, not fp32
};

// g_d of one term at one pixel (offset i), and dprox/dlam * (g_v - g_u') for the lambda gradient
__device__ __forceinline__ float zb_gd(const BwdTerm& tm, long i, float lam, float& lam_term, int hist_bf16) {
  const float gv = tm.gv ? tm.gv[i] : 0.f, gu = tm.gun ? tm.gun[i] : 0.f;
  const float diff = gv - gu, v = dpx_hist_load(tm.v, hist_bf16, i);
  float J, dl;
  if (tm.prox == DPX_PROX_NORM1) {
    J = v != 0.f ? 1.f : 0.f;
    dl = v > 0.f ? -1.f : (v < 0.f ? 1.f : 0.f);
  } else if (tm.prox == DPX_PROX_NONNEG) {
    J = v > 0.f ? 1.f : 0.f;
    dl = 0.f;
  } else {
    const float s = 1.f / (1.f + 2.f * lam);
    J = s;
    dl = -2.f * v * s;                                   // d = v (1 + 2 lam):  -2 d s^2 = -2 v s
  }
  lam_term = diff * dl;
  return fmaf(J, diff, gu);
}

// grid (blocks, B): one image per blockIdx.y so that the lambda partial sums stay per image
__global__ void __launch_bounds__(256) k_zupdate_bwd(float* __restrict__ gx, BwdPack T, float* __restrict__ part, int C, int H, int W) {
  __shared__ float sh[16];
  const int b = blockIdx.y;
  const long npb = (long)C * H * W, base = (long)b * npb;
  float lsum[DPX_MAX_TERMS];
#pragma unroll
  for (int t = 0; t < DPX_MAX_TERMS; ++t) lsum[t] = 0.f;
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npb; p += (long)gridDim.x * 256) {
    const int w = (int)(p % W);
    const long row = p / W;
    const int h = (int)(row % H);
    const long i = base + p;
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < DPX_MAX_TERMS; ++t) {
      if (t < T.n) {
        const BwdTerm& tm = T.t[t];
        const float lam = tm.lam ? tm.lam[b] * tm.alpha : 0.f;
        float lt, dummy;
        const float gd = zb_gd(tm, i, lam, lt, T.hist_bf16);
        lsum[t] += lt;
        tm.gu[i] = gd;
        if (tm.linop == DPX_LIN_IDENTITY) {
          acc += gd;
        } else if (tm.linop == DPX_LIN_GRAD_W) {           // adjoint: y[w-1] - y[w]
          const long il = base + row * W + (w == 0 ? W - 1 : w - 1);
          acc += zb_gd(tm, il, lam, dummy, T.hist_bf16) - gd;
        } else {                                            // grad_H adjoint: y[h-1] - y[h]
          const long iu = i + (long)((h == 0 ? H - 1 : h - 1) - h) * W;
          acc += zb_gd(tm, iu, lam, dummy, T.hist_bf16) - gd;
        }
      }
    }
    gx[i] = acc;
  }
  for (int t = 0; t < T.n; ++t) {
    const float s = ad_block_sum(lsum[t], sh);
    if (threadIdx.x == 0) part[((long)t * gridDim.y + b) * gridDim.x + blockIdx.x] = s * T.t[t].alpha;
  }
}

// the same stage, four pixels of a row per thread (W % 4 == 0): 16-byte accesses for every plane; the left neighbour of the first
// pixel and the row above are the only extra loads of the two stencil adjoints
__device__ __forceinline__ void zb_gd4(const BwdTerm& tm, long i, float lam, int hist_bf16, float (&gd)[4], float (&lt)[4]) {
  const float4 gv = tm.gv ? *(const float4*)(tm.gv + i) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 gu = tm.gun ? *(const float4*)(tm.gun + i) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 v4 = dpx_hist_load4(tm.v, hist_bf16, i);
  const float gva[4] = {gv.x, gv.y, gv.z, gv.w}, gua[4] = {gu.x, gu.y, gu.z, gu.w}, va[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float diff = gva[k] - gua[k], v = va[k];
    float J, dl;
    if (tm.prox == DPX_PROX_NORM1) {
      J = v != 0.f ? 1.f : 0.f;
      dl = v > 0.f ? -1.f : (v < 0.f ? 1.f : 0.f);
    } else if (tm.prox == DPX_PROX_NONNEG) {
      J = v > 0.f ? 1.f : 0.f;
      dl = 0.f;
    } else {
      const float s = 1.f / (1.f + 2.f * lam);
      J = s;
      dl = -2.f * v * s;
    }
    lt[k] = diff * dl;
    gd[k] = fmaf(J, diff, gua[k]);
  }
}
__global__ void __launch_bounds__(256) k_zupdate_bwd4(float* __restrict__ gx, BwdPack T, float* __restrict__ part, int C, int H, int W) {
  __shared__ float sh[16];
  const int b = blockIdx.y;
  const long npb = (long)C * H * W, base = (long)b * npb;
  float lsum[DPX_MAX_TERMS];
#pragma unroll
  for (int t = 0; t < DPX_MAX_TERMS; ++t) lsum[t] = 0.f;
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < npb / 4; q += (long)gridDim.x * 256) {
    const long p = q * 4;
    const int w = (int)(p % W);
    const long row = p / W;
    const int h = (int)(row % H);
    const long i = base + p;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < DPX_MAX_TERMS; ++t) {
      if (t < T.n) {
        const BwdTerm& tm = T.t[t];
        const float lam = tm.lam ? tm.lam[b] * tm.alpha : 0.f;
        float gd[4], lt[4];
        zb_gd4(tm, i, lam, T.hist_bf16, gd, lt);
        lsum[t] += (lt[0] + lt[1]) + (lt[2] + lt[3]);
        *(float4*)(tm.gu + i) = make_float4(gd[0], gd[1], gd[2], gd[3]);
        if (tm.linop == DPX_LIN_IDENTITY) {
#pragma unroll
          for (int k = 0; k < 4; ++k) acc[k] += gd[k];
        } else if (tm.linop == DPX_LIN_GRAD_W) {           // adjoint: y[w-1] - y[w]
          float dummy;
          const float left = zb_gd(tm, base + row * W + (w == 0 ? W - 1 : w - 1), lam, dummy, T.hist_bf16);
          acc[0] += left - gd[0];
#pragma unroll
          for (int k = 1; k < 4; ++k) acc[k] += gd[k - 1] - gd[k];
        } else {                                            // grad_H adjoint: y[h-1] - y[h]
          float up[4], dl[4];
          zb_gd4(tm, i + (long)((h == 0 ? H - 1 : h - 1) - h) * W, lam, T.hist_bf16, up, dl);
#pragma unroll
          for (int k = 0; k < 4; ++k) acc[k] += up[k] - gd[k];
        }
      }
    }
    *(float4*)(gx + i) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  }
  for (int t = 0; t < T.n; ++t) {
    const float s = ad_block_sum(lsum[t], sh);
    if (threadIdx.x == 0) part[((long)t * gridDim.y + b) * gridDim.x + blockIdx.x] = s * T.t[t].alpha;
  }
}

struct LinCodes {
  int linop[DPX_MAX_TERMS];
  int n;
  int hist_bf16;
};

// part[b][blk] = - sum_p g[p] * (sum_i K_i^T K_i x)[p]
__global__ void __launch_bounds__(256) k_solve_rho_grad(const float* __restrict__ g, const float* __restrict__ x, LinCodes L,
                                                         float* __restrict__ part, int C, int H, int W) {
  __shared__ float sh[16];
  const int b = blockIdx.y;
  const long npb = (long)C * H * W, base = (long)b * npb;
  float cI = 0.f;
  bool hasW = false, hasH = false;
  int nW = 0, nH = 0;
  for (int t = 0; t < L.n; ++t) {
    if (L.linop[t] == DPX_LIN_IDENTITY) cI += 1.f;
    else if (L.linop[t] == DPX_LIN_GRAD_W) { hasW = true; ++nW; }
    else { hasH = true; ++nH; }
  }
  float acc = 0.f;
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npb; p += (long)gridDim.x * 256) {
    const int w = (int)(p % W);
    const long row = p / W;
    const int h = (int)(row % H);
    const long i = base + p;
    const float xc = dpx_hist_load(x, L.hist_bf16, i);
    float lx = cI * xc;
    if (hasW) {
      const float xl = dpx_hist_load(x, L.hist_bf16, base + row * W + (w == 0 ? W - 1 : w - 1)), xr = dpx_hist_load(x, L.hist_bf16, base + row * W + (w + 1 == W ? 0 : w + 1));
      lx += (float)nW * (2.f * xc - xl - xr);             // K^T K of the circular forward difference
    }
    if (hasH) {
      const float xu = dpx_hist_load(x, L.hist_bf16, i + (long)((h == 0 ? H - 1 : h - 1) - h) * W), xd = dpx_hist_load(x, L.hist_bf16, i + (long)((h + 1 == H ? 0 : h + 1) - h) * W);
      lx += (float)nH * (2.f * xc - xu - xd);
    }
    acc = fmaf(g[i], lx, acc);
  }
  const float s = ad_block_sum(acc, sh);
  if (threadIdx.x == 0) part[(long)b * gridDim.x + blockIdx.x] = -s;
}

struct RhsBwdPack {
  int linop[DPX_MAX_TERMS];
  float* gv[DPX_MAX_TERMS];
  float* gu[DPX_MAX_TERMS];
  const float* gu_add[DPX_MAX_TERMS];      // nullable: gu = gu_add - gv  (the z stage's share of the gradient w.r.t. the dual)
  int n;
  int hist_bf16;
};

// g_v_i = rho_b K_i g, g_u_i = (gu_add_i) - g_v_i; part[b][blk] = sum g * rhs  (the finishing pass divides by rho)
__global__ void __launch_bounds__(256) k_rhs_bwd(const float* __restrict__ g, const float* __restrict__ rhs, const float* __restrict__ rho,
                                                  RhsBwdPack T, float* __restrict__ part, int C, int H, int W) {
  __shared__ float sh[16];
  const int b = blockIdx.y;
  const long npb = (long)C * H * W, base = (long)b * npb;
  const float r = rho[b];
  float acc = 0.f;
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npb; p += (long)gridDim.x * 256) {
    const int w = (int)(p % W);
    const long row = p / W;
    const int h = (int)(row % H);
    const long i = base + p;
    const float gc = g[i];
    acc = fmaf(gc, dpx_hist_load(rhs, T.hist_bf16, i), acc);
#pragma unroll
    for (int t = 0; t < DPX_MAX_TERMS; ++t) {
      if (t < T.n) {
        float kg;
        if (T.linop[t] == DPX_LIN_IDENTITY) kg = gc;
        else if (T.linop[t] == DPX_LIN_GRAD_W) kg = g[base + row * W + (w + 1 == W ? 0 : w + 1)] - gc;
        else kg = g[i + (long)((h + 1 == H ? 0 : h + 1) - h) * W] - gc;
        kg *= r;
        if (T.gv[t]) T.gv[t][i] = kg;
        if (T.gu[t]) T.gu[t][i] = (T.gu_add[t] ? T.gu_add[t][i] : 0.f) - kg;
      }
    }
  }
  const float s = ad_block_sum(acc, sh);
  if (threadIdx.x == 0) part[(long)b * gridDim.x + blockIdx.x] = s;
}

// k_solve_rho_grad and k_rhs_bwd in one pass over g (the unrolled backward loop runs them back to back on the same g = M_rho g_x):
// part_a[b][blk] = - sum g (sum K_i^T K_i x),  part_b[b][blk] = sum g rhs,  g_v_i = rho_b K_i g,  g_u_i = gu_add_i - g_v_i
__global__ void __launch_bounds__(256) k_solve_rhs_bwd(const float* __restrict__ g, const float* __restrict__ x, const float* __restrict__ rhs,
                                                        const float* __restrict__ rho, RhsBwdPack T, float* __restrict__ part_a,
                                                        float* __restrict__ part_b, int C, int H, int W) {
  __shared__ float sh[16];
  const int b = blockIdx.y;
  const long npb = (long)C * H * W, base = (long)b * npb;
  const float r = rho[b];
  float cI = 0.f;
  int nW = 0, nH = 0;
  for (int t = 0; t < T.n; ++t) {
    if (T.linop[t] == DPX_LIN_IDENTITY) cI += 1.f;
    else if (T.linop[t] == DPX_LIN_GRAD_W) ++nW;
    else ++nH;
  }
  float acc_a = 0.f, acc_b = 0.f;
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npb; p += (long)gridDim.x * 256) {
    const int w = (int)(p % W);
    const long row = p / W;
    const int h = (int)(row % H);
    const long i = base + p;
    const long il = base + row * W + (w == 0 ? W - 1 : w - 1), ir = base + row * W + (w + 1 == W ? 0 : w + 1);
    const long iu = i + (long)((h == 0 ? H - 1 : h - 1) - h) * W, id = i + (long)((h + 1 == H ? 0 : h + 1) - h) * W;
    const float gc = g[i];
    const float xc = dpx_hist_load(x, T.hist_bf16, i);
    float lx = cI * xc;
    if (nW) lx += (float)nW * (2.f * xc - dpx_hist_load(x, T.hist_bf16, il) - dpx_hist_load(x, T.hist_bf16, ir));
    if (nH) lx += (float)nH * (2.f * xc - dpx_hist_load(x, T.hist_bf16, iu) - dpx_hist_load(x, T.hist_bf16, id));
    acc_a = fmaf(gc, lx, acc_a);
    acc_b = fmaf(gc, dpx_hist_load(rhs, T.hist_bf16, i), acc_b);
#pragma unroll
    for (int t = 0; t < DPX_MAX_TERMS; ++t) {
      if (t < T.n) {
        float kg;
        if (T.linop[t] == DPX_LIN_IDENTITY) kg = gc;
        else if (T.linop[t] == DPX_LIN_GRAD_W) kg = g[ir] - gc;
        else kg = g[id] - gc;
        kg *= r;
        if (T.gv[t]) T.gv[t][i] = kg;
        if (T.gu[t]) T.gu[t][i] = (T.gu_add[t] ? T.gu_add[t][i] : 0.f) - kg;
      }
    }
  }
  const float sa = ad_block_sum(acc_a, sh);
  __syncthreads();
  const float sb = ad_block_sum(acc_b, sh);
  if (threadIdx.x == 0) {
    part_a[(long)b * gridDim.x + blockIdx.x] = -sa;
    part_b[(long)b * gridDim.x + blockIdx.x] = sb;
  }
}

// ... four pixels of a row per thread (W % 4 == 0)
__global__ void __launch_bounds__(256) k_solve_rhs_bwd4(const float* __restrict__ g, const float* __restrict__ x, const float* __restrict__ rhs,
                                                         const float* __restrict__ rho, RhsBwdPack T, float* __restrict__ part_a,
                                                         float* __restrict__ part_b, int C, int H, int W) {
  __shared__ float sh[16];
  const int b = blockIdx.y;
  const long npb = (long)C * H * W, base = (long)b * npb;
  const float r = rho[b];
  float cI = 0.f;
  int nW = 0, nH = 0;
  for (int t = 0; t < T.n; ++t) {
    if (T.linop[t] == DPX_LIN_IDENTITY) cI += 1.f;
    else if (T.linop[t] == DPX_LIN_GRAD_W) ++nW;
    else ++nH;
  }
  float acc_a = 0.f, acc_b = 0.f;
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < npb / 4; q += (long)gridDim.x * 256) {
    const long p = q * 4;
    const int w = (int)(p % W);
    const long row = p / W;
    const int h = (int)(row % H);
    const long i = base + p, rowb = base + row * W;
    const long up = (long)((h == 0 ? H - 1 : h - 1) - h) * W, dn = (long)((h + 1 == H ? 0 : h + 1) - h) * W;
    const float4 g4 = *(const float4*)(g + i);
    const float ga[4] = {g4.x, g4.y, g4.z, g4.w};
    const float4 x4 = dpx_hist_load4(x, T.hist_bf16, i), r4 = dpx_hist_load4(rhs, T.hist_bf16, i);
    const float xa[4] = {x4.x, x4.y, x4.z, x4.w}, ra[4] = {r4.x, r4.y, r4.z, r4.w};
    float lx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) lx[k] = cI * xa[k];
    if (nW) {
      const float xl = dpx_hist_load(x, T.hist_bf16, rowb + (w == 0 ? W - 1 : w - 1)), xr = dpx_hist_load(x, T.hist_bf16, rowb + (w + 4 == W ? 0 : w + 4));
      const float xe[6] = {xl, xa[0], xa[1], xa[2], xa[3], xr};
#pragma unroll
      for (int k = 0; k < 4; ++k) lx[k] += (float)nW * (2.f * xe[k + 1] - xe[k] - xe[k + 2]);
    }
    float gdn[4] = {0.f, 0.f, 0.f, 0.f};
    if (nH) {
      const float4 xu = dpx_hist_load4(x, T.hist_bf16, i + up), xd = dpx_hist_load4(x, T.hist_bf16, i + dn);
      const float xua[4] = {xu.x, xu.y, xu.z, xu.w}, xda[4] = {xd.x, xd.y, xd.z, xd.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) lx[k] += (float)nH * (2.f * xa[k] - xua[k] - xda[k]);
      const float4 gd4 = *(const float4*)(g + i + dn);
      gdn[0] = gd4.x; gdn[1] = gd4.y; gdn[2] = gd4.z; gdn[3] = gd4.w;
    }
    const float gright = nW ? g[rowb + (w + 4 == W ? 0 : w + 4)] : 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      acc_a = fmaf(ga[k], lx[k], acc_a);
      acc_b = fmaf(ga[k], ra[k], acc_b);
    }
#pragma unroll
    for (int t = 0; t < DPX_MAX_TERMS; ++t) {
      if (t < T.n) {
        float kg[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (T.linop[t] == DPX_LIN_IDENTITY) kg[k] = ga[k];
          else if (T.linop[t] == DPX_LIN_GRAD_W) kg[k] = (k < 3 ? ga[(k + 1) & 3] : gright) - ga[k];
          else kg[k] = gdn[k] - ga[k];
          kg[k] *= r;
        }
        if (T.gv[t]) *(float4*)(T.gv[t] + i) = make_float4(kg[0], kg[1], kg[2], kg[3]);
        if (T.gu[t]) {
          const float4 a4 = T.gu_add[t] ? *(const float4*)(T.gu_add[t] + i) : make_float4(0.f, 0.f, 0.f, 0.f);
          *(float4*)(T.gu[t] + i) = make_float4(a4.x - kg[0], a4.y - kg[1], a4.z - kg[2], a4.w - kg[3]);
        }
      }
    }
  }
  const float sa = ad_block_sum(acc_a, sh);
  __syncthreads();
  const float sb = ad_block_sum(acc_b, sh);
  if (threadIdx.x == 0) {
    part_a[(long)b * gridDim.x + blockIdx.x] = -sa;
    part_b[(long)b * gridDim.x + blockIdx.x] = sb;
  }
}

__global__ void k_ad_finish(const float* __restrict__ part, float* __restrict__ out, int nblk, const float* __restrict__ div,
                            const float* __restrict__ add) {
  __shared__ float sh[16];
  float acc = 0.f;
  for (int i = threadIdx.x; i < nblk; i += blockDim.x) acc += part[(long)blockIdx.x * nblk + i];
  acc = ad_block_sum(acc, sh);
  if (threadIdx.x == 0) out[blockIdx.x] = (div ? acc / div[blockIdx.x] : acc) + (add ? add[blockIdx.x] : 0.f);
}

// the three reductions of one unrolled backward iteration in one launch: block j < n B: glam[j] = sum part_lam[j][*];
// block n B + b: grho[b] = sum part_a[b][*] + (sum part_b[b][*]) / rho[b]     (same summation order as three k_ad_finish launches)
__global__ void k_ad_finish_iter(const float* __restrict__ part_lam, const float* __restrict__ part_a, const float* __restrict__ part_b,
                                 float* __restrict__ glam, float* __restrict__ grho, const float* __restrict__ rho, int nB, int nblk) {
  __shared__ float sh[16];
  const int j = blockIdx.x;
  if (j < nB) {
    float acc = 0.f;
    for (int i = threadIdx.x; i < nblk; i += blockDim.x) acc += part_lam[(long)j * nblk + i];
    acc = ad_block_sum(acc, sh);
    if (threadIdx.x == 0) glam[j] = acc;
  } else {
    const int b = j - nB;
    float a = 0.f, c = 0.f;
    for (int i = threadIdx.x; i < nblk; i += blockDim.x) a += part_a[(long)b * nblk + i];
    a = ad_block_sum(a, sh);
    __syncthreads();
    for (int i = threadIdx.x; i < nblk; i += blockDim.x) c += part_b[(long)b * nblk + i];
    c = ad_block_sum(c, sh);
    if (threadIdx.x == 0) grho[b] = a + c / rho[b];
  }
}

// the same for ALL iterations of the two-kernel backward loop at once (blockIdx.y = stage k = 0 .. nst - 1, the stage of loop iteration
// it = T - 1 - k): its partial sums sit at part + k * stride as [n B rows of lambda][B rows a][B rows b], nblk slots per row; it finishes
// glam[(it - 1) n B + j] and grho[it B + b] with rho_tab[it B + b].  One launch behind the loop instead of one per iteration.
__global__ void k_ad_finish_all(const float* __restrict__ part, long stride, float* __restrict__ glam, float* __restrict__ grho,
                                const float* __restrict__ rho_tab, int nB, int B, int nblk, int T) {
  __shared__ float sh[16];
  const int j = blockIdx.x, it = T - 1 - (int)blockIdx.y;
  const float* p = part + (long)blockIdx.y * stride;
  if (j < nB) {
    float acc = 0.f;
    for (int i = threadIdx.x; i < nblk; i += blockDim.x) acc += p[(long)j * nblk + i];
    acc = ad_block_sum(acc, sh);
    if (threadIdx.x == 0) glam[(long)(it - 1) * nB + j] = acc;
  } else {
    const int b = j - nB;
    const float* pa = p + (long)nB * nblk;
    const float* pb = pa + (long)B * nblk;
    float a = 0.f, c = 0.f;
    for (int i = threadIdx.x; i < nblk; i += blockDim.x) a += pa[(long)b * nblk + i];
    a = ad_block_sum(a, sh);
    __syncthreads();
    for (int i = threadIdx.x; i < nblk; i += blockDim.x) c += pb[(long)b * nblk + i];
    c = ad_block_sum(c, sh);
    if (threadIdx.x == 0) grho[(long)it * B + b] = a + c / rho_tab[(long)it * B + b];
  }
}

// Gradient of the Fourier-domain x-update w.r.t. the OTF of a convolutional data term (end-to-end optics: the PSF of conv_doe is
// trained through the unrolled solver, reference README.md:93-116, linop/conv.py:81-156).  With X = (R + conj(O) Y + eps) / D,
// D = |O|^2 + rho sum|G_i|^2 + eps, A = F(g_rhs) = F(g_x) / D, all transforms unnormalised:
//     dL/dO = (1 / HW) sum_b [ conj(A_b) Y_b - 2 Re(A_b conj(X_b)) O ]        (dL/dRe O + i dL/dIm O, O shared by the batch)
// A, X, Y: [B][C][HW] complex (full spectra, natural order; X and Y nullable), O: [C][HW], G: [C][HW].  With X = NULL it is the
// gradient of a plain product: y = F^-1(O F(x)) has dL/dO = (1/HW) sum_b conj(F(x)) F(g) (A = F(x), Y = F(g)), the adjoint
// y = F^-1(conj(O) F(x)) has dL/dO = (1/HW) sum_b conj(F(g)) F(x) (A = F(g), Y = F(x)).
__global__ void k_otf_grad(const float2* __restrict__ A, const float2* __restrict__ X, const float2* __restrict__ Y,
                           const float2* __restrict__ O, float2* __restrict__ G, int B, long chw, float scale, int accumulate) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < chw; i += (long)gridDim.x * blockDim.x) {
    const float2 o = O[i];
    float2 acc = make_float2(0.f, 0.f);
    for (int b = 0; b < B; ++b) {
      const float2 a = A[(long)b * chw + i];
      if (X) {
        const float2 x = X[(long)b * chw + i];
        const float w = 2.f * (a.x * x.x + a.y * x.y);               // 2 Re(A conj X)
        acc.x -= w * o.x;
        acc.y -= w * o.y;
      }
      if (Y) {
        const float2 y = Y[(long)b * chw + i];
        acc.x += a.x * y.x + a.y * y.y;                               // conj(A) Y
        acc.y += a.x * y.y - a.y * y.x;
      }
    }
    acc = make_float2(acc.x * scale, acc.y * scale);
    if (accumulate) acc = make_float2(acc.x + G[i].x, acc.y + G[i].y);
    G[i] = acc;
  }
}

// ---- the rhs stage of iteration `it` and the z stage of iteration `it - 1` in ONE pass (the unrolled backward loop runs them back to
// back; W % 4 == 0).  Between them the staged form writes g_v_i, g_u_i (2 n planes) and reads them again; here a thread forms
//     g_v_i = rho K_i g,   g_u_i = a_i - g_v_i,   g_d_i = J_i (g_v_i - g_u_i) + g_u_i      (a_i: the previous z stage's share, zb_gd)
// for its four pixels and -- for the stencil adjoints -- for the pixel to their left / the four pixels above (recomputed from g, a_i,
// v_i there: reads that mostly hit the cache), and leaves  g_x = sum_i K_i^T g_d_i,  a_i' = g_d_i  and the three reductions' partial
// sums (d/d rho of iteration `it`: parts a and b as k_solve_rhs_bwd4; d/d lam_i of iteration `it - 1` as k_zupdate_bwd4).
// 10 plane passes instead of 16, one launch instead of two.  a_in and a_out must be different buffers (neighbours read a_in).
struct FusedBwdTerm {
  int linop, prox;
  float alpha;
  const float* lam;       // [B], iteration it - 1
  const float* v;         // saved prox output of iteration it - 1 (fp32 or bf16 history plane)
  const float* a_in;      // the z stage's share of d/du from the previous backward step (nullable = 0)
  float* a_out;           // g_d of this step
};
struct FusedBwdPack {
  FusedBwdTerm t[DPX_MAX_TERMS];
  int n;
  int hist_bf16;
};
__device__ __forceinline__ void fb_gd4(const FusedBwdTerm& tm, const float (&kg)[4], long i, float lam, int hb, float (&gd)[4], float (&lt)[4]) {
  const float4 a4 = tm.a_in ? *(const float4*)(tm.a_in + i) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 v4 = dpx_hist_load4(tm.v, hb, i);
  const float aa[4] = {a4.x, a4.y, a4.z, a4.w}, va[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float gu = aa[k] - kg[k], diff = kg[k] - gu, v = va[k];
    float J, dl;
    if (tm.prox == DPX_PROX_NORM1) {
      J = v != 0.f ? 1.f : 0.f;
      dl = v > 0.f ? -1.f : (v < 0.f ? 1.f : 0.f);
    } else if (tm.prox == DPX_PROX_NONNEG) {
      J = v > 0.f ? 1.f : 0.f;
      dl = 0.f;
    } else {
      const float s = 1.f / (1.f + 2.f * lam);
      J = s;
      dl = -2.f * v * s;
    }
    lt[k] = diff * dl;
    gd[k] = fmaf(J, diff, gu);
  }
}
__device__ __forceinline__ float fb_gd1(const FusedBwdTerm& tm, float kg, long i, float lam, int hb) {
  const float a = tm.a_in ? tm.a_in[i] : 0.f, v = dpx_hist_load(tm.v, hb, i);
  const float gu = a - kg, diff = kg - gu;
  float J;
  if (tm.prox == DPX_PROX_NORM1) J = v != 0.f ? 1.f : 0.f;
  else if (tm.prox == DPX_PROX_NONNEG) J = v > 0.f ? 1.f : 0.f;
  else J = 1.f / (1.f + 2.f * lam);
  return fmaf(J, diff, gu);
}
// counter != NULL: the LAST workgroup to arrive also finishes the three reductions (what k_ad_finish_iter does in its own launch: the same
// sums in the same order) -- glam[t B + b] (iteration it - 1) and grho[b] = sum part_a + (sum part_b) / rho_b (iteration it).
__global__ void __launch_bounds__(256) k_rhs_z_bwd4(const float* __restrict__ g, const float* __restrict__ x, const float* __restrict__ rhs,
                                                     const float* __restrict__ rho, FusedBwdPack T, float* __restrict__ gx,
                                                     float* __restrict__ part_a, float* __restrict__ part_b, float* __restrict__ part_lam, int C,
                                                     int H, int W, unsigned* __restrict__ counter, float* __restrict__ glam,
                                                     float* __restrict__ grho) {
  __shared__ float sh[16];
  __shared__ int shlast;
  const int b = blockIdx.y, hb = T.hist_bf16;
  const long npb = (long)C * H * W, base = (long)b * npb;
  const float r = rho[b];
  float cI = 0.f;
  int nW = 0, nH = 0;
  for (int t = 0; t < T.n; ++t) {
    if (T.t[t].linop == DPX_LIN_IDENTITY) cI += 1.f;
    else if (T.t[t].linop == DPX_LIN_GRAD_W) ++nW;
    else ++nH;
  }
  float acc_a = 0.f, acc_b = 0.f, lsum[DPX_MAX_TERMS];
#pragma unroll
  for (int t = 0; t < DPX_MAX_TERMS; ++t) lsum[t] = 0.f;
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < npb / 4; q += (long)gridDim.x * 256) {
    const long p = q * 4;
    const int w = (int)(p % W);
    const long row = p / W;
    const int h = (int)(row % H);
    const long i = base + p, rowb = base + row * W;
    const long up = (long)((h == 0 ? H - 1 : h - 1) - h) * W, dn = (long)((h + 1 == H ? 0 : h + 1) - h) * W;
    const long il = rowb + (w == 0 ? W - 1 : w - 1), ir = rowb + (w + 4 == W ? 0 : w + 4);
    const float4 g4 = *(const float4*)(g + i);
    const float ga[4] = {g4.x, g4.y, g4.z, g4.w};
    // ---- the two rho reductions of iteration `it` (k_solve_rhs_bwd4)
    {
      const float4 x4 = dpx_hist_load4(x, hb, i), r4 = dpx_hist_load4(rhs, hb, i);
      const float xa[4] = {x4.x, x4.y, x4.z, x4.w}, ra[4] = {r4.x, r4.y, r4.z, r4.w};
      float lx[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) lx[k] = cI * xa[k];
      if (nW) {
        const float xe[6] = {dpx_hist_load(x, hb, il), xa[0], xa[1], xa[2], xa[3], dpx_hist_load(x, hb, ir)};
#pragma unroll
        for (int k = 0; k < 4; ++k) lx[k] += (float)nW * (2.f * xe[k + 1] - xe[k] - xe[k + 2]);
      }
      if (nH) {
        const float4 xu = dpx_hist_load4(x, hb, i + up), xd = dpx_hist_load4(x, hb, i + dn);
        const float xua[4] = {xu.x, xu.y, xu.z, xu.w}, xda[4] = {xd.x, xd.y, xd.z, xd.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) lx[k] += (float)nH * (2.f * xa[k] - xua[k] - xda[k]);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        acc_a = fmaf(ga[k], lx[k], acc_a);
        acc_b = fmaf(ga[k], ra[k], acc_b);
      }
    }
    // ---- g_v, g_u, g_d of every term at the own pixels and at the stencil neighbours; g_x = sum K^T g_d
    const float gright = nW ? g[ir] : 0.f, gleft = nW ? g[il] : 0.f;
    float gdn[4] = {0.f, 0.f, 0.f, 0.f}, gup[4] = {0.f, 0.f, 0.f, 0.f};
    if (nH) {
      const float4 d4 = *(const float4*)(g + i + dn), u4 = *(const float4*)(g + i + up);
      gdn[0] = d4.x; gdn[1] = d4.y; gdn[2] = d4.z; gdn[3] = d4.w;
      gup[0] = u4.x; gup[1] = u4.y; gup[2] = u4.z; gup[3] = u4.w;
    }
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < DPX_MAX_TERMS; ++t) {
      if (t < T.n) {
        const FusedBwdTerm& tm = T.t[t];
        const float lam = tm.lam ? tm.lam[b] * tm.alpha : 0.f;
        float kg[4], gd[4], lt[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (tm.linop == DPX_LIN_IDENTITY) kg[k] = ga[k];
          else if (tm.linop == DPX_LIN_GRAD_W) kg[k] = (k < 3 ? ga[(k + 1) & 3] : gright) - ga[k];
          else kg[k] = gdn[k] - ga[k];
          kg[k] *= r;
        }
        fb_gd4(tm, kg, i, lam, hb, gd, lt);
        lsum[t] += (lt[0] + lt[1]) + (lt[2] + lt[3]);
        *(float4*)(tm.a_out + i) = make_float4(gd[0], gd[1], gd[2], gd[3]);
        if (tm.linop == DPX_LIN_IDENTITY) {
#pragma unroll
          for (int k = 0; k < 4; ++k) acc[k] += gd[k];
        } else if (tm.linop == DPX_LIN_GRAD_W) {           // adjoint: y[w-1] - y[w]; the left pixel's g_v = rho (g[w] - g[w-1])
          const float left = fb_gd1(tm, r * (ga[0] - gleft), il, lam, hb);
          acc[0] += left - gd[0];
#pragma unroll
          for (int k = 1; k < 4; ++k) acc[k] += gd[k - 1] - gd[k];
        } else {                                            // grad_H adjoint: y[h-1] - y[h]; the row above has g_v = rho (g[h] - g[h-1])
          float kgu[4], gdu[4], ltu[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) kgu[k] = r * (ga[k] - gup[k]);
          fb_gd4(tm, kgu, i + up, lam, hb, gdu, ltu);
#pragma unroll
          for (int k = 0; k < 4; ++k) acc[k] += gdu[k] - gd[k];
        }
      }
    }
    *(float4*)(gx + i) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  }
  const float sa = ad_block_sum(acc_a, sh);
  __syncthreads();
  const float sb = ad_block_sum(acc_b, sh);
  if (threadIdx.x == 0) {
    dpx_st_agent(part_a + (long)b * gridDim.x + blockIdx.x, -sa);
    dpx_st_agent(part_b + (long)b * gridDim.x + blockIdx.x, sb);
  }
  for (int t = 0; t < T.n; ++t) {
    __syncthreads();
    const float s = ad_block_sum(lsum[t], sh);
    if (threadIdx.x == 0) dpx_st_agent(part_lam + ((long)t * gridDim.y + b) * gridDim.x + blockIdx.x, s * T.t[t].alpha);
  }
  if (!counter) return;
  if (!dpx_last_block(counter, gridDim.x * gridDim.y, &shlast)) return;
  const int nblk = gridDim.x, B = gridDim.y, nB = T.n * B;
  for (int j = 0; j < nB + B; ++j) {
    __syncthreads();
    if (j < nB) {
      float acc = 0.f;
      for (int i = threadIdx.x; i < nblk; i += blockDim.x) acc += dpx_ld_agent(part_lam + (long)j * nblk + i);
      acc = ad_block_sum(acc, sh);
      if (threadIdx.x == 0) glam[j] = acc;
    } else {
      const int bb = j - nB;
      float a = 0.f, c = 0.f;
      for (int i = threadIdx.x; i < nblk; i += blockDim.x) a += dpx_ld_agent(part_a + (long)bb * nblk + i);
      a = ad_block_sum(a, sh);
      __syncthreads();
      for (int i = threadIdx.x; i < nblk; i += blockDim.x) c += dpx_ld_agent(part_b + (long)bb * nblk + i);
      c = ad_block_sum(c, sh);
      if (threadIdx.x == 0) grho[bb] = a + c / rho[bb];
    }
  }
}

static int ad_blocks(long npb) {
  long g = (npb + 256 * 8 - 1) / (256 * 8);
  return (int)(g > 512 ? 512 : (g < 1 ? 1 : g));
}

}  // namespace dpx

using namespace dpx;

extern "C" size_t dpx_admm_bwd_ws_bytes(int B, int C, int H, int W) {
  return (size_t)DPX_MAX_TERMS * B * ad_blocks((long)C * H * W) * sizeof(float);
}

namespace dpx {
int ad_partial_blocks(int C, int H, int W) { return ad_blocks((long)C * H * W); }
// the stage kernels of the unrolled backward pass without their finishing launches: partial sums [rows][nblk] into `part`
int zupdate_bwd_partials(float* gx, const dpx_bwd_term* terms, int nterms, float* part, int hist_bf16, int B, int C, int H, int W, hipStream_t s) {
  BwdPack T;
  T.n = nterms;
  T.hist_bf16 = hist_bf16;
  for (int i = 0; i < nterms; ++i) {
    DPX_REQUIRE(terms[i].v && terms[i].gu, "dpx_admm_zupdate_bwd: term %d lacks v / gu", i);
    T.t[i] = BwdTerm{terms[i].linop, terms[i].prox, terms[i].alpha, terms[i].lam, terms[i].v, terms[i].gv, terms[i].gu_new, terms[i].gu};
  }
  if (W % 4 == 0) DPX_LAUNCH("k_zupdate_bwd", k_zupdate_bwd4, dim3(ad_blocks((long)C * H * W), B), dim3(256), 0, s, gx, T, part, C, H, W);
  else DPX_LAUNCH("k_zupdate_bwd", k_zupdate_bwd, dim3(ad_blocks((long)C * H * W), B), dim3(256), 0, s, gx, T, part, C, H, W);
  return launch_status("dpx_admm_zupdate_bwd");
}
int solve_rhs_bwd_partials(const float* g, const float* x, const float* rhs, const float* rho, const int* linops, int nterms, float* const* gv,
                           float* const* gu, const float* const* gu_add, float* part_a, float* part_b, int hist_bf16, int B, int C, int H, int W,
                           hipStream_t s) {
  RhsBwdPack T;
  T.n = nterms;
  T.hist_bf16 = hist_bf16;
  for (int i = 0; i < nterms; ++i) {
    T.linop[i] = linops[i];
    T.gv[i] = gv[i];
    T.gu[i] = gu[i];
    T.gu_add[i] = gu_add ? gu_add[i] : nullptr;
  }
  if (W % 4 == 0)
    DPX_LAUNCH("k_solve_rhs_bwd", k_solve_rhs_bwd4, dim3(ad_blocks((long)C * H * W), B), dim3(256), 0, s, g, x, rhs, rho, T, part_a, part_b, C, H, W);
  else
    DPX_LAUNCH("k_solve_rhs_bwd", k_solve_rhs_bwd, dim3(ad_blocks((long)C * H * W), B), dim3(256), 0, s, g, x, rhs, rho, T, part_a, part_b, C, H, W);
  return launch_status("dpx_admm_unrolled_backward");
}
// glam == NULL: only the rho reductions; grho == NULL: only the lambda reductions; nblk: partial sums per row
int finish_iter_n(const float* part_lam, const float* part_a, const float* part_b, float* glam, float* grho, const float* rho, int nterms, int B,
                  int nblk, hipStream_t s) {
  const int nB = glam ? nterms * B : 0, nR = grho ? B : 0;
  if (nB + nR == 0) return DPX_OK;
  DPX_LAUNCH("k_ad_finish_iter", k_ad_finish_iter, dim3(nB + nR), dim3(256), 0, s, part_lam, part_a, part_b, glam, grho, rho, nB, nblk);
  return launch_status("dpx_admm_unrolled_backward");
}
int finish_all(const float* part, long stride, float* glam, float* grho, const float* rho_tab, int nterms, int B, int nblk, int T, int nst,
               hipStream_t s) {
  DPX_LAUNCH("k_ad_finish_all", k_ad_finish_all, dim3(nterms * B + B, nst), dim3(256), 0, s, part, stride, glam, grho, rho_tab, nterms * B, B, nblk, T);
  return launch_status("dpx_admm_unrolled_backward");
}
int finish_iter(const float* part_lam, const float* part_a, const float* part_b, float* glam, float* grho, const float* rho, int nterms, int B,
                int C, int H, int W, hipStream_t s) {
  return finish_iter_n(part_lam, part_a, part_b, glam, grho, rho, nterms, B, ad_blocks((long)C * H * W), s);
}
// rhs stage of iteration `it` + z stage of iteration `it - 1` (k_rhs_z_bwd4); false: the planes do not fit it (W % 4)
bool rhs_z_bwd_fused(const float* g, const float* x, const float* rhs, const float* rho, const dpx_bwd_term* terms, int nterms, const float* const* a_in,
                     float* const* a_out, float* gx, float* part_a, float* part_b, float* part_lam, int hist_bf16, int B, int C, int H, int W,
                     hipStream_t s, unsigned* counter, float* glam, float* grho) {
  if (W % 4) return false;
  FusedBwdPack T;
  T.n = nterms;
  T.hist_bf16 = hist_bf16;
  for (int i = 0; i < nterms; ++i) T.t[i] = FusedBwdTerm{terms[i].linop, terms[i].prox, terms[i].alpha, terms[i].lam, terms[i].v, a_in[i], a_out[i]};
  DPX_LAUNCH("k_rhs_z_bwd", k_rhs_z_bwd4, dim3(ad_blocks((long)C * H * W), B), dim3(256), 0, s, g, x, rhs, rho, T, gx, part_a, part_b, part_lam, C, H, W,
             counter, glam, grho);
  return true;
}
}  // namespace dpx

extern "C" int dpx_admm_zupdate_bwd(float* gx, const dpx_bwd_term* terms, int nterms, float* glam, int B, int C, int H, int W, void* ws,
                                    dpx_stream_t stream) {
  DPX_REQUIRE(gx && terms && glam && ws && nterms >= 1 && nterms <= DPX_MAX_TERMS && B > 0 && C > 0 && H > 0 && W > 0,
              "dpx_admm_zupdate_bwd: bad arguments");
  BwdPack T;
  T.n = nterms;
  T.hist_bf16 = 0;
  for (int i = 0; i < nterms; ++i) {
    DPX_REQUIRE(terms[i].v && terms[i].gu, "dpx_admm_zupdate_bwd: term %d lacks v / gu", i);
    DPX_REQUIRE(terms[i].linop >= DPX_LIN_IDENTITY && terms[i].linop <= DPX_LIN_GRAD_W && terms[i].prox >= DPX_PROX_NORM1 &&
                    terms[i].prox <= DPX_PROX_SUMSQ, "dpx_admm_zupdate_bwd: term %d has an unknown linop / prox code", i);
    T.t[i] = BwdTerm{terms[i].linop, terms[i].prox, terms[i].alpha, terms[i].lam, terms[i].v, terms[i].gv, terms[i].gu_new, terms[i].gu};
  }
  const int nblk = ad_blocks((long)C * H * W);
  hipStream_t s = (hipStream_t)stream;
  if (W % 4 == 0) DPX_LAUNCH("k_zupdate_bwd", k_zupdate_bwd4, dim3(nblk, B), dim3(256), 0, s, gx, T, (float*)ws, C, H, W);
  else DPX_LAUNCH("k_zupdate_bwd", k_zupdate_bwd, dim3(nblk, B), dim3(256), 0, s, gx, T, (float*)ws, C, H, W);
  DPX_LAUNCH("k_ad_finish", k_ad_finish, dim3(nterms * B), dim3(256), 0, s, (const float*)ws, glam, nblk, (const float*)nullptr, (const float*)nullptr);
  return launch_status("dpx_admm_zupdate_bwd");
}

extern "C" int dpx_admm_solve_rho_grad(const float* g_rhs, const float* x, const int* linops, int nterms, float* grho, int B, int C, int H,
                                       int W, void* ws, dpx_stream_t stream) {
  DPX_REQUIRE(g_rhs && x && grho && ws && (linops || nterms == 0) && nterms >= 0 && nterms <= DPX_MAX_TERMS,
              "dpx_admm_solve_rho_grad: bad arguments");
  LinCodes L;
  L.n = nterms;
  L.hist_bf16 = 0;
  for (int i = 0; i < nterms; ++i) L.linop[i] = linops[i];
  const int nblk = ad_blocks((long)C * H * W);
  hipStream_t s = (hipStream_t)stream;
  DPX_LAUNCH("k_solve_rho_grad", k_solve_rho_grad, dim3(nblk, B), dim3(256), 0, s, g_rhs, x, L, (float*)ws, C, H, W);
  DPX_LAUNCH("k_ad_finish", k_ad_finish, dim3(B), dim3(256), 0, s, (const float*)ws, grho, nblk, (const float*)nullptr, (const float*)nullptr);
  return launch_status("dpx_admm_solve_rho_grad");
}

namespace dpx {
// dpx_admm_rhs_bwd with the two sums that follow it in the unrolled backward pass folded in: gu[i] = gu_add[i] - gv[i] and
// grho = <g, rhs> / rho + grho_add (both additions nullable)
int rhs_bwd_impl(const float* g, const float* rhs, const float* rho, const int* linops, int nterms, float* const* gv, float* const* gu,
                 const float* const* gu_add, float* grho, const float* grho_add, int hist_bf16, int B, int C, int H, int W, void* ws, hipStream_t s) {
  RhsBwdPack T;
  T.n = nterms;
  T.hist_bf16 = hist_bf16;
  for (int i = 0; i < nterms; ++i) {
    T.linop[i] = linops[i];
    T.gv[i] = gv[i];
    T.gu[i] = gu[i];
    T.gu_add[i] = gu_add ? gu_add[i] : nullptr;
  }
  const int nblk = ad_blocks((long)C * H * W);
  DPX_LAUNCH("k_rhs_bwd", k_rhs_bwd, dim3(nblk, B), dim3(256), 0, s, g, rhs, rho, T, (float*)ws, C, H, W);
  if (grho)                                             // (NULL: the caller finishes the partial sums in `ws` itself)
    DPX_LAUNCH("k_ad_finish", k_ad_finish, dim3(B), dim3(256), 0, s, (const float*)ws, grho, nblk, rho, grho_add);
  return launch_status("dpx_admm_rhs_bwd");
}
}  // namespace dpx

extern "C" int dpx_admm_rhs_bwd(const float* g, const float* rhs, const float* rho, const int* linops, int nterms, float* const* gv,
                                float* const* gu, float* grho, int B, int C, int H, int W, void* ws, dpx_stream_t stream) {
  DPX_REQUIRE(g && rhs && rho && linops && gv && gu && grho && ws && nterms >= 1 && nterms <= DPX_MAX_TERMS,
              "dpx_admm_rhs_bwd: bad arguments");
  return rhs_bwd_impl(g, rhs, rho, linops, nterms, gv, gu, nullptr, grho, nullptr, 0, B, C, H, W, ws, (hipStream_t)stream);
}

extern "C" int dpx_otf_grad(const void* A, const void* X, const void* Y, const void* O, void* G, int B, int C, int H, int W, int accumulate,
                            dpx_stream_t stream) {
  DPX_REQUIRE(A && (X || Y) && O && G && B > 0 && C > 0 && H > 0 && W > 0, "dpx_otf_grad: bad arguments");
  const long chw = (long)C * H * W;
  DPX_LAUNCH("k_otf_grad", k_otf_grad, dim3(grid_for(chw, 256, 4096)), dim3(256), 0, (hipStream_t)stream, (const float2*)A, (const float2*)X,
             (const float2*)Y, (const float2*)O, (float2*)G, B, chw, 1.0f / ((float)H * (float)W), accumulate);
  return launch_status("dpx_otf_grad");
}
