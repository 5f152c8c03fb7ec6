// Streaming (HBM-bound) kernels of the ADMM/PGD iteration: stencils, proximal operators, the fused
// z/dual update, the x-update right-hand side, AXPY-style linear combinations and batched dots.
//
// They replace the reference's chains of eager elementwise torch ops:
//   * v_i = prox(K_i x + u_i), u_i += K_i x - v_i           dprox/algo/admm.py:54-57
//   * Ktb + rho * sum_i K_i^T (v_i - u_i)                    dprox/proxfn/sum_square.py:126-135, algo/admm.py:51
//   * soft-threshold / nonneg / v/(1+2 lam)                 dprox/proxfn/norm.py:6-27, nonneg.py:10-11, sum_square.py:26-27
//   * grad forward/adjoint (a 2-tap circular stencil the reference evaluates with two full FFTs)
//                                                           dprox/linop/grad.py:8-23, linop/conv.py:31-41
//   * bdot and the CG AXPYs                                  dprox/linalg/solve/solver_cg.py:7-22,109-129
// All of them are one coalesced float4 pass over each operand; halo values of the stencils come
// from L2 (the neighbouring row/pixel was just streamed by the same or the adjacent workgroup).
#include <cstdlib>

#include "dpx_cg_dev.h"
#include "dpx_prox_dev.h"

namespace dpx {


// ---------------------------------------------------------------------------------------------
// z / dual update.  One thread = 4 consecutive pixels of a row (VEC=4) or one pixel (VEC=1).
// ---------------------------------------------------------------------------------------------
template <int VEC>
__global__ void k_zupdate(const float* __restrict__ x, TermPack T, int B, int C, int H, int W) {
  const int Wv = W / VEC;
  const long total = (long)B * C * H * Wv;
  for (long i = (long)xcd_block() * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int wv = (int)(i % Wv);
    const long row = i / Wv;                 // (b*C + c)*H + h
    const int h = (int)(row % H);
    const long plane = row / H;
    const int b = (int)(plane / C);
    const long off = row * W + (long)wv * VEC;
    float xv[VEC + 1];
    if constexpr (VEC == 4) {
      const float4 q = *(const float4*)(x + off);
      xv[0] = q.x; xv[1] = q.y; xv[2] = q.z; xv[3] = q.w;
    } else {
      xv[0] = x[off];
    }
    bool need_w = false, need_h = false;
    for (int t = 0; t < T.n; ++t) {
      need_w |= T.t[t].linop == DPX_LIN_GRAD_W;
      need_h |= T.t[t].linop == DPX_LIN_GRAD_H;
    }
    if (need_w) xv[VEC] = x[row * W + ((wv + 1) * VEC) % W];
    float xd[VEC];
    if (need_h) {
      const long offd = (plane * H + (h + 1 == H ? 0 : h + 1)) * W + (long)wv * VEC;
      if constexpr (VEC == 4) {
        const float4 q = *(const float4*)(x + offd);
        xd[0] = q.x; xd[1] = q.y; xd[2] = q.z; xd[3] = q.w;
      } else {
        xd[0] = x[offd];
      }
    }
    for (int t = 0; t < T.n; ++t) {
      const dpx_term tm = T.t[t];
      const float lam = tm.lam ? tm.lam[b] * tm.alpha : 0.f;
      float uu[VEC], vv[VEC];
      if (tm.prox != DPX_PROX_EXTERNAL || true) {
        if constexpr (VEC == 4) {
          const float4 q = *(const float4*)(tm.u + off);
          uu[0] = q.x; uu[1] = q.y; uu[2] = q.z; uu[3] = q.w;
        } else {
          uu[0] = tm.u[off];
        }
      }
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        float kx;
        if (tm.linop == DPX_LIN_IDENTITY) kx = xv[e];
        else if (tm.linop == DPX_LIN_GRAD_W) kx = xv[e + 1] - xv[e];
        else kx = xd[e] - xv[e];
        const float d = kx + uu[e];
        if (tm.prox == DPX_PROX_EXTERNAL) {
          vv[e] = d;                                         // the denoiser consumes d; u finished afterwards
        } else {
          vv[e] = prox_eval(tm.prox, d, lam);
          uu[e] = d - vv[e];
        }
      }
      float* uo = tm.u_out ? tm.u_out : tm.u;            // out-of-place dual (differentiable mode keeps u_in)
      if constexpr (VEC == 4) {
        *(float4*)(tm.v + off) = make_float4(vv[0], vv[1], vv[2], vv[3]);
        if (tm.prox != DPX_PROX_EXTERNAL) *(float4*)(uo + off) = make_float4(uu[0], uu[1], uu[2], uu[3]);
      } else {
        tm.v[off] = vv[0];
        if (tm.prox != DPX_PROX_EXTERNAL) uo[off] = uu[0];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// rhs = ktb + sum_i rho_b * K_i^T (v_i - u_i)
// ---------------------------------------------------------------------------------------------
template <int VEC>
__global__ void k_rhs(float* __restrict__ rhs, const float* __restrict__ ktb, const float* __restrict__ rho, TermPack T,
                      int B, int C, int H, int W) {
  const int Wv = W / VEC;
  const long total = (long)B * C * H * Wv;
  for (long i = (long)xcd_block() * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int wv = (int)(i % Wv);
    const long row = i / Wv;
    const int h = (int)(row % H);
    const long plane = row / H;
    const int b = (int)(plane / C);
    const long off = row * W + (long)wv * VEC;
    const float r = rho[b];
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
    for (int t = 0; t < T.n; ++t) {
      const dpx_term tm = T.t[t];
      float y[VEC + 1];   // y[1..VEC] = (v-u) at this pixel group, y[0] = left neighbour
      if constexpr (VEC == 4) {
        const float4 a = *(const float4*)(tm.v + off), c = *(const float4*)(tm.u + off);
        y[1] = a.x - c.x; y[2] = a.y - c.y; y[3] = a.z - c.z; y[4] = a.w - c.w;
      } else {
        y[1] = tm.v[off] - tm.u[off];
      }
      if (tm.linop == DPX_LIN_IDENTITY) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] += y[e + 1];
      } else if (tm.linop == DPX_LIN_GRAD_W) {
        const long left = row * W + (wv == 0 ? W - 1 : wv * VEC - 1);
        y[0] = tm.v[left] - tm.u[left];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] += y[e] - y[e + 1];
      } else {
        const long offu = (plane * H + (h == 0 ? H - 1 : h - 1)) * W + (long)wv * VEC;
        float yu[VEC];
        if constexpr (VEC == 4) {
          const float4 a = *(const float4*)(tm.v + offu), c = *(const float4*)(tm.u + offu);
          yu[0] = a.x - c.x; yu[1] = a.y - c.y; yu[2] = a.z - c.z; yu[3] = a.w - c.w;
        } else {
          yu[0] = tm.v[offu] - tm.u[offu];
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] += yu[e] - y[e + 1];
      }
    }
    if constexpr (VEC == 4) {
      float4 k = ktb ? *(const float4*)(ktb + off) : make_float4(0.f, 0.f, 0.f, 0.f);
      *(float4*)(rhs + off) = make_float4(fmaf(r, acc[0], k.x), fmaf(r, acc[1], k.y), fmaf(r, acc[2], k.z), fmaf(r, acc[3], k.w));
    } else {
      rhs[off] = fmaf(r, acc[0], ktb ? ktb[off] : 0.f);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// z / dual update of iteration t AND the right-hand side of iteration t + 1 in one pass (planes off the two-kernel iteration):
//   d = K_i x + u_i ; v_i = prox_i(d) ; u_i' = d - v_i ;  rhs = ktb + rho' sum_i K_i^T (v_i - u_i')
// k_zupdate + k_rhs move 12 planes (x, u_i in; v_i, u_i' out; v_i, u_i' in again; rhs out: two gradient terms), this pass 8 (6 without
// the v stores, 4 with neither v nor dual stores: half-quadratic splitting): the
// adjoint stencils need v - u' at the left / upper neighbour, which the thread RECOMPUTES from x and u there (cache hits: the
// neighbour's own thread reads the same lines) instead of waiting for another thread's stores.  The incoming duals are therefore read
// at neighbours while the outgoing ones are written: u must be double-buffered (terms[i].u_out != terms[i].u).  Same expressions in the
// same order as the two kernels: bit-identical v, u', rhs.
// ---------------------------------------------------------------------------------------------
template <int VEC>
__global__ void k_zupdate_rhs(const float* __restrict__ x, float* __restrict__ rhs, const float* __restrict__ ktb,
                              const float* __restrict__ rho, TermPack T, int dual, int emit_v, int B, int C, int H, int W) {
  const int Wv = W / VEC;
  const long total = (long)B * C * H * Wv;
  bool need_w = false, need_h = false;
  for (int t = 0; t < T.n; ++t) {
    need_w |= T.t[t].linop == DPX_LIN_GRAD_W;
    need_h |= T.t[t].linop == DPX_LIN_GRAD_H;
  }
  for (long i = (long)xcd_block() * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int wv = (int)(i % Wv);
    const long row = i / Wv;                 // (b*C + c)*H + h
    const int h = (int)(row % H);
    const long plane = row / H;
    const int b = (int)(plane / C);
    const long off = row * W + (long)wv * VEC;
    const long left = row * W + (wv == 0 ? W - 1 : wv * VEC - 1);                     // left neighbour of the group's first pixel
    const long offu = (plane * H + (h == 0 ? H - 1 : h - 1)) * W + (long)wv * VEC;      // the group one row up
    const float r = rho[b];
    float xv[VEC + 1], xd[VEC], xu[VEC], xl = 0.f;
    if constexpr (VEC == 4) {
      const float4 q = *(const float4*)(x + off);
      xv[0] = q.x; xv[1] = q.y; xv[2] = q.z; xv[3] = q.w;
    } else {
      xv[0] = x[off];
    }
    if (need_w) {
      xv[VEC] = x[row * W + ((wv + 1) * VEC) % W];
      xl = x[left];
    }
    if (need_h) {
      const long offd = (plane * H + (h + 1 == H ? 0 : h + 1)) * W + (long)wv * VEC;
      if constexpr (VEC == 4) {
        const float4 q = *(const float4*)(x + offd), p = *(const float4*)(x + offu);
        xd[0] = q.x; xd[1] = q.y; xd[2] = q.z; xd[3] = q.w;
        xu[0] = p.x; xu[1] = p.y; xu[2] = p.z; xu[3] = p.w;
      } else {
        xd[0] = x[offd];
        xu[0] = x[offu];
      }
    }
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
    for (int t = 0; t < T.n; ++t) {
      const dpx_term tm = T.t[t];
      const float lam = tm.lam ? tm.lam[b] * tm.alpha : 0.f;
      const bool nodual = tm.reserved & DPX_TERM_NO_DUAL;      // half-quadratic splitting: the incoming duals are zero and are not fetched
      float uu[VEC], vv[VEC];
      if (nodual) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) uu[e] = 0.f;
      } else if constexpr (VEC == 4) {
        const float4 q = *(const float4*)(tm.u + off);
        uu[0] = q.x; uu[1] = q.y; uu[2] = q.z; uu[3] = q.w;
      } else {
        uu[0] = tm.u[off];
      }
      float y[VEC + 1];   // y[1..VEC] = v - u' at this pixel group, y[0] = left neighbour (as in k_rhs)
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        float kx;
        if (tm.linop == DPX_LIN_IDENTITY) kx = xv[e];
        else if (tm.linop == DPX_LIN_GRAD_W) kx = xv[e + 1] - xv[e];
        else kx = xd[e] - xv[e];
        const float d = kx + uu[e];
        vv[e] = prox_eval(tm.prox, d, lam);
        const float uin = uu[e];
        uu[e] = d - vv[e];
        y[e + 1] = vv[e] - (dual ? uu[e] : uin);            // (dual = 0: the duals are not advanced -- the next right-hand side sees the incoming ones)
      }
      // (v is read by nobody inside the loop -- the next right-hand side is formed here -- and duals that are not advanced go nowhere:
      //  emit_v = 0 / dual = 0 skip those stores; the caller's last z / dual stage, dpx_admm_zupdate, writes the final state)
      if constexpr (VEC == 4) {
        if (emit_v) *(float4*)(tm.v + off) = make_float4(vv[0], vv[1], vv[2], vv[3]);
        if (dual) *(float4*)(tm.u_out + off) = make_float4(uu[0], uu[1], uu[2], uu[3]);
      } else {
        if (emit_v) tm.v[off] = vv[0];
        if (dual) tm.u_out[off] = uu[0];
      }
      if (tm.linop == DPX_LIN_IDENTITY) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] += y[e + 1];
      } else if (tm.linop == DPX_LIN_GRAD_W) {
        const float uin = nodual ? 0.f : tm.u[left];
        const float d = (xv[0] - xl) + uin;                              // the left neighbour's own update, recomputed
        const float vl = prox_eval(tm.prox, d, lam);
        const float ul = d - vl;
        y[0] = vl - (dual ? ul : uin);
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] += y[e] - y[e + 1];
      } else {
        float ua[VEC], yu[VEC];
        if (nodual) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) ua[e] = 0.f;
        } else if constexpr (VEC == 4) {
          const float4 q = *(const float4*)(tm.u + offu);
          ua[0] = q.x; ua[1] = q.y; ua[2] = q.z; ua[3] = q.w;
        } else {
          ua[0] = tm.u[offu];
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const float d = (xv[e] - xu[e]) + ua[e];                        // the upper neighbour's own update, recomputed
          const float vu = prox_eval(tm.prox, d, lam);
          const float un = d - vu;
          yu[e] = vu - (dual ? un : ua[e]);
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] += yu[e] - y[e + 1];
      }
    }
    if constexpr (VEC == 4) {
      float4 k = ktb ? *(const float4*)(ktb + off) : make_float4(0.f, 0.f, 0.f, 0.f);
      *(float4*)(rhs + off) = make_float4(fmaf(r, acc[0], k.x), fmaf(r, acc[1], k.y), fmaf(r, acc[2], k.z), fmaf(r, acc[3], k.w));
    } else {
      rhs[off] = fmaf(r, acc[0], ktb ? ktb[off] : 0.f);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// stand-alone stencil, prox, linear combination
// ---------------------------------------------------------------------------------------------
__global__ void k_grad(const float* __restrict__ x, float* __restrict__ y, int dim, int adjoint, long planes, int H, int W) {
  const long total = planes * H * W;
  for (long i = (long)xcd_block() * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int w = (int)(i % W);
    const long row = i / W;
    const int h = (int)(row % H);
    long j;
    if (dim == 1) {
      const int wn = adjoint ? (w == 0 ? W - 1 : w - 1) : (w + 1 == W ? 0 : w + 1);
      j = row * W + wn;
    } else {
      const int hn = adjoint ? (h == 0 ? H - 1 : h - 1) : (h + 1 == H ? 0 : h + 1);
      j = (row - h + hn) * W + w;
    }
    y[i] = x[j] - x[i];
  }
}

__global__ void k_prox(int kind, const float* __restrict__ v, float* __restrict__ out, const float* __restrict__ lam,
                       float alpha, const float* __restrict__ off, int B, long npb) {
  const long total = (long)B * npb;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / npb);
    const float o = off ? off[i] : 0.f;
    out[i] = prox_eval(kind, v[i] - o, (lam ? lam[b] : 0.f) * alpha) + o;
  }
}

struct LinPack {
  const float* x[4];
  const float* cb[4];
  float c[4];
  int n;
};
__global__ void k_lincomb(float* __restrict__ out, LinPack L, int B, long npb) {
  const long total = (long)B * npb;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / npb);
    float acc = 0.f;
    for (int t = 0; t < L.n; ++t) {
      const float c = L.c[t] * (L.cb[t] ? L.cb[t][b] : 1.f);
      acc = (t == 0) ? c * L.x[t][i] : fmaf(c, L.x[t][i], acc);
    }
    out[i] = acc;
  }
}

// ---------------------------------------------------------------------------------------------
// batched dot products: per-block partials (wave shuffles + one LDS hop), then a tiny finishing pass.
// Deterministic (no atomics): the CG iterates must be reproducible run to run.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) sh[wid] = v;
  __syncthreads();
  const int nw = (blockDim.x + 63) >> 6;
  float r = (threadIdx.x < nw) ? sh[threadIdx.x] : 0.f;
  if (wid == 0) r = wave_sum(r);
  return r;   // valid in wave 0
}

// grid (nblk, B, Bj): partial[(bi*Bj + bj)*nblk + blk] = sum over a slice of <x[bi], y[bj or bi]>
__global__ void k_dot_partial(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ partial,
                              long npb, int gram) {
  __shared__ float sh[16];
  const int bi = blockIdx.y, bj = gram ? blockIdx.z : bi;
  const float* xa = x + (long)bi * npb;
  const float* yb = y + (long)bj * npb;
  float acc = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npb; i += (long)gridDim.x * blockDim.x)
    acc = fmaf(xa[i], yb[i], acc);
  acc = block_sum(acc, sh);
  if (threadIdx.x == 0) partial[((long)bi * gridDim.z + blockIdx.z) * gridDim.x + blockIdx.x] = acc;
}
__global__ void k_dot_finish(const float* __restrict__ partial, float* __restrict__ out, int nblk) {
  __shared__ float sh[16];
  float acc = 0.f;
  for (int i = threadIdx.x; i < nblk; i += blockDim.x) acc += partial[(long)blockIdx.x * nblk + i];
  acc = block_sum(acc, sh);
  if (threadIdx.x == 0) out[blockIdx.x] = acc;
}

// B x B Gram matrix, B <= 32: a workgroup stages a [32][GR_CH] slab of the residuals in LDS and forms all B*B partial dot
// products from it (k_dot_partial with a (nblk, B, B) grid re-reads every row B times).  64 threads cover the 32 x 32 outputs
// with 4 x 4 register tiles (8 LDS reads per 16 FMAs), the 4 waves split the slab's columns; one slab per workgroup
// iteration and up to 1024 workgroups, so that a 32 x 320^2 residual (800 slabs) fills the chip: 125 us -> ~10 us.
// Summation order is fixed (per-wave partials are added in wave order, slabs by the finishing kernel).
constexpr int GR_CH = 128, GR_P = GR_CH + 1;
// sx: 32 * GR_P floats, red: 3 * 64 * 16 floats of shared memory
// UPDATE: the slab load also applies the pending CG update of the previous iteration, x += alpha_b p, r -= alpha_b A p (upd_*; every
// element of r belongs to exactly one slab of one workgroup), and the Gram products are those of the updated residual.
template <bool AGENT_STORE, bool UPDATE = false>
__device__ __forceinline__ void gram_tile_body(const float* __restrict__ r, float* __restrict__ partial, int B, long npb, int nblk, float* sx,
                                               float* red, float* upd_r = nullptr, float* upd_x = nullptr, const float* upd_p = nullptr,
                                               const float* upd_Ap = nullptr, const float* alpha = nullptr) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, ti = (lane >> 3) * 4, tj = (lane & 7) * 4;
  float acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[a][q] = 0.f;
  for (long c0 = (long)blockIdx.x * GR_CH; c0 < npb; c0 += (long)gridDim.x * GR_CH) {
    __syncthreads();
    for (int e = tid; e < 32 * GR_CH; e += 256) {
      const int b = e / GR_CH, k = e % GR_CH;
      float rv = 0.f;
      if (b < B && c0 + k < npb) {
        const long at = (long)b * npb + c0 + k;
        rv = r[at];
        if constexpr (UPDATE) {
          const float al = alpha[b];
          upd_x[at] = fmaf(al, upd_p[at], upd_x[at]);
          rv = fmaf(-al, upd_Ap[at], rv);
          upd_r[at] = rv;
        }
      }
      sx[b * GR_P + k] = rv;
    }
    __syncthreads();
#pragma unroll 4
    for (int k = wave * (GR_CH / 4); k < (wave + 1) * (GR_CH / 4); ++k) {
      float xi[4], xj[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) { xi[a] = sx[(ti + a) * GR_P + k]; xj[a] = sx[(tj + a) * GR_P + k]; }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[a][q] = fmaf(xi[a], xj[q], acc[a][q]);
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q) red[((wave - 1) * 64 + lane) * 16 + a * 4 + q] = acc[a][q];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float v = ((acc[a][q] + red[(0 * 64 + lane) * 16 + a * 4 + q]) + red[(1 * 64 + lane) * 16 + a * 4 + q]) + red[(2 * 64 + lane) * 16 + a * 4 + q];
        if (ti + a < B && tj + q < B) {
          float* dst = partial + ((long)(ti + a) * B + (tj + q)) * nblk + blockIdx.x;
          if (AGENT_STORE) dpx_st_agent(dst, v);
          else *dst = v;
        }
      }
  }
}
__global__ void __launch_bounds__(256) k_gram_tile(const float* __restrict__ r, float* __restrict__ partial, int B, long npb, int nblk) {
  __shared__ float sx[32 * GR_P];
  __shared__ float red[3 * 64 * 16];
  gram_tile_body<false>(r, partial, B, npb, nblk, sx, red);
}

// The Gram pass of a device-controlled CG iteration with its two followers folded in: the LAST workgroup to arrive adds up the
// slabs' partial products (one wave per entry, fixed order) and runs the stop rule / beta update on the finished matrix
// (cg_test_block) -- k_gram_tile -> k_dot_finish -> k_cg_test in one launch.  B <= 32.
// UPDATE: + the x / r update of the previous iteration in the slab load (k_cg_update folded in: alpha_b = gamma_b / <p_b, A p_b> from the
// state the previous launches left).  host_flags (nullable): a host-mapped copy of (done, n_done) written by the finishing workgroup --
// the host polls it instead of copying the flags back with a transfer per iteration: one 8-byte word, low half = done | n_done << 1, high
// half = host_tag, which tells the host that THIS launch's test has run (dpx_cg_masked_fft spins on it: no event, no marker packet).
template <bool UPDATE>
__global__ void __launch_bounds__(256) k_gram_tile_test(float* __restrict__ r, float* __restrict__ partial, float* __restrict__ G, CgState S,
                                                        long npb, int nblk, unsigned* __restrict__ counter, float init_rtol, float* __restrict__ x,
                                                        const float* __restrict__ p, const float* __restrict__ Ap, int* __restrict__ host_flags, int host_tag) {
  __shared__ double rawd[64 * 65];                        // the slabs' staging (28.8 KB), then the test's matrix (33.3 KB)
  char* raw = (char*)rawd;
  __shared__ int shf[3];
  __shared__ float alpha_s[32];
  if (S.flags()[0]) return;                               // (uniform: the solve has converged, this launch ran ahead)
  const int B = S.B;
  float* sx = (float*)raw;
  float* red = sx + 32 * GR_P;
  if constexpr (UPDATE) {
    if (threadIdx.x < B) alpha_s[threadIdx.x] = S.gamma()[threadIdx.x] / S.pAp()[threadIdx.x];
    __syncthreads();
    gram_tile_body<true, true>(r, partial, B, npb, nblk, sx, red, r, x, p, Ap, alpha_s);
  } else {
    gram_tile_body<true>(r, partial, B, npb, nblk, sx, red);
  }
  if (!dpx_last_block(counter, (unsigned)nblk, &shf[2])) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int e = wave; e < B * B; e += 4) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;          // (four independent chains: the loads of a wave overlap)
    const float* pe = partial + (long)e * nblk;
    int i = lane;
    for (; i + 192 < nblk; i += 256) {
      a0 += dpx_ld_agent(pe + i);
      a1 += dpx_ld_agent(pe + i + 64);
      a2 += dpx_ld_agent(pe + i + 128);
      a3 += dpx_ld_agent(pe + i + 192);
    }
    for (; i < nblk; i += 64) a0 += dpx_ld_agent(pe + i);
    const float acc = wave_sum((a0 + a1) + (a2 + a3));
    if (lane == 0) G[e] = acc;
  }
  __syncthreads();
  cg_test_block(S, G, (double*)raw, shf, init_rtol);
  if (host_flags) {
    __syncthreads();
    if (threadIdx.x == 0) {
      // (done, n_done, tag) in ONE 8-byte store the host reads in one piece: nothing else travels through memory with it (the iterate is
      //  consumed by later launches of this stream), so no system-scope fence -- each one writes the L2's dirty lines back -- is needed
      const unsigned long long word = ((unsigned long long)(unsigned)host_tag << 32) | (unsigned)((S.flags()[1] << 1) | (S.flags()[0] ? 1 : 0));
      __hip_atomic_store((unsigned long long*)host_flags, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// Tuning aid (tools/build_variant.sh par_trace -DDPX_PAR_TRACE; never in the shipped library): thread 0 of every workgroup of the small Gram /
// stop-test launch stamps the 100 MHz real-time counter at its phase boundaries; tools/gram_trace.py reads the stamps of the last launch.
#ifdef DPX_PAR_TRACE
__device__ unsigned long long dpx_gram_trace_buf[256 * 8];
#define DPX_GSTAMP(i)                                                                                             \
  do {                                                                                                            \
    if (threadIdx.x == 0 && blockIdx.x < 256) dpx_gram_trace_buf[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
}  // namespace dpx
extern "C" int dpx_dbg_gram_trace(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(dpx::dpx_gram_trace_buf), (size_t)n * sizeof(unsigned long long));
}
namespace dpx {
#else
#define DPX_GSTAMP(i) ((void)0)
#endif

// The same launch for small batches (B <= 8, n_per_batch a multiple of 4): the slab kernel above pads every batch to 32 x 32 products
// and runs 800 workgroups on a 4 x 320^2 residual (19.6 us per CG iteration of config 4's shard, most of it zero rows and the
// finishing workgroup's 16 x 800 partial sums).  Here a thread holds the B values of four neighbouring elements (float4 per image),
// accumulates the B (B + 1) / 2 distinct products in registers, and ~100 workgroups leave one partial per product: the same
// interface (partial [B * B][nblk], finish + stop rule by the last workgroup to arrive), fixed summation order.
template <int BT, bool UPDATE>
__global__ void __launch_bounds__(256) k_gram_small_test(float* __restrict__ r, float* __restrict__ partial, float* __restrict__ G, CgState S, long npb,
                                                         int nblk, unsigned* __restrict__ counter, float init_rtol, float* __restrict__ x,
                                                         const float* __restrict__ p, const float* __restrict__ Ap, int* __restrict__ host_flags, int host_tag) {
  constexpr int NP = BT * (BT + 1) / 2;
  __shared__ double rawd[64 * 65];                        // the test's matrix (cg_test_block)
  __shared__ float red[4 * NP];
  __shared__ int shf[3];
  __shared__ float alpha_s[BT];
  __shared__ float Gs[BT * BT];
  // (the solve's `done` word and the step lengths are requested together with the first elements: one memory round trip at the head of the
  //  launch instead of three -- flag, then gamma / <p, Ap> behind a barrier, then the data; the flag is looked at before the first store)
  DPX_GSTAMP(0);
  const int dn = S.flags()[0];
  const int B = S.B, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  float alpha_r[BT];
  if constexpr (UPDATE) {
#pragma unroll
    for (int b = 0; b < BT; ++b) alpha_r[b] = b < B ? S.gamma()[b] / S.pAp()[b] : 0.f;
  }
  (void)alpha_s;
  float acc[NP];
#pragma unroll
  for (int e = 0; e < NP; ++e) acc[e] = 0.f;
  const long n4 = npb / 4;
  bool checked = false;
  for (long i = (long)blockIdx.x * 256 + tid; i < n4; i += (long)gridDim.x * 256) {
    float4 rv[BT], pvs[UPDATE ? BT : 1], qvs[UPDATE ? BT : 1], xvs[UPDATE ? BT : 1];
#pragma unroll
    for (int b = 0; b < BT; ++b) {
      rv[b] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (b < B) {
        const long at = (long)b * n4 + i;
        rv[b] = ((const float4*)r)[at];
        if constexpr (UPDATE) {
          pvs[b] = ((const float4*)p)[at];
          qvs[b] = ((const float4*)Ap)[at];
          xvs[b] = ((float4*)x)[at];
        }
      }
    }
    if (!checked) {                                       // (uniform: the solve has converged, this launch ran ahead)
      if (dn) return;
      checked = true;
      DPX_GSTAMP(1);
    }
#pragma unroll
    for (int b = 0; b < BT; ++b) {
      if (b < B) {
        const long at = (long)b * n4 + i;
        if constexpr (UPDATE) {
          const float al = alpha_r[b];
          const float4 pv = pvs[b], qv = qvs[b], xv = xvs[b];
          ((float4*)x)[at] = make_float4(fmaf(al, pv.x, xv.x), fmaf(al, pv.y, xv.y), fmaf(al, pv.z, xv.z), fmaf(al, pv.w, xv.w));
          rv[b] = make_float4(fmaf(-al, qv.x, rv[b].x), fmaf(-al, qv.y, rv[b].y), fmaf(-al, qv.z, rv[b].z), fmaf(-al, qv.w, rv[b].w));
          ((float4*)r)[at] = rv[b];
        }
      }
    }
    int e = 0;
#pragma unroll
    for (int a = 0; a < BT; ++a)
#pragma unroll
      for (int q = a; q < BT; ++q, ++e)
        acc[e] = fmaf(rv[a].w, rv[q].w, fmaf(rv[a].z, rv[q].z, fmaf(rv[a].y, rv[q].y, fmaf(rv[a].x, rv[q].x, acc[e]))));
  }
  if (dn) return;                                         // (threads without an element of their own)
  DPX_GSTAMP(2);
#pragma unroll
  for (int e = 0; e < NP; ++e) {
    const float v = wave_sum(acc[e]);
    if (lane == 0) red[wave * NP + e] = v;
  }
  __syncthreads();
  if (tid < NP) {
    const float v = ((red[tid] + red[NP + tid]) + red[2 * NP + tid]) + red[3 * NP + tid];
    int a = 0, e0 = 0;                                   // pair index tid -> (a, q), a <= q
    while (tid >= e0 + (BT - a)) { e0 += BT - a; ++a; }
    const int q = a + (tid - e0);
    if (a < B && q < B) {
      dpx_st_agent(partial + ((long)a * B + q) * nblk + blockIdx.x, v);
      if (q != a) dpx_st_agent(partial + ((long)q * B + a) * nblk + blockIdx.x, v);
    }
  }
  DPX_GSTAMP(3);
  if (!dpx_last_block(counter, (unsigned)nblk, &shf[2])) return;
  DPX_GSTAMP(4);
  // the slabs' partial products added up: one wave per distinct product (a <= q; the mirrored entry holds the same partials), every wave's
  // loads requested before the first sum (one memory round trip for the finishing workgroup instead of one per product), the same
  // additions in the same order as ever; the finished matrix stays in shared memory for the test (and goes to G for whoever looks)
  constexpr int EPW = (NP + 3) / 4;
  float ldv[EPW][4];
#pragma unroll
  for (int k = 0; k < EPW; ++k) {
    const int e = wave + 4 * k;
    int a = 0, e0 = 0;
    while (e < NP && e >= e0 + (BT - a)) { e0 += BT - a; ++a; }
    const int q = a + (e - e0);
    const bool on = e < NP && q < B;
    const float* pe = partial + ((long)a * B + q) * nblk;
#pragma unroll
    for (int j = 0; j < 4; ++j) ldv[k][j] = (on && lane + 64 * j < nblk) ? dpx_ld_agent(pe + lane + 64 * j) : 0.f;
  }
#pragma unroll
  for (int k = 0; k < EPW; ++k) {
    const int e = wave + 4 * k;
    int a = 0, e0 = 0;
    while (e < NP && e >= e0 + (BT - a)) { e0 += BT - a; ++a; }
    const int q = a + (e - e0);
    if (e >= NP || q >= B) continue;
    float a0 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (lane + 64 * j < nblk) a0 += ldv[k][j];
    const float v = wave_sum(a0);
    if (lane == 0) {
      Gs[a * B + q] = v;
      Gs[q * B + a] = v;
      G[a * B + q] = v;
      G[q * B + a] = v;
    }
  }
  __syncthreads();
  DPX_GSTAMP(5);
  cg_test_block(S, Gs, rawd, shf, init_rtol);
  DPX_GSTAMP(6);
  if (host_flags) {
    __syncthreads();
    if (threadIdx.x == 0) {
      // (done, n_done, tag) in ONE 8-byte store the host reads in one piece: nothing else travels through memory with it (the iterate is
      //  consumed by later launches of this stream), so no system-scope fence -- each one writes the L2's dirty lines back -- is needed
      const unsigned long long word = ((unsigned long long)(unsigned)host_tag << 32) | (unsigned)((S.flags()[1] << 1) | (S.flags()[0] ? 1 : 0));
      __hip_atomic_store((unsigned long long*)host_flags, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  DPX_GSTAMP(7);
}

static int gram_blocks(long npb) {
  long g = (npb + GR_CH - 1) / GR_CH;
  if (g > 1024) g = 1024;
  if (g < 1) g = 1;
  return (int)g;
}

static int dot_blocks(long npb) {
  long g = (npb + 256 * 8 - 1) / (256 * 8);
  if (g > 256) g = 256;
  if (g < 1) g = 1;
  return (int)g;
}

static bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

static int pack_terms(TermPack& T, const dpx_term* terms, int n, const char* who, bool& vec_ok) {
  if (n < 0 || n > DPX_MAX_TERMS || (n > 0 && !terms)) {
    set_error("%s: nterms must be in [0, %d]", who, DPX_MAX_TERMS);
    return DPX_ERR_ARG;
  }
  T.n = n;
  for (int i = 0; i < n; ++i) {
    T.t[i] = terms[i];
    if (!terms[i].v || !terms[i].u) {
      set_error("%s: term %d has a null state pointer", who, i);
      return DPX_ERR_ARG;
    }
    if (terms[i].linop < DPX_LIN_IDENTITY || terms[i].linop > DPX_LIN_GRAD_W || terms[i].prox < 0 || terms[i].prox > DPX_PROX_EXTERNAL) {
      set_error("%s: term %d has an unknown linop/prox code", who, i);
      return DPX_ERR_ARG;
    }
    vec_ok &= aligned16(terms[i].v) && aligned16(terms[i].u) && aligned16(terms[i].u_out);
  }
  return DPX_OK;
}

}  // namespace dpx

using namespace dpx;

extern "C" int dpx_admm_zupdate(const float* x, const dpx_term* terms, int nterms, int B, int C, int H, int W,
                                dpx_stream_t stream) {
  DPX_REQUIRE(x && B > 0 && C > 0 && H > 0 && W > 0, "dpx_admm_zupdate: bad arguments");
  TermPack T;
  bool vec = (W % 4 == 0) && aligned16(x);
  int rc = pack_terms(T, terms, nterms, "dpx_admm_zupdate", vec);
  if (rc) return rc;
  if (nterms == 0) return DPX_OK;
  const long n = (long)B * C * H * W;
  if (vec)
    DPX_LAUNCH("k_zupdate", (k_zupdate<4>), dim3(grid_for8(n / 4, 256, 8192)), dim3(256), 0, (hipStream_t)stream, x, T, B, C, H, W);
  else
    DPX_LAUNCH("k_zupdate", (k_zupdate<1>), dim3(grid_for8(n, 256, 8192)), dim3(256), 0, (hipStream_t)stream, x, T, B, C, H, W);
  return launch_status("dpx_admm_zupdate");
}

extern "C" int dpx_admm_rhs(float* rhs, const float* ktb, const float* rho, const dpx_term* terms, int nterms, int B, int C,
                            int H, int W, dpx_stream_t stream) {
  DPX_REQUIRE(rhs && rho && B > 0 && C > 0 && H > 0 && W > 0, "dpx_admm_rhs: bad arguments");
  TermPack T;
  bool vec = (W % 4 == 0) && aligned16(rhs) && (!ktb || aligned16(ktb));
  int rc = pack_terms(T, terms, nterms, "dpx_admm_rhs", vec);
  if (rc) return rc;
  const long n = (long)B * C * H * W;
  if (vec)
    DPX_LAUNCH("k_rhs", (k_rhs<4>), dim3(grid_for8(n / 4, 256, 8192)), dim3(256), 0, (hipStream_t)stream, rhs, ktb, rho, T, B, C, H, W);
  else
    DPX_LAUNCH("k_rhs", (k_rhs<1>), dim3(grid_for8(n, 256, 8192)), dim3(256), 0, (hipStream_t)stream, rhs, ktb, rho, T, B, C, H, W);
  return launch_status("dpx_admm_rhs");
}

// rhs = rho_b sum_i K_i^T K_i x0: dpx_admm_rhs for a state that comes straight from ADMM.initialize (algo/admm.py:61-67: v_i = K_i x0, u_i = 0)
// without that state -- the same differences in the same order (K^T (K x0 - 0) for grad: (x[h] - x[h-1]) - (x[h+1] - x[h]), circular)
__global__ void k_rhs_fresh(float* __restrict__ rhs, const float* __restrict__ x0, const float* __restrict__ rho, int nI, int nH, int nW, int B, int C,
                            int H, int W) {
  const long total = (long)B * C * H * W;
  for (long i = (long)xcd_block() * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int w = (int)(i % W);
    const long row = i / W;
    const int h = (int)(row % H);
    const long plane = row / H;
    const float* p = x0 + plane * H * W;
    const float c = p[(long)h * W + w];
    float acc = 0.f;
    for (int k = 0; k < nI; ++k) acc += c;
    if (nH) {
      const float up = p[(long)(h ? h - 1 : H - 1) * W + w], dn = p[(long)(h + 1 < H ? h + 1 : 0) * W + w];
      const float t = (c - up) - (dn - c);
      for (int k = 0; k < nH; ++k) acc += t;
    }
    if (nW) {
      const float lf = p[(long)h * W + (w ? w - 1 : W - 1)], rt = p[(long)h * W + (w + 1 < W ? w + 1 : 0)];
      const float t = (c - lf) - (rt - c);
      for (int k = 0; k < nW; ++k) acc += t;
    }
    rhs[i] = rho[plane / C] * acc;
  }
}
extern "C" int dpx_admm_rhs_fresh(float* rhs, const float* x0, const float* rho, const int* linops, int nterms, int B, int C, int H, int W,
                                  dpx_stream_t stream) {
  DPX_REQUIRE(rhs && x0 && rho && linops && rhs != x0 && nterms >= 1 && nterms <= DPX_MAX_TERMS && B > 0 && C > 0 && H > 0 && W > 0,
              "dpx_admm_rhs_fresh: bad arguments");
  int nI = 0, nH = 0, nW = 0;
  for (int i = 0; i < nterms; ++i) {
    DPX_REQUIRE(linops[i] >= DPX_LIN_IDENTITY && linops[i] <= DPX_LIN_GRAD_W, "dpx_admm_rhs_fresh: term %d: operator %d", i, linops[i]);
    nI += linops[i] == DPX_LIN_IDENTITY;
    nH += linops[i] == DPX_LIN_GRAD_H;
    nW += linops[i] == DPX_LIN_GRAD_W;
  }
  const long n = (long)B * C * H * W;
  DPX_LAUNCH("k_rhs_fresh", k_rhs_fresh, dim3(grid_for8(n, 256, 8192)), dim3(256), 0, (hipStream_t)stream, rhs, x0, rho, nI, nH, nW, B, C, H, W);
  return launch_status("dpx_admm_rhs_fresh");
}

extern "C" int dpx_admm_zupdate_rhs(const float* x, const dpx_term* terms, int nterms, float* rhs, const float* ktb, const float* rho_next,
                                    int dual, int emit_v, int B, int C, int H, int W, dpx_stream_t stream) {
  DPX_REQUIRE(x && rhs && rho_next && rhs != x && B > 0 && C > 0 && H > 0 && W > 0 && nterms > 0, "dpx_admm_zupdate_rhs: bad arguments");
  TermPack T;
  bool vec = (W % 4 == 0) && aligned16(x) && aligned16(rhs) && (!ktb || aligned16(ktb));
  int rc = pack_terms(T, terms, nterms, "dpx_admm_zupdate_rhs", vec);
  if (rc) return rc;
  for (int i = 0; i < nterms; ++i)
    DPX_REQUIRE(terms[i].prox != DPX_PROX_EXTERNAL && terms[i].u_out && terms[i].u_out != terms[i].u,
                "dpx_admm_zupdate_rhs: term %d needs a closed-form prox and a double-buffered dual (u_out != u)", i);
  for (int i = 0; i < nterms; ++i)
    DPX_REQUIRE(!(terms[i].reserved & DPX_TERM_NO_DUAL) || !dual, "dpx_admm_zupdate_rhs: DPX_TERM_NO_DUAL (term %d) excludes dual = 1", i);
  const long n = (long)B * C * H * W;
  if (vec)
    DPX_LAUNCH("k_zupdate_rhs", (k_zupdate_rhs<4>), dim3(grid_for8(n / 4, 256, 8192)), dim3(256), 0, (hipStream_t)stream, x, rhs, ktb, rho_next, T,
               dual, emit_v, B, C, H, W);
  else
    DPX_LAUNCH("k_zupdate_rhs", (k_zupdate_rhs<1>), dim3(grid_for8(n, 256, 8192)), dim3(256), 0, (hipStream_t)stream, x, rhs, ktb, rho_next, T, dual,
               emit_v, B, C, H, W);
  return launch_status("dpx_admm_zupdate_rhs");
}

extern "C" int dpx_grad(const float* x, float* y, int dim, int adjoint, int B, int C, int H, int W, dpx_stream_t stream) {
  DPX_REQUIRE(x && y && x != y, "dpx_grad: null or aliased pointers");
  DPX_REQUIRE(dim == 0 || dim == 1, "dpx_grad: dim must be 0 (H) or 1 (W)");
  DPX_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "dpx_grad: bad shape");
  const long n = (long)B * C * H * W;
  DPX_LAUNCH("k_grad", k_grad, dim3(grid_for8(n, 256, 8192)), dim3(256), 0, (hipStream_t)stream, x, y, dim, adjoint, (long)B * C, H, W);
  return launch_status("dpx_grad");
}

extern "C" int dpx_prox(int kind, const float* v, float* out, const float* lam, float alpha, const float* off, int B,
                        long n_per_batch, dpx_stream_t stream) {
  DPX_REQUIRE(v && out && B > 0 && n_per_batch > 0, "dpx_prox: bad arguments");
  DPX_REQUIRE(kind >= DPX_PROX_NORM1 && kind <= DPX_PROX_SUMSQ, "dpx_prox: unknown kind %d", kind);
  DPX_LAUNCH("k_prox", k_prox, dim3(grid_for(B * n_per_batch, 256, 8192)), dim3(256), 0, (hipStream_t)stream, kind, v, out, lam,
                     alpha, off, B, n_per_batch);
  return launch_status("dpx_prox");
}

// backward of k_prox at the point d (t = d - offset, l = lam_b * alpha):
//   soft-threshold: J = [|t| > l],  dprox/dl = -sign(t) [|t| > l]
//   nonneg        : J = [t > 0],    dprox/dl = 0
//   sum-squares   : J = 1/(1 + 2l), dprox/dl = -2 t / (1 + 2l)^2
__global__ void k_prox_bwd(int kind, const float* __restrict__ d, const float* __restrict__ g, float* __restrict__ gd,
                           float* __restrict__ dlam, const float* __restrict__ lam, float alpha, const float* __restrict__ off, int B,
                           long npb) {
  const long total = (long)B * npb;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / npb);
    const float t = d[i] - (off ? off[i] : 0.f);
    const float l = (lam ? lam[b] : 0.f) * alpha;
    float J, dl;
    if (kind == DPX_PROX_NORM1) {
      const bool pass = fabsf(t) > l;
      J = pass ? 1.f : 0.f;
      dl = pass ? (t > 0.f ? -1.f : 1.f) : 0.f;
    } else if (kind == DPX_PROX_NONNEG) {
      J = t > 0.f ? 1.f : 0.f;
      dl = 0.f;
    } else {
      const float s = 1.f / (1.f + 2.f * l);
      J = s;
      dl = -2.f * t * s * s;
    }
    if (gd) gd[i] = J * (g ? g[i] : 1.f);
    if (dlam) dlam[i] = dl;
  }
}

extern "C" int dpx_prox_bwd(int kind, const float* d, const float* g, float* gd, float* dlam, const float* lam, float alpha,
                            const float* offset, int B, long n_per_batch, dpx_stream_t stream) {
  DPX_REQUIRE(d && (gd || dlam) && (g || !gd) && B > 0 && n_per_batch > 0, "dpx_prox_bwd: bad arguments");
  DPX_REQUIRE(kind >= DPX_PROX_NORM1 && kind <= DPX_PROX_SUMSQ, "dpx_prox_bwd: unknown kind %d", kind);
  DPX_LAUNCH("k_prox_bwd", k_prox_bwd, dim3(grid_for(B * n_per_batch, 256, 8192)), dim3(256), 0, (hipStream_t)stream, kind, d, g, gd, dlam,
             lam, alpha, offset, B, n_per_batch);
  return launch_status("dpx_prox_bwd");
}

extern "C" int dpx_lincomb(float* out, int n, const float* const* x, const float* coef, const float* const* coef_b, int B,
                           long n_per_batch, dpx_stream_t stream) {
  DPX_REQUIRE(out && x && coef && n >= 1 && n <= 4 && B > 0 && n_per_batch > 0, "dpx_lincomb: bad arguments");
  LinPack L;
  L.n = n;
  for (int i = 0; i < n; ++i) {
    DPX_REQUIRE(x[i], "dpx_lincomb: operand %d is null", i);
    L.x[i] = x[i];
    L.c[i] = coef[i];
    L.cb[i] = coef_b ? coef_b[i] : nullptr;
  }
  DPX_LAUNCH("k_lincomb", k_lincomb, dim3(grid_for(B * n_per_batch, 256, 8192)), dim3(256), 0, (hipStream_t)stream, out, L, B, n_per_batch);
  return launch_status("dpx_lincomb");
}

__global__ void __launch_bounds__(256) k_mul(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ out, long npb,
                                             int w_images) {
  const int b = blockIdx.y;
  const float* wb = w + (w_images > 1 ? (long)b * npb : 0L);
  for (long i = blockIdx.x * 256L + threadIdx.x; i < npb; i += (long)gridDim.x * 256L) out[(long)b * npb + i] = x[(long)b * npb + i] * wb[i];
}

extern "C" int dpx_mul(const float* x, const float* w, float* out, int B, long n_per_image, int w_images, dpx_stream_t stream) {
  DPX_REQUIRE(x && w && out && B > 0 && n_per_image > 0, "dpx_mul: bad arguments");
  DPX_REQUIRE(w_images == 1 || w_images == B, "dpx_mul: the weight must hold 1 or B=%d images (got %d)", B, w_images);
  DPX_LAUNCH("k_mul", k_mul, dim3(grid_for(n_per_image, 256, 2048), B, 1), dim3(256), 0, (hipStream_t)stream, x, w, out, n_per_image, w_images);
  return launch_status("dpx_mul");
}

__global__ void __launch_bounds__(256) k_wss_prox(const float* __restrict__ v, const float* __restrict__ ktb, int ktb_images,
                                                  const float* __restrict__ diag, int diag_images, const float* __restrict__ lam,
                                                  float* __restrict__ out, long npb) {
  const int b = blockIdx.y;
  const float l = lam[b];
  const float* kb = ktb + (ktb_images > 1 ? (long)b * npb : 0L);
  const float* db = diag + (diag_images > 1 ? (long)b * npb : 0L);
  for (long i = blockIdx.x * 256L + threadIdx.x; i < npb; i += (long)gridDim.x * 256L)
    out[(long)b * npb + i] = (kb[i] + l * v[(long)b * npb + i]) / (db[i] + l);
}

extern "C" int dpx_wss_prox(const float* v, const float* ktb, int ktb_images, const float* diag, int diag_images, const float* lam,
                            float* out, int B, long n_per_image, dpx_stream_t stream) {
  DPX_REQUIRE(v && ktb && diag && lam && out && B > 0 && n_per_image > 0, "dpx_wss_prox: bad arguments");
  DPX_REQUIRE((ktb_images == 1 || ktb_images == B) && (diag_images == 1 || diag_images == B), "dpx_wss_prox: tables must hold 1 or B images");
  DPX_LAUNCH("k_wss_prox", k_wss_prox, dim3(grid_for(n_per_image, 256, 2048), B, 1), dim3(256), 0, (hipStream_t)stream, v, ktb, ktb_images, diag,
             diag_images, lam, out, n_per_image);
  return launch_status("dpx_wss_prox");
}

__global__ void __launch_bounds__(256) k_mul_color(const float* __restrict__ x, const float* __restrict__ srf, float* __restrict__ out,
                                                   int transpose, int C, int C2, long hw) {
  const int b = blockIdx.y;
  const int cin = transpose ? C2 : C, cout = transpose ? C : C2;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < (long)cout * hw; i += (long)gridDim.x * 256L) {
    const long p = i % hw;
    const int co = (int)(i / hw);
    float acc = 0.f;
    for (int ci = 0; ci < cin; ++ci) {
      const float s = transpose ? srf[(long)co * C2 + ci] : srf[(long)ci * C2 + co];
      acc = fmaf(s, x[((long)b * cin + ci) * hw + p], acc);
    }
    out[((long)b * cout + co) * hw + p] = acc;
  }
}

extern "C" int dpx_mul_color(const float* x, const float* srf, float* out, int transpose, int B, int C, int C2, long hw,
                             dpx_stream_t stream) {
  DPX_REQUIRE(x && srf && out && B > 0 && C > 0 && C2 > 0 && hw > 0, "dpx_mul_color: bad arguments");
  const long n = (long)(transpose ? C : C2) * hw;
  DPX_LAUNCH("k_mul_color", k_mul_color, dim3(grid_for(n, 256, 2048), B, 1), dim3(256), 0, (hipStream_t)stream, x, srf, out, transpose, C, C2, hw);
  return launch_status("dpx_mul_color");
}

// ---- closed-form super-resolution data term (proxfn/fast/sr.py:45-126) -------------------------------------------------
__global__ void __launch_bounds__(256) k_upsample_zero(const float* __restrict__ y, float* __restrict__ out, int sf, long planes, int h, int w) {
  const int H = h * sf, W = w * sf;
  const long total = planes * H * W;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
    const int xx = (int)(i % W);
    const long r = i / W;
    const int yy = (int)(r % H);
    const long p = r / H;
    out[i] = (yy % sf == 0 && xx % sf == 0) ? y[(p * h + yy / sf) * w + xx / sf] : 0.f;
  }
}

__global__ void __launch_bounds__(256) k_cplx_mul(float2* __restrict__ out, const float2* __restrict__ a, const float2* __restrict__ bb,
                                                  int conj_a, long npb, int a_images) {
  const int b = blockIdx.y;
  const float2* ab = a + (a_images > 1 ? (long)b * npb : 0L);
  for (long i = blockIdx.x * 256L + threadIdx.x; i < npb; i += (long)gridDim.x * 256L) {
    float2 av = ab[i];
    if (conj_a) av.y = -av.y;
    const float2 bv = bb[(long)b * npb + i];
    out[(long)b * npb + i] = make_float2(av.x * bv.x - av.y * bv.y, av.x * bv.y + av.y * bv.x);
  }
}

// one thread per output frequency; the sf*sf aliases of its block are gathered from L2
__global__ void __launch_bounds__(256) k_sisr_update(const float2* __restrict__ FRin, float2* __restrict__ FX, const float2* __restrict__ FB,
                                                     int fb_planes, const float* __restrict__ lam, float I, int sf, int C, int H, int W) {
  const int b = blockIdx.z, c = blockIdx.y;
  const int h = H / sf, w = W / sf;
  const long plane = ((long)b * C + c) * H * W;
  const float2* fb = FB + (fb_planes == 1 ? 0L : (fb_planes == C ? (long)c : ((long)b * C + c))) * H * W;
  const float il = I * lam[b];
  const float inv_n = 1.f / (float)(sf * sf);
  for (long i = blockIdx.x * 256L + threadIdx.x; i < (long)H * W; i += (long)gridDim.x * 256L) {
    const int v = (int)(i % W), u = (int)(i / W);
    const int u0 = u % h, v0 = v % w;
    float2 fbr = make_float2(0.f, 0.f);
    float invw = 0.f;
    for (int a = 0; a < sf; ++a)
      for (int e = 0; e < sf; ++e) {
        const long k = (long)(u0 + a * h) * W + (v0 + e * w);
        const float2 B2 = fb[k], R = FRin[plane + k];
        fbr.x += B2.x * R.x - B2.y * R.y;
        fbr.y += B2.x * R.y + B2.y * R.x;
        invw += B2.x * B2.x + B2.y * B2.y;
      }
    fbr.x *= inv_n;
    fbr.y *= inv_n;
    invw *= inv_n;
    const float d = invw + il;
    const float2 q = make_float2(fbr.x / d, fbr.y / d);            // invWBR
    const float2 Bc = fb[i], R = FRin[plane + i];
    // FBC * invWBR = conj(B) * q
    const float2 t = make_float2(Bc.x * q.x + Bc.y * q.y, Bc.x * q.y - Bc.y * q.x);
    const float den = il + 1e-9f;
    FX[plane + i] = make_float2((R.x - t.x) / den, (R.y - t.y) / den);
  }
}

extern "C" int dpx_upsample_zero(const float* y, float* out, int sf, long planes, int h, int w, dpx_stream_t stream) {
  DPX_REQUIRE(y && out && sf >= 1 && planes > 0 && h > 0 && w > 0, "dpx_upsample_zero: bad arguments");
  DPX_LAUNCH("k_upsample_zero", k_upsample_zero, dim3(grid_for(planes * h * sf * w * sf, 256, 8192)), dim3(256), 0, (hipStream_t)stream, y, out,
             sf, planes, h, w);
  return launch_status("dpx_upsample_zero");
}

extern "C" int dpx_cplx_mul(void* out, const void* a, const void* bb, int conj_a, int B, long n_per_image, int a_images,
                            dpx_stream_t stream) {
  DPX_REQUIRE(out && a && bb && B > 0 && n_per_image > 0 && (a_images == 1 || a_images == B), "dpx_cplx_mul: bad arguments");
  DPX_LAUNCH("k_cplx_mul", k_cplx_mul, dim3(grid_for(n_per_image, 256, 2048), B, 1), dim3(256), 0, (hipStream_t)stream, (float2*)out,
             (const float2*)a, (const float2*)bb, conj_a, n_per_image, a_images);
  return launch_status("dpx_cplx_mul");
}

extern "C" int dpx_sisr_update(void* FR, const void* FB, int fb_planes, const float* lam, float I, int sf, int B, int C, int H, int W,
                               dpx_stream_t stream) {
  DPX_REQUIRE(FR && FB && lam && sf >= 1 && B > 0 && C > 0 && H > 0 && W > 0, "dpx_sisr_update: bad arguments");
  DPX_REQUIRE(H % sf == 0 && W % sf == 0, "dpx_sisr_update: the scale factor %d must divide the image size %dx%d", sf, H, W);
  DPX_REQUIRE(fb_planes == 1 || fb_planes == C || fb_planes == B * C, "dpx_sisr_update: FB must have 1, C or B*C planes (got %d)", fb_planes);
  // out of place inside: aliases of other frequencies are read while this one is written -> use a second buffer view:
  // FR is [2][B][C][H][W]: the first half is the input, the second half receives FX (the caller passes both)
  const long n = (long)B * C * H * W;
  DPX_LAUNCH("k_sisr_update", k_sisr_update, dim3(grid_for((long)H * W, 256, 1024), C, B), dim3(256), 0, (hipStream_t)stream,
             (const float2*)FR, (float2*)FR + n, (const float2*)FB, fb_planes, lam, I, sf, C, H, W);
  return launch_status("dpx_sisr_update");
}

__global__ void __launch_bounds__(256) k_cplx_scale(float2* __restrict__ out, const float2* __restrict__ a, const float* __restrict__ w,
                                                    long npb, int w_images) {
  const int b = blockIdx.y;
  const float* wb = w + (w_images > 1 ? (long)b * npb : 0L);
  for (long i = blockIdx.x * 256L + threadIdx.x; i < npb; i += (long)gridDim.x * 256L) {
    const float2 v = a[(long)b * npb + i];
    const float s = wb[i];
    out[(long)b * npb + i] = make_float2(s * v.x, s * v.y);
  }
}

extern "C" int dpx_cplx_scale(void* out, const void* a, const float* w, int B, long n_per_image, int w_images, dpx_stream_t stream) {
  DPX_REQUIRE(out && a && w && B > 0 && n_per_image > 0 && (w_images == 1 || w_images == B), "dpx_cplx_scale: bad arguments");
  DPX_LAUNCH("k_cplx_scale", k_cplx_scale, dim3(grid_for(n_per_image, 256, 2048), B, 1), dim3(256), 0, (hipStream_t)stream, (float2*)out,
             (const float2*)a, w, n_per_image, w_images);
  return launch_status("dpx_cplx_scale");
}

// ---- complex-iterate arithmetic of the CS-MRI solver (contrib/csmri.py:156-171, proxfn/fast/csmri.py:14-25) ----
struct CplxPack {
  const void* x[4];
  float c[4];
  int cx[4];
  int n;
};
template <bool OUT_COMPLEX>
__global__ void __launch_bounds__(256) k_cplx_lincomb(void* __restrict__ out, CplxPack L, long n) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
    float re = 0.f, im = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k < L.n) {
        if (L.cx[k]) {
          const float2 v = ((const float2*)L.x[k])[i];
          re = fmaf(L.c[k], v.x, re);
          im = fmaf(L.c[k], v.y, im);
        } else {
          re = fmaf(L.c[k], ((const float*)L.x[k])[i], re);
        }
      }
    }
    if (OUT_COMPLEX) ((float2*)out)[i] = make_float2(re, im);
    else ((float*)out)[i] = re;
  }
}

__global__ void __launch_bounds__(256) k_csmri_update(float2* __restrict__ z, const float2* __restrict__ y, const unsigned char* __restrict__ mask,
                                                       int mask_images, const float* __restrict__ lam, float num_psi, long n_per_image) {
  const int b = blockIdx.y;
  const float l = lam[b];
  const float den = 1.f + l * num_psi;
  const unsigned char* mk = mask + (mask_images > 1 ? (long)b * n_per_image : 0L);
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n_per_image; i += (long)gridDim.x * 256L) {
    if (mk[i]) {
      const long j = (long)b * n_per_image + i;
      const float2 zz = z[j], yy = y[j];
      z[j] = make_float2((l * zz.x + yy.x) / den, (l * zz.y + yy.y) / den);
    }
  }
}

extern "C" int dpx_csmri_update(void* z, const void* y, const unsigned char* mask, int mask_images, const float* lam, float num_psi, int B,
                                long n_per_image, dpx_stream_t stream) {
  DPX_REQUIRE(z && y && mask && lam && B > 0 && n_per_image > 0, "dpx_csmri_update: bad arguments");
  DPX_REQUIRE(mask_images == 1 || mask_images == B, "dpx_csmri_update: mask must hold 1 or B=%d images (got %d)", B, mask_images);
  DPX_LAUNCH("k_csmri_update", k_csmri_update, dim3(grid_for(n_per_image, 256, 2048), B, 1), dim3(256), 0, (hipStream_t)stream, (float2*)z,
             (const float2*)y, mask, mask_images, lam, num_psi, n_per_image);
  return launch_status("dpx_csmri_update");
}

extern "C" int dpx_cplx_lincomb(void* out, int out_complex, int n, const void* const* x, const int* x_complex, const float* coef,
                                long n_elems, dpx_stream_t stream) {
  DPX_REQUIRE(out && x && x_complex && coef && n >= 1 && n <= 4 && n_elems > 0, "dpx_cplx_lincomb: bad arguments");
  CplxPack L;
  L.n = n;
  for (int i = 0; i < 4; ++i) {
    L.x[i] = i < n ? x[i] : nullptr;
    L.c[i] = i < n ? coef[i] : 0.f;
    L.cx[i] = i < n ? (x_complex[i] != 0) : 0;
    DPX_REQUIRE(i >= n || x[i], "dpx_cplx_lincomb: operand %d is null", i);
  }
  if (out_complex)
    DPX_LAUNCH("k_cplx_lincomb", k_cplx_lincomb<true>, dim3(grid_for(n_elems, 256, 8192)), dim3(256), 0, (hipStream_t)stream, out, L, n_elems);
  else
    DPX_LAUNCH("k_cplx_lincomb", k_cplx_lincomb<false>, dim3(grid_for(n_elems, 256, 8192)), dim3(256), 0, (hipStream_t)stream, out, L, n_elems);
  return launch_status("dpx_cplx_lincomb");
}

extern "C" size_t dpx_bdot_ws_bytes(int B, long n_per_batch) {
  const int nb = dot_blocks(n_per_batch) > gram_blocks(n_per_batch) ? dot_blocks(n_per_batch) : gram_blocks(n_per_batch);
  return (size_t)B * B * nb * sizeof(float);
}

extern "C" int dpx_bdot(const float* x, const float* y, float* out, int B, long n_per_batch, void* ws, dpx_stream_t stream) {
  DPX_REQUIRE(x && y && out && ws && B > 0 && n_per_batch > 0, "dpx_bdot: bad arguments");
  const int nblk = dot_blocks(n_per_batch);
  DPX_LAUNCH("k_dot_partial", k_dot_partial, dim3(nblk, B, 1), dim3(256), 0, (hipStream_t)stream, x, y, (float*)ws, n_per_batch, 0);
  DPX_LAUNCH("k_dot_finish", k_dot_finish, dim3(B), dim3(256), 0, (hipStream_t)stream, (const float*)ws, out, nblk);
  return launch_status("dpx_bdot");
}

namespace dpx {
// Gram pass + finish + stop rule in one launch (dpx_cg_masked_fft's fused iteration, B <= 32); ws: B * B * gram_blocks floats
// x / p / Ap non-null: the pending update x += alpha p, r -= alpha A p of the previous iteration is applied on the way (r is written)
int gram_test_fused(float* r, float* G, void* state, int B, long n_per_batch, void* ws, unsigned* counter, float init_rtol, float* x, const float* p,
                    const float* Ap, int* host_flags, int host_tag, hipStream_t s) {
  const int env_blk = 0;                              // (0 = by size)
  int nblk = gram_blocks(n_per_batch);
  // the finishing workgroup adds up B * B * nblk partial products: for larger batches fewer, longer slab walks (B = 32: 64 workgroups)
  if (B > 8) {
    int cap = 65536 / (B * B);
    cap = cap < 32 ? 32 : cap;
    if (nblk > cap) nblk = cap;
  }
  if (env_blk > 0 && env_blk < nblk) nblk = env_blk;
  const bool al16 = ((size_t)r % 16 == 0) && (!x || (((size_t)x % 16 == 0) && ((size_t)p % 16 == 0) && ((size_t)Ap % 16 == 0)));
  if (B <= 8 && n_per_batch % 4 == 0 && al16 && tune(TUNE_CG_GRAM_SMALL) != 2) {
    int nb = (int)((n_per_batch / 4 + 255) / 256);
    nb = nb > 256 ? 256 : (nb < 1 ? 1 : nb);
    if (env_blk > 0 && env_blk < nb) nb = env_blk;
    const CgState S{(float*)state, B};
#define DPX_GRAM_SMALL(BT)                                                                                                                       \
  do {                                                                                                                                           \
    if (x)                                                                                                                                       \
      DPX_LAUNCH("k_gram_small_test_upd", (k_gram_small_test<BT, true>), dim3(nb), dim3(256), 0, s, r, (float*)ws, G, S, n_per_batch, nb, counter,  \
                 init_rtol, x, p, Ap, host_flags, host_tag);                                                                                               \
    else                                                                                                                                         \
      DPX_LAUNCH("k_gram_small_test", (k_gram_small_test<BT, false>), dim3(nb), dim3(256), 0, s, r, (float*)ws, G, S, n_per_batch, nb, counter,     \
                 init_rtol, x, p, Ap, host_flags, host_tag);                                                                                               \
  } while (0)
    if (B <= 1) DPX_GRAM_SMALL(1);
    else if (B <= 2) DPX_GRAM_SMALL(2);
    else if (B <= 4) DPX_GRAM_SMALL(4);
    else DPX_GRAM_SMALL(8);
#undef DPX_GRAM_SMALL
    return launch_status("gram_test_fused");
  }
  if (x)
    DPX_LAUNCH("k_gram_tile_test_upd", k_gram_tile_test<true>, dim3(nblk), dim3(256), 0, s, r, (float*)ws, G, CgState{(float*)state, B}, n_per_batch,
               nblk, counter, init_rtol, x, p, Ap, host_flags, host_tag);
  else
    DPX_LAUNCH("k_gram_tile_test", k_gram_tile_test<false>, dim3(nblk), dim3(256), 0, s, r, (float*)ws, G, CgState{(float*)state, B}, n_per_batch,
               nblk, counter, init_rtol, x, p, Ap, host_flags, host_tag);
  return launch_status("gram_test_fused");
}
}  // namespace dpx

extern "C" int dpx_bgram(const float* r, float* out, int B, long n_per_batch, void* ws, dpx_stream_t stream) {
  DPX_REQUIRE(r && out && ws && B > 0 && n_per_batch > 0, "dpx_bgram: bad arguments");
  const int nblk = B <= 32 ? gram_blocks(n_per_batch) : dot_blocks(n_per_batch);
  if (B <= 32)
    DPX_LAUNCH("k_gram_tile", k_gram_tile, dim3(nblk), dim3(256), 0, (hipStream_t)stream, r, (float*)ws, B, n_per_batch, nblk);
  else
    DPX_LAUNCH("k_dot_partial", k_dot_partial, dim3(nblk, B, B), dim3(256), 0, (hipStream_t)stream, r, r, (float*)ws, n_per_batch, 1);
  DPX_LAUNCH("k_dot_finish", k_dot_finish, dim3(B * B), dim3(256), 0, (hipStream_t)stream, (const float*)ws, out, nblk);
  return launch_status("dpx_bgram");
}
