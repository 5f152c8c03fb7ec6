// Power-of-two planes (H in {256,512,1024}, W in {256,...,2048}): the rFFT2 -> operator -> irFFT2 pipeline on
// the register-radix transform of dpx_fft_reg.h.
//
//   k_rows_r2c_p2 : one wave (or part of one) per image row: row read as (x[2n], x[2n+1]) complex pairs with
//                   coalesced 8-byte loads, length-W/2 transform in registers, real-input untangling with
//                   in-wave shuffles, half spectrum written with coalesced stores; the (real) Nyquist bins go
//                   to a small side array [P][H] so that every column of the main array is an ordinary
//                   complex column (no packed DC/Nyquist special case in the column kernel).
//   k_cols_p2     : a tile of COLS adjacent spectrum columns of one plane per workgroup: forward column
//                   transform, per-frequency operator on registers, inverse column transform, in place.
//   k_rows_c2r_p2 : inverse of the first.
// Every element crosses HBM once per kernel (8 B/element algorithmic traffic each) and LDS twice per 1-D
// transform.
#include "dpx_fft_reg.h"

#include <cstdlib>
#include <type_traits>

namespace dpx {

// ---------------------------------------------------------------------------------------------
// rows
// ---------------------------------------------------------------------------------------------
// Register <-> index maps of one row (M = W/2 complex points, T lanes, V = M/T values per lane).  Power-of-two M: register m holds
// point t + m T before and after a transform.  M = 3 * 2^k (fft_reg_x3): register m = r VS + a holds point 3 (t + a T) + r in the
// image domain and bin (t + a T) + r M/3 in the frequency domain.  krow: the part of the frequency map that does not depend on t;
// the bin M - k that the real-input (un)tangling pairs with bin k sits in lane (T - t) % T, register pne(m) (t != 0) / peq(m) (t == 0).
template <int M, int T> struct RowMap {
  // (M = 384 on T = 64 lanes: the one-wave 6 * 8 * 8 transform fft384_wave with the power-of-two register map -- the two-kernel
  //  iteration's 768-wide rows; the staged kernels keep M = 384 on 16 lanes with the interleaved map)
  static constexpr bool W384 = (M == 384 && T == 64);
  static constexpr bool R3 = (M % 3 == 0) && !W384;
  static constexpr int V = M / T, VS = R3 ? V / 3 : V, N = R3 ? M / 3 : M;
  static constexpr int LDS_SLOTS = R3 ? 3 * LdsSeq<N>::SLOTS : LdsSeq<M>::SLOTS;
  __device__ static constexpr int krow(int m) { return R3 ? (m % VS) * T + (m / VS) * N : m * T; }
  __device__ static constexpr int pne(int m) { return R3 ? (2 - m / VS) * VS + (VS - 1 - m % VS) : V - 1 - m; }
  __device__ static constexpr int peq(int m) {
    return R3 ? ((m % VS) ? (2 - m / VS) * VS + (VS - m % VS) : ((3 - m / VS) % 3) * VS) : (V - m) % V;
  }
  // the lane's pixel pairs of one image row (row: the row's first pair).  M = 3 * 2^k: a lane owns 3 adjacent pairs per a -- 24 contiguous
  // bytes, moved as 16 + 8 (pair-by-pair the three 24-byte-strided loads take 3.5x longer: 104 us instead of 30 for 8x3x768^2)
  typedef float row_f4 __attribute__((ext_vector_type(4), aligned(8)));
  typedef float row_f2 __attribute__((ext_vector_type(2), aligned(8)));
  template <int NT = 0> __device__ static __forceinline__ void load_pairs(float2 (&v)[V], const float2* __restrict__ row, int t) {
    if constexpr (R3) {
#pragma unroll
      for (int a = 0; a < VS; ++a) {
        const float* p = (const float*)(row + 3 * (t + a * T));
        const row_f4 q = NT ? __builtin_nontemporal_load((const row_f4*)p) : *(const row_f4*)p;
        const row_f2 w = NT ? __builtin_nontemporal_load((const row_f2*)(p + 4)) : *(const row_f2*)(p + 4);
        v[a] = make_float2(q.x, q.y);
        v[VS + a] = make_float2(q.z, q.w);
        v[2 * VS + a] = make_float2(w.x, w.y);
      }
    } else {
#pragma unroll
      for (int m = 0; m < V; ++m) v[m] = ld_stream<NT>(row + t + m * T);
    }
  }
  template <class F> __device__ static __forceinline__ void store_pairs(float2* __restrict__ row, int t, F value) {
    if constexpr (R3) {
#pragma unroll
      for (int a = 0; a < VS; ++a) {
        float* p = (float*)(row + 3 * (t + a * T));
        const float2 v0 = value(a), v1 = value(VS + a), v2 = value(2 * VS + a);
        row_f4 q = {v0.x, v0.y, v1.x, v1.y};
        row_f2 w = {v2.x, v2.y};
        *(row_f4*)p = q;
        *(row_f2*)(p + 4) = w;
      }
    } else {
#pragma unroll
      for (int m = 0; m < V; ++m) row[t + m * T] = value(m);
    }
  }
  // spectrum offset of bin krow(m) relative to the lane's bin t (column-tile-major: bin k of image row h at ((k >> 3) H + h) 8 + (k & 7))
  __device__ static constexpr size_t koff(int m, int H) { return (size_t)(krow(m) / SPEC_TILE) * H * SPEC_TILE; }
  template <int DIR> __device__ static __forceinline__ void fft(float2 (&v)[V], float2* lds, int t, const float2* __restrict__ twW) {
    if constexpr (W384) fft384_wave_tab<DIR>(v, lds, t, twW, 2, WaveSync());
    else if constexpr (R3) fft_reg_x3<N, T, DIR>(v, lds, t, twW, 2, WaveSync());
    else fft_reg<M, T, DIR>(v, lds, t, twW, 2, WaveSync());
  }
  // Z (the transform of the row read as complex pairs) -> X (the row's half spectrum); returns the (real) Nyquist bin in lane t = 0
  //   X[k] = E[k] + w^k O[k], E = (Z[k] + conj Z[M-k])/2, O = -i (Z[k] - conj Z[M-k])/2
  __device__ static __forceinline__ float untangle(const float2 (&v)[V], float2 (&X)[V], int t, int lane, const float2* __restrict__ twW) {
    const int plane = (lane & ~(T - 1)) | ((T - t) & (T - 1));
    float nyq = 0.f;
#pragma unroll
    for (int m = 0; m < V; ++m) {
      const float2 got = make_float2(__shfl(v[pne(m)].x, plane), __shfl(v[pne(m)].y, plane));
      float2 own = v[peq(m)];
      if constexpr (R3) {     // (there the select between two elements of v becomes one dynamically indexed access: the array would live in scratch)
        DPX_OPAQUE(own.x);
        DPX_OPAQUE(own.y);
      }
      const float2 zm = cconj(t == 0 ? own : got);
      const int k = t + krow(m);
      const float2 zk = v[m];
      if (k == 0) {
        X[m] = make_float2(zk.x + zk.y, 0.f);                               // DC (real)
        nyq = zk.x - zk.y;
      } else {
        const float2 e = cscale(cadd(zk, zm), 0.5f);
        const float2 d = cscale(csub(zk, zm), 0.5f);
        X[m] = cadd(e, cmul(make_float2(d.y, -d.x), twW[k]));
      }
    }
    return nyq;
  }
  // the inverse: X (+ the Nyquist bin xn) -> the complex sequence whose inverse transform is the row's pixel pairs
  __device__ static __forceinline__ void tangle(const float2 (&X)[V], float2 (&v)[V], float xn, int t, int lane, const float2* __restrict__ twW) {
    const int plane = (lane & ~(T - 1)) | ((T - t) & (T - 1));
#pragma unroll
    for (int m = 0; m < V; ++m) {
      const float2 got = make_float2(__shfl(X[pne(m)].x, plane), __shfl(X[pne(m)].y, plane));
      float2 own = X[peq(m)];
      if constexpr (R3) {
        DPX_OPAQUE(own.x);
        DPX_OPAQUE(own.y);
      }
      const float2 xm = cconj(t == 0 ? own : got);
      const int k = t + krow(m);
      const float2 xk = X[m];
      if (k == 0) {
        v[m] = make_float2(xk.x + xn, xk.x - xn);       // DC and Nyquist bins of a real row are real: imaginary round-off dropped
      } else {
        const float2 e = cadd(xk, xm);
        const float2 d = cmulc(csub(xk, xm), twW[k]);
        v[m] = make_float2(e.x - d.y, e.y + d.x);
      }
    }
  }
};

template <int M, int T>
__global__ void __launch_bounds__(256) k_rows_r2c_p2(const float* __restrict__ x, float2* __restrict__ spec, float2* __restrict__ side,
                                                      int nrows, int H, const float2* __restrict__ twW) {
  using RM = RowMap<M, T>;
  constexpr int V = M / T, SPB = 256 / T, S = RM::LDS_SLOTS;
  __shared__ float2 lds[SPB * S];
  const int tid = threadIdx.x, seq = tid / T, t = tid % T;
  const int row = blockIdx.x * SPB + seq;
  const bool live = row < nrows;
  float2 v[V];
  RM::load_pairs(v, (const float2*)(x + (size_t)(live ? row : 0) * (2 * M)), t);
  RM::template fft<-1>(v, lds + seq * S, t, twW);
  float2 X[V];
  const float nyq = RM::untangle(v, X, t, tid & 63, twW);
  const int rr = live ? row : 0, pl = rr / H, hh = rr - pl * H;
  float2* out = spec + (size_t)pl * H * M + (size_t)hh * SPEC_TILE + (t % SPEC_TILE) + (size_t)(t / SPEC_TILE) * H * SPEC_TILE;
  if (live && t == 0) side[row] = make_float2(nyq, 0.f);
  if (live) {
#pragma unroll
    for (int m = 0; m < V; ++m) out[RM::koff(m, H)] = X[m];
  }
}

template <int M, int T>
__global__ void __launch_bounds__(256) k_rows_c2r_p2(const float2* __restrict__ spec, const float2* __restrict__ side,
                                                      float* __restrict__ y, int nrows, int H, const float2* __restrict__ twW, float scale) {
  using RM = RowMap<M, T>;
  constexpr int V = M / T, SPB = 256 / T, S = RM::LDS_SLOTS;
  __shared__ float2 lds[SPB * S];
  const int tid = threadIdx.x, seq = tid / T, t = tid % T;
  const int row = blockIdx.x * SPB + seq;
  const bool live = row < nrows;
  const int rr = live ? row : 0, pl = rr / H, hh = rr - pl * H;
  const float2* in = spec + (size_t)pl * H * M + (size_t)hh * SPEC_TILE + (t % SPEC_TILE) + (size_t)(t / SPEC_TILE) * H * SPEC_TILE;
  float2 X[V], v[V];
#pragma unroll
  for (int m = 0; m < V; ++m) X[m] = in[RM::koff(m, H)];
  RM::tangle(X, v, side[rr].x, t, tid & 63, twW);
  RM::template fft<+1>(v, lds + seq * S, t, twW);
  if (live) RM::store_pairs((float2*)(y + (size_t)rr * (2 * M)), t, [&](int m) { return make_float2(v[m].x * scale, v[m].y * scale); });
}

// The seed of the two-kernel iteration in one pass: the row transform of  rho_b sum_i K_i^T (v_i - u_i)  (dpx_admm_rhs followed by
// dpx_rfft_rows: 112 + 60 us at 8x3x1024^2, and the right-hand side image written and read back in between).  One wave per image
// row; the stencils' neighbours come from the adjacent lane (grad_W) / from the row above, read again (grad_H).
struct SeedTerms {
  const float* v[DPX_MAX_TERMS];
  const float* u[DPX_MAX_TERMS];
  int linop[DPX_MAX_TERMS];
  int n;
  const float* x0;       // FRESH: the iterate the state was initialised from (v_i = K_i x0, u_i = 0: admm.py:61-67)
};
// FRESH: the state comes straight from ADMM.initialize -- v_i = K_i x0 and u_i = 0 -- so v_i - u_i is recomputed from x0's rows (the
// same fp32 differences K.forward stored, minus an exact zero: bit-identical spectra) and the pass reads ONE image instead of 2 n.
template <int M, int T, bool FRESH>
__global__ void __launch_bounds__(256) k_seed_rows(SeedTerms S_, const float* __restrict__ rho, float2* __restrict__ spec, float2* __restrict__ side,
                                                    int nrows, int H, int C, const float2* __restrict__ twW) {
  using RM = RowMap<M, T>;
  static_assert(!RM::R3, "power-of-two rows only (the two-kernel iteration's widths)");
  constexpr int V = M / T, SPB = 256 / T, S = RM::LDS_SLOTS;
  __shared__ float2 lds[SPB * S];
  const int tid = threadIdx.x, seq = tid / T, t = tid % T;
  const int row = blockIdx.x * SPB + seq;
  const bool live = row < nrows;
  const int rr = live ? row : 0, pl = rr / H, hh = rr - pl * H;
  const int lane = tid & 63, lbase = lane & ~(T - 1);
  const size_t here = (size_t)rr * M, above = ((size_t)pl * H + (hh == 0 ? H - 1 : hh - 1)) * M;      // float2 offsets of the two rows
  const size_t below = ((size_t)pl * H + (hh == H - 1 ? 0 : hh + 1)) * M;
  float2 acc[V];
#pragma unroll
  for (int m = 0; m < V; ++m) acc[m] = make_float2(0.f, 0.f);
  float2 xh[V], xab[V], xbe[V];                           // FRESH: this row of x0 and its two neighbours, requested together (one round trip)
  if constexpr (FRESH) {
    const float2* xr = (const float2*)S_.x0;
    bool has_h = false;
    for (int i = 0; i < S_.n; ++i) has_h |= S_.linop[i] == DPX_LIN_GRAD_H;
#pragma unroll
    for (int m = 0; m < V; ++m) xh[m] = xr[here + t + m * T];
    if (has_h) {
#pragma unroll
      for (int m = 0; m < V; ++m) {
        xab[m] = xr[above + t + m * T];
        xbe[m] = xr[below + t + m * T];
      }
    }
  }
  for (int i = 0; i < S_.n; ++i) {
    const float2* vr = (const float2*)S_.v[i];
    const float2* ur = (const float2*)S_.u[i];
    float2 y[V];
    if constexpr (FRESH) {
      if (S_.linop[i] == DPX_LIN_IDENTITY) {
#pragma unroll
        for (int m = 0; m < V; ++m) y[m] = xh[m];
      } else if (S_.linop[i] == DPX_LIN_GRAD_W) {           // x[w+1] - x[w]; pixel 2n+2 is the neighbour lane's .x
#pragma unroll
        for (int m = 0; m < V; ++m) {
          const float nx_same = __shfl(xh[m].x, lbase | ((t + 1) & (T - 1)));
          const float nx_wrap = __shfl(xh[(m + 1) % V].x, lbase);
          const float xr_ = (t == T - 1) ? nx_wrap : nx_same;
          y[m] = make_float2(xh[m].y - xh[m].x, xr_ - xh[m].y);
        }
      } else {                                              // x[h+1] - x[h]
#pragma unroll
        for (int m = 0; m < V; ++m) y[m] = csub(xbe[m], xh[m]);
      }
    } else {
#pragma unroll
      for (int m = 0; m < V; ++m) y[m] = csub(vr[here + t + m * T], ur[here + t + m * T]);
    }
    if (S_.linop[i] == DPX_LIN_IDENTITY) {
#pragma unroll
      for (int m = 0; m < V; ++m) acc[m] = cadd(acc[m], y[m]);
    } else if (S_.linop[i] == DPX_LIN_GRAD_W) {           // adjoint: y[w-1] - y[w]; pixel 2n-1 is the left lane's .y
#pragma unroll
      for (int m = 0; m < V; ++m) {
        const float l_same = __shfl(y[m].y, lbase | ((t + T - 1) & (T - 1)));
        const float l_wrap = __shfl(y[(m + V - 1) % V].y, lbase | (T - 1));
        const float left = (t == 0) ? l_wrap : l_same;
        acc[m] = make_float2(acc[m].x + (left - y[m].x), acc[m].y + (y[m].x - y[m].y));
      }
    } else {                                              // grad_H adjoint: y[h-1] - y[h]
#pragma unroll
      for (int m = 0; m < V; ++m) {
        float2 yu;
        if constexpr (FRESH) yu = csub(xh[m], xab[m]);
        else yu = csub(vr[above + t + m * T], ur[above + t + m * T]);
        acc[m] = cadd(acc[m], csub(yu, y[m]));
      }
    }
  }
  const float r = rho[pl / C];
  float2 v[V], X[V];
#pragma unroll
  for (int m = 0; m < V; ++m) v[m] = cscale(acc[m], r);
  RM::template fft<-1>(v, lds + seq * S, t, twW);
  const float nyq = RM::untangle(v, X, t, lane, twW);
  float2* out = spec + (size_t)pl * H * M + (size_t)hh * SPEC_TILE + (t % SPEC_TILE) + (size_t)(t / SPEC_TILE) * H * SPEC_TILE;
  if (live && t == 0) side[row] = make_float2(nyq, 0.f);
  if (live) {
#pragma unroll
    for (int m = 0; m < V; ++m) out[RM::koff(m, H)] = X[m];
  }
}

#ifndef DPX_PGD_LD_NT
#define DPX_PGD_LD_NT 0
#endif
#ifndef DPX_PGD_ST
#define DPX_PGD_ST 0
#endif
__device__ __forceinline__ float pgd_prox(int kind, float d, float lam) {     // (dpx_elementwise.hip::prox_eval, closed forms only)
  if (kind == DPX_PROX_NORM1) {
    const float m = fmaxf(fabsf(d) - lam, 0.f);
    return d > 0.f ? m : (d < 0.f ? -m : 0.f * m);
  }
  if (kind == DPX_PROX_NONNEG) return fmaxf(d, 0.f);
  return d / (1.f + 2.f * lam);
}

// One row pass of a proximal-gradient iteration (reference dprox/algo/pgd.py:26-54 with a circular-convolution data term):
//   inverse row transform of the column-processed spectrum  (= K^T K x, the Gram operator is ONE multiply by |OTF|^2)
//   -> y = x - rho_b (K^T K x - K^T b) -> x' = prox(y, alpha lam_b) -> forward row transform of x'  (the next iteration's input).
// k_rows_c2r_p2 and k_rows_r2c_p2 back to back on one wave's row, with the forward step and the prox on registers in between:
// with the column kernel that is 2 launches and 28 B / element per iteration instead of 5 launches and 56.
template <int M, int T>
__global__ void __launch_bounds__(256) k_pgd_rows(const float2* __restrict__ spec_in, float2* __restrict__ spec_out, float* __restrict__ x,
                                                   const float* __restrict__ ktb, const float* __restrict__ rho, const float* __restrict__ lam,
                                                   float alpha, int prox, int nrows, int H, int C, const float2* __restrict__ twW) {
  using RM = RowMap<M, T>;
  constexpr int V = M / T, SPB = 256 / T, S = RM::LDS_SLOTS;
  __shared__ float2 lds[SPB * S];
  const int tid = threadIdx.x, seq = tid / T, t = tid % T;
  const int row = blockIdx.x * SPB + seq;
  const bool live = row < nrows;
  const int rr = live ? row : 0, pl = rr / H, hh = rr - pl * H, bi = pl / C;
  const size_t toff = (size_t)pl * H * M + (size_t)hh * SPEC_TILE + (t % SPEC_TILE) + (size_t)(t / SPEC_TILE) * H * SPEC_TILE;
  const float2* side_in = spec_in + (size_t)nrows * M;
  float2 X[V], v[V], xo[V], kb[V];
#pragma unroll
  for (int m = 0; m < V; ++m) X[m] = ld_stream<DPX_PGD_LD_NT>(spec_in + toff + RM::koff(m, H));
  const float xn0 = side_in[rr].x;
  // the iterate and K^T b are requested with the spectrum: ONE memory round trip in front of the transform
  float2* xr = (float2*)(x + (size_t)rr * (2 * M));
  RM::load_pairs(xo, xr, t);
  if (ktb) {
    RM::template load_pairs<1>(kb, (const float2*)(ktb + (size_t)rr * (2 * M)), t);
  } else {
#pragma unroll
    for (int m = 0; m < V; ++m) kb[m] = make_float2(0.f, 0.f);
  }
  RM::tangle(X, v, xn0, t, tid & 63, twW);
  RM::template fft<+1>(v, lds + seq * S, t, twW);     // v[m] = two adjacent pixels of K^T K x
  const float r = rho[bi], th = lam ? lam[bi] * alpha : 0.f;
#pragma unroll
  for (int m = 0; m < V; ++m) {
    float2 y = make_float2(xo[m].x - r * (v[m].x - kb[m].x), xo[m].y - r * (v[m].y - kb[m].y));
    v[m] = make_float2(pgd_prox(prox, y.x, th), pgd_prox(prox, y.y, th));
  }
  if (live) RM::store_pairs(xr, t, [&](int m) { return v[m]; });
  if (!spec_out) return;                                              // (block-uniform: the last iteration needs no spectrum)
  WaveSync()();
  RM::template fft<-1>(v, lds + seq * S, t, twW);
  const float nyq = RM::untangle(v, X, t, tid & 63, twW);
  float2* side_out = spec_out + (size_t)nrows * M;
  if (live && t == 0) side_out[row] = make_float2(nyq, 0.f);
  if (live) {
#pragma unroll
    for (int m = 0; m < V; ++m) st_stream<DPX_PGD_ST>(spec_out + toff + RM::koff(m, H), X[m]);
  }
}

// ---------------------------------------------------------------------------------------------
// per-frequency operator (same semantics as dpx_fft.hip::spec_op)
// ---------------------------------------------------------------------------------------------
template <int OP>
__device__ __forceinline__ float2 spec_op_p2(float2 z, const SpecArgs& A, unsigned tix, float rho_b) {
  if constexpr (OP == OP_MUL) {
    return cscale(cmul(z, A.otf[tix]), A.scale);
  } else if constexpr (OP == OP_MULCONJ) {
    return cscale(cmulc(z, A.otf[tix]), A.scale);
  } else {
    const float2 dd = A.dd[tix];
    const float den = fmaf(rho_b, dd.y, dd.x) + A.eps;
    const float inv = A.scale * DPX_RCP(den);
    return make_float2((z.x + A.eps_num) * inv, z.y * inv);
  }
}

// ---------------------------------------------------------------------------------------------
// columns
// ---------------------------------------------------------------------------------------------
// DBG (tuning experiments only, default 0): bit0 = skip both transforms, bit1 = skip the operator's table loads
#ifndef DPX_COLS_LD_NT
#define DPX_COLS_LD_NT 0
#endif
#ifndef DPX_COLS_ADD_NT
#define DPX_COLS_ADD_NT 1       // the data spectrum is re-read once per iteration, ~1 GB of other traffic later: stream it (see dpx_iter.hip)
#endif
#ifndef DPX_COLS_ST
#define DPX_COLS_ST 1           // write-through, as for the row kernel's spectrum
#endif
#ifndef DPX_COLS_TBL_NT
#define DPX_COLS_TBL_NT 0
#endif
#ifndef DPX_COLS_TW_GLOBAL
#define DPX_COLS_TW_GLOBAL 1
#endif
constexpr int COLS_LD_NT = DPX_COLS_LD_NT, COLS_ADD_NT = DPX_COLS_ADD_NT, COLS_ST = DPX_COLS_ST;   // cache policy (dpx_common.h)
#ifndef DPX_COLS_BATCH_INNER
#define DPX_COLS_BATCH_INNER 1
#endif
#ifndef DPX_COLS_PERSIST
#define DPX_COLS_PERSIST 0       // > 0: persistent workgroups, that many per CU (tuning experiment)
#endif
// The Nyquist column of a plane (side array [P][H]) rides through the two transforms as the imaginary part of the plane's DC
// column: both are spectra of real sequences, so z = dc + i ny is ONE complex column, separated by Hermitian symmetry around
// the per-frequency operator (a = (Z[k] + conj Z[N-k]) / 2, i b = (Z[k] - conj Z[N-k]) / 2), as the generic path does
// (dpx_fft.hip).  The HBM layout does not change: the lanes of the DC column read and write the side array themselves.
// With separate workgroups for the Nyquist columns the launch had 3 x 512 + 3 workgroups at 8x3x1024^2 for 512 resident
// ones, and the three extra ran alone in a fourth round: 75.7 -> 68.9 us without them.
// The workgroup's work; PACK = its column c == 0 is the plane's packed (DC, Nyquist) column.  Two instantiations per kernel so that
// the packed variant's extra live values (and spills) stay out of the register allocation of the other 63 workgroups in 64.
// Tuning aid (variant builds with -DDPX_PAR_TRACE only, tools/par_trace.py): 100 MHz real-time stamps of every wave of the first 256 workgroups
#ifdef DPX_PAR_TRACE
__device__ unsigned long long dpx_cols_trace_buf[256 * 16 * 10];
#define DPX_CSTAMP(i)                                                                                                       \
  do {                                                                                                                      \
    if ((threadIdx.x & 63) == 0 && bid < 256) dpx_cols_trace_buf[(bid * 16 + (threadIdx.x >> 6)) * 10 + (i)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
extern "C" int dpx_dbg_cols_trace(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(dpx_cols_trace_buf), (size_t)n * sizeof(unsigned long long));
}
#else
#define DPX_CSTAMP(i) ((void)0)
#endif
template <int H, int T, int COLS, int OP, int DBG, bool PACK>
__device__ __forceinline__ void cols_body(const float2* __restrict__ spec_in, float2* __restrict__ spec_out, const SpecArgs& A, int C, int Ws, int P,
                                          const float2* __restrict__ twH, int bid, int nmain, bool is_side, int p, int j, unsigned sub_off) {
  constexpr int V = H / T;
  constexpr int S0 = LdsSeq<H>::SLOTS;
  constexpr int S = S0 + ((36 - S0 % 32) % 32);     // S % 32 == 4: adjacent columns start 8 banks apart
  // H = 3 * 2^k (fft_reg_x3): register m = r * V/3 + a holds image row 3 (t + a T) + r before / after the kernel and frequency row
  // (t + a T) + r H/3 in between; power-of-two H: row t + m T on both sides.  hrow / krow: the part of those maps that does not depend on t.
  constexpr bool R3 = (H % 3 == 0);
  constexpr int VS = R3 ? V / 3 : V, TMUL = R3 ? 3 : 1;
  auto hrow = [](int m) { return R3 ? 3 * (m % VS) * T + m / VS : m * T; };
  auto krow = [](int m) { return R3 ? (m % VS) * T + (m / VS) * (H / 3) : m * T; };
  HIP_DYNAMIC_SHARED(float2, smem_p2)
  DPX_CSTAMP(0);
  const int tid = threadIdx.x, c = tid % COLS, t = tid / COLS;
  // uniform (scalar) bases + 32-bit per-thread element offsets: one address VGPR per access
  size_t ubase;                                       // element offset of the tile's plane / of the side array
  unsigned off0, off0k, step, toff0, tbase;           // data offset of image row t TMUL / of frequency row t, offset of one row, table offset of row t, table base
  constexpr bool pack0 = PACK;
  if (!is_side) {
    ubase = (size_t)p * H * Ws + (size_t)j * H * SPEC_TILE;   // tile-major main part: element (row r, col c) of a tile at r*TILE + c
    off0k = (unsigned)((c / SPEC_TILE) * H * SPEC_TILE + t * SPEC_TILE + (c % SPEC_TILE)) + sub_off;
    off0 = off0k + (unsigned)((TMUL - 1) * t * SPEC_TILE);
    step = (unsigned)SPEC_TILE;
    toff0 = off0k;
    tbase = (unsigned)((p % C) * H * Ws + j * H * SPEC_TILE);
  } else {
    p = (bid - nmain) * COLS + c;
    if (p >= P) p = P - 1;                            // (P not a multiple of COLS: duplicate work, identical values)
    ubase = (size_t)P * H * Ws;
    off0k = (unsigned)(p * H + t);
    off0 = off0k + (unsigned)((TMUL - 1) * t);
    step = 1u;
    toff0 = (unsigned)((p % C) * H + t);
    tbase = (unsigned)C * H * Ws;
  }
  const int bi = p / C;
  const char* pin = (const char*)(spec_in + ubase);
  char* pout = (char*)(spec_out + ubase);
  float2* lds = smem_p2 + c * S;
  // the H column twiddles, shared by the workgroup: an LDS copy -- except for H = 2048, where the copy's 16 KB would leave room for
  // one 4-column workgroup per CU instead of two (81920 B each without it): there they are read from the table (L1 / L2 resident)
  constexpr bool TWG = DPX_COLS_TW_GLOBAL && (H >= 2048);
  const float2* twl = TWG ? twH : smem_p2 + COLS * S;
  float2 v[V];
  // Input tile.  DMA_IN: by LDS-DMA, 16 bytes per lane (a wave instruction moves 16 rows x 64 bytes = 1 KB instead of 8 x 64), into
  // the exchange buffer, each wave fetching exactly the rows its own lanes hold (piece j = rows t + T (2j + half), the table stage's
  // lane map: value m of this lane lands at float2 index m * 64 + lane of the wave's image).  Piece j of wave w lands in column region j
  // at slots [136 w, 136 w + 128) -- precisely the slots (17 t + m, t = 8 w .. 8 w + 7) this wave overwrites itself in the first pass,
  // after it has read them (one wave's LDS operations execute in order): no other wave's stage is touched, no barrier is needed.
  // Measured at 8x3x1024^2 (round 4, alternating runs on one box): k_cols_p2 72.3 us with it, 69.6 - 70.7 us without -- half the load
  // instructions and no address arithmetic, but the tile crosses the LDS once more and the first pass waits for a full vmcnt(0): OFF.
#ifndef DPX_COLS_DMA_IN
#define DPX_COLS_DMA_IN 0
#endif
  constexpr bool DMA_IN = DPX_COLS_DMA_IN && DPX_COLS_PACK0 && !R3 && COLS == 8 && V == 16 && SPEC_TILE == 16 && !DBG;
  if constexpr (DMA_IN) {
    constexpr int RPW = 64 / COLS, LPR = COLS / 2;
    const int lane_ = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane_ >> 5, li = lane_ & 31;
    const float2* src = (const float2*)pin + (unsigned)((RPW * wv + li / LPR + T * half) * SPEC_TILE + (li % LPR) * 2) + sub_off;
    float2* stg = smem_p2 + wv * (V * RPW + RPW);        // 17 * 8 slots per wave and column region
#pragma unroll
    for (int j = 0; j < V / 2; ++j) dpx_glds16<COLS_LD_NT>(src + j * 2 * T * SPEC_TILE, stg + j * S);
  } else {
#pragma unroll
    for (int m = 0; m < V; ++m) v[m] = ld_stream<COLS_LD_NT>((const float2*)(pin + (off0 + step * hrow(m)) * 8u));
  }
  const bool dc_lane = PACK && c == 0;               // the lanes carrying the packed (DC, Nyquist) column
  float sidev[PACK ? V : 1];
  if (dc_lane) {                                      // (row spectra: both columns are real on entry)
    const float* sidef = (const float*)(spec_in + (size_t)P * H * Ws + (size_t)p * H + TMUL * t);
#pragma unroll
    for (int m = 0; m < V; ++m) sidev[PACK ? m : 0] = sidef[2 * hrow(m)];
  }
  // DPX_COLS_PACK_EARLY: the packed column's Nyquist denominators are requested with the tile (V floats per lane of the packed
  // column's instantiation, alive through the forward transform) instead of behind the operator, where their memory round trip
  // made the launch's three packed workgroups its last to finish (1 x 3 x 1024^2: 17.0 us against 13.6 for the others)
#ifndef DPX_COLS_PACK_EARLY
#define DPX_COLS_PACK_EARLY 1
#endif
  constexpr bool PACK_EARLY = DPX_COLS_PACK_EARLY && PACK && OP == OP_SOLVE;
  float dnb_early[PACK_EARLY ? V : 1];
  if constexpr (PACK_EARLY) {
    if (dc_lane) {
      const float rb = A.rho ? A.rho[p / C] : 0.f;
      const float2* tsd0 = A.dd + (unsigned)C * H * Ws + (unsigned)(p % C) * H + t;
#pragma unroll
      for (int m = 0; m < V; ++m) {
        const float2 db = tsd0[krow(m)];
        dnb_early[m] = fmaf(rb, db.y, db.x) + A.eps;
      }
    }
  }
  if constexpr (!TWG) {
    for (int i = tid; i < H; i += T * COLS) smem_p2[COLS * S + i] = twH[i];
  }
  if constexpr (DMA_IN) {
    dpx_wait_vm<0>();                                 // this wave's pieces have landed (and its twiddle / side loads returned)
    const float2* stg = smem_p2 + __builtin_amdgcn_readfirstlane(tid >> 6) * (V * (64 / COLS) + 64 / COLS) + (tid & 63);
#pragma unroll
    for (int m = 0; m < V; ++m) v[m] = stg[(m >> 1) * S + (m & 1) * 64];
  }
  if (dc_lane) {
#pragma unroll
    for (int m = 0; m < V; ++m) v[m].y = sidev[PACK ? m : 0];
  }
  const float rho_b = (OP == OP_SOLVE && A.rho) ? A.rho[bi] : 0.f;
  const char* add = (OP == OP_SOLVE && A.add) ? (const char*)(A.add + ubase) : nullptr;
  // (no barrier here: the twiddle copy is first read in the second pass, behind the first pass's barrier)

  // The operator's table values of the tile (SOLVE: the interleaved denominators) are fetched by LDS-DMA into the
  // transform's exchange buffer while it is idle -- between the forward transform's last LDS read and the inverse
  // transform's first write -- so that they cost no registers; each wave fetches exactly the rows its own lanes use
  // (lane-linear image: float2 index m*64 + lane of the wave's 8 KB), so its own vmcnt wait is all the
  // synchronisation the data needs.  The data-spectrum values (HBM) are requested in one batch right behind them:
  // ONE memory round trip between the two transforms.
#ifndef DPX_COLS_MUL_DMA
#define DPX_COLS_MUL_DMA 1         // the multiply operators stage their OTF values the same way (direct loads at the operator: 81.5 us at 8x3x1024^2)
#endif
  constexpr bool DMA_TABLE = (OP == OP_SOLVE || DPX_COLS_MUL_DMA) && (COLS == 8 || COLS == 4) && (V % 2 == 0) && !(DBG & 2);
#ifndef DPX_COLS_EARLY_ADD
#define DPX_COLS_EARLY_ADD 1       // with the spectra served from the Infinity Cache the data spectrum is the HBM stream: start it one pass earlier (78.8 -> 76.0 us)
#endif
  constexpr bool EARLY_ADD = DPX_COLS_EARLY_ADD;        // request the data spectrum before the last pass's arithmetic
  constexpr int NAV = (OP == OP_SOLVE || DPX_COLS_PACK0) ? V : 1;
  float2 av[NAV];                                     // data spectrum (SOLVE); MUL: the packed column's correction
#pragma unroll
  for (int m = 0; m < NAV; ++m) av[m] = make_float2(0.f, 0.f);
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float2* tstage = smem_p2 + wave * (64 * V);
  auto fetch_table = [&]() {
    DPX_CSTAMP(8);
    if (DMA_TABLE && !is_side) {
      DPX_LDS_BARRIER();                                // every wave has read its last-pass inputs
      // piece j: rows T*(2j + half) + RPW*wave + li/LPR (half = lane / 32, li = lane % 32), columns 2*(li % LPR), +1
      constexpr int RPW = 64 / COLS, LPR = COLS / 2;      // rows per wave and 16-byte lanes per row
      int ln = lane;
      DPX_OPAQUE(ln);                                   // derive the source address here, not at kernel entry
      const int half = ln >> 5, li = ln & 31;
      const float2* src = (OP == OP_SOLVE ? A.dd : A.otf) + tbase + (unsigned)((RPW * wave + li / LPR) * SPEC_TILE + (li % LPR) * 2) + sub_off;
      if constexpr (R3) {                                 // (pieces 2j, 2j+1 are not a constant distance apart)
#pragma unroll
        for (int j = 0; j < V / 2; ++j) dpx_glds16<DPX_COLS_TBL_NT>(src + (half ? krow(2 * j + 1) : krow(2 * j)) * SPEC_TILE, tstage + j * 128);
      } else {
        src += T * half * SPEC_TILE;
#pragma unroll
        for (int j = 0; j < V / 2; ++j) dpx_glds16<DPX_COLS_TBL_NT>(src + j * 2 * T * SPEC_TILE, tstage + j * 128);
      }
    }
    if constexpr (OP == OP_SOLVE && EARLY_ADD) {
      unsigned offa = off0k;
      DPX_OPAQUE(offa);
      if (add) {
#pragma unroll
        for (int m = 0; m < V; ++m) av[m] = ld_stream<COLS_ADD_NT>((const float2*)(add + (offa + step * krow(m)) * 8u));
      }
    }
    DPX_CSTAMP(9);
  };
#ifdef DPX_PAR_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  DPX_CSTAMP(1);
#endif
  if (!(DBG & 1)) {
    if constexpr (R3) fft_reg_x3<H / 3, T, -1>(v, lds, t, twl, 1, BlockSync(), fetch_table);
    else fft_reg<H, T, -1>(v, lds, t, twl, 1, BlockSync(), fetch_table);
  }
  DPX_CSTAMP(2);
  __builtin_amdgcn_sched_barrier(0);
#ifdef DPX_COLS_PRIO
  __builtin_amdgcn_s_setprio(DPX_COLS_PRIO);          // experiment: workgroups past their forward transform go first
#endif
  // Packed column (its workgroup only, PACK): Z = A + iB is split into A = (Z + Zc)/2 and iB = (Z - Zc)/2, Zc[k] = conj Z[N-k], so
  // that the operator's factors of the DC column (fA) and of the Nyquist column (fB) reach their own parts:
  //     op(Z) = fA (A + eps) + fB (iB + i eps) = fA (Z + eps) + (fB - fA) iB + i fB eps.
  // The partner bins live in other waves: one exchange through an LDS buffer of H bins -- the 8 (S - H) slots behind the table
  // stage + the launch's extra bytes behind the twiddles.   pack_split: zval(m) -> Z[t + m T] of this lane, consume(m, iB).
  constexpr int XSLACK = COLS * (S - H);
  auto xslot = [&](int i) { return i < XSLACK ? smem_p2 + COLS * H + i : smem_p2 + COLS * S + (TWG ? 0 : H) + (i - XSLACK); };
  auto pack_split = [&](auto zval, auto consume) {
    DPX_LDS_BARRIER();                                // (no table stage on this path: every wave's last-pass reads are done)
    if (dc_lane) {
#pragma unroll
      for (int m = 0; m < V; ++m) xslot(t + krow(m))[0] = zval(m);
    }
    DPX_LDS_BARRIER();
    if (dc_lane) {
#pragma unroll
      for (int m = 0; m < V; ++m) {
        const int k = t + krow(m);
        consume(m, cscale(csub(zval(m), cconj(xslot(k ? H - k : 0)[0])), 0.5f));
      }
    }
  };
  const float2* tside = (OP == OP_SOLVE ? A.dd : A.otf) + (unsigned)C * H * Ws + (unsigned)(p % C) * H + t;    // Nyquist column's table values
  if constexpr (PACK && OP != OP_SOLVE) {
    // multiply: no division by the DC factor is possible, so the correction (oB - oA) iB is formed in front of the operator and added behind it
    // (its table values are loaded where they are used: three workgroups of the launch, and the registers matter more than their latency)
    pack_split([&](int m) { return v[m]; },
               [&](int m, float2 ib) {
                 const float2 dd = cscale(csub(tside[krow(m)], A.otf[tbase + toff0 + step * krow(m)]), A.scale);
                 av[m] = OP == OP_MULCONJ ? cmulc(ib, dd) : cmul(ib, dd);
               });
  }
  if constexpr (OP == OP_SOLVE) {
    unsigned offa = off0k;
    DPX_OPAQUE(offa);
    if (!EARLY_ADD && add) {
#pragma unroll
      for (int m = 0; m < V; ++m) av[m] = ld_stream<COLS_ADD_NT>((const float2*)(add + (offa + step * krow(m)) * 8u));
    }
    if (DMA_TABLE && !is_side) {
      if (add) dpx_wait_vm<V>();                        // the V data-spectrum loads above may stay in flight
      else dpx_wait_vm<0>();
#pragma unroll
      for (int m = 0; m < V; ++m) {
        const float2 dv = tstage[m * 64 + (offa & 0) + lane];
        const float2 z = cadd(v[m], av[m]);
        const float den = fmaf(rho_b, dv.y, dv.x) + A.eps;
        const float inv = A.scale * DPX_RCP(den);
        v[m] = make_float2((z.x + A.eps_num) * inv, z.y * inv);
      }
    } else {
#pragma unroll
      for (int m = 0; m < V; ++m) {
        const float2 z = cadd(v[m], av[m]);
        if (DBG & 2) v[m] = cscale(z, A.scale);
        else v[m] = spec_op_p2<OP>(z, A, tbase + toff0 + step * krow(m), rho_b);
        if (V > 8 && (m & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else {
#pragma unroll
    for (int m = 0; m < V; ++m) {
      if (DMA_TABLE && !is_side) {
        if (m == 0) dpx_wait_vm<0>();
        const float2 o = tstage[m * 64 + lane];
        v[m] = cscale(OP == OP_MULCONJ ? cmulc(v[m], o) : cmul(v[m], o), A.scale);
      } else {
        v[m] = spec_op_p2<OP>(v[m], A, tbase + toff0 + step * krow(m), rho_b);
      }
      if (DPX_COLS_PACK0) v[m] = cadd(v[m], av[m]);     // (zero except in the packed column)
      if (V > 8 && (m & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
  }
  if constexpr (PACK && OP == OP_SOLVE) {
    // solve: the operator above produced fA (Z + eps); Z is recovered from it (fA > 0), split, and the Nyquist part re-weighted
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");                    // (keeps the loads below behind the operator: hoisted across it they spill)
    const float iscale = 1.0f / A.scale;
    float dnb[V], g[V];                               // the Nyquist column's and the DC column's denominators
    if (!DMA_TABLE) DPX_LDS_BARRIER();                  // (with the table stage, its barrier already separates the last-pass reads from these writes)
    if (dc_lane) {
      int ts = t;
      DPX_OPAQUE(ts);
      const float2* tsd = A.dd + (unsigned)C * H * Ws + (unsigned)(p % C) * H + ts;
#pragma unroll
      for (int m = 0; m < V; ++m) {
        if constexpr (PACK_EARLY) {
          dnb[m] = dnb_early[m];
        } else {
          const float2 db = tsd[krow(m)];
          dnb[m] = fmaf(rho_b, db.y, db.x) + A.eps;
        }
      }
#pragma unroll
      for (int m = 0; m < V; ++m) {
        const float2 da = (DMA_TABLE && !(DBG & 2)) ? tstage[m * 64 + lane] : A.dd[tbase + toff0 + step * krow(m)];
        g[m] = fmaf(rho_b, da.y, da.x) + A.eps;
        xslot(t + krow(m))[0] = make_float2(fmaf(v[m].x, g[m] * iscale, -A.eps_num), v[m].y * (g[m] * iscale));
      }
    }
    DPX_LDS_BARRIER();
    if (dc_lane) {
#pragma unroll
      for (int m = 0; m < V; ++m) {
        const int k = t + krow(m);
        const float2 z = make_float2(fmaf(v[m].x, g[m] * iscale, -A.eps_num), v[m].y * (g[m] * iscale));
        const float2 ib = cscale(csub(z, cconj(xslot(k ? H - k : 0)[0])), 0.5f);
        const float fb = A.scale * DPX_RCP(dnb[m]), df = fb - A.scale * DPX_RCP(g[m]);
        v[m] = make_float2(fmaf(df, ib.x, v[m].x), fmaf(df, ib.y, fmaf(fb, A.eps_num, v[m].y)));
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  DPX_CSTAMP(3);
  DPX_LDS_BARRIER();
  DPX_CSTAMP(4);
  if (!(DBG & 1)) {
    if constexpr (R3) fft_reg_x3<H / 3, T, +1>(v, lds, t, twl, 1, BlockSync());
    else fft_reg<H, T, +1>(v, lds, t, twl, 1, BlockSync());
  }
  DPX_CSTAMP(5);
  unsigned off1 = off0;
  DPX_OPAQUE(off1);       // do not keep the load offsets alive for the stores
#pragma unroll
  for (int m = 0; m < V; ++m) st_stream<COLS_ST>((float2*)(pout + (off1 + step * hrow(m)) * 8u), dc_lane ? make_float2(v[m].x, 0.f) : v[m]);
  if (dc_lane) {                                      // both columns are real again: Re -> DC column, Im -> side array
    int te = threadIdx.x;
    DPX_OPAQUE(te);                                   // (address re-derived here instead of kept alive through the kernel)
    float2* so = spec_out + (size_t)P * H * Ws + (size_t)p * H + TMUL * (te / COLS);
#pragma unroll
    for (int m = 0; m < V; ++m) st_stream<COLS_ST>(so + hrow(m), make_float2(v[m].y, 0.f));
  }
  DPX_CSTAMP(6);
#ifdef DPX_PAR_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  DPX_CSTAMP(7);
#endif
}


template <int H, int T, int COLS, int OP, int DBG = 0>
#ifndef DPX_COLS_WPE
#define DPX_COLS_WPE ((H % 3 == 0 || (H >= 2048 && !DPX_COLS_TW_GLOBAL)) ? 2 : (T * COLS) >= 512 ? 4 : 3)     // waves per SIMD the register budget is sized for (H = 768: 61 KB of LDS -> 2 workgroups of 4 waves per CU)
#endif
__global__ void __launch_bounds__(T* COLS, DPX_COLS_WPE) k_cols_p2(const float2* __restrict__ spec_in, float2* __restrict__ spec_out, SpecArgs A,
                                                      int C, int Ws, int P, const float2* __restrict__ twH, int total_blocks) {
  const int tiles = Ws / COLS, nmain = P * tiles;
#if DPX_COLS_PERSIST
  // persistent form: the launch has at most DPX_COLS_PERSIST workgroups per CU's worth of blocks; each walks block ids bid, bid + grid, ...
  // (same XCD: the grid is a multiple of 8) -- the next tile's loads are issued while this tile's stores drain
  for (int bid = blockIdx.x; bid < total_blocks; bid += gridDim.x) {
#else
  const int bid = blockIdx.x;
#endif
  // main tiles: COLS adjacent columns of plane p.  side tiles (DPX_COLS_PACK0 = 0 only): the Nyquist columns of COLS consecutive planes.
  const bool is_side = bid >= nmain;                  // block-uniform
  int p = 0, j = 0;                                   // plane and first spectrum tile of this workgroup
  unsigned sub_off = 0;                               // column offset of this workgroup inside a wider spectrum tile
  if (!is_side) {
    p = bid / tiles;
    if constexpr (COLS >= SPEC_TILE) {
      j = (bid - p * tiles) * (COLS / SPEC_TILE);
    } else {
      // SPEC_TILE / COLS workgroups share a tile (and its 128-byte lines): they are placed 8 block ids apart so that
      // the round-robin block -> XCD assignment puts them on the same XCD, i.e. behind the same L2
      constexpr int SUB = SPEC_TILE / COLS;
      const int tiles_w = Ws / SPEC_TILE, Bn = P / C;   // spectrum tiles per plane, images
      if (DPX_COLS_BATCH_INNER && (C * tiles_w) % 8 == 0) {
        // ... and so are the Bn images' workgroups of one (channel, tile): the operator's table (denominators / OTF) is
        // shared by the batch, and with the images innermost on one XCD its lines are fetched from HBM once instead of
        // once per image (the streams in between would evict them from the 4 MB L2).
        // bid = 8 * ((g * Bn + b) * SUB + sub) + xcd,  (channel, tile) = g * 8 + xcd
        const int GB = (DPX_COLS_BATCH_INNER > 1 && Bn % DPX_COLS_BATCH_INNER == 0) ? DPX_COLS_BATCH_INNER : Bn;   // images per inner group
        const int per = 8 * GB * SUB, u = bid / per, r = bid - u * per, ng = (C * tiles_w) / 8;
        const int bg = u / ng, g = u - bg * ng;
        const int xs = r % 8, sidx = r / 8, ct = g * 8 + xs;
        // (channel, tile) pairs in tile-major order: the workgroups holding a packed (DC, Nyquist) column -- tile 0 of every channel,
        // a few microseconds longer than the rest -- are dispatched first instead of ending up in the launch's last round
        const int cch = DPX_COLS_PACK0 ? ct % C : ct / tiles_w;
        j = DPX_COLS_PACK0 ? ct / C : ct - cch * tiles_w;
        p = (bg * GB + sidx / SUB) * C + cch;
        sub_off = (unsigned)((sidx % SUB) * COLS);
      } else {
        const int q = bid - p * tiles;
        j = (q / (8 * SUB)) * 8 + (q % 8);
        sub_off = (unsigned)(((q % (8 * SUB)) / 8) * COLS);
      }
    }
  }
  if (DPX_COLS_PACK0 && !is_side && j == 0 && sub_off == 0)
    cols_body<H, T, COLS, OP, DBG, true>(spec_in, spec_out, A, C, Ws, P, twH, bid, nmain, is_side, p, j, sub_off);
  else
    cols_body<H, T, COLS, OP, DBG, false>(spec_in, spec_out, A, C, Ws, P, twH, bid, nmain, is_side, p, j, sub_off);
#if DPX_COLS_PERSIST
    __syncthreads();                                  // the inverse transform's last LDS reads / the next tile's first writes
  }
#endif
}

// Tuning probe (DPX_DEBUG_COLS=4, wrong results by design): the column kernel's HBM traffic with 16-byte accesses and
// no arithmetic -- the ceiling a wide-access version of k_cols_p2 could reach at the same occupancy.

// ---------------------------------------------------------------------------------------------
// host dispatch
// ---------------------------------------------------------------------------------------------
size_t pow2_spec_elems(int P, int H, int W) { return (size_t)P * H * (W / 2) + (size_t)P * H; }

bool pow2_path_available(int H, int W) {
  const bool hok = (H == 256 || H == 384 || H == 512 || H == 768 || H == 1024 || H == 1536 || H == 2048);
  const bool wok = (W == 256 || W == 512 || W == 768 || W == 1024 || W == 1536 || W == 2048);
  return hok && wok;
}

template <int M, int T>
static void launch_rows(bool fwd, const float* x, float2* spec, float* y, int nrows, int H, const float2* twW, float scale, hipStream_t s) {
  constexpr int SPB = 256 / T;
  const dim3 grid((nrows + SPB - 1) / SPB);
  float2* side = spec + (size_t)nrows * M;           // [P][H] Nyquist bins behind the main [P][H][W/2] array
  if (fwd)
    DPX_LAUNCH("k_rows_r2c_p2", (k_rows_r2c_p2<M, T>), grid, dim3(256), 0, s, x, spec, side, nrows, H, twW);
  else
    DPX_LAUNCH("k_rows_c2r_p2", (k_rows_c2r_p2<M, T>), grid, dim3(256), 0, s, (const float2*)spec, (const float2*)side, y, nrows, H, twW, scale);
}

static void rows_dispatch(bool fwd, int W, int H, const float* x, float2* spec, float* y, int nrows, const float2* twW, float scale, hipStream_t s) {
  switch (W) {
    case 256: launch_rows<128, 16>(fwd, x, spec, y, nrows, H, twW, scale, s); break;
    case 512: launch_rows<256, 32>(fwd, x, spec, y, nrows, H, twW, scale, s); break;
    case 768: launch_rows<384, 16>(fwd, x, spec, y, nrows, H, twW, scale, s); break;
    case 1536: launch_rows<768, 32>(fwd, x, spec, y, nrows, H, twW, scale, s); break;
    case 1024: launch_rows<512, 64>(fwd, x, spec, y, nrows, H, twW, scale, s); break;
    default: launch_rows<1024, 64>(fwd, x, spec, y, nrows, H, twW, scale, s); break;
  }
}

template <int H, int T, int COLS, int OP, int DBG = 0>
static void launch_cols(const float2* spec, float2* spec_out, const SpecArgs& A, int P, int C, int Ws, const float2* twH, hipStream_t s) {
  constexpr int S0 = LdsSeq<H>::SLOTS;
  constexpr int S = S0 + ((36 - S0 % 32) % 32);
  // + the rest of the packed column's H-bin exchange buffer (DPX_COLS_PACK0): 80 KB per workgroup at H = 1024, still two per CU
  constexpr bool TWG = DPX_COLS_TW_GLOBAL && (H >= 2048);      // (cols_body: no LDS copy of the twiddles)
  const size_t sh = (size_t)(COLS * S + (TWG ? 0 : H) + (DPX_COLS_PACK0 && H > COLS * (S - H) ? H - COLS * (S - H) : 0)) * sizeof(float2);
  static bool attr_done = false;
  if (!attr_done && sh > 48 * 1024) {
    hipFuncSetAttribute((const void*)k_cols_p2<H, T, COLS, OP, DBG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    attr_done = true;
  }
  const int total_blocks = P * (Ws / COLS) + (DPX_COLS_PACK0 ? 0 : (P + COLS - 1) / COLS);
  int grid = total_blocks;
#if DPX_COLS_PERSIST
  {
    const int cap = 256 * DPX_COLS_PERSIST;
    if (grid > cap) grid = cap;
  }
#endif
  DPX_LAUNCH("k_cols_p2", (k_cols_p2<H, T, COLS, OP, DBG>), dim3(grid), dim3(T * COLS), sh, s, spec, spec_out, A, C, Ws, P, twH, total_blocks);
}

#ifndef DPX_COLS_WG
#define DPX_COLS_WG (SPEC_TILE > 8 ? 8 : SPEC_TILE)
#endif
constexpr int COLS_WG = DPX_COLS_WG;   // columns per workgroup of the column kernel

template <int OP>
static void cols_dispatch(int H, const float2* spec, float2* spec_out, const SpecArgs& A, int P, int C, int Ws, const float2* twH, hipStream_t s) {
  // (measured and removed -- round 5: 4-column workgroups for launches of a few planes, 16.9 -> 22.8 us; the transform-free / wide-access
  //  probe forms of this kernel behind the debug knob: DESIGN.md section 3)
  switch (H) {
    case 256: launch_cols<256, 32, COLS_WG, OP>(spec, spec_out, A, P, C, Ws, twH, s); break;
    case 512: launch_cols<512, 64, COLS_WG, OP>(spec, spec_out, A, P, C, Ws, twH, s); break;
    case 384: launch_cols<384, 16, COLS_WG, OP>(spec, spec_out, A, P, C, Ws, twH, s); break;
    case 768: launch_cols<768, 32, COLS_WG, OP>(spec, spec_out, A, P, C, Ws, twH, s); break;
    case 1536: launch_cols<1536, 64, COLS_WG, OP>(spec, spec_out, A, P, C, Ws, twH, s); break;
    case 2048: launch_cols<2048, 128, 4, OP>(spec, spec_out, A, P, C, Ws, twH, s); break;      // (4 columns: 98 KB of LDS per workgroup)
    default: launch_cols<1024, 64, COLS_WG, OP>(spec, spec_out, A, P, C, Ws, twH, s); break;
  }
}

bool seed_rows_seq_pow2(const int* linops, int n, const float* rho, const float* x0, float2* spec, int P, int C, int H, int W, const void* table,
                        hipStream_t s);
template <int M, int T>
static void launch_seed(const SeedTerms& S_, const float* rho, float2* spec, int nrows, int H, int C, const float2* twW, hipStream_t s) {
  constexpr int SPB = 256 / T;
  if (S_.x0)
    DPX_LAUNCH("k_seed_rows", (k_seed_rows<M, T, true>), dim3((nrows + SPB - 1) / SPB), dim3(256), 0, s, S_, rho, spec, spec + (size_t)nrows * M, nrows, H,
               C, twW);
  else
    DPX_LAUNCH("k_seed_rows", (k_seed_rows<M, T, false>), dim3((nrows + SPB - 1) / SPB), dim3(256), 0, s, S_, rho, spec, spec + (size_t)nrows * M, nrows, H,
               C, twW);
}
int seed_rows_pow2(const dpx_term* terms, int nterms, const float* rho, const float* x0, float2* spec, int B, int C, int H, int W, const void* table,
                   hipStream_t stream) {
  SeedTerms S_{};
  S_.n = nterms;
  S_.x0 = x0;
  for (int i = 0; i < nterms; ++i) {
    S_.v[i] = terms[i].v;
    S_.u[i] = terms[i].u;
    S_.linop[i] = terms[i].linop;
  }
  const int nrows = B * C * H;
  if (x0) {                                             // the streaming form (dpx_iter.hip) where the plane fits it
    int ops[DPX_MAX_TERMS];
    for (int i = 0; i < nterms; ++i) ops[i] = terms[i].linop;
    if (seed_rows_seq_pow2(ops, nterms, rho, x0, spec, B * C, C, H, W, table, stream)) return launch_status("dpx_admm_seed_rows");
  }
  switch (W) {
    case 256: launch_seed<128, 16>(S_, rho, spec, nrows, H, C, tw_rows(table), stream); break;
    case 512: launch_seed<256, 32>(S_, rho, spec, nrows, H, C, tw_rows(table), stream); break;
    case 768: launch_seed<384, 64>(S_, rho, spec, nrows, H, C, tw_rows(table), stream); break;
    default: launch_seed<512, 64>(S_, rho, spec, nrows, H, C, tw_rows(table), stream); break;
  }
  return launch_status("dpx_admm_seed_rows");
}

int cols_solve_pow2(const float2* spec_in, float2* spec_out, const SpecArgs& A, int P, int C, int H, int W, const void* table,
                    hipStream_t stream) {
  cols_dispatch<OP_SOLVE>(H, spec_in, spec_out, A, P, C, W / 2, tw_cols(table, W), stream);
  return launch_status("cols_solve_pow2");
}

int rows_r2c_pow2(const float* x, float2* spec, int P, int H, int W, const void* table, hipStream_t stream) {
  rows_dispatch(true, W, H, x, spec, nullptr, P * H, tw_rows(table), 1.0f, stream);
  return launch_status("rows_r2c_pow2");
}

int rows_c2r_pow2(const float2* spec, float* y, int P, int H, int W, const void* table, hipStream_t stream) {
  rows_dispatch(false, W, H, nullptr, (float2*)spec, y, P * H, tw_rows(table), 1.0f, stream);
  return launch_status("rows_c2r_pow2");
}

int spectral_apply_pow2(const float* x, float* y, int op, const SpecArgs& A, int B, int C, int H, int W,
                        const void* table, void* ws, hipStream_t stream) {
  const int P = B * C, Ws = W / 2;
  float2* spec = (float2*)ws;
  float2* spec2 = spec + pow2_spec_elems(P, H, W);     // the column pass is out of place
  rows_dispatch(true, W, H, x, spec, nullptr, P * H, tw_rows(table), 1.0f, stream);
  switch (op) {
    case OP_MUL: cols_dispatch<OP_MUL>(H, spec, spec2, A, P, C, Ws, tw_cols(table, W), stream); break;
    case OP_MULCONJ: cols_dispatch<OP_MULCONJ>(H, spec, spec2, A, P, C, Ws, tw_cols(table, W), stream); break;
    default: cols_dispatch<OP_SOLVE>(H, spec, spec2, A, P, C, Ws, tw_cols(table, W), stream); break;
  }
  rows_dispatch(false, W, H, nullptr, spec2, y, P * H, tw_rows(table), 1.0f, stream);
  return launch_status("spectral_apply_pow2");
}


template <int M, int T>
static void launch_pgd_rows(const float2* sin, float2* sout, float* x, const float* ktb, const float* rho, const float* lam, float alpha, int prox,
                            int nrows, int H, int C, const float2* twW, hipStream_t s) {
  constexpr int SPB = 256 / T;
  DPX_LAUNCH("k_pgd_rows", (k_pgd_rows<M, T>), dim3((nrows + SPB - 1) / SPB), dim3(256), 0, s, sin, sout, x, ktb, rho, lam, alpha, prox, nrows, H,
             C, twW);
}

bool pgd_rows_seq_pow2(const float2* sin, float2* sout, float* x, const float* ktb, const float* rho, const float* lam, float alpha, int prox, int P,
                       int C, int H, int W, const void* table, hipStream_t s);        // dpx_iter.hip: the streaming row pass

int pgd_run_pow2(float* x, const float* ktb, const void* gram_otf, int prox, float alpha, const float* rho_tab, const float* lam_tab, int T,
                 int B, int C, int H, int W, const void* table, void* ws, hipStream_t stream) {
  const int P = B * C, Ws = W / 2;
  float2* spec = (float2*)ws;
  float2* spec2 = spec + pow2_spec_elems(P, H, W);
  SpecArgs A{};
  A.otf = (const float2*)gram_otf;
  A.scale = 1.0f / ((float)H * (float)W);
  rows_dispatch(true, W, H, x, spec, nullptr, P * H, tw_rows(table), 1.0f, stream);
  for (int it = 0; it < T; ++it) {
    cols_dispatch<OP_MUL>(H, spec, spec2, A, P, C, Ws, tw_cols(table, W), stream);
    float2* sout = it + 1 < T ? spec : nullptr;
    const float* rho = rho_tab + (size_t)it * B;
    const float* lam = lam_tab ? lam_tab + (size_t)it * B : nullptr;
    if (pgd_rows_seq_pow2(spec2, sout, x, ktb, rho, lam, alpha, prox, P, C, H, W, table, stream)) continue;
    switch (W) {
      case 256: launch_pgd_rows<128, 16>(spec2, sout, x, ktb, rho, lam, alpha, prox, P * H, H, C, tw_rows(table), stream); break;
      case 512: launch_pgd_rows<256, 32>(spec2, sout, x, ktb, rho, lam, alpha, prox, P * H, H, C, tw_rows(table), stream); break;
      case 768: launch_pgd_rows<384, 16>(spec2, sout, x, ktb, rho, lam, alpha, prox, P * H, H, C, tw_rows(table), stream); break;
      case 1536: launch_pgd_rows<768, 32>(spec2, sout, x, ktb, rho, lam, alpha, prox, P * H, H, C, tw_rows(table), stream); break;
      case 1024: launch_pgd_rows<512, 64>(spec2, sout, x, ktb, rho, lam, alpha, prox, P * H, H, C, tw_rows(table), stream); break;
      default: launch_pgd_rows<1024, 64>(spec2, sout, x, ktb, rho, lam, alpha, prox, P * H, H, C, tw_rows(table), stream); break;
    }
  }
  return launch_status("dpx_pgd_run");
}

}  // namespace dpx

using namespace dpx;

// T proximal-gradient iterations  x <- prox(x - rho_t (K^T K x - K^T b), alpha lam_t)  in place, 2 launches per iteration
// (reference dprox/algo/pgd.py:26-54; K a circular convolution: gram_otf = the |OTF|^2 table, ktb = K^T b or NULL).
extern "C" int dpx_pgd_supported(int H, int W, int prox) {
  return pow2_path_available(H, W) && (prox == DPX_PROX_NORM1 || prox == DPX_PROX_NONNEG || prox == DPX_PROX_SUMSQ);
}
extern "C" int dpx_pgd_run(float* x, const float* ktb, const void* gram_otf, int prox, float alpha, const float* rho_tab, const float* lam_tab,
                           int T, int B, int C, int H, int W, const void* table, void* spectrum_ws, dpx_stream_t stream) {
  DPX_REQUIRE(x && gram_otf && rho_tab && table && spectrum_ws && T >= 0 && B > 0 && C > 0, "dpx_pgd_run: bad arguments");
  DPX_REQUIRE(dpx_pgd_supported(H, W, prox), "dpx_pgd_run: unsupported plane %dx%d / prox %d", H, W, prox);
  if (T == 0) return DPX_OK;
  return pgd_run_pow2(x, ktb, gram_otf, prox, alpha, rho_tab, lam_tab, T, B, C, H, W, table, spectrum_ws, (hipStream_t)stream);
}
