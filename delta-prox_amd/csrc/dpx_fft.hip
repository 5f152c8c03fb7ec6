// Generic (any H, W) batched 2-D real FFT pipeline of the x-update and of conv/adjoint:
//
//   rows r2c  ->  [ columns c2c  ->  per-frequency operator  ->  columns c2c^-1 ]  ->  rows c2r
//
// Replaces the reference's eager `torch.fft.fftn -> (F + eps)/(diag + eps) -> ifftn -> real -> float`
// (dprox/proxfn/sum_square.py:150-156) and `fftn -> FB * Fx -> ifftn -> real` (dprox/linop/conv.py:31-41).
//
// This file is the size-generic path: LDS-resident Stockham autosort passes with a run-time radix
// list (2,3,4,5,7,8,11,13 in registers, any other prime by an O(p) per-output pass), half-length
// complex transform for even row lengths, Nyquist column packed into column 0.  Power-of-two
// planes take the register-radix kernels of dpx_fft_pow2.hip instead.
#include <cstdlib>

#include "dpx_cg_dev.h"
#include "dpx_fft_reg.h"
#include "dpx_prox_dev.h"

namespace dpx {

// ---------------------------------------------------------------------------------------------
// plans and tables
// ---------------------------------------------------------------------------------------------
Plan1D make_plan(int n) {
  Plan1D p;
  p.n = n;
  p.nf = 0;
  int m = n;
  const int pref[] = {8, 4, 2, 3, 5, 7, 11, 13};
  for (int r : pref)
    while (m % r == 0 && m > 1) {
      p.radix[p.nf++] = r;
      m /= r;
    }
  for (int q = 17; m > 1; q += 2)
    while (m % q == 0) {
      p.radix[p.nf++] = q;
      m /= q;
    }
  return p;
}

__global__ void k_twiddle_table(float2* tw, int n) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  double s, c;
  sincospi(-2.0 * (double)t / (double)n, &s, &c);   // exact at multiples of 1/4 turn
  tw[t] = make_float2((float)c, (float)s);
}

// ---------------------------------------------------------------------------------------------
// precision-generic complex helpers: CT = float2 (hot path) or double2 (one-off fp64 data spectrum)
// ---------------------------------------------------------------------------------------------
template <class CT> struct cx_traits;
template <> struct cx_traits<float2> { typedef float real; };
template <> struct cx_traits<double2> { typedef double real; };
template <class CT> __device__ __forceinline__ CT mk(typename cx_traits<CT>::real x, typename cx_traits<CT>::real y) {
  CT r;
  r.x = x;
  r.y = y;
  return r;
}
__device__ __forceinline__ float gfma(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ double gfma(double a, double b, double c) { return fma(a, b, c); }
template <class CT> __device__ __forceinline__ CT gmul(CT a, CT b) { return mk<CT>(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 gmul(float2 a, float2 b) { return cmul(a, b); }
template <class CT> __device__ __forceinline__ CT gadd(CT a, CT b) { return mk<CT>(a.x + b.x, a.y + b.y); }
template <class CT> __device__ __forceinline__ CT gsub(CT a, CT b) { return mk<CT>(a.x - b.x, a.y - b.y); }
template <int DIR, class CT> __device__ __forceinline__ CT gmul_i(CT a) { return DIR < 0 ? mk<CT>(a.y, -a.x) : mk<CT>(-a.y, a.x); }
// twiddle source: fp32 table lookup, or exact on-the-fly evaluation for the fp64 path
template <class CT> struct Twid;
template <> struct Twid<float2> {
  const float2* t;
  int len;
  __device__ __forceinline__ float2 get(long long idx) const { return t[idx]; }
};
template <> struct Twid<double2> {
  const double2* t;  // fp64 table exp(-2 pi i idx / len) (k_twiddle_table_f64); evaluating sincospi at every use instead put its
                     // argument reduction into scratch memory: 4.0 GB of scratch writes per 100 MB image
  int len;           // table length the indices refer to
  __device__ __forceinline__ double2 get(long long idx) const { return t[idx]; }
};

// ---------------------------------------------------------------------------------------------
// in-register butterflies
// ---------------------------------------------------------------------------------------------
template <int DIR, class CT> __device__ __forceinline__ void bfly2(CT& a, CT& b) {
  CT t = gsub(a, b);
  a = gadd(a, b);
  b = t;
}
template <int DIR, class CT> __device__ __forceinline__ void bfly4(CT (&v)[4]) {
  CT s0 = gadd(v[0], v[2]), d0 = gsub(v[0], v[2]);
  CT s1 = gadd(v[1], v[3]), d1 = gmul_i<DIR>(gsub(v[1], v[3]));
  v[0] = gadd(s0, s1);
  v[2] = gsub(s0, s1);
  v[1] = gadd(d0, d1);
  v[3] = gsub(d0, d1);
}
template <int DIR, class CT> __device__ __forceinline__ void bfly8(CT (&v)[8]) {
  typedef typename cx_traits<CT>::real RT;
  const RT h = (RT)0.70710678118654752440;
  CT e[4] = {v[0], v[2], v[4], v[6]};
  CT o[4] = {v[1], v[3], v[5], v[7]};
  bfly4<DIR, CT>(e);
  bfly4<DIR, CT>(o);
  // o[m] *= W_8^m  (W_8 = exp(DIR * i*pi/4))
  CT t1 = DIR < 0 ? mk<CT>((o[1].x + o[1].y) * h, (o[1].y - o[1].x) * h)
                      : mk<CT>((o[1].x - o[1].y) * h, (o[1].y + o[1].x) * h);
  CT t2 = gmul_i<DIR>(o[2]);
  CT t3 = DIR < 0 ? mk<CT>((o[3].y - o[3].x) * h, -(o[3].x + o[3].y) * h)
                      : mk<CT>(-(o[3].x + o[3].y) * h, (o[3].x - o[3].y) * h);
  v[0] = gadd(e[0], o[0]);
  v[4] = gsub(e[0], o[0]);
  v[1] = gadd(e[1], t1);
  v[5] = gsub(e[1], t1);
  v[2] = gadd(e[2], t2);
  v[6] = gsub(e[2], t2);
  v[3] = gadd(e[3], t3);
  v[7] = gsub(e[3], t3);
}
// odd radix on the half-size form: with a_k = v_k + v_{R-k}, b_k = v_k - v_{R-k} (k = 1 .. (R-1)/2)
//   X_j, X_{R-j} = (v_0 + sum_k cos(2 pi j k / R) a_k)  -+ i (sum_k sin(2 pi j k / R) b_k)          (forward; inverse: +-)
// -- (R-1)^2 / 2 real multiply-adds on complex operands instead of (R-1)^2 complex products (radix 5: 36 instead of 120 instructions;
// the size-generic passes are co-limited by vector-instruction issue).  w[n] = exp(DIR * 2 pi i n / R), n = 0 .. R-1: cosines and sines
// are the table's own values.  Explicit fma: both forms of the size-generic kernels (and the fp64 data-spectrum passes) share this
// function and must round identically wherever it is inlined.
template <int R, int DIR, class CT> __device__ __forceinline__ void bfly_odd_roots(CT (&v)[R], const CT (&w)[R]) {
  typedef typename cx_traits<CT>::real RT;
  constexpr int HR = (R - 1) / 2;
  CT a[HR + 1], b[HR + 1];
#pragma unroll
  for (int k = 1; k <= HR; ++k) {
    a[k] = gadd(v[k], v[R - k]);
    b[k] = gsub(v[k], v[R - k]);
  }
  const CT v0 = v[0];
  CT x0 = v0;
#pragma unroll
  for (int k = 1; k <= HR; ++k) x0 = gadd(x0, a[k]);
  v[0] = x0;
#pragma unroll
  for (int j = 1; j <= HR; ++j) {
    CT t = v0, u = mk<CT>((RT)0, (RT)0);
#pragma unroll
    for (int k = 1; k <= HR; ++k) {
      const int n = (j * k) % R;
      const RT c = w[n].x, sn = DIR > 0 ? w[n].y : -w[n].y;      // cos, sin of 2 pi n / R
      t = mk<CT>(gfma(c, a[k].x, t.x), gfma(c, a[k].y, t.y));
      u = mk<CT>(gfma(sn, b[k].x, u.x), gfma(sn, b[k].y, u.y));
    }
    const CT r = gmul_i<DIR>(u);                                  // -+ i u
    v[j] = gadd(t, r);
    v[R - j] = gsub(t, r);
  }
}
// odd prime radix, roots of unity read from the transform's own table: W_R^t = tw[t * rstride]
template <int R, int DIR, class CT> __device__ __forceinline__ void bfly_odd(CT (&v)[R], const Twid<CT>& tw, int rstride) {
  CT w[R];
#pragma unroll
  for (int t = 0; t < R; ++t) {
    w[t] = tw.get((long long)t * rstride);
    if (DIR > 0) w[t].y = -w[t].y;
  }
  bfly_odd_roots<R, DIR, CT>(v, w);
}
template <int R, int DIR, class CT> __device__ __forceinline__ void bfly(CT (&v)[R], const Twid<CT>& tw, int rstride) {
  if constexpr (R == 2) bfly2<DIR, CT>(v[0], v[1]);
  else if constexpr (R == 4) bfly4<DIR, CT>(v);
  else if constexpr (R == 8) bfly8<DIR, CT>(v);
  else bfly_odd<R, DIR, CT>(v, tw, rstride);
}

// LDS index maps of a sequence: plain, or one pad slot per 8 elements (fp64 passes: the first pass stores element 8 j + m from lane j,
// a 128-byte stride = every lane on the same banks; padded, lane j lands 144 bytes on)
struct IdxPlain {
  __device__ static __forceinline__ int at(int i) { return i; }
};
struct IdxPad8 {
  __device__ static __forceinline__ int at(int i) { return i + (i >> 3); }
};

// ---------------------------------------------------------------------------------------------
// one Stockham autosort pass over `nseq` sequences held in LDS (sequence s at base + s*ld)
//   out[(j/Ns)*Ns*R + k + m*Ns] = DFT_R_m( in[j + m'*N/R] * W_{Ns*R}^{k*m'} ),  k = j % Ns
// tw is a table of length tlen = N * tscale with tw[t] = exp(-2 pi i t / tlen)
// ---------------------------------------------------------------------------------------------
template <int R, int DIR, class CT, class IX = IdxPlain>
__device__ void stockham_pass(const CT* __restrict__ in, CT* __restrict__ out, int N, int Ns,
                              const Twid<CT>& tw, int tscale, int nseq, int ld, int tid, int nthr) {
  const int nb = N / R;
  const int twm = (N / (Ns * R)) * tscale;
  const int rstride = nb * tscale;
  // (power-of-two butterfly counts: shifts and masks instead of two integer divisions per butterfly -- a third of a pass's
  //  instructions on the fp64 path, whose butterflies are one per thread and pass)
  const bool p2 = (nb & (nb - 1)) == 0 && (Ns & (Ns - 1)) == 0;
  const int lnb = 31 - __builtin_clz((unsigned)(nb > 0 ? nb : 1));
  for (int i = tid; i < nseq * nb; i += nthr) {
    const int s = p2 ? (i >> lnb) : i / nb, j = i - s * nb;
    const int k = p2 ? (j & (Ns - 1)) : j % Ns;
    const CT* src = in + s * ld;
    CT* dst = out + s * ld;
    CT v[R];
#pragma unroll
    for (int m = 0; m < R; ++m) v[m] = src[IX::at(j + m * nb)];
    if (Ns > 1) {
#pragma unroll
      for (int m = 1; m < R; ++m) {
        CT w = tw.get((long long)k * m * twm);
        if (DIR > 0) w.y = -w.y;
        v[m] = gmul(v[m], w);
      }
    }
    bfly<R, DIR, CT>(v, tw, rstride);
    const int j0 = (j - k) * R + k;
#pragma unroll
    for (int m = 0; m < R; ++m) dst[IX::at(j0 + m * Ns)] = v[m];
  }
}

// any radix: one output per work item, O(R) each
template <int DIR, class CT, class IX = IdxPlain>
__device__ void stockham_pass_any(const CT* __restrict__ in, CT* __restrict__ out, int N, int Ns, int R,
                                  const Twid<CT>& tw, int tscale, int nseq, int ld, int tid, int nthr) {
  const int nb = N / R;
  const long long e1 = N / (Ns * R), e2 = nb;
  for (int i = tid; i < nseq * N; i += nthr) {
    const int s = i / N, r = i - s * N;
    const int q = r / nb, j = r - q * nb;
    const int k = j % Ns;
    const CT* src = in + s * ld;
    CT acc = mk<CT>(0, 0);
    for (int m = 0; m < R; ++m) {
      long long e = ((long long)k * m * e1 + (long long)q * m * e2) % N;
      CT w = tw.get(e * tscale);
      if (DIR > 0) w.y = -w.y;
      acc = gadd(acc, gmul(src[IX::at(j + m * nb)], w));
    }
    out[s * ld + IX::at((j - k) * R + k + q * Ns)] = acc;
  }
}

// full transform of nseq LDS-resident sequences, ping-ponging a <-> b; returns the buffer holding
// the result (natural order).  Every thread of the block must call it.
template <int DIR, class CT, class IX = IdxPlain>
__device__ CT* fft_lds(CT* a, CT* b, const Plan1D& plan, const Twid<CT>& tw, int tscale,
                       int nseq, int ld, int tid, int nthr) {
  const int N = plan.n;
  int Ns = 1;
  for (int f = 0; f < plan.nf; ++f) {
    const int R = plan.radix[f];
    switch (R) {
      case 2: stockham_pass<2, DIR, CT, IX>(a, b, N, Ns, tw, tscale, nseq, ld, tid, nthr); break;
      case 3: stockham_pass<3, DIR, CT, IX>(a, b, N, Ns, tw, tscale, nseq, ld, tid, nthr); break;
      case 4: stockham_pass<4, DIR, CT, IX>(a, b, N, Ns, tw, tscale, nseq, ld, tid, nthr); break;
      case 5: stockham_pass<5, DIR, CT, IX>(a, b, N, Ns, tw, tscale, nseq, ld, tid, nthr); break;
      case 7: stockham_pass<7, DIR, CT, IX>(a, b, N, Ns, tw, tscale, nseq, ld, tid, nthr); break;
      case 8: stockham_pass<8, DIR, CT, IX>(a, b, N, Ns, tw, tscale, nseq, ld, tid, nthr); break;
      case 11: stockham_pass<11, DIR, CT, IX>(a, b, N, Ns, tw, tscale, nseq, ld, tid, nthr); break;
      case 13: stockham_pass<13, DIR, CT, IX>(a, b, N, Ns, tw, tscale, nseq, ld, tid, nthr); break;
      default: stockham_pass_any<DIR, CT, IX>(a, b, N, Ns, R, tw, tscale, nseq, ld, tid, nthr); break;
    }
    __syncthreads();
    CT* t = a;
    a = b;
    b = t;
    Ns *= R;
  }
  return a;
}

// ---------------------------------------------------------------------------------------------
// rows: real -> half spectrum   (even W: length-W/2 complex transform of (x[2n], x[2n+1]) pairs)
// ---------------------------------------------------------------------------------------------
template <bool EVEN>
__global__ void k_rows_r2c(const float* __restrict__ x, float2* __restrict__ spec, int W, int nrows, Plan1D plan,
                           const float2* __restrict__ twW, int rpb) {
  HIP_DYNAMIC_SHARED(float2, smem)
  const int M = plan.n, ld = M + 1, Ws = (W + 1) / 2;
  float2* a = smem;
  float2* b = smem + rpb * ld;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int row0 = blockIdx.x * rpb;
  const int nseq = min(rpb, nrows - row0);
  for (int i = tid; i < nseq * M; i += nthr) {
    const int s = i / M, n = i - s * M;
    const float* xr = x + (size_t)(row0 + s) * W;
    a[s * ld + n] = EVEN ? make_float2(xr[2 * n], xr[2 * n + 1]) : make_float2(xr[n], 0.f);
  }
  __syncthreads();
  const Twid<float2> twd{twW, W};
  const float2* z = fft_lds<-1, float2>(a, b, plan, twd, EVEN ? 2 : 1, nseq, ld, tid, nthr);
  for (int i = tid; i < nseq * Ws; i += nthr) {
    const int s = i / Ws, k = i - s * Ws;
    const float2* zs = z + s * ld;
    float2 X;
    if (!EVEN) {
      X = zs[k];
    } else if (k == 0) {
      X = make_float2(zs[0].x + zs[0].y, zs[0].x - zs[0].y);   // (DC, Nyquist) packed
    } else {
      const float2 zk = zs[k], zm = cconj(zs[M - k]);
      const float2 e = cscale(cadd(zk, zm), 0.5f);
      const float2 d = cscale(csub(zk, zm), 0.5f);
      const float2 o = make_float2(d.y, -d.x);                 // -i * d
      X = cadd(e, cmul(o, twW[k]));
    }
    spec[(size_t)(row0 + s) * Ws + k] = X;
  }
}

// rows: half spectrum -> real (unnormalised inverse times `scale`)
template <bool EVEN>
__global__ void k_rows_c2r(const float2* __restrict__ spec, float* __restrict__ y, int W, int nrows, Plan1D plan,
                           const float2* __restrict__ twW, int rpb, float scale) {
  HIP_DYNAMIC_SHARED(float2, smem)
  const int M = plan.n, ld = M + 1, Ws = (W + 1) / 2;
  float2* a = smem;
  float2* b = smem + rpb * ld;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int row0 = blockIdx.x * rpb;
  const int nseq = min(rpb, nrows - row0);
  for (int i = tid; i < nseq * Ws; i += nthr) {
    const int s = i / Ws, k = i - s * Ws;
    b[s * ld + k] = spec[(size_t)(row0 + s) * Ws + k];
  }
  __syncthreads();
  if (EVEN) {
    for (int i = tid; i < nseq * M; i += nthr) {
      const int s = i / M, k = i - s * M;
      const float2* X = b + s * ld;
      float2 zp;
      if (k == 0) {
        zp = make_float2(X[0].x + X[0].y, X[0].x - X[0].y);
      } else {
        const float2 xk = X[k], xm = cconj(X[M - k]);
        const float2 e = cadd(xk, xm);
        const float2 d = cmulc(csub(xk, xm), twW[k]);          // * w^{-k}
        zp = make_float2(e.x - d.y, e.y + d.x);                // e + i d
      }
      a[s * ld + k] = zp;
    }
  } else {
    for (int i = tid; i < nseq * W; i += nthr) {
      const int s = i / W, k = i - s * W;
      const float2* X = b + s * ld;
      a[s * ld + k] = (k < Ws) ? X[k] : cconj(X[W - k]);
    }
  }
  __syncthreads();
  const Twid<float2> twd{twW, W};
  const float2* z = fft_lds<+1, float2>(a, b, plan, twd, EVEN ? 2 : 1, nseq, ld, tid, nthr);
  for (int i = tid; i < nseq * M; i += nthr) {
    const int s = i / M, n = i - s * M;
    float* yr = y + (size_t)(row0 + s) * W;
    const float2 v = z[s * ld + n];
    if (EVEN) {
      yr[2 * n] = v.x * scale;
      yr[2 * n + 1] = v.y * scale;
    } else {
      yr[n] = v.x * scale;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// the per-frequency operator
// ---------------------------------------------------------------------------------------------
template <int OP>
__device__ __forceinline__ float2 spec_op(float2 z, const SpecArgs& A, size_t tix, float rho_b) {
  if constexpr (OP == OP_MUL) {
    return cscale(cmul(z, A.otf[tix]), A.scale);
  } else if constexpr (OP == OP_MULCONJ) {
    return cscale(cmulc(z, A.otf[tix]), A.scale);
  } else {
    const float2 dd = A.dd[tix];
    const float den = fmaf(rho_b, dd.y, dd.x) + A.eps;
    const float inv = A.scale / den;
    return make_float2((z.x + A.eps_num) * inv, z.y * inv);
  }
}

// the same operator with its table value fetched separately (k_cols_il requests a batch of them before the arithmetic)
template <int OP> __device__ __forceinline__ float2 spec_table(const SpecArgs& A, size_t tix) {
  if constexpr (OP == OP_SOLVE) return A.dd[tix];
  else return A.otf[tix];
}
template <int OP> __device__ __forceinline__ float2 spec_op_t(float2 z, const SpecArgs& A, float2 tv, float rho_b) {
  if constexpr (OP == OP_MUL) {
    return cscale(cmul(z, tv), A.scale);
  } else if constexpr (OP == OP_MULCONJ) {
    return cscale(cmulc(z, tv), A.scale);
  } else {
    const float den = fmaf(rho_b, tv.y, tv.x) + A.eps;
    const float inv = A.scale / den;
    return make_float2((z.x + A.eps_num) * inv, z.y * inv);
  }
}

// columns: forward c2c, operator, inverse c2c, for a tile of CT columns of one plane
template <int OP>
__global__ void k_cols(float2* __restrict__ spec, SpecArgs A, int C, int H, int W, Plan1D plan,
                       const float2* __restrict__ twH, int CT) {
  HIP_DYNAMIC_SHARED(float2, smem)
  const int Ws = (W + 1) / 2, ld = H + 1;
  const bool packed = (W % 2 == 0);
  float2* a = smem;
  float2* b = smem + CT * ld;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int p = blockIdx.y, l0 = blockIdx.x * CT;
  const int nseq = min(CT, Ws - l0);
  const int ch = p % C, bi = p / C;
  float2* base = spec + (size_t)p * H * Ws;
  for (int i = tid; i < H * nseq; i += nthr) {
    const int r = i / nseq, c = i - r * nseq;
    a[c * ld + r] = base[(size_t)r * Ws + l0 + c];
  }
  __syncthreads();
  const Twid<float2> twd{twH, H};
  float2* z = fft_lds<-1, float2>(a, b, plan, twd, 1, nseq, ld, tid, nthr);
  float2* other = (z == a) ? b : a;
  const float rho_b = (OP == OP_SOLVE && A.rho) ? A.rho[bi] : 0.f;
  const size_t tmain = (size_t)ch * H * Ws;
  const size_t tside = (size_t)C * H * Ws + (size_t)ch * H;
  if (OP == OP_SOLVE && A.add) {                      // data spectrum F(K^T b), accumulated in the Fourier domain
    const float2* add = A.add + (size_t)p * H * Ws;
    for (int i = tid; i < H * nseq; i += nthr) {
      const int k = i / nseq, c = i - k * nseq;
      z[c * ld + k] = cadd(z[c * ld + k], add[(size_t)k * Ws + l0 + c]);
    }
    __syncthreads();
  }
  for (int i = tid; i < H * nseq; i += nthr) {
    const int k = i / nseq, c = i - k * nseq;
    if (packed && l0 + c == 0) continue;
    z[c * ld + k] = spec_op<OP>(z[c * ld + k], A, tmain + (size_t)k * Ws + l0 + c, rho_b);
  }
  if (packed && l0 == 0) {
    // column 0 holds DC + i*Nyquist of real-valued columns: separate by Hermitian symmetry
    for (int k = tid; k <= H / 2; k += nthr) {
      const int k2 = (H - k) % H;
      const float2 zk = z[k], zm = cconj(z[k2]);
      const float2 Ak = cscale(cadd(zk, zm), 0.5f);
      const float2 d = cscale(csub(zk, zm), 0.5f);
      const float2 Bk = make_float2(d.y, -d.x);                       // d / i
      const float2 A1 = spec_op<OP>(Ak, A, tmain + (size_t)k * Ws, rho_b);
      const float2 B1 = spec_op<OP>(Bk, A, tside + k, rho_b);
      const float2 A2 = spec_op<OP>(cconj(Ak), A, tmain + (size_t)k2 * Ws, rho_b);
      const float2 B2 = spec_op<OP>(cconj(Bk), A, tside + k2, rho_b);
      other[k] = make_float2(A1.x - B1.y, A1.y + B1.x);               // A' + i B'
      other[k2] = make_float2(A2.x - B2.y, A2.y + B2.x);
    }
    __syncthreads();
    for (int k = tid; k < H; k += nthr) z[k] = other[k];
  }
  __syncthreads();
  const float2* r = fft_lds<+1, float2>(z, other, plan, twd, 1, nseq, ld, tid, nthr);
  for (int i = tid; i < H * nseq; i += nthr) {
    const int row = i / nseq, c = i - row * nseq;
    base[(size_t)row * Ws + l0 + c] = r[c * ld + row];
  }
}

// ---------------------------------------------------------------------------------------------
// Size-generic transforms, second form (round 4): CT sequences INTERLEAVED in one LDS image, passes in place.
//
//   element n of sequence c lives at slot n * LD + c  (LD = CT + 1: the pad keeps strided butterfly outputs off one bank group; CT = 1: LD = 1)
//
// The kernels above give every sequence its own LDS line and ping-pong between two buffers: with 2 * (H + 1) * 8 bytes per column a
// workgroup of the column pass holds 3 columns at H = 1000 -- 24-byte pieces of every 4 KB spectrum row, i.e. a third of each 64-byte
// request used -- and every butterfly pays two integer divisions to find its sequence and position.  Here
//   * the COLUMN pass gives a workgroup of 512 threads CT = 8 columns (64-byte pieces, what the column kernel of the power-of-two planes
//     moves); a work item is (butterfly j, column c) with c = item % CT, so the index arithmetic is shifts and one reciprocal multiply and
//     the twiddles of a butterfly are fetched once per 8 lanes (same address: broadcast);
//   * the ROW passes give every row to a ONE-WAVE workgroup (CT = 1, 64 threads): no workgroup barrier at all, ~20 independent waves per
//     CU in different phases, so one wave's loads and stores overlap the others' passes (eight rows interleaved on 512 threads: 95 us per
//     pass at 8 x 3 x 1000 x 1000, four on 256: 76, two on 128: 68, one on 64: 60);
//   * a pass reads all its operands into registers, synchronises and writes them back to the SAME buffer (at most 16 + R values per
//     thread): half the LDS, two column workgroups per CU up to H = 1100;
//   * the pass twiddles are requested together with the operands in front of the barrier, and every loop of run-time length issues its
//     global loads in batches (written one element at a time, each iteration waits for its own load);
//   * odd radices use the half-size butterfly (bfly_odd_roots: 36 instead of 120 instructions at radix 5).
// Same Stockham order, same butterflies, same table twiddles as stockham_pass: results are bit-identical to the first form's
// (knob generic_interleaved = 0 keeps the first form; pinned in tests/parity_cases.py case_generic_interleaved).
// Radices 2, 3, 4, 5, 7, 8, 11; lengths with another prime factor, or beyond 16 * 512 / CT elements per sequence, stay on the first form.
// Measured (8 x 3 x 1000 x 1000, ADMM TV iteration): k_cols 363 -> 166 us, k_rows_r2c / c2r 165 / 155 -> 60 / 61 us, the iteration 0.91 ->
// 0.52 ms.  Where the column pass's 166 us go (probes, same launch): load + operator + store without any pass 120 us (384 MB incl. the
// denominators: 3.2 TB/s through 64-byte pieces at a 4008-byte stride), the passes alone 79 us; a wave-per-column form (no workgroup
// barrier inside the passes) and a staggered start of the two workgroups of a CU were measured and change nothing; non-temporal
// loads of the spectrum and data-spectrum values: 150 -> 166 us.
// ---------------------------------------------------------------------------------------------
constexpr int IL_NT = 512;        // threads per workgroup
constexpr int IL_MAXE = 16;       // sequence elements per thread: N * CT <= IL_MAXE * IL_NT
constexpr int IL_UB = 8;          // global loads in flight per thread in the load / operator phases

struct BlockSyncAll {
  __device__ __forceinline__ void operator()() const { __syncthreads(); }
};

template <int R, int DIR> __device__ __forceinline__ void bfly_roots(float2 (&v)[R], const float2 (&w)[R]) {
  if constexpr (R == 2) bfly2<DIR, float2>(v[0], v[1]);
  else if constexpr (R == 4) bfly4<DIR, float2>(v);
  else if constexpr (R == 8) bfly8<DIR, float2>(v);
  else bfly_odd_roots<R, DIR, float2>(v, w);             // (bfly_odd with the roots of unity fetched once per pass)
}

template <int R, int DIR, int CT, int NT, class Sync = BlockSyncAll>
__device__ __forceinline__ void il_pass(float2* __restrict__ a, int N, int Ns, const float2* __restrict__ tw, int tscale, int tid) {
  constexpr int LD = CT > 1 ? CT + 1 : 1;
  constexpr int NIT = (IL_MAXE + R - 1) / R;
  DPX_OPAQUE(tid);      // (every pass derives its own index registers: hoisted out of the pass loop, those of all eight radices stay alive -- 256 VGPRs and spills)
  const int nb = N / R, total = nb * CT;
  const int twm = (N / (Ns * R)) * tscale, rstride = nb * tscale;
  const float rcp_ns = 1.0f / (float)Ns;
  float2 w[R];
  if constexpr (R != 2 && R != 4 && R != 8) {
#pragma unroll
    for (int t = 0; t < R; ++t) {
      w[t] = tw[t * rstride];
      if (DIR > 0) w[t].y = -w[t].y;
    }
  }
  // the pass twiddles W_{Ns R}^{k m} are requested together with the operands, in front of the barrier (their L1 / L2 latency hides
  // behind it); radix 11 fetches them behind it (20 more registers than the 128 of four waves per SIMD hold)
  constexpr bool EARLY = R <= 8;
  constexpr int NTW = EARLY ? R - 1 : 1;
  float2 v[NIT][R], pw[NIT][NTW];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = tid + it * NT;
    if (i < total) {
      const int c = i % CT, j = i / CT;
#pragma unroll
      for (int m = 0; m < R; ++m) v[it][m] = a[(j + m * nb) * LD + c];
      if (EARLY && Ns > 1) {
        const int k = j - __mul24(Ns, (int)(((float)j + 0.5f) * rcp_ns));
        const int kt = __mul24(k, twm);                       // (the index of W^{k m}: m kt by additions; index products fit 24 bits -- a full 32-bit multiply costs four additions)
        int idx = kt;
#pragma unroll
        for (int m = 1; m < R; ++m) {
          pw[it][EARLY ? m - 1 : 0] = tw[idx];
          idx += kt;
        }
      }
    }
  }
  Sync()();
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = tid + it * NT;
    if (i < total) {
      const int c = i % CT, j = i / CT;
      // k = j % Ns: j < 2^13 and q * Ns <= j, so (j + 0.5) / Ns is at least 0.5 / Ns >= q * 2^-13 away from an integer -- far above
      // the product's rounding error (q * 2^-23)
      const int k = j - __mul24(Ns, (int)(((float)j + 0.5f) * rcp_ns));
      if (Ns > 1) {
        const int kt = __mul24(k, twm);
#pragma unroll
        for (int m = 1; m < R; ++m) {
          float2 t = EARLY ? pw[it][EARLY ? m - 1 : 0] : tw[m * kt];
          if (DIR > 0) t.y = -t.y;
          v[it][m] = gmul(v[it][m], t);
        }
      }
      bfly_roots<R, DIR>(v[it], w);
      const int j0 = __mul24(j - k, R) + k;
#pragma unroll
      for (int m = 0; m < R; ++m) a[(j0 + m * Ns) * LD + c] = v[it][m];
    }
  }
  Sync()();
}

// Tuning aid (-DDPX_PAR_TRACE builds only): phase stamps of k_rows_c2r_il's one-wave workgroups (tools/ilrow_trace.py)
#ifdef DPX_PAR_TRACE
__device__ unsigned long long dpx_ilrow_trace_buf[4096 * 8];
#define DPX_ROWSTAMP(i)                                                                                             \
  do {                                                                                                            \
    if (threadIdx.x == 0 && blockIdx.x < 4096) dpx_ilrow_trace_buf[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
}  // namespace dpx
extern "C" int dpx_dbg_ilrow_trace(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(dpx::dpx_ilrow_trace_buf), (size_t)n * sizeof(unsigned long long));
}
namespace dpx {
#else
#define DPX_ROWSTAMP(i) ((void)0)
#endif
// in-place transform of the CT interleaved sequences; the caller synchronises in front, the last pass behind
template <int DIR, int CT, int NT, class Sync = BlockSyncAll>
__device__ __forceinline__ void fft_il(float2* __restrict__ a, const Plan1D& plan, const float2* __restrict__ tw, int tscale, int tid) {
  const int N = plan.n;
  int Ns = 1;
  for (int f = 0; f < plan.nf; ++f) {
    const int R = plan.radix[f];
    switch (R) {
      case 2: il_pass<2, DIR, CT, NT, Sync>(a, N, Ns, tw, tscale, tid); break;
      case 3: il_pass<3, DIR, CT, NT, Sync>(a, N, Ns, tw, tscale, tid); break;
      case 4: il_pass<4, DIR, CT, NT, Sync>(a, N, Ns, tw, tscale, tid); break;
      case 5: il_pass<5, DIR, CT, NT, Sync>(a, N, Ns, tw, tscale, tid); break;
      case 7: il_pass<7, DIR, CT, NT, Sync>(a, N, Ns, tw, tscale, tid); break;
      case 8: il_pass<8, DIR, CT, NT, Sync>(a, N, Ns, tw, tscale, tid); break;
      default: il_pass<11, DIR, CT, NT, Sync>(a, N, Ns, tw, tscale, tid); break;
    }
    Ns *= R;
#ifdef DPX_PAR_TRACE
    if (NT == 64 && DIR > 0 && f < 4) DPX_ROWSTAMP(2 + f);
#endif
  }
}

// sequences per workgroup for a length (0: the length stays on the first form)
static int il_seqs(const Plan1D& plan) {
  for (int f = 0; f < plan.nf; ++f) {
    const int r = plan.radix[f];
    if (!(r == 2 || r == 3 || r == 4 || r == 5 || r == 7 || r == 8 || r == 11)) return 0;      // (radix 13 in this form: 169 VGPRs)
  }
  if (plan.n < 2) return 0;
  if (plan.n * 8 <= IL_MAXE * IL_NT) return 8;
  if (plan.n * 4 <= IL_MAXE * IL_NT) return 4;
  return 0;
}

// Where a one-wave row transform's time goes (tools/ilrow_trace.py, 8 x 3 x 1000 x 1000: 500 complex points, passes 4 x 5 x 5 x 5; a workgroup lives 12 us,
// 16 of them per CU): row loaded + untangled 2.6 us, first pass 1.5, the three passes with twiddles 2.0 each, row stored 1.5.  A pass is ~450 vector
// instructions per wave (60 % of them integer / move), but it is not issue-bound: 12 % fewer issue cycles (twiddle indices by additions, 24-bit index
// products) bought 1 - 4 %, compile-time lengths and strides 4 % on the column passes (not adopted) -- a pass is a chain of shared-memory round trips, a
// twiddle gather and two barriers, and the waves of a CU walk it in step.
// (the loops of run-time length below issue their global loads in batches of UB: written one element at a time, each iteration waits for
//  its own load -- M / NT dependent round trips per phase, which is what a one-wave workgroup's row kernel then consists of)
// TWL (both row kernels, as in k_cols_il): the M twiddles of the transform in shared memory behind the rows
template <bool EVEN, int CT, int NT, bool TWL>
__global__ void __launch_bounds__(NT, 4) k_rows_r2c_il(const float* __restrict__ x, float2* __restrict__ spec, int W, int nrows, Plan1D plan,
                                                       const float2* __restrict__ twW) {
  HIP_DYNAMIC_SHARED(float2, smem)
  constexpr int LD = CT > 1 ? CT + 1 : 1;
  constexpr int UB = CT >= 4 ? 2 : 8;
  const int M = plan.n, Ws = (W + 1) / 2;
  float2* a = smem;
  const int tid = threadIdx.x;
  const float2* twp = twW;
  int tsc = EVEN ? 2 : 1;
  if constexpr (TWL) {
    float2* tl = smem + (size_t)M * LD;
    for (int i = tid; i < M; i += NT) tl[i] = twW[i * tsc];
    twp = tl;
    tsc = 1;
  }
  const int row0 = blockIdx.x * CT;
  const int nseq = min(CT, nrows - row0);
#pragma unroll
  for (int s = 0; s < CT; ++s) {
    const float* xr = x + (size_t)(row0 + s) * W;
    for (int n0 = tid; n0 < M; n0 += UB * NT) {
      float2 v[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int n = n0 + u * NT;
        v[u] = make_float2(0.f, 0.f);
        if (s < nseq && n < M) v[u] = EVEN ? *(const float2*)(xr + 2 * n) : make_float2(xr[n], 0.f);
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int n = n0 + u * NT;
        if (n < M) a[n * LD + s] = v[u];
      }
    }
  }
  __syncthreads();
  fft_il<-1, CT, NT>(a, plan, twp, tsc, tid);
#pragma unroll
  for (int s = 0; s < CT; ++s) {
    if (s >= nseq) break;
    float2* out = spec + (size_t)(row0 + s) * Ws;
    for (int k0 = tid; k0 < Ws; k0 += UB * NT) {
      float2 tw[UB];
      if (EVEN) {
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          const int k = k0 + u * NT;
          tw[u] = k < Ws ? twW[k] : make_float2(0.f, 0.f);
        }
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int k = k0 + u * NT;
        if (k >= Ws) continue;
        float2 X;
        if (!EVEN) {
          X = a[k * LD + s];
        } else if (k == 0) {
          const float2 z0 = a[s];
          X = make_float2(z0.x + z0.y, z0.x - z0.y);               // (DC, Nyquist) packed
        } else {
          const float2 zk = a[k * LD + s], zm = cconj(a[(M - k) * LD + s]);
          const float2 e = cscale(cadd(zk, zm), 0.5f);
          const float2 d = cscale(csub(zk, zm), 0.5f);
          const float2 o = make_float2(d.y, -d.x);                 // -i * d
          X = cadd(e, cmul(o, tw[u]));
        }
        out[k] = X;
      }
    }
  }
}

template <bool EVEN, int CT, int NT, bool TWL>
__global__ void __launch_bounds__(NT, 4) k_rows_c2r_il(const float2* __restrict__ spec, float* __restrict__ y, int W, int nrows, Plan1D plan,
                                                       const float2* __restrict__ twW, float scale) {
  HIP_DYNAMIC_SHARED(float2, smem)
  constexpr int LD = CT > 1 ? CT + 1 : 1;
  constexpr int UB = CT >= 4 ? 1 : 4;
  const int M = plan.n, Ws = (W + 1) / 2;
  float2* a = smem;
  const int tid = threadIdx.x;
  const float2* twp = twW;
  int tsc = EVEN ? 2 : 1;
  if constexpr (TWL) {
    float2* tl = smem + (size_t)M * LD;
    for (int i = tid; i < M; i += NT) tl[i] = twW[i * tsc];
    twp = tl;
    tsc = 1;
  }
  DPX_ROWSTAMP(0);
  const int row0 = blockIdx.x * CT;
  const int nseq = min(CT, nrows - row0);
  // half spectrum -> the transform's input, pair (k, M - k) by one thread (in place)
#pragma unroll
  for (int s = 0; s < CT; ++s) {
    const float2* X = spec + (size_t)(row0 + s) * Ws;
    if (EVEN) {
      for (int k0 = tid; k0 <= M / 2; k0 += UB * NT) {
        float2 xa[UB], xb[UB], wa[UB], wb[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          const int k = k0 + u * NT, k2 = M - k;
          xa[u] = xb[u] = wa[u] = wb[u] = make_float2(0.f, 0.f);
          if (s < nseq && k <= M / 2) {
            xa[u] = X[k];
            if (k != 0) {
              xb[u] = X[k2];
              wa[u] = twW[k];
              wb[u] = twW[k2];
            }
          }
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          const int k = k0 + u * NT, k2 = M - k;
          if (k > M / 2) continue;
          float2 p0 = make_float2(0.f, 0.f), p1 = p0;
          if (s < nseq) {
            if (k == 0) {
              const float2 x0 = xa[u];
              p0 = make_float2(x0.x + x0.y, x0.x - x0.y);
            } else {
              {
                const float2 xm = cconj(xb[u]);
                const float2 e = cadd(xa[u], xm);
                const float2 d = cmulc(csub(xa[u], xm), wa[u]);        // * w^{-k}
                p0 = make_float2(e.x - d.y, e.y + d.x);              // e + i d
              }
              {
                const float2 xm = cconj(xa[u]);
                const float2 e = cadd(xb[u], xm);
                const float2 d = cmulc(csub(xb[u], xm), wb[u]);
                p1 = make_float2(e.x - d.y, e.y + d.x);
              }
            }
          }
          a[k * LD + s] = p0;
          if (k != 0 && k2 != k) a[k2 * LD + s] = p1;
        }
      }
    } else {
      for (int k0 = tid; k0 < Ws; k0 += UB * NT) {
        float2 v[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          const int k = k0 + u * NT;
          v[u] = make_float2(0.f, 0.f);
          if (s < nseq && k < Ws) v[u] = X[k];
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          const int k = k0 + u * NT;
          if (k >= Ws) continue;
          a[k * LD + s] = v[u];
          if (k > 0) a[(W - k) * LD + s] = cconj(v[u]);
        }
      }
    }
  }
  __syncthreads();
  DPX_ROWSTAMP(1);
  fft_il<+1, CT, NT>(a, plan, twp, tsc, tid);
  DPX_ROWSTAMP(6);
#pragma unroll
  for (int s = 0; s < CT; ++s) {
    if (s >= nseq) break;
    float* yr = y + (size_t)(row0 + s) * W;
    for (int n = tid; n < M; n += NT) {
      const float2 v = a[n * LD + s];
      if (EVEN) *(float2*)(yr + 2 * n) = make_float2(v.x * scale, v.y * scale);
      else yr[n] = v.x * scale;
    }
  }
  DPX_ROWSTAMP(7);
}

// Tuning aid (tools/build_variant.sh par_trace -DDPX_PAR_TRACE; never in the shipped library): thread 0 of the first 2048 workgroups of k_cols_il
// stamps the 100 MHz real-time counter at the phase boundaries; tools/il_trace.py reads the stamps of the last launch.
#ifdef DPX_PAR_TRACE
__device__ unsigned long long dpx_il_trace_buf[2048 * 8];
#define DPX_ILSTAMP(i)                                                                                             \
  do {                                                                                                            \
    if (threadIdx.x == 0 && blockIdx.x < 2048) dpx_il_trace_buf[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
}  // namespace dpx
extern "C" int dpx_dbg_il_trace(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(dpx::dpx_il_trace_buf), (size_t)n * sizeof(unsigned long long));
}
namespace dpx {
#else
#define DPX_ILSTAMP(i) ((void)0)
#endif

// columns: forward c2c, operator, inverse c2c for CT adjacent columns of one plane (the arithmetic of k_cols)
// TWL: the H twiddles of the column length sit in shared memory behind the tile (copied with the tile's loads): a pass of the size-generic
// transform gathers R - 1 of them per butterfly -- from the global table that is 14 scattered 8-byte loads per thread and pass, each lane of a
// wave in another cache line, and the CU's vector L1 looks lines up one at a time: the in-kernel timeline (tools/il_trace.py) had the four
// forward passes of a 1000-point column tile at 15 - 17 us of a workgroup's 42 - 47.
template <int OP, int CT, int NT, bool TWL>
__global__ void __launch_bounds__(NT, 4) k_cols_il(float2* __restrict__ spec, SpecArgs A, int C, int H, int W, Plan1D plan,
                                                   const float2* __restrict__ twH, int P) {
  HIP_DYNAMIC_SHARED(float2, smem)
  constexpr int LD = CT > 1 ? CT + 1 : 1;
  const int Ws = (W + 1) / 2;
  const bool packed = (W % 2 == 0);
  float2* a = smem;
  const int tid = threadIdx.x;
  // workgroup -> (image, channel, column tile).  The hardware deals workgroup i to XCD i % 8; the work is dealt so that ONE XCD takes
  //   * the B images of a (channel, tile) in consecutive slots -- they read the same 64 KB of denominators: from that XCD's L2 for all
  //     but the first -- and
  //   * IL_KB = 8 ADJACENT tiles (64 columns: 512 bytes of every spectrum row) one behind the other: a tile's 64-byte pieces start at
  //     32 r mod 128 in row r (a row is 8 Ws bytes, no multiple of the 128-byte line), so each piece shares lines with its neighbours;
  //     with the neighbours behind eight different L2s every line crossed the fabric twice or more (FETCH_SIZE: 2.5 x the
  //     algorithmic read at 8 x 3 x 1000 x 1000) -- behind one L2 the second toucher hits.
  // unit u = (channel, block of IL_KB tiles) -> XCD u % 8; a launch is 8 x the longest XCD list, the padding workgroups leave at once.
  constexpr int IL_KB = 8;
  const int ntile = (Ws + CT - 1) / CT, nb = P / C;
  int p, tile;
  {
    const int nblk = (ntile + IL_KB - 1) / IL_KB, U = C * nblk;
    int u = blockIdx.x % 8, item = blockIdx.x / 8, cnt = 0;
    for (; u < U; u += 8) {
      const int tb = u % nblk;
      cnt = min(IL_KB, ntile - tb * IL_KB) * nb;
      if (item < cnt) break;
      item -= cnt;
    }
    if (u >= U) return;
    DPX_ILSTAMP(0);
    tile = (u % nblk) * IL_KB + item / nb;
    p = (item % nb) * C + u / nblk;
  }
  const int l0 = tile * CT;
  const int nseq = min(CT, Ws - l0);
  const int ch = p % C, bi = p / C;
  float2* base = spec + (size_t)p * H * Ws;
  const int c = tid % CT;
  const bool live = c < nseq;
  // (global loads in batches of IL_UB: a loop of run-time length keeps ONE load in flight per thread -- 16 dependent round trips per phase)
  constexpr int RS = NT / CT;
  for (int r0 = tid / CT; r0 < H; r0 += IL_UB * RS) {
    float2 t[IL_UB];
#pragma unroll
    for (int u = 0; u < IL_UB; ++u) {
      const int r = r0 + u * RS;
      t[u] = (live && r < H) ? base[(size_t)r * Ws + l0 + c] : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < IL_UB; ++u) {
      const int r = r0 + u * RS;
      if (r < H) a[r * LD + c] = t[u];
    }
  }
  const float2* twp = twH;
  if constexpr (TWL) {
    float2* tl = smem + (size_t)H * LD;
    for (int i = tid; i < H; i += NT) tl[i] = twH[i];
    twp = tl;
  }
  __syncthreads();
  DPX_ILSTAMP(1);
  fft_il<-1, CT, NT>(a, plan, twp, 1, tid);
  DPX_ILSTAMP(2);
  const float rho_b = (OP == OP_SOLVE && A.rho) ? A.rho[bi] : 0.f;
  const size_t tmain = (size_t)ch * H * Ws;
  const size_t tside = (size_t)C * H * Ws + (size_t)ch * H;
  const float2* add = (OP == OP_SOLVE && A.add) ? A.add + (size_t)p * H * Ws : nullptr;   // data spectrum F(K^T b), accumulated in the Fourier domain
  const bool pk0 = packed && l0 + c == 0;
  if (live)
    for (int k0 = tid / CT; k0 < H; k0 += IL_UB * RS) {
      float2 z[IL_UB], ad[IL_UB], tb[IL_UB];
#pragma unroll
      for (int u = 0; u < IL_UB; ++u) {
        const int k = k0 + u * RS;
        ad[u] = tb[u] = make_float2(0.f, 0.f);
        if (k < H) {
          if (add) ad[u] = add[(size_t)k * Ws + l0 + c];
          if (!pk0) tb[u] = spec_table<OP>(A, tmain + (size_t)k * Ws + l0 + c);
          z[u] = a[k * LD + c];
        }
      }
#pragma unroll
      for (int u = 0; u < IL_UB; ++u) {
        const int k = k0 + u * RS;
        if (k < H) {
          float2 zz = z[u];
          if (add) zz = cadd(zz, ad[u]);
          if (!pk0) zz = spec_op_t<OP>(zz, A, tb[u], rho_b);
          a[k * LD + c] = zz;
        }
      }
    }
  if (packed && l0 == 0) {
    // column 0 holds DC + i*Nyquist of real-valued columns: separate by Hermitian symmetry; the pair (k, H - k) by one thread, in place
    __syncthreads();
    for (int k = tid; k <= H / 2; k += NT) {
      const int k2 = (H - k) % H;
      const float2 zk = a[k * LD], zm = cconj(a[k2 * LD]);
      const float2 Ak = cscale(cadd(zk, zm), 0.5f);
      const float2 d = cscale(csub(zk, zm), 0.5f);
      const float2 Bk = make_float2(d.y, -d.x);                       // d / i
      const float2 A1 = spec_op<OP>(Ak, A, tmain + (size_t)k * Ws, rho_b);
      const float2 B1 = spec_op<OP>(Bk, A, tside + k, rho_b);
      const float2 A2 = spec_op<OP>(cconj(Ak), A, tmain + (size_t)k2 * Ws, rho_b);
      const float2 B2 = spec_op<OP>(cconj(Bk), A, tside + k2, rho_b);
      a[k * LD] = make_float2(A1.x - B1.y, A1.y + B1.x);               // A' + i B'
      a[k2 * LD] = make_float2(A2.x - B2.y, A2.y + B2.x);
    }
  }
  __syncthreads();
  DPX_ILSTAMP(3);
  fft_il<+1, CT, NT>(a, plan, twp, 1, tid);
  DPX_ILSTAMP(4);
  if (live)
    for (int r0 = tid / CT; r0 < H; r0 += IL_UB * RS) {
      float2 t[IL_UB];
#pragma unroll
      for (int u = 0; u < IL_UB; ++u) {
        const int r = r0 + u * RS;
        if (r < H) t[u] = a[r * LD + c];
      }
#pragma unroll
      for (int u = 0; u < IL_UB; ++u) {
        const int r = r0 + u * RS;
        if (r < H) base[(size_t)r * Ws + l0 + c] = t[u];
      }
    }
  DPX_ILSTAMP(5);
}

// ---------------------------------------------------------------------------------------------
// one-off fp64 forward transform of the data term:  spec = op(OTF) * F(b)   (packed fp32 half spectrum)
// Accumulating F(K^T b) in the Fourier domain keeps the large, iteration-invariant part of the
// right-hand side out of the per-iteration fp32 transforms (only the small increment
// rho * sum K_i^T (v_i - u_i) goes through them), which is what holds the iterates within 1e-5 of the
// reference's although the x-update amplifies transform round-off by up to 1/min(denominator).
// ---------------------------------------------------------------------------------------------
// The fp64 twiddles come from a table (k_twiddle_table_f64, once per call, in the workspace): evaluated on the fly, every
// sincospi's argument reduction went through scratch memory (4.0 GB of scratch writes for a 100 MB image at 8x3x1024^2).
// Intermediate fp64 half spectrum: COLUMN-TILE-MAJOR [P][ceil(Wh/CT)][H][CT] (Wh = W/2+1 columns, CT = 4 or 2): the row kernel
// writes RPB adjacent rows x CT columns = one contiguous RPB * CT * 16-byte piece per tile, the column kernel reads its tile as
// ONE linear block (the row-major intermediate cost it one 16-byte element per 8 KB stride: 16x the algorithmic traffic).
__global__ void k_twiddle_table_f64(double2* tw, int n) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  double s, c;
  sincospi(-2.0 * (double)t / (double)n, &s, &c);
  tw[t] = make_double2(c, s);
}

// rows: RPB image rows per workgroup; even W: length-W/2 complex transform of the (x[2n], x[2n+1]) pairs + real-input untangling
template <bool EVEN>
__global__ void __launch_bounds__(512) k_rows_r2c_f64(const float* __restrict__ x, double2* __restrict__ spec, int W, int nrows, int H, Plan1D plan,
                               const double2* __restrict__ tw64, int rpb, int CT) {
  HIP_DYNAMIC_SHARED(double2, smem64)
  const int M = plan.n, ld = M + M / 8 + 1, Wh = W / 2 + 1, NTL = (Wh + CT - 1) / CT;
  double2* a = smem64;
  double2* b = smem64 + rpb * ld;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int row0 = blockIdx.x * rpb;
  const int nseq = min(rpb, nrows - row0);
  for (int i = tid; i < nseq * M; i += nthr) {
    const int s = i / M, n = i - s * M;
    const float* xr = x + (size_t)(row0 + s) * W;
    if (EVEN) {
      const float2 v = ((const float2*)xr)[n];
      a[s * ld + IdxPad8::at(n)] = make_double2((double)v.x, (double)v.y);
    } else {
      a[s * ld + IdxPad8::at(n)] = make_double2((double)xr[n], 0.0);
    }
  }
  __syncthreads();
  const Twid<double2> twd{tw64, W};
  const double2* z = fft_lds<-1, double2, IdxPad8>(a, b, plan, twd, EVEN ? 2 : 1, nseq, ld, tid, nthr);
  // (tile, row, column-in-tile) order: the rows of a workgroup are adjacent in every tile.  nthr is a multiple of CT * rpb: a thread
  // keeps its (row, column-in-tile) and walks the tiles -- no division in the loop
  const int kk = tid % CT, s = (tid / CT) % rpb, kt0 = tid / (CT * rpb), kstep = nthr / (CT * rpb);
  const int row = row0 + (s < nseq ? s : 0), p = row / H, h = row - p * H;
  const double2* zs = z + s * ld;
  for (int kt = kt0; kt < NTL && s < nseq; kt += kstep) {
    const int k = kt * CT + kk;
    if (k >= Wh) continue;
    double2 X;
    if (!EVEN) {
      X = zs[IdxPad8::at(k)];
    } else if (k == 0) {
      X = make_double2(zs[0].x + zs[0].y, 0.0);
    } else if (k == M) {
      X = make_double2(zs[0].x - zs[0].y, 0.0);
    } else {
      const double2 zk = zs[IdxPad8::at(k)], zmr = zs[IdxPad8::at(M - k)], zm = make_double2(zmr.x, -zmr.y);
      const double2 e = make_double2(0.5 * (zk.x + zm.x), 0.5 * (zk.y + zm.y));
      const double2 d = make_double2(0.5 * (zk.x - zm.x), 0.5 * (zk.y - zm.y));
      const double2 o = make_double2(d.y, -d.x);                 // -i * d
      X = gadd(e, gmul(o, tw64[k]));
    }
    spec[(((size_t)p * NTL + kt) * H + h) * CT + kk] = X;
  }
}

// columns of the fp64 half spectrum, CT per workgroup -> packed fp32 spectrum times the OTF.  Even W: the Nyquist column
// (l = W/2) has to meet the DC column (packed layouts carry it as the imaginary part of column 0), so workgroup 0 takes it in its
// last slot and the workgroup that would have held it takes column CT-1 instead.
__global__ void __launch_bounds__(512) k_cols_fwd_f64(const double2* __restrict__ spec, float2* __restrict__ out, const float2* __restrict__ otf,
                               int conj_otf, int accumulate, int C, int H, int W, Plan1D plan, int side_layout, int P,
                               const double2* __restrict__ tw64, int CT) {
  HIP_DYNAMIC_SHARED(double2, smem64)
  const int Wh = W / 2 + 1, Ws = (W + 1) / 2, ld = H + H / 8 + 1, NTL = (Wh + CT - 1) / CT;
  const bool packed = (W % 2 == 0);
  double2* a = smem64;
  double2* b = smem64 + CT * ld;
  const int tid = threadIdx.x, nthr = blockDim.x;
  // The workgroups whose columns share the 128-byte lines of the fp32 output / OTF table (16 columns = 16 / CT tiles) are given block
  // ids 8 apart inside a group of 8 * 16 / CT: the round-robin block -> XCD assignment puts them behind the same L2 at nearly the
  // same time, so a table line is fetched once instead of four times (measured: 609 MB of reads for 214 MB algorithmic before) and
  // the four 32-byte pieces of an output line meet in that L2.
  const int p = blockIdx.y;
  int lt;
  {
    const int TPL = 16 / CT;                                // tiles per 128-byte line of the fp32 layouts (8 for CT = 2)
    const int id = blockIdx.x, g = id / (8 * TPL), r = id - g * (8 * TPL);
    lt = TPL * (g * 8 + (r & 7)) + (r >> 3);
  }
  if (lt >= NTL) return;                                   // (block-uniform: the grid is padded to a multiple of 8 * TPL)
  const int ch = p % C;
  const int nyq = W / 2, tn = packed ? nyq / CT : 0;
  auto col_of = [&](int c) {                              // spectrum column of slot c of this workgroup (-1: none)
    int l = lt * CT + c;
    if (packed && tn != 0) {
      if (lt == 0 && c == CT - 1) l = nyq;
      else if (lt == tn && c == nyq % CT) l = CT - 1;
    }
    return l < Wh ? l : -1;
  };
  const double2* base = spec + (size_t)p * NTL * H * CT;
  // (nthr is a multiple of CT: a thread keeps its column, rows advance by nthr / CT -- no division in the loops)
  const int c_own = tid % CT, r_own = tid / CT, r_step = nthr / CT;
  const int l_own = col_of(c_own);
  {
    const double2* src = base + ((size_t)(l_own >= 0 ? l_own / CT : 0) * H) * CT + (l_own >= 0 ? l_own % CT : 0);
    double2* dst = a + c_own * ld;
    for (int r = r_own; r < H; r += r_step) dst[IdxPad8::at(r)] = l_own >= 0 ? src[(size_t)r * CT] : make_double2(0.0, 0.0);
  }
  __syncthreads();
  const Twid<double2> twd{tw64, H};
  const double2* z = fft_lds<-1, double2, IdxPad8>(a, b, plan, twd, 1, CT, ld, tid, nthr);
  const size_t tmain = (size_t)ch * H * Ws, tside = (size_t)C * H * Ws + (size_t)ch * H;
  float2* o = out + (size_t)p * H * Ws;
  int nyq_slot = -1;                                        // the slot holding the Nyquist column, if this workgroup has the DC column
  if (packed && lt == 0) nyq_slot = tn != 0 ? CT - 1 : nyq % CT;
  for (int k = r_own; k < H; k += r_step) {
    const int c = c_own, l = l_own;
    if (l < 0 || (packed && l == nyq)) break;              // (the Nyquist column is consumed by the DC column's lanes)
    double2 v = z[c * ld + IdxPad8::at(k)];
    if (otf) {
      const float2 t = otf[tmain + spec_main_index(side_layout, H, Ws, k, l)];
      const double tr = t.x, ti = conj_otf ? -(double)t.y : (double)t.y;
      v = make_double2(v.x * tr - v.y * ti, v.x * ti + v.y * tr);
    }
    if (l == 0 && nyq_slot >= 0) {
      double2 n = z[nyq_slot * ld + IdxPad8::at(k)];
      if (otf) {
        const float2 t = otf[tside + k];
        const double tr = t.x, ti = conj_otf ? -(double)t.y : (double)t.y;
        n = make_double2(n.x * tr - n.y * ti, n.x * ti + n.y * tr);
      }
      if (side_layout && !DPX_COLS_PACK0) {              // Nyquist column -> side array [P][H]
        float2* sd = out + (size_t)P * H * Ws + (size_t)p * H + k;
        if (accumulate) {
          const float2 old = *sd;
          *sd = make_float2((float)((double)old.x + n.x), (float)((double)old.y + n.y));
        } else {
          *sd = make_float2((float)n.x, (float)n.y);
        }
      } else {
        v = make_double2(v.x - n.y, v.y + n.x);        // packed: A + i B (power-of-two planes too, see dpx_common.h)
      }
    }
    float2* dst = o + spec_main_index(side_layout, H, Ws, k, l);
    if (accumulate) {
      const float2 old = *dst;
      *dst = make_float2((float)((double)old.x + v.x), (float)((double)old.y + v.y));
    } else {
      *dst = make_float2((float)v.x, (float)v.y);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// complex 2-D transform with the centring shifts fused (utils.fft2 / ifft2 of the reference:
// ifftshift -> fft2(norm='ortho') -> fftshift, dprox/utils/misc.py:164-193) -- the building block of
// user-defined CS-MRI operators (mask * fft2(x)).  Any size (same Stockham core as above).
//   centred: input index (i + N/2) % N is read at position i, output bin k is stored at (k + N/2) % N
// ---------------------------------------------------------------------------------------------
template <int DIR>
__global__ void k_crows(const float2* __restrict__ in, float2* __restrict__ out, int W, int nrows, Plan1D plan,
                        const float2* __restrict__ twW, int rpb, int centred, float scale) {
  HIP_DYNAMIC_SHARED(float2, smem)
  const int ld = W + 1, hs = centred ? W / 2 : 0;
  float2* a = smem;
  float2* b = smem + rpb * ld;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int row0 = blockIdx.x * rpb;
  const int nseq = min(rpb, nrows - row0);
  for (int i = tid; i < nseq * W; i += nthr) {
    const int s = i / W, n = i - s * W;
    int src = n + hs;
    if (src >= W) src -= W;
    a[s * ld + n] = in[(size_t)(row0 + s) * W + src];
  }
  __syncthreads();
  const Twid<float2> twd{twW, W};
  const float2* z = fft_lds<DIR, float2>(a, b, plan, twd, 1, nseq, ld, tid, nthr);
  for (int i = tid; i < nseq * W; i += nthr) {
    const int s = i / W, k = i - s * W;
    int dst = k + hs;
    if (dst >= W) dst -= W;
    out[(size_t)(row0 + s) * W + dst] = cscale(z[s * ld + k], scale);
  }
}

template <int DIR>
__global__ void k_ccols(float2* __restrict__ data, int H, int W, Plan1D plan, const float2* __restrict__ twH, int CT, int centred) {
  HIP_DYNAMIC_SHARED(float2, smem)
  const int ld = H + 1, hs = centred ? H / 2 : 0;
  float2* a = smem;
  float2* b = smem + CT * ld;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int p = blockIdx.y, l0 = blockIdx.x * CT;
  const int nseq = min(CT, W - l0);
  float2* base = data + (size_t)p * H * W;
  for (int i = tid; i < H * nseq; i += nthr) {
    const int r = i / nseq, c = i - r * nseq;
    int src = r + hs;
    if (src >= H) src -= H;
    a[c * ld + r] = base[(size_t)src * W + l0 + c];
  }
  __syncthreads();
  const Twid<float2> twd{twH, H};
  const float2* z = fft_lds<DIR, float2>(a, b, plan, twd, 1, nseq, ld, tid, nthr);
  __syncthreads();      // (all loads of this tile happened before the first pass; stores below are in place)
  for (int i = tid; i < H * nseq; i += nthr) {
    const int k = i / nseq, c = i - k * nseq;
    int dst = k + hs;
    if (dst >= H) dst -= H;
    base[(size_t)dst * W + l0 + c] = z[c * ld + k];
  }
}

// ---- the masked-Fourier normal operator  Ap = Re F^-1 (mask^2 . F p) + c rho_b p  in three launches (CG matvec of the CS-MRI data
// term, dpx_cg_masked_fft): the same transforms, index shifts and scale factors as  dpx_cfft2 -> mask^2 -> dpx_cfft2^-1  with the
// real -> complex copy, the mask pass and the real-part / rho pass folded into the neighbouring transform kernels.
//   k_crows_real_in : rows of the real image p -> centred orthonormal row spectra (complex)
//   k_ccols_mask    : forward column transform -> * mask^2 -> inverse column transform, in place (one LDS residency)
//   k_crows_real_out: inverse row transform -> real part + c rho_b p -> Ap
// dir_r != NULL: the CG direction update rides in the load -- p = r + beta_b p (solver_cg.py:111-115) is formed here, written back
// to `in` (= p) and transformed, instead of a k_cg_direction launch in front
__global__ void k_crows_real_in(float* __restrict__ in, float2* __restrict__ out, int W, int nrows, Plan1D plan,
                                const float2* __restrict__ twW, int rpb, float scale, const float* __restrict__ dir_r,
                                const float* __restrict__ beta, const int* __restrict__ done, int rows_per_image) {
  HIP_DYNAMIC_SHARED(float2, smem)
  if (done && done[0]) return;                          // (block-uniform: the solve has converged, this launch ran ahead)
  const int ld = W + 1, hs = W / 2;
  float2* a = smem;
  float2* b = smem + rpb * ld;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int row0 = blockIdx.x * rpb;
  const int nseq = min(rpb, nrows - row0);
  for (int i = tid; i < nseq * W; i += nthr) {
    const int s = i / W, n = i - s * W;
    int src = n + hs;
    if (src >= W) src -= W;
    const size_t e = (size_t)(row0 + s) * W + src;
    float v = in[e];
    if (dir_r) {
      v = fmaf(beta[(row0 + s) / rows_per_image], v, dir_r[e]);
      in[e] = v;
    }
    a[s * ld + n] = make_float2(v, 0.f);
  }
  __syncthreads();
  const Twid<float2> twd{twW, W};
  const float2* z = fft_lds<-1, float2>(a, b, plan, twd, 1, nseq, ld, tid, nthr);
  for (int i = tid; i < nseq * W; i += nthr) {
    const int s = i / W, k = i - s * W;
    int dst = k + hs;
    if (dst >= W) dst -= W;
    out[(size_t)(row0 + s) * W + dst] = cscale(z[s * ld + k], scale);
  }
}

__global__ void k_ccols_mask(float2* __restrict__ data, const float* __restrict__ mask2, int mask_images, int H, int W, Plan1D plan,
                             const float2* __restrict__ twH, int CT, int square, const int* __restrict__ done) {
  HIP_DYNAMIC_SHARED(float2, smem)
  if (done && done[0]) return;
  const int ld = H + 1, hs = H / 2;
  float2* a = smem;
  float2* b = smem + CT * ld;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int p = blockIdx.y, l0 = blockIdx.x * CT;
  const int nseq = min(CT, W - l0);
  float2* base = data + (size_t)p * H * W;
  const float* mk = mask2 + (mask_images == 1 ? (size_t)0 : (size_t)p * H * W);
  for (int i = tid; i < H * nseq; i += nthr) {
    const int r = i / nseq, c = i - r * nseq;
    int src = r + hs;
    if (src >= H) src -= H;
    a[c * ld + r] = base[(size_t)src * W + l0 + c];
  }
  __syncthreads();
  const Twid<float2> twd{twH, H};
  float2* z = fft_lds<-1, float2>(a, b, plan, twd, 1, nseq, ld, tid, nthr);
  // bin k would be stored at centred row (k + hs) % H, which is where the inverse transform reads its input r = k from: the two
  // shifts cancel, only the mask is indexed with the centred row
  for (int i = tid; i < H * nseq; i += nthr) {
    const int k = i / nseq, c = i - k * nseq;
    int row = k + hs;
    if (row >= H) row -= H;
    const float mv = mk[(size_t)row * W + l0 + c];
    z[c * ld + k] = cscale(z[c * ld + k], square ? mv * mv : mv);     // (square: `mask2` holds the mask itself)
  }
  __syncthreads();
  const float2* y = fft_lds<+1, float2>(z, z == a ? b : a, plan, twd, 1, nseq, ld, tid, nthr);
  for (int i = tid; i < H * nseq; i += nthr) {
    const int k = i / nseq, c = i - k * nseq;
    int dst = k + hs;
    if (dst >= H) dst -= H;
    base[(size_t)dst * W + l0 + c] = y[c * ld + k];
  }
}

// dot_partial != NULL (rpb divides rows_per_image): <p_b, A p_b> rides in the store -- every workgroup leaves the sum over its rows,
// the LAST one to arrive adds them up per image in a fixed order and writes pAp_b into the CG state (k_dot_partial + k_dot_finish)
__global__ void k_crows_real_out(const float2* __restrict__ in, float* __restrict__ out, const float* __restrict__ pin, const float* __restrict__ rho,
                                 float c, const int* __restrict__ done, int rows_per_image, int W, int nrows, Plan1D plan,
                                 const float2* __restrict__ twW, int rpb, float scale, float* __restrict__ dot_partial,
                                 unsigned* __restrict__ counter, float* __restrict__ pAp, int B) {
  HIP_DYNAMIC_SHARED(float2, smem)
  __shared__ float shred[16];
  __shared__ int shlast;
  if (done && done[0]) return;                          // (block-uniform: the solve has converged, this launch ran ahead)
  const int ld = W + 1, hs = W / 2;
  float2* a = smem;
  float2* b = smem + rpb * ld;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int row0 = blockIdx.x * rpb;
  const int nseq = min(rpb, nrows - row0);
  for (int i = tid; i < nseq * W; i += nthr) {
    const int s = i / W, n = i - s * W;
    int src = n + hs;
    if (src >= W) src -= W;
    a[s * ld + n] = in[(size_t)(row0 + s) * W + src];
  }
  __syncthreads();
  const Twid<float2> twd{twW, W};
  const float2* z = fft_lds<+1, float2>(a, b, plan, twd, 1, nseq, ld, tid, nthr);
  float dacc = 0.f;
  for (int i = tid; i < nseq * W; i += nthr) {
    const int s = i / W, k = i - s * W;
    int dst = k + hs;
    if (dst >= W) dst -= W;
    const size_t e = (size_t)(row0 + s) * W + dst;
    const float pv = pin[e];
    const float av = fmaf(c * rho[(row0 + s) / rows_per_image], pv, z[s * ld + k].x * scale);
    out[e] = av;
    dacc = fmaf(pv, av, dacc);
  }
  if (!dot_partial) return;
  {
    // block sum (wave shuffles, then the waves' sums through LDS)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dacc += __shfl_xor(dacc, o);
    if ((tid & 63) == 0) shred[tid >> 6] = dacc;
    __syncthreads();
    if (tid == 0) {
      float t = 0.f;
      for (int w = 0; w < (nthr + 63) / 64; ++w) t += shred[w];
      dpx_st_agent(dot_partial + blockIdx.x, t);
    }
  }
  if (!dpx_last_block(counter, gridDim.x, &shlast)) return;
  const int bpi = rows_per_image / rpb, wave = tid >> 6, lane = tid & 63;          // workgroups per image
  for (int bimg = wave; bimg < B; bimg += (nthr + 63) / 64) {
    float t = 0.f;
    for (int i = lane; i < bpi; i += 64) t += dpx_ld_agent(dot_partial + (size_t)bimg * bpi + i);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
    if (lane == 0) pAp[bimg] = t;
  }
}

// ---- the same three kernels for planes of 64 R1 x 64 R1 (R1 = 5: 320 x 320, the CS-MRI configuration; R1 = 6: 384 x 384) with every
// 1-D transform on ONE wave's registers (fftR64_wave, dpx_fft_reg.h: radix R1 * 8 * 8, wave-local exchanges): a workgroup = 4 waves
// = 4 rows / 4 columns; no workgroup barrier inside a transform, no index arithmetic of the size-generic Stockham passes, the
// twiddles of a lane in 14 registers.  Same shifts, scale factors, mask handling, direction update and <p, Ap> hand-over as above.
template <int R1>
__global__ void __launch_bounds__(256) k_crows_real_in_w(float* __restrict__ in, float2* __restrict__ out, int nrows, const float2* __restrict__ twW,
                                                         float scale, const float* __restrict__ dir_r, const float* __restrict__ beta,
                                                         const int* __restrict__ done, int rows_per_image) {
  constexpr int W = 64 * R1, hs = W / 2, S = LdsSeq<W>::SLOTS;
  __shared__ float2 lds[4 * S];
  // (the solve's `done` word is looked at AFTER this launch's loads have been issued and before its first store: one memory round trip at
  //  the head of the launch instead of two -- these launches are a few microseconds long)
  const int dn = done ? done[0] : 0;
  const int tid = threadIdx.x, t = tid & 63, wv = tid >> 6;
  const int row = blockIdx.x * 4 + wv;
  if (row >= nrows) return;                              // (wave-uniform)
  TwR64<R1> tw;
  tw.load(t, twW, 1);
  const float bt = dir_r ? beta[row / rows_per_image] : 0.f;
  float xin[R1], rin[R1];
#pragma unroll
  for (int m = 0; m < R1; ++m) {
    int src = t + 64 * m + hs;
    if (src >= W) src -= W;
    const size_t e = (size_t)row * W + src;
    xin[m] = in[e];
    rin[m] = dir_r ? dir_r[e] : 0.f;
  }
  if (dn) return;
  float2 v[R1];
#pragma unroll
  for (int m = 0; m < R1; ++m) {
    int src = t + 64 * m + hs;
    if (src >= W) src -= W;
    const size_t e = (size_t)row * W + src;
    float x = xin[m];
    if (dir_r) {
      x = fmaf(bt, x, rin[m]);
      in[e] = x;
    }
    v[m] = make_float2(x, 0.f);
  }
  fftR64_wave<R1, -1>(v, lds + wv * S, t, tw, WaveSync());
#pragma unroll
  for (int m = 0; m < R1; ++m) {
    int dst = t + 64 * m + hs;
    if (dst >= W) dst -= W;
    out[(size_t)row * W + dst] = cscale(v[m], scale);
  }
}

template <int R1>
__global__ void __launch_bounds__(256) k_ccols_mask_w(float2* __restrict__ data, const float* __restrict__ mask, int mask_images, int W,
                                                      const float2* __restrict__ twH, int square, const int* __restrict__ done) {
  constexpr int H = 64 * R1, hs = H / 2, S = LdsSeq<H>::SLOTS, ld = H + 1;
  __shared__ float2 col[4 * ld];
  __shared__ float2 lds[4 * S];
  const int dn = done ? done[0] : 0;                     // (looked at behind the loads, see k_crows_real_in_w)
  const int tid = threadIdx.x, t = tid & 63, wv = tid >> 6;
  const int p = blockIdx.y, l0 = blockIdx.x * 4;
  const int nseq = min(4, W - l0);
  float2* base = data + (size_t)p * H * W;
  const float* mk = mask + (mask_images == 1 ? (size_t)0 : (size_t)p * H * W);
  for (int i = tid; i < H * nseq; i += 256) {            // the workgroup's columns, rows shifted to the transform's order
    const int r = i / nseq, c = i - r * nseq;
    int src = r + hs;
    if (src >= H) src -= H;
    col[c * ld + r] = base[(size_t)src * W + l0 + c];
  }
  if (dn) return;                                        // (uniform over the launch)
  __syncthreads();
  if (wv < nseq) {
    TwR64<R1> tw;
    tw.load(t, twH, 1);
    float2 v[R1];
#pragma unroll
    for (int m = 0; m < R1; ++m) v[m] = col[wv * ld + t + 64 * m];
    fftR64_wave<R1, -1>(v, lds + wv * S, t, tw, WaveSync());
    // bin k sits at centred row (k + hs) % H, which is where the inverse transform reads its input from: the shifts cancel, only the
    // mask is indexed with the centred row
#pragma unroll
    for (int m = 0; m < R1; ++m) {
      int row = t + 64 * m + hs;
      if (row >= H) row -= H;
      const float mv = mk[(size_t)row * W + l0 + wv];
      v[m] = cscale(v[m], square ? mv * mv : mv);
    }
    WaveSync()();
    fftR64_wave<R1, +1>(v, lds + wv * S, t, tw, WaveSync());
#pragma unroll
    for (int m = 0; m < R1; ++m) col[wv * ld + t + 64 * m] = v[m];
  }
  __syncthreads();
  for (int i = tid; i < H * nseq; i += 256) {
    const int k = i / nseq, c = i - k * nseq;
    int dst = k + hs;
    if (dst >= H) dst -= H;
    base[(size_t)dst * W + l0 + c] = col[c * ld + k];
  }
}

template <int R1>
__global__ void __launch_bounds__(256) k_crows_real_out_w(const float2* __restrict__ in, float* __restrict__ out, const float* __restrict__ pin,
                                                          const float* __restrict__ rho, float c, const int* __restrict__ done, int rows_per_image,
                                                          int nrows, const float2* __restrict__ twW, float scale,
                                                          float* __restrict__ dot_partial, unsigned* __restrict__ counter, float* __restrict__ pAp,
                                                          int B) {
  constexpr int W = 64 * R1, hs = W / 2, S = LdsSeq<W>::SLOTS;
  __shared__ float2 lds[4 * S];
  __shared__ float shred[4];
  __shared__ int shlast;
  const int dn = done ? done[0] : 0;                     // (looked at behind the loads, see k_crows_real_in_w)
  const int tid = threadIdx.x, t = tid & 63, wv = tid >> 6;
  const int row = blockIdx.x * 4 + wv;                   // (nrows is a multiple of 4: rows_per_image is)
  TwR64<R1> tw;
  tw.load(t, twW, 1);
  float2 v[R1];
  float pvs[R1];
#pragma unroll
  for (int m = 0; m < R1; ++m) {
    int src = t + 64 * m + hs;
    if (src >= W) src -= W;
    v[m] = in[(size_t)row * W + src];
    pvs[m] = pin[(size_t)row * W + src];                 // (the direction at the output position t + 64 m + hs of this lane: the same index)
  }
  const float cr = c * rho[row / rows_per_image];
  if (dn) return;
  fftR64_wave<R1, +1>(v, lds + wv * S, t, tw, WaveSync());
  float dacc = 0.f;
#pragma unroll
  for (int m = 0; m < R1; ++m) {
    int dst = t + 64 * m + hs;
    if (dst >= W) dst -= W;
    const size_t e = (size_t)row * W + dst;
    const float pv = pvs[m];
    const float av = fmaf(cr, pv, v[m].x * scale);
    out[e] = av;
    dacc = fmaf(pv, av, dacc);
  }
  if (!dot_partial) return;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) dacc += __shfl_xor(dacc, o);
  if (t == 0) shred[wv] = dacc;
  __syncthreads();
  if (tid == 0) dpx_st_agent(dot_partial + blockIdx.x, ((shred[0] + shred[1]) + shred[2]) + shred[3]);
  if (!dpx_last_block(counter, gridDim.x, &shlast)) return;
  const int bpi = rows_per_image / 4;                    // workgroups per image
  for (int bimg = wv; bimg < B; bimg += 4) {
    float a = 0.f;
    for (int i = t; i < bpi; i += 64) a += dpx_ld_agent(dot_partial + (size_t)bimg * bpi + i);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if (t == 0) pAp[bimg] = a;
  }
}

// r / beta: the direction update in front (NULL: none); done_fwd / done_out: the converged flag the forward kernels / the last kernel
// look at (NULL: none); square: `mask` holds the mask (1) or its square (0); dotws / counter / pAp: the <p, Ap> hand-over (NULL: none)
template <int R1>
static int masked_normal_apply_wave(float* p, const float* r, const float* beta, float* Ap, float2* z, const float* mask, int mask_images, int square,
                                    const float* rho, float c, const int* done_fwd, const int* done_out, float* dotws, unsigned* counter, float* pAp,
                                    int B, const void* table, hipStream_t s, const char* what) {
  constexpr int N = 64 * R1;
  const float scale = 1.0f / (float)N;                   // 1 / sqrt(H W)
  const dim3 grow(B * N / 4), gcol(N / 4, B);
  DPX_LAUNCH("k_crows_real_in", (k_crows_real_in_w<R1>), grow, dim3(256), 0, s, p, z, B * N, tw_rows(table), scale, r, beta, done_fwd, N);
  DPX_LAUNCH("k_ccols_mask", (k_ccols_mask_w<R1>), gcol, dim3(256), 0, s, z, mask, mask_images, N, tw_cols(table, N), square, done_fwd);
  DPX_LAUNCH("k_crows_real_out", (k_crows_real_out_w<R1>), grow, dim3(256), 0, s, (const float2*)z, Ap, (const float*)p, rho, c, done_out, N, B * N,
             tw_rows(table), scale, dotws, counter, pAp, B);
  return launch_status(what);
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
size_t pow2_spec_elems(int P, int H, int W);
int spectral_apply_pow2(const float* x, float* y, int op, const SpecArgs& a, int B, int C, int H, int W,
                        const void* table, void* ws, hipStream_t stream);

static int rows_per_block(int M) {
  int r = 4096 / (M + 1);
  if (r < 1) r = 1;
  if (r > 16) r = 16;
  return r;
}

template <class K> static void il_lds_attr(K kernel, size_t sh) {
  if (sh > 48 * 1024) hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
}
template <bool EVEN, int CT, int NT>
static void launch_rows_il_t(bool fwd, const float* x, float2* spec, float* y, int W, int nrows, const Plan1D& prow, const float2* twW, hipStream_t s) {
  const size_t sh = (size_t)prow.n * (CT > 1 ? CT + 1 : 1) * sizeof(float2);
  const dim3 grid((nrows + CT - 1) / CT);
  // (TWL = false: measured on the one-wave row workgroups, the table copy halves the workgroups a CU holds and costs more than the gathers it
  //  saves -- 8 x 3 x 1000 x 1000: 62.9 -> 68.5 us, 1080 x 1920: 113 -> 170 us; four one-wave rows per workgroup around ONE shared copy were measured too: 64.7 -> 71 us;
  //  the instantiation stays for A/B builds)
  if (fwd) {
    il_lds_attr(k_rows_r2c_il<EVEN, CT, NT, false>, sh);
    DPX_LAUNCH("k_rows_r2c_il", (k_rows_r2c_il<EVEN, CT, NT, false>), grid, dim3(NT), sh, s, x, spec, W, nrows, prow, twW);
  } else {
    il_lds_attr(k_rows_c2r_il<EVEN, CT, NT, false>, sh);
    DPX_LAUNCH("k_rows_c2r_il", (k_rows_c2r_il<EVEN, CT, NT, false>), grid, dim3(NT), sh, s, (const float2*)spec, y, W, nrows, prow, twW, 1.0f);
  }
}
// ct: 1 = one row per one-wave workgroup (row lengths up to 1024 complex points: the rule), 8 = eight rows interleaved on 512 threads (A/B),
// 4 = four rows on 512 threads (lengths up to 2048)
static void launch_rows_il(bool fwd, bool even, int ct, const float* x, float2* spec, float* y, int W, int nrows, const Plan1D& prow,
                           const float2* twW, hipStream_t s) {
  if (ct == 1) {
    if (even) launch_rows_il_t<true, 1, 64>(fwd, x, spec, y, W, nrows, prow, twW, s);
    else launch_rows_il_t<false, 1, 64>(fwd, x, spec, y, W, nrows, prow, twW, s);
  } else if (ct == 8) {
    if (even) launch_rows_il_t<true, 8, 512>(fwd, x, spec, y, W, nrows, prow, twW, s);
    else launch_rows_il_t<false, 8, 512>(fwd, x, spec, y, W, nrows, prow, twW, s);
  } else {
    if (even) launch_rows_il_t<true, 4, 512>(fwd, x, spec, y, W, nrows, prow, twW, s);
    else launch_rows_il_t<false, 4, 512>(fwd, x, spec, y, W, nrows, prow, twW, s);
  }
}
template <int OP, int CT, int NT>
static void launch_cols_il_t(float2* spec, const SpecArgs& A, int P, int C, int H, int W, const Plan1D& pcol, const float2* twH, hipStream_t s) {
  const size_t sh = (size_t)H * (CT + 1) * sizeof(float2);
  // 8 x the longest per-XCD work list (k_cols_il: unit u = (channel, block of 8 adjacent tiles) -> XCD u % 8)
  const int ntile = (spec_cols(W) + CT - 1) / CT, nblk = (ntile + 7) / 8, U = C * nblk, nb = P / C;
  int longest = 0;
  for (int x = 0; x < 8; ++x) {
    int n = 0;
    for (int u = x; u < U; u += 8) n += ((u % nblk) * 8 + 8 <= ntile ? 8 : ntile - (u % nblk) * 8) * nb;
    longest = n > longest ? n : longest;
  }
  const dim3 grid(8 * longest);
  // the twiddle table in shared memory when the workgroups per CU stay what they are without it (knob il_tw_lds = 0: never)
  const size_t sh_tw = sh + (size_t)H * sizeof(float2), lds_cu = 160 * 1024;
  if (sh_tw <= lds_cu && lds_cu / sh_tw >= (lds_cu / sh > 2 ? 2 : lds_cu / sh)) {
    il_lds_attr(k_cols_il<OP, CT, NT, true>, sh_tw);
    DPX_LAUNCH("k_cols_il", (k_cols_il<OP, CT, NT, true>), grid, dim3(NT), sh_tw, s, spec, A, C, H, W, pcol, twH, P);
    return;
  }
  il_lds_attr(k_cols_il<OP, CT, NT, false>, sh);
  DPX_LAUNCH("k_cols_il", (k_cols_il<OP, CT, NT, false>), grid, dim3(NT), sh, s, spec, A, C, H, W, pcol, twH, P);
}
template <int CT, int NT>
static void launch_cols_il_o(int op, float2* spec, const SpecArgs& A, int P, int C, int H, int W, const Plan1D& pcol, const float2* twH, hipStream_t s) {
  if (op == OP_MUL) launch_cols_il_t<OP_MUL, CT, NT>(spec, A, P, C, H, W, pcol, twH, s);
  else if (op == OP_MULCONJ) launch_cols_il_t<OP_MULCONJ, CT, NT>(spec, A, P, C, H, W, pcol, twH, s);
  else launch_cols_il_t<OP_SOLVE, CT, NT>(spec, A, P, C, H, W, pcol, twH, s);
}
// ct: 8 = eight columns on 512 threads, 4 = four columns on 512 threads (column lengths above 1024)
static void launch_cols_il(int op, int ct, float2* spec, const SpecArgs& A, int P, int C, int H, int W, const Plan1D& pcol, const float2* twH,
                           hipStream_t s) {
  if (ct == 8) launch_cols_il_o<8, 512>(op, spec, A, P, C, H, W, pcol, twH, s);
  else launch_cols_il_o<4, 512>(op, spec, A, P, C, H, W, pcol, twH, s);
}

int spectral_apply(const float* x, float* y, int op, const SpecArgs& A, int B, int C, int H, int W,
                   const void* table, void* ws, hipStream_t stream) {
  if (pow2_path_available(H, W)) return spectral_apply_pow2(x, y, op, A, B, C, H, W, table, ws, stream);
  const int P = B * C, Ws = spec_cols(W);
  const bool even = (W % 2 == 0);
  const int M = even ? W / 2 : W;
  const Plan1D prow = make_plan(M), pcol = make_plan(H);
  float2* spec = (float2*)ws;
  const int nrows = P * H;
  const int rpb = rows_per_block(M);
  const size_t shrow = (size_t)2 * rpb * (M + 1) * sizeof(float2);
  const dim3 grow((nrows + rpb - 1) / rpb);
  const int il = tune(TUNE_GENERIC_INTERLEAVED);          // 1 = rows and columns, 2 = rows only, 3 = columns only, 4 = both, rows eight to a workgroup (A/B)
  int rct = (il == 1 || il == 2 || il == 4) ? il_seqs(prow) : 0;
  const int cct = (il == 1 || il == 3 || il == 4) ? il_seqs(pcol) : 0;
  if (rct == 8 && il != 4) rct = 1;                      // a row per one-wave workgroup: no workgroup barrier, the waves of a CU drift apart
  if (!rct && shrow > 160 * 1024) {
    set_error("row length %d too large for the LDS-resident generic FFT", W);
    return DPX_ERR_UNSUPPORTED;
  }
  // second form of the size-generic kernels (interleaved sequences, passes in place): knob generic_interleaved, default on
  if (rct) {
    launch_rows_il(true, even, rct, x, spec, y, W, nrows, prow, tw_rows(table), stream);
  } else if (even)
    DPX_LAUNCH("k_rows_r2c", (k_rows_r2c<true>), grow, dim3(256), shrow, stream, x, spec, W, nrows, prow, tw_rows(table), rpb);
  else
    DPX_LAUNCH("k_rows_r2c", (k_rows_r2c<false>), grow, dim3(256), shrow, stream, x, spec, W, nrows, prow, tw_rows(table), rpb);
  if (cct) launch_cols_il(op, cct, spec, A, P, C, H, W, pcol, tw_cols(table, W), stream);

  int CT = (int)(60 * 1024 / (2 * (size_t)(H + 1) * sizeof(float2)));
  if (CT < 1) CT = 1;
  if (CT > 16) CT = 16;
  const size_t shcol = (size_t)2 * CT * (H + 1) * sizeof(float2);
  if (!cct && shcol > 160 * 1024) {
    set_error("column length %d too large for the LDS-resident generic FFT", H);
    return DPX_ERR_UNSUPPORTED;
  }
  const dim3 gcol((Ws + CT - 1) / CT, P);
  const float2* twH = tw_cols(table, W);
  if (shcol > 60 * 1024) {
    hipFuncSetAttribute((const void*)k_cols<OP_MUL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shcol);
    hipFuncSetAttribute((const void*)k_cols<OP_MULCONJ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shcol);
    hipFuncSetAttribute((const void*)k_cols<OP_SOLVE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shcol);
  }
  if (!cct) switch (op) {
    case OP_MUL: DPX_LAUNCH("k_cols", (k_cols<OP_MUL>), gcol, dim3(256), shcol, stream, spec, A, C, H, W, pcol, twH, CT); break;
    case OP_MULCONJ: DPX_LAUNCH("k_cols", (k_cols<OP_MULCONJ>), gcol, dim3(256), shcol, stream, spec, A, C, H, W, pcol, twH, CT); break;
    default: DPX_LAUNCH("k_cols", (k_cols<OP_SOLVE>), gcol, dim3(256), shcol, stream, spec, A, C, H, W, pcol, twH, CT); break;
  }
  if (rct)
    launch_rows_il(false, even, rct, x, spec, y, W, nrows, prow, tw_rows(table), stream);
  else if (even)
    DPX_LAUNCH("k_rows_c2r", (k_rows_c2r<true>), grow, dim3(256), shrow, stream, spec, y, W, nrows, prow, tw_rows(table), rpb, 1.0f);
  else
    DPX_LAUNCH("k_rows_c2r", (k_rows_c2r<false>), grow, dim3(256), shrow, stream, spec, y, W, nrows, prow, tw_rows(table), rpb, 1.0f);
  return launch_status("spectral_apply");
}

// does a plane fit the LDS-resident transforms of masked_normal_apply?  (the same arithmetic as below)
bool masked_normal_fits(int H, int W) {
  const int rpb = rows_per_block(W);
  int CT = (int)(60 * 1024 / (2 * (size_t)(H + 1) * sizeof(float2)));
  CT = CT < 1 ? 1 : (CT > 16 ? 16 : CT);
  return (size_t)2 * rpb * (W + 1) * sizeof(float2) <= 64 * 1024 && (size_t)2 * CT * (H + 1) * sizeof(float2) <= 64 * 1024;
}

// z: one complex [B][H][W] scratch plane set.  Returns DPX_ERR_UNSUPPORTED for planes beyond the LDS-resident transform.
int masked_normal_apply(const float* p, float* Ap, float2* z, const float* mask2, int mask_images, const float* rho, float c, const int* done,
                        int B, int H, int W, const void* table, hipStream_t s) {
  if (H == W && tune(TUNE_CG_WAVE_FFT) != 2) {         // (as in the fused iteration below)
    const char* what = "masked_normal_apply";
    if (H == 320)
      return masked_normal_apply_wave<5>((float*)p, nullptr, nullptr, Ap, z, mask2, mask_images, 0, rho, c, nullptr, done, nullptr, nullptr, nullptr, B,
                                         table, s, what);
    if (H == 384)
      return masked_normal_apply_wave<6>((float*)p, nullptr, nullptr, Ap, z, mask2, mask_images, 0, rho, c, nullptr, done, nullptr, nullptr, nullptr, B,
                                         table, s, what);
  }
  const Plan1D prow = make_plan(W), pcol = make_plan(H);
  const int rpb = rows_per_block(W);
  const size_t shrow = (size_t)2 * rpb * (W + 1) * sizeof(float2);
  int CT = (int)(60 * 1024 / (2 * (size_t)(H + 1) * sizeof(float2)));
  CT = CT < 1 ? 1 : (CT > 16 ? 16 : CT);
  const size_t shcol = (size_t)2 * CT * (H + 1) * sizeof(float2);
  if (shrow > 64 * 1024 || shcol > 64 * 1024) {
    set_error("masked_normal_apply: plane %dx%d too large for the LDS-resident transform", H, W);
    return DPX_ERR_UNSUPPORTED;
  }
  const float scale = 1.0f / sqrtf((float)H * (float)W);
  const dim3 grow((B * H + rpb - 1) / rpb), gcol((W + CT - 1) / CT, B);
  DPX_LAUNCH("k_crows_real_in", k_crows_real_in, grow, dim3(256), shrow, s, (float*)p, z, W, B * H, prow, tw_rows(table), rpb, scale, (const float*)nullptr,
             (const float*)nullptr, (const int*)nullptr, H);
  DPX_LAUNCH("k_ccols_mask", k_ccols_mask, gcol, dim3(256), shcol, s, z, mask2, mask_images, H, W, pcol, tw_cols(table, W), CT, 0, (const int*)nullptr);
  DPX_LAUNCH("k_crows_real_out", k_crows_real_out, grow, dim3(256), shrow, s, (const float2*)z, Ap, p, rho, c, done, H, W, B * H, prow,
             tw_rows(table), rpb, scale, (float*)nullptr, (unsigned*)nullptr, (float*)nullptr, B);
  return launch_status("masked_normal_apply");
}

// rows per row workgroup of the fused iteration: the largest divisor of H the LDS-resident transform holds (a workgroup's rows then
// belong to one image: its <p, Ap> share is one number)
// ... and, for a small batch, few enough rows that the launch has a workgroup for every CU: these launches are latency-bound (4 x 320^2:
// 1.6 MB), and 128 workgroups of 10 rows each walk every Stockham pass twice (400 butterflies on 256 threads) on half the chip.
static int fused_rpb(int B, int H, int W) {
  int r = rows_per_block(W);
  const int knob = tune(TUNE_CG_ROWS_PER_WG);
  if (knob > 0 && knob < r) r = knob;
  while (r > 1 && (H % r || (knob <= 0 && (long)B * H / r < 320))) --r;      // (measured at 4 x 320^2: 10 / 5 / 4 / 2 rows: 0.757 / 0.721 / 0.706 / 0.720 ms per outer iteration)
  return r;
}
static int fused_ct(int B, int H, int W) {
  int CT = (int)(60 * 1024 / (2 * (size_t)(H + 1) * sizeof(float2)));
  CT = CT < 1 ? 1 : (CT > 16 ? 16 : CT);
  const int knob = tune(TUNE_CG_COLS_PER_WG);
  if (knob > 0) return knob < CT ? knob : CT;
  while (CT > 4 && (long)B * ((W + CT - 1) / CT) < 320) --CT;      // (at least 4 columns = 32-byte pieces of a row; 11 / 8 / 5 / 4 columns: 0.737 / 0.720 / 0.714 / 0.706 ms)
  return CT;
}
size_t masked_normal_fused_ws_floats(int B, int H, int W) { return (size_t)B * H + 8; }      // (one partial per workgroup; at most one workgroup per row)

// The matvec of dpx_cg_masked_fft's fused iteration: the three launches above with the CG direction update in front (p = r + beta p
// formed in the first kernel's load) and <p, Ap> behind (partial sums in the last kernel's store, finished by its last workgroup
// into the CG state).  `mask` is the mask itself (squared on the fly).  dotws: masked_normal_fused_ws_floats floats.
int masked_normal_apply_fused(float* p, const float* r, float* Ap, float2* z, const float* mask, int mask_images, const float* rho, float c,
                              float* state, float* dotws, unsigned* counter, int B, int H, int W, const void* table, hipStream_t s) {
  // 320 x 320 / 384 x 384 planes: every transform on one wave's registers (knob cg_wave_fft = 2 keeps the size-generic kernels)
  if (H == W && tune(TUNE_CG_WAVE_FFT) != 2) {
    const CgState S{state, B};
    const char* what = "masked_normal_apply_fused";
    if (H == 320)
      return masked_normal_apply_wave<5>(p, r, S.beta(), Ap, z, mask, mask_images, 1, rho, c, S.flags(), S.flags(), dotws, counter, S.pAp(), B, table, s, what);
    if (H == 384)
      return masked_normal_apply_wave<6>(p, r, S.beta(), Ap, z, mask, mask_images, 1, rho, c, S.flags(), S.flags(), dotws, counter, S.pAp(), B, table, s, what);
  }
  const Plan1D prow = make_plan(W), pcol = make_plan(H);
  const int rpb = fused_rpb(B, H, W);
  const size_t shrow = (size_t)2 * rpb * (W + 1) * sizeof(float2);
  const int CT = fused_ct(B, H, W);
  const size_t shcol = (size_t)2 * CT * (H + 1) * sizeof(float2);
  const CgState S{state, B};
  const int* done = S.flags();
  const float scale = 1.0f / sqrtf((float)H * (float)W);
  const dim3 grow(B * H / rpb), gcol((W + CT - 1) / CT, B);
  DPX_LAUNCH("k_crows_real_in", k_crows_real_in, grow, dim3(256), shrow, s, p, z, W, B * H, prow, tw_rows(table), rpb, scale, r, (const float*)S.beta(), done,
             H);
  DPX_LAUNCH("k_ccols_mask", k_ccols_mask, gcol, dim3(256), shcol, s, z, mask, mask_images, H, W, pcol, tw_cols(table, W), CT, 1, done);
  DPX_LAUNCH("k_crows_real_out", k_crows_real_out, grow, dim3(256), shrow, s, (const float2*)z, Ap, (const float*)p, rho, c, done, H, W, B * H, prow,
             tw_rows(table), rpb, scale, dotws, counter, S.pAp(), B);
  return launch_status("masked_normal_apply_fused");
}

}  // namespace dpx

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
using namespace dpx;

extern "C" size_t dpx_fft_table_bytes(int H, int W) { return (size_t)(H + W) * sizeof(float2); }

extern "C" int dpx_fft_table_init(void* table, int H, int W, dpx_stream_t stream) {
  DPX_REQUIRE(table && H > 0 && W > 0, "dpx_fft_table_init: bad arguments");
  float2* t = (float2*)table;
  DPX_LAUNCH("k_twiddle_table", k_twiddle_table, dim3((W + 255) / 256), dim3(256), 0, (hipStream_t)stream, t, W);
  DPX_LAUNCH("k_twiddle_table", k_twiddle_table, dim3((H + 255) / 256), dim3(256), 0, (hipStream_t)stream, t + W, H);
  return launch_status("dpx_fft_table_init");
}

// two half-spectrum buffers (the power-of-two column pass works out of place); power-of-two planes keep the
// Nyquist bins in a side array [P][H] behind the main [P][H][W/2] array instead of packing them into column 0
extern "C" size_t dpx_spectrum_bytes(int P, int H, int W) {
  const size_t one = pow2_path_available(H, W) ? pow2_spec_elems(P, H, W) : (size_t)P * H * spec_cols(W);
  return 2 * one * sizeof(float2);
}

extern "C" int dpx_fft_conv(const float* x, float* y, const void* otf, int conj_otf, int B, int C, int H, int W,
                            const void* table, void* ws, dpx_stream_t stream) {
  DPX_REQUIRE(x && y && otf && table && ws, "dpx_fft_conv: null pointer");
  DPX_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "dpx_fft_conv: bad shape");
  SpecArgs a{};
  a.otf = (const float2*)otf;
  a.scale = 1.0f / ((float)H * (float)W);
  return spectral_apply(x, y, conj_otf ? OP_MULCONJ : OP_MUL, a, B, C, H, W, table, ws, (hipStream_t)stream);
}

// geometry of the fp64 data-spectrum pass: CT columns per column workgroup (LDS: two images of CT sequences of H fp64 points),
// RPB rows per row workgroup; 0 = the plane does not fit the LDS-resident transform
static size_t ds_ld(int n) { return (size_t)n + n / 8 + 1; }      // padded LDS length of one fp64 sequence (IdxPad8)
static int ds_ct(int H) {
  // two columns per workgroup (256 threads, two workgroups per CU: one loads while the other transforms) beat four (512 threads, one per
  // CU) at 8x3x1024^2: column pass 234 vs 267 us, row pass 164 vs 156 us (64-byte instead of 128-byte pieces); DPX_DS_CT=4 forces four
  const int env = 0;
  if (env == 4 && 4 * 2 * ds_ld(H) * sizeof(double2) <= 160 * 1024) return 4;
  return 2 * 2 * ds_ld(H) * sizeof(double2) <= 160 * 1024 ? 2 : 0;
}
static int ds_rpb(int W) {
  const int M = (W % 2 == 0) ? W / 2 : W;
  // (measured at 8x3x1024^2, 256 threads: 2 rows per workgroup -- 4 workgroups per CU -- 174 us, 4 rows 207 us, 1 row 225 us)
  for (int r = 2; r >= 1; r >>= 1)
    if ((size_t)r * 2 * ds_ld(M) * sizeof(double2) <= 40 * 1024) return r;
  return 2 * ds_ld(M) * sizeof(double2) <= 160 * 1024 ? 1 : 0;
}
static size_t ds_spec_elems(int P, int H, int W) {
  const int CT = ds_ct(H) ? ds_ct(H) : 4, Wh = W / 2 + 1;
  return (size_t)P * ((Wh + CT - 1) / CT) * H * CT;
}
// workspace: the tile-major fp64 half spectrum + the two fp64 twiddle tables
extern "C" size_t dpx_data_spectrum_ws_bytes(int P, int H, int W) { return (ds_spec_elems(P, H, W) + (size_t)W + (size_t)H) * sizeof(double2); }

extern "C" int dpx_data_spectrum(const float* b, const void* otf, int conj_otf, void* spec_out, int accumulate, int B, int C,
                                 int H, int W, void* ws, dpx_stream_t stream) {
  DPX_REQUIRE(b && spec_out && ws, "dpx_data_spectrum: null pointer");
  DPX_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "dpx_data_spectrum: bad shape");
  const int P = B * C, Wh = W / 2 + 1;
  const int env_rpb = tune(TUNE_DS_RPB), env_rt = tune(TUNE_DS_ROW_THREADS), env_ct = tune(TUNE_DS_COL_THREADS);     // tuning
  const int CT = ds_ct(H);
  int rpb = ds_rpb(W);
  if (env_rpb && env_rpb <= rpb) rpb = env_rpb;
  if (!CT || !rpb) {
    set_error("dpx_data_spectrum: plane %dx%d too large for the LDS-resident fp64 transform", H, W);
    return DPX_ERR_UNSUPPORTED;
  }
  const bool even = (W % 2 == 0);
  const int M = even ? W / 2 : W;
  const size_t shrow = (size_t)rpb * 2 * ds_ld(M) * sizeof(double2), shcol = (size_t)CT * 2 * ds_ld(H) * sizeof(double2);
  hipStream_t s = (hipStream_t)stream;
  double2* spec64 = (double2*)ws;
  double2* twW = spec64 + ds_spec_elems(P, H, W);
  double2* twH = twW + W;
  DPX_LAUNCH("k_twiddle_table_f64", k_twiddle_table_f64, dim3((W + 255) / 256), dim3(256), 0, s, twW, W);
  DPX_LAUNCH("k_twiddle_table_f64", k_twiddle_table_f64, dim3((H + 255) / 256), dim3(256), 0, s, twH, H);
  const int nrows = P * H;
  if (even) {
    if (shrow > 48 * 1024) hipFuncSetAttribute((const void*)k_rows_r2c_f64<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shrow);
    DPX_LAUNCH("k_rows_r2c_f64", k_rows_r2c_f64<true>, dim3((nrows + rpb - 1) / rpb), dim3(env_rt ? env_rt : 256), shrow, s, b, spec64, W, nrows, H, make_plan(M),
               (const double2*)twW, rpb, CT);
  } else {
    if (shrow > 48 * 1024) hipFuncSetAttribute((const void*)k_rows_r2c_f64<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shrow);
    DPX_LAUNCH("k_rows_r2c_f64", k_rows_r2c_f64<false>, dim3((nrows + rpb - 1) / rpb), dim3(256), shrow, s, b, spec64, W, nrows, H, make_plan(M),
               (const double2*)twW, rpb, CT);
  }
  if (shcol > 48 * 1024) hipFuncSetAttribute((const void*)k_cols_fwd_f64, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shcol);
  DPX_LAUNCH("k_cols_fwd_f64", k_cols_fwd_f64, dim3((((Wh + CT - 1) / CT + 8 * (16 / CT) - 1) / (8 * (16 / CT))) * (8 * (16 / CT)), P), dim3(env_ct ? env_ct : (CT >= 4 ? 512 : 256)), shcol, s, (const double2*)spec64,
             (float2*)spec_out, (const float2*)otf, conj_otf, accumulate, C, H, W, make_plan(H),
             pow2_path_available(H, W) ? 1 : 0, P, (const double2*)twH, CT);
  return launch_status("dpx_data_spectrum");
}

extern "C" int dpx_fourier_solve(const float* rhs, float* x, const void* spec_add, const void* dd, const float* rho, float eps,
                                 int B, int C, int H, int W, const void* table, void* ws, dpx_stream_t stream) {
  DPX_REQUIRE(rhs && x && table && ws && rho && dd, "dpx_fourier_solve: null pointer");
  DPX_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "dpx_fourier_solve: bad shape");
  SpecArgs a{};
  a.add = (const float2*)spec_add;
  a.dd = (const float2*)dd;
  a.rho = rho;
  a.eps = eps;
  a.eps_num = eps;
  a.scale = 1.0f / ((float)H * (float)W);
  return spectral_apply(rhs, x, OP_SOLVE, a, B, C, H, W, table, ws, (hipStream_t)stream);
}

extern "C" int dpx_fourier_apply_inv(const float* g, float* out, const void* dd, const float* rho, float eps, int B, int C, int H, int W,
                                     const void* table, void* ws, dpx_stream_t stream) {
  DPX_REQUIRE(g && out && table && ws && rho && dd, "dpx_fourier_apply_inv: null pointer");
  DPX_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "dpx_fourier_apply_inv: bad shape");
  SpecArgs a{};
  a.dd = (const float2*)dd;
  a.rho = rho;
  a.eps = eps;
  a.eps_num = 0.f;
  a.scale = 1.0f / ((float)H * (float)W);
  return spectral_apply(g, out, OP_SOLVE, a, B, C, H, W, table, ws, (hipStream_t)stream);
}

extern "C" int dpx_cfft2(const void* in, void* out, int inverse, int centred, int ortho, int P, int H, int W, const void* table,
                         dpx_stream_t stream) {
  DPX_REQUIRE(in && out && table && in != out, "dpx_cfft2: null or aliased pointers");
  DPX_REQUIRE(P > 0 && H > 0 && W > 0, "dpx_cfft2: bad shape");
  hipStream_t s = (hipStream_t)stream;
  const Plan1D prow = make_plan(W), pcol = make_plan(H);
  const int rpb = rows_per_block(W);
  const size_t shrow = (size_t)2 * rpb * (W + 1) * sizeof(float2);
  int CT = (int)(60 * 1024 / (2 * (size_t)(H + 1) * sizeof(float2)));
  CT = CT < 1 ? 1 : (CT > 16 ? 16 : CT);
  const size_t shcol = (size_t)2 * CT * (H + 1) * sizeof(float2);
  if (shrow > 64 * 1024 || shcol > 64 * 1024) {
    set_error("dpx_cfft2: plane %dx%d too large for the LDS-resident transform", H, W);
    return DPX_ERR_UNSUPPORTED;
  }
  const float scale = ortho ? 1.0f / sqrtf((float)H * (float)W) : (inverse ? 1.0f / ((float)H * (float)W) : 1.0f);
  const dim3 grow((P * H + rpb - 1) / rpb), gcol((W + CT - 1) / CT, P);
  if (!inverse) {
    DPX_LAUNCH("k_crows", (k_crows<-1>), grow, dim3(256), shrow, s, (const float2*)in, (float2*)out, W, P * H, prow, tw_rows(table), rpb, centred, scale);
    DPX_LAUNCH("k_ccols", (k_ccols<-1>), gcol, dim3(256), shcol, s, (float2*)out, H, W, pcol, tw_cols(table, W), CT, centred);
  } else {
    DPX_LAUNCH("k_crows", (k_crows<+1>), grow, dim3(256), shrow, s, (const float2*)in, (float2*)out, W, P * H, prow, tw_rows(table), rpb, centred, scale);
    DPX_LAUNCH("k_ccols", (k_ccols<+1>), gcol, dim3(256), shcol, s, (float2*)out, H, W, pcol, tw_cols(table, W), CT, centred);
  }
  return launch_status("dpx_cfft2");
}
