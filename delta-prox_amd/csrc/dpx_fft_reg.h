// Register-resident radix passes for power-of-two transforms (gfx950).
//
// A length-N complex transform is carried by T threads holding V = N/T values each.  Three Stockham
// passes: radix V in registers -> (LDS) -> radix RM = N/V^2 -> (LDS) -> radix V in registers, so an
// element crosses LDS twice per transform (2 writes + 2 reads) instead of once per radix-2/4/8 pass.
// On entry v[m] = x[t + m*T]; on exit v[m] = X[t + m*T] -- the same striding, so a forward transform's
// output registers feed the inverse transform's first pass directly (the per-frequency operator of the
// x-update runs on registers between the two).
//
// LDS image of one sequence: element i lives at slot i + (i >> 4) (one pad slot per 16) which keeps the
// "thread t writes V consecutive elements" pattern of the first pass off a single bank.
#pragma once
#include "dpx_common.h"

namespace dpx {

__device__ __forceinline__ int lds_slot(int i) { return i + (i >> 4); }
template <int N> struct LdsSeq { static constexpr int SLOTS = N + N / 16; };

// Experiment, OFF by default (-DDPX_FFT_BASEOFF=1 enables it): the slot of every LDS access of the three passes as  base(t) +
// compile-time constant, so that the constant becomes the instruction's immediate offset (written as lds_slot(t-dependent index +
// constant) the compiler re-derives `i + (i >> 4)` per access).  Valid for the splits the kernels use (checked exhaustively against
// lds_slot on the host; the emulator tests pass with it).  Measured at 8x3x1024^2, alternating runs on one box: the column kernel's
// static instruction count 5468 -> 4895 (v_ashrrev 153 -> 29, LDS instructions 400 -> 304: paired ds_read2 / ds_write2), and it got
// SLOWER -- k_cols_p2 70.5 -> 73.5 us, k_iter_rows_seq 104.5 -> 105.0 us: the kernel is bound by its dependent chain, not by VALU
// issue, and the paired LDS instructions lengthen that chain.
#ifndef DPX_FFT_BASEOFF
#define DPX_FFT_BASEOFF 0
#endif
template <int N, int T> struct LdsIdx {
  static constexpr int V = N / T, RM = N / (V * V);
  static constexpr bool FAST = DPX_FFT_BASEOFF && (T % 16 == 0) && (V == 8 || V == 16) && (T % V == 0) &&
                               (RM == 1 || ((N / RM) % 16 == 0 && RM % 2 == 0));
  // pass A writes element t V + m
  __device__ static __forceinline__ int base_a(int t) { return V == 16 ? 17 * t : 8 * t + (t >> 1); }
  // pass B reads element t + i T + mm N/RM; pass C reads element t + m T
  __device__ static __forceinline__ int base_t(int t) { return t + (t >> 4); }
  __device__ static constexpr int off_br(int i, int mm) { return (i * T + mm * (N / RM)) + (i * T + mm * (N / RM)) / 16; }
  __device__ static constexpr int off_c(int m) { return m * T + (m * T) / 16; }
  // pass B writes element (jb - k) RM + k + mm V,  jb = t + i T,  k = jb % V = t % V
  __device__ static __forceinline__ int base_bw(int t) {
    const int k = t % V, tb = t - k;
    return RM * tb + k + (RM * tb) / 16;
  }
  __device__ static constexpr int off_bw(int i, int mm) { return RM * i * T + (RM * i * T) / 16 + mm * V + (V == 16 ? mm : (mm >> 1)); }
};

// ---- small in-register DFTs, natural order in and out ------------------------------------------------
template <int DIR> __device__ __forceinline__ void rdft2(float2& a, float2& b) {
  const float2 t = csub(a, b);
  a = cadd(a, b);
  b = t;
}
template <int DIR> __device__ __forceinline__ void rdft4(float2& a0, float2& a1, float2& a2, float2& a3) {
  const float2 s0 = cadd(a0, a2), d0 = csub(a0, a2);
  const float2 s1 = cadd(a1, a3), e1 = csub(a1, a3);
  a0 = cadd(s0, s1);
  a2 = csub(s0, s1);
  a1 = cadd_rot<DIR>(d0, e1);          // d0 +- (-+i) e1: the rotation is the butterfly's operand selection
  a3 = csub_rot<DIR>(d0, e1);
}
// multiply by exp(DIR * i * pi * q / 8), q = 1..7, constants folded
template <int DIR, int Q> __device__ __forceinline__ float2 rot16(float2 a) {
  constexpr float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, h = 0.70710678118654752440f;
  constexpr float cr = (Q == 1) ? c1 : (Q == 2) ? h : (Q == 3) ? s1 : (Q == 4) ? 0.f : (Q == 5) ? -s1 : (Q == 6) ? -h : -c1;
  constexpr float sr = (Q == 1) ? s1 : (Q == 2) ? h : (Q == 3) ? c1 : (Q == 4) ? 1.f : (Q == 5) ? c1 : (Q == 6) ? h : s1;
  constexpr float si = DIR < 0 ? -sr : sr;
  return cmul_const(a, cr, si);
}
template <int DIR> __device__ __forceinline__ void rdft8(float2 (&v)[8]) {
  rdft4<DIR>(v[0], v[2], v[4], v[6]);     // even samples -> E[0..3] in v[0],v[2],v[4],v[6]
  rdft4<DIR>(v[1], v[3], v[5], v[7]);     // odd samples  -> O[0..3] in v[1],v[3],v[5],v[7]
  const float2 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
  const float2 o0 = v[1], o1 = rot16<DIR, 2>(v[3]), o2 = v[5], o3 = rot16<DIR, 6>(v[7]);
  v[0] = cadd(e0, o0); v[4] = csub(e0, o0);
  v[1] = cadd(e1, o1); v[5] = csub(e1, o1);
  v[2] = cadd_rot<DIR>(e2, o2); v[6] = csub_rot<DIR>(e2, o2);
  v[3] = cadd(e3, o3); v[7] = csub(e3, o3);
}
template <int DIR> __device__ __forceinline__ void rdft16(float2 (&v)[16]) {
  float2 e[8] = {v[0], v[2], v[4], v[6], v[8], v[10], v[12], v[14]};
  float2 o[8] = {v[1], v[3], v[5], v[7], v[9], v[11], v[13], v[15]};
  rdft8<DIR>(e);
  rdft8<DIR>(o);
  o[1] = rot16<DIR, 1>(o[1]);
  o[2] = rot16<DIR, 2>(o[2]);
  o[3] = rot16<DIR, 3>(o[3]);
  o[5] = rot16<DIR, 5>(o[5]);
  o[6] = rot16<DIR, 6>(o[6]);
  o[7] = rot16<DIR, 7>(o[7]);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (k == 4) {                        // o[4] carries the factor -+i: folded into the butterfly
      v[k] = cadd_rot<DIR>(e[k], o[k]);
      v[k + 8] = csub_rot<DIR>(e[k], o[k]);
    } else {
      v[k] = cadd(e[k], o[k]);
      v[k + 8] = csub(e[k], o[k]);
    }
  }
}
template <int R, int DIR> __device__ __forceinline__ void rdft(float2 (&v)[R]) {
  if constexpr (R == 2) rdft2<DIR>(v[0], v[1]);
  else if constexpr (R == 4) rdft4<DIR>(v[0], v[1], v[2], v[3]);
  else if constexpr (R == 8) rdft8<DIR>(v);
  else rdft16<DIR>(v);
}

template <int DIR> __device__ __forceinline__ float2 twmul(float2 a, float2 w) {
  return DIR < 0 ? cmul(a, w) : cmulc(a, w);
}

// ---- the three-pass transform -------------------------------------------------------------------------
// lds: this sequence's LdsSeq<N>::SLOTS float2 slots; tw: exp(-2 pi i k / (N*TWS)) table, stride TWS;
// sync(): barrier over (at least) the T threads of the sequence.  Ends with all LDS reads done but NOT
// synchronised: call sync() before the same LDS region is written again.
struct NoHook {
  __device__ __forceinline__ void operator()() const {}
};
// after_reads(): called once, right after the last LDS read of the transform has been issued and before the last
// pass's arithmetic -- the place to start memory traffic that should overlap that arithmetic.
template <int N, int T, int DIR, class Sync, class Hook = NoHook>
__device__ __forceinline__ void fft_reg(float2 (&v)[N / T], float2* __restrict__ lds, int t, const float2* __restrict__ tw,
                                        int tws, Sync sync, Hook after_reads = Hook()) {
  DPX_OPAQUE(t);      // every call re-derives its few index registers instead of keeping all of them alive
  constexpr int V = N / T;
  constexpr int RM = N / (V * V);
  static_assert(V * V * RM == N && (RM == 1 || RM == 2 || RM == 4 || RM == 8), "unsupported N/T split");
  static_assert(RM <= V, "middle radix must fit the per-thread registers");
  using IX = LdsIdx<N, T>;
  // pass A: radix V, stride 1, no twiddles
  rdft<V, DIR>(v);
  if constexpr (IX::FAST) {
    float2* pa = lds + IX::base_a(t);
#pragma unroll
    for (int m = 0; m < V; ++m) pa[m] = v[m];
  } else {
#pragma unroll
    for (int m = 0; m < V; ++m) lds[lds_slot(t * V + m)] = v[m];
  }
  sync();
  if constexpr (RM > 1) {
    // pass B: radix RM, Ns = V; this thread owns butterflies jb = t + i*T
    constexpr int NB = V / RM;
    if constexpr (IX::FAST) {
      const float2* pr = lds + IX::base_t(t);
#pragma unroll
      for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int mm = 0; mm < RM; ++mm) v[i * RM + mm] = pr[IX::off_br(i, mm)];
    } else {
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int jb = t + i * T;
#pragma unroll
        for (int mm = 0; mm < RM; ++mm) v[i * RM + mm] = lds[lds_slot(jb + mm * (N / RM))];
      }
    }
    sync();
    float2* pw = lds + (IX::FAST ? IX::base_bw(t) : 0);
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int jb = t + i * T;
      const int k = jb % V;
      float2 a[RM];
#pragma unroll
      for (int mm = 0; mm < RM; ++mm) a[mm] = v[i * RM + mm];
#pragma unroll
      for (int mm = 1; mm < RM; ++mm) a[mm] = twmul<DIR>(a[mm], tw[(k * mm * V) * tws]);   // W_{V*RM}^{k*mm}
      rdft<RM, DIR>(a);
      if constexpr (IX::FAST) {
#pragma unroll
        for (int mm = 0; mm < RM; ++mm) pw[IX::off_bw(i, mm)] = a[mm];
      } else {
        const int j0 = (jb - k) * RM + k;
#pragma unroll
        for (int mm = 0; mm < RM; ++mm) lds[lds_slot(j0 + mm * V)] = a[mm];
      }
    }
    sync();
  }
  // pass C: radix V, Ns = N/V = T.  Twiddles W_N^{t*m} are exact table values; they are applied in groups
  // of four so that at most four of them are live at a time (register pressure at V = 16).
  if constexpr (IX::FAST) {
    const float2* pc = lds + IX::base_t(t);
#pragma unroll
    for (int m = 0; m < V; ++m) v[m] = pc[IX::off_c(m)];
  } else {
#pragma unroll
    for (int m = 0; m < V; ++m) v[m] = lds[lds_slot(t + m * T)];
  }
  after_reads();
  const unsigned tstep = (unsigned)(t * tws);
#pragma unroll
  for (int m = 1; m < V; ++m) {
    v[m] = twmul<DIR>(v[m], tw[tstep * (unsigned)m]);
    if (V > 8 && (m & 3) == 3) __builtin_amdgcn_sched_barrier(0);
  }
  rdft<V, DIR>(v);
}

// ---- length 3N: three interleaved length-N transforms + one radix-3 butterfly on registers -------------------------------------------
// v[r * VS + a], VS = N / T:   on entry (DIR < 0) x[3 (t + a T) + r]          -> on exit X[(t + a T) + r N]
//                              on entry (DIR > 0) X[(t + a T) + r N]          -> on exit x[3 (t + a T) + r]
// (decimation in time forward, in frequency backward: the two index maps are each other's inverse, so the operator between a forward
// and an inverse transform works on registers exactly as in the power-of-two case -- it only has to address its tables with the
// frequency map).  The three sub-transforms run pass by pass in three LDS regions of LdsSeq<N>::SLOTS slots: the same number of
// barriers as one power-of-two transform.  tw: exp(-2 pi i k / (3N * tws)) table, stride tws.
template <int DIR> __device__ __forceinline__ void rdft3(float2& a0, float2& a1, float2& a2) {
  constexpr float h = 0.86602540378443864676f;
  const float2 s = cadd(a1, a2), d = cscale(csub(a1, a2), h);
  const float2 m = make_float2(fmaf(-0.5f, s.x, a0.x), fmaf(-0.5f, s.y, a0.y));
  a0 = cadd(a0, s);
  a1 = cadd_rot<DIR>(m, d);
  a2 = csub_rot<DIR>(m, d);
}
template <int N, int T, int DIR, class Sync, class Hook = NoHook>
__device__ __forceinline__ void fft_reg_x3(float2 (&v)[3 * N / T], float2* __restrict__ lds, int t, const float2* __restrict__ tw, int tws,
                                           Sync sync, Hook after_reads = Hook()) {
  DPX_OPAQUE(t);
  constexpr int V = N / T, RM = N / (V * V), SL = LdsSeq<N>::SLOTS;
  static_assert(V * V * RM == N && (RM == 1 || RM == 2 || RM == 4 || RM == 8) && RM <= V, "unsupported N/T split");
  if constexpr (DIR > 0) {                                   // X[k + q N] -> G_r[k] = conj(w^{r k}) sum_q w3^{+r q} X[k + q N]
#pragma unroll
    for (int a = 0; a < V; ++a) {
      rdft3<DIR>(v[a], v[V + a], v[2 * V + a]);
      const int k = t + a * T;
      v[V + a] = twmul<DIR>(v[V + a], tw[k * tws]);
      v[2 * V + a] = twmul<DIR>(v[2 * V + a], tw[2 * k * tws]);
    }
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    float2 a[V];
#pragma unroll
    for (int m = 0; m < V; ++m) a[m] = v[r * V + m];
    rdft<V, DIR>(a);
#pragma unroll
    for (int m = 0; m < V; ++m) lds[r * SL + lds_slot(t * V + m)] = a[m];
  }
  sync();
  if constexpr (RM > 1) {
    constexpr int NB = V / RM;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int mm = 0; mm < RM; ++mm) v[r * V + i * RM + mm] = lds[r * SL + lds_slot(t + i * T + mm * (N / RM))];
    sync();
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int jb = t + i * T;
        const int k = jb % V;
        float2 a[RM];
#pragma unroll
        for (int mm = 0; mm < RM; ++mm) a[mm] = v[r * V + i * RM + mm];
#pragma unroll
        for (int mm = 1; mm < RM; ++mm) a[mm] = twmul<DIR>(a[mm], tw[(k * mm * V) * 3 * tws]);   // W_{V*RM}^{k*mm}
        rdft<RM, DIR>(a);
        const int j0 = (jb - k) * RM + k;
#pragma unroll
        for (int mm = 0; mm < RM; ++mm) lds[r * SL + lds_slot(j0 + mm * V)] = a[mm];
      }
    sync();
  }
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int m = 0; m < V; ++m) v[r * V + m] = lds[r * SL + lds_slot(t + m * T)];
  after_reads();
  const unsigned tstep = (unsigned)(t * 3 * tws);
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    float2 a[V];
    a[0] = v[r * V];
#pragma unroll
    for (int m = 1; m < V; ++m) a[m] = twmul<DIR>(v[r * V + m], tw[tstep * (unsigned)m]);
    rdft<V, DIR>(a);
#pragma unroll
    for (int m = 0; m < V; ++m) v[r * V + m] = a[m];
    __builtin_amdgcn_sched_barrier(0);
  }
  if constexpr (DIR < 0) {                                   // F_r[k] -> X[k + q N] = sum_r w3^{r q} w^{r k} F_r[k]
#pragma unroll
    for (int a = 0; a < V; ++a) {
      const int k = t + a * T;
      v[V + a] = twmul<DIR>(v[V + a], tw[k * tws]);
      v[2 * V + a] = twmul<DIR>(v[2 * V + a], tw[2 * k * tws]);
      rdft3<DIR>(v[a], v[V + a], v[2 * V + a]);
    }
  }
}

// ---- length 384 = 6 * 8 * 8 on ONE wave (T = 64 lanes, 6 values per lane), natural order in and out -----------------------------------
// v[m] = x[t + 64 m] on entry, X[t + 64 m] on exit -- the striding of the power-of-two transforms above, so the kernels that own
// "pixel pairs t + m T" of a row (the streaming row kernels of the two-kernel iteration) take 768-wide rows unchanged: the reference's
// own patch size is 768 x 768 (contrib/optic/utils.py:158-166).  Stockham passes of radix 6 (all 64 lanes), 8 and 8 (48 butterflies each:
// lanes 48 .. 63 idle) and one redistribution read; every exchange is wave-local (sync = wave barrier).
//   pass A: j = t < 64    in x[j + 64 m]              out[6 j + m]
//   pass B: j < 48, k = j % 6: in[j + 48 m] W_48^{k m}  out[8 (j - k) + k + 6 m]
//   pass C: j < 48:        in[j + 48 m] W_384^{j m}     out[j + 48 m]
// twb[m - 1] = W_48^{(t % 6) m}, twc[m - 1] = W_384^{t m}, m = 1 .. 7 (lanes t < 48), forward-direction values.
template <int DIR> __device__ __forceinline__ void rdft6(float2 (&v)[6]) {
  float2 e0 = v[0], e1 = v[2], e2 = v[4], o0 = v[1], o1 = v[3], o2 = v[5];
  rdft3<DIR>(e0, e1, e2);
  rdft3<DIR>(o0, o1, o2);
  constexpr float h = 0.86602540378443864676f, si = DIR < 0 ? -h : h;
  o1 = cmul_const(o1, 0.5f, si);                       // W_6
  o2 = cmul_const(o2, -0.5f, si);                      // W_6^2
  v[0] = cadd(e0, o0); v[3] = csub(e0, o0);
  v[1] = cadd(e1, o1); v[4] = csub(e1, o1);
  v[2] = cadd(e2, o2); v[5] = csub(e2, o2);
}
struct Tw384 {
  float2 b[7], c[7];
  __device__ __forceinline__ void load(int t, const float2* __restrict__ tw, int tws) {      // tw[n tws] = exp(-2 pi i n / 384)
    const int j = t < 48 ? t : 0, k = j % 6;
#pragma unroll
    for (int m = 1; m < 8; ++m) {
      b[m - 1] = tw[(8 * k * m) * tws];
      c[m - 1] = tw[(j * m) * tws];
    }
  }
};
template <int DIR, class Sync>
__device__ __forceinline__ void fft384_wave(float2 (&v)[6], float2* __restrict__ lds, int t, const Tw384& W, Sync sync) {
  rdft6<DIR>(v);
#pragma unroll
  for (int m = 0; m < 6; ++m) lds[lds_slot(6 * t + m)] = v[m];
  sync();
  float2 a[8];
  const bool act = t < 48;
  const int j = act ? t : 0, k = j % 6;
#pragma unroll
  for (int m = 0; m < 8; ++m) a[m] = lds[lds_slot(j + 48 * m)];
  sync();
#pragma unroll
  for (int m = 1; m < 8; ++m) a[m] = twmul<DIR>(a[m], W.b[m - 1]);
  rdft8<DIR>(a);
  if (act) {
#pragma unroll
    for (int m = 0; m < 8; ++m) lds[lds_slot(8 * (j - k) + k + 6 * m)] = a[m];
  }
  sync();
#pragma unroll
  for (int m = 0; m < 8; ++m) a[m] = lds[lds_slot(j + 48 * m)];
  sync();
#pragma unroll
  for (int m = 1; m < 8; ++m) a[m] = twmul<DIR>(a[m], W.c[m - 1]);
  rdft8<DIR>(a);
  if (act) {
#pragma unroll
    for (int m = 0; m < 8; ++m) lds[lds_slot(j + 48 * m)] = a[m];
  }
  sync();
#pragma unroll
  for (int m = 0; m < 6; ++m) v[m] = lds[lds_slot(t + 64 * m)];
}

// ---- the same scheme for any length R1 * 64 (R1 = 5: 320 points, the plane size of the CS-MRI configuration; R1 = 6: 384) ------------
// radix-5 butterfly (Winograd form: 5 real multiplications per component), natural order
template <int DIR> __device__ __forceinline__ void rdft5(float2 (&v)[5]) {
  constexpr float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;      // cos 72, cos 144 degrees
  constexpr float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;       // sin 72, sin 144 degrees
  const float2 t1 = cadd(v[1], v[4]), t2 = cadd(v[2], v[3]), t3 = csub(v[1], v[4]), t4 = csub(v[2], v[3]);
  const float2 m1 = make_float2(fmaf(c2, t2.x, fmaf(c1, t1.x, v[0].x)), fmaf(c2, t2.y, fmaf(c1, t1.y, v[0].y)));
  const float2 m2 = make_float2(fmaf(c1, t2.x, fmaf(c2, t1.x, v[0].x)), fmaf(c1, t2.y, fmaf(c2, t1.y, v[0].y)));
  const float2 q1 = make_float2(fmaf(s2, t4.x, s1 * t3.x), fmaf(s2, t4.y, s1 * t3.y));
  const float2 q2 = make_float2(fmaf(-s1, t4.x, s2 * t3.x), fmaf(-s1, t4.y, s2 * t3.y));
  v[0] = cadd(v[0], cadd(t1, t2));
  v[1] = cadd_rot<DIR>(m1, q1);           // forward: m1 - i q1
  v[4] = csub_rot<DIR>(m1, q1);
  v[2] = cadd_rot<DIR>(m2, q2);
  v[3] = csub_rot<DIR>(m2, q2);
}
template <int R1> struct TwR64 {
  float2 b[7], c[7];
  // tw[n tws] = exp(-2 pi i n / (64 R1));  b[m - 1] = W_{8 R1}^{(j % R1) m},  c[m - 1] = W_{64 R1}^{j m},  j = t < 8 R1
  __device__ __forceinline__ void load(int t, const float2* __restrict__ tw, int tws) {
    const int j = t < 8 * R1 ? t : 0, k = j % R1;
#pragma unroll
    for (int m = 1; m < 8; ++m) {
      b[m - 1] = tw[(8 * k * m) * tws];
      c[m - 1] = tw[(j * m) * tws];
    }
  }
};
// v[m] = x[t + 64 m] -> X[t + 64 m], m < R1; lds: LdsSeq<64 R1>::SLOTS float2 of this wave
template <int R1, int DIR, class Sync>
__device__ __forceinline__ void fftR64_wave(float2 (&v)[R1], float2* __restrict__ lds, int t, const TwR64<R1>& W, Sync sync) {
  static_assert(R1 == 5 || R1 == 6, "first-pass radix 5 or 6");
  constexpr int NB = 8 * R1;                           // radix-8 butterflies per pass
  if constexpr (R1 == 5) rdft5<DIR>(v);
  else rdft6<DIR>(v);
#pragma unroll
  for (int m = 0; m < R1; ++m) lds[lds_slot(R1 * t + m)] = v[m];
  sync();
  float2 a[8];
  const bool act = t < NB;
  const int j = act ? t : 0, k = j % R1;
#pragma unroll
  for (int m = 0; m < 8; ++m) a[m] = lds[lds_slot(j + NB * m)];
  sync();
#pragma unroll
  for (int m = 1; m < 8; ++m) a[m] = twmul<DIR>(a[m], W.b[m - 1]);
  rdft8<DIR>(a);
  if (act) {
#pragma unroll
    for (int m = 0; m < 8; ++m) lds[lds_slot(8 * (j - k) + k + R1 * m)] = a[m];
  }
  sync();
#pragma unroll
  for (int m = 0; m < 8; ++m) a[m] = lds[lds_slot(j + NB * m)];
  sync();
#pragma unroll
  for (int m = 1; m < 8; ++m) a[m] = twmul<DIR>(a[m], W.c[m - 1]);
  rdft8<DIR>(a);
  if (act) {
#pragma unroll
    for (int m = 0; m < 8; ++m) lds[lds_slot(j + NB * m)] = a[m];
  }
  sync();
#pragma unroll
  for (int m = 0; m < R1; ++m) v[m] = lds[lds_slot(t + 64 * m)];
}

// the same transform with its twiddles read from a table on the spot (kernels that transform one row per thread group)
template <int DIR, class Sync>
__device__ __forceinline__ void fft384_wave_tab(float2 (&v)[6], float2* __restrict__ lds, int t, const float2* __restrict__ tw, int tws, Sync sync) {
  Tw384 W;
  W.load(t, tw, tws);
  fft384_wave<DIR>(v, lds, t, W, sync);
}

// The twiddles a thread needs depend only on its index t: kernels that transform many sequences with the same
// thread mapping (one row after another) load them once into registers and reuse them for every transform.
template <int N, int T, bool KEEPB = true> struct TwRegs {
  static constexpr int V = N / T, RM = N / (V * V), NB = (RM > 1) ? V / RM : 0;
  float2 b[(RM > 1 && KEEPB) ? NB * (RM - 1) : 1];
  float2 c[V - 1];
  const float2* twb_;      // pass-B twiddles W_{V*RM}^j at twb_[j * bstride_] (the global table, or a 64-entry LDS copy)
  int bstride_;
  __device__ __forceinline__ void load(int t, const float2* __restrict__ tw, int tws) {
    twb_ = tw;
    bstride_ = V * tws;
    if constexpr (RM > 1 && KEEPB) {
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int k = (t + i * T) % V;
#pragma unroll
        for (int mm = 1; mm < RM; ++mm) b[i * (RM - 1) + mm - 1] = tw[(k * mm * V) * tws];
      }
    }
#pragma unroll
    for (int m = 1; m < V; ++m) c[m - 1] = tw[(t * m) * tws];
  }
};

// 384 points on one wave: the twiddle registers of fft384_wave behind the same interface (twb_ / bstride_ are accepted and unused)
template <bool KEEPB> struct TwRegs<384, 64, KEEPB> {
  Tw384 w;
  const float2* twb_;
  int bstride_;
  __device__ __forceinline__ void load(int t, const float2* __restrict__ tw, int tws) {
    twb_ = tw;
    bstride_ = tws;
    w.load(t, tw, tws);
  }
};

// fft_reg with preloaded twiddles (same passes, same LDS image)
template <int N, int T, int DIR, bool KEEPB, class Sync>
__device__ __forceinline__ void fft_reg_tw(float2 (&v)[N / T], float2* __restrict__ lds, int t, const TwRegs<N, T, KEEPB>& W, Sync sync) {
  if constexpr (N == 384 && T == 64) {
    fft384_wave<DIR>(v, lds, t, W.w, sync);
    return;
  } else {
  constexpr int V = N / T;
  constexpr int RM = N / (V * V);
  using IX = LdsIdx<N, T>;
  rdft<V, DIR>(v);
  if constexpr (IX::FAST) {
    float2* pa = lds + IX::base_a(t);
#pragma unroll
    for (int m = 0; m < V; ++m) pa[m] = v[m];
  } else {
#pragma unroll
    for (int m = 0; m < V; ++m) lds[lds_slot(t * V + m)] = v[m];
  }
  sync();
  if constexpr (RM > 1) {
    constexpr int NB = V / RM;
    if constexpr (IX::FAST) {
      const float2* pr = lds + IX::base_t(t);
#pragma unroll
      for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int mm = 0; mm < RM; ++mm) v[i * RM + mm] = pr[IX::off_br(i, mm)];
    } else {
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int jb = t + i * T;
#pragma unroll
        for (int mm = 0; mm < RM; ++mm) v[i * RM + mm] = lds[lds_slot(jb + mm * (N / RM))];
      }
    }
    sync();
    float2* pw = lds + (IX::FAST ? IX::base_bw(t) : 0);
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int jb = t + i * T;
      const int k = jb % V;
      float2 a[RM];
#pragma unroll
      for (int mm = 0; mm < RM; ++mm) a[mm] = v[i * RM + mm];
#pragma unroll
      for (int mm = 1; mm < RM; ++mm) {
        if constexpr (KEEPB) a[mm] = twmul<DIR>(a[mm], W.b[i * (RM - 1) + mm - 1]);
        else a[mm] = twmul<DIR>(a[mm], W.twb_[(k * mm) * W.bstride_]);
      }
      rdft<RM, DIR>(a);
      if constexpr (IX::FAST) {
#pragma unroll
        for (int mm = 0; mm < RM; ++mm) pw[IX::off_bw(i, mm)] = a[mm];
      } else {
        const int j0 = (jb - k) * RM + k;
#pragma unroll
        for (int mm = 0; mm < RM; ++mm) lds[lds_slot(j0 + mm * V)] = a[mm];
      }
    }
    sync();
  }
  if constexpr (IX::FAST) {
    const float2* pc = lds + IX::base_t(t);
#pragma unroll
    for (int m = 0; m < V; ++m) v[m] = pc[IX::off_c(m)];
  } else {
#pragma unroll
    for (int m = 0; m < V; ++m) v[m] = lds[lds_slot(t + m * T)];
  }
#pragma unroll
  for (int m = 1; m < V; ++m) v[m] = twmul<DIR>(v[m], W.c[m - 1]);
  rdft<V, DIR>(v);
  }
}

struct BlockSync {
  __device__ __forceinline__ void operator()() const { DPX_LDS_BARRIER(); }
};
struct WaveSync {
  __device__ __forceinline__ void operator()() const { __builtin_amdgcn_wave_barrier(); }
};

}  // namespace dpx
