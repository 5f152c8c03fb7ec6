// The unrolled ADMM backward pass on power-of-two planes: TWO kernels per backward iteration, the mirror image of the forward loop
// (dpx_iter.hip).  Reference: torch.autograd through `UnrolledSolver` (algo/specialization/unroll.py:21-58, algo/admm.py:49-59); what
// is differentiated is restated in dpx_autodiff.hip.
//
//   k_cols_p2<OP_SOLVE> (dpx_fft_pow2.hip): the transposed x-update  g_rhs^ = g_x^ / (|O|^2 + rho sum |G_i|^2 + eps), spectrum in / out
//   k_bwd_rows (this file): for a band of R image rows of one plane
//        inverse row FFT (finishes g_rhs of iteration t)
//     -> the rhs stage of iteration t and the z stage of iteration t - 1 (what k_rhs_z_bwd4 does on images):
//          g_v_i = rho K_i g_rhs,  g_u_i = a_i - g_v_i,  g_d_i = J_i (g_v_i - g_u_i) + g_u_i  (J_i from the saved prox output v_i),
//          a_i' = g_d_i,  g_x = sum_i K_i^T g_d_i,  and the partial sums of d/d rho_t (<g, sum K^T K x> and <g, rhs>) and d/d lam_i
//     -> forward row FFT of g_x: the first half of the next transposed x-update
//   g_rhs and g_x never touch HBM (g_rhs is emitted on request: the offsets' gradient needs it as an image).  Per backward iteration
//   and element: one spectrum in, one out, a_i in / out and the history planes x, rhs, v_i -- against the staged form's extra
//   g_rhs write + read and g_x write + read (4 plane passes) and two more launches.
//
// Row dependencies as in k_iter_rows: grad along H couples row h with h + 1 (K) and h - 1 (K^T); the SPB row sequences of a workgroup
// advance through the band in lock step, a ring of SPB + 2 rows of g and of the grad_H term's g_d lives in LDS, one halo row above /
// below the band is recomputed (R + 2 inverse transforms for R rows).
#include <cstdlib>
#include <cstring>

#ifndef DPX_PK_ASM
#define DPX_PK_ASM 0                                    // (as in dpx_iter.hip)
#endif
#include "dpx_fft_reg.h"
#include "dpx_bwd_dev.h"

namespace dpx {

// M = W / 2 pixel pairs per row, T lanes per row, SPB = 256 / T rows in flight per workgroup, NT terms.  Partial sums: one slot per
// workgroup, [row][nblk] with nblk = C * bands workgroups per image (part_lam rows = term * B + image).
// (three or four terms, and two with an fp32 history: one wave per SIMD -- up to 512 registers -- instead of 2 - 28 spilled ones; tools/spill_check.py)
template <int M, int T, int NT, bool HB>
__global__ void __launch_bounds__(256, (NT >= 3 || (NT == 2 && !HB)) ? 1 : 2) k_bwd_rows(const float2* __restrict__ spec_in, float2* __restrict__ spec_out, BwdRowTerms TT,
                                                   const float* __restrict__ rho_b, float* __restrict__ part_a, float* __restrict__ part_b,
                                                   float* __restrict__ part_lam, int B, int C, int H, int R, int bands, int P,
                                                   const float2* __restrict__ twW) {
  constexpr int V = M / T, SPB = 256 / T, S = LdsSeq<M>::SLOTS, RING = SPB + 2;     // (row qz - 1 = q - 2 is read while row q + SPB - 1 arrives)
  HIP_DYNAMIC_SHARED(float2, smem_bw)
  __shared__ float red[4 * (2 + DPX_MAX_TERMS)];
  float2* fft_lds = smem_bw;                          // SPB * S
  float2* gring = smem_bw + SPB * S;                  // RING rows of M float2 (pixel pairs) of g_rhs
  float2* wring = gring + RING * M;                   // RING rows of g_d of the grad_H term
  const int tid = threadIdx.x, j = tid / T, t = tid % T;
  const int lane = tid & 63, lbase = lane & ~(T - 1);
  const int pl = blockIdx.x / bands, band = blockIdx.x - pl * bands;
  const int r0 = band * R, Rb = min(R, H - r0);        // (the last band of a plane may be shorter)
  const int bi = pl / C, ci = pl - bi * C;
  const size_t plane_pairs = (size_t)pl * H * M;
  const float2* sin_main = spec_in + (size_t)pl * H * M;
  const float2* sin_side = spec_in + (size_t)P * H * M + (size_t)pl * H;
  float2* sout_main = spec_out + (size_t)pl * H * M;
  float2* sout_side = spec_out + (size_t)P * H * M + (size_t)pl * H;
  const unsigned tile_off = (unsigned)((t % SPEC_TILE) + (t / SPEC_TILE) * H * SPEC_TILE);   // bin t of a row in the tile-major spectrum
  const unsigned tile_step = (unsigned)((T / SPEC_TILE) * H * SPEC_TILE);            // bin t + m*T
  const float rho = rho_b[bi];
  float2* myfft = fft_lds + j * S;
  int hterm = -1, nW = 0, nH = 0;
  float cI = 0.f;
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    if (TT.t[i].linop == DPX_LIN_GRAD_H) hterm = i, ++nH;
    else if (TT.t[i].linop == DPX_LIN_GRAD_W) ++nW;
    else cI += 1.f;
  }
  const int nsteps = (Rb + 2 + SPB - 1) / SPB;
  const int pair = lbase | ((T - t) & (T - 1));         // lane holding bin M-k for this lane's bin k
  const int lnext = lbase | ((t + 1) & (T - 1)), lprev = lbase | ((t + T - 1) & (T - 1));


  float acc_a = 0.f, acc_b = 0.f, lsum[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) lsum[i] = 0.f;

  // prefetch the spectrum row of step 0
  float2 X[V];
  float xn;
  {
    const int h = (r0 - 1 + j + H) % H;
    const float2* in = sin_main + (unsigned)h * SPEC_TILE + tile_off;
#pragma unroll
    for (int m = 0; m < V; ++m) X[m] = in[tile_step * m];
    xn = sin_side[h].x;
  }

  for (int s = 0; s < nsteps; ++s) {
    const int q = s * SPB + j;                          // row (relative to r0 - 1) this sequence transforms
    const bool a_live = q <= Rb + 1;
    int h = r0 - 1 + q;                                 // circular rows: at most one wrap either way
    h = h < 0 ? h + H : (h >= H ? h - H : h);
    const int qz = q - 1;                               // row this sequence works on (g[qz] from LDS, g[qz + 1] own)
    const bool z_own = qz >= 1 && qz <= Rb;             // rows of this band (row qz = 0 is the halo above)
    int hz = r0 - 1 + qz;
    hz = hz < 0 ? hz + H : (hz >= H ? hz - H : hz);
    const size_t rowz = plane_pairs + (size_t)hz * M;   // pair index of the row's first pair
    // ---------------- phase A: inverse row transform of row q ----------------
    float2 ga[V];
#pragma unroll
    for (int m = 0; m < V; ++m) {
      const float2 got = make_float2(__shfl(X[V - 1 - m].x, pair), __shfl(X[V - 1 - m].y, pair));
      const float2 xm = cconj(t == 0 ? X[(V - m) % V] : got);
      const int k = t + m * T;
      const float2 xk = X[m];
      if (k == 0) {
        ga[m] = make_float2(xk.x + xn, xk.x - xn);
      } else {
        const float2 e = cadd(xk, xm);
        const float2 d = cmulc(csub(xk, xm), twW[k]);
        ga[m] = make_float2(e.x - d.y, e.y + d.x);
      }
    }
    WaveSync()();
    fft_reg<M, T, +1>(ga, myfft, t, twW, 2, WaveSync());   // ga[m] = (g[2n], g[2n+1]), n = t + m*T
    if (a_live) {
      float2* gr = gring + (q % RING) * M;
#pragma unroll
      for (int m = 0; m < V; ++m) gr[t + m * T] = ga[m];
      if (TT.g_out && q >= 1 && q <= Rb) {
        float2* go = (float2*)TT.g_out + plane_pairs + (size_t)h * M;
        if (TT.g_acc) {
#pragma unroll
          for (int m = 0; m < V; ++m) go[t + m * T] = cadd(go[t + m * T], ga[m]);
        } else {
#pragma unroll
          for (int m = 0; m < V; ++m) go[t + m * T] = ga[m];
        }
      }
    }
    DPX_LDS_BARRIER();
    // ---------------- phase B: the two stages on row qz ----------------
    float2 acc[V];                                        // sum_i K_i^T g_d_i (row-local parts)
#pragma unroll
    for (int m = 0; m < V; ++m) acc[m] = make_float2(0.f, 0.f);
    {
      // every lane runs the arithmetic (the shuffles need converged T-lane groups); only memory accesses are predicated
      const float2* gc_row = gring + ((qz + RING) % RING) * M;
      float2 gc[V];
#pragma unroll
      for (int m = 0; m < V; ++m) gc[m] = gc_row[t + m * T];
      // -- the two rho reductions of iteration t over the band's own rows: <g, rhs> and <g, L x> = <L g, x>, L = sum_i K_i^T K_i
      //    (symmetric: formed from the three g rows at hand -- the ring's rows qz - 1, qz and the own row qz + 1 -- so that one row of x is
      //    read instead of three)
      {
        const float2* gp_row = gring + ((qz - 1 + 2 * RING) % RING) * M;
#pragma unroll
        for (int m = 0; m < V; ++m) {
          float2 lg = make_float2(cI * gc[m].x, cI * gc[m].y);
          if (nW) {
            const float r_same = __shfl(gc[m].x, lnext), r_wrap = __shfl(gc[(m + 1) % V].x, lbase);
            const float l_same = __shfl(gc[m].y, lprev), l_wrap = __shfl(gc[(m + V - 1) % V].y, lbase | (T - 1));
            const float right = (t == T - 1) ? r_wrap : r_same, left = (t == 0) ? l_wrap : l_same;
            lg.x += (float)nW * (2.f * gc[m].x - left - gc[m].y);
            lg.y += (float)nW * (2.f * gc[m].y - gc[m].x - right);
          }
          if (nH) {
            const float2 gp = gp_row[t + m * T];
            lg.x += (float)nH * (2.f * gc[m].x - gp.x - ga[m].x);
            lg.y += (float)nH * (2.f * gc[m].y - gp.y - ga[m].y);
          }
          // (every row index is a valid row of the plane: the loads are unconditional, rows outside the band are left out by the selects)
          const float2 xr = hist_pair<HB>(TT.x, rowz + t + m * T);
          const float2 rr = hist_pair<HB>(TT.rhs, rowz + t + m * T);
          const float pa = fmaf(lg.y, xr.y, lg.x * xr.x), pb = fmaf(gc[m].y, rr.y, gc[m].x * rr.x);
          acc_a += z_own ? pa : 0.f;
          acc_b += z_own ? pb : 0.f;
        }
      }
      __builtin_amdgcn_sched_barrier(0);                  // (keeps the terms' loads from being hoisted above: registers)
      // -- g_v, g_u, g_d of every term; g_x = sum K^T g_d
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const BwdRowTerm tm = TT.t[i];
        const float lam = tm.lam ? tm.lam[bi] * tm.alpha : 0.f;
        const float sq = 1.f / (1.f + 2.f * lam);
        float2 av[V], vv[V], w[V];
#pragma unroll
        for (int m = 0; m < V; ++m) {
          av[m] = ((const float2*)tm.a_in)[rowz + t + m * T];
          vv[m] = hist_pair<HB>(tm.v, rowz + t + m * T);
        }
        if (tm.linop == DPX_LIN_IDENTITY) {
#pragma unroll
          for (int m = 0; m < V; ++m) w[m] = gc[m];
        } else if (tm.linop == DPX_LIN_GRAD_H) {
#pragma unroll
          for (int m = 0; m < V; ++m) w[m] = make_float2(ga[m].x - gc[m].x, ga[m].y - gc[m].y);
        } else {                                        // grad_W: g[w+1] - g[w]; pixel 2n+2 is the neighbour lane's .x
#pragma unroll
          for (int m = 0; m < V; ++m) {
            const float nx_same = __shfl(gc[m].x, lnext);
            const float nx_wrap = __shfl(gc[(m + 1) % V].x, lbase);
            const float gr = (t == T - 1) ? nx_wrap : nx_same;
            w[m] = make_float2(gc[m].y - gc[m].x, gr - gc[m].y);
          }
        }
        float lt;
        if (tm.prox == DPX_PROX_NORM1) lt = bwd_gd_row<1, V>(sq, rho, w, av, vv);
        else if (tm.prox == DPX_PROX_NONNEG) lt = bwd_gd_row<2, V>(sq, rho, w, av, vv);
        else lt = bwd_gd_row<0, V>(sq, rho, w, av, vv);
        lsum[i] += z_own ? lt : 0.f;
        if (z_own) {
#pragma unroll
          for (int m = 0; m < V; ++m) ((float2*)tm.a_out)[rowz + t + m * T] = w[m];
        }
        if (tm.linop == DPX_LIN_IDENTITY) {
#pragma unroll
          for (int m = 0; m < V; ++m) acc[m] = cadd(acc[m], w[m]);
        } else if (tm.linop == DPX_LIN_GRAD_W) {          // adjoint: y[w-1] - y[w]; pixel 2n-1 is the left lane's .y
#pragma unroll
          for (int m = 0; m < V; ++m) {
            const float l_same = __shfl(w[m].y, lprev);
            const float l_wrap = __shfl(w[(m + V - 1) % V].y, lbase | (T - 1));
            const float wl = (t == 0) ? l_wrap : l_same;
            acc[m] = make_float2(acc[m].x + (wl - w[m].x), acc[m].y + (w[m].x - w[m].y));
          }
        } else {                                          // grad_H: g_d goes to the ring, its adjoint is formed in phase C
          float2* wr = wring + ((qz + RING) % RING) * M;  // (rows that are not live land in slots nobody reads before they are rewritten)
#pragma unroll
          for (int m = 0; m < V; ++m) wr[t + m * T] = w[m];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    DPX_LDS_BARRIER();
    // ---- issue the spectrum loads of the next step's phase A: the forward transform below covers them ----
    if (s + 1 < nsteps && q + SPB <= Rb + 1) {
      int hn = r0 - 1 + q + SPB;
      hn = hn >= H ? hn - H : hn;
      const float2* in = sin_main + (unsigned)hn * SPEC_TILE + tile_off;
#pragma unroll
      for (int m = 0; m < V; ++m) X[m] = in[tile_step * m];
      xn = sin_side[hn].x;
    }
    // ---------------- phase C: g_x of row qz and its forward row transform ----------------
    {
      float2 z[V];
      if (hterm >= 0) {                                    // grad_H adjoint: y[h-1] - y[h]
        const float2* wp = wring + ((qz - 1 + RING) % RING) * M;
        const float2* wc = wring + ((qz + RING) % RING) * M;
#pragma unroll
        for (int m = 0; m < V; ++m) {
          const float2 up = wp[t + m * T], cu = wc[t + m * T];
          acc[m] = make_float2(acc[m].x + (up.x - cu.x), acc[m].y + (up.y - cu.y));
        }
      }
#pragma unroll
      for (int m = 0; m < V; ++m) z[m] = acc[m];
      WaveSync()();
      fft_reg<M, T, -1>(z, myfft, t, twW, 2, WaveSync());
      float2* out = sout_main + (unsigned)hz * SPEC_TILE + tile_off;
#pragma unroll
      for (int m = 0; m < V; ++m) {
        const float2 got = make_float2(__shfl(z[V - 1 - m].x, pair), __shfl(z[V - 1 - m].y, pair));
        const float2 zm = cconj(t == 0 ? z[(V - m) % V] : got);
        const int k = t + m * T;
        const float2 zk = z[m];
        float2 Xo;
        if (k == 0) {
          Xo = make_float2(zk.x + zk.y, 0.f);
          if (z_own) sout_side[hz] = make_float2(zk.x - zk.y, 0.f);
        } else {
          const float2 e = cscale(cadd(zk, zm), 0.5f);
          const float2 d = cscale(csub(zk, zm), 0.5f);
          Xo = cadd(e, cmul(make_float2(d.y, -d.x), twW[k]));
        }
        if (z_own) out[tile_step * m] = Xo;
      }
    }
    // ring rows read in phase B / C are rewritten by the next phase A / B, each behind a barrier; the transform scratch is wave-local
  }

  // ---- the workgroup's partial sums, one slot each (the finishing launch adds the slots of an image in index order) ----
  {
    const int wave = tid >> 6;
    float vals[2 + NT];
    vals[0] = bwd_wave_sum(acc_a);
    vals[1] = bwd_wave_sum(acc_b);
#pragma unroll
    for (int i = 0; i < NT; ++i) vals[2 + i] = bwd_wave_sum(lsum[i]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
      for (int e = 0; e < 2 + NT; ++e) red[wave * (2 + DPX_MAX_TERMS) + e] = vals[e];
    }
    __syncthreads();
    if (tid < 2 + NT) {
      const int W4 = 2 + DPX_MAX_TERMS;
      const float sum = ((red[tid] + red[W4 + tid]) + red[2 * W4 + tid]) + red[3 * W4 + tid];
      const int nblk = C * bands;
      const long slot = (long)ci * bands + band;
      if (tid == 0) part_a[(long)bi * nblk + slot] = -sum;
      else if (tid == 1) part_b[(long)bi * nblk + slot] = sum;
      else part_lam[((long)(tid - 2) * B + bi) * nblk + slot] = sum * TT.t[tid - 2].alpha;
    }
  }
}

static size_t bwd_rows_lds(int M, int T) {
  const int SPB = 256 / T, S = M + M / 16;
  return (size_t)(SPB * S + 2 * (SPB + 2) * M) * sizeof(float2);
}

template <int M, int T, int NT, bool HB>
static void launch_bwd_rows_hb(const float2* sin, float2* sout, const BwdRowTerms& TT, const float* rho, float* part_a, float* part_b, float* part_lam,
                               int B, int C, int H, int R, int bands, const float2* twW, hipStream_t s) {
  const size_t sh = bwd_rows_lds(M, T);
  static bool attr = false;
  if (!attr && sh > 48 * 1024) {
    hipFuncSetAttribute((const void*)k_bwd_rows<M, T, NT, HB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    attr = true;
  }
  const int P = B * C;
  DPX_LAUNCH("k_bwd_rows", (k_bwd_rows<M, T, NT, HB>), dim3(P * bands), dim3(256), sh, s, sin, sout, TT, rho, part_a, part_b, part_lam, B, C, H, R, bands, P,
             twW);
}
template <int M, int T, int NT>
static void launch_bwd_rows_nt(const float2* sin, float2* sout, const BwdRowTerms& TT, const float* rho, float* part_a, float* part_b, float* part_lam,
                               int B, int C, int H, int R, int bands, const float2* twW, hipStream_t s) {
  if (TT.hist_bf16) launch_bwd_rows_hb<M, T, NT, true>(sin, sout, TT, rho, part_a, part_b, part_lam, B, C, H, R, bands, twW, s);
  else launch_bwd_rows_hb<M, T, NT, false>(sin, sout, TT, rho, part_a, part_b, part_lam, B, C, H, R, bands, twW, s);
}
template <int M, int T>
static void launch_bwd_rows(const float2* sin, float2* sout, const BwdRowTerms& TT, const float* rho, float* part_a, float* part_b, float* part_lam,
                            int B, int C, int H, int R, int bands, const float2* twW, hipStream_t s) {
  switch (TT.n) {
    case 1: launch_bwd_rows_nt<M, T, 1>(sin, sout, TT, rho, part_a, part_b, part_lam, B, C, H, R, bands, twW, s); break;
    case 2: launch_bwd_rows_nt<M, T, 2>(sin, sout, TT, rho, part_a, part_b, part_lam, B, C, H, R, bands, twW, s); break;
    case 3: launch_bwd_rows_nt<M, T, 3>(sin, sout, TT, rho, part_a, part_b, part_lam, B, C, H, R, bands, twW, s); break;
    default: launch_bwd_rows_nt<M, T, 4>(sin, sout, TT, rho, part_a, part_b, part_lam, B, C, H, R, bands, twW, s); break;
  }
}

// rows per band: two lock-step rounds of the workgroup's SPB row sequences (2 SPB - 2 rows + the two halo rows); knob unroll_bwd_band
static int bwd_rows_band(int H, int W) {
  const int SPB = 256 / (W / 16);
  int R = tune(TUNE_UNROLL_BWD_BAND);
  if (R <= 0) R = 2 * SPB - 2;
  if (R < 2) R = 2;
  return R > H ? H : R;
}

// workgroups (= partial-sum slots) per image of the launch below; 0: planes this kernel does not take
int bwd_rows_slots(int B, int C, int H, int W, int max_slots) {
  if (!(W == 256 || W == 512 || W == 1024) || !pow2_path_available(H, W) || H % 16) return 0;
  const int own = bwd_rows_par_own(B * C, H, W);       // > 0: the row-parallel kernel (dpx_bwd_rows_par.hip) takes the launch
  const int R = own ? own : bwd_rows_band(H, W), bands = (H + R - 1) / R;
  return C * bands <= max_slots ? C * bands : 0;
}

// One backward iteration's row half.  spec_in: the column kernel's output (g_rhs^ of iteration t); spec_out: the row transform of g_x.
// x / rhs: history of iteration t; terms[i].v / lam: iteration t - 1; partial sums as bwd_rows_slots(...) slots per image.
int bwd_rows_fused(const void* spec_in, void* spec_out, const float* x, const float* rhs, const float* rho, const dpx_bwd_term* terms, int nterms,
                   const float* const* a_in, float* const* a_out, float* g_out, int g_acc, float* part_a, float* part_b, float* part_lam, int hist_bf16, int B,
                   int C, int H, int W, const void* table, hipStream_t s) {
  BwdRowTerms TT;
  TT.n = nterms;
  TT.hist_bf16 = hist_bf16;
  TT.x = x;
  TT.rhs = rhs;
  TT.g_out = g_out;
  TT.g_acc = g_acc;
  for (int i = 0; i < DPX_MAX_TERMS; ++i) TT.t[i] = BwdRowTerm{DPX_LIN_IDENTITY, DPX_PROX_NONNEG, 0.f, nullptr, nullptr, nullptr, nullptr};
  for (int i = 0; i < nterms; ++i) TT.t[i] = BwdRowTerm{terms[i].linop, terms[i].prox, terms[i].alpha, terms[i].lam, terms[i].v, a_in[i], a_out[i]};
  for (int i = 0; i < nterms; ++i) DPX_REQUIRE(a_in[i] && a_out[i] && terms[i].v, "dpx_admm_unrolled_backward: term %d lacks a plane of the row kernel", i);
  const float2* tw = tw_rows(table);
  if (const int own = bwd_rows_par_own(B * C, H, W))
    return bwd_rows_par_launch((const float2*)spec_in, (float2*)spec_out, TT, rho, part_a, part_b, part_lam, B, C, H, W, (H + own - 1) / own, tw, s);
  const int R = bwd_rows_band(H, W), bands = (H + R - 1) / R;
  switch (W) {
    case 256: launch_bwd_rows<128, 16>((const float2*)spec_in, (float2*)spec_out, TT, rho, part_a, part_b, part_lam, B, C, H, R, bands, tw, s); break;
    case 512: launch_bwd_rows<256, 32>((const float2*)spec_in, (float2*)spec_out, TT, rho, part_a, part_b, part_lam, B, C, H, R, bands, tw, s); break;
    default: launch_bwd_rows<512, 64>((const float2*)spec_in, (float2*)spec_out, TT, rho, part_a, part_b, part_lam, B, C, H, R, bands, tw, s); break;
  }
  return launch_status("dpx_admm_unrolled_backward");
}

}  // namespace dpx
