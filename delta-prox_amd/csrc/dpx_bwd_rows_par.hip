// The row half of the unrolled ADMM backward iteration with the rows of a band SIDE BY SIDE (the backward mirror of k_iter_rows_par,
// dpx_iter_par.hip): BASELINE config 5 differentiates 4 x 3 x 512 x 512 -- twelve planes, which k_bwd_rows' lock-step bands (two rounds
// of eight row sequences per workgroup, LDS rings, two workgroup barriers per round) cannot spread over the chip.  Here a workgroup of
// 16 waves holds 16 * G rows (G = 64 / T rows per wave), every T-lane group does ONE inverse transform of g_rhs^, the rhs stage of
// iteration t and the z stage of iteration t - 1 on its row, and ONE forward transform of g_x; the stencil neighbours travel through LDS:
//     phase A   spectrum row q by LDS-DMA -> untangle -> inverse row transform -> g[q] to the exchange buffer             (barrier)
//     phase B   g[q - 1], g[q + 1] from the neighbours -> the two rho reductions; g_v, g_u, g_d of every term (a_i, v_i, x, rhs from
//               HBM), g_d of the grad_H term to LDS                                                                          (barrier)
//     phase C   g_d[q - 1] from the neighbour -> g_x = sum K^T g_d -> forward row transform -> spectrum row out
// Same expressions as k_bwd_rows (dpx_bwd_dev.h); the partial sums of d/d rho_t and d/d lam_i are grouped differently (one slot per
// workgroup here as there, but a workgroup is another set of rows), i.e. equal to round-off.
// Reference: torch.autograd through UnrolledSolver (algo/specialization/unroll.py:21-58, algo/admm.py:49-59).
#ifndef DPX_FFT_BASEOFF
#define DPX_FFT_BASEOFF 1
#endif
#include "dpx_bwd_dev.h"

namespace dpx {

// Registers (round 6): at most 128 per lane in every instantiation (94 - 120 used), i.e. TWO 8-wave workgroups per CU.  Round 5's build used 158 -
// 192 and ran one workgroup per CU -- config 5's twelve 512-row planes are 444 workgroups: two rounds, 38.8 us per launch against the forward
// kernel's 21.4.  What brought it down, without touching the arithmetic: the two double-precision rho sums are reduced over the wave and parked in the
// workgroup's slot array right behind their loop instead of living through the term loop (this alone: 137 -> 96), the history and incoming-adjoint
// values are loaded two values at a time, and the forward transform's twiddle registers are loaded again in front of phase C.
template <int M, int T, int NT, bool HB, int NW>
__global__ void __launch_bounds__(64 * NW, 4) k_bwd_rows_par(const float2* __restrict__ spec_in, float2* __restrict__ spec_out,
                                                            const float2* __restrict__ twW, const float* __restrict__ rho_b,
                                                            float* __restrict__ part_a, float* __restrict__ part_b, float* __restrict__ part_lam,
                                                            int B, int C, int H, int bands, int P, BwdRowTerms TT) {
  constexpr int V = M / T, G = 64 / T, S = LdsSeq<M>::SLOTS, D = V / 2, RM = M / (V * V);
  constexpr int STG = 64 * V, PERWAVE = G * S + STG + 32, RW = NW * G;
  static_assert(V == 8 && STG + 32 == G * S, "row-parallel geometry");
  HIP_DYNAMIC_SHARED(float2, smem_bp)
  __shared__ double red[NW * (2 + DPX_MAX_TERMS)];
  float2* twl = smem_bp;
  float2* twb = smem_bp + M;
  float2* waves = twb + 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane / T, t = lane % T, lbase = lane & ~(T - 1);
  float2* wl = waves + wave * PERWAVE;
  float2* myfft = wl + g * S;                           // inverse transform's scratch; then g_d of the grad_H term for the row below
  float2* stX = wl + G * S;                             // DMA staging; then g of this wave's rows; then the forward transform's scratch
  float* stN = (float*)(stX + STG);

  const int pl = blockIdx.x / bands, band = blockIdx.x - pl * bands;
  const int rbase = H / bands, rrem = H - rbase * bands;
  const int r0 = band * rbase + (band < rrem ? band : rrem);
  const int R = rbase + (band < rrem ? 1 : 0);        // own rows; R + 2 <= RW (the launcher's rule)
  const int bi = pl / C, ci = pl - bi * C;
  const float rho = rho_b[bi];
  int hterm = -1, nW = 0, nH = 0;
  float cI = 0.f;
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    if (TT.t[i].linop == DPX_LIN_GRAD_H) hterm = i, ++nH;
    else if (TT.t[i].linop == DPX_LIN_GRAD_W) ++nW;
    else cI += 1.f;
  }
  const int q = wave * G + g;                           // this group's row: image row r0 - 1 + q
  const int qlast = R + 1;
  const int qc = q <= qlast ? q : qlast;                // (idle groups repeat the last row: in-bounds addresses, nothing stored)
  int h = r0 - 1 + qc;
  h = h < 0 ? h + H : (h >= H ? h - H : h);
  const bool wave_live = wave * G <= qlast;
  const bool z_live = q <= R;                           // rows r0 - 1 .. r0 + R - 1 run the two stages (row 0: for its g_d only)
  const bool own = q >= 1 && q <= R;
  const size_t rowz = (size_t)pl * H * M + (size_t)h * M;

  const unsigned e0 = 2u * t;
  const unsigned xoff = (unsigned)pl * H * M + (e0 / SPEC_TILE) * H * SPEC_TILE + (e0 % SPEC_TILE);
  const unsigned xstep = (unsigned)(2 * T / SPEC_TILE) * H * SPEC_TILE;
  const unsigned noff = (unsigned)P * H * M + (unsigned)pl * H;
  const int lnext = lbase | ((t + 1) & (T - 1)), lprev = lbase | ((t + T - 1) & (T - 1));
  auto stage_idx = [&](int e) { return ((e >> 1) / T) * 128 + g * 2 * T + ((e >> 1) % T) * 2 + (e & 1); };

  if (wave_live) {
#pragma unroll
    for (int i = 0; i < D; ++i) dpx_glds16<0>(spec_in + xoff + (unsigned)h * SPEC_TILE + xstep * i, stX + i * 128);
    dpx_glds4<0>(spec_in + noff + h, stN);
  }
  for (int i = tid + 64 * NW; i < M; i += 64 * NW) twl[i] = twW[i];      // (M > 64 NW: the eight-wave build at 1024-wide rows)
  const float2 tw_a = twW[tid < M ? tid : 0];
  const float2 tw_b = twW[((tid & 63) * (M / (V * RM)) * 2) % (2 * M)];
  TwRegs<M, T, false> twr;
  twr.load(t, twW, 2);
  twr.twb_ = twb;
  twr.bstride_ = 1;
  if (tid < M) twl[tid] = tw_a;
  if (tid < 64) twb[tid] = tw_b;
  __syncthreads();
  // ---------------- phase A: inverse row transform of row q ----------------
  float2 ga[V];
#pragma unroll
  for (int m = 0; m < V; ++m) ga[m] = make_float2(0.f, 0.f);
  if (wave_live) {
    dpx_wait_vm<0>();
    {
      float2 Xk[V], Xm[V];
#pragma unroll
      for (int m = 0; m < V; ++m) {
        const int k = t + m * T;
        Xk[m] = stX[stage_idx(k)];
        Xm[m] = stX[stage_idx((M - k) % M)];
      }
      const float xn = stN[g * T];
      dpx_wait_lds();
#pragma unroll
      for (int m = 0; m < V; ++m) {
        const int k = t + m * T;
        const float2 xk = Xk[m], xm = cconj(Xm[m]);
        if (k == 0) {
          ga[m] = make_float2(xk.x + xn, xk.x - xn);
        } else {
          const float2 e = cadd(xk, xm);
          const float2 d = cmulc(csub(xk, xm), twl[k]);
          ga[m] = make_float2(e.x - d.y, e.y + d.x);
        }
      }
    }
    WaveSync()();
    fft_reg_tw<M, T, +1, false>(ga, myfft, t, twr, WaveSync());   // ga[m] = (g[2n], g[2n+1]), n = t + m*T
    if (TT.g_out && own) {
      float2* go = (float2*)TT.g_out + rowz;
      if (TT.g_acc) {
#pragma unroll
        for (int m = 0; m < V; ++m) go[t + m * T] = cadd(go[t + m * T], ga[m]);
      } else {
#pragma unroll
        for (int m = 0; m < V; ++m) go[t + m * T] = ga[m];
      }
    }
    float2* gb = stX + g * M + t;
#pragma unroll
    for (int m = 0; m < V; ++m) gb[m * T] = ga[m];
  }
  DPX_LDS_BARRIER();
  // ---------------- phase B: the two stages on row q (g[q] = ga; g[q - 1], g[q + 1] from the neighbours) ----------------
  float2 acc[V];
#pragma unroll
  for (int m = 0; m < V; ++m) acc[m] = make_float2(0.f, 0.f);
  // the partial sums of d/d rho_t are signed products that cancel to ~1e-5 of their magnitude: summed in double inside the workgroup
  // (a dozen additions per lane and row), one rounding per slot
  double acc_a = 0.0, acc_b = 0.0;
  float lsum[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) lsum[i] = 0.f;
  if (wave_live) {
    const int qn = (q + 1 < RW) ? q + 1 : q;
    const float2* gnb = waves + (qn / G) * PERWAVE + G * S + (qn % G) * M + t;      // g[q + 1] (read where it is used: registers)
    // -- the two rho reductions of iteration t over the band's own rows: <g, rhs> and <g, L x> = <L g, x>, L = sum_i K_i^T K_i
    {
      const int qp = q >= 1 ? q - 1 : 0;
      const float2* gpb = waves + (qp / G) * PERWAVE + G * S + (qp % G) * M + t;
      constexpr int QS = 2;                   // values per batch of history loads (fp32 history: one -- 128 registers)
#pragma unroll
      for (int m0 = 0; m0 < V; m0 += QS) {
        float2 xr[QS], rr[QS];
#pragma unroll
        for (int k = 0; k < QS; ++k) {
          xr[k] = hist_pair<HB>(TT.x, rowz + t + (m0 + k) * T);
          rr[k] = hist_pair<HB>(TT.rhs, rowz + t + (m0 + k) * T);
        }
#pragma unroll
        for (int k = 0; k < QS; ++k) {
          const int m = m0 + k;
          float2 lg = make_float2(cI * ga[m].x, cI * ga[m].y);
          if (nW) {
            const float r_same = __shfl(ga[m].x, lnext), r_wrap = __shfl(ga[(m + 1) % V].x, lbase);
            const float l_same = __shfl(ga[m].y, lprev), l_wrap = __shfl(ga[(m + V - 1) % V].y, lbase | (T - 1));
            const float right = (t == T - 1) ? r_wrap : r_same, left = (t == 0) ? l_wrap : l_same;
            lg.x += (float)nW * (2.f * ga[m].x - left - ga[m].y);
            lg.y += (float)nW * (2.f * ga[m].y - ga[m].x - right);
          }
          if (nH) {
            const float2 gp = gpb[m * T], gn = gnb[m * T];
            lg.x += (float)nH * (2.f * ga[m].x - gp.x - gn.x);
            lg.y += (float)nH * (2.f * ga[m].y - gp.y - gn.y);
          }
          const float pa = fmaf(lg.y, xr[k].y, lg.x * xr[k].x), pb = fmaf(ga[m].y, rr[k].y, ga[m].x * rr[k].x);
          acc_a += own ? (double)pa : 0.0;
          acc_b += own ? (double)pb : 0.0;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // (the two double sums leave the registers here: reduced over the wave and parked in the workgroup's slot array)
    {
      double va = acc_a, vb = acc_b;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        va += __shfl_xor(va, o);
        vb += __shfl_xor(vb, o);
      }
      if (lane == 0) {
        red[wave * (2 + DPX_MAX_TERMS) + 0] = va;
        red[wave * (2 + DPX_MAX_TERMS) + 1] = vb;
      }
    }
    __builtin_amdgcn_sched_barrier(0);                  // (keeps the terms' loads from being hoisted above: registers)
    // -- g_v, g_u, g_d of every term; g_x = sum K^T g_d
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const BwdRowTerm tm = TT.t[i];
      const float lam = tm.lam ? tm.lam[bi] * tm.alpha : 0.f;
      const float sq = 1.f / (1.f + 2.f * lam);
      float2 w[V];
      if (tm.linop == DPX_LIN_IDENTITY) {
#pragma unroll
        for (int m = 0; m < V; ++m) w[m] = ga[m];
      } else if (tm.linop == DPX_LIN_GRAD_H) {
#pragma unroll
        for (int m = 0; m < V; ++m) {
          const float2 gn = gnb[m * T];
          w[m] = make_float2(gn.x - ga[m].x, gn.y - ga[m].y);
        }
      } else {                                          // grad_W: g[w+1] - g[w]; pixel 2n+2 is the neighbour lane's .x
#pragma unroll
        for (int m = 0; m < V; ++m) {
          const float nx_same = __shfl(ga[m].x, lnext);
          const float nx_wrap = __shfl(ga[(m + 1) % V].x, lbase);
          const float gr = (t == T - 1) ? nx_wrap : nx_same;
          w[m] = make_float2(ga[m].y - ga[m].x, gr - ga[m].y);
        }
      }
      // the prox stage in quarters of the row's values: a quarter's incoming adjoint and history values in registers at a time (the sum `lt`
      // runs over m = 0 .. V - 1 in order, as bwd_gd_row does)
      float lt = 0.f;
      constexpr int QP = 2;
#pragma unroll
      for (int m0 = 0; m0 < V; m0 += QP) {
        float2 av[QP], vv[QP];
#pragma unroll
        for (int k = 0; k < QP; ++k) {
          av[k] = ((const float2*)tm.a_in)[rowz + t + (m0 + k) * T];
          vv[k] = hist_pair<HB>(tm.v, rowz + t + (m0 + k) * T);
        }
#pragma unroll
        for (int k = 0; k < QP; ++k) {
          const int m = m0 + k;
          if (tm.prox == DPX_PROX_NORM1) {
            w[m].x = bwd_gd<1>(sq, rho * w[m].x, av[k].x, vv[k].x, lt);
            w[m].y = bwd_gd<1>(sq, rho * w[m].y, av[k].y, vv[k].y, lt);
          } else if (tm.prox == DPX_PROX_NONNEG) {
            w[m].x = bwd_gd<2>(sq, rho * w[m].x, av[k].x, vv[k].x, lt);
            w[m].y = bwd_gd<2>(sq, rho * w[m].y, av[k].y, vv[k].y, lt);
          } else {
            w[m].x = bwd_gd<0>(sq, rho * w[m].x, av[k].x, vv[k].x, lt);
            w[m].y = bwd_gd<0>(sq, rho * w[m].y, av[k].y, vv[k].y, lt);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      lsum[i] += own ? lt : 0.f;
      if (own) {
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
          const float wlft = (t == 0) ? l_wrap : l_same;
          acc[m] = make_float2(acc[m].x + (wlft - w[m].x), acc[m].y + (w[m].x - w[m].y));
        }
      } else {                                          // grad_H: g_d goes to the row below through LDS, its adjoint is formed in phase C
        float2* wb = myfft + t;                         // (every live group writes: its own copy is read back in phase C)
#pragma unroll
        for (int m = 0; m < V; ++m) wb[m * T] = w[m];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  DPX_LDS_BARRIER();
  // ---------------- phase C: g_x of row q and its forward row transform ----------------
  const bool wave_own = wave * G <= R && wave * G + G - 1 >= 1;
  if (wave_own) {
    if (hterm >= 0) {
      const int qp = q >= 1 ? q - 1 : 0;
      const float2* wb = waves + (qp / G) * PERWAVE + (qp % G) * S + t;
#pragma unroll
      for (int m = 0; m < V; ++m) {
        const float2 up = wb[m * T], cu = myfft[t + m * T];
        acc[m] = make_float2(acc[m].x + (up.x - cu.x), acc[m].y + (up.y - cu.y));
      }
    }
    float2* fwd = stX + g * S;
    int t_c = t;
    DPX_OPAQUE(t_c);                                    // (phase A's twiddle registers are not carried through phase B: loaded again, 14 L2 hits)
    TwRegs<M, T, false> twc;
    twc.load(t_c, twW, 2);
    twc.twb_ = twb;
    twc.bstride_ = 1;
    WaveSync()();
    fft_reg_tw<M, T, -1, false>(acc, fwd, t, twc, WaveSync());
    const unsigned tile_off_c = (unsigned)pl * H * M + (unsigned)((t_c % SPEC_TILE) + (t_c / SPEC_TILE) * H * SPEC_TILE);
    const unsigned tile_step_c = (unsigned)((T / SPEC_TILE) * H * SPEC_TILE);
    const int pair_c = lbase | ((T - t_c) & (T - 1));
    float2* out = spec_out + tile_off_c + (unsigned)h * SPEC_TILE;
#pragma unroll
    for (int m = 0; m < V; ++m) {
      const float2 got = make_float2(__shfl(acc[V - 1 - m].x, pair_c), __shfl(acc[V - 1 - m].y, pair_c));
      const float2 zm = cconj(t == 0 ? acc[(V - m) % V] : got);
      const int k = t + m * T;
      const float2 zk = acc[m];
      float2 Xo;
      if (k == 0) {
        Xo = make_float2(zk.x + zk.y, 0.f);
        if (own) spec_out[noff + h] = make_float2(zk.x - zk.y, 0.f);
      } else {
        const float2 e = cscale(cadd(zk, zm), 0.5f);
        const float2 d = cscale(csub(zk, zm), 0.5f);
        Xo = cadd(e, cmul(make_float2(d.y, -d.x), twl[k]));
      }
      if (own) out[tile_step_c * m] = Xo;
    }
  }
  // ---- the workgroup's partial sums, one slot each (the finishing launch adds the slots of an image in index order) ----
  {
    auto wave_sum_d = [](double v) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      return v;
    };
    double vals[2 + NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) vals[2 + i] = wave_sum_d((double)lsum[i]);
    if (lane == 0) {
      if (!wave_live) red[wave * (2 + DPX_MAX_TERMS) + 0] = red[wave * (2 + DPX_MAX_TERMS) + 1] = 0.0;     // (idle waves: their two slots were not written above)
#pragma unroll
      for (int e = 2; e < 2 + NT; ++e) red[wave * (2 + DPX_MAX_TERMS) + e] = vals[e];
    }
    __syncthreads();
    if (tid < 2 + NT) {
      constexpr int W4 = 2 + DPX_MAX_TERMS;
      double sumd = 0.0;
#pragma unroll
      for (int w = 0; w < NW; ++w) sumd += red[w * W4 + tid];
      const float sum = (float)sumd;
      const int nblk = C * bands;
      const long slot = (long)ci * bands + band;
      if (tid == 0) part_a[(long)bi * nblk + slot] = -sum;
      else if (tid == 1) part_b[(long)bi * nblk + slot] = sum;
      else part_lam[((long)(tid - 2) * B + bi) * nblk + slot] = sum * TT.t[tid - 2].alpha;
    }
  }
}

#ifndef DPX_BWD_PAR_NW
#define DPX_BWD_PAR_NW 8           // waves per workgroup (two workgroups per CU; 16: one)
#endif
constexpr int BWD_PAR_NW = DPX_BWD_PAR_NW;

template <int M, int T, int NT, bool HB>
static void launch_bp_hb(const float2* sin, float2* sout, const BwdRowTerms& TT, const float* rho, float* part_a, float* part_b, float* part_lam,
                         int B, int C, int H, int bands, const float2* twW, hipStream_t s) {
  constexpr int NW = BWD_PAR_NW, G = 64 / T, S = M + M / 16, V = M / T;
  const size_t sh = (size_t)(M + 64 + NW * (G * S + 64 * V + 32)) * sizeof(float2);
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute((const void*)k_bwd_rows_par<M, T, NT, HB, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    attr = true;
  }
  const int P = B * C;
  DPX_LAUNCH("k_bwd_rows_par", (k_bwd_rows_par<M, T, NT, HB, NW>), dim3(P * bands), dim3(64 * NW), sh, s, sin, sout, twW, rho, part_a, part_b, part_lam, B,
             C, H, bands, P, TT);
}
template <int M, int T, int NT>
static void launch_bp_nt(const float2* sin, float2* sout, const BwdRowTerms& TT, const float* rho, float* part_a, float* part_b, float* part_lam,
                         int B, int C, int H, int bands, const float2* twW, hipStream_t s) {
  if (TT.hist_bf16) launch_bp_hb<M, T, NT, true>(sin, sout, TT, rho, part_a, part_b, part_lam, B, C, H, bands, twW, s);
  else launch_bp_hb<M, T, NT, false>(sin, sout, TT, rho, part_a, part_b, part_lam, B, C, H, bands, twW, s);
}
template <int M, int T>
static void launch_bp(const float2* sin, float2* sout, const BwdRowTerms& TT, const float* rho, float* part_a, float* part_b, float* part_lam,
                      int B, int C, int H, int bands, const float2* twW, hipStream_t s) {
  switch (TT.n) {
    case 1: launch_bp_nt<M, T, 1>(sin, sout, TT, rho, part_a, part_b, part_lam, B, C, H, bands, twW, s); break;
    case 2: launch_bp_nt<M, T, 2>(sin, sout, TT, rho, part_a, part_b, part_lam, B, C, H, bands, twW, s); break;
    case 3: launch_bp_nt<M, T, 3>(sin, sout, TT, rho, part_a, part_b, part_lam, B, C, H, bands, twW, s); break;
    default: launch_bp_nt<M, T, 4>(sin, sout, TT, rho, part_a, part_b, part_lam, B, C, H, bands, twW, s); break;
  }
}

// own rows per workgroup (0: the lock-step kernel keeps the launch): launches of up to `unroll_bwd_par_max_rows` rows (planes x H; the
// library's rule: 12288 -- config 5's twelve 512-row planes in one round of 16-wave workgroups), knob < 0 = never
int bwd_rows_par_own(int P, int H, int W) {
  if (!(W == 256 || W == 512 || W == 1024)) return 0;
  const int knob = tune(TUNE_UNROLL_BWD_PAR_MAX_ROWS);
  const long max_rows = knob > 0 ? knob : (knob < 0 ? 0 : 12288);
  if ((long)P * H > max_rows) return 0;
  return BWD_PAR_NW * (64 / (W / 16)) - 2;
}
int bwd_rows_par_launch(const float2* sin, float2* sout, const BwdRowTerms& TT, const float* rho, float* part_a, float* part_b, float* part_lam, int B,
                        int C, int H, int W, int bands, const float2* twW, hipStream_t s) {
  switch (W) {
    case 256: launch_bp<128, 16>(sin, sout, TT, rho, part_a, part_b, part_lam, B, C, H, bands, twW, s); break;
    case 512: launch_bp<256, 32>(sin, sout, TT, rho, part_a, part_b, part_lam, B, C, H, bands, twW, s); break;
    default: launch_bp<512, 64>(sin, sout, TT, rho, part_a, part_b, part_lam, B, C, H, bands, twW, s); break;
  }
  return launch_status("dpx_admm_unrolled_backward");
}

}  // namespace dpx
