// The row pass of the two-kernel ADMM iteration for launches with FEW planes (one rank's shard of a batch split over 8 GPUs: 1 x 3 x
// 1024 x 1024; the reference's own example is ONE 768 x 1024 RGB image, examples/applications/deconv.py:1-16).
//
// k_iter_rows_seq (dpx_iter.hip) lets one wave walk down a band row by row: HBM streams continuously, which is what 24 planes need.
// Three planes cannot fill the chip that way -- 768 waves of 4-row bands, each a dependent chain of 6 inverse transforms, 5 z-updates
// and 4 forward transforms: 27 us at 1 x 3 x 1024^2 whatever the band length (profiles/r5_shard_probe.log), a quarter of the vector
// units busy.  Here the rows of a band are transformed SIDE BY SIDE: a workgroup of NW waves holds NW * G rows (G = 64 / T rows per
// wave), every T-lane group does ONE inverse transform, ONE z / dual update and ONE forward transform, and the two stencil neighbours
// travel through LDS:
//     phase A   spectrum row q by LDS-DMA -> untangle -> inverse row transform -> x[q] to the exchange buffer           (barrier)
//     phase B   x[q + 1] from the neighbour's buffer -> v = prox(K x + u), u' = K x + u - v, w = v - u' of the grad_H term to LDS   (barrier)
//     phase C   w[q - 1] from the neighbour -> rho' sum K^T (v - u') -> forward row transform -> spectrum row out
// Rows 0 and NW G - 1 of a workgroup are the halo of its band (NW G - 2 own rows: 14 of 16 at 1024-wide rows), i.e. 1.14 x the
// transforms instead of the 1.5 x of 4-row bands, and the dependent chain of a wave is one row long.  Same arithmetic in the same
// order as k_iter_rows_seq (same untangling, fft_reg_tw, z-update expressions, accumulation order of the K^T terms): BIT-identical
// results, so sub-batch chains / shards that fall under this kernel agree with the batch run that does not.
// LDS per wave: the transform scratch (G S slots; after phase A: the w exchange buffer) + the DMA staging of the spectrum row (G M + 32
// slots; after the untangling: the x exchange buffer; in phase C: the forward transform's scratch -- its G M + 32 slots are exactly
// G S when M / T = 8) = 17 KB, 16 waves + twiddles: 141 KB, one workgroup of 1024 threads per CU.
// Reference: one iteration = dprox/algo/admm.py:49-59; x-update proxfn/sum_square.py:123-156.
// The transforms' LDS accesses as base(t) + immediate offset (dpx_fft_reg.h: LdsIdx, the same LDS image): this kernel runs every
// transform once, so the slot arithmetic the streaming kernel hoists out of its row loop would be paid per row here.
#ifndef DPX_FFT_BASEOFF
#define DPX_FFT_BASEOFF 1
#endif
#include "dpx_iter_dev.h"

// Tuning aid (tools/build_variant.sh par_trace -DDPX_PAR_TRACE; never in the shipped library): every wave of the first 256 workgroups
// stamps the 100 MHz real-time counter at the phase boundaries; tools/par_trace.py reads the stamps of the last launch.
#ifdef DPX_PAR_TRACE
__device__ unsigned long long dpx_par_trace_buf[256 * 16 * 10];
#define DPX_STAMP(i)                                                                                                       \
  do {                                                                                                                     \
    if (lane == 0 && blockIdx.x < 256) dpx_par_trace_buf[(blockIdx.x * 16 + wave) * 10 + (i)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
extern "C" int dpx_dbg_par_trace(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(dpx_par_trace_buf), (size_t)n * sizeof(unsigned long long));
}
#else
#define DPX_STAMP(i) ((void)0)
#endif

namespace dpx {

template <int M, int T, int NT, bool DUAL, bool VXU, int NW>
__global__ void __launch_bounds__(64 * NW, 1) k_iter_rows_par(const float2* __restrict__ spec_in, float2* __restrict__ spec_out,
                                                             const float2* __restrict__ twW, const float* __restrict__ rho_next,
                                                             float* __restrict__ x_out, int emit_v, int C, int H, int bands, int P, IterTerms TT) {
  // (argument order: the pointers and sizes the first loads need come first -- the leading 16 dwords of the kernel arguments are
  //  preloaded into scalar registers at wave launch, -mllvm -amdgpu-kernarg-preload-count=16 -- the by-value term table last)
  constexpr int V = M / T, G = 64 / T, S = LdsSeq<M>::SLOTS, D = V / 2, RM = M / (V * V);
  constexpr int STG = 64 * V;                           // float2 per staged row set of one wave (G rows of M)
  constexpr int PERWAVE = G * S + STG + 32;
  constexpr int RW = NW * G;                            // rows in flight per workgroup
  static_assert(V == 8 && STG + 32 == G * S, "the staging area doubles as the forward transform's scratch");
  HIP_DYNAMIC_SHARED(float2, smem_pr)
  float2* twl = smem_pr;                                // untangling twiddles exp(-i pi k / M), k < M
  float2* twb = smem_pr + M;                            // pass-B twiddles W_{V*RM}^j, j < 64
  float2* waves = twb + 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane / T, t = lane % T, lbase = lane & ~(T - 1);
  float2* wl = waves + wave * PERWAVE;
  float2* myfft = wl + g * S;                           // inverse transform's scratch; then this row's w (grad_H term) for the row below
  float2* stX = wl + G * S;                             // DMA staging of the spectrum rows; then x of this wave's rows; then forward scratch
  float* stN = (float*)(stX + STG);
  DPX_STAMP(0);
  static_assert(M <= 64 * NW, "one untangling twiddle per thread");

  // xonly (emit_v == 2, no-dual instantiation only): the last pass of a solve() that hands back x alone -- inverse transforms and x stores
  const bool xonly = !DUAL && emit_v == 2;
  const int halo = xonly ? 0 : 1;
  // `bands` bands per plane; the first H % bands of them are one row longer; R + 2 halo <= RW (the launcher's rule)
  const int pl = blockIdx.x / bands, bb = blockIdx.x - pl * bands;
  const int rbase = H / bands, rrem = H - rbase * bands;
  const int r0 = bb * rbase + (bb < rrem ? bb : rrem);
  const int R = rbase + (bb < rrem ? 1 : 0);
  const int bi = pl / C;
  const float rho = rho_next ? rho_next[bi] : 0.f;
  float lamv[NT];
  int hterm = -1;
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    lamv[i] = TT.t[i].lam ? TT.t[i].lam[bi] * TT.t[i].alpha : 0.f;
    if (TT.t[i].linop == DPX_LIN_GRAD_H) hterm = i;
  }
  const int q = wave * G + g;                           // this group's row: image row r0 - halo + q
  const int qlast = R + 2 * halo - 1;                   // last live row of the workgroup
  const bool a_live = q <= qlast;
  const int qc = a_live ? q : qlast;                    // (idle groups repeat the last row: in-bounds addresses, nothing stored)
  int h = r0 - halo + qc;
  h = h < 0 ? h + H : (h >= H ? h - H : h);
  const bool wave_live = wave * G <= qlast;             // wave-uniform: some group of this wave has a row

  // per-lane element offsets (float2 units), as in k_iter_rows_seq
  const unsigned e0 = 2u * t;
  const unsigned xoff = (unsigned)pl * H * M + (e0 / SPEC_TILE) * H * SPEC_TILE + (e0 % SPEC_TILE);
  const unsigned xstep = (unsigned)(2 * T / SPEC_TILE) * H * SPEC_TILE;
  const unsigned noff = (unsigned)P * H * M + (unsigned)pl * H;
  const unsigned tile_off = (unsigned)pl * H * M + (unsigned)((t % SPEC_TILE) + (t / SPEC_TILE) * H * SPEC_TILE);
  const unsigned tile_step = (unsigned)((T / SPEC_TILE) * H * SPEC_TILE);
  const int pair = lbase | ((T - t) & (T - 1));
  auto stage_idx = [&](int e) { return ((e >> 1) / T) * 128 + g * 2 * T + ((e >> 1) % T) * 2 + (e & 1); };

  // ONE memory round trip in front of phase A: the spectrum row by LDS-DMA first (older than every load the compiler counts, so its
  // waits cover it), then all twiddle loads back to back, the LDS copies behind them
  if (wave_live) {
#pragma unroll
    for (int i = 0; i < D; ++i) dpx_glds16<R_LDX>(spec_in + xoff + (unsigned)h * SPEC_TILE + xstep * i, stX + i * 128);
    dpx_glds4<R_LDX>(spec_in + noff + h, stN);
  }
  const float2 tw_a = twW[tid < M ? tid : 0];
  const float2 tw_b = twW[((tid & 63) * (M / (V * RM)) * 2) % (2 * M)];
  TwRegs<M, T, false> twr;
  twr.load(t, twW, 2);
  twr.twb_ = twb;
  twr.bstride_ = 1;
#ifdef DPX_PAR_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  DPX_STAMP(9);
#endif
  if (tid < M) twl[tid] = tw_a;
  if (tid < 64) twb[tid] = tw_b;
  __syncthreads();                                      // the twiddle copies
  DPX_STAMP(1);
  // ---------------- phase A: inverse row transform of row q ----------------
  float2 xa[V];
  float2 ureg[NT][V];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int m = 0; m < V; ++m) ureg[n][m] = make_float2(0.f, 0.f);
  const bool z_live = !xonly && q <= R;                 // rows r0 - 1 .. r0 + R - 1 get a z / dual update
  const bool own = xonly ? a_live : (q >= 1 && q <= R);
  if (wave_live) {
    dpx_wait_vm<0>();
    DPX_STAMP(2);
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
      // the dual rows of phase B: requested now, the inverse transform covers their latency
      if constexpr (DUAL) {
        if (z_live && TT.u_live) {
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            const float2* urow = (const float2*)TT.t[n].u_in + (unsigned)pl * H * M + (unsigned)h * M + t;
#pragma unroll
            for (int m = 0; m < V; ++m) ureg[n][m] = ld_stream<R_LDU>(urow + m * T);
          }
        }
      }
#pragma unroll
      for (int m = 0; m < V; ++m) {
        const int k = t + m * T;
        const float2 xk = Xk[m], xm = cconj(Xm[m]);
        if (k == 0) {
          xa[m] = make_float2(xk.x + xn, xk.x - xn);
        } else {
          const float2 e = cadd(xk, xm);
          const float2 d = cmulc(csub(xk, xm), twl[k]);
          xa[m] = make_float2(e.x - d.y, e.y + d.x);
        }
      }
    }
    WaveSync()();
    fft_reg_tw<M, T, +1, false>(xa, myfft, t, twr, WaveSync());   // xa[m] = (x[2n], x[2n+1]), n = t + m*T
    DPX_STAMP(3);
    if (x_out && own) {
      const size_t xo = (size_t)pl * H * M + (size_t)h * M + t;
#pragma unroll
      for (int m = 0; m < V; ++m) dpx_emit_pair(x_out, TT.emit_bf16, xo + m * T, xa[m]);
    }
    if (!xonly) {
      float2* xb = stX + g * M + t;                     // (this wave has read its staging area: dpx_wait_lds above)
#pragma unroll
      for (int m = 0; m < V; ++m) xb[m * T] = xa[m];
    }
  }
  if (xonly) return;
  DPX_LDS_BARRIER();
  DPX_STAMP(4);
  // ---------------- phase B: z / dual update of row q (x[q] = xa, x[q + 1] from the neighbour) ----------------
  float2 acc[V], cpost[NT][V], wown[V];
#pragma unroll
  for (int m = 0; m < V; ++m) acc[m] = wown[m] = make_float2(0.f, 0.f);
  const unsigned hz = (unsigned)h;
  if (wave_live) {
    float2 xnx[V];                                      // x[q + 1]
    {
      const int qn = (q + 1 < RW) ? q + 1 : q;
      const float2* xb = waves + (qn / G) * PERWAVE + G * S + (qn % G) * M + t;
#pragma unroll
      for (int m = 0; m < V; ++m) xnx[m] = xb[m * T];
    }
    const float dualf = DUAL ? TT.dual : 0.f;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const IterTerm tm = TT.t[n];
      const float lam = lamv[n];
      float2 d[V];
      if (tm.linop == DPX_LIN_IDENTITY) {
#pragma unroll
        for (int m = 0; m < V; ++m) d[m] = make_float2(fmaf(dualf, ureg[n][m].x, xa[m].x), fmaf(dualf, ureg[n][m].y, xa[m].y));
      } else if (tm.linop == DPX_LIN_GRAD_H) {
#pragma unroll
        for (int m = 0; m < V; ++m) d[m] = make_float2(fmaf(dualf, ureg[n][m].x, xnx[m].x - xa[m].x), fmaf(dualf, ureg[n][m].y, xnx[m].y - xa[m].y));
      } else {                                        // grad_W: x[w+1] - x[w]; pixel 2n+2 is the neighbour lane's .x
#pragma unroll
        for (int m = 0; m < V; ++m) {
          const float nx_same = __shfl(xa[m].x, lbase | ((t + 1) & (T - 1)));
          const float nx_wrap = __shfl(xa[(m + 1) % V].x, lbase);
          const float xr = (t == T - 1) ? nx_wrap : nx_same;
          d[m] = make_float2(fmaf(dualf, ureg[n][m].x, xa[m].y - xa[m].x), fmaf(dualf, ureg[n][m].y, xr - xa[m].y));
        }
      }
      float2 tq[VXU ? V : 1];                              // VXU: t = q + x (the dual in front of this v-update)
      if constexpr (VXU) {
#pragma unroll
        for (int m = 0; m < V; ++m) {
          tq[m] = make_float2(ureg[n][m].x + xa[m].x, ureg[n][m].y + xa[m].y);
          d[m] = make_float2(d[m].x + xa[m].x, d[m].y + xa[m].y);      // (d was K x + q: + x)
        }
      }
      float2 v[V];
      if (tm.prox == DPX_PROX_NORM1) {
        soft_threshold_pairs<V>(d, v, lam);
      } else if (tm.prox == DPX_PROX_NONNEG) {
#pragma unroll
        for (int m = 0; m < V; ++m) v[m] = make_float2(fmaxf(d[m].x, 0.f), fmaxf(d[m].y, 0.f));
      } else {
#pragma unroll
        for (int m = 0; m < V; ++m) v[m] = make_float2(prox1(DPX_PROX_SUMSQ, d[m].x, lam), prox1(DPX_PROX_SUMSQ, d[m].y, lam));
      }
      float2 w[V];
#pragma unroll
      for (int m = 0; m < V; ++m) {
        if constexpr (VXU) {
          const float2 un = csub(tq[m], v[m]);                 // q' = t - v
          w[m] = make_float2(-un.x, -un.y);                    // v - t
          d[m] = un;
        } else {
          const float2 un = csub(d[m], v[m]);
          w[m] = make_float2(fmaf(-dualf, un.x, v[m].x), fmaf(-dualf, un.y, v[m].y));
          d[m] = un;
        }
      }
      if (own) {
        if constexpr (DUAL) {
          float2* uo = (float2*)tm.u_out + (unsigned)pl * H * M + hz * M + t;
#pragma unroll
          for (int m = 0; m < V; ++m) st_stream<R_STU>(uo + m * T, d[m]);
        }
        if (emit_v) {
          const size_t vo = (size_t)pl * H * M + (size_t)hz * M + t;
#pragma unroll
          for (int m = 0; m < V; ++m) dpx_emit_pair(tm.v_out, TT.emit_bf16, vo + m * T, v[m]);
        }
      }
      // this term's contribution to K^T (v - u'): accumulated in term order (k_iter_rows_seq's order); the grad_H term's needs the row
      // above and is formed in phase C, so the terms behind it wait in cpost
      float2 c[V];
      if (tm.linop == DPX_LIN_IDENTITY) {
#pragma unroll
        for (int m = 0; m < V; ++m) c[m] = w[m];
      } else if (tm.linop == DPX_LIN_GRAD_W) {          // adjoint: y[w-1] - y[w]; pixel 2n-1 is the left lane's .y
#pragma unroll
        for (int m = 0; m < V; ++m) {
          const float l_same = __shfl(w[m].y, lbase | ((t + T - 1) & (T - 1)));
          const float l_wrap = __shfl(w[(m + V - 1) % V].y, lbase | (T - 1));
          const float wlft = (t == 0) ? l_wrap : l_same;
          c[m] = make_float2(wlft - w[m].x, w[m].x - w[m].y);
        }
      } else {                                          // grad_H: w goes to the row below through LDS
#pragma unroll
        for (int m = 0; m < V; ++m) {
          wown[m] = w[m];
          c[m] = make_float2(0.f, 0.f);
        }
        if (z_live) {
          float2* wb = myfft + t;
#pragma unroll
          for (int m = 0; m < V; ++m) wb[m * T] = w[m];
        }
      }
#pragma unroll
      for (int m = 0; m < V; ++m) {
        if (hterm >= 0 && n > hterm) cpost[n][m] = c[m];
        else if (n != hterm) acc[m] = cadd(acc[m], c[m]);
      }
    }
  }
  DPX_STAMP(5);
  DPX_LDS_BARRIER();
  DPX_STAMP(6);
  // ---------------- phase C: right-hand-side increment of row q and its forward row transform ----------------
  const bool wave_own = wave * G <= R && wave * G + G - 1 >= 1;      // wave-uniform: some group of this wave owns a row of the band
  if (wave_own && rho_next) {
    if (hterm >= 0) {
      float2 wprev[V];
      const int qp = q >= 1 ? q - 1 : 0;
      const float2* wb = waves + (qp / G) * PERWAVE + (qp % G) * S + t;
#pragma unroll
      for (int m = 0; m < V; ++m) wprev[m] = wb[m * T];
#pragma unroll
      for (int m = 0; m < V; ++m) acc[m] = make_float2(acc[m].x + (wprev[m].x - wown[m].x), acc[m].y + (wprev[m].y - wown[m].y));
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        if (n > hterm) {
#pragma unroll
          for (int m = 0; m < V; ++m) acc[m] = cadd(acc[m], cpost[n][m]);
        }
      }
    }
    float2 z[V];
#pragma unroll
    for (int m = 0; m < V; ++m) z[m] = make_float2(rho * acc[m].x, rho * acc[m].y);
    if (TT.rhs_out && own) {
      const size_t ro = (size_t)pl * H * M + (size_t)hz * M + t;
#pragma unroll
      for (int m = 0; m < V; ++m) dpx_emit_pair(TT.rhs_out, TT.emit_bf16, ro + m * T, z[m]);
    }
    float2* fwd = stX + g * S;                          // (the x exchange buffer: read by the row above before the barrier)
    WaveSync()();
    fft_reg_tw<M, T, -1, false>(z, fwd, t, twr, WaveSync());
    DPX_STAMP(7);
    float2* out = spec_out + tile_off + hz * SPEC_TILE;
#pragma unroll
    for (int m = 0; m < V; ++m) {
      const float2 got = make_float2(__shfl(z[V - 1 - m].x, pair), __shfl(z[V - 1 - m].y, pair));
      const float2 zm = cconj(t == 0 ? z[(V - m) % V] : got);
      const int k = t + m * T;
      const float2 zk = z[m];
      float2 Xo;
      if (k == 0) {
        Xo = make_float2(zk.x + zk.y, 0.f);
        if (own) st_stream<R_STX>(spec_out + noff + hz, make_float2(zk.x - zk.y, 0.f));
      } else {
        const float2 e = cscale(cadd(zk, zm), 0.5f);
        const float2 d = cscale(csub(zk, zm), 0.5f);
        Xo = cadd(e, cmul(make_float2(d.y, -d.x), twl[k]));
      }
      if (own) st_stream<R_STX>(out + tile_step * m, Xo);
    }
    DPX_STAMP(8);
  }
}

template <int M, int T, int NT, bool DUAL, bool VXU>
static void launch_par_d(const float2* sin, float2* sout, const IterTerms& TT, const float* rho_next, float* x_out, int emit_v, int C, int H, int P,
                         const float2* twW, hipStream_t s) {
  // 16 waves (128 registers each) for the one- and two-term forms; three or four terms and the v, x, u order keep more rows of duals / split
  // variables in registers: 8 waves of 256 registers (no scratch in any instantiation: tools/spill_check.py)
  constexpr int NW = (NT >= 3 || VXU) ? 8 : 16, G = 64 / T, S = M + M / 16, V = M / T, RW = NW * G;
  const size_t sh = (size_t)(M + 64 + NW * (G * S + 64 * V + 32)) * sizeof(float2);
  static unsigned long long attr = 0;                   // one bit per device: the 140 KB opt-in is a per-device function attribute
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!(attr >> (dev & 63) & 1ull)) {
    hipFuncSetAttribute((const void*)k_iter_rows_par<M, T, NT, DUAL, VXU, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    attr |= 1ull << (dev & 63);
  }
  const bool xonly = !DUAL && emit_v == 2;
  const int rows = xonly ? RW : RW - 2;                 // own rows per workgroup
  const int bands = (H + rows - 1) / rows;
  DPX_LAUNCH(VXU ? "k_iter_rows_par_vxu" : (DUAL ? "k_iter_rows_par" : "k_iter_rows_par_nodual"), (k_iter_rows_par<M, T, NT, DUAL, VXU, NW>),
             dim3(P * bands), dim3(64 * NW), sh, s, sin, sout, twW, rho_next, x_out, emit_v, C, H, bands, P, TT);
}
template <int M, int T, int NT>
static void launch_par_nt(const float2* sin, float2* sout, const IterTerms& TT, const float* rho_next, float* x_out, int emit_v, int C, int H, int P,
                          const float2* twW, hipStream_t s) {
  const bool keep_dual = false;
  // (the same choice of instantiation as launch_iter_rows_seq_nt, dpx_iter.hip)
  if (emit_v == 2 && x_out && !rho_next) launch_par_d<M, T, NT, false, false>(sin, sout, TT, rho_next, x_out, 2, C, H, P, twW, s);
  else if (TT.vxu) launch_par_d<M, T, NT, true, true>(sin, sout, TT, rho_next, x_out, emit_v ? 1 : 0, C, H, P, twW, s);
  else if (TT.dual == 0.f && !keep_dual) launch_par_d<M, T, NT, false, false>(sin, sout, TT, rho_next, x_out, emit_v ? 1 : 0, C, H, P, twW, s);
  else launch_par_d<M, T, NT, true, false>(sin, sout, TT, rho_next, x_out, emit_v ? 1 : 0, C, H, P, twW, s);
}
template <int M, int T>
static void launch_par(const float2* sin, float2* sout, const IterTerms& TT, const float* rho_next, float* x_out, int emit_v, int C, int H, int P,
                       const float2* twW, hipStream_t s) {
  switch (TT.n) {
    case 1: launch_par_nt<M, T, 1>(sin, sout, TT, rho_next, x_out, emit_v, C, H, P, twW, s); break;
    case 2: launch_par_nt<M, T, 2>(sin, sout, TT, rho_next, x_out, emit_v, C, H, P, twW, s); break;
    case 3: launch_par_nt<M, T, 3>(sin, sout, TT, rho_next, x_out, emit_v, C, H, P, twW, s); break;
    default: launch_par_nt<M, T, 4>(sin, sout, TT, rho_next, x_out, emit_v, C, H, P, twW, s); break;
  }
}

// Row-parallel kernel for launches that cannot fill the chip with band walkers: P * H rows up to `iter_par_max_rows` (knob; the
// library's rule: 8192 rows of 1024 / 512 / 256 pixels -- 1 .. 2 rounds of one 16-wave workgroup per CU), or forced (iter_rows = 3).
bool launch_iter_rows_par(const float2* sin, float2* sout, const IterTerms& TT, const float* rho_next, float* x_out, int emit_v, int C, int H, int W,
                          int P, const float2* twW, hipStream_t s, bool forced) {
  if (W != 1024 && W != 512 && W != 256) return false;
  if (!forced) {
    const int knob = tune(TUNE_ITER_PAR_MAX_ROWS);
    const long max_rows = knob > 0 ? knob : (knob < 0 ? 0 : 8192);
    if ((long)P * H > max_rows) return false;
    // (a few 256-wide planes -- config 1 -- stay on the lock-step kernel's 8-row bands: 0.580 ms per 20-iteration solve against 0.606 here)
    if (W <= 256 && (long)P * H <= 4096 && knob == 0) return false;
  }
  switch (W) {
    case 256: launch_par<128, 16>(sin, sout, TT, rho_next, x_out, emit_v, C, H, P, twW, s); break;
    case 512: launch_par<256, 32>(sin, sout, TT, rho_next, x_out, emit_v, C, H, P, twW, s); break;
    default: launch_par<512, 64>(sin, sout, TT, rho_next, x_out, emit_v, C, H, P, twW, s); break;
  }
  return true;
}

}  // namespace dpx
