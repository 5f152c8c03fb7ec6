// The fused ADMM iteration for power-of-two planes: TWO kernels per iteration.
//
//   k_iter_cols (= k_cols_p2<OP_SOLVE>, dpx_fft_pow2.hip): column FFT -> (+ data spectrum, / denominator) -> inverse
//                 column FFT, spectrum in, spectrum out.
//   k_iter_rows (this file): for a band of R image rows of one plane
//        inverse row FFT (finishes the x-update)                     x = irFFT2[...]          sum_square.py:150-152
//     -> z / dual update of every Psi term on the fresh rows          v = prox(Kx + u), u += Kx - v   admm.py:54-57
//     -> right-hand side increment of the NEXT x-update               rho' * sum K_i^T (v_i - u_i)    sum_square.py:126-135
//     -> forward row FFT of that increment                            first half of the next rFFT2
//   The image x and the split variables v_i never touch HBM inside the loop (they are emitted on request: last
//   iteration or a user callback); per iteration and element the loop moves one spectrum in, one spectrum
//   out, u_i in and u_i out per term.  Reference: one iteration = dprox/algo/admm.py:49-59.
//
// Row dependencies (grad along H couples row h with h+1 in K and h-1 in K^T) are handled inside a workgroup:
// its SPB row-sequences advance through the band in lock step, a ring of SPB+1 rows of x and of (v-u) lives in
// LDS, and one halo row above / below the band is recomputed (R+2 inverse transforms for R rows).
#include <cstdlib>
#include <cstring>
#include <type_traits>

// The hand-written VOP3P complex primitives (dpx_common.h: DPX_PK_ASM) are OFF in this file: the same arithmetic in the same rounding
// order from the C forms (bit-identical results).  Measured, round 4, alternating runs on one box at 8x3x1024^2: k_iter_rows_seq
// 110.1 - 111.4 us with the asm forms (1850 vector instructions, 193 VGPRs) against 108.0 us without (2028, 224) -- this kernel sits on
// the memory system's copy rate, and hipcc schedules the opaque asm pairs worse than its own code; the column kernel
// (dpx_fft_pow2.hip) gains 2 - 3 us from them.
#ifndef DPX_PK_ASM
#define DPX_PK_ASM 0
#endif
#include "dpx_fft_reg.h"
#include "dpx_iter_dev.h"

namespace dpx {

// M = W/2 complex points per row, T threads per row, SPB = 256/T rows in flight per workgroup, NT terms.
// Global loads are issued one phase ahead of their use (u rows at the top of phase A, the next spectrum row at
// the top of phase C) so that the transforms cover the HBM latency; 2 workgroups (8 waves) share a CU.
// (1024-wide rows with three or four terms: one wave per SIMD -- 512 registers -- instead of 14 / 28 spilled ones; tools/spill_check.py)
template <int M, int T, int NT>
__global__ void __launch_bounds__(256, (M == 512 && NT >= 3) ? 1 : 2) k_iter_rows(const float2* __restrict__ spec_in, float2* __restrict__ spec_out, IterTerms TT,
                                                    const float* __restrict__ rho_next, float* __restrict__ x_out, int emit_v,
                                                    int C, int H, int R, int P, const float2* __restrict__ twW) {
  constexpr int V = M / T, SPB = 256 / T, S = LdsSeq<M>::SLOTS, RING = SPB + 1;
  HIP_DYNAMIC_SHARED(float2, smem_it)
  float2* fft_lds = smem_it;                          // SPB * S
  float2* xring = smem_it + SPB * S;                  // RING rows of M float2 (pixel pairs)
  float2* wring = xring + RING * M;                   // RING rows of (v-u) of the grad_H term
  const int tid = threadIdx.x, j = tid / T, t = tid % T;
  const int lane = tid & 63, lbase = lane & ~(T - 1);
  const int bands = H / R;
  const int pl = blockIdx.x / bands, r0 = (blockIdx.x - pl * bands) * R;
  const int bi = pl / C;
  const size_t plane_px = (size_t)pl * H * (2 * M);
  const float2* sin_main = spec_in + (size_t)pl * H * M;
  const float2* sin_side = spec_in + (size_t)P * H * M + (size_t)pl * H;
  float2* sout_main = spec_out + (size_t)pl * H * M;
  float2* sout_side = spec_out + (size_t)P * H * M + (size_t)pl * H;
  const unsigned tile_off = (unsigned)((t % SPEC_TILE) + (t / SPEC_TILE) * H * SPEC_TILE);   // bin t of a row in the tile-major spectrum
  const unsigned tile_step = (unsigned)((T / SPEC_TILE) * H * SPEC_TILE);            // bin t + m*T
  const float rho = rho_next ? rho_next[bi] : 0.f;
  float2* myfft = fft_lds + j * S;
  int hterm = -1;
#pragma unroll
  for (int i = 0; i < NT; ++i)
    if (TT.t[i].linop == DPX_LIN_GRAD_H) hterm = i;
  const int nsteps = (R + 2 + SPB - 1) / SPB;
  const int pair = lbase | ((T - t) & (T - 1));         // lane holding bin M-k for this lane's bin k

  // the last-pass transform twiddles depend on t only: loaded once for the whole band
  TwRegs<M, T, false> twr;
  twr.load(t, twW, 2);

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
    const bool a_live = q <= R + 1;
    int h = r0 - 1 + q;                                 // circular rows: at most one wrap either way
    h = h < 0 ? h + H : (h >= H ? h - H : h);
    const int qz = q - 1;                               // row this sequence updates (x[qz] from LDS, x[qz+1] own)
    const bool z_live = qz >= 0 && qz <= R;
    const bool z_own = qz >= 1 && qz <= R;              // rows of this band (row qz = 0 is the halo above)
    int hz = r0 - 1 + qz;
    hz = hz < 0 ? hz + H : (hz >= H ? hz - H : hz);
    // ---- issue the u loads of phase B now: the inverse transform below covers their latency ----
    float2 ureg[NT][V];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const float2* urow = (const float2*)(TT.t[i].u_in + plane_px + (size_t)hz * (2 * M));
#pragma unroll
      for (int m = 0; m < V; ++m) ureg[i][m] = (z_live && TT.u_live) ? urow[t + m * T] : make_float2(0.f, 0.f);    // (u_live = 0: the duals count as zero, DPX_TERM_U_ZERO)
    }
    // ---------------- phase A: inverse row transform of row q ----------------
    float2 xa[V];
#pragma unroll
    for (int m = 0; m < V; ++m) {
      const float2 got = make_float2(__shfl(X[V - 1 - m].x, pair), __shfl(X[V - 1 - m].y, pair));
      const float2 xm = cconj(t == 0 ? X[(V - m) % V] : got);
      const int k = t + m * T;
      const float2 xk = X[m];
      if (k == 0) {
        xa[m] = make_float2(xk.x + xn, xk.x - xn);
      } else {
        const float2 e = cadd(xk, xm);
        const float2 d = cmulc(csub(xk, xm), twW[k]);
        xa[m] = make_float2(e.x - d.y, e.y + d.x);
      }
    }
    WaveSync()();
    fft_reg_tw<M, T, +1, false>(xa, myfft, t, twr, WaveSync());   // xa[m] = (x[2n], x[2n+1]), n = t + m*T
    if (a_live) {
      float2* xr = xring + (q % RING) * M;
#pragma unroll
      for (int m = 0; m < V; ++m) xr[t + m * T] = xa[m];
      if (x_out && q >= 1 && q <= R) {
        const size_t xo = (plane_px + (size_t)h * (2 * M)) / 2 + t;
#pragma unroll
        for (int m = 0; m < V; ++m) dpx_emit_pair(x_out, TT.emit_bf16, xo + m * T, xa[m]);
      }
    }
    DPX_LDS_BARRIER();
    // ---------------- phase B: z / dual update of row qz ----------------
    float2 acc[V];                                        // K^T (v - u) accumulated over the terms (row-local parts)
#pragma unroll
    for (int m = 0; m < V; ++m) acc[m] = make_float2(0.f, 0.f);
    {
      // every lane runs the arithmetic (the shuffles need converged T-lane groups); only memory writes are predicated
      const float2* xc_row = xring + ((qz + RING) % RING) * M;
      float2 xc[V];
#pragma unroll
      for (int m = 0; m < V; ++m) xc[m] = xc_row[t + m * T];
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const IterTerm tm = TT.t[i];
        const float lam = tm.lam ? tm.lam[bi] * tm.alpha : 0.f;
        float2 w[V];
#pragma unroll
        for (int m = 0; m < V; ++m) {
          const float2 uu = ureg[i][m];
          float2 kx;
          if (tm.linop == DPX_LIN_IDENTITY) {
            kx = xc[m];
          } else if (tm.linop == DPX_LIN_GRAD_H) {
            kx = make_float2(xa[m].x - xc[m].x, xa[m].y - xc[m].y);
          } else {                                      // grad_W: x[w+1] - x[w]; pixel 2n+2 is the neighbour lane's .x
            const float nx_same = __shfl(xc[m].x, lbase | ((t + 1) & (T - 1)));
            const float nx_wrap = __shfl(xc[(m + 1) % V].x, lbase);
            const float xr = (t == T - 1) ? nx_wrap : nx_same;
            kx = make_float2(xc[m].y - xc[m].x, xr - xc[m].y);
          }
          float dx, dy, vx, vy, ux, uy;
          if (TT.vxu) {                                       // (kernel-uniform) t = q + x, d = K x + t, v = prox(d), q' = t - v, w = v - t
            const float tx = uu.x + xc[m].x, ty = uu.y + xc[m].y;
            dx = kx.x + tx, dy = kx.y + ty;
            vx = prox1(tm.prox, dx, lam), vy = prox1(tm.prox, dy, lam);
            ux = tx - vx, uy = ty - vy;
            w[m] = make_float2(-ux, -uy);
          } else {
            dx = fmaf(TT.dual, uu.x, kx.x), dy = fmaf(TT.dual, uu.y, kx.y);
            vx = prox1(tm.prox, dx, lam), vy = prox1(tm.prox, dy, lam);
            ux = dx - vx, uy = dy - vy;
            w[m] = make_float2(fmaf(-TT.dual, ux, vx), fmaf(-TT.dual, uy, vy));
          }
          if (z_own) {
            ((float2*)(tm.u_out + plane_px + (size_t)hz * (2 * M)))[t + m * T] = make_float2(ux, uy);
            if (emit_v) dpx_emit_pair(tm.v_out, TT.emit_bf16, (plane_px + (size_t)hz * (2 * M)) / 2 + t + m * T, make_float2(vx, vy));
          }
        }
        // this term's contribution to K^T (v - u) that needs no other row
        if (tm.linop == DPX_LIN_IDENTITY) {
#pragma unroll
          for (int m = 0; m < V; ++m) acc[m] = cadd(acc[m], w[m]);
        } else if (tm.linop == DPX_LIN_GRAD_W) {          // adjoint: y[w-1] - y[w]; pixel 2n-1 is the left lane's .y
#pragma unroll
          for (int m = 0; m < V; ++m) {
            const float l_same = __shfl(w[m].y, lbase | ((t + T - 1) & (T - 1)));
            const float l_wrap = __shfl(w[(m + V - 1) % V].y, lbase | (T - 1));
            const float wl = (t == 0) ? l_wrap : l_same;
            acc[m] = make_float2(acc[m].x + (wl - w[m].x), acc[m].y + (w[m].x - w[m].y));
          }
        } else {                                          // grad_H: (v - u) goes to the ring, its adjoint is formed in phase C
          if (z_live) {
            float2* wr = wring + (qz % RING) * M;
#pragma unroll
            for (int m = 0; m < V; ++m) wr[t + m * T] = w[m];
          }
        }
      }
    }
    DPX_LDS_BARRIER();
    // ---- issue the spectrum loads of the next step's phase A: the forward transform below covers them ----
    if (s + 1 < nsteps && q + SPB <= R + 1) {
      int hn = r0 - 1 + q + SPB;
      hn = hn >= H ? hn - H : hn;
      const float2* in = sin_main + (unsigned)hn * SPEC_TILE + tile_off;
#pragma unroll
      for (int m = 0; m < V; ++m) X[m] = in[tile_step * m];
      xn = sin_side[hn].x;
    }
    // ---------------- phase C: right-hand-side increment of row qz and its forward row transform ----
    if (rho_next) {
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
      for (int m = 0; m < V; ++m) z[m] = make_float2(rho * acc[m].x, rho * acc[m].y);
      if (TT.rhs_out && z_own) {
        const size_t ro = (plane_px + (size_t)hz * (2 * M)) / 2 + t;
#pragma unroll
        for (int m = 0; m < V; ++m) dpx_emit_pair(TT.rhs_out, TT.emit_bf16, ro + m * T, z[m]);
      }
      WaveSync()();
      fft_reg_tw<M, T, -1, false>(z, myfft, t, twr, WaveSync());
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
    // ring rows read in phase B / C are rewritten by the next phase A / B, each behind a barrier; the transform
    // scratch is wave-local (WaveSync before reuse).
  }
}

static size_t iter_rows_lds(int M, int T) {
  const int SPB = 256 / T, S = M + M / 16;
  return (size_t)(SPB * S + 2 * (SPB + 1) * M) * sizeof(float2);
}

template <int M, int T, int NT>
static void launch_iter_rows_nt(const float2* sin, float2* sout, const IterTerms& TT, const float* rho_next, float* x_out, int emit_v,
                                int C, int H, int R, int P, const float2* twW, hipStream_t s) {
  const size_t sh = iter_rows_lds(M, T);
  static bool attr = false;
  if (!attr && sh > 48 * 1024) {
    hipFuncSetAttribute((const void*)k_iter_rows<M, T, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    attr = true;
  }
  DPX_LAUNCH("k_iter_rows", (k_iter_rows<M, T, NT>), dim3(P * (H / R)), dim3(256), sh, s, sin, sout, TT, rho_next, x_out, emit_v, C,
             H, R, P, twW);
}
template <int M, int T>
static void launch_iter_rows(const float2* sin, float2* sout, const IterTerms& TT, const float* rho_next, float* x_out, int emit_v,
                             int C, int H, int R, int P, const float2* twW, hipStream_t s) {
  switch (TT.n) {
    case 1: launch_iter_rows_nt<M, T, 1>(sin, sout, TT, rho_next, x_out, emit_v, C, H, R, P, twW, s); break;
    case 2: launch_iter_rows_nt<M, T, 2>(sin, sout, TT, rho_next, x_out, emit_v, C, H, R, P, twW, s); break;
    case 3: launch_iter_rows_nt<M, T, 3>(sin, sout, TT, rho_next, x_out, emit_v, C, H, R, P, twW, s); break;
    default: launch_iter_rows_nt<M, T, 4>(sin, sout, TT, rho_next, x_out, emit_v, C, H, R, P, twW, s); break;
  }
}


// ------------------------------------------------------------------------------------------------------------
// k_iter_rows_seq: the same band update as k_iter_rows, organised so that HBM streams continuously.
//
// One T-lane group (a whole wave when T = 64) owns a band of R rows and walks down it row by row:
//   * the neighbouring rows the stencils need (x[h] for grad_H, (v-u)[h-1] for its adjoint) are simply the
//     previous step's registers -- no LDS rings, no workgroup barriers, waves never wait for each other;
//   * the next spectrum row and the next u rows are fetched by LDS-DMA (global_load_lds, no VGPRs) one full step
//     ahead into a wave-private staging area, so every wave always has ~(1 + NT) rows in flight while it
//     computes; the waits are hand-counted vmcnt values (loads, LDS-DMA and stores retire in issue order), never 0
//     in the steady state, so posted stores and the prefetch stay in flight across them;
//   * the inverse untangling reads bin k and bin M-k straight from the staging area (no cross-lane shuffles).
// Per band: R+2 inverse transforms, R+1 z-updates, R forward transforms (halo = one row above, one below).
// The loop contains no compiler-visible global load (twiddles live in LDS / registers, per-image scalars are
// read before the loop): hipcc therefore never inserts a vmcnt wait of its own that would drain the prefetch.
// Cache policy of the row kernel's four streams (dpx_common.h): spectrum in / u in (loads, 1 = nt), u out / spectrum out
// (stores, 2 = nt).  The two spectra are the hand-over between the two kernels of an iteration (written by one, read by the
// next: 2 x 100 MB at 8x3x1024^2, which the 256 MB Infinity Cache can hold); the dual variables come back a whole iteration
// later and the data spectrum is re-read once per iteration -- streaming those past the caches (`nt`) leaves the cache to the
// spectra.  Measured (12-variant matrix, gpurun_out/ab3.log): all plain 4692 it/s, this setting 5147 it/s.
// (DPX_R_LDX / _LDU / _STU / _STX and the constants R_LDX ... R_STX: dpx_iter_dev.h)
// DUAL = false: half-quadratic splitting (TT.dual == 0, DPX_TERM_NO_DUAL) -- the duals are neither fetched nor stored (the general
// kernel streams 16 of its 24 B per element for planes nobody reads: 36 -> 20 B per element and iteration with the column kernel);
// every wait count below is the general one with the dual streams' operations taken out (NU = 0 terms with a dual).
// VXU = true (DUAL only): the update order v, x, u (see IterTerms::vxu) -- two more additions per element and term, its own instantiation
// so that the headline kernel's instruction stream stays as it is.
// (four terms -- their staging areas leave room for one workgroup per CU anyway -- and the 512-wide v, x, u form with three: one wave per SIMD,
//  i.e. up to 512 registers, instead of 2 - 22 spilled ones; tools/spill_check.py)
template <int M, int T, int NT, bool DUAL, bool VXU = false>
__global__ void __launch_bounds__(256, (NT >= 4 || (NT == 3 && VXU && M == 256)) ? 1 : 2) k_iter_rows_seq(const float2* __restrict__ spec_in, float2* __restrict__ spec_out, IterTerms TT,
                                                        const float* __restrict__ rho_next, float* __restrict__ x_out, int emit_v,
                                                        int C, int H, int bands, int P, const float2* __restrict__ twW) {
  constexpr int V = M / T, G = 64 / T, S = LdsSeq<M>::SLOTS, D = V / 2, RM = M / (V * V);
  constexpr int STG = 64 * V;                           // float2 per staged row set of one wave (G rows)
  constexpr int NU = DUAL ? NT : 0;                   // terms whose dual is streamed
  constexpr int PERWAVE = G * S + STG + 32 + NU * STG;
  HIP_DYNAMIC_SHARED(float2, smem_sq)
  float2* twl = smem_sq;                                // untangling twiddles exp(-i pi k / M), k < M
  float2* twb = smem_sq + M;                            // pass-B twiddles W_{V*RM}^j, j < 64
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane / T, t = lane % T, lbase = lane & ~(T - 1);
  float2* wl = twb + 64 + wave * PERWAVE;
  float2* myfft = wl + g * S;
  float2* stX = wl + G * S;
  float* stN = (float*)(stX + STG);
  float2* stU = stX + STG + 32;
  for (int i = tid; i < M; i += 256) twl[i] = twW[i];
  if (tid < 64) twb[tid] = twW[(tid * (M / (V * RM)) * 2) % (2 * M)];   // W_{2M}^{2 j M/(V RM)} = W_{V RM}^j  (j < 64 = V*RM)
  TwRegs<M, T, false> twr;
  twr.load(t, twW, 2);
  twr.twb_ = twb;
  twr.bstride_ = 1;

  // `bands` bands per plane; the first H % bands of them are one row longer
  const int band = (blockIdx.x * 4 + wave) * G + g;
  const int pl = band / bands, bb = band - pl * bands;
  const int rbase = H / bands, rrem = H - rbase * bands;
  const int r0 = bb * rbase + (bb < rrem ? bb : rrem);
  const int R = rbase + (bb < rrem ? 1 : 0);
  const int Rmax = rbase + (rrem ? 1 : 0);            // loop bound of the whole wave (its groups differ by at most one row)
  const int bi = pl / C;
  const float rho = rho_next ? rho_next[bi] : 0.f;
  float lamv[NT];
  int hterm = -1;
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    lamv[i] = TT.t[i].lam ? TT.t[i].lam[bi] * TT.t[i].alpha : 0.f;
    if (TT.t[i].linop == DPX_LIN_GRAD_H) hterm = i;
  }
  __syncthreads();                                      // the only workgroup barrier: the twiddle copies

  // per-lane element offsets (float2 units)
  const unsigned e0 = 2u * t;                           // first of the two elements this lane fetches per DMA piece
  const unsigned xoff = (unsigned)pl * H * M + (e0 / SPEC_TILE) * H * SPEC_TILE + (e0 % SPEC_TILE);
  const unsigned xstep = (unsigned)(2 * T / SPEC_TILE) * H * SPEC_TILE;   // elements e0 + 2T*i
  const unsigned noff = (unsigned)P * H * M + (unsigned)pl * H;
  const unsigned usc = (unsigned)TT.u_live;
  const unsigned uoff = (unsigned)pl * H * M * usc + e0;      // row-major image rows: M float2 per row
  const unsigned tile_off = (unsigned)pl * H * M + (unsigned)((t % SPEC_TILE) + (t / SPEC_TILE) * H * SPEC_TILE);
  const unsigned tile_step = (unsigned)((T / SPEC_TILE) * H * SPEC_TILE);
  const int pair = lbase | ((T - t) & (T - 1));
  auto rowof = [&](int q) { int h = r0 - 1 + q; return h < 0 ? h + H : (h >= H ? h - H : h); };
  auto stage_idx = [&](int e) { return ((e >> 1) / T) * 128 + g * 2 * T + ((e >> 1) % T) * 2 + (e & 1); };
  auto issue_x = [&](int h) {
#pragma unroll
    for (int i = 0; i < D; ++i) dpx_glds16<R_LDX>(spec_in + xoff + (unsigned)h * SPEC_TILE + xstep * i, stX + i * 128);
    dpx_glds4<R_LDX>(spec_in + noff + h, stN);
  };
  auto issue_u = [&](int h) {
#pragma unroll
    for (int n = 0; n < NU; ++n) {
      const float2* urow = (const float2*)TT.t[n].u_in + uoff + (unsigned)h * M * usc;
#pragma unroll
      for (int i = 0; i < D; ++i) dpx_glds16<R_LDU>(urow + 2 * T * i, stU + n * STG + i * 128);
    }
  };
  // lower bounds of the vector-memory operations issued AFTER the awaited DMA (a wait count must never exceed the real
  // number, or it could return early).  The V spectral stores of phase C only exist when a next right-hand side is
  // produced (rho_next != NULL); the emit stores (x_out, v_out) are never counted.
  constexpr int NX_STEADY = (NU * (D + V) + V) > 63 ? 63 : (NU * (D + V) + V);
  constexpr int NU_LAST = (NU * V + V) > 63 ? 63 : (NU * V + V);
  constexpr int NU_STEADY = (NU * V + V + D + 1) > 63 ? 63 : (NU * V + V + D + 1);
  constexpr int NX_STEADY_NS = (NU * (D + V)) > 63 ? 63 : (NU * (D + V));
  constexpr int NU_LAST_NS = (NU * V) > 63 ? 63 : (NU * V);
  constexpr int NU_STEADY_NS = (NU * V + D + 1) > 63 ? 63 : (NU * V + D + 1);
  // LATE_U (three and four terms): the dual rows are read from the staging area term by term inside the update loop instead of all
  // at once in front of it (NT x V register pairs less alive through the loop: no scratch for ADMM with TV + nonneg / + norm1), so the
  // next dual rows are requested BEHIND the loop -- the dual stores and v emits of a step then sit in FRONT of that request and drop
  // out of the count of operations issued after it
  constexpr bool LATE_U = DUAL && NT >= 3;
  constexpr int NU_LAST_L = V, NU_STEADY_L = V + D + 1, NU_LAST_NS_L = 0, NU_STEADY_NS_L = D + 1;
  const bool has_spec = rho_next != nullptr;
  // An emitting pass (the last one of a call; every one under a callback) also stores x (V per row, phase A, behind issue_x) and v
  // (NT V per row, phase B, behind issue_u).  With bands of equal length every steady-state wait has them behind the awaited DMA
  // as well, so they are added to its count -- without them the wave would also wait for its emit stores of the previous row
  // (the pass took 1.8 x a plain one instead of the 1.33 x its bytes ask for).  emode: 0 none / unequal bands, 1 x, 2 x and v.
  constexpr int EX = V, EV = NT * V;
  // xonly (emit_v == 2, no-dual instantiation only): the last pass of a solve() that hands back x alone -- inverse transforms of
  // the band's own rows and their x stores, nothing else (no halo rows, no z / dual update: 8 B per element)
  const bool xonly = !DUAL && emit_v == 2;
  const int qfirst = xonly ? 1 : 0, qlast = xonly ? Rmax : Rmax + 1;
  const int emode = (x_out && rrem == 0) ? ((emit_v && !xonly) ? 2 : 1) : 0;
  auto wait_emit = [&](auto nbase, auto with_x, auto with_v) {
    constexpr int N0 = decltype(nbase)::value, WX = decltype(with_x)::value, WV = decltype(with_v)::value;
    constexpr int N1 = (N0 + WX * EX) > 63 ? 63 : (N0 + WX * EX), N2 = (N0 + WX * EX + WV * EV) > 63 ? 63 : (N0 + WX * EX + WV * EV);
    if (emode == 2) dpx_wait_vm<N2>();
    else if (emode == 1) dpx_wait_vm<N1>();
    else dpx_wait_vm<N0>();
  };

  issue_x(rowof(qfirst));
  issue_u(rowof(0));
  float2 xprev[V], wprev[V];
#pragma unroll
  for (int m = 0; m < V; ++m) xprev[m] = wprev[m] = make_float2(0.f, 0.f);

  for (int q = qfirst; q <= qlast; ++q) {
    // ---------------- phase A: inverse row transform of row q ----------------
    if (q >= 3) {
      if (has_spec) wait_emit(std::integral_constant<int, NX_STEADY>(), std::integral_constant<int, 1>(), std::integral_constant<int, 1>());
      else wait_emit(std::integral_constant<int, NX_STEADY_NS>(), std::integral_constant<int, 1>(), std::integral_constant<int, 1>());
    }
    else if (q == 1) dpx_wait_vm<0>();
    else dpx_wait_vm<NU * D>();
    float2 xa[V];
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
      if (q < qlast) issue_x(rowof(q + 1));
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
    if (x_out && q >= 1 && q <= R) {
      const size_t xo = (size_t)pl * H * M + (size_t)rowof(q) * M + t;
#pragma unroll
      for (int m = 0; m < V; ++m) dpx_emit_pair(x_out, TT.emit_bf16, xo + m * T, xa[m]);
    }
    if (q >= 1 && !xonly) {
      // ---------------- phase B: z / dual update of row qz = q - 1 (x[qz] = xprev, x[qz+1] = xa) ----------------
      const int qz = q - 1;
      const bool own = qz >= 1 && qz <= R;
      const unsigned hz = (unsigned)rowof(qz);
      float2 ureg[NT][V];
      const float dualf = DUAL ? TT.dual : 0.f;
      if constexpr (DUAL) {
        if (q >= 3) {
          using I0 = std::integral_constant<int, 0>;
          using I1 = std::integral_constant<int, 1>;
          if constexpr (LATE_U) {                       // (the v emits of the previous step sit in front of the awaited request)
            if (has_spec) {
              if (q <= Rmax) wait_emit(std::integral_constant<int, NU_STEADY_L>(), I1(), I0());
              else wait_emit(std::integral_constant<int, NU_LAST_L>(), I0(), I0());
            } else {
              if (q <= Rmax) wait_emit(std::integral_constant<int, NU_STEADY_NS_L>(), I1(), I0());
              else wait_emit(std::integral_constant<int, NU_LAST_NS_L>(), I0(), I0());
            }
          } else if (has_spec) {
            if (q <= Rmax) wait_emit(std::integral_constant<int, NU_STEADY>(), I1(), I1());
            else wait_emit(std::integral_constant<int, NU_LAST>(), I0(), I1());      // (row Rmax + 1 is a halo row: no x store in its phase A)
          } else {
            if (q <= Rmax) wait_emit(std::integral_constant<int, NU_STEADY_NS>(), I1(), I1());
            else wait_emit(std::integral_constant<int, NU_LAST_NS>(), I0(), I1());
          }
        } else {
          dpx_wait_vm<D + 1>();
        }
        if constexpr (!LATE_U) {
#pragma unroll
          for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int m = 0; m < V; ++m) ureg[n][m] = stU[n * STG + stage_idx(t + m * T)];
          dpx_wait_lds();
          if (qz < Rmax) issue_u(rowof(qz + 1));
        }
      } else {
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int m = 0; m < V; ++m) ureg[n][m] = make_float2(0.f, 0.f);
      }
      float2 acc[V];
#pragma unroll
      for (int m = 0; m < V; ++m) acc[m] = make_float2(0.f, 0.f);
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const IterTerm tm = TT.t[n];
        const float lam = lamv[n];
        if constexpr (LATE_U) {
#pragma unroll
          for (int m = 0; m < V; ++m) ureg[n][m] = stU[n * STG + stage_idx(t + m * T)];
        }
        // d = K x + u   (the operator / prox codes are wave-uniform: one branch per term, not per element)
        float2 d[V];
        if (tm.linop == DPX_LIN_IDENTITY) {
#pragma unroll
          for (int m = 0; m < V; ++m) d[m] = make_float2(fmaf(dualf, ureg[n][m].x, xprev[m].x), fmaf(dualf, ureg[n][m].y, xprev[m].y));
        } else if (tm.linop == DPX_LIN_GRAD_H) {
#pragma unroll
          for (int m = 0; m < V; ++m) d[m] = make_float2(fmaf(dualf, ureg[n][m].x, xa[m].x - xprev[m].x), fmaf(dualf, ureg[n][m].y, xa[m].y - xprev[m].y));
        } else {                                        // grad_W: x[w+1] - x[w]; pixel 2n+2 is the neighbour lane's .x
#pragma unroll
          for (int m = 0; m < V; ++m) {
            const float nx_same = __shfl(xprev[m].x, lbase | ((t + 1) & (T - 1)));
            const float nx_wrap = __shfl(xprev[(m + 1) % V].x, lbase);
            const float xr = (t == T - 1) ? nx_wrap : nx_same;
            d[m] = make_float2(fmaf(dualf, ureg[n][m].x, xprev[m].y - xprev[m].x), fmaf(dualf, ureg[n][m].y, xr - xprev[m].y));
          }
        }
        float2 tq[VXU ? V : 1];                              // VXU: t = q + x (the dual in front of this v-update)
        if constexpr (VXU) {
#pragma unroll
          for (int m = 0; m < V; ++m) {
            tq[m] = make_float2(ureg[n][m].x + xprev[m].x, ureg[n][m].y + xprev[m].y);
            d[m] = make_float2(d[m].x + xprev[m].x, d[m].y + xprev[m].y);      // (d was K x + q: + x)
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
        if (tm.linop == DPX_LIN_IDENTITY) {
#pragma unroll
          for (int m = 0; m < V; ++m) acc[m] = cadd(acc[m], w[m]);
        } else if (tm.linop == DPX_LIN_GRAD_W) {          // adjoint: y[w-1] - y[w]; pixel 2n-1 is the left lane's .y
#pragma unroll
          for (int m = 0; m < V; ++m) {
            const float l_same = __shfl(w[m].y, lbase | ((t + T - 1) & (T - 1)));
            const float l_wrap = __shfl(w[(m + V - 1) % V].y, lbase | (T - 1));
            const float wlft = (t == 0) ? l_wrap : l_same;
            acc[m] = make_float2(acc[m].x + (wlft - w[m].x), acc[m].y + (w[m].x - w[m].y));
          }
        } else {                                          // grad_H adjoint: y[h-1] - y[h], the row above is last step's w
#pragma unroll
          for (int m = 0; m < V; ++m) {
            acc[m] = make_float2(acc[m].x + (wprev[m].x - w[m].x), acc[m].y + (wprev[m].y - w[m].y));
            wprev[m] = w[m];
          }
        }
      }
      if constexpr (LATE_U) {
        dpx_wait_lds();                                 // (the loop's reads of the staging area have returned)
        if (qz < Rmax) issue_u(rowof(qz + 1));
      }
      // ---------------- phase C: right-hand-side increment of row qz and its forward row transform ----------------
      if (own && rho_next) {
        float2 z[V];
#pragma unroll
        for (int m = 0; m < V; ++m) z[m] = make_float2(rho * acc[m].x, rho * acc[m].y);
        if (TT.rhs_out) {
          const size_t ro = (size_t)pl * H * M + (size_t)hz * M + t;
#pragma unroll
          for (int m = 0; m < V; ++m) dpx_emit_pair(TT.rhs_out, TT.emit_bf16, ro + m * T, z[m]);
        }
        WaveSync()();
        fft_reg_tw<M, T, -1, false>(z, myfft, t, twr, WaveSync());
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
            st_stream<R_STX>(spec_out + noff + hz, make_float2(zk.x - zk.y, 0.f));
          } else {
            const float2 e = cscale(cadd(zk, zm), 0.5f);
            const float2 d = cscale(csub(zk, zm), 0.5f);
            Xo = cadd(e, cmul(make_float2(d.y, -d.x), twl[k]));
          }
          st_stream<R_STX>(out + tile_step * m, Xo);
        }
      }
    }
#pragma unroll
    for (int m = 0; m < V; ++m) xprev[m] = xa[m];
  }
}

static size_t iter_rows_seq_lds(int M, int T, int NU) {
  const int V = M / T, G = 64 / T, S = M + M / 16, STG = 64 * V;
  return (size_t)(M + 64 + 4 * (G * S + STG + 32 + NU * STG)) * sizeof(float2);
}
template <int M, int T, int NT, bool DUAL, bool VXU = false>
static void launch_iter_rows_seq_d(const float2* sin, float2* sout, const IterTerms& TT, const float* rho_next, float* x_out, int emit_v,
                                   int C, int H, int R, int P, const float2* twW, hipStream_t s) {
  const size_t sh = iter_rows_seq_lds(M, T, DUAL ? NT : 0);
  static bool attr = false;
  if (!attr && sh > 48 * 1024) {
    hipFuncSetAttribute((const void*)k_iter_rows_seq<M, T, NT, DUAL, VXU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    attr = true;
  }
  const int groups = P * R, per_block = 4 * (64 / T);   // R = bands per plane here
  DPX_LAUNCH(VXU ? "k_iter_rows_seq_vxu" : (DUAL ? "k_iter_rows_seq" : "k_iter_rows_seq_nodual"), (k_iter_rows_seq<M, T, NT, DUAL, VXU>), dim3(groups / per_block),
             dim3(256), sh, s, sin, sout, TT, rho_next, x_out, emit_v, C, H, R, P, twW);
}
template <int M, int T, int NT>
static void launch_iter_rows_seq_nt(const float2* sin, float2* sout, const IterTerms& TT, const float* rho_next, float* x_out, int emit_v,
                                    int C, int H, int R, int P, const float2* twW, hipStream_t s) {
  const bool keep_dual = false;                                 // (half-quadratic splitting on the general kernel: bit-identical, 36 instead of 20 B per pixel)
  // emit_v == 2 (x only: the last pass of a solve() that returns x alone) runs on the no-dual instantiation whatever the solver
  if (emit_v == 2 && x_out && !rho_next) launch_iter_rows_seq_d<M, T, NT, false>(sin, sout, TT, rho_next, x_out, 2, C, H, R, P, twW, s);
  else if (TT.vxu) launch_iter_rows_seq_d<M, T, NT, true, true>(sin, sout, TT, rho_next, x_out, emit_v ? 1 : 0, C, H, R, P, twW, s);
  else if (TT.dual == 0.f && !keep_dual) launch_iter_rows_seq_d<M, T, NT, false>(sin, sout, TT, rho_next, x_out, emit_v ? 1 : 0, C, H, R, P, twW, s);
  else launch_iter_rows_seq_d<M, T, NT, true>(sin, sout, TT, rho_next, x_out, emit_v ? 1 : 0, C, H, R, P, twW, s);
}
template <int M, int T>
static void launch_iter_rows_seq(const float2* sin, float2* sout, const IterTerms& TT, const float* rho_next, float* x_out, int emit_v,
                                 int C, int H, int R, int P, const float2* twW, hipStream_t s) {
  switch (TT.n) {
    case 1: launch_iter_rows_seq_nt<M, T, 1>(sin, sout, TT, rho_next, x_out, emit_v, C, H, R, P, twW, s); break;
    case 2: launch_iter_rows_seq_nt<M, T, 2>(sin, sout, TT, rho_next, x_out, emit_v, C, H, R, P, twW, s); break;
    case 3: launch_iter_rows_seq_nt<M, T, 3>(sin, sout, TT, rho_next, x_out, emit_v, C, H, R, P, twW, s); break;
    default: launch_iter_rows_seq_nt<M, T, 4>(sin, sout, TT, rho_next, x_out, emit_v, C, H, R, P, twW, s); break;
  }
}

// ------------------------------------------------------------------------------------------------------------
// k_pgd_rows_seq: the row pass of a proximal-gradient iteration (k_pgd_rows, dpx_fft_pow2.hip; reference dprox/algo/pgd.py:26-54) on
// the streaming structure of k_iter_rows_seq: one T-lane group walks down a band of rows; the next row's spectrum, iterate and K^T b
// arrive by LDS-DMA one step ahead (hand-counted vmcnt waits, never 0 in the steady state).  No row couples to its neighbours here
// (the prox acts on x itself), so there is no halo: per row one inverse transform, the forward step and the prox on registers, one
// forward transform.  20 B per pixel: spectrum in, x in, K^T b in, x out, spectrum out.
//   after the awaited spectrum DMA of row q (issued in step q-1): NR D (row DMAs of q) + V (x stores) [+ V spectrum stores]
//   after the awaited row DMAs of row q (issued in step q-1)    : V [+ V] stores of step q-1, + D + 1 (spectrum DMA of q+1) unless q is the last row
template <int M, int T, bool KTB>
__global__ void __launch_bounds__(256, 2) k_pgd_rows_seq(const float2* __restrict__ spec_in, float2* __restrict__ spec_out, float* __restrict__ x,
                                                       const float* __restrict__ ktb, const float* __restrict__ rho, const float* __restrict__ lam,
                                                       float alpha, int prox, int C, int H, int bands, int P, const float2* __restrict__ twW) {
  constexpr int V = M / T, G = 64 / T, S = LdsSeq<M>::SLOTS, D = V / 2, RM = M / (V * V);
  constexpr int STG = 64 * V;
  constexpr int NR = KTB ? 2 : 1;                       // image-row streams: x, K^T b
  constexpr int PERWAVE = G * S + STG + 32 + NR * STG;
  HIP_DYNAMIC_SHARED(float2, smem_pq)
  float2* twl = smem_pq;
  float2* twb = smem_pq + M;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane / T, t = lane % T, lbase = lane & ~(T - 1);
  float2* wl = twb + 64 + wave * PERWAVE;
  float2* myfft = wl + g * S;
  float2* stX = wl + G * S;
  float* stN = (float*)(stX + STG);
  float2* stR = stX + STG + 32;
  for (int i = tid; i < M; i += 256) twl[i] = twW[i];
  if (tid < 64) twb[tid] = twW[(tid * (M / (V * RM)) * 2) % (2 * M)];
  TwRegs<M, T, false> twr;
  twr.load(t, twW, 2);
  twr.twb_ = twb;
  twr.bstride_ = 1;
  const int band = (blockIdx.x * 4 + wave) * G + g;
  const int pl = band / bands, bb = band - pl * bands;
  const int R = H / bands, r0 = bb * R;                 // equal bands (H and bands are powers of two)
  const int bi = pl / C;
  const float rr = rho[bi], th = lam ? lam[bi] * alpha : 0.f;
  __syncthreads();
  const unsigned e0 = 2u * t;
  const unsigned xoff = (unsigned)pl * H * M + (e0 / SPEC_TILE) * H * SPEC_TILE + (e0 % SPEC_TILE);
  const unsigned xstep = (unsigned)(2 * T / SPEC_TILE) * H * SPEC_TILE;
  const unsigned noff = (unsigned)P * H * M + (unsigned)pl * H;
  const unsigned uoff = (unsigned)pl * H * M + e0;
  const unsigned tile_off = (unsigned)pl * H * M + (unsigned)((t % SPEC_TILE) + (t / SPEC_TILE) * H * SPEC_TILE);
  const unsigned tile_step = (unsigned)((T / SPEC_TILE) * H * SPEC_TILE);
  const int pair = lbase | ((T - t) & (T - 1));
  auto stage_idx = [&](int e) { return ((e >> 1) / T) * 128 + g * 2 * T + ((e >> 1) % T) * 2 + (e & 1); };
  auto issue_x = [&](int h) {
#pragma unroll
    for (int i = 0; i < D; ++i) dpx_glds16<R_LDX>(spec_in + xoff + (unsigned)h * SPEC_TILE + xstep * i, stX + i * 128);
    dpx_glds4<R_LDX>(spec_in + noff + h, stN);
  };
  auto issue_r = [&](int h) {
    const float2* xrow = (const float2*)x + uoff + (unsigned)h * M;
#pragma unroll
    for (int i = 0; i < D; ++i) dpx_glds16<0>(xrow + 2 * T * i, stR + i * 128);
    if constexpr (KTB) {
      const float2* krow = (const float2*)ktb + uoff + (unsigned)h * M;
#pragma unroll
      for (int i = 0; i < D; ++i) dpx_glds16<1>(krow + 2 * T * i, stR + STG + i * 128);
    }
  };
  constexpr int NXS = NR * D + 2 * V, NXS_NS = NR * D + V;
  constexpr int NRS = 2 * V + D + 1, NRS_LAST = 2 * V, NRS_NS = V + D + 1, NRS_LAST_NS = V;
  static_assert(NXS <= 63 && NRS <= 63, "vmcnt is a 6-bit counter");
  const bool has_spec = spec_out != nullptr;
  issue_x(r0);
  issue_r(r0);
  for (int q = 0; q < R; ++q) {
    const int h = r0 + q;
    // ---------------- phase A: inverse row transform of row h ----------------
    if (q == 0) dpx_wait_vm<NR * D>();
    else if (has_spec) dpx_wait_vm<NXS>();
    else dpx_wait_vm<NXS_NS>();
    float2 xa[V];
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
      if (q + 1 < R) issue_x(h + 1);
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
    fft_reg_tw<M, T, +1, false>(xa, myfft, t, twr, WaveSync());   // xa[m] = two adjacent pixels of K^T K x
    // ---------------- phase B: forward step + prox on the row, store x ----------------
    if (q == 0) {
      if (R > 1) dpx_wait_vm<D + 1>();
      else dpx_wait_vm<0>();
    } else if (has_spec) {
      if (q + 1 < R) dpx_wait_vm<NRS>();
      else dpx_wait_vm<NRS_LAST>();
    } else {
      if (q + 1 < R) dpx_wait_vm<NRS_NS>();
      else dpx_wait_vm<NRS_LAST_NS>();
    }
    float2 xo[V], kb[V];
#pragma unroll
    for (int m = 0; m < V; ++m) {
      xo[m] = stR[stage_idx(t + m * T)];
      kb[m] = KTB ? stR[STG + stage_idx(t + m * T)] : make_float2(0.f, 0.f);
    }
    dpx_wait_lds();
    if (q + 1 < R) issue_r(h + 1);
    float2 z[V];
#pragma unroll
    for (int m = 0; m < V; ++m) {
      const float2 y = make_float2(xo[m].x - rr * (xa[m].x - kb[m].x), xo[m].y - rr * (xa[m].y - kb[m].y));
      z[m] = make_float2(prox1(prox, y.x, th), prox1(prox, y.y, th));
    }
    {
      float2* xw = (float2*)x + (unsigned)pl * H * M + (unsigned)h * M + t;
#pragma unroll
      for (int m = 0; m < V; ++m) xw[m * T] = z[m];
    }
    // ---------------- phase C: forward row transform of the new row ----------------
    if (has_spec) {
      WaveSync()();
      fft_reg_tw<M, T, -1, false>(z, myfft, t, twr, WaveSync());
      float2* out = spec_out + tile_off + (unsigned)h * SPEC_TILE;
#pragma unroll
      for (int m = 0; m < V; ++m) {
        const float2 got = make_float2(__shfl(z[V - 1 - m].x, pair), __shfl(z[V - 1 - m].y, pair));
        const float2 zm = cconj(t == 0 ? z[(V - m) % V] : got);
        const int k = t + m * T;
        const float2 zk = z[m];
        float2 Xo;
        if (k == 0) {
          Xo = make_float2(zk.x + zk.y, 0.f);
          spec_out[noff + h] = make_float2(zk.x - zk.y, 0.f);
        } else {
          const float2 e = cscale(cadd(zk, zm), 0.5f);
          const float2 d = cscale(csub(zk, zm), 0.5f);
          Xo = cadd(e, cmul(make_float2(d.y, -d.x), twl[k]));
        }
        out[tile_step * m] = Xo;
      }
    }
  }
}

// run-time overrides shared with the ADMM row kernels (dpx_admm_iter_config): rows_mode 2 keeps the plain kernels, bands_per_plane > 0
// fixes the band count
extern int g_rows_mode_pgd, g_rows_band_pgd, g_chain_share;
template <int M, int T>
static bool launch_pgd_rows_seq(const float2* sin, float2* sout, float* x, const float* ktb, const float* rho, const float* lam, float alpha, int prox,
                                int C, int H, int P, const float2* twW, hipStream_t s) {
  constexpr int V = M / T, G = 64 / T, S = M + M / 16, STG = 64 * V;
  // bands per plane: a power of two, ~3 rounds of the resident T-lane groups (2 workgroups of 4 waves per CU) -- there is no halo
  // to amortise here, and shorter bands even out the tail (8x3x1024^2: 64 / 128 / 256 bands 0.158 / 0.150 / 0.147 ms per iteration)
  int nb = (256 * 2 * 4 * G) / (P * g_chain_share);
  int p2 = 1;
  while (p2 < nb) p2 <<= 1;
  nb = 2 * p2;
  const int band_env = 0;
  if (band_env) nb = band_env;
  if (g_rows_band_pgd > 0) nb = g_rows_band_pgd;
  if (nb > H) nb = H;
  const int per_block = 4 * G;
  if (nb < 1 || H % nb || (P * nb) % per_block) return false;
  const size_t sh1 = (size_t)(M + 64 + 4 * (G * S + STG + 32 + 1 * STG)) * sizeof(float2), sh2 = (size_t)(M + 64 + 4 * (G * S + STG + 32 + 2 * STG)) * sizeof(float2);
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute((const void*)k_pgd_rows_seq<M, T, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh2);
    hipFuncSetAttribute((const void*)k_pgd_rows_seq<M, T, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh1);
    attr = true;
  }
  const dim3 grid(P * nb / per_block);
  if (ktb)
    DPX_LAUNCH("k_pgd_rows_seq", (k_pgd_rows_seq<M, T, true>), grid, dim3(256), sh2, s, sin, sout, x, ktb, rho, lam, alpha, prox, C, H, nb, P, twW);
  else
    DPX_LAUNCH("k_pgd_rows_seq", (k_pgd_rows_seq<M, T, false>), grid, dim3(256), sh1, s, sin, sout, x, ktb, rho, lam, alpha, prox, C, H, nb, P, twW);
  return true;
}
// false: the plane / batch does not fit the streaming kernel (the caller keeps k_pgd_rows)
bool pgd_rows_seq_pow2(const float2* sin, float2* sout, float* x, const float* ktb, const float* rho, const float* lam, float alpha, int prox, int P,
                       int C, int H, int W, const void* table, hipStream_t s) {
  const bool plain = false;                               // (the plain row kernel serves the planes the streaming one has no partition for)
  if (plain || g_rows_mode_pgd == 2) return false;
  switch (W) {
    case 256: return launch_pgd_rows_seq<128, 16>(sin, sout, x, ktb, rho, lam, alpha, prox, C, H, P, tw_rows(table), s);
    case 512: return launch_pgd_rows_seq<256, 32>(sin, sout, x, ktb, rho, lam, alpha, prox, C, H, P, tw_rows(table), s);
    case 1024: return launch_pgd_rows_seq<512, 64>(sin, sout, x, ktb, rho, lam, alpha, prox, C, H, P, tw_rows(table), s);
    case 768: return launch_pgd_rows_seq<384, 64>(sin, sout, x, ktb, rho, lam, alpha, prox, C, H, P, tw_rows(table), s);
    default: return false;
  }
}

// ------------------------------------------------------------------------------------------------------------
// k_seed_rows_seq: the seed pass of a solve that starts from ADMM.initialize's state (k_seed_rows<FRESH>, dpx_fft_pow2.hip: the row
// transform of rho_0 sum_i K_i^T K_i x0, formed from x0's rows alone) on the streaming structure: one T-lane group walks down a band,
// the next row of x0 arrives by LDS-DMA while the current one is transformed; the band's two halo rows (grad_H couples a row to
// both neighbours) are read once per band instead of twice per row.  8 B per pixel + the halo: image in, spectrum out.  The stencil
// arithmetic is k_seed_rows' (same differences, same order).
// Two rows are in flight per group (two staging areas): with one, a launch has ~8 MB outstanding, short of what 8 TB/s needs.
//   after the awaited DMA of row q (issued in step q-2): the V (+1) spectrum stores of steps q-2 and q-1, where those steps produced
//   a row, and the D pieces of row q+1 (issued in step q-1)
#ifndef DPX_SEED_STX
#define DPX_SEED_STX DPX_R_STX
#endif
constexpr int SEED_STX = DPX_SEED_STX;        // cache policy of the seed pass's spectrum stores (0 plain, 1 write-through, 2 nt)
struct SeedOps {
  int linop[DPX_MAX_TERMS];
  int n;
};
template <int M, int T>
__global__ void __launch_bounds__(256, 2) k_seed_rows_seq(SeedOps SO, const float* __restrict__ rho, const float* __restrict__ x0,
                                                        float2* __restrict__ spec_out, int C, int H, int bands, int P, const float2* __restrict__ twW) {
  constexpr int V = M / T, G = 64 / T, S = LdsSeq<M>::SLOTS, D = V / 2, RM = M / (V * V);
  constexpr int STG = 64 * V;
  constexpr int PERWAVE = G * S + 2 * STG;
  HIP_DYNAMIC_SHARED(float2, smem_sd)
  float2* twl = smem_sd;
  float2* twb = smem_sd + M;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane / T, t = lane % T, lbase = lane & ~(T - 1);
  float2* wl = twb + 64 + wave * PERWAVE;
  float2* myfft = wl + g * S;
  float2* stR = wl + G * S;
  for (int i = tid; i < M; i += 256) twl[i] = twW[i];
  if (tid < 64) twb[tid] = twW[(tid * (M / (V * RM)) * 2) % (2 * M)];
  TwRegs<M, T, false> twr;
  twr.load(t, twW, 2);
  twr.twb_ = twb;
  twr.bstride_ = 1;
  const int band = (blockIdx.x * 4 + wave) * G + g;
  const int pl = band / bands, bb = band - pl * bands;
  const int R = H / bands, r0 = bb * R;                 // equal bands
  const float rr = rho[pl / C];
  bool has_h = false;
  for (int i = 0; i < SO.n; ++i) has_h |= SO.linop[i] == DPX_LIN_GRAD_H;
  __syncthreads();
  const unsigned e0 = 2u * t;
  const unsigned noff = (unsigned)P * H * M + (unsigned)pl * H;
  const unsigned uoff = (unsigned)pl * H * M + e0;
  const unsigned tile_off = (unsigned)pl * H * M + (unsigned)((t % SPEC_TILE) + (t / SPEC_TILE) * H * SPEC_TILE);
  const unsigned tile_step = (unsigned)((T / SPEC_TILE) * H * SPEC_TILE);
  const int pair = lbase | ((T - t) & (T - 1));
  auto rowof = [&](int q) { int h = r0 - 1 + q; return h < 0 ? h + H : (h >= H ? h - H : h); };
  auto stage_idx = [&](int e) { return ((e >> 1) / T) * 128 + g * 2 * T + ((e >> 1) % T) * 2 + (e & 1); };
  auto issue_r = [&](int h, int slot) {
    const float2* xrow = (const float2*)x0 + uoff + (unsigned)h * M;
#pragma unroll
    for (int i = 0; i < D; ++i) dpx_glds16<0>(xrow + 2 * T * i, stR + slot * STG + i * 128);
  };
  // rows r0 - 1 + q; with a grad_H term step q reads row q and produces row q - 1 (q = 0 and R + 1: the halo rows), without one it
  // produces the row it reads
  const int qfirst = has_h ? 0 : 1, qlast = has_h ? R + 1 : R;
  issue_r(rowof(qfirst), qfirst & 1);
  issue_r(rowof(qfirst + 1), (qfirst + 1) & 1);         // (bands have >= 4 rows)
  float2 xc[V], xp[V];
#pragma unroll
  for (int m = 0; m < V; ++m) xc[m] = xp[m] = make_float2(0.f, 0.f);
  for (int q = qfirst; q <= qlast; ++q) {
    // lower bounds of what was issued behind row q's DMA (steps produce a row from q = qfirst + 2 on at the latest)
    const int qq = q - qfirst;
    if (q == qlast) {
      if (qq >= 4) dpx_wait_vm<2 * V>();
      else dpx_wait_vm<0>();
    } else if (qq >= 4) dpx_wait_vm<2 * V + D>();
    else if (qq == 3) dpx_wait_vm<V + D>();
    else dpx_wait_vm<D>();
    float2 xn[V];
#pragma unroll
    for (int m = 0; m < V; ++m) xn[m] = stR[(q & 1) * STG + stage_idx(t + m * T)];
    dpx_wait_lds();
    if (q + 2 <= qlast) issue_r(rowof(q + 2), q & 1);
    if (has_h ? q >= 2 : true) {
      float2 ce[V];
#pragma unroll
      for (int m = 0; m < V; ++m) ce[m] = has_h ? xc[m] : xn[m];
      float2 acc[V];
#pragma unroll
      for (int m = 0; m < V; ++m) acc[m] = make_float2(0.f, 0.f);
      for (int i = 0; i < SO.n; ++i) {
        const int op = SO.linop[i];
        if (op == DPX_LIN_IDENTITY) {
#pragma unroll
          for (int m = 0; m < V; ++m) acc[m] = cadd(acc[m], ce[m]);
        } else if (op == DPX_LIN_GRAD_W) {
          float2 y[V];
#pragma unroll
          for (int m = 0; m < V; ++m) {                   // x[w+1] - x[w]; pixel 2n+2 is the neighbour lane's .x
            const float nx_same = __shfl(ce[m].x, lbase | ((t + 1) & (T - 1)));
            const float nx_wrap = __shfl(ce[(m + 1) % V].x, lbase);
            const float xr_ = (t == T - 1) ? nx_wrap : nx_same;
            y[m] = make_float2(ce[m].y - ce[m].x, xr_ - ce[m].y);
          }
#pragma unroll
          for (int m = 0; m < V; ++m) {                   // adjoint: y[w-1] - y[w]; pixel 2n-1 is the left lane's .y
            const float l_same = __shfl(y[m].y, lbase | ((t + T - 1) & (T - 1)));
            const float l_wrap = __shfl(y[(m + V - 1) % V].y, lbase | (T - 1));
            const float left = (t == 0) ? l_wrap : l_same;
            acc[m] = make_float2(acc[m].x + (left - y[m].x), acc[m].y + (y[m].x - y[m].y));
          }
        } else {                                          // grad_H: (x[h] - x[h-1]) - (x[h+1] - x[h])
#pragma unroll
          for (int m = 0; m < V; ++m) {
            const float2 y = csub(xn[m], ce[m]);
            const float2 yu = csub(ce[m], xp[m]);
            acc[m] = cadd(acc[m], csub(yu, y));
          }
        }
      }
      const unsigned hz = (unsigned)rowof(has_h ? q - 1 : q);
      float2 z[V];
#pragma unroll
      for (int m = 0; m < V; ++m) z[m] = cscale(acc[m], rr);
      WaveSync()();
      fft_reg_tw<M, T, -1, false>(z, myfft, t, twr, WaveSync());
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
          st_stream<SEED_STX>(spec_out + noff + hz, make_float2(zk.x - zk.y, 0.f));
        } else {
          const float2 e = cscale(cadd(zk, zm), 0.5f);
          const float2 d = cscale(csub(zk, zm), 0.5f);
          Xo = cadd(e, cmul(make_float2(d.y, -d.x), twl[k]));
        }
        st_stream<SEED_STX>(out + tile_step * m, Xo);
      }
    }
#pragma unroll
    for (int m = 0; m < V; ++m) {
      xp[m] = xc[m];
      xc[m] = xn[m];
    }
  }
}

template <int M, int T>
static bool launch_seed_rows_seq(const SeedOps& SO, const float* rho, const float* x0, float2* spec, int C, int H, int P, const float2* twW, hipStream_t s) {
  constexpr int V = M / T, G = 64 / T, S = M + M / 16, STG = 64 * V;
  // bands per plane: the row kernel's rule (every T-lane group of the launch resident, rounded up to a power of two)
  int nb = (256 * 2 * 4 * G) / (P * g_chain_share), p2 = 1;
  while (p2 < nb) p2 <<= 1;
  nb = p2;
  const int band_env = 0;
  if (band_env) nb = band_env;
  if (nb > H / 4) nb = H / 4;
  const int per_block = 4 * G;
  if (nb < 1 || H % nb || (P * nb) % per_block) return false;
  const size_t sh = (size_t)(M + 64 + 4 * (G * S + 2 * STG)) * sizeof(float2);
  static bool attr = false;
  if (!attr && sh > 48 * 1024) {
    hipFuncSetAttribute((const void*)k_seed_rows_seq<M, T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    attr = true;
  }
  DPX_LAUNCH("k_seed_rows_seq", (k_seed_rows_seq<M, T>), dim3(P * nb / per_block), dim3(256), sh, s, SO, rho, x0, spec, C, H, nb, P, twW);
  return true;
}
// false: the plane / batch does not fit the streaming kernel (the caller keeps k_seed_rows<FRESH>)
bool seed_rows_seq_pow2(const int* linops, int n, const float* rho, const float* x0, float2* spec, int P, int C, int H, int W, const void* table,
                        hipStream_t s) {
  const bool plain = false;
  if (plain || g_rows_mode_pgd == 2) return false;                 // (dpx_admm_iter_config: 2 = the plain kernels)
  SeedOps SO{};
  SO.n = n;
  for (int i = 0; i < n; ++i) SO.linop[i] = linops[i];
  switch (W) {
    case 256: return launch_seed_rows_seq<128, 16>(SO, rho, x0, spec, C, H, P, tw_rows(table), s);
    case 512: return launch_seed_rows_seq<256, 32>(SO, rho, x0, spec, C, H, P, tw_rows(table), s);
    case 1024: return launch_seed_rows_seq<512, 64>(SO, rho, x0, spec, C, H, P, tw_rows(table), s);
    case 768: return launch_seed_rows_seq<384, 64>(SO, rho, x0, spec, C, H, P, tw_rows(table), s);
    default: return false;
  }
}

size_t pow2_spec_elems(int P, int H, int W);
int cols_solve_pow2(const float2* spec_in, float2* spec_out, const SpecArgs& A, int P, int C, int H, int W, const void* table,
                    hipStream_t stream);
int rows_r2c_pow2(const float* x, float2* spec, int P, int H, int W, const void* table, hipStream_t stream);
int seed_rows_pow2(const dpx_term* terms, int nterms, const float* rho, const float* x0, float2* spec, int B, int C, int H, int W, const void* table,
                   hipStream_t stream);

}  // namespace dpx

using namespace dpx;
namespace dpx {
int iter_rows_impl(const void* spec_in, void* spec_out, const dpx_term* terms, int nterms, const float* rho_next, float* x_out, int emit_v,
                   float* rhs_out, int emit_bf16, int B, int C, int H, int W, const void* table, dpx_stream_t stream);
}

static int terms_ok(const dpx_term* terms, int nterms) {
  if (nterms < 1 || nterms > DPX_MAX_TERMS || !terms) return 0;
  int nh = 0;
  for (int i = 0; i < nterms; ++i) {
    if (terms[i].linop < DPX_LIN_IDENTITY || terms[i].linop > DPX_LIN_GRAD_W) return 0;
    if (terms[i].prox < DPX_PROX_NORM1 || terms[i].prox > DPX_PROX_SUMSQ) return 0;
    nh += terms[i].linop == DPX_LIN_GRAD_H;
  }
  return nh <= 1;
}

// run-time overrides of the row-kernel choice (tests / tuning): rows_mode 0 = automatic, 1 = streaming kernel, 2 = lock-step
// ring-buffer kernel; bands_per_plane 0 = automatic.  The environment variables DPX_ITER_ROWS / DPX_ITER_BAND set the defaults.
static int g_rows_mode = -1, g_rows_band = -1;
namespace dpx { int g_rows_mode_pgd = 0, g_rows_band_pgd = 0, g_chain_share = 1; int iter_seq_bands(int P, int H, int W, int forced); }
// Sub-batches of one solve running as independent chains on separate streams (dprox/algo/fused.py) share the GPU: the row kernels size
// their bands for `chains` x the planes of one call.  A tuning hint only -- results do not depend on the band partition.
// Bands per plane of the streaming row kernel (k_iter_rows_seq) for a launch of P planes of H x W: as many as keep every T-lane group
// of the launch resident at once (2 workgroups of 4 waves per CU, times the chains that share the GPU), rounded UP to a power of two
// (H is one: bands of equal length keep the waves of a workgroup in step, and ~1.5 rounds of resident groups beat one round of
// unequal bands -- 8x3x1024^2: 128 bands of 8 rows 106 us, 85 bands of 12-13 rows 109 us, 96 / 102 / 136 bands 127-133 us), at least
// `iter_band_min_rows` (4) rows long, and such that the groups fill whole workgroups.  A partition always exists: if no count below
// the rule's fills whole workgroups (P odd and large), the smallest count that does is taken (per_block / gcd(P, per_block) <= 16).
int dpx::iter_seq_bands(int P, int H, int W, int forced) {
  const int T = W == 768 ? 64 : W / 16, G = 64 / T, per_block = 4 * G;
  int nb = (256 * 2 * 4 * G) / (P * dpx::g_chain_share);
  {
    int p2 = 1;
    while (p2 < nb) p2 <<= 1;
    nb = p2;
  }
  if (forced) nb = forced;
  const int min_rows_knob = tune(TUNE_ITER_BAND_MIN_ROWS);
  const int min_rows = min_rows_knob > 0 ? min_rows_knob : 4;
  if (nb > H / min_rows) nb = H / min_rows;
  if (nb < 1) nb = 1;
  while (nb > 1 && (P * nb) % per_block) --nb;
  if ((P * nb) % per_block) {
    int a = P, b = per_block;
    while (b) { const int r = a % b; a = b; b = r; }
    const int last_resort = per_block / a;
    // (bands shorter than the rule's minimum -- one or two rows at H = 32 -- are not what the streaming kernel's tests walk: planes other than
    //  768-wide ones then stay on the lock-step kernel, which the caller falls back to when the count returned here fills no whole workgroups)
    if (W == 768 || H / last_resort >= min_rows) nb = last_resort;
  }
  return nb;
}

extern "C" int dpx_admm_iter_bands(int planes, int H, int W) {
  if (planes < 1 || H < 16 || !(W == 256 || W == 512 || W == 768 || W == 1024)) return 0;
  return dpx::iter_seq_bands(planes, H, W, g_rows_band > 0 ? g_rows_band : tune(TUNE_ITER_BAND));
}
extern "C" int dpx_admm_iter_share(int chains) {
  DPX_REQUIRE(chains >= 1 && chains <= 16, "dpx_admm_iter_share: chains must be in [1, 16]");
  dpx::g_chain_share = chains;
  return DPX_OK;
}
extern "C" int dpx_admm_iter_config(int rows_mode, int bands_per_plane) {
  DPX_REQUIRE(rows_mode >= 0 && rows_mode <= 3 && bands_per_plane >= 0, "dpx_admm_iter_config: bad arguments");
  g_rows_mode = rows_mode;
  g_rows_band = bands_per_plane;
  dpx::g_rows_mode_pgd = rows_mode;                      // (dpx_pgd_run's row pass follows the same switch: 2 = the plain kernel)
  dpx::g_rows_band_pgd = bands_per_plane;
  return DPX_OK;
}

extern "C" int dpx_admm_iter_supported(int H, int W, const dpx_term* terms, int nterms) {
  // (2048-wide planes: 16 values per thread spilled 121 - 279 registers in the lock-step row kernel -- 33 ps per pixel and iteration
  //  against 14 on the staged kernels, which the callers use; the instantiation is gone since round 5)
  // 768-wide rows: M = 384 = 6 * 8 * 8 on one wave per row (fft384_wave, 6 values per lane) in the streaming kernels; 1536: staged kernels only
  const bool wok = W == 256 || W == 512 || W == 768 || W == 1024;
  return pow2_path_available(H, W) && wok && H % 16 == 0 && terms_ok(terms, nterms);
}

extern "C" int dpx_rfft_rows(const float* x, void* spec, int B, int C, int H, int W, const void* table, dpx_stream_t stream) {
  DPX_REQUIRE(x && spec && table, "dpx_rfft_rows: null pointer");
  DPX_REQUIRE(pow2_path_available(H, W), "dpx_rfft_rows: only power-of-two planes use the two-kernel iteration");
  return rows_r2c_pow2(x, (float2*)spec, B * C, H, W, table, (hipStream_t)stream);
}

// spec = row transform of rho_b sum_i K_i^T (v_i - u_i): dpx_admm_rhs (ktb = NULL) + dpx_rfft_rows in one pass (the seed of dpx_admm_run)
extern "C" int dpx_admm_seed_rows(void* spec, const float* rho, const dpx_term* terms, int nterms, int B, int C, int H, int W, const void* table,
                                  dpx_stream_t stream) {
  DPX_REQUIRE(spec && rho && terms && table, "dpx_admm_seed_rows: null pointer");
  DPX_REQUIRE(dpx_admm_iter_supported(H, W, terms, nterms), "dpx_admm_seed_rows: unsupported problem (plane %dx%d, %d terms)", H, W, nterms);
  for (int i = 0; i < nterms; ++i) DPX_REQUIRE(terms[i].v && terms[i].u, "dpx_admm_seed_rows: term %d lacks v / u", i);
  return seed_rows_pow2(terms, nterms, rho, nullptr, (float2*)spec, B, C, H, W, table, (hipStream_t)stream);
}

// the same seed for a state that comes straight from ADMM.initialize (admm.py:61-67: v_i = K_i x0, u_i = 0): the pass recomputes
// v_i - u_i from x0 (bit-identical: the same fp32 differences, minus an exact zero) and reads one image instead of 2 nterms
extern "C" int dpx_admm_seed_rows_fresh(void* spec, const float* rho, const float* x0, const dpx_term* terms, int nterms, int B, int C, int H, int W,
                                        const void* table, dpx_stream_t stream) {
  DPX_REQUIRE(spec && rho && terms && table && x0, "dpx_admm_seed_rows_fresh: null pointer");
  DPX_REQUIRE(dpx_admm_iter_supported(H, W, terms, nterms), "dpx_admm_seed_rows_fresh: unsupported problem (plane %dx%d, %d terms)", H, W, nterms);
  return seed_rows_pow2(terms, nterms, rho, x0, (float2*)spec, B, C, H, W, table, (hipStream_t)stream);
}

extern "C" int dpx_admm_iter_cols(const void* spec_in, void* spec_out, const void* spec_add, const void* dd, const float* rho,
                                  float eps, int B, int C, int H, int W, const void* table, dpx_stream_t stream) {
  DPX_REQUIRE(spec_in && spec_out && dd && rho && table, "dpx_admm_iter_cols: null pointer");
  DPX_REQUIRE(pow2_path_available(H, W), "dpx_admm_iter_cols: unsupported plane size %dx%d", H, W);
  SpecArgs a{};
  a.dd = (const float2*)dd;
  a.add = (const float2*)spec_add;
  a.rho = rho;
  a.eps = eps;
  a.eps_num = eps;
  a.scale = 1.0f / ((float)H * (float)W);
  return cols_solve_pow2((const float2*)spec_in, (float2*)spec_out, a, B * C, C, H, W, table, (hipStream_t)stream);
}

extern "C" int dpx_admm_iter_rows(const void* spec_in, void* spec_out, const dpx_term* terms, int nterms, const float* rho_next,
                                  float* x_out, int emit_v, int B, int C, int H, int W, const void* table, dpx_stream_t stream) {
  return dpx::iter_rows_impl(spec_in, spec_out, terms, nterms, rho_next, x_out, emit_v, nullptr, 0, B, C, H, W, table, stream);
}

// + rhs_out (nullable): the right-hand-side increment handed to the next x-update, also written as an image;
// + emit_bf16: x_out, terms[i].v and rhs_out are bf16 planes (written with round-to-nearest-even)
int dpx::iter_rows_impl(const void* spec_in, void* spec_out, const dpx_term* terms, int nterms, const float* rho_next, float* x_out, int emit_v,
                        float* rhs_out, int emit_bf16, int B, int C, int H, int W, const void* table, dpx_stream_t stream) {
  DPX_REQUIRE(spec_in && table && (spec_out || !rho_next), "dpx_admm_iter_rows: null pointer");
  DPX_REQUIRE(dpx_admm_iter_supported(H, W, terms, nterms), "dpx_admm_iter_rows: unsupported problem (plane %dx%d, %d terms)", H, W, nterms);
  IterTerms TT;
  TT.n = nterms;
  TT.rhs_out = rho_next ? rhs_out : nullptr;
  TT.emit_bf16 = emit_bf16;
  TT.dual = (nterms > 0 && (terms[0].reserved & DPX_TERM_NO_DUAL)) ? 0.f : 1.f;
  TT.u_live = (nterms > 0 && (terms[0].reserved & DPX_TERM_U_ZERO)) ? 0 : 1;
  TT.vxu = (nterms > 0 && (terms[0].reserved & DPX_TERM_VXU)) ? 1 : 0;
  DPX_REQUIRE(!(TT.vxu && TT.dual == 0.f), "dpx_admm_iter_rows: DPX_TERM_VXU and DPX_TERM_NO_DUAL exclude each other");
  for (int i = 0; i < nterms; ++i) {
    DPX_REQUIRE(!(terms[i].reserved & DPX_TERM_NO_DUAL) == (TT.dual != 0.f), "dpx_admm_iter_rows: DPX_TERM_NO_DUAL must be set on every term or on none");
    DPX_REQUIRE(!(terms[i].reserved & DPX_TERM_U_ZERO) == (TT.u_live != 0), "dpx_admm_iter_rows: DPX_TERM_U_ZERO must be set on every term or on none");
    DPX_REQUIRE(!(terms[i].reserved & DPX_TERM_VXU) == (TT.vxu == 0), "dpx_admm_iter_rows: DPX_TERM_VXU must be set on every term or on none");
    DPX_REQUIRE(terms[i].u && (terms[i].u_out) && (!emit_v || terms[i].v), "dpx_admm_iter_rows: term %d lacks u / u_out / v", i);
    DPX_REQUIRE(terms[i].u != terms[i].u_out, "dpx_admm_iter_rows: u must be double-buffered (u_out != u)");
    TT.t[i] = IterTerm{terms[i].linop, terms[i].prox, terms[i].alpha, terms[i].lam, terms[i].u, terms[i].u_out, terms[i].v};
  }
  const int P = B * C;
  const float2* tw = tw_rows(table);
  hipStream_t s = (hipStream_t)stream;
  const float2* sin = (const float2*)spec_in;
  float2* sout = (float2*)spec_out;
  // Streaming kernel (one T-lane group per band): bands as long as possible while still >= ~2 waves per SIMD-pair
  // of the chip; DPX_ITER_ROWS=lockstep keeps the ring-buffer kernel (A/B timing), DPX_ITER_BAND overrides the number of bands per plane.
  const int mode_knob = tune(TUNE_ITER_ROWS), band_env0 = tune(TUNE_ITER_BAND);
  const char* mode_env = mode_knob == 1 ? "seq" : (mode_knob == 2 ? "lockstep" : (mode_knob == 3 ? "par" : nullptr));
  const char* mode = g_rows_mode > 0 ? (g_rows_mode == 1 ? "seq" : (g_rows_mode == 2 ? "lockstep" : "par")) : (g_rows_mode == 0 ? nullptr : mode_env);
  // launches of a few planes: the rows of a band side by side (dpx_iter_par.hip; bit-identical to the streaming kernel)
  if (!mode || !strcmp(mode, "par")) {
    if (launch_iter_rows_par(sin, sout, TT, rho_next, x_out, emit_v, C, H, W, P, tw, s, mode != nullptr)) return launch_status("dpx_admm_iter_rows");
  }
  const int band_env = g_rows_band >= 0 ? g_rows_band : band_env0;
  // (small launches -- a few 256-wide planes -- are latency-bound: the ring-buffer kernel's row-parallel bands finish ~10 %
  //  sooner there than the streaming kernel's sequential ones; measured crossover between 256- and 512-wide planes)
  const bool tiny = W <= 256 && (long)P * H <= 4096 && !(mode && !strcmp(mode, "seq"));
  if ((!(mode && !strcmp(mode, "lockstep")) && W <= 1024 && !tiny) || W == 768) {      // (768-wide rows exist on the streaming kernel only)
    const int T = W == 768 ? 64 : W / 16, G = 64 / T, per_block = 4 * G;
    const int nb = dpx::iter_seq_bands(P, H, W, band_env);
    if (nb >= 1 && (P * nb) % per_block == 0) {
      switch (W) {
        case 256: launch_iter_rows_seq<128, 16>(sin, sout, TT, rho_next, x_out, emit_v, C, H, nb, P, tw, s); break;
        case 512: launch_iter_rows_seq<256, 32>(sin, sout, TT, rho_next, x_out, emit_v, C, H, nb, P, tw, s); break;
        case 768: launch_iter_rows_seq<384, 64>(sin, sout, TT, rho_next, x_out, emit_v, C, H, nb, P, tw, s); break;
        default: launch_iter_rows_seq<512, 64>(sin, sout, TT, rho_next, x_out, emit_v, C, H, nb, P, tw, s); break;
      }
      return launch_status("dpx_admm_iter_rows");
    }
  }
  DPX_REQUIRE(W != 768, "dpx_admm_iter_rows: no band partition of %d planes of %d rows for the 768-wide streaming kernel", P, H);
  const int r_env = tune(TUNE_ITER_R);   // tuning: rows per band of the ring-buffer kernel
  // (256-wide planes: 16 rows are in flight per workgroup, so a band of 8 rows + its 2 halo rows is ONE step of the kernel instead
  //  of two -- these launches are latency-bound: config 1 0.78 -> 0.62 ms per 20-iteration solve)
  const int R = r_env ? r_env : (W <= 256 ? 8 : 16);
  switch (W) {
    case 256: launch_iter_rows<128, 16>(sin, sout, TT, rho_next, x_out, emit_v ? 1 : 0, C, H, R, P, tw, s); break;
    case 512: launch_iter_rows<256, 32>(sin, sout, TT, rho_next, x_out, emit_v ? 1 : 0, C, H, R, P, tw, s); break;
    default: launch_iter_rows<512, 64>(sin, sout, TT, rho_next, x_out, emit_v ? 1 : 0, C, H, R, P, tw, s); break;
  }
  return launch_status("dpx_admm_iter_rows");
}

// Runs `n_iters` consecutive iterations (it0 .. it0 + n_iters - 1 of `total_iters`) without returning to the host
// language: 2 kernel launches per iteration.  rho_tab is [total_iters][B], lam_tabs[i] is [total_iters][B] for term i.
// spec_a holds the row-transformed right-hand side on entry (dpx_rfft_rows) and on exit (unless the solve ended);
// u_i alternate between terms[i].u and terms[i].u_out: the return value (>= 0) is 0 if the current u_i are in
// terms[i].u after the call, 1 if they are in terms[i].u_out.  x / v_i are written only by the final iteration of
// the call when emit_last is set.
extern "C" int dpx_admm_run(void* spec_a, void* spec_b, const void* spec_add, const void* dd, const dpx_term* terms, int nterms,
                            const float* rho_tab, const float* const* lam_tabs, float eps, int it0, int n_iters, int total_iters,
                            float* x_out, int emit_last, int B, int C, int H, int W, const void* table, dpx_stream_t stream) {
  DPX_REQUIRE(spec_a && spec_b && dd && terms && rho_tab && lam_tabs && table, "dpx_admm_run: null pointer");
  DPX_REQUIRE(n_iters >= 0 && it0 >= 0 && it0 + n_iters <= total_iters, "dpx_admm_run: bad iteration range");
  DPX_REQUIRE(!emit_last || x_out, "dpx_admm_run: emit_last needs x_out");
  dpx_term cur[DPX_MAX_TERMS];
  DPX_REQUIRE(nterms >= 1 && nterms <= DPX_MAX_TERMS, "dpx_admm_run: nterms");
  for (int i = 0; i < nterms; ++i) cur[i] = terms[i];
  int parity = 0;
  const bool cols_inplace = false;                           // (column pass in place: a round-2 experiment, no gain)
  for (int k = 0; k < n_iters; ++k) {
    const int it = it0 + k;
    const bool last_of_solve = (it == total_iters - 1);
    const bool emit = emit_last && (k == n_iters - 1);
    if (cols_inplace) {
      // column pass in place (every workgroup reads its whole tile before it writes it); the row pass then needs the other buffer
      void* t = spec_a; spec_a = spec_b; spec_b = t;
      int rc0 = dpx_admm_iter_cols(spec_b, spec_b, spec_add, dd, rho_tab + (size_t)it * B, eps, B, C, H, W, table, stream);
      if (rc0) return rc0;
    }
    int rc = cols_inplace ? 0 : dpx_admm_iter_cols(spec_a, spec_b, spec_add, dd, rho_tab + (size_t)it * B, eps, B, C, H, W, table, stream);
    if (rc) return rc;
    for (int i = 0; i < nterms; ++i) {
      cur[i].lam = lam_tabs[i] ? lam_tabs[i] + (size_t)it * B : nullptr;
      if (k > 0) cur[i].reserved &= ~DPX_TERM_U_ZERO;      // (only the duals the call starts from can be the fresh zeros)
      cur[i].u = parity ? terms[i].u_out : terms[i].u;
      cur[i].u_out = parity ? terms[i].u : terms[i].u_out;
    }
    rc = dpx_admm_iter_rows(spec_b, last_of_solve ? nullptr : spec_a, cur, nterms,
                            last_of_solve ? nullptr : rho_tab + (size_t)(it + 1) * B, emit ? x_out : nullptr,
                            emit ? (emit_last == 2 && last_of_solve ? 2 : 1) : 0, B, C,
                            H, W, table, stream);
    if (rc) return rc;
    parity ^= 1;
  }
  return parity;
}

// Stream fork / join for the sub-batch chains, with events kept per (host thread, device): `to[i]` waits for everything issued on `from`
// so far / `into` waits for everything issued on `from[i]` so far.  Streams equal to the other side are skipped.
namespace {
struct ChainEvents {
  hipEvent_t ev[DPX_MAX_CHAINS + 1];
  bool ok = false;
};
ChainEvents* chain_events(const char* who) {
  constexpr int MAXDEV = 16;
  static thread_local ChainEvents evs[MAXDEV];
  int devid = 0;
  if (hipGetDevice(&devid) != hipSuccess || devid < 0 || devid >= MAXDEV) {
    dpx::set_error("%s: hipGetDevice failed / device id %d out of range", who, devid);
    return nullptr;
  }
  ChainEvents& E = evs[devid];
  if (!E.ok) {
    for (int i = 0; i <= DPX_MAX_CHAINS; ++i)
      if (hipEventCreateWithFlags(&E.ev[i], hipEventDisableTiming) != hipSuccess) {
        for (int k = 0; k < i; ++k) hipEventDestroy(E.ev[k]);
        dpx::set_error("%s: hipEventCreate failed", who);
        return nullptr;
      }
    E.ok = true;
  }
  return &E;
}
}  // namespace
extern "C" int dpx_stream_fork(dpx_stream_t from, const dpx_stream_t* to, int n) {
  DPX_REQUIRE(to && n >= 0 && n <= DPX_MAX_CHAINS, "dpx_stream_fork: bad arguments");
  ChainEvents* E = chain_events("dpx_stream_fork");
  if (!E) return DPX_ERR_LAUNCH;
  bool recorded = false;
  for (int i = 0; i < n; ++i) {
    if (to[i] == from) continue;
    if (!recorded && hipEventRecord(E->ev[DPX_MAX_CHAINS], (hipStream_t)from) != hipSuccess) {
      set_error("dpx_stream_fork: hipEventRecord failed");
      return DPX_ERR_LAUNCH;
    }
    recorded = true;
    if (hipStreamWaitEvent((hipStream_t)to[i], E->ev[DPX_MAX_CHAINS], 0) != hipSuccess) {
      set_error("dpx_stream_fork: hipStreamWaitEvent failed");
      return DPX_ERR_LAUNCH;
    }
  }
  return DPX_OK;
}
extern "C" int dpx_stream_join(dpx_stream_t into, const dpx_stream_t* from, int n) {
  DPX_REQUIRE(from && n >= 0 && n <= DPX_MAX_CHAINS, "dpx_stream_join: bad arguments");
  ChainEvents* E = chain_events("dpx_stream_join");
  if (!E) return DPX_ERR_LAUNCH;
  for (int i = 0; i < n; ++i) {
    if (from[i] == into) continue;
    if (hipEventRecord(E->ev[i], (hipStream_t)from[i]) != hipSuccess || hipStreamWaitEvent((hipStream_t)into, E->ev[i], 0) != hipSuccess) {
      set_error("dpx_stream_join: event record / wait failed");
      return DPX_ERR_LAUNCH;
    }
  }
  return DPX_OK;
}

// The same loop for `nchains` sub-batches of one solve, each on its own stream (the images of a batch never exchange data: one
// chain's column pass can run beside another chain's row pass).  Launches are issued chain by chain within an iteration, so that every
// chain has work queued from the first microseconds on; the chains then run freely.  (DPX_CHAIN_LOCKSTEP=1, measured and kept as a
// switch: the column passes ordered by events -- chain c's of iteration k behind chain c-1's, chain 0's of iteration k+1 behind the last
// chain's of iteration k.  It removes the drift between the chains -- left to the queues' arbitration one chain of two finishes ~12 %
// earlier -- but also forbids column pass beside column pass and row pass beside row pass: 0.174 -> 0.202 ms per iteration at 8x3x1024^2.)
// Returns the dual-buffer parity (the same for every chain), < 0 on error.
extern "C" int dpx_admm_run_chains(const dpx_chain* chains, int nchains, const void* dd, int nterms, float eps, int it0, int n_iters,
                                   int total_iters, int emit_last, int C, int H, int W, const void* table) {
  DPX_REQUIRE(chains && nchains >= 1 && nchains <= DPX_MAX_CHAINS && dd && table, "dpx_admm_run_chains: bad arguments");
  DPX_REQUIRE(n_iters >= 0 && it0 >= 0 && it0 + n_iters <= total_iters, "dpx_admm_run_chains: bad iteration range");
  DPX_REQUIRE(nterms >= 1 && nterms <= DPX_MAX_TERMS, "dpx_admm_run_chains: nterms");
  for (int c = 0; c < nchains; ++c) {
    const dpx_chain& ch = chains[c];
    DPX_REQUIRE(ch.spec_a && ch.spec_b && ch.terms && ch.rho_tab && ch.lam_tabs && ch.B >= 1, "dpx_admm_run_chains: chain %d: null pointer", c);
    DPX_REQUIRE(!emit_last || ch.x_out, "dpx_admm_run_chains: emit_last needs x_out (chain %d)", c);
  }
  ChainEvents* Ep = chain_events("dpx_admm_run_chains");
  if (!Ep) return DPX_ERR_LAUNCH;
  ChainEvents& E = *Ep;
  const bool lockstep = false;                          // (chains in lock step: measured slower, round 3; the ordered form below stays for reference runs)
  const bool ordered = lockstep && nchains > 1;
  dpx_term cur[DPX_MAX_CHAINS][DPX_MAX_TERMS];
  for (int c = 0; c < nchains; ++c)
    for (int i = 0; i < nterms; ++i) cur[c][i] = chains[c].terms[i];
  // the seed passes of all chains first (it0 == 0 only): issued back to back from here, nothing of the host language between them and the loop
  for (int c = 0; c < nchains; ++c) {
    const dpx_chain& ch = chains[c];
    if (!ch.seed) continue;
    DPX_REQUIRE(it0 == 0 && n_iters > 0, "dpx_admm_run_chains: a chain can only be seeded at the start of the solve (chain %d)", c);
    DPX_REQUIRE(ch.seed == 1 || (ch.seed == 2 && ch.seed_x0), "dpx_admm_run_chains: chain %d: seed mode %d", c, ch.seed);
    for (int i = 0; i < nterms; ++i) cur[c][i].lam = ch.lam_tabs[i];
    const int rc = ch.seed == 2 ? dpx_admm_seed_rows_fresh(ch.spec_a, ch.rho_tab, ch.seed_x0, cur[c], nterms, ch.B, C, H, W, table, ch.stream)
                                : dpx_admm_seed_rows(ch.spec_a, ch.rho_tab, cur[c], nterms, ch.B, C, H, W, table, ch.stream);
    if (rc) return rc;
  }
  int parity = 0;
  for (int k = 0; k < n_iters; ++k) {
    const int it = it0 + k;
    const bool last_of_solve = (it == total_iters - 1);
    const bool emit = emit_last && (k == n_iters - 1);
    for (int c = 0; c < nchains; ++c) {
      const dpx_chain& ch = chains[c];
      hipStream_t s = (hipStream_t)ch.stream;
      if (ordered && (k > 0 || c > 0)) {
        if (hipStreamWaitEvent(s, E.ev[(c + nchains - 1) % nchains], 0) != hipSuccess) {
          set_error("dpx_admm_run_chains: hipStreamWaitEvent failed");
          return DPX_ERR_LAUNCH;
        }
      }
      int rc = dpx_admm_iter_cols(ch.spec_a, ch.spec_b, ch.spec_add, dd, ch.rho_tab + (size_t)it * ch.B, eps, ch.B, C, H, W, table, ch.stream);
      if (rc) return rc;
      if (ordered && hipEventRecord(E.ev[c], s) != hipSuccess) {
        set_error("dpx_admm_run_chains: hipEventRecord failed");
        return DPX_ERR_LAUNCH;
      }
      for (int i = 0; i < nterms; ++i) {
        cur[c][i].lam = ch.lam_tabs[i] ? ch.lam_tabs[i] + (size_t)it * ch.B : nullptr;
        if (k > 0) cur[c][i].reserved &= ~DPX_TERM_U_ZERO;
        cur[c][i].u = parity ? ch.terms[i].u_out : ch.terms[i].u;
        cur[c][i].u_out = parity ? ch.terms[i].u : ch.terms[i].u_out;
      }
      rc = dpx_admm_iter_rows(ch.spec_b, last_of_solve ? nullptr : ch.spec_a, cur[c], nterms,
                              last_of_solve ? nullptr : ch.rho_tab + (size_t)(it + 1) * ch.B, emit ? ch.x_out : nullptr,
                              emit ? (emit_last == 2 && last_of_solve ? 2 : 1) : 0, ch.B, C, H, W, table, ch.stream);
      if (rc) return rc;
    }
    parity ^= 1;
  }
  return parity;
}
