// Unrolled ADMM with closed-form proxes, forward and backward, as two C-side loops (config 5: specialize(method='unroll'),
// reference dprox/algo/specialization/unroll.py:14-58 differentiating dprox/algo/admm.py:49-59 with PyTorch autograd).
//
// This file sequences stage kernels without returning to the host language between stages (issued from Python, a 10-iteration
// training step's launches cost ~4 ms of issue time for ~2.3 ms of run time).  Forward: on power-of-two planes the two-kernel
// iteration of dpx_admm_run (dpx_iter.hip) whose row kernel emits the history -- x, v_i, the next right-hand side; fp32 or bf16 --,
// otherwise dpx_admm_rhs / dpx_fourier_solve / dpx_admm_zupdate.  Backward: the stage kernels of dpx_autodiff.hip with the sums
// between the stages folded in and one finishing launch per iteration for its three reductions.
//
// History buffer (caller-owned): iteration `it` occupies (2 + 2 n) planes of px = B*C*H*W floats:
//     [rhs][x][v_0 .. v_{n-1}][u_0 .. u_{n-1}]
// the backward pass reads rhs, x and v_i of every iteration (u_i only as the forward's running state).
//
// bf16 mode (BASELINE config 5 is quoted in bf16): the forward iteration, every transform and every reduction stay fp32 -- the
// x-update divides by |H|^2 + rho sum|G|^2, which amplifies round-off by up to 1e5, bf16 state would lose
// the iterate -- but what is KEPT for the backward pass is stored in bf16: per iteration rhs, x and v_i (u_i is not needed),
// (2 + n) x 2 bytes per pixel instead of (2 + 2n) x 4.  The soft-threshold / clipping masks the backward reads from v_i survive
// the rounding exactly (v = 0 stays 0); d loss / d rho_t (inner products with x and rhs) carry the bf16 rounding, ~1e-3 relative.
#include "dpx_common.h"

using namespace dpx;

namespace dpx {
int ad_partial_blocks(int C, int H, int W);
int zupdate_bwd_partials(float* gx, const dpx_bwd_term* terms, int nterms, float* part, int hist_bf16, int B, int C, int H, int W, hipStream_t s);
int solve_rhs_bwd_partials(const float* g, const float* x, const float* rhs, const float* rho, const int* linops, int nterms, float* const* gv,
                           float* const* gu, const float* const* gu_add, float* part_a, float* part_b, int hist_bf16, int B, int C, int H, int W,
                           hipStream_t s);
int finish_iter(const float* part_lam, const float* part_a, const float* part_b, float* glam, float* grho, const float* rho, int nterms, int B,
                int C, int H, int W, hipStream_t s);
int iter_rows_impl(const void* spec_in, void* spec_out, const dpx_term* terms, int nterms, const float* rho_next, float* x_out, int emit_v,
                   float* rhs_out, int emit_bf16, int B, int C, int H, int W, const void* table, dpx_stream_t stream);   // dpx_iter.hip
int finish_iter_n(const float* part_lam, const float* part_a, const float* part_b, float* glam, float* grho, const float* rho, int nterms, int B,
                  int nblk, hipStream_t s);
int finish_all(const float* part, long stride, float* glam, float* grho, const float* rho_tab, int nterms, int B, int nblk, int T, int nst,
               hipStream_t s);
int bwd_rows_slots(int B, int C, int H, int W, int max_slots);   // dpx_bwd_rows.hip
int bwd_rows_fused(const void* spec_in, void* spec_out, const float* x, const float* rhs, const float* rho, const dpx_bwd_term* terms, int nterms,
                   const float* const* a_in, float* const* a_out, float* g_out, int g_acc, float* part_a, float* part_b, float* part_lam, int hist_bf16,
                   int B, int C, int H, int W, const void* table, hipStream_t s);
int cols_solve_pow2(const float2* spec_in, float2* spec_out, const SpecArgs& A, int P, int C, int H, int W, const void* table, hipStream_t stream);
int rows_r2c_pow2(const float* x, float2* spec, int P, int H, int W, const void* table, hipStream_t stream);
int rows_c2r_pow2(const float2* spec, float* y, int P, int H, int W, const void* table, hipStream_t stream);   // dpx_fft_pow2.hip
bool rhs_z_bwd_fused(const float* g, const float* x, const float* rhs, const float* rho, const dpx_bwd_term* terms, int nterms, const float* const* a_in,
                     float* const* a_out, float* gx, float* part_a, float* part_b, float* part_lam, int hist_bf16, int B, int C, int H, int W,
                     hipStream_t s, unsigned* counter, float* glam, float* grho);   // dpx_autodiff.hip
int rhs_bwd_impl(const float* g, const float* rhs, const float* rho, const int* linops, int nterms, float* const* gv, float* const* gu,
                 const float* const* gu_add, float* grho, const float* grho_add, int hist_bf16, int B, int C, int H, int W, void* ws,
                 hipStream_t s);   // dpx_autodiff.hip
// fp32 planes -> consecutive bf16 planes of a history slot, round-to-nearest-even
struct PlanePack {
  float* p[2 + DPX_MAX_TERMS];
  int n;
};
__global__ void k_hist_pack_bf16(PlanePack P, unsigned short* __restrict__ slot, long px) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < px * P.n; i += (long)gridDim.x * blockDim.x) {
    const int k = (int)(i / px);
    const unsigned u = __float_as_uint(P.p[k][i - k * px]);
    slot[i] = (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
  }
}
}  // namespace dpx

namespace {
struct Hist {
  float* base;
  size_t px;
  int n;
  float* rhs(int it) const { return base + (size_t)it * (2 + 2 * n) * px; }
  float* x(int it) const { return rhs(it) + px; }
  float* v(int it, int i) const { return rhs(it) + (size_t)(2 + i) * px; }
  float* u(int it, int i) const { return rhs(it) + (size_t)(2 + n + i) * px; }
};
}  // namespace

#define DPX_TRY(call)            \
  do {                           \
    const int dpx_rc_ = (call);  \
    if (dpx_rc_ != DPX_OK) return dpx_rc_; \
  } while (0)

extern "C" size_t dpx_admm_unrolled_hist_bytes(int nterms, int T, int B, int C, int H, int W) {
  return (size_t)T * (2 + 2 * nterms) * B * C * H * W * sizeof(float);
}

extern "C" int dpx_admm_unrolled_forward(float* hist, const float* const* v0, const float* const* u0, const int* linops, const int* proxes,
                                         const float* alphas, int nterms, const float* rho_tab, const float* const* lam_tabs, int T,
                                         const void* spec_add, const void* dd, float eps, int B, int C, int H, int W, const void* table,
                                         void* spectrum_ws, const float* fresh_x0, dpx_stream_t stream) {
  DPX_REQUIRE(hist && v0 && u0 && linops && proxes && alphas && rho_tab && lam_tabs && dd && table && spectrum_ws,
              "dpx_admm_unrolled_forward: null pointer");
  DPX_REQUIRE(nterms >= 1 && nterms <= DPX_MAX_TERMS && T >= 1 && B > 0 && C > 0 && H > 0 && W > 0, "dpx_admm_unrolled_forward: bad sizes");
  const Hist h{hist, (size_t)B * C * H * W, nterms};
  {
    // power-of-two planes with stencil terms: the two-kernel iteration of dpx_admm_run, its row kernel emitting the history (x, v_i,
    // the next right-hand side; u_i are its outputs anyway) -- 2 launches per iteration instead of 5
    dpx_term probe[DPX_MAX_TERMS];
    for (int i = 0; i < nterms; ++i) probe[i] = dpx_term{linops[i], proxes[i], alphas[i], 0, lam_tabs[i], h.v(0, i), (float*)u0[i], h.u(0, i)};
    if (dpx_admm_iter_supported(H, W, probe, nterms) && dpx_spectrum_bytes(B * C, H, W) > 0) {
      char* spec_a = (char*)spectrum_ws;
      char* spec_b = spec_a + dpx_spectrum_bytes(B * C, H, W) / 2;
      dpx_term rt[DPX_MAX_TERMS];
      for (int i = 0; i < nterms; ++i) rt[i] = dpx_term{linops[i], proxes[i], 1.0f, 0, nullptr, (float*)v0[i], (float*)u0[i], nullptr};
      // fresh_x0: the state is ADMM.initialize(x0) untouched (v_i = K_i x0, u_i = 0) and need not exist -- the first right-hand side is formed
      // from x0 and the first iteration does not stream the (zero) duals (DPX_TERM_U_ZERO: row 0 of plane 0 of every image of u0[i] is zero)
      if (fresh_x0) DPX_TRY(dpx_admm_rhs_fresh(h.rhs(0), fresh_x0, rho_tab, linops, nterms, B, C, H, W, stream));
      else DPX_TRY(dpx_admm_rhs(h.rhs(0), nullptr, rho_tab, rt, nterms, B, C, H, W, stream));
      DPX_TRY(dpx_rfft_rows(h.rhs(0), spec_a, B, C, H, W, table, stream));
      for (int it = 0; it < T; ++it) {
        const float* rho = rho_tab + (size_t)it * B;
        DPX_TRY(dpx_admm_iter_cols(spec_a, spec_b, spec_add, dd, rho, eps, B, C, H, W, table, stream));
        dpx_term zt[DPX_MAX_TERMS];
        for (int i = 0; i < nterms; ++i)
          zt[i] = dpx_term{linops[i], proxes[i], alphas[i], (it == 0 && fresh_x0) ? DPX_TERM_U_ZERO : 0, lam_tabs[i] + (size_t)it * B, h.v(it, i),
                           it ? h.u(it - 1, i) : (float*)u0[i], h.u(it, i)};
        const bool last = it == T - 1;
        DPX_TRY(iter_rows_impl(spec_b, last ? nullptr : spec_a, zt, nterms, last ? nullptr : rho_tab + (size_t)(it + 1) * B, h.x(it), 1,
                               last ? nullptr : h.rhs(it + 1), 0, B, C, H, W, table, stream));
      }
      return DPX_OK;
    }
  }
  DPX_REQUIRE(!fresh_x0, "dpx_admm_unrolled_forward: fresh_x0 needs the two-kernel iteration (plane %d x %d is not on it): pass the initial state", H, W);
  for (int it = 0; it < T; ++it) {
    dpx_term rt[DPX_MAX_TERMS], zt[DPX_MAX_TERMS];
    for (int i = 0; i < nterms; ++i) {
      float* pv = it ? h.v(it - 1, i) : (float*)v0[i];
      float* pu = it ? h.u(it - 1, i) : (float*)u0[i];
      rt[i] = dpx_term{linops[i], proxes[i], 1.0f, 0, nullptr, pv, pu, nullptr};
      zt[i] = dpx_term{linops[i], proxes[i], alphas[i], 0, lam_tabs[i] + (size_t)it * B, h.v(it, i), pu, h.u(it, i)};
    }
    const float* rho = rho_tab + (size_t)it * B;
    DPX_TRY(dpx_admm_rhs(h.rhs(it), nullptr, rho, rt, nterms, B, C, H, W, stream));
    DPX_TRY(dpx_fourier_solve(h.rhs(it), h.x(it), spec_add, dd, rho, eps, B, C, H, W, table, spectrum_ws, stream));
    DPX_TRY(dpx_admm_zupdate(h.x(it), zt, nterms, B, C, H, W, stream));
  }
  return DPX_OK;
}

extern "C" size_t dpx_admm_unrolled_hist_bytes_bf16(int nterms, int T, int B, int C, int H, int W) {
  return (size_t)T * (2 + nterms) * B * C * H * W * sizeof(unsigned short);
}
// fp32 working planes of the bf16-history forward: rhs, x, and two generations of v_i / u_i
extern "C" size_t dpx_admm_unrolled_work_bytes_bf16(int nterms, int B, int C, int H, int W) {
  return (size_t)(2 + 4 * nterms) * B * C * H * W * sizeof(float);
}

// Same iteration as dpx_admm_unrolled_forward in fp32 working planes; after every iteration rhs, x, v_i are rounded into the
// bf16 history.  The final state is written to x_out, v_out[i], u_out[i] (fp32).
extern "C" int dpx_admm_unrolled_forward_bf16(void* hist_bf16, void* work, float* x_out, float* const* v_out, float* const* u_out,
                                              const float* const* v0, const float* const* u0, const int* linops, const int* proxes,
                                              const float* alphas, int nterms, const float* rho_tab, const float* const* lam_tabs, int T,
                                              const void* spec_add, const void* dd, float eps, int B, int C, int H, int W,
                                              const void* table, void* spectrum_ws, const float* fresh_x0, dpx_stream_t stream) {
  DPX_REQUIRE(hist_bf16 && work && x_out && v_out && u_out && v0 && u0 && linops && proxes && alphas && rho_tab && lam_tabs && dd && table &&
                  spectrum_ws, "dpx_admm_unrolled_forward_bf16: null pointer");
  DPX_REQUIRE(nterms >= 1 && nterms <= DPX_MAX_TERMS && T >= 1 && B > 0 && C > 0 && H > 0 && W > 0, "dpx_admm_unrolled_forward_bf16: bad sizes");
  const int n = nterms;
  const size_t px = (size_t)B * C * H * W;
  float* w = (float*)work;
  float* rhs_w = w;
  float* x_w = w + px;
  auto vw = [&](int gen, int i) { return w + (size_t)(2 + gen * 2 * n + i) * px; };
  auto uw = [&](int gen, int i) { return w + (size_t)(2 + gen * 2 * n + n + i) * px; };
  unsigned short* hist = (unsigned short*)hist_bf16;
  {
    // power-of-two planes: the two-kernel iteration with the row kernel emitting x, v_i and the next right-hand side (see
    // dpx_admm_unrolled_forward): 3 launches per iteration instead of 6
    dpx_term probe[DPX_MAX_TERMS];
    for (int i = 0; i < n; ++i) probe[i] = dpx_term{linops[i], proxes[i], alphas[i], 0, lam_tabs[i], vw(0, i), (float*)u0[i], uw(0, i)};
    if (dpx_admm_iter_supported(H, W, probe, n) && dpx_spectrum_bytes(B * C, H, W) > 0) {
      // ... and it writes them as bf16 straight into the history slot [rhs][x][v_i] (nothing but u_i and the spectra is read back by
      // the next iteration); only rhs(0) and the last iteration's fp32 outputs go through the packing kernel
      auto slot = [&](int it) { return hist + (size_t)it * (2 + n) * px; };
      char* spec_a = (char*)spectrum_ws;
      char* spec_b = spec_a + dpx_spectrum_bytes(B * C, H, W) / 2;
      dpx_term rt[DPX_MAX_TERMS];
      for (int i = 0; i < n; ++i) rt[i] = dpx_term{linops[i], proxes[i], 1.0f, 0, nullptr, (float*)v0[i], (float*)u0[i], nullptr};
      if (fresh_x0) DPX_TRY(dpx_admm_rhs_fresh(rhs_w, fresh_x0, rho_tab, linops, n, B, C, H, W, stream));      // (see dpx_admm_unrolled_forward)
      else DPX_TRY(dpx_admm_rhs(rhs_w, nullptr, rho_tab, rt, n, B, C, H, W, stream));
      DPX_TRY(dpx_rfft_rows(rhs_w, spec_a, B, C, H, W, table, stream));
      PlanePack P;
      P.n = 1;
      P.p[0] = rhs_w;
      DPX_LAUNCH("k_hist_pack_bf16", k_hist_pack_bf16, dim3(grid_for((long)px, 256, 8192)), dim3(256), 0, (hipStream_t)stream, P, slot(0), (long)px);
      for (int it = 0; it < T; ++it) {
        const bool last = it == T - 1;
        dpx_term zt[DPX_MAX_TERMS];
        for (int i = 0; i < n; ++i) {
          float* pu = it ? uw((it - 1) & 1, i) : (float*)u0[i];
          float* nv = last ? v_out[i] : (float*)(slot(it) + (size_t)(2 + i) * px);
          float* nu = last ? u_out[i] : uw(it & 1, i);
          zt[i] = dpx_term{linops[i], proxes[i], alphas[i], (it == 0 && fresh_x0) ? DPX_TERM_U_ZERO : 0, lam_tabs[i] + (size_t)it * B, nv, pu, nu};
        }
        DPX_TRY(dpx_admm_iter_cols(spec_a, spec_b, spec_add, dd, rho_tab + (size_t)it * B, eps, B, C, H, W, table, stream));
        DPX_TRY(iter_rows_impl(spec_b, last ? nullptr : spec_a, zt, n, last ? nullptr : rho_tab + (size_t)(it + 1) * B,
                               last ? x_out : (float*)(slot(it) + px), 1, last ? nullptr : (float*)slot(it + 1), last ? 0 : 1, B, C, H, W, table,
                               stream));
      }
      P.n = 1 + n;
      P.p[0] = x_out;
      for (int i = 0; i < n; ++i) P.p[1 + i] = v_out[i];
      DPX_LAUNCH("k_hist_pack_bf16", k_hist_pack_bf16, dim3(grid_for((long)(px * (1 + n)), 256, 8192)), dim3(256), 0, (hipStream_t)stream, P,
                 slot(T - 1) + px, (long)px);
      return launch_status("dpx_admm_unrolled_forward_bf16");
    }
  }
  for (int it = 0; it < T; ++it) {
    const bool last = it == T - 1;
    dpx_term rt[DPX_MAX_TERMS], zt[DPX_MAX_TERMS];
    float* xo = last ? x_out : x_w;
    PlanePack P;
    P.n = 2 + n;
    P.p[0] = rhs_w;
    P.p[1] = xo;
    for (int i = 0; i < n; ++i) {
      float* pv = it ? vw((it - 1) & 1, i) : (float*)v0[i];
      float* pu = it ? uw((it - 1) & 1, i) : (float*)u0[i];
      float* nv = last ? v_out[i] : vw(it & 1, i);
      float* nu = last ? u_out[i] : uw(it & 1, i);
      rt[i] = dpx_term{linops[i], proxes[i], 1.0f, 0, nullptr, pv, pu, nullptr};
      zt[i] = dpx_term{linops[i], proxes[i], alphas[i], 0, lam_tabs[i] + (size_t)it * B, nv, pu, nu};
      P.p[2 + i] = nv;
    }
    const float* rho = rho_tab + (size_t)it * B;
    DPX_TRY(dpx_admm_rhs(rhs_w, nullptr, rho, rt, n, B, C, H, W, stream));
    DPX_TRY(dpx_fourier_solve(rhs_w, xo, spec_add, dd, rho, eps, B, C, H, W, table, spectrum_ws, stream));
    DPX_TRY(dpx_admm_zupdate(xo, zt, n, B, C, H, W, stream));
    DPX_LAUNCH("k_hist_pack_bf16", k_hist_pack_bf16, dim3(grid_for((long)(px * (2 + n)), 256, 8192)), dim3(256), 0, (hipStream_t)stream, P,
               hist + (size_t)it * (2 + n) * px, (long)px);
  }
  return launch_status("dpx_admm_unrolled_forward_bf16");
}

// workspace: (4 + 6 n) planes + 2 B floats, followed by the partial sums of one iteration's three reductions ((n + 2) B rows)
extern "C" size_t dpx_admm_unrolled_bwd_ws_bytes(int nterms, int B, int C, int H, int W) {
  return ((size_t)(4 + 6 * nterms) * B * C * H * W + 2 * (size_t)B + 64) * sizeof(float) +
         (size_t)(DPX_MAX_TERMS + 2) * B * ad_partial_blocks(C, H, W) * sizeof(float);
}

// gx / gv_in[i] / gu_in[i]: gradients w.r.t. the final x, v_i, u_i (any may be NULL = 0).
// Out: gv0[i], gu0[i] (w.r.t. the initial split / dual variables), grho [T][B], glam [T][n][B], goff[k] (w.r.t. the k-th
// Omega offset: K g_rhs summed over the iterations; off_otf[k] = that term's OTF table or NULL for the identity; goff[k]
// NULL = not wanted).
static int unrolled_backward_impl(const float* hist, const unsigned short* hist16, const float* gx, const float* const* gv_in, const float* const* gu_in,
                                          float* const* gv0, float* const* gu0, float* grho, float* glam, float* const* goff,
                                          const void* const* off_otf, int n_off, const int* linops, const int* proxes, const float* alphas,
                                          int nterms, const float* rho_tab, const float* const* lam_tabs, int T, const void* dd, float eps,
                                          int B, int C, int H, int W, const void* table, void* spectrum_ws, void* ws, dpx_stream_t stream) {
  DPX_REQUIRE((hist || hist16) && gv0 && gu0 && grho && glam && linops && proxes && alphas && rho_tab && lam_tabs && dd && table &&
                  spectrum_ws && ws, "dpx_admm_unrolled_backward: null pointer");
  DPX_REQUIRE(nterms >= 1 && nterms <= DPX_MAX_TERMS && T >= 1 && B > 0 && n_off >= 0 && (n_off == 0 || (goff && off_otf)),
              "dpx_admm_unrolled_backward: bad sizes");
  const int n = nterms;
  const size_t px = (size_t)B * C * H * W;
  // fp32 history: pointers into it; bf16 history: iteration `it` is expanded into the (2 + n) staging planes first
  const Hist h32{(float*)hist, px, n};
  // (bf16 history: the stage kernels read the 16-bit slots themselves)
  const int hb = hist ? 0 : 1;
  auto slot16 = [&](int it) { return hist16 + (size_t)it * (2 + n) * px; };
  auto H_rhs = [&](int it) { return hist ? h32.rhs(it) : (float*)slot16(it); };
  auto H_x = [&](int it) { return hist ? h32.x(it) : (float*)(slot16(it) + px); };
  auto H_v = [&](int it, int i) { return hist ? h32.v(it, i) : (float*)(slot16(it) + (size_t)(2 + i) * px); };
  float* w = (float*)ws;
  float* gxz = w;
  float* gtot = w + px;
  float* grhs = w + 2 * px;
  float* tmp = w + 3 * px;
  float* gu_a = w + 4 * px;                  // n planes
  float* gu_b = gu_a + n * px;               // n planes
  float* set[2] = {gu_b + n * px, gu_b + 3 * n * px};   // each: gv[n] then gu[n]
  float* rho_a = set[1] + 2 * n * px;
  float* part_lam = rho_a + ((2 * B + 63) / 64) * 64;      // partial sums: [n B][nblk], [B][nblk], [B][nblk]
  float* part_a = part_lam + (size_t)DPX_MAX_TERMS * B * ad_partial_blocks(C, H, W);
  float* part_b = part_a + (size_t)B * ad_partial_blocks(C, H, W);
  const float one2[2] = {1.f, 1.f};
  const float* cur_gv[DPX_MAX_TERMS];
  const float* cur_gu[DPX_MAX_TERMS];
  for (int i = 0; i < n; ++i) {
    cur_gv[i] = gv_in ? gv_in[i] : nullptr;
    cur_gu[i] = gu_in ? gu_in[i] : nullptr;
  }
  bool off_started[DPX_MAX_TERMS] = {false, false, false, false};
  DPX_REQUIRE(n_off <= DPX_MAX_TERMS, "dpx_admm_unrolled_backward: at most %d offsets", DPX_MAX_TERMS);
  // the offsets' share of one iteration: goff_k += K_k g_rhs
  auto add_offsets = [&]() -> int {
    for (int k = 0; k < n_off; ++k) {
      if (!goff[k]) continue;
      if (off_otf[k]) {
        float* dst = off_started[k] ? tmp : goff[k];
        DPX_TRY(dpx_fft_conv(grhs, dst, off_otf[k], 0, B, C, H, W, table, spectrum_ws, stream));
        if (off_started[k]) {
          const float* xs[2] = {goff[k], tmp};
          DPX_TRY(dpx_lincomb(goff[k], 2, xs, one2, nullptr, B, (long)(px / B), stream));
        }
      } else {
        const float* xs[2] = {grhs, goff[k]};
        DPX_TRY(dpx_lincomb(goff[k], off_started[k] ? 2 : 1, xs, one2, nullptr, B, (long)(px / B), stream));
      }
      off_started[k] = true;
    }
    return DPX_OK;
  };
  // ---- power-of-two planes: TWO launches per backward iteration (+ the finishing launch), the mirror image of the forward loop --
  //        z(T-1) | rows | cols(T-1) | [rows^-1 + rhs(T-1) + z(T-2) + rows] | cols(T-2) | ... | cols(0) | rows^-1 | rhs(0)
  //      g_rhs and g_x stay in the Fourier domain between the stages (k_bwd_rows, dpx_bwd_rows.hip).  The offsets' gradient
  //      sum_t K g_rhs_t = K sum_t g_rhs_t is formed once at the end from the sum of the g_rhs images (emitted by the row kernel only
  //      when an offset gradient is wanted).  Knob unroll_bwd_staged: 0 = this loop where it applies, 2 = the image-domain fused
  //      stage below, 1 = the staged loop.
  const int slots = (T > 1 && tune(TUNE_UNROLL_BWD_STAGED) == 0) ? bwd_rows_slots(B, C, H, W, ad_partial_blocks(C, H, W)) : 0;
  if (slots > 0 && dpx_spectrum_bytes(B * C, H, W) > 0) {
    const hipStream_t st = (hipStream_t)stream;
    const int P = B * C;
    float2* spec_a = (float2*)spectrum_ws;
    float2* spec_b = (float2*)((char*)spectrum_ws + dpx_spectrum_bytes(P, H, W) / 2);
    float* abuf[2] = {gu_a, gu_b};
    int cur = 0;
    bool want_off = false;
    for (int k = 0; k < n_off; ++k) want_off = want_off || goff[k];
    {
      const int it = T - 1;
      dpx_bwd_term bt[DPX_MAX_TERMS];
      for (int i = 0; i < n; ++i)
        bt[i] = dpx_bwd_term{linops[i], proxes[i], alphas[i], 0, lam_tabs[i] + (size_t)it * B, H_v(it, i), cur_gv[i], cur_gu[i], abuf[cur] + i * px};
      DPX_TRY(zupdate_bwd_partials(gxz, bt, n, part_lam, hb, B, C, H, W, st));
      DPX_TRY(finish_iter(part_lam, part_a, part_b, glam + (size_t)it * n * B, nullptr, rho_tab + (size_t)it * B, n, B, C, H, W, st));
    }
    const float* g = gxz;
    if (gx) {
      const float* xs[2] = {gx, gxz};
      DPX_TRY(dpx_lincomb(gtot, 2, xs, one2, nullptr, B, (long)(px / B), stream));
      g = gtot;
    }
    DPX_TRY(rows_r2c_pow2(g, spec_a, P, H, W, table, st));
    SpecArgs sa{};
    sa.dd = (const float2*)dd;
    sa.eps = eps;
    sa.eps_num = 0.f;
    sa.scale = 1.0f / ((float)H * (float)W);
    // the iterations' partial sums side by side in the planes only the staged loop uses, finished by ONE launch behind the loop (a
    // finishing launch per iteration is 9 of the loop's 27 launches); if they do not fit: one finishing launch per iteration
    const long pstride = (long)(n + 2) * B * slots;
    const bool all_at_once = (size_t)(T - 1) * pstride <= (size_t)2 * n * px;
    for (int it = T - 1; it >= 1; --it) {
      const float* rho = rho_tab + (size_t)it * B;
      sa.rho = rho;
      if (all_at_once) {
        part_lam = set[0] + (size_t)(T - 1 - it) * pstride;
        part_a = part_lam + (size_t)n * B * slots;
        part_b = part_a + (size_t)B * slots;
      }
      DPX_TRY(cols_solve_pow2(spec_a, spec_b, sa, P, C, H, W, table, st));
      dpx_bwd_term bt[DPX_MAX_TERMS];
      const float* ain[DPX_MAX_TERMS];
      float* aout[DPX_MAX_TERMS];
      for (int i = 0; i < n; ++i) {
        bt[i] = dpx_bwd_term{linops[i], proxes[i], alphas[i], 0, lam_tabs[i] + (size_t)(it - 1) * B, H_v(it - 1, i), nullptr, nullptr, nullptr};
        ain[i] = abuf[cur] + i * px;
        aout[i] = abuf[cur ^ 1] + i * px;
      }
      DPX_TRY(bwd_rows_fused(spec_b, spec_a, H_x(it), H_rhs(it), rho, bt, n, ain, aout, want_off ? tmp : nullptr, it != T - 1, part_a, part_b, part_lam, hb,
                             B, C, H, W, table, st));
      if (!all_at_once) DPX_TRY(finish_iter_n(part_lam, part_a, part_b, glam + (size_t)(it - 1) * n * B, grho + (size_t)it * B, rho, n, B, slots, st));
      cur ^= 1;
    }
    if (all_at_once) {
      DPX_TRY(finish_all(set[0], pstride, glam, grho, rho_tab, n, B, slots, T, T - 1, st));
      part_lam = rho_a + ((2 * B + 63) / 64) * 64;          // (the last stage below: the workspace's own partial-sum rows again)
      part_a = part_lam + (size_t)DPX_MAX_TERMS * B * ad_partial_blocks(C, H, W);
      part_b = part_a + (size_t)B * ad_partial_blocks(C, H, W);
    }
    {
      const float* rho = rho_tab;
      sa.rho = rho;
      DPX_TRY(cols_solve_pow2(spec_a, spec_b, sa, P, C, H, W, table, st));
      DPX_TRY(rows_c2r_pow2(spec_b, grhs, P, H, W, table, st));
      if (want_off) {
        const float* xs[2] = {grhs, tmp};
        DPX_TRY(dpx_lincomb(tmp, 2, xs, one2, nullptr, B, (long)(px / B), stream));      // sum_t g_rhs_t
        for (int k = 0; k < n_off; ++k) {
          if (!goff[k]) continue;
          if (off_otf[k]) {
            DPX_TRY(dpx_fft_conv(tmp, goff[k], off_otf[k], 0, B, C, H, W, table, spectrum_ws, stream));
          } else {
            const float* one[1] = {tmp};
            DPX_TRY(dpx_lincomb(goff[k], 1, one, one2, nullptr, B, (long)(px / B), stream));
          }
        }
      }
      const float* gua[DPX_MAX_TERMS];
      for (int i = 0; i < n; ++i) gua[i] = abuf[cur] + (size_t)i * px;
      DPX_TRY(solve_rhs_bwd_partials(grhs, H_x(0), H_rhs(0), rho, linops, n, gv0, gu0, gua, part_a, part_b, hb, B, C, H, W, st));
      DPX_TRY(finish_iter(part_lam, part_a, part_b, nullptr, grho, rho, n, B, C, H, W, st));
    }
    return DPX_OK;
  }
  // ---- the loop with the rhs stage of iteration `it` and the z stage of iteration `it - 1` as ONE pass (k_rhs_z_bwd4, W % 4 == 0):
  //        z(T-1) | solve(T-1) | [rhs(T-1) + z(T-2)] | solve(T-2) | ... | [rhs(1) + z(0)] | solve(0) | rhs(0)
  //      4 launches per iteration (3 of them the transform) + 1 finishing launch instead of 5 + 1, no g_v / g_u planes in between.
  //      Knob unroll_bwd_staged = 1 keeps the staged loop below (A/B and tests).
  if (W % 4 == 0 && tune(TUNE_UNROLL_BWD_STAGED) != 1) {
    const hipStream_t st = (hipStream_t)stream;
    float* abuf[2] = {gu_a, gu_b};
    int cur = 0;
    // (experiment) the fused stage's last workgroup finishes the iteration's reductions itself (one arrival counter in the workspace's
    // scalar slot, zeroed once per call and reset by its last user) instead of a finishing launch per iteration.
    // Knob unroll_bwd_fold_finish, OFF: measured at 4x3x512^2 the step went from 1.55 - 1.61 to 1.78 ms -- 1536 workgroups taking a ticket
    // on one counter (~12 ns each) and one workgroup adding up 12 x 384 partials cost more than the 4.6-us finishing launch they replace.
    unsigned* counter = tune(TUNE_UNROLL_BWD_FOLD_FINISH) ? (unsigned*)rho_a : nullptr;
    if (counter && T > 1 && hipMemsetAsync(counter, 0, sizeof(unsigned), st) != hipSuccess) {
      set_error("dpx_admm_unrolled_backward: hipMemsetAsync failed");
      return DPX_ERR_LAUNCH;
    }
    {
      const int it = T - 1;
      dpx_bwd_term bt[DPX_MAX_TERMS];
      for (int i = 0; i < n; ++i)
        bt[i] = dpx_bwd_term{linops[i], proxes[i], alphas[i], 0, lam_tabs[i] + (size_t)it * B, H_v(it, i), cur_gv[i], cur_gu[i], abuf[cur] + i * px};
      DPX_TRY(zupdate_bwd_partials(gxz, bt, n, part_lam, hb, B, C, H, W, st));
      DPX_TRY(finish_iter(part_lam, part_a, part_b, glam + (size_t)it * n * B, nullptr, rho_tab + (size_t)it * B, n, B, C, H, W, st));
    }
    const float* g = gxz;
    if (gx) {
      const float* xs[2] = {gx, gxz};
      DPX_TRY(dpx_lincomb(gtot, 2, xs, one2, nullptr, B, (long)(px / B), stream));
      g = gtot;
    }
    for (int it = T - 1; it >= 1; --it) {
      const float* rho = rho_tab + (size_t)it * B;
      DPX_TRY(dpx_fourier_apply_inv(g, grhs, dd, rho, eps, B, C, H, W, table, spectrum_ws, stream));
      DPX_TRY(add_offsets());
      dpx_bwd_term bt[DPX_MAX_TERMS];
      const float* ain[DPX_MAX_TERMS];
      float* aout[DPX_MAX_TERMS];
      for (int i = 0; i < n; ++i) {
        bt[i] = dpx_bwd_term{linops[i], proxes[i], alphas[i], 0, lam_tabs[i] + (size_t)(it - 1) * B, H_v(it - 1, i), nullptr, nullptr, nullptr};
        ain[i] = abuf[cur] + i * px;
        aout[i] = abuf[cur ^ 1] + i * px;
      }
      float* glam_it = glam + (size_t)(it - 1) * n * B;
      float* grho_it = grho + (size_t)it * B;
      if (!rhs_z_bwd_fused(grhs, H_x(it), H_rhs(it), rho, bt, n, ain, aout, gxz, part_a, part_b, part_lam, hb, B, C, H, W, st, counter,
                           glam_it, grho_it)) {
        set_error("dpx_admm_unrolled_backward: fused stage refused a plane it was selected for");
        return DPX_ERR_LAUNCH;
      }
      if (!counter) DPX_TRY(finish_iter(part_lam, part_a, part_b, glam_it, grho_it, rho, n, B, C, H, W, st));
      cur ^= 1;
      g = gxz;
    }
    {
      const float* rho = rho_tab;
      DPX_TRY(dpx_fourier_apply_inv(g, grhs, dd, rho, eps, B, C, H, W, table, spectrum_ws, stream));
      DPX_TRY(add_offsets());
      const float* gua[DPX_MAX_TERMS];
      for (int i = 0; i < n; ++i) gua[i] = abuf[cur] + (size_t)i * px;
      DPX_TRY(solve_rhs_bwd_partials(grhs, H_x(0), H_rhs(0), rho, linops, n, gv0, gu0, gua, part_a, part_b, hb, B, C, H, W, st));
      DPX_TRY(finish_iter(part_lam, part_a, part_b, nullptr, grho, rho, n, B, C, H, W, st));
    }
    return DPX_OK;
  }
  for (int it = T - 1; it >= 0; --it) {
    const float* rho = rho_tab + (size_t)it * B;
    dpx_bwd_term bt[DPX_MAX_TERMS];
    for (int i = 0; i < n; ++i)
      bt[i] = dpx_bwd_term{linops[i], proxes[i], alphas[i], 0, lam_tabs[i] + (size_t)it * B, H_v(it, i), cur_gv[i], cur_gu[i], gu_a + i * px};
    DPX_TRY(zupdate_bwd_partials(gxz, bt, n, part_lam, hb, B, C, H, W, (hipStream_t)stream));
    const float* g = gxz;
    if (it == T - 1 && gx) {
      const float* xs[2] = {gx, gxz};
      DPX_TRY(dpx_lincomb(gtot, 2, xs, one2, nullptr, B, (long)(px / B), stream));
      g = gtot;
    }
    DPX_TRY(dpx_fourier_apply_inv(g, grhs, dd, rho, eps, B, C, H, W, table, spectrum_ws, stream));
    for (int k = 0; k < n_off; ++k) {
      if (!goff[k]) continue;
      if (off_otf[k]) {
        float* dst = off_started[k] ? tmp : goff[k];
        DPX_TRY(dpx_fft_conv(grhs, dst, off_otf[k], 0, B, C, H, W, table, spectrum_ws, stream));
        if (off_started[k]) {
          const float* xs[2] = {goff[k], tmp};
          DPX_TRY(dpx_lincomb(goff[k], 2, xs, one2, nullptr, B, (long)(px / B), stream));
        }
      } else {
        const float* xs[2] = {grhs, goff[k]};
        DPX_TRY(dpx_lincomb(goff[k], off_started[k] ? 2 : 1, xs, one2, nullptr, B, (long)(px / B), stream));
      }
      off_started[k] = true;
    }
    // gradients w.r.t. the previous iteration's v_i, u_i (the last step writes the caller's outputs directly)
    float* nv[DPX_MAX_TERMS];
    float* nu[DPX_MAX_TERMS];
    for (int i = 0; i < n; ++i) {
      nv[i] = it ? set[it & 1] + (size_t)i * px : gv0[i];
      nu[i] = it ? set[it & 1] + (size_t)(n + i) * px : gu0[i];
    }
    // the x-update's rho gradient and the rhs stage in one pass over g_rhs, with the sum of the two stages' shares of the dual
    // gradient folded in: gu_prev_i = gu_a_i - gv_i
    const float* gua[DPX_MAX_TERMS];
    for (int i = 0; i < n; ++i) gua[i] = gu_a + (size_t)i * px;
    DPX_TRY(solve_rhs_bwd_partials(grhs, H_x(it), H_rhs(it), rho, linops, n, nv, nu, gua, part_a, part_b, hb, B, C, H, W, (hipStream_t)stream));
    // ... and the iteration's three reductions (d/d lam_i, the two shares of d/d rho) finished by one launch
    DPX_TRY(finish_iter(part_lam, part_a, part_b, glam + (size_t)it * n * B, grho + (size_t)it * B, rho, n, B, C, H, W, (hipStream_t)stream));
    for (int i = 0; i < n; ++i) {
      cur_gv[i] = nv[i];
      cur_gu[i] = nu[i];
    }
  }
  return DPX_OK;
}

extern "C" int dpx_admm_unrolled_backward(const float* hist, const float* gx, const float* const* gv_in, const float* const* gu_in,
                                          float* const* gv0, float* const* gu0, float* grho, float* glam, float* const* goff,
                                          const void* const* off_otf, int n_off, const int* linops, const int* proxes, const float* alphas,
                                          int nterms, const float* rho_tab, const float* const* lam_tabs, int T, const void* dd, float eps,
                                          int B, int C, int H, int W, const void* table, void* spectrum_ws, void* ws, dpx_stream_t stream) {
  DPX_REQUIRE(hist, "dpx_admm_unrolled_backward: null history");
  return unrolled_backward_impl(hist, nullptr, gx, gv_in, gu_in, gv0, gu0, grho, glam, goff, off_otf, n_off, linops, proxes, alphas, nterms,
                                rho_tab, lam_tabs, T, dd, eps, B, C, H, W, table, spectrum_ws, ws, stream);
}

// bf16 history: the stage kernels read the 16-bit slots directly, the workspace is that of the fp32 pass
extern "C" size_t dpx_admm_unrolled_bwd_ws_bytes_bf16(int nterms, int B, int C, int H, int W) {
  return dpx_admm_unrolled_bwd_ws_bytes(nterms, B, C, H, W);
}
extern "C" int dpx_admm_unrolled_backward_bf16(const void* hist_bf16, const float* gx, const float* const* gv_in, const float* const* gu_in,
                                               float* const* gv0, float* const* gu0, float* grho, float* glam, float* const* goff,
                                               const void* const* off_otf, int n_off, const int* linops, const int* proxes,
                                               const float* alphas, int nterms, const float* rho_tab, const float* const* lam_tabs, int T,
                                               const void* dd, float eps, int B, int C, int H, int W, const void* table, void* spectrum_ws,
                                               void* ws, dpx_stream_t stream) {
  DPX_REQUIRE(hist_bf16 && ws, "dpx_admm_unrolled_backward_bf16: null pointer");
  return unrolled_backward_impl(nullptr, (const unsigned short*)hist_bf16, gx, gv_in, gu_in, gv0, gu0, grho, glam, goff, off_otf, n_off, linops,
                                proxes, alphas, nterms, rho_tab, lam_tabs, T, dd, eps, B, C, H, W, table, spectrum_ws, ws, stream);
}
