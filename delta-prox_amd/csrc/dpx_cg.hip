// Device-side control of the conjugate-gradient loop (reference dprox/linalg/solve/solver_cg.py:95-129).
//
// The operator application A(p) stays a sequence of the caller's kernels (any LinOp graph); everything else of an iteration
// -- the stop rule, beta, the direction update, alpha, the x / r updates -- is decided and executed on the GPU from a small
// state block, so the host issues a whole CG solve without a single read-back:
//
//   dpx_bgram(r) -> dpx_cg_test -> dpx_cg_direction -> [A(p): caller's kernels] -> dpx_bdot(p, Ap -> state) -> dpx_cg_update
//
// Stop rule (solver_cg.py:103-104): torch.linalg.norm(ravel(r), 2) <= cg_tol_i for ALL images i, i.e. the spectral norm of the
// [B, N] residual matrix against the smallest tolerance:  lambda_max(R R^T) <= tau^2, tau = min_i rtol * ||b_i||.  That is a
// statement about the B x B Gram matrix G only, and it holds iff  tau^2 I - G  is positive semidefinite -- decided here by an
// LDL^T factorisation in float64 on one wavefront (B <= 64; pivots >= 0), no eigen-solver needed.  Once the test succeeds
// the state's `done` flag is set and every later control kernel of the solve returns immediately (the iterate is frozen at
// exactly the reference's exit point; `n_done` = the iteration index the reference prints in "Converged at CG Iter").
#include <chrono>
#include <thread>
#include <cstdlib>

#include "dpx_cg_dev.h"

namespace dpx {

__global__ void k_cg_init(CgState S, const float* __restrict__ bnorm2, float rtol) {
  const int i = threadIdx.x;
  if (i < S.B) {
    const float nb = sqrtf(fmaxf(bnorm2[i], 0.f));          // ||b_i||  (torch.linalg.norm(ravel(b), 2, dim=-1))
    const float t = rtol * nb;                              // cg_tol_i in float32 like the reference
    S.tol2()[i] = t * t;
    S.gamma()[i] = 0.f;
    S.gamma_prev()[i] = 1.f;
    S.beta()[i] = 0.f;
    S.pAp()[i] = 1.f;
  }
  if (i == 0) {
    S.flags()[0] = 0;
    S.flags()[1] = -1;
    S.flags()[2] = 0;
    S.flags()[3] = 0;
  }
}

// one wavefront: the stop rule and beta / gamma of the next iteration (cg_test_block, dpx_cg_dev.h)
__global__ void __launch_bounds__(64) k_cg_test(CgState S, const float* __restrict__ G) {
  __shared__ double M[64 * 65];
  __shared__ int sh[2];
  cg_test_block(S, G, M, sh, -1.f);
}

// p = r + beta_b * p        (solver_cg.py:111-115; beta = 0 in the first iteration)
__global__ void k_cg_direction(float* __restrict__ p, const float* __restrict__ r, CgState S, long npb4) {
  if (S.flags()[0]) return;
  const int b = blockIdx.y;
  const float beta = S.beta()[b];
  float4* pb = (float4*)p + (long)b * npb4;
  const float4* rb = (const float4*)r + (long)b * npb4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npb4; i += (long)gridDim.x * blockDim.x) {
    const float4 rv = rb[i], pv = pb[i];
    pb[i] = make_float4(fmaf(beta, pv.x, rv.x), fmaf(beta, pv.y, rv.y), fmaf(beta, pv.z, rv.z), fmaf(beta, pv.w, rv.w));
  }
}
__global__ void k_cg_direction1(float* __restrict__ p, const float* __restrict__ r, CgState S, long npb) {
  if (S.flags()[0]) return;
  const int b = blockIdx.y;
  const float beta = S.beta()[b];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npb; i += (long)gridDim.x * blockDim.x)
    p[(long)b * npb + i] = fmaf(beta, p[(long)b * npb + i], r[(long)b * npb + i]);
}

// alpha_b = gamma_b / <p_b, A p_b>;  x += alpha p;  r -= alpha A p       (solver_cg.py:122-127)
__global__ void k_cg_update(float* __restrict__ x, float* __restrict__ r, const float* __restrict__ p, const float* __restrict__ Ap, CgState S,
                            long npb4) {
  if (S.flags()[0]) return;
  const int b = blockIdx.y;
  const float alpha = S.gamma()[b] / S.pAp()[b];
  float4* xb = (float4*)x + (long)b * npb4;
  float4* rb = (float4*)r + (long)b * npb4;
  const float4* pb = (const float4*)p + (long)b * npb4;
  const float4* qb = (const float4*)Ap + (long)b * npb4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npb4; i += (long)gridDim.x * blockDim.x) {
    const float4 pv = pb[i], qv = qb[i], xv = xb[i], rv = rb[i];
    xb[i] = make_float4(fmaf(alpha, pv.x, xv.x), fmaf(alpha, pv.y, xv.y), fmaf(alpha, pv.z, xv.z), fmaf(alpha, pv.w, xv.w));
    rb[i] = make_float4(fmaf(-alpha, qv.x, rv.x), fmaf(-alpha, qv.y, rv.y), fmaf(-alpha, qv.z, rv.z), fmaf(-alpha, qv.w, rv.w));
  }
}
__global__ void k_cg_update1(float* __restrict__ x, float* __restrict__ r, const float* __restrict__ p, const float* __restrict__ Ap, CgState S,
                             long npb) {
  if (S.flags()[0]) return;
  const int b = blockIdx.y;
  const float alpha = S.gamma()[b] / S.pAp()[b];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npb; i += (long)gridDim.x * blockDim.x) {
    const long e = (long)b * npb + i;
    x[e] = fmaf(alpha, p[e], x[e]);
    r[e] = fmaf(-alpha, Ap[e], r[e]);
  }
}

// start of a fused solve: r = b, x = p = 0, control flags and the two arrival counters cleared (one launch instead of a copy, two
// fills and the state initialisation; the tolerances are set by the first stop test from the Gram diagonal)
__global__ void k_cgm_start(float* __restrict__ x, float* __restrict__ r, float* __restrict__ p, const float* __restrict__ b, long n, int* __restrict__ flags,
                            unsigned* __restrict__ counters) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    r[i] = b[i];
    x[i] = 0.f;
    p[i] = 0.f;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    flags[0] = 0;
    flags[1] = -1;
    flags[2] = 0;
    flags[3] = 0;
    counters[0] = 0u;
    counters[1] = 0u;
  }
}

__global__ void k_square(float* __restrict__ out, const float* __restrict__ w, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = w[i] * w[i];
}

}  // namespace dpx

using namespace dpx;

namespace dpx {
int masked_normal_apply(const float* p, float* Ap, float2* z, const float* mask2, int mask_images, const float* rho, float c, const int* done,
                        int B, int H, int W, const void* table, hipStream_t s);     // dpx_fft.hip
int masked_normal_apply_fused(float* p, const float* r, float* Ap, float2* z, const float* mask, int mask_images, const float* rho, float c,
                              float* state, float* dotws, unsigned* counter, int B, int H, int W, const void* table, hipStream_t s);
size_t masked_normal_fused_ws_floats(int B, int H, int W);
int gram_test_fused(float* r, float* G, void* state, int B, long n_per_batch, void* ws, unsigned* counter, float init_rtol, float* x, const float* p,
                    const float* Ap, int* host_flags, int host_tag, hipStream_t s);   // dpx_elementwise.hip
}

extern "C" size_t dpx_cg_state_bytes(int B) { return B > 0 ? (size_t)(5 * B + 4) * sizeof(float) : 0; }

extern "C" int dpx_cg_init(void* state, const float* bnorm2, float rtol, int B, dpx_stream_t stream) {
  DPX_REQUIRE(state && bnorm2 && B >= 1 && B <= 64, "dpx_cg_init: the device-side stop rule handles 1..64 systems per solve, got %d", B);
  DPX_LAUNCH("k_cg_init", k_cg_init, dim3(1), dim3(64), 0, (hipStream_t)stream, CgState{(float*)state, B}, bnorm2, rtol);
  return launch_status("dpx_cg_init");
}

extern "C" int dpx_cg_test(void* state, const float* gram, int B, dpx_stream_t stream) {
  DPX_REQUIRE(state && gram && B >= 1 && B <= 64, "dpx_cg_test: bad arguments");
  DPX_LAUNCH("k_cg_test", k_cg_test, dim3(1), dim3(64), 0, (hipStream_t)stream, CgState{(float*)state, B}, gram);
  return launch_status("dpx_cg_test");
}

extern "C" int dpx_cg_direction(float* p, const float* r, void* state, int B, long n_per_batch, dpx_stream_t stream) {
  DPX_REQUIRE(p && r && state && B >= 1 && n_per_batch > 0, "dpx_cg_direction: bad arguments");
  const CgState S{(float*)state, B};
  if (n_per_batch % 4 == 0 && ((size_t)p % 16 == 0) && ((size_t)r % 16 == 0))
    DPX_LAUNCH("k_cg_direction", k_cg_direction, dim3(grid_for(n_per_batch / 4, 256, 1024), B), dim3(256), 0, (hipStream_t)stream, p, r, S,
               n_per_batch / 4);
  else
    DPX_LAUNCH("k_cg_direction", k_cg_direction1, dim3(grid_for(n_per_batch, 256, 1024), B), dim3(256), 0, (hipStream_t)stream, p, r, S, n_per_batch);
  return launch_status("dpx_cg_direction");
}

extern "C" int dpx_cg_update(float* x, float* r, const float* p, const float* Ap, void* state, int B, long n_per_batch, dpx_stream_t stream) {
  DPX_REQUIRE(x && r && p && Ap && state && B >= 1 && n_per_batch > 0, "dpx_cg_update: bad arguments");
  const CgState S{(float*)state, B};
  const bool al = ((size_t)x % 16 == 0) && ((size_t)r % 16 == 0) && ((size_t)p % 16 == 0) && ((size_t)Ap % 16 == 0);
  if (n_per_batch % 4 == 0 && al)
    DPX_LAUNCH("k_cg_update", k_cg_update, dim3(grid_for(n_per_batch / 4, 256, 1024), B), dim3(256), 0, (hipStream_t)stream, x, r, p, Ap, S,
               n_per_batch / 4);
  else
    DPX_LAUNCH("k_cg_update", k_cg_update1, dim3(grid_for(n_per_batch, 256, 1024), B), dim3(256), 0, (hipStream_t)stream, x, r, p, Ap, S, n_per_batch);
  return launch_status("dpx_cg_update");
}

/* zero-fill on the stream (hipMemsetAsync): the solver's fresh iterates / workspaces */
extern "C" int dpx_zero(void* p, size_t bytes, dpx_stream_t stream) {
  DPX_REQUIRE(p || bytes == 0, "dpx_zero: null pointer");
  if (bytes && hipMemsetAsync(p, 0, bytes, (hipStream_t)stream) != hipSuccess) {
    set_error("dpx_zero: hipMemsetAsync failed");
    return DPX_ERR_LAUNCH;
  }
  return DPX_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// The whole CG solve of config 4's x-update in one call:  (A^H A + n_identity * rho_b I) x = b,  A = mask * fft2c (centred,
// orthonormal), x0 = 0 -- least_squares.solve_cg (proxfn/sum_square.py:158-197) over linalg/solve/solver_cg.py:56-136 for the
// masked-Fourier data term.  The operator application is dpx_cfft2 -> mask^2 -> dpx_cfft2^-1 -> real part + n rho p; the loop
// control is the device-side state machine above; the host side of THIS function only issues kernels and looks at the `done`
// flag LAG iterations late through a pinned buffer (never blocks on the newest work).  Returns the exit iteration (>= 0, the
// number the reference prints in "Converged at CG Iter"; max_iters if it did not converge) or a negative status.
// ws (dpx_cg_masked_fft_ws_bytes): r, p, Ap [B n] floats; two complex [B n] buffers (the second one unused since the operator
// became three fused launches); mask^2; state; Gram; dot workspace.
// ---------------------------------------------------------------------------------------------------------------------
namespace dpx { bool masked_normal_fits(int H, int W); }
// 1 when dpx_cg_masked_fft takes this batch of planes (B <= 64 images, planes within its LDS-resident transforms): callers fall
// back to the generic cg() loop on the same primitives otherwise
extern "C" int dpx_cg_masked_fft_supported(int B, int H, int W) { return B >= 1 && B <= 64 && H > 0 && W > 0 && dpx::masked_normal_fits(H, W) ? 1 : 0; }
constexpr int DPX_CG_MAX_DEVICES = 64;
extern "C" size_t dpx_cg_masked_fft_ws_bytes(int B, int H, int W, int mask_images) {
  const size_t n = (size_t)H * W;
  return (3 * B * n + 4 * B * n + (size_t)mask_images * n + 5 * B + 4 + (size_t)B * B + 64) * sizeof(float) + dpx_bdot_ws_bytes(B, (long)n) + 256 +
         (dpx::masked_normal_fused_ws_floats(B, H, W) + 8) * sizeof(float);
}

// Spin until the device has stored `tag` at *slot (host-coherent memory, written by the finishing workgroup of a launch already in the
// stream, together with the word next to it: one 8-byte store).  Once a millisecond (by the clock, not by a spin count) it looks at the stream:
// a drained stream makes every store of its kernels visible whatever the platform does with device stores to pinned memory in mid-kernel, so
// the answer is then final -- the tag, or false (also for a failed stream).  The one-minute limit guards against a store that never comes
// although the stream drains normally -- it only runs while the stream reports work in flight AFTER the tag's own launch could have run, i.e.
// it is restarted whenever the stream makes progress the host can see: a stream with minutes of earlier work in front of the solve is waited for.
static bool wait_for_tag(volatile int* slot, int tag, hipStream_t s) {
  using clock = std::chrono::steady_clock;
  auto t_query = clock::now(), t_limit = t_query;
  for (unsigned spins = 1;; ++spins) {
    if (__atomic_load_n((const int*)slot, __ATOMIC_ACQUIRE) == tag) return true;
    if ((spins & 0x3ffu) == 0) {                         // (the clock itself is read every 1024 spins: tens of microseconds)
      const auto now = clock::now();
      if (now - t_query >= std::chrono::milliseconds(1)) {
        t_query = now;
        const hipError_t q = hipStreamQuery(s);
        if (q != hipErrorNotReady) return q == hipSuccess && __atomic_load_n((const int*)slot, __ATOMIC_ACQUIRE) == tag;
        if (now - t_limit > std::chrono::seconds(60)) {
          // a minute of "not ready": fall back to the blocking wait instead of failing a solve that merely queued behind a long job
          if (hipStreamSynchronize(s) != hipSuccess) return false;
          return __atomic_load_n((const int*)slot, __ATOMIC_ACQUIRE) == tag;
        }
      }
    }
#if defined(__x86_64__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield" ::: "memory");
#else
    std::this_thread::yield();
#endif
  }
}

namespace dpx {
// the fused branch (B <= cg_fused_max_b, not cg_unfused) takes this batch: its start state (r = b, x = p = 0, flags and arrival counters
// cleared -- k_cgm_start) can then be written by the producer of b itself (dpx_admm_cg_pnp_iter's tail pass) through cg_masked_fft_start_ptrs
bool cg_masked_fft_is_fused(int B) {
  const int fused_max_b = tune(TUNE_CG_FUSED_MAX_B);
  return B <= (fused_max_b > 32 ? 32 : fused_max_b) && tune(TUNE_CG_UNFUSED) == 0;
}
// the one place that knows how dpx_cg_masked_fft carves up its workspace (dpx_cg_masked_fft_ws_bytes)
struct CgWsLayout {
  float *r, *p, *Ap;
  float2 *z0, *z1;
  float *mask2, *state, *gram, *dotws, *fdot;
  unsigned* counters;
};
static CgWsLayout cg_ws_layout(void* ws, int B, int H, int W, int mask_images) {
  const size_t n = (size_t)H * W;
  CgWsLayout L;
  L.r = (float*)ws;
  L.p = L.r + (size_t)B * n;
  L.Ap = L.p + (size_t)B * n;
  L.z0 = (float2*)(L.Ap + (size_t)B * n + (((size_t)3 * B * n) & 1));      // (8-byte aligned also when 3 B n is odd; the slack is in ws_bytes)
  L.z1 = L.z0 + (size_t)B * n;
  L.mask2 = (float*)(L.z1 + (size_t)B * n);
  L.state = L.mask2 + (size_t)mask_images * n;
  L.gram = L.state + 5 * B + 4;
  L.dotws = L.gram + (((size_t)B * B + 63) / 64) * 64;
  L.fdot = (float*)((char*)L.dotws + dpx_bdot_ws_bytes(B, (long)n));                           // (fused branch)
  L.counters = (unsigned*)(L.fdot + dpx::masked_normal_fused_ws_floats(B, H, W));
  return L;
}
CgStartPtrs cg_masked_fft_start_ptrs(void* ws, int B, int H, int W, int mask_images) {
  const CgWsLayout L = cg_ws_layout(ws, B, H, W, mask_images);
  return CgStartPtrs{L.r, L.p, (int*)(L.state + 5 * B), L.counters};
}
}  // namespace dpx

extern "C" int dpx_cg_masked_fft(float* x, const float* b, const float* mask, int mask_images, const float* rho, float n_identity, float rtol,
                                 int max_iters, int B, int H, int W, const void* table, void* ws, dpx_stream_t stream) {
  return dpx::cg_masked_fft_run(x, b, mask, mask_images, rho, n_identity, rtol, max_iters, B, H, W, table, ws, false, nullptr, stream);
}

// started: the start state is in place already (see cg_masked_fft_start_ptrs; b is not looked at); fused branch only
int dpx::cg_masked_fft_run(float* x, const float* b, const float* mask, int mask_images, const float* rho, float n_identity, float rtol, int max_iters,
                           int B, int H, int W, const void* table, void* ws, bool started, CgSpeculate* spec, dpx_stream_t stream) {
  DPX_REQUIRE(x && (b || started) && mask && rho && table && ws && B >= 1 && B <= 64 && H > 0 && W > 0 && max_iters >= 0 && (mask_images == 1 || mask_images == B),
              "dpx_cg_masked_fft: bad arguments (B = %d must be 1..64)", B);
  DPX_REQUIRE(!started || cg_masked_fft_is_fused(B), "dpx_cg_masked_fft: a pre-started solve needs the fused branch (B = %d)", B);
  hipStream_t s = (hipStream_t)stream;
  const long n = (long)H * W;
  const CgWsLayout WL = cg_ws_layout(ws, B, H, W, mask_images);
  float *r = WL.r, *p = WL.p, *Ap = WL.Ap, *mask2 = WL.mask2, *state = WL.state, *gram = WL.gram, *dotws = WL.dotws;
  float2 *z0 = WL.z0, *z1 = WL.z1;
  int* flags = (int*)(state + 5 * B);
  float* pAp = state + 3 * B;

  // pinned ring for the flag read-backs + its events: host-side resources, one set per (host thread, device) -- an event belongs to
  // the device that was current when it was created, and two host threads solving at once must not share slots (this function
  // returns only after it has waited for its own last event, so a thread has at most one solve in flight)
  struct Ring {
    int* pin = nullptr;
    int* pin_dev = nullptr;     // the same memory as the device addresses it
    hipEvent_t ev[4];
    int hint = -1;              // exit iteration of this thread's previous fused solve on this device (-1: none yet)
    unsigned seq = 0;           // solves issued by this thread on this device: the high bits of the tags the test kernels store
  };
  static thread_local Ring rings[DPX_CG_MAX_DEVICES];
  int devid = 0;
  if (hipGetDevice(&devid) != hipSuccess || devid < 0 || devid >= DPX_CG_MAX_DEVICES) {
    set_error("dpx_cg_masked_fft: hipGetDevice failed / device id %d out of range", devid);
    return DPX_ERR_LAUNCH;
  }
  Ring& R = rings[devid];
  if (!R.pin) {
    int* pnew = nullptr;
    if (hipHostMalloc((void**)&pnew, 4 * 4 * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
      set_error("dpx_cg_masked_fft: hipHostMalloc failed");
      return DPX_ERR_LAUNCH;
    }
    for (int i = 0; i < 4; ++i)
      if (hipEventCreateWithFlags(&R.ev[i], hipEventDisableTiming) != hipSuccess) {
        for (int k = 0; k < i; ++k) hipEventDestroy(R.ev[k]);
        hipHostFree(pnew);
        set_error("dpx_cg_masked_fft: hipEventCreate failed");
        return DPX_ERR_LAUNCH;
      }
    void* dptr = nullptr;
    if (hipHostGetDevicePointer(&dptr, pnew, 0) != hipSuccess || !dptr) {
      for (int k = 0; k < 4; ++k) hipEventDestroy(R.ev[k]);
      hipHostFree(pnew);
      set_error("dpx_cg_masked_fft: hipHostGetDevicePointer failed");
      return DPX_ERR_LAUNCH;
    }
    R.pin = pnew;
    R.pin_dev = (int*)dptr;
  }
  int* pin = R.pin;
  int* pin_dev = R.pin_dev;
  hipEvent_t* ev = R.ev;
  for (int i = 0; i < 16; ++i) pin[i] = 0;             // nothing left over from an earlier solve can read as "done"
#define CG_HIP(call)                                                        \
  do {                                                                      \
    if ((call) != hipSuccess) {                                             \
      set_error("dpx_cg_masked_fft: %s failed (%s)", #call, hipGetErrorString(hipGetLastError())); \
      return DPX_ERR_LAUNCH;                                                \
    }                                                                       \
  } while (0)
#define CG_TRY(call)                 \
  do {                               \
    const int rc_ = (call);          \
    if (rc_ != DPX_OK) return rc_;   \
  } while (0)
  const int n_it = max_iters < (int)((long)B * n) ? max_iters : (int)((long)B * n);
  constexpr int LAG = 2;
  static_assert(LAG < 4, "the four-slot ring of stop-test results and the 1024-iteration tag wrap are safe only while fewer than 4 tests are in flight");
  int done_it = n_it;
  bool done = false;
  int last = -1;
  // ---- B <= 8: the fused iteration -- 5 launches (Gram pass + its finish + the stop rule; the operator's three kernels with the
  //      direction update in the first one's load and <p, Ap> in the last one's store; the x / r update) instead of 10, and one
  //      launch instead of seven in front of the loop.  The partial sums of a launch are finished by its LAST workgroup to arrive
  //      (dpx_last_block: no workgroup waits for another); same state machine, same exit iteration.
  const bool unfused = tune(TUNE_CG_UNFUSED) != 0;      // (A/B and tests: the step-by-step sequence below)
  const bool split_update = tune(TUNE_CG_SPLIT_UPDATE) != 0;      // (A/B: the 5-launch form with its own update kernel and a flag transfer)
  // (the kernels hold up to 32 residuals, DPX_CG_FUSED_MAX_B: measured at 16 / 32 images of 320^2: 2.28 / 4.64 ms per outer iteration
  //  against 2.25 / 4.08 on the step-by-step sequence -- the Gram pass with its finish in one workgroup stops paying beyond 8)
  const int fused_max_b = tune(TUNE_CG_FUSED_MAX_B);
  if (B <= (fused_max_b > 32 ? 32 : fused_max_b) && !unfused) {
    float* fdot = WL.fdot;
    unsigned* counters = WL.counters;
    if (!started)
      DPX_LAUNCH("k_cgm_start", k_cgm_start, dim3(grid_for((long)B * n, 256, 1024)), dim3(256), 0, s, x, r, p, b, (long)B * n, flags, counters);
    // How the host learns that the test of iteration j has run: the finishing workgroup of that launch stores (done | n_done << 1, tag_j) as one 8-byte word into slot
    // j & 3 of a host-coherent ring, and the host spins on the tag -- slots in order, so that a launch
    // that found the solve converged already (and stores nothing) is never waited for.  No event, no marker packet between the launches
    // (an event per iteration left ~5.5 us of idle stream in front of every test kernel: knob cg_event_wait = 1 is that form).
    const bool poll = !split_update && tune(TUNE_CG_EVENT_WAIT) == 0;
    const unsigned seq = ++R.seq;
    const int hint = (spec && spec->hint >= 0) ? spec->hint : R.hint;      // (a caller that knows better: the same solve of the previous run)
    auto tag_of = [&](int j) { return (int)((((seq & 0x1fffffu) << 10) | (unsigned)((j & 1023) + 1)) & 0x7fffffffu); };
    // (a slot the test kernel stored: low word = done | n_done << 1, high word = tag; a slot copied from the device flags -- the
    //  split_update form -- : (done, n_done, ...) as they are)
    auto slot_done = [&](int j) { return split_update ? pin[(j & 3) * 4] != 0 : (pin[(j & 3) * 4] & 1) != 0; };
    auto slot_it = [&](int j) { return split_update ? pin[(j & 3) * 4 + 1] : pin[(j & 3) * 4] >> 1; };
    int inspected = 0;                                       // slots of iterations [0, inspected) have been looked at
    auto inspect = [&](int upto) -> int {                    // 1: converged (done / done_it set), 0: not yet, < 0: error
      for (; inspected <= upto; ++inspected) {
        volatile int* sl = pin + (inspected & 3) * 4;
        if (!wait_for_tag(sl + 1, tag_of(inspected), s)) {
          set_error("dpx_cg_masked_fft: the stop test of iteration %d never reported (stream failed or timed out)", inspected);
          return DPX_ERR_LAUNCH;
        }
        if (sl[0] & 1) {
          done = true;
          done_it = sl[0] >> 1;
          return 1;
        }
      }
      return 0;
    };
    for (int it = 0; it < n_it; ++it) {
      if (it >= LAG) {
        if (poll) {
          const int rc = inspect(it - LAG);
          if (rc < 0) return rc;
          if (rc) break;
        } else {
          CG_HIP(hipEventSynchronize(ev[(it - LAG) & 3]));
          if (slot_done(it - LAG)) {
            done = true;
            done_it = slot_it(it - LAG);
            break;
          }
        }
      }
      // 4 launches: [x / r update of the previous iteration + Gram pass + stop rule, flags to the pinned slot] + the operator's three
      if (split_update) {
        CG_TRY(dpx::gram_test_fused(r, gram, state, B, n, dotws, counters, it == 0 ? rtol : -1.f, nullptr, nullptr, nullptr, nullptr, 0, s));
      } else {
        CG_TRY(dpx::gram_test_fused(r, gram, state, B, n, dotws, counters, it == 0 ? rtol : -1.f, it > 0 ? x : nullptr, p, Ap, pin_dev + (it & 3) * 4,
                                    tag_of(it), s));
        // Consecutive solves of one outer loop exit at the same iteration almost always (config 4: 2, 3, 3, 3, ...).  At the iteration the
        // previous solve stopped at, look at THIS test's flag right away -- one host round trip, which the end of the solve pays
        // anyway -- instead of finding out two iterations (seven empty launches) later.  A miss costs that one wait.
        if (it == hint && it > 0 && !tune(TUNE_CG_NO_HINT)) {
          if (spec && poll && !spec->launched) {            // the caller's next stage, predicated on this test's verdict (CgSpeculate; the
                                                            //  in-order look at the slots below is what makes `valid` exact)
            spec->launched = true;
            CG_TRY(spec->launch(spec->ctx, flags, stream));
          }
          if (poll) {
            const int rc = inspect(it);
            if (rc < 0) return rc;
            if (rc) {
              last = -1;                                    // (nothing pending behind this point: the update below is not needed either)
              if (spec) spec->valid = spec->launched;        // (done now or earlier: the flag was set when the predicated launch ran)
              break;
            }
          } else {
            CG_HIP(hipEventRecord(ev[it & 3], s));
            CG_HIP(hipEventSynchronize(ev[it & 3]));
            if (slot_done(it)) {
              done = true;
              done_it = slot_it(it);
              last = -1;
              if (spec) spec->valid = spec->launched;
              break;
            }
          }
        }
      }
      CG_TRY(dpx::masked_normal_apply_fused(p, r, Ap, z0, mask, mask_images, rho, n_identity, state, fdot, counters + 1, B, H, W, table, s));
      if (split_update) {
        CG_TRY(dpx_cg_update(x, r, p, Ap, state, B, n, stream));
        CG_HIP(hipMemcpyAsync(pin + (it & 3) * 4, flags, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
      }
      if (!poll) CG_HIP(hipEventRecord(ev[it & 3], s));
      last = it;
    }
    if (!split_update && last >= 0 && !done) CG_TRY(dpx_cg_update(x, r, p, Ap, state, B, n, stream));     // the last iteration's update (max_iters reached without convergence)
    if (!done && last >= 0) {
      // the iterations the loop did not look at yet, oldest first (a launch behind the converged one leaves its slot untouched)
      if (poll) {
        const int rc = inspect(last);
        if (rc < 0) return rc;
      } else {
        CG_HIP(hipEventSynchronize(ev[last & 3]));
        for (int j = (last - LAG + 1 > 0 ? last - LAG + 1 : 0); j <= last; ++j)
          if (slot_done(j)) {
            done_it = slot_it(j);
            break;
          }
      }
    }
    R.hint = done_it;
    const int st = launch_status("dpx_cg_masked_fft");
    return st != DPX_OK ? st : done_it;
  }
  // r = b, x = p = 0, tolerances
  CG_HIP(hipMemcpyAsync(r, b, (size_t)B * n * sizeof(float), hipMemcpyDeviceToDevice, s));
  CG_HIP(hipMemsetAsync(x, 0, (size_t)B * n * sizeof(float), s));
  CG_HIP(hipMemsetAsync(p, 0, (size_t)B * n * sizeof(float), s));
  DPX_LAUNCH("k_square", k_square, dim3(grid_for((long)mask_images * n, 256, 1024)), dim3(256), 0, s, mask2, mask, (long)mask_images * n);
  CG_TRY(dpx_bdot(b, b, gram, B, n, dotws, stream));                      // <b_i, b_i> (gram reused as scratch)
  CG_TRY(dpx_cg_init(state, gram, rtol, B, stream));
  for (int it = 0; it < n_it; ++it) {
    if (it >= LAG) {
      CG_HIP(hipEventSynchronize(ev[(it - LAG) & 3]));
      if (pin[((it - LAG) & 3) * 4]) {
        done = true;
        done_it = pin[((it - LAG) & 3) * 4 + 1];
        break;
      }
    }
    CG_TRY(dpx_bgram(r, gram, B, n, dotws, stream));
    CG_TRY(dpx_cg_test(state, gram, B, stream));
    CG_TRY(dpx_cg_direction(p, r, state, B, n, stream));
    // Ap = Re F^-1 mask^2 F p + n rho p: three launches (row transform of the real p, column transform - mask - inverse column
    // transform, inverse row transform + real part + rho p) instead of the seven of cfft2 / mask / cfft2^-1 around two copies
    CG_TRY(masked_normal_apply(p, Ap, z0, mask2, mask_images, rho, n_identity, (const int*)flags, B, H, W, table, s));
    CG_TRY(dpx_bdot(p, Ap, pAp, B, n, dotws, stream));
    CG_TRY(dpx_cg_update(x, r, p, Ap, state, B, n, stream));
    CG_HIP(hipMemcpyAsync(pin + (it & 3) * 4, flags, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
    CG_HIP(hipEventRecord(ev[it & 3], s));
    last = it;
  }
  if (!done && last >= 0) {
    CG_HIP(hipEventSynchronize(ev[last & 3]));
    if (pin[(last & 3) * 4]) done_it = pin[(last & 3) * 4 + 1];
  }
#undef CG_TRY
#undef CG_HIP
  const int st = launch_status("dpx_cg_masked_fft");
  return st != DPX_OK ? st : done_it;
}
