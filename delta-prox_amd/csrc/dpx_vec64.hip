// Vector primitives of the Krylov solvers in float64, and the max-|x| reduction of pcg's stop rule.
//
// The reference's solvers (dprox/linalg/solve/solver_cg.py:56-233: cg, cg2, pcg) are dtype-generic torch code and its own tests
// run them on float64 systems at rtol 1e-8 (tests/linalg/test_linear_solver.py:57-80, test_linear_solver_batch.py:27-49,
// test_linear_solver_torch.py:33-57).  The hot path of this backend is float32; these kernels let the same solvers keep a float64
// right-hand side in float64 on the device instead of rounding it: batched dots and the B x B Gram matrix (deterministic: per-block
// partials + a finishing pass, no atomics), linear combinations, and max |x| -- for float32 and float64.
#include "dpx_common.h"

namespace dpx {

template <class T> __device__ __forceinline__ T wave_sum_t(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
template <class T> __device__ __forceinline__ T wave_max_t(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const T w = __shfl_xor(v, o);
    v = w > v ? w : v;
  }
  return v;
}
// sum (MAX = false) or maximum over the block; valid in thread 0
template <class T, bool MAX> __device__ __forceinline__ T block_reduce_t(T v, T* sh) {
  v = MAX ? wave_max_t(v) : wave_sum_t(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wid] = v;
  __syncthreads();
  const int nw = (blockDim.x + 63) >> 6;
  T r = (threadIdx.x < nw) ? sh[threadIdx.x] : T(0);
  if (wid == 0) r = MAX ? wave_max_t(r) : wave_sum_t(r);
  return r;
}

// grid (nblk, B, Bj): partial[(bi * Bj + bj) * nblk + blk] = sum over a slice of <x[bi], y[gram ? bj : bi]>
__global__ void k_dot_partial_f64(const double* __restrict__ x, const double* __restrict__ y, double* __restrict__ partial, long npb, int gram) {
  __shared__ double sh[16];
  const int bi = blockIdx.y, bj = gram ? blockIdx.z : bi;
  const double* xa = x + (long)bi * npb;
  const double* yb = y + (long)bj * npb;
  double acc = 0.0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npb; i += (long)gridDim.x * blockDim.x) acc = fma(xa[i], yb[i], acc);
  acc = block_reduce_t<double, false>(acc, sh);
  if (threadIdx.x == 0) partial[((long)bi * gridDim.z + blockIdx.z) * gridDim.x + blockIdx.x] = acc;
}
__global__ void k_sum_finish_f64(const double* __restrict__ partial, double* __restrict__ out, int nblk) {
  __shared__ double sh[16];
  double acc = 0.0;
  for (int i = threadIdx.x; i < nblk; i += blockDim.x) acc += partial[(long)blockIdx.x * nblk + i];
  acc = block_reduce_t<double, false>(acc, sh);
  if (threadIdx.x == 0) out[blockIdx.x] = acc;
}

struct LinPack64 {
  const double* x[4];
  const double* cb[4];
  double c[4];
  int n;
};
__global__ void k_lincomb_f64(double* __restrict__ out, LinPack64 L, int B, long npb) {
  const long total = (long)B * npb;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / npb);
    double acc = 0.0;
    for (int t = 0; t < L.n; ++t) {
      const double c = L.c[t] * (L.cb[t] ? L.cb[t][b] : 1.0);
      acc = (t == 0) ? c * L.x[t][i] : fma(c, L.x[t][i], acc);
    }
    out[i] = acc;
  }
}

// partial[blk] = max |x| over a slice, then out[0] = max over the partials (same kernel, nblk = 1 block over the partials)
template <class T> __global__ void k_absmax(const T* __restrict__ x, T* __restrict__ out, long n) {
  __shared__ T sh[16];
  T m = T(0);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const T a = x[i] < T(0) ? -x[i] : x[i];
    m = a > m ? a : m;
  }
  m = block_reduce_t<T, true>(m, sh);
  if (threadIdx.x == 0) out[blockIdx.x] = m;
}

static int red_blocks(long n) {
  long g = (n + 256 * 8 - 1) / (256 * 8);
  return (int)(g > 256 ? 256 : (g < 1 ? 1 : g));
}

}  // namespace dpx

using namespace dpx;

extern "C" size_t dpx_bdot_f64_ws_bytes(int B, long n_per_batch) { return (size_t)B * B * red_blocks(n_per_batch) * sizeof(double); }

extern "C" int dpx_bdot_f64(const double* x, const double* y, double* out, int B, long n_per_batch, void* ws, dpx_stream_t stream) {
  DPX_REQUIRE(x && y && out && ws && B > 0 && n_per_batch > 0, "dpx_bdot_f64: bad arguments");
  const int nblk = red_blocks(n_per_batch);
  DPX_LAUNCH("k_dot_partial_f64", k_dot_partial_f64, dim3(nblk, B, 1), dim3(256), 0, (hipStream_t)stream, x, y, (double*)ws, n_per_batch, 0);
  DPX_LAUNCH("k_sum_finish_f64", k_sum_finish_f64, dim3(B), dim3(256), 0, (hipStream_t)stream, (const double*)ws, out, nblk);
  return launch_status("dpx_bdot_f64");
}

extern "C" int dpx_bgram_f64(const double* r, double* out, int B, long n_per_batch, void* ws, dpx_stream_t stream) {
  DPX_REQUIRE(r && out && ws && B > 0 && B <= 1024 && n_per_batch > 0, "dpx_bgram_f64: bad arguments");
  const int nblk = red_blocks(n_per_batch);
  DPX_LAUNCH("k_dot_partial_f64", k_dot_partial_f64, dim3(nblk, B, B), dim3(256), 0, (hipStream_t)stream, r, r, (double*)ws, n_per_batch, 1);
  DPX_LAUNCH("k_sum_finish_f64", k_sum_finish_f64, dim3(B * B), dim3(256), 0, (hipStream_t)stream, (const double*)ws, out, nblk);
  return launch_status("dpx_bgram_f64");
}

extern "C" int dpx_lincomb_f64(double* out, int n, const double* const* x, const double* coef, const double* const* coef_b, int B,
                               long n_per_batch, dpx_stream_t stream) {
  DPX_REQUIRE(out && x && coef && n >= 1 && n <= 4 && B > 0 && n_per_batch > 0, "dpx_lincomb_f64: bad arguments");
  LinPack64 L;
  L.n = n;
  for (int i = 0; i < n; ++i) {
    DPX_REQUIRE(x[i], "dpx_lincomb_f64: operand %d is null", i);
    L.x[i] = x[i];
    L.c[i] = coef[i];
    L.cb[i] = coef_b ? coef_b[i] : nullptr;
  }
  DPX_LAUNCH("k_lincomb_f64", k_lincomb_f64, dim3(grid_for(B * n_per_batch, 256, 8192)), dim3(256), 0, (hipStream_t)stream, out, L, B, n_per_batch);
  return launch_status("dpx_lincomb_f64");
}

// out[0] = max_i |x_i|; is_f64 selects the element type of x, out and ws (ws: 256 elements)
extern "C" int dpx_absmax(const void* x, void* out, long n, int is_f64, void* ws, dpx_stream_t stream) {
  DPX_REQUIRE(x && out && ws && n > 0, "dpx_absmax: bad arguments");
  const int nblk = red_blocks(n);
  hipStream_t s = (hipStream_t)stream;
  if (is_f64) {
    DPX_LAUNCH("k_absmax", k_absmax<double>, dim3(nblk), dim3(256), 0, s, (const double*)x, (double*)ws, n);
    DPX_LAUNCH("k_absmax", k_absmax<double>, dim3(1), dim3(256), 0, s, (const double*)ws, (double*)out, (long)nblk);
  } else {
    DPX_LAUNCH("k_absmax", k_absmax<float>, dim3(nblk), dim3(256), 0, s, (const float*)x, (float*)ws, n);
    DPX_LAUNCH("k_absmax", k_absmax<float>, dim3(1), dim3(256), 0, s, (const float*)ws, (float*)out, (long)nblk);
  }
  return launch_status("dpx_absmax");
}
