// PSF -> OTF / |OTF|^2 tables, evaluated as a direct fp64 DFT of the (small) kernel on the device.
//
// Restates what the reference builds on the host with NumPy: conv._FB -> psf2otf(kernel, [H, W, C])
// (dprox/linop/conv.py:23-29, dprox/utils/psf2otf.py:11-40): zero-pad 'post', roll the centre
// floor(size/2) to the origin, n-D DFT over H, W *and C*, used afterwards as a per-channel 2-D
// multiplier.  Because the result multiplies the spectrum of a real image and only the real part of
// the inverse transform is kept (conv.py:34,40), the operator only sees the Hermitian part of the OTF,
// which is what the `otf` table stores (the channel phase factor enters through its cosine); the
// `diag` table is |full OTF|^2 exactly like conv.get_diag (conv.py:46-53).
#include <cstdlib>

#include "dpx_common.h"

namespace dpx {

__global__ void k_psf2otf(const double* __restrict__ psf, int kh, int kw, int kc, int C, int H, int W,
                          float2* __restrict__ otf, float* __restrict__ diag, float weight, int accumulate, int tiled) {
  const int Ws = (W + 1) / 2;
  const bool even = (W % 2 == 0);
  const long nmain = (long)C * H * Ws, nside = (long)C * H;
  const long total = nmain + (even ? nside : 0);
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    int c, k, l;
    if (idx < nmain) {
      c = (int)(idx / ((long)H * Ws));
      const long r = idx - (long)c * H * Ws;
      k = (int)(r / Ws);
      l = (int)(r - (long)k * Ws);
    } else {
      const long r = idx - nmain;
      c = (int)(r / H);
      k = (int)(r - (long)c * H);
      l = W / 2;
    }
    const int ci = kh / 2, cj = kw / 2, cm = kc / 2;
    double fr = 0.0, fi = 0.0;     // full OTF
    double hr = 0.0, hi = 0.0;     // Hermitian part (what a real-in / real-out operator applies)
    for (int m = 0; m < kc; ++m) {
      double sr = 0.0, si = 0.0;   // S_m[k,l] = sum_ij psf[i,j,m] e^{-2 pi i (k (i-ci)/H + l (j-cj)/W)}
      for (int i = 0; i < kh; ++i) {
        long long ph = ((long long)k * (i - ci)) % H;
        if (ph < 0) ph += H;
        double sh, chh;
        sincospi(-2.0 * (double)ph / (double)H, &sh, &chh);
        double rr = 0.0, ri = 0.0;
        for (int j = 0; j < kw; ++j) {
          long long pw = ((long long)l * (j - cj)) % W;
          if (pw < 0) pw += W;
          double sw, cw;
          sincospi(-2.0 * (double)pw / (double)W, &sw, &cw);
          const double a = psf[((long)i * kw + j) * kc + m];
          rr += a * cw;
          ri += a * sw;
        }
        sr += rr * chh - ri * sh;
        si += rr * sh + ri * chh;
      }
      long long pc = ((long long)c * (m - cm)) % C;
      if (pc < 0) pc += C;
      double sc, cc;
      sincospi(-2.0 * (double)pc / (double)C, &sc, &cc);
      fr += sr * cc - si * sc;
      fi += sr * sc + si * cc;
      hr += sr * cc;
      hi += si * cc;
    }
    const size_t pos = idx < nmain ? (size_t)c * H * Ws + spec_main_index(tiled, H, Ws, k, l) : (size_t)idx;
    if (otf) otf[pos] = make_float2((float)hr, (float)hi);
    if (diag) {
      const double d = (double)weight * (fr * fr + fi * fi);
      diag[pos] = accumulate ? (float)((double)diag[pos] + d) : (float)d;
    }
  }
}

// The same table, evaluated separably: one workgroup = one tile of PT_K frequency rows x PT_L frequency columns of one channel.
//   R[m][i][l] = sum_j psf[i,j,m] e^{-2 pi i l (j-cj)/W}   (kc kh PT_L values, kw sincospi each)      -- the column factor, once per tile
//   (the column phases e^{-2 pi i l (j-cj)/W} themselves: PT_L kw sincospi per tile, shared by the kh kernel rows)
//   E[k][i]    = e^{-2 pi i k (i-ci)/H}                    (PT_K kh values)                           -- the row factor, once per tile
//   S_m[k,l]   = sum_i R[m][i][l] E[k][i]                  (kh complex fp64 products per entry instead of kh (1 + kw) sincospi)
// Same operations in the same order as k_psf2otf on every entry (bit-identical tables); 15 x 15 PSF at 3 x 1024^2: 0.84 ms -> ~0.03 ms.
constexpr int PT_K = 64, PT_L = 16;
__global__ void __launch_bounds__(256) k_psf2otf_tiled(const double* __restrict__ psf, int kh, int kw, int kc, int C, int H, int W,
                                                        float2* __restrict__ otf, float* __restrict__ diag, float weight, int accumulate, int tiled) {
  HIP_DYNAMIC_SHARED(double2, smem_otf)
  const int Ws = (W + 1) / 2;
  const bool even = (W % 2 == 0);
  const int Wl = even ? W / 2 + 1 : Ws;                 // frequency columns 0 .. Wl-1 (the last one is the side part for even W)
  const int ltiles = (Wl + PT_L - 1) / PT_L, ktiles = (H + PT_K - 1) / PT_K;
  int bid = blockIdx.x;
  const int lt = bid % ltiles; bid /= ltiles;
  const int kt = bid % ktiles;
  const int c = bid / ktiles;
  const int k0 = kt * PT_K, l0 = lt * PT_L;
  double2* Rt = smem_otf;                               // [kc][kh][PT_L]
  double2* Et = smem_otf + (size_t)kc * kh * PT_L;      // [PT_K][kh]
  const int ci = kh / 2, cj = kw / 2, cm = kc / 2;
  double2* Ft = Et + (size_t)PT_K * kh;                 // [PT_L][kw]: the column phases, shared by the kh kernel rows
  for (int e = threadIdx.x; e < PT_L * kw; e += blockDim.x) {
    const int j = e % kw, l = l0 + e / kw;
    long long pw = ((long long)l * (j - cj)) % W;
    if (pw < 0) pw += W;
    double sw, cw;
    sincospi(-2.0 * (double)pw / (double)W, &sw, &cw);
    Ft[e] = make_double2(cw, sw);
  }
  __syncthreads();
  for (int e = threadIdx.x; e < kc * kh * PT_L; e += blockDim.x) {
    const int ll = e % PT_L, i = (e / PT_L) % kh, m = e / (PT_L * kh);
    double rr = 0.0, ri = 0.0;
    for (int j = 0; j < kw; ++j) {
      const double2 f = Ft[ll * kw + j];
      const double a = psf[((long)i * kw + j) * kc + m];
      rr += a * f.x;
      ri += a * f.y;
    }
    Rt[e] = make_double2(rr, ri);
  }
  for (int e = threadIdx.x; e < PT_K * kh; e += blockDim.x) {
    const int i = e % kh, k = k0 + e / kh;
    long long ph = ((long long)k * (i - ci)) % H;
    if (ph < 0) ph += H;
    double sh, chh;
    sincospi(-2.0 * (double)ph / (double)H, &sh, &chh);
    Et[e] = make_double2(chh, sh);
  }
  __syncthreads();
  for (int e = threadIdx.x; e < PT_K * PT_L; e += blockDim.x) {
    const int ll = e % PT_L, kk = e / PT_L;
    const int k = k0 + kk, l = l0 + ll;
    if (k >= H || l >= Wl) continue;
    double fr = 0.0, fi = 0.0, hr = 0.0, hi = 0.0;
    for (int m = 0; m < kc; ++m) {
      double sr = 0.0, si = 0.0;
      for (int i = 0; i < kh; ++i) {
        const double2 r = Rt[((size_t)m * kh + i) * PT_L + ll], w = Et[kk * kh + i];
        sr += r.x * w.x - r.y * w.y;
        si += r.x * w.y + r.y * w.x;
      }
      long long pc = ((long long)c * (m - cm)) % C;
      if (pc < 0) pc += C;
      double sc, cc;
      sincospi(-2.0 * (double)pc / (double)C, &sc, &cc);
      fr += sr * cc - si * sc;
      fi += sr * sc + si * cc;
      hr += sr * cc;
      hi += si * cc;
    }
    const size_t pos = l < Ws ? (size_t)c * H * Ws + spec_main_index(tiled, H, Ws, k, l) : (size_t)C * H * Ws + (size_t)c * H + k;
    if (otf) otf[pos] = make_float2((float)hr, (float)hi);
    if (diag) {
      const double d = (double)weight * (fr * fr + fi * fi);
      diag[pos] = accumulate ? (float)((double)diag[pos] + d) : (float)d;
    }
  }
}

// real table <-> full [C][H][W] array (Hermitian-symmetric extension: T(-k,-l) = T(k,l))
__global__ void k_table_to_full(const float* __restrict__ tab, float* __restrict__ full, int C, int H, int W, int tiled) {
  const int Ws = (W + 1) / 2;
  const long total = (long)C * H * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int l0 = (int)(i % W);
    const long r = i / W;
    const int k0 = (int)(r % H), c = (int)(r / H);
    int k = k0, l = l0;
    if (2 * l0 > W) { l = W - l0; k = (H - k0) % H; }
    float v;
    if (W % 2 == 0 && l == W / 2) v = tab[(size_t)C * H * Ws + (size_t)c * H + k];
    else v = tab[(size_t)c * H * Ws + spec_main_index(tiled, H, Ws, k, l)];
    full[i] = v;
  }
}
__global__ void k_table_from_full(const float* __restrict__ full, float* __restrict__ tab, int C, int H, int W, int tiled) {
  const int Ws = (W + 1) / 2;
  const long nmain = (long)C * H * Ws, total = nmain + ((W % 2 == 0) ? (long)C * H : 0);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    if (i < nmain) {
      const int l = (int)(i % Ws);
      const long r = i / Ws;
      const int k = (int)(r % H), c = (int)(r / H);
      tab[(size_t)c * H * Ws + spec_main_index(tiled, H, Ws, k, l)] = full[((size_t)c * H + k) * W + l];
    } else {
      const long r = i - nmain;
      const int k = (int)(r % H), c = (int)(r / H);
      tab[i] = full[((size_t)c * H + k) * W + W / 2];
    }
  }
}
// complex OTF given on the full grid [C][H][W] (conv_doe: transform of a padded, centred PSF) -> half-spectrum table
__global__ void k_otf_from_full(const float2* __restrict__ full, float2* __restrict__ tab, int C, int H, int W, int tiled) {
  const int Ws = (W + 1) / 2;
  const long nmain = (long)C * H * Ws, total = nmain + ((W % 2 == 0) ? (long)C * H : 0);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    if (i < nmain) {
      const int l = (int)(i % Ws);
      const long r = i / Ws;
      const int k = (int)(r % H), c = (int)(r / H);
      tab[(size_t)c * H * Ws + spec_main_index(tiled, H, Ws, k, l)] = full[((size_t)c * H + k) * W + l];
    } else {
      const long r = i - nmain;
      const int k = (int)(r % H), c = (int)(r / H);
      tab[i] = full[((size_t)c * H + k) * W + W / 2];
    }
  }
}
// dd = (d0 + c0, d1 + c1) interleaved; element order is irrelevant (same opaque layout in and out)
__global__ void k_denominator_pack(const float* __restrict__ d0, float c0, const float* __restrict__ d1, float c1,
                                   float2* __restrict__ dd, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    dd[i] = make_float2((d0 ? d0[i] : 0.f) + c0, (d1 ? d1[i] : 0.f) + c1);
}

}  // namespace dpx

using namespace dpx;

extern "C" int dpx_table_to_full(const void* table, float* full, int C, int H, int W, dpx_stream_t stream) {
  DPX_REQUIRE(table && full && C > 0 && H > 0 && W > 0, "dpx_table_to_full: bad arguments");
  DPX_LAUNCH("k_table_to_full", k_table_to_full, dim3(grid_for((long)C * H * W, 256, 4096)), dim3(256), 0, (hipStream_t)stream,
             (const float*)table, full, C, H, W, pow2_path_available(H, W) ? 1 : 0);
  return launch_status("dpx_table_to_full");
}
extern "C" int dpx_table_from_full(const float* full, void* table, int C, int H, int W, dpx_stream_t stream) {
  DPX_REQUIRE(table && full && C > 0 && H > 0 && W > 0, "dpx_table_from_full: bad arguments");
  DPX_LAUNCH("k_table_from_full", k_table_from_full, dim3(grid_for((long)table_elems(C, H, W), 256, 4096)), dim3(256), 0,
             (hipStream_t)stream, full, (float*)table, C, H, W, pow2_path_available(H, W) ? 1 : 0);
  return launch_status("dpx_table_from_full");
}
extern "C" int dpx_otf_from_full(const void* full, void* otf, int C, int H, int W, dpx_stream_t stream) {
  DPX_REQUIRE(otf && full && C > 0 && H > 0 && W > 0, "dpx_otf_from_full: bad arguments");
  DPX_LAUNCH("k_otf_from_full", k_otf_from_full, dim3(grid_for((long)table_elems(C, H, W), 256, 4096)), dim3(256), 0, (hipStream_t)stream,
             (const float2*)full, (float2*)otf, C, H, W, pow2_path_available(H, W) ? 1 : 0);
  return launch_status("dpx_otf_from_full");
}
extern "C" size_t dpx_denominator_bytes(int C, int H, int W) { return table_elems(C, H, W) * sizeof(float2); }
extern "C" int dpx_denominator_pack(const void* d0, float c0, const void* d1, float c1, void* dd, int C, int H, int W,
                                    dpx_stream_t stream) {
  DPX_REQUIRE(dd && C > 0 && H > 0 && W > 0, "dpx_denominator_pack: bad arguments");
  const long n = (long)table_elems(C, H, W);
  DPX_LAUNCH("k_denominator_pack", k_denominator_pack, dim3(grid_for(n, 256, 4096)), dim3(256), 0, (hipStream_t)stream,
             (const float*)d0, c0, (const float*)d1, c1, (float2*)dd, n);
  return launch_status("dpx_denominator_pack");
}

extern "C" size_t dpx_otf_bytes(int C, int H, int W) { return table_elems(C, H, W) * sizeof(float2); }
extern "C" size_t dpx_diag_bytes(int C, int H, int W) { return table_elems(C, H, W) * sizeof(float); }

extern "C" int dpx_psf2otf(const double* psf, int kh, int kw, int kc, int C, int H, int W, void* otf, void* diag,
                           float weight, int accumulate, dpx_stream_t stream) {
  DPX_REQUIRE(psf && (otf || diag), "dpx_psf2otf: null pointer");
  DPX_REQUIRE(kh > 0 && kw > 0 && kc > 0 && C > 0 && H > 0 && W > 0, "dpx_psf2otf: bad shape");
  DPX_REQUIRE(kh <= H && kw <= W && kc <= C, "dpx_psf2otf: outsize [%d,%d,%d] cannot be smaller than the PSF [%d,%d,%d]",
              H, W, C, kh, kw, kc);   // psf2otf.py:53-54
  const long total = (long)table_elems(C, H, W);
  const size_t sh = ((size_t)kc * kh * PT_L + (size_t)PT_K * kh + (size_t)PT_L * kw) * sizeof(double2);
  const bool direct = false;                               // (the entry-by-entry kernel serves kernels whose tile does not fit 64 KB of LDS)
  if (sh <= 64 * 1024 && !direct) {
    const int Wl = (W % 2 == 0) ? W / 2 + 1 : (W + 1) / 2;
    const long blocks = (long)C * ((H + PT_K - 1) / PT_K) * ((Wl + PT_L - 1) / PT_L);
    static bool attr = false;
    if (!attr) {
      hipFuncSetAttribute((const void*)k_psf2otf_tiled, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
      attr = true;
    }
    DPX_LAUNCH("k_psf2otf_tiled", k_psf2otf_tiled, dim3((unsigned)blocks), dim3(256), sh, (hipStream_t)stream, psf, kh, kw, kc, C, H, W,
               (float2*)otf, (float*)diag, weight, accumulate, pow2_path_available(H, W) ? 1 : 0);
    return launch_status("dpx_psf2otf");
  }
  DPX_LAUNCH("k_psf2otf", k_psf2otf, dim3(grid_for(total, 256, 4096)), dim3(256), 0, (hipStream_t)stream, psf, kh, kw, kc, C, H,
                     W, (float2*)otf, (float*)diag, weight, accumulate, pow2_path_available(H, W) ? 1 : 0);
  return launch_status("dpx_psf2otf");
}
