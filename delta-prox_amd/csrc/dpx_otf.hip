// PSF -> OTF / |OTF|^2 tables, evaluated as a direct fp64 DFT of the (small) kernel on the device.
//
// Restates what the reference builds on the host with NumPy: conv._FB -> psf2otf(kernel, [H, W, C])
// (dprox/linop/conv.py:23-29, dprox/utils/psf2otf.py:11-40): zero-pad 'post', roll the centre
// floor(size/2) to the origin, n-D DFT over H, W *and C*, used afterwards as a per-channel 2-D
// multiplier.  Because the result multiplies the spectrum of a real image and only the real part of
// the inverse transform is kept (conv.py:34,40), the operator only sees the Hermitian part of the OTF,
// which is what the `otf` table stores (the channel phase factor enters through its cosine); the
// `diag` table is |full OTF|^2 exactly like conv.get_diag (conv.py:46-53).
#include "dpx_common.h"

namespace dpx {

__global__ void k_psf2otf(const double* __restrict__ psf, int kh, int kw, int kc, int C, int H, int W,
                          float2* __restrict__ otf, float* __restrict__ diag, float weight, int accumulate) {
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
    if (otf) otf[idx] = make_float2((float)hr, (float)hi);
    if (diag) {
      const double d = (double)weight * (fr * fr + fi * fi);
      diag[idx] = accumulate ? (float)((double)diag[idx] + d) : (float)d;
    }
  }
}

}  // namespace dpx

using namespace dpx;

extern "C" size_t dpx_otf_bytes(int C, int H, int W) { return table_elems(C, H, W) * sizeof(float2); }
extern "C" size_t dpx_diag_bytes(int C, int H, int W) { return table_elems(C, H, W) * sizeof(float); }

extern "C" int dpx_psf2otf(const double* psf, int kh, int kw, int kc, int C, int H, int W, void* otf, void* diag,
                           float weight, int accumulate, dpx_stream_t stream) {
  DPX_REQUIRE(psf && (otf || diag), "dpx_psf2otf: null pointer");
  DPX_REQUIRE(kh > 0 && kw > 0 && kc > 0 && C > 0 && H > 0 && W > 0, "dpx_psf2otf: bad shape");
  DPX_REQUIRE(kh <= H && kw <= W && kc <= C, "dpx_psf2otf: outsize [%d,%d,%d] cannot be smaller than the PSF [%d,%d,%d]",
              H, W, C, kh, kw, kc);   // psf2otf.py:53-54
  const long total = (long)table_elems(C, H, W);
  DPX_LAUNCH("k_psf2otf", k_psf2otf, dim3(grid_for(total, 256, 4096)), dim3(256), 0, (hipStream_t)stream, psf, kh, kw, kc, C, H,
                     W, (float2*)otf, (float*)diag, weight, accumulate);
  return launch_status("dpx_psf2otf");
}
