// Elementwise / resampling layers of the U-Net denoiser behind deep_prior (reference
// dprox/proxfn/pnp/denoisers/models/unet/unet.py:34-135, wrapper.py:206-221): the convolutions run on the matrix-core kernel of
// dpx_ffdnet.hip (dpx_conv2d_leaky: bias + LeakyReLU(0.2) fused); this file holds what sits between them --
//   MaxPool2d(2)                                   unet.py:80      (floor: an odd last row / column is dropped)
//   Upsample(x2, bilinear, align_corners=True) + F.pad to the skip tensor's size + torch.cat([skip, up], dim=1)   unet.py:96-117
//     -> one kernel that writes the interpolated, zero-padded tensor straight into its channel slice of the concatenated buffer,
//        and a channel-slice copy for the skip half
// and their adjoints for the backward-data pass.  All HBM-bound streaming kernels (4 B in / 4 B out per element or less).
#include "dpx_common.h"

namespace dpx {

// y[b, c, i, j] = max of the 2x2 window; ties keep the first maximum in (dy, dx) scan order like ATen's max_pool2d
__global__ void k_maxpool2(const float* __restrict__ x, float* __restrict__ y, long n_out, int H, int W, int Ho, int Wo) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_out; i += (long)gridDim.x * blockDim.x) {
    const int xo = (int)(i % Wo);
    long r = i / Wo;
    const int yo = (int)(r % Ho);
    const long pc = r / Ho;
    const float* p = x + (pc * H + 2 * yo) * W + 2 * xo;
    y[i] = fmaxf(fmaxf(p[0], p[1]), fmaxf(p[W], p[W + 1]));
  }
}

// gx = gradient routed to the window's (first) maximum; inputs outside every window (odd last row / column) get 0
__global__ void k_maxpool2_bwd(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ gx, long n_in, int H, int W, int Ho,
                               int Wo) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_in; i += (long)gridDim.x * blockDim.x) {
    const int xx = (int)(i % W);
    long r = i / W;
    const int yy = (int)(r % H);
    const long pc = r / H;
    const int yo = yy >> 1, xo = xx >> 1;
    float g = 0.f;
    if (yo < Ho && xo < Wo) {
      const float* p = x + (pc * H + 2 * yo) * W + 2 * xo;
      const float v[4] = {p[0], p[1], p[W], p[W + 1]};
      int am = 0;
#pragma unroll
      for (int k = 1; k < 4; ++k)
        if (v[k] > v[am]) am = k;
      if (am == (yy & 1) * 2 + (xx & 1)) g = gy[(pc * Ho + yo) * Wo + xo];
    }
    gx[i] = g;
  }
}

// dst[b, c0 + c, :, :] = src[b, c, :, :]      (the skip half of torch.cat([skip, up], dim=1))
__global__ void k_copy_into_channels(const float* __restrict__ src, float* __restrict__ dst, long n, int C, long plane, int Ctot, int c0) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long px = i % plane, bc = i / plane;
    const int c = (int)(bc % C);
    const long b = bc / C;
    dst[((b * Ctot) + c0 + c) * plane + px] = src[i];
  }
}
// the reverse (gradient of the concatenation w.r.t. one of its halves)
__global__ void k_copy_from_channels(const float* __restrict__ src, float* __restrict__ dst, long n, int C, long plane, int Ctot, int c0) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long px = i % plane, bc = i / plane;
    const int c = (int)(bc % C);
    const long b = bc / C;
    dst[i] = src[((b * Ctot) + c0 + c) * plane + px];
  }
}

// source coordinate of output index o for align_corners=True: o * (n_in - 1) / (n_out - 1) in float32 like ATen
// (area_pixel_compute_scale / _source_index), index clamped, lambda = fractional part
__device__ __forceinline__ void lin_coord(int o, float scale, int n_in, int& i0, int& i1, float& l1) {
  const float s = scale * (float)o;
  i0 = (int)s;
  if (i0 > n_in - 1) i0 = n_in - 1;
  i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
  l1 = s - (float)i0;
}

// dst[b, c0 + c, py + i, px + j] = bilinear x2 of src[b, c] (align_corners=True) for (i, j) inside [0, 2h) x [0, 2w), 0 elsewhere
// (F.pad with a non-negative size difference; dst plane is Hd x Wd)
__global__ void k_upsample2_into(const float* __restrict__ src, float* __restrict__ dst, long n, int C, int h, int w, int Ctot, int c0, int Hd,
                                 int Wd, int py, int px, float sh, float sw) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int xd = (int)(i % Wd);
    long r = i / Wd;
    const int yd = (int)(r % Hd);
    const long bc = r / Hd;
    const int c = (int)(bc % C);
    const long b = bc / C;
    const int yo = yd - py, xo = xd - px;
    float v = 0.f;
    if (yo >= 0 && yo < 2 * h && xo >= 0 && xo < 2 * w) {
      int y0, y1, x0, x1;
      float ly, lx;
      lin_coord(yo, sh, h, y0, y1, ly);
      lin_coord(xo, sw, w, x0, x1, lx);
      const float* p = src + bc * (long)h * w;
      const float wy0 = 1.f - ly, wx0 = 1.f - lx;
      // ATen's separable evaluation order: rows interpolated along x first, then combined along y
      const float top = wx0 * p[(long)y0 * w + x0] + lx * p[(long)y0 * w + x1];
      const float bot = wx0 * p[(long)y1 * w + x0] + lx * p[(long)y1 * w + x1];
      v = wy0 * top + ly * bot;
    }
    dst[((b * Ctot + c0 + c) * (long)Hd + yd) * Wd + xd] = v;
  }
}

// adjoint of k_upsample2_into w.r.t. src: gsrc[b, c, y, x] = sum over the output pixels whose stencil touches (y, x).
// With scale = (n-1)/(2n-1) < 1/2 an input sample contributes to at most 5 consecutive outputs per axis: gather over that window.
__global__ void k_upsample2_into_bwd(const float* __restrict__ gdst, float* __restrict__ gsrc, long n, int C, int h, int w, int Ctot, int c0,
                                     int Hd, int Wd, int py, int px, float sh, float sw) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % w);
    long r = i / w;
    const int y = (int)(r % h);
    const long bc = r / h;
    const int c = (int)(bc % C);
    const long b = bc / C;
    const float* g = gdst + (b * Ctot + c0 + c) * (long)Hd * Wd;
    // candidate outputs: o with floor(scale * o) in {y - 1, y}
    const int oy_lo = max(0, 2 * y - 3), oy_hi = min(2 * h - 1, 2 * y + 3);
    const int ox_lo = max(0, 2 * x - 3), ox_hi = min(2 * w - 1, 2 * x + 3);
    float acc = 0.f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      int y0, y1;
      float ly;
      lin_coord(oy, sh, h, y0, y1, ly);
      float wy = 0.f;
      if (y0 == y) wy += 1.f - ly;
      if (y1 == y) wy += ly;
      if (wy == 0.f) continue;
      const int yd = oy + py;
      if (yd < 0 || yd >= Hd) continue;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        int x0, x1;
        float lx;
        lin_coord(ox, sw, w, x0, x1, lx);
        float wx = 0.f;
        if (x0 == x) wx += 1.f - lx;
        if (x1 == x) wx += lx;
        const int xd = ox + px;
        if (wx != 0.f && xd >= 0 && xd < Wd) acc += wy * wx * g[(long)yd * Wd + xd];
      }
    }
    gsrc[i] = acc;
  }
}

// gin = g * (y > 0 ? 1 : slope)      (backward of the fused LeakyReLU epilogue from the saved output: y > 0 <=> pre-activation > 0)
__global__ void k_leaky_bwd(const float* __restrict__ y, const float* __restrict__ g, float* __restrict__ gin, long n, float slope) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) gin[i] = y[i] > 0.f ? g[i] : slope * g[i];
}

}  // namespace dpx

using namespace dpx;

extern "C" int dpx_maxpool2(const float* x, float* y, int B, int C, int H, int W, dpx_stream_t stream) {
  DPX_REQUIRE(x && y && B > 0 && C > 0 && H >= 2 && W >= 2, "dpx_maxpool2: bad arguments (plane %dx%d)", H, W);
  const int Ho = H / 2, Wo = W / 2;
  const long n = (long)B * C * Ho * Wo;
  DPX_LAUNCH("k_maxpool2", k_maxpool2, dim3(grid_for(n, 256, 8192)), dim3(256), 0, (hipStream_t)stream, x, y, n, H, W, Ho, Wo);
  return launch_status("dpx_maxpool2");
}

extern "C" int dpx_maxpool2_bwd(const float* x, const float* gy, float* gx, int B, int C, int H, int W, dpx_stream_t stream) {
  DPX_REQUIRE(x && gy && gx && B > 0 && C > 0 && H >= 2 && W >= 2, "dpx_maxpool2_bwd: bad arguments");
  const long n = (long)B * C * H * W;
  DPX_LAUNCH("k_maxpool2_bwd", k_maxpool2_bwd, dim3(grid_for(n, 256, 8192)), dim3(256), 0, (hipStream_t)stream, x, gy, gx, n, H, W, H / 2, W / 2);
  return launch_status("dpx_maxpool2_bwd");
}

extern "C" int dpx_copy_channels(const float* src, float* dst, int to_slice, int B, int C, int H, int W, int Ctot, int c0, dpx_stream_t stream) {
  DPX_REQUIRE(src && dst && B > 0 && C > 0 && H > 0 && W > 0 && c0 >= 0 && c0 + C <= Ctot, "dpx_copy_channels: bad channel slice [%d, %d) of %d", c0,
              c0 + C, Ctot);
  const long n = (long)B * C * H * W;
  if (to_slice)
    DPX_LAUNCH("k_copy_into_channels", k_copy_into_channels, dim3(grid_for(n, 256, 8192)), dim3(256), 0, (hipStream_t)stream, src, dst, n, C,
               (long)H * W, Ctot, c0);
  else
    DPX_LAUNCH("k_copy_from_channels", k_copy_from_channels, dim3(grid_for(n, 256, 8192)), dim3(256), 0, (hipStream_t)stream, src, dst, n, C,
               (long)H * W, Ctot, c0);
  return launch_status("dpx_copy_channels");
}

static float align_corners_scale(int n_in, int n_out) { return n_out > 1 ? (float)(n_in - 1) / (float)(n_out - 1) : 0.f; }

extern "C" int dpx_upsample2_into(const float* src, float* dst, int B, int C, int h, int w, int Ctot, int c0, int Hd, int Wd, dpx_stream_t stream) {
  DPX_REQUIRE(src && dst && B > 0 && C > 0 && h > 0 && w > 0 && c0 >= 0 && c0 + C <= Ctot, "dpx_upsample2_into: bad arguments");
  DPX_REQUIRE(Hd >= 2 * h && Wd >= 2 * w, "dpx_upsample2_into: target plane %dx%d smaller than the upsampled %dx%d", Hd, Wd, 2 * h, 2 * w);
  const long n = (long)B * C * Hd * Wd;
  DPX_LAUNCH("k_upsample2_into", k_upsample2_into, dim3(grid_for(n, 256, 8192)), dim3(256), 0, (hipStream_t)stream, src, dst, n, C, h, w, Ctot, c0,
             Hd, Wd, (Hd - 2 * h) / 2, (Wd - 2 * w) / 2, align_corners_scale(h, 2 * h), align_corners_scale(w, 2 * w));
  return launch_status("dpx_upsample2_into");
}

extern "C" int dpx_upsample2_into_bwd(const float* gdst, float* gsrc, int B, int C, int h, int w, int Ctot, int c0, int Hd, int Wd,
                                      dpx_stream_t stream) {
  DPX_REQUIRE(gdst && gsrc && B > 0 && C > 0 && h > 0 && w > 0 && c0 >= 0 && c0 + C <= Ctot && Hd >= 2 * h && Wd >= 2 * w,
              "dpx_upsample2_into_bwd: bad arguments");
  const long n = (long)B * C * h * w;
  DPX_LAUNCH("k_upsample2_into_bwd", k_upsample2_into_bwd, dim3(grid_for(n, 256, 8192)), dim3(256), 0, (hipStream_t)stream, gdst, gsrc, n, C, h, w,
             Ctot, c0, Hd, Wd, (Hd - 2 * h) / 2, (Wd - 2 * w) / 2, align_corners_scale(h, 2 * h), align_corners_scale(w, 2 * w));
  return launch_status("dpx_upsample2_into_bwd");
}

extern "C" int dpx_leaky_relu_bwd(const float* y, const float* g, float* gin, long n, float neg_slope, dpx_stream_t stream) {
  DPX_REQUIRE(y && g && gin && n > 0, "dpx_leaky_relu_bwd: bad arguments");
  DPX_LAUNCH("k_leaky_bwd", k_leaky_bwd, dim3(grid_for(n, 256, 8192)), dim3(256), 0, (hipStream_t)stream, y, g, gin, n, neg_slope);
  return launch_status("dpx_leaky_relu_bwd");
}
