// Weight / bias gradients of the FFDNet layers from the C8 planes of the split-arithmetic training path (dpx_ffdnet_backward_bf16_w): the
// kernel is in dpx_wgrad_c8_dev.h; this translation unit instantiates it for the layer shapes (MT, NT <= 3: up to 96 channels either side) and
// the two arithmetic modes and finishes the per-workgroup partial sums in a fixed order (k_wgrad_c8_reduce).   Reference: the gradients
// autograd forms for network_ffdnet.py:54-68's convolutions when deep_prior(trainable=True) hands the denoiser's parameters to the
// optimiser (proxfn/pnp/prior.py:52-60, algo/primitives.py:124-205).
#include "dpx_common.h"
#include "dpx_mma_dev.h"
#include "dpx_wgrad_c8_dev.h"

namespace dpx {
unsigned* f16_overflow_flag();                                      // dpx_conv_bf16.hip

// gw[co][ci][tap], gb[co] = the sums over the NG workgroups' partial slices, in a fixed order (the same bits run to run), times *mul.  A thread owns
// one element of the kernel's tiled layout (consecutive threads read consecutive floats of every slice) and writes it where it belongs.
__global__ void __launch_bounds__(256) k_wgrad_c8_reduce(const float* __restrict__ part, const float* __restrict__ part_b, float* __restrict__ gw,
                                                         float* __restrict__ gb, int NG, int Cout, int Cin, int MT, int NT, const float* __restrict__ mul) {
  __shared__ float shb[256];
  const float ms = mul ? *mul : 1.f;
  const int CoP = MT * 32, CiP = NT * 32;
  const long ne = (long)CoP * CiP * 9;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < ne; e += (long)gridDim.x * blockDim.x) {
    const int u = (int)(e >> 10), r = (int)(e & 1023);
    const int q = r >> 8, ln = (r >> 2) & 63, i = q * 4 + (r & 3);
    const int tap = u % 9, tile = u / 9, nt = tile % NT, mt = tile / NT;
    const int co = mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * (ln >> 5), ci = nt * 32 + (ln & 31);
    // eight interleaved running sums (slice g goes to sum g % 8), joined in a fixed order: eight loads in flight
    float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float* p0 = part + e;
    int g = 0;
    for (; g + 8 <= NG; g += 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) a8[j] += p0[(size_t)(g + j) * ne];
    }
    for (int j = 0; g < NG; ++g, ++j) a8[j] += p0[(size_t)g * ne];
    if (co < Cout && ci < Cin) gw[((long)co * Cin + ci) * 9 + tap] = (((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]))) * ms;
  }
  // the bias gradients: one workgroup per output channel, a thread per slice, a fixed tree over the threads
  for (int co = blockIdx.x; co < Cout; co += gridDim.x) {
    float acc = 0.f;
    for (int g = threadIdx.x; g < NG; g += 256) acc += part_b[((size_t)g * CoP + co) * 2] + part_b[((size_t)g * CoP + co) * 2 + 1];
    __syncthreads();
    shb[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if ((int)threadIdx.x < w) shb[threadIdx.x] += shb[threadIdx.x + w];
      __syncthreads();
    }
    if (threadIdx.x == 0) gb[co] = shb[0] * ms;
  }
}

constexpr int WC_NG = 256;                                          // persistent workgroups = partial slices (one per CU)

size_t wgrad_c8_ws_floats(int cout_max, int cin_max) {
  const size_t cop = (size_t)(cout_max + 31) / 32 * 32, cip = (size_t)(cin_max + 31) / 32 * 32;
  return (size_t)WC_NG * cop * cip * 9 + (size_t)WC_NG * cop * 2;
}

template <int MT, int NT, int MODE>
static void launch_wc(const float* G, const float* A, float* part, float* part_b, int Gg, int Ga, int B, int H, int W, int nstrips, long nchunks,
                      int NG, hipStream_t s) {
  typedef WcGeom<MT, NT, MODE> GM;
  static unsigned long long attr = 0;                   // one bit per device: the opt-in beyond 64 KB is a per-device function attribute
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!(attr >> (dev & 63) & 1ull)) {
    hipFuncSetAttribute((const void*)k_wgrad_c8<MT, NT, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, GM::LDS_BYTES);
    attr |= 1ull << (dev & 63);
  }
  DPX_LAUNCH("k_wgrad_c8", (k_wgrad_c8<MT, NT, MODE>), dim3(NG), dim3(WC_NW * 64), GM::LDS_BYTES, s, G, A, part, part_b, Gg, Ga, B, H, W, nstrips,
             nchunks, f16_overflow_flag());
}
// the shapes an FFDNet stack has: first layer (MT, 1), hidden layers (MT, MT), last layer (1, NT)
template <int MODE>
static void launch_wc_mt(int mt, int nt, const float* G, const float* A, float* part, float* part_b, int Gg, int Ga, int B, int H, int W, int nstrips,
                         long nchunks, int NG, hipStream_t s) {
#define DPX_WC(M_, N_) launch_wc<M_, N_, MODE>(G, A, part, part_b, Gg, Ga, B, H, W, nstrips, nchunks, NG, s)
  switch (mt * 4 + nt) {
    case 5: DPX_WC(1, 1); break;
    case 6: DPX_WC(1, 2); break;
    case 7: DPX_WC(1, 3); break;
    case 9: DPX_WC(2, 1); break;
    case 10: DPX_WC(2, 2); break;
    case 13: DPX_WC(3, 1); break;
    case 15: DPX_WC(3, 3); break;
    default: set_error("k_wgrad_c8: no instantiation for %d x %d channel blocks", mt, nt); break;
  }
#undef DPX_WC
}

// G: C8 [B][Gg][H][W][8], A: C8 [B][Ga][H][W][8]; gw: [Cout][Cin_w][9], gb: [Cout] (Cout <= 8 Gg, Cin_w <= 8 Ga, both <= 96);
// mode 3: split-f16 (G scaled into the binary16 range by the caller), 6: split-bf16; mul (device, nullable): the sums leave multiplied by *mul
void launch_wgrad_c8(int mode, const float* G, const float* A, float* gw, float* gb, int Cout, int Cin_w, int Gg, int Ga, int B, int H, int W,
                     float* ws, const float* mul, hipStream_t s) {
  const int MT = (Cout + 31) / 32, NT = (Cin_w + 31) / 32, CoP = MT * 32, CiP = NT * 32;
  const int nstrips = (W + WC_WT - 1) / WC_WT;
  const long nchunks = (long)B * nstrips * H;
  int NG = WC_NG;
  if ((long)NG > nchunks) NG = (int)nchunks;
  float* part = ws;
  float* part_b = ws + (size_t)NG * CoP * CiP * 9;
  if (mode == 3) launch_wc_mt<3>(MT, NT, G, A, part, part_b, Gg, Ga, B, H, W, nstrips, nchunks, NG, s);
  else launch_wc_mt<6>(MT, NT, G, A, part, part_b, Gg, Ga, B, H, W, nstrips, nchunks, NG, s);
  DPX_LAUNCH("k_wgrad_c8_reduce", k_wgrad_c8_reduce, dim3(grid_for((long)CoP * CiP * 9, 256, 1024)), dim3(256), 0, s, (const float*)part,
             (const float*)part_b, gw, gb, NG, Cout, Cin_w, MT, NT, mul);
}
}  // namespace dpx

// ---- the kernel as a C entry point (the FFDNet backward pass calls launch_wgrad_c8 directly) --------------------------------------------
extern "C" size_t dpx_conv3x3_wgrad_c8_ws_bytes(int cout, int cin) { return dpx::wgrad_c8_ws_floats(cout, cin) * sizeof(float); }

extern "C" int dpx_conv3x3_wgrad_c8(const float* g, const float* a, float* gw, float* gb, int cout, int cin, int g_groups, int a_groups, int mode,
                                    const float* mul, int B, int H, int W, void* ws, dpx_stream_t stream) {
  using namespace dpx;
  DPX_REQUIRE(g && a && gw && gb && ws, "dpx_conv3x3_wgrad_c8: null pointer");
  DPX_REQUIRE(B > 0 && H > 0 && W > 0 && cout > 0 && cin > 0 && cout <= 96 && cin <= 96 && cout <= 8 * g_groups && cin <= 8 * a_groups &&
                  g_groups <= 12 && a_groups <= 12 && (mode == 3 || mode == 6),
              "dpx_conv3x3_wgrad_c8: unsupported configuration (cout=%d cin=%d groups %d / %d mode=%d)", cout, cin, g_groups, a_groups, mode);
  const int mt = (cout + 31) / 32, nt = (cin + 31) / 32;
  DPX_REQUIRE(mt == nt || mt == 1 || nt == 1, "dpx_conv3x3_wgrad_c8: %d x %d blocks of 32 channels are not instantiated (equal, or one of them 1)", mt, nt);
  DPX_REQUIRE((size_t)a_groups * H * W * 32 < ((size_t)1 << 32), "dpx_conv3x3_wgrad_c8: plane %dx%d too large", H, W);
  launch_wgrad_c8(mode, g, a, gw, gb, cout, cin, g_groups, a_groups, B, H, W, (float*)ws, mul, (hipStream_t)stream);
  return launch_status("dpx_conv3x3_wgrad_c8");
}
#ifdef DPX_WC_TRACE
extern "C" int dpx_dbg_wc_trace(unsigned long long* host, int n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(dpx::dpx_wc_trace_buf), sizeof(unsigned long long) * (n < 512 ? n : 512)) == hipSuccess ? 0 : -1;
}
#endif
