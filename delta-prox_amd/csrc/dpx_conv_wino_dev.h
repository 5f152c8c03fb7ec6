// 3x3 convolution layers as Winograd F(2x2, 3x3) on the f16 matrix cores at fp32 accuracy (arithmetic mode 4, "split-f16 Winograd"):
//   Y = A^T [ (G g G^T) . (B^T d B) ] A   per 2 x 2 output tile and (cin, cout) pair  --  16 products instead of 36, i.e. 2.25 x fewer
//   matrix instructions than k_conv3x3_bf16<.., 3> for the layers of network_ffdnet.py:54-68 / basicblock.py:61-98.
// Included by dpx_conv_bf16.hip (shares its split helpers and the range-trap flag).
//
// Per Winograd position p = (xi, nu) the layer is a GEMM  M_p[cout][tile] = sum_cin U_p[cout][cin] V_p[cin][tile]  on
// v_mfma_f32_32x32x16_f16 with BOTH operands split into two binary16 terms:
//   V = vh + vl / 2^11  (split2_f16_pair, as the direct kernel: 22 significant bits at every magnitude below the binary16 range),
//   U' = 2^s U = uh + ul (s: a per-layer power of two that puts max |U'| in [2^13, 2^14); ul = half(U' - uh) UNSCALED -- small ones are
//   binary16 subnormals, which the matrix instruction honours (tools/probe_wino_prereq.hip): absolute error <= 2^-25, 2^-38 of the layer's
//   largest weight), and a third plane uh2 = 2^-11 uh derived in registers (v_pk_mul_f16: an exponent shift).  Three products
//       ul vh  |  uh2 vl'  |  uh vh            (vl' = 2^11 vl;  dropped: ul vl <= 2^-22 |U V|)
//   land on ONE accumulator at the scale 2^s -- the direct kernel's second accumulator (its cross terms live at 2^11) would not fit: the
//   accumulators of a workgroup are 16 positions x cout x tiles.
// U = G g G^T is computed in float64 and split from there (k_wn_pack_weights); the input transform B^T d B is additions in fp32
// (two roundings), the output transform A^T M A additions in fp32.
//
// Workgroup = 8 waves, output tile 8 rows x 32 columns = 64 Winograd tiles (two matrix-instruction column blocks `nb`) x all output channels.
// Wave w owns the positions (xi = w >> 1, nu in {2 (w & 1), 2 (w & 1) + 1}) for both column blocks: 2 x MT x 2 accumulators (192 registers at
// MT = 3), nobody else needs its weights -- they stream by LDS-DMA into a wave-private ring of two position slots (no workgroup barrier) --
// and nobody else needs its B operands -- every lane builds them in registers from the landing buffer (fp32 input pixels of the current 16-channel
// chunk, double-buffered, ONE workgroup barrier per chunk): two rows x two columns of its tile's 4 x 4 patch per position, conflict-free
// ds_read_b128 by construction of the landing layout (the DMA lands every 16-byte piece where its reader wants it).
// Epilogue: the 16 -> 4 output transform crosses waves: per 32-channel block every wave leaves its two nu-partial sums in LDS (131 KB), then
// thread (tile, channel quad) adds the eight pieces up in a fixed order, scales by 2^-s, adds the bias, applies the ReLU and stores 2 x 2 pixels x 4 channels.
#pragma once
#include <type_traits>

namespace dpx {

constexpr int WN_TH = 8, WN_TW = 32, WN_ROWS = WN_TH + 2;
constexpr int WN_LINE = 640;                                          // one (k-group, half, row) line: [column parity 2][17 + 3 pad] pieces of 16 bytes
constexpr int WN_LAND_BYTES = 2 * 2 * WN_ROWS * WN_LINE;              // 25600 = 25 DMA instructions of 1 KB
constexpr int WN_LAND_INSTR = WN_LAND_BYTES / 1024;
__host__ __device__ constexpr int wn_pos_bytes(int MT) { return MT * 2048; }              // [mt][plane 2][k-group 2][cout 32][8] binary16
__host__ __device__ constexpr size_t wn_smem_bytes(int MT) {
  const size_t main_loop = 2 * WN_LAND_BYTES + 8 * 2 * wn_pos_bytes(MT), epilogue = 8 * 16 * 1024;
  return main_loop > epilogue ? main_loop : epilogue;
}
// packed layer: [chunk][position 16][mt][plane][k-group][cout 32][8] binary16, bias fp32 [MT * 32], 2^-s + 15 unused floats, 64 zero bytes
static inline size_t wn_layer_bytes(int cin, int cout) {
  const int chunks = (cin + 15) / 16, MT = (cout + 31) / 32;
  return (size_t)chunks * 16 * wn_pos_bytes(MT) + (size_t)MT * 32 * 4 + 64 + 64;
}

// B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]: row r = d[first] + sign d[second]
__host__ __device__ constexpr int wn_first(int r) { return r == 0 ? 0 : (r == 2 ? 2 : 1); }
__host__ __device__ constexpr int wn_second(int r) { return r == 0 ? 2 : (r == 1 ? 2 : (r == 2 ? 1 : 3)); }
__host__ __device__ constexpr float wn_sign(int r) { return r == 1 ? 1.f : -1.f; }

// largest |U| of a layer (U = G g G^T, G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]) as the bits of a non-negative float
__device__ __forceinline__ double wn_u(const float* g9, int xi, int nu) {
  double r[3];                                                         // (G g)[xi][b]
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    const double g0 = g9[b], g1 = g9[3 + b], g2 = g9[6 + b];
    r[b] = xi == 0 ? g0 : (xi == 3 ? g2 : (xi == 1 ? 0.5 * (g0 + g1 + g2) : 0.5 * (g0 - g1 + g2)));
  }
  return nu == 0 ? r[0] : (nu == 3 ? r[2] : (nu == 1 ? 0.5 * (r[0] + r[1] + r[2]) : 0.5 * (r[0] - r[1] + r[2])));
}
__global__ void k_wn_umax(const float* __restrict__ w, int n_filters, unsigned* __restrict__ out) {
  float m = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)n_filters * 16; i += (long)gridDim.x * blockDim.x) {
    const int p = (int)(i % 16);
    m = fmaxf(m, fabsf((float)wn_u(w + (i / 16) * 9, p >> 2, p & 3)));
  }
  atomicMax(out, __float_as_uint(m));
}
__device__ __forceinline__ unsigned short wn_f16_bits_d(double x) {
  const _Float16 h = (_Float16)x;
  unsigned short b;
  __builtin_memcpy(&b, &h, 2);
  return b;
}
__device__ __forceinline__ double wn_f16_value(unsigned short b) {
  _Float16 h;
  __builtin_memcpy(&h, &b, 2);
  return (double)h;
}
// w [cout][cin][9] fp32, b [cout] -> packed Winograd layer; umax: k_wn_umax's result for this layer
__global__ void k_wn_pack_weights(const float* __restrict__ w, const float* __restrict__ b, unsigned short* __restrict__ dst, int cin, int cout,
                                  const unsigned* __restrict__ umax) {
  const int chunks = (cin + 15) / 16, MT = (cout + 31) / 32, M32 = MT * 32;
  const long nw = (long)chunks * 16 * MT * 1024;                        // 16-bit elements
  float* tail = (float*)(dst + nw);
  // 2^s with max |U| 2^s in [2^13, 2^14)
  const float um = __uint_as_float(*umax);
  int e = 0;
  if (um > 0.f) frexpf(um, &e);                                        // um = f 2^e, f in [0.5, 1)  ->  um in [2^(e-1), 2^e)
  const int s = um > 0.f ? 14 - e : 0;
  const double scale = ldexp(1.0, s);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nw + M32 + 32; i += (long)gridDim.x * blockDim.x) {
    if (i >= nw) {
      const int k = (int)(i - nw);
      tail[k] = k < M32 ? ((k < cout && b) ? b[k] : 0.f) : (k == M32 ? (float)ldexp(1.0, -s) : 0.f);
      continue;
    }
    const int j = (int)(i % 8);
    long r = i / 8;
    const int co32 = (int)(r % 32);
    r /= 32;
    const int kg = (int)(r % 2);
    r /= 2;
    const int plane = (int)(r % 2);
    r /= 2;
    const int mt = (int)(r % MT);
    r /= MT;
    const int p = (int)(r % 16), chunk = (int)(r / 16);
    const int ci = chunk * 16 + kg * 8 + j, co = mt * 32 + co32;
    unsigned short v = 0;
    if (co < cout && ci < cin) {
      const double u = wn_u(w + ((long)co * cin + ci) * 9, p >> 2, p & 3) * scale;
      const unsigned short h = wn_f16_bits_d(u);
      v = plane == 0 ? h : wn_f16_bits_d(u - wn_f16_value(h));
    }
    dst[i] = v;
  }
}

__device__ __forceinline__ void wn_wait_vm(int n) {                    // s_waitcnt vmcnt(n) for a wave-uniform run-time n
#ifdef DPX_EMULATED
  dpx_wait_vm<0>();
#else
  switch (n) {
#define DPX_WN_CASE(k) case k: dpx_wait_vm<k>(); break;
    DPX_WN_CASE(0) DPX_WN_CASE(1) DPX_WN_CASE(2) DPX_WN_CASE(3) DPX_WN_CASE(4) DPX_WN_CASE(5) DPX_WN_CASE(6) DPX_WN_CASE(7) DPX_WN_CASE(8)
    DPX_WN_CASE(9) DPX_WN_CASE(10) DPX_WN_CASE(11) DPX_WN_CASE(12) DPX_WN_CASE(13) DPX_WN_CASE(14) DPX_WN_CASE(15) DPX_WN_CASE(16)
    DPX_WN_CASE(17) DPX_WN_CASE(18) DPX_WN_CASE(19) DPX_WN_CASE(20)
#undef DPX_WN_CASE
    default: dpx_wait_vm<0>(); break;
  }
#endif
}

typedef _Float16 wn_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ uint4 wn_shift11(uint4 a) {                 // every binary16 element times 2^-11 (exact above the subnormal range)
  wn_f16x8 v = __builtin_bit_cast(wn_f16x8, a);
  v = v * (_Float16)0.00048828125f;
  return __builtin_bit_cast(uint4, v);
}

// Tuning aid (tools/build_variant.sh wn_trace -DDPX_WN_TRACE; never in the shipped library): the waves of workgroup 0 stamp the shader clock
// along their SECOND tile; tools/wino_trace.py prints the timeline of the last launch.
#ifdef DPX_WN_TRACE
__device__ unsigned long long dpx_wn_trace_buf[8 * 64];
#define DPX_WN_STAMP(i)                                                                                                       \
  do {                                                                                                                       \
    if (MT > 1 && lane == 0 && blockIdx.x == 0 && tile == (int)gridDim.x && (i) < 64) dpx_wn_trace_buf[wv * 64 + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define DPX_WN_STAMP(i) ((void)0)
#endif

template <int MT, bool RELU>
__global__ void __launch_bounds__(512, 1) k_conv3x3_wino(const float* __restrict__ in, float* __restrict__ out, const char* __restrict__ wpk, int Gin,
                                                         int Gout, int H, int W, int tiles_x, int tiles_img, int tiles_all) {
  constexpr int POSB = wn_pos_bytes(MT), NWP = 2 * MT;                 // bytes / DMA instructions of one position's weights
  HIP_DYNAMIC_SHARED(char, smem_wn)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xi = wv >> 1, nup = wv & 1;
  const int n = lane & 31, kg = lane >> 5;
  const int chunks = Gin / 2;
  char* wring = smem_wn + 2 * WN_LAND_BYTES + wv * 2 * POSB;           // this wave's two position slots
  const float* bias = (const float*)(wpk + (size_t)chunks * 16 * POSB);
  // persistent workgroups (one per CU: the launch of a 149 KB / 8-wave workgroup costs microseconds, a tile ~10): tile t of the launch =
  // (image, tile row, tile column), dealt round-robin
  for (int tile = blockIdx.x; tile < tiles_all; tile += gridDim.x) {
  const int b = tile / tiles_img, trem = tile - b * tiles_img;
  const int ty0 = trem / tiles_x, tx0 = trem - ty0 * tiles_x;
  const int y0 = ty0 * WN_TH, x0 = tx0 * WN_TW;
  const float* inb = in + (size_t)b * Gin * H * W * 8;
  if (tile != (int)blockIdx.x) DPX_LDS_BARRIER();                      // the previous tile's epilogue has read its partial sums: the LDS is free again
  DPX_WN_STAMP(0);

  // ---- landing pieces of this lane: DMA instruction k * 8 + wv, piece (..) * 64 + lane -> (k-group, half, row, column) of the input tile.
  // Pieces outside the image (zero padding) and the layout's pad slots fetch the chunk's first 16 bytes and are zeroed by their lane once landed.
  int lane_t = lane;                                                   // (opaque per tile: the geometry below is recomputed for every tile instead
  DPX_OPAQUE(lane_t);                                                  //  of living in ~20 registers across the tile loop)
  constexpr int NLAND = 4;                                             // 25 instructions over 8 waves: wave 0 has four, the others repeat their first one
  unsigned poff[4];
  unsigned pzero = 0u;                                                 // bit k: piece k of this lane is a zero
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int li = k * 8 + wv < WN_LAND_INSTR ? k * 8 + wv : wv;       // this wave's k-th landing instruction
    const int q = li * 64 + lane_t;
    const int line = q / 40, slot = q - line * 40;
    const int par = slot / 20, idx = slot - par * 20;
    const int g = line / (2 * WN_ROWS), rem = line - g * 2 * WN_ROWS, half = rem / WN_ROWS, row = rem - half * WN_ROWS;
    const int yy = y0 + row - 1, xx = x0 + 2 * idx + par - 1;
    const bool ok = idx < 17 && yy >= 0 && yy < H && xx >= 0 && xx < W;
    poff[k] = ok ? (unsigned)((((size_t)g * H * W + (size_t)yy * W + xx) * 8 + half * 4) * sizeof(float)) : 0u;
    if (!ok) pzero |= 1u << k;
  }
  const bool any_zero = __any(pzero != 0u);                            // (wave-uniform: interior tiles skip the zero fill)
  auto issue_land = [&](int c) {
    const float* cb = inb + (size_t)(2 * c) * H * W * 8;
    char* dst = smem_wn + (c & 1) * WN_LAND_BYTES;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      dpx_glds16_s(cb, poff[k], dst + (k * 8 + wv < WN_LAND_INSTR ? k * 8 + wv : wv) * 1024);
  };
  auto zero_land = [&](int c) {                                        // behind the wait for land(c), in front of the barrier
    if (!any_zero) return;
    char* dst = smem_wn + (c & 1) * WN_LAND_BYTES + lane * 16;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (pzero & (1u << k)) *(float4*)(dst + (k * 8 + wv < WN_LAND_INSTR ? k * 8 + wv : wv) * 1024) = make_float4(0.f, 0.f, 0.f, 0.f);
  };
  auto issue_w = [&](int c, int which) {                               // position (xi, 2 nup + which) of chunk c -> slot `which`
    const char* src = wpk + ((size_t)c * 16 + xi * 4 + 2 * nup + which) * POSB;
    char* dst = wring + which * POSB;
#pragma unroll
    for (int i = 0; i < NWP; ++i) dpx_glds16_s(src + i * 1024, lane * 16, dst + i * 1024);
  };

  float f16_max = 0.f;
  f32x16 acc[2][MT][2];
#pragma unroll
  for (int v = 0; v < 2; ++v)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[v][mt][nb][i] = 0.f;
  // this lane's tile inside a column block: (n >> 4, n & 15); byte offset of its patch's first pixel in a landing buffer
  const int lbase = kg * (2 * WN_ROWS * WN_LINE) + (n >> 4) * (2 * WN_LINE) + (n & 15) * 16;
  const int ra = wn_first(xi) * WN_LINE, rb = wn_second(xi) * WN_LINE;
  const float sxi = wn_sign(xi);

  // ---- B operands, built in registers.  V = (d[ra][cA] + sxi d[rb][cA]) + snu (d[ra][cB] + sxi d[rb][cB]) for this lane's tile and 8 channels:
  // per half (4 channels) four 16-byte reads, 12 fused multiply-adds (exact additions), one split into two binary16 pairs.
  const int nu0 = 2 * nup;
  int oAB[2][2];                                                       // [v][A / B]: byte offset of the position's two patch columns
  float snu[2];
#pragma unroll
  for (int v = 0; v < 2; ++v) {
    const int cA = wn_first(nu0 + v), cB = wn_second(nu0 + v);
    oAB[v][0] = (cA & 1) * 320 + (cA >> 1) * 16;
    oAB[v][1] = (cB & 1) * 320 + (cB >> 1) * 16;
    snu[v] = wn_sign(nu0 + v);
  }
  struct Ld {
    float4 aA, bA, aB, bB;
  };
  auto ld_half = [&](const char* land, int u, int h, Ld& L) {          // unit u = (v = u >> 1, nb = u & 1)
    const char* th = land + (u & 1) * (4 * WN_LINE) + h * (WN_ROWS * WN_LINE);
    const int oA = oAB[u >> 1][0], oB = oAB[u >> 1][1];
    L.aA = *(const float4*)(th + ra + oA);
    L.bA = *(const float4*)(th + rb + oA);
    L.aB = *(const float4*)(th + ra + oB);
    L.bB = *(const float4*)(th + rb + oB);
  };
  auto xf_half = [&](int u, const Ld& L, unsigned& h0, unsigned& h1, unsigned& l0, unsigned& l1) {
    const float s = snu[u >> 1];
    const float v0 = fmaf(s, fmaf(sxi, L.bB.x, L.aB.x), fmaf(sxi, L.bA.x, L.aA.x));
    const float v1 = fmaf(s, fmaf(sxi, L.bB.y, L.aB.y), fmaf(sxi, L.bA.y, L.aA.y));
    const float v2 = fmaf(s, fmaf(sxi, L.bB.z, L.aB.z), fmaf(sxi, L.bA.z, L.aA.z));
    const float v3 = fmaf(s, fmaf(sxi, L.bB.w, L.aB.w), fmaf(sxi, L.bA.w, L.aA.w));
    split2_f16_pair(v0, v1, h0, l0);
    split2_f16_pair(v2, v3, h1, l1);
    f16_max = fmaxf(f16_max, fmaxf(fmaxf(fabsf(v0), fabsf(v1)), fmaxf(fabsf(v2), fabsf(v3))));
  };
  struct Af {
    uint4 uh, ul;
  };
  auto ld_a = [&](int v, int mt) {
    const char* slot = wring + v * POSB + lane * 16 + mt * 2048;
    Af a;
    a.uh = *(const uint4*)slot;
    a.ul = *(const uint4*)(slot + 1024);
    return a;
  };

  issue_land(0);
  issue_w(0, 0);
  issue_w(0, 1);
  auto chunk_body = [&](int c, auto more_c) {                          // (two instantiations: waits with compile-time counts, no branches inside)
    constexpr bool more = decltype(more_c)::value;
    DPX_WN_STAMP(1 + c * 8);
    dpx_wait_vm<2 * NWP>();                                            // (in flight behind land(c): the two position slots of this chunk)
    DPX_WN_STAMP(2 + c * 8);
    zero_land(c);
    DPX_LDS_BARRIER();                                                 // land(c) complete; everybody has left chunk c - 1: its landing buffer is free
    DPX_WN_STAMP(3 + c * 8);
    if (more) issue_land(c + 1);
    const char* land = smem_wn + (c & 1) * WN_LAND_BYTES + lbase;
    // The chunk is four units (position v, column block nb) of MT steps of three matrix instructions.  While a unit's instructions run the NEXT
    // unit's B operand is built -- its reads go out in front of step 0, the first half is transformed and the second half's reads go out in front
    // of step 1, the second half is transformed in front of step 2 -- and every step's weights are read one step ahead: at MT = 3 the 64 registers
    // beside the 192 accumulators hold two weight fragments (16), two B operands (16), four landed pieces (16) and the addresses.
    unsigned bw[8];                                                    // the current unit's operand: hi words 0 .. 3, lo words 4 .. 7
    {
      Ld L;
      ld_half(land, 0, 0, L);
      xf_half(0, L, bw[0], bw[1], bw[4], bw[5]);
      ld_half(land, 0, 1, L);
      xf_half(0, L, bw[2], bw[3], bw[6], bw[7]);
    }
    dpx_wait_vm<NWP + (more ? NLAND : 0)>();                           // slot 0 landed (in flight behind it: slot 1, land(c + 1))
    Af a_cur = ld_a(0, 0);
    __builtin_amdgcn_sched_barrier(0);
    DPX_WN_STAMP(4 + c * 8);
    static_assert(MT <= 2, "96 output channels: 192 accumulators + the pipeline's 60 registers do not fit a 256-register wave (DESIGN.md section 9.2)");
    constexpr bool PIPE = MT < 3;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int v = u >> 1, nb = u & 1;
      unsigned nw[8];
      Ld L;
      if (!PIPE && u > 0) {
        ld_half(land, u, 0, L);
        xf_half(u, L, bw[0], bw[1], bw[4], bw[5]);
        __builtin_amdgcn_sched_barrier(0);
        ld_half(land, u, 1, L);
        xf_half(u, L, bw[2], bw[3], bw[6], bw[7]);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        // in front of the step's matrix instructions: the next unit's operand advances one stage (its reads were issued a step ago)
        if (PIPE && u < 3) {
          if (mt == 0) ld_half(land, u + 1, 0, L);
          if (mt == 1) {
            xf_half(u + 1, L, nw[0], nw[1], nw[4], nw[5]);
            ld_half(land, u + 1, 1, L);
          }
          if (mt == 2) xf_half(u + 1, L, nw[2], nw[3], nw[6], nw[7]);
        }
        __builtin_amdgcn_sched_barrier(0);
        const uint4 bh = make_uint4(bw[0], bw[1], bw[2], bw[3]), bl = make_uint4(bw[4], bw[5], bw[6], bw[7]);
        // (the cross terms are 2^-11 of the leading product: their place inside the fp32 sum is immaterial; uh's registers become uh2's)
        acc[v][mt][nb] = mfma_f16(a_cur.ul, bh, acc[v][mt][nb]);
        acc[v][mt][nb] = mfma_f16(a_cur.uh, bh, acc[v][mt][nb]);
        acc[v][mt][nb] = mfma_f16(wn_shift11(a_cur.uh), bl, acc[v][mt][nb]);
        __builtin_amdgcn_sched_barrier(0);
        // behind them: the next step's weights into the same registers (the other wave of this SIMD covers the read's latency)
        if (mt + 1 < MT) a_cur = ld_a(v, mt + 1);
        else if (u < 3) {
          if (u == 1) dpx_wait_vm<(more ? NLAND : 0)>();               // slot 1 landed (in flight behind it: land(c + 1))
          a_cur = ld_a((u + 1) >> 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (PIPE && u < 3) {                                             // (what MT < 3 steps did not reach)
        if (MT == 1) {
          xf_half(u + 1, L, nw[0], nw[1], nw[4], nw[5]);
          ld_half(land, u + 1, 1, L);
        }
        if (MT <= 2) xf_half(u + 1, L, nw[2], nw[3], nw[6], nw[7]);
#pragma unroll
        for (int k = 0; k < 8; ++k) bw[k] = nw[k];
        __builtin_amdgcn_sched_barrier(0);
      }
      if ((u & 1) && more) {                                           // both units of position v are done: its slot takes the next chunk's weights
        dpx_wait_lds();
        issue_w(c + 1, v);
      }
      DPX_WN_STAMP(5 + u + c * 8);
    }
  };
  for (int c = 0; c + 1 < chunks; ++c) chunk_body(c, std::true_type());
  chunk_body(chunks - 1, std::false_type());
  if (!(f16_max <= 6.0e4f)) atomicOr(&g_f16_overflow, 1u);             // (NaN counts)

  // ---- epilogue: Y[i][j] = sum_xi At[i][xi] sum_nu M[xi][nu] At[j][nu],  At = [1 1 1 0; 0 1 -1 -1] --------------------------------------
  DPX_WN_STAMP(49);
  const float inv_scale = bias[MT * 32];
  const int rw = tid >> 6, rnb = rw & 1, rq = rw >> 1;                  // reader thread: column block, register quad; its lane = (k-group, tile)
  const int rty = 2 * rnb + (n >> 4), rtx = n & 15;
  const int yy0 = y0 + 2 * rty, xx0 = x0 + 2 * rtx;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    DPX_LDS_BARRIER();                                                 // the main loop's / the previous block's LDS reads are over
    DPX_WN_STAMP(50 + mt * 4);
    // this wave's partial sums over its two positions: Z[j] = sum_nu At[j][nu] M[nu];  nup = 0: (M0 + M1, M1), nup = 1: (M2, -M2 - M3)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float4 m0 = make_float4(acc[0][mt][nb][4 * q], acc[0][mt][nb][4 * q + 1], acc[0][mt][nb][4 * q + 2], acc[0][mt][nb][4 * q + 3]);
        float4 m1 = make_float4(acc[1][mt][nb][4 * q], acc[1][mt][nb][4 * q + 1], acc[1][mt][nb][4 * q + 2], acc[1][mt][nb][4 * q + 3]);
        const float4 s = make_float4(m0.x + m1.x, m0.y + m1.y, m0.z + m1.z, m0.w + m1.w);
        const float4 z0 = nup ? m0 : s;
        const float4 z1 = nup ? make_float4(-s.x, -s.y, -s.z, -s.w) : m1;
        *(float4*)(smem_wn + ((((wv * 2 + 0) * 2 + nb) * 4 + q) * 1024) + lane * 16) = z0;
        *(float4*)(smem_wn + ((((wv * 2 + 1) * 2 + nb) * 4 + q) * 1024) + lane * 16) = z1;
      }
    DPX_WN_STAMP(51 + mt * 4);
    DPX_LDS_BARRIER();
    DPX_WN_STAMP(52 + mt * 4);
    const int cl = mt * 32 + 8 * rq + 4 * kg;                           // first of this thread's 4 output channels
    const int cg = mt * 4 + rq;
    const float4 bs = *(const float4*)(bias + cl);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float4 S[4];
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const float4 a = *(const float4*)(smem_wn + (((((2 * x) * 2 + j) * 2 + rnb) * 4 + rq) * 1024) + lane * 16);
        const float4 c2 = *(const float4*)(smem_wn + (((((2 * x + 1) * 2 + j) * 2 + rnb) * 4 + rq) * 1024) + lane * 16);
        S[x] = make_float4(a.x + c2.x, a.y + c2.y, a.z + c2.z, a.w + c2.w);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float4 y;
        if (i == 0) y = make_float4((S[0].x + S[1].x) + S[2].x, (S[0].y + S[1].y) + S[2].y, (S[0].z + S[1].z) + S[2].z, (S[0].w + S[1].w) + S[2].w);
        else y = make_float4((S[1].x - S[2].x) - S[3].x, (S[1].y - S[2].y) - S[3].y, (S[1].z - S[2].z) - S[3].z, (S[1].w - S[2].w) - S[3].w);
        float4 v = make_float4(fmaf(y.x, inv_scale, bs.x), fmaf(y.y, inv_scale, bs.y), fmaf(y.z, inv_scale, bs.z), fmaf(y.w, inv_scale, bs.w));
        if (RELU) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
        const int yy = yy0 + i, xx = xx0 + j;
        if (cg < Gout && yy < H && xx < W)
          *(float4*)(out + (size_t)b * Gout * H * W * 8 + (((size_t)cg * H + yy) * W + xx) * 8 + 4 * kg) = v;
      }
    }
  }
  DPX_WN_STAMP(62);
  }  // tile loop
}
#ifdef DPX_WN_TRACE
}  // namespace dpx
extern "C" int dpx_dbg_wn_trace(unsigned long long* host, int n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(dpx::dpx_wn_trace_buf), sizeof(unsigned long long) * (n < 512 ? n : 512)) == hipSuccess ? 0 : -1;
}
namespace dpx {
#endif

template <int MT>
static void launch_wino(bool relu, const float* in, float* out, const char* wpk, int Gin, int Gout, int B, int H, int W, hipStream_t s) {
  const int tx = (W + WN_TW - 1) / WN_TW, ty = (H + WN_TH - 1) / WN_TH;
  const size_t sh = wn_smem_bytes(MT);
#ifdef DPX_EMULATED
  const int ncu = 3;                                                   // (the host emulator: a few tiles per workgroup)
#else
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
  }
#endif
  const int tiles_all = tx * ty * B, nwg = tiles_all < ncu ? tiles_all : ncu;
  static bool attr[2] = {false, false};
  if (!attr[relu]) {
    if (relu) hipFuncSetAttribute((const void*)k_conv3x3_wino<MT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    else hipFuncSetAttribute((const void*)k_conv3x3_wino<MT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    attr[relu] = true;
  }
  if (relu) DPX_LAUNCH("k_conv3x3_wino", (k_conv3x3_wino<MT, true>), dim3(nwg), dim3(512), sh, s, in, out, wpk, Gin, Gout, H, W, tx, tx * ty, tiles_all);
  else DPX_LAUNCH("k_conv3x3_wino", (k_conv3x3_wino<MT, false>), dim3(nwg), dim3(512), sh, s, in, out, wpk, Gin, Gout, H, W, tx, tx * ty, tiles_all);
}
static void launch_wino_mt(int mt, bool relu, const float* in, float* out, const char* wpk, int Gin, int Gout, int B, int H, int W, hipStream_t s) {
  switch (mt) {
    case 1: launch_wino<1>(relu, in, out, wpk, Gin, Gout, B, H, W, s); break;
    default: launch_wino<2>(relu, in, out, wpk, Gin, Gout, B, H, W, s); break;     // (up to 64 output channels: see bx_layer_is_wino)
  }
}

}  // namespace dpx
