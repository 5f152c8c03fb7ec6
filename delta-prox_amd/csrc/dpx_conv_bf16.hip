// 3x3 convolution layers at fp32 accuracy on the bf16 matrix cores ("split-bf16": x = hi + mid + lo, three bf16 terms that
// together hold all 24 mantissa bits of an fp32 value, split by truncation so that the decomposition is EXACT).
//
//   y = sum_k w_k a_k  with  w = wh + wm + wl,  a = ah + am + al   =>   six products of order <= 2 are accumulated in fp32 by
//   v_mfma_f32_32x32x16_bf16:   wh ah | wh am | wm ah | wh al | wl ah | wm am        (dropped: wm al, wl am, wl al <= 2^-24 |w a|,
//   the size of one fp32 rounding of the product itself).  6 x 32 cycles per 16 input channels and 32 x 32 outputs instead of
//   8 x 64 cycles with v_mfma_f32_32x32x2_f32: 2.67x the fp32-matrix-core rate.   MODE = 1 keeps only  wh ah  (plain bf16).
//
// Data layout ("C8"): activations [B][C/8][H][W][8] fp32 -- eight channels of a pixel are 32 contiguous bytes, so that
//   * the producing layer's D tile (lane = pixel, 4 consecutive output channels per register quad) is written with 16-byte
//     stores, 1 KB contiguous per wave instruction,
//   * a (pixel, 8-channel group) is exactly the B operand of one lane of the 32x32x16 MFMA.
// Activations stay fp32 in HBM (4 B / element as before); the workgroup splits the staged tile once per 16-channel chunk.
//
// Workgroup = 512 threads = 8 waves, output tile 16 rows x 32 columns x all output channels (MT x 32); wave w owns rows 2w, 2w+1.
// Per 16-channel chunk:  LDS-DMA landing buffer (fp32, next chunk in flight while this one multiplies) -> split pass -> bf16 tile
// [plane][k-group][row][col][8];  weights arrive by LDS-DMA in slots of 3 taps, pre-split at pack time, ring of 2 slots, one
// workgroup barrier per slot.  LDS: 39 + 57 + 54 KB (MT = 3): one workgroup per CU, two waves per SIMD.
#include "dpx_common.h"
#include "dpx_prox_dev.h"
#include "dpx_mma_dev.h"

// Tuning probes (wrong results by design; DESIGN.md section 3 has what they measured): 1 tap-independent fragment reads,
// 2 no split pass, 4 no DMA / waits / barriers, 8 no DMA (barriers stay), 16 DMA never waited for,
// 32 / 64 every activation piece from one cached KB of zeros / of image data.
#ifndef DPX_BX_DBG
#define DPX_BX_DBG 0
#endif
namespace dpx {

__device__ unsigned g_f16_overflow = 0u;                              // set by any split-f16 layer that met |x| > 6e4 (NaN counts)
// (for the kernels of other translation units -- dpx_wgrad_c8.hip -- which receive the word's address as an argument)
unsigned* f16_overflow_flag() {
  static unsigned* p = nullptr;
  if (!p) hipGetSymbolAddress((void**)&p, HIP_SYMBOL(g_f16_overflow));
  return p;
}

constexpr int BX_TW = 32, BX_TH = 16, BX_ROWS = BX_TH + 2, BX_COLS = BX_TW + 2;
constexpr int BX_UNITS = 2 * BX_ROWS * BX_COLS;                      // (k-group, row, col) units of 8 channels: 1224
constexpr int BX_PIECES = 2 * BX_UNITS;                              // 16-byte LDS-DMA pieces of one fp32 chunk: 2448
constexpr int BX_LAND_BYTES = ((BX_PIECES + 63) / 64) * 1024;        // 39936 (whole 1 KB instructions)
constexpr int BX_PLANE_BYTES = BX_UNITS * 16;                        // one bf16 plane of the tile: 19584
constexpr int BX_TILE_BYTES = 3 * BX_PLANE_BYTES;
constexpr int BX_NPI = (BX_PIECES + 511) / 512;                      // DMA instructions per thread and chunk: 5
// weight planes of an arithmetic mode (1 plain bf16 and 6 split-bf16: three plane slots; 3 split-f16: two)
__host__ __device__ constexpr int bx_planes(int mode) { return mode == 3 ? 2 : 3; }
__host__ __device__ constexpr int bx_tap_bytes(int MT, int NPW) { return NPW * 2 * MT * 32 * 16; }     // [plane][k-group][cout][8] 16-bit
__host__ __device__ constexpr int bx_slot_bytes(int MT, int NPW) { return 3 * bx_tap_bytes(MT, NPW); }  // 3 taps

// packed layer: [chunk][tap group 3][tap 3][plane NPW][k-group 2][cout MT*32][8] 16-bit, then bias fp32 [MT*32], then 64 zero bytes
static inline size_t bx_layer_bytes(int cin, int cout, int NPW = 3) {
  const int chunks = (cin + 15) / 16, MT = (cout + 31) / 32;
  return (size_t)chunks * 3 * bx_slot_bytes(MT, NPW) + (size_t)MT * 32 * 4 + 64;
}

// w [cout][cin][9] fp32, b [cout] (nullable) -> packed layer.  mode 6: three exact planes; mode 1: plane 0 = RNE bf16, others 0.
// tr: the backward-data layer of w -- (cin, cout) are ITS channel counts (the forward layer's cout, cin), the tensor is the forward
// layer's [cin][cout][9] and the taps are flipped: v = w[ci][co][8 - tap]
__global__ void k_bx_pack_weights(const float* __restrict__ w, const float* __restrict__ b, unsigned short* __restrict__ dst, int cin, int cout,
                                  int mode, int tr) {
  const int chunks = (cin + 15) / 16, MT = (cout + 31) / 32, M32 = MT * 32, NPW = bx_planes(mode);
  const long nw = (long)chunks * 9 * NPW * 2 * M32 * 8;                // 16-bit elements
  float* bias = (float*)(dst + nw);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nw + M32 + 16; i += (long)gridDim.x * blockDim.x) {
    if (i >= nw) {
      const int k = (int)(i - nw);
      bias[k] = (k < M32 && k < cout && b) ? b[k] : 0.f;
      continue;
    }
    const int j = (int)(i % 8);
    long r = i / 8;
    const int co = (int)(r % M32);
    r /= M32;
    const int kg = (int)(r % 2);
    r /= 2;
    const int plane = (int)(r % NPW);
    r /= NPW;
    const int tap = (int)(r % 9), chunk = (int)(r / 9);               // [chunk][tap group][tap in group] == [chunk][tap]
    const int ci = chunk * 16 + kg * 8 + j;
    const float v = (co < cout && ci < cin) ? (tr ? w[((long)ci * cout + co) * 9 + (8 - tap)] : w[((long)co * cin + ci) * 9 + tap]) : 0.f;
    unsigned h, m, l;
    if (mode == 1) {
      h = bf16_rne(v);
      m = l = 0u;
    } else if (mode == 3) {
      split2_f16(v, h, m);
      l = 0u;
      if (fabsf(v) > 6.0e4f) atomicOr(&g_f16_overflow, 1u);
    } else {
      split3(v, h, m, l);
    }
    dst[i] = (unsigned short)((plane == 0 ? h : (plane == 1 ? m : l)) >> 16);
  }
}

// FFDNet input stage into the C8 layout: replicate-pad to even size, pixel-unshuffle(2) (channel = c*4 + dy*2 + dx), sigma map as
// channel 4C, zero fill up to a multiple of 16 channels               (network_ffdnet.py:56-63)
// p8 != 0: the "P8" form of the split-f16 operand planes -- a (pixel, 8-channel group) unit is [hi: 8 x binary16][lo: 8 x binary16], the
// same 32 bytes as its eight fp32 values, already split (split2_f16): what the consuming layer's LDS-DMA lands IS its operand tile.
__global__ void k_bx_pack_in(const float* __restrict__ x, const float* __restrict__ sigma, float* __restrict__ a, int B, int C, int H, int W,
                             int H2, int W2, int G, int p8) {
  const long total = (long)B * G * H2 * W2 * 8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int j = (int)(i % 8);
    long r = i / 8;
    const int x2 = (int)(r % W2);
    r /= W2;
    const int y2 = (int)(r % H2);
    r /= H2;
    const int g = (int)(r % G), b = (int)(r / G);
    const int ch = g * 8 + j;
    float v = 0.f;
    if (ch < 4 * C) {
      const int c = ch >> 2, dy = (ch >> 1) & 1, dx = ch & 1;
      const int yy = min(2 * y2 + dy, H - 1), xx = min(2 * x2 + dx, W - 1);
      v = x[(((long)b * C + c) * H + yy) * W + xx];
    } else if (ch == 4 * C) {
      v = sigma[b];
    }
    if (p8) {
      unsigned hw, lw;
      split2_f16(v, hw, lw);
      unsigned short* unit = (unsigned short*)a + (i - j) * 2;          // 32 bytes per unit
      unit[j] = (unsigned short)(hw >> 16);
      unit[8 + j] = (unsigned short)(lw >> 16);
      if (fabsf(v) > 6.0e4f) atomicOr(&g_f16_overflow, 1u);
    } else {
      a[i] = v;
    }
  }
}

// C8 [B][G][H2][W2][8] (first 4C channels) -> y [B][C][H][W]: PixelShuffle(2) + crop      (network_ffdnet.py:65-67)
__global__ void k_bx_unpack_out(const float* __restrict__ o, float* __restrict__ y, int B, int C, int H, int W, int H2, int W2, int G) {
  const long total = (long)B * C * H * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int xx = (int)(i % W);
    long r = i / W;
    const int yy = (int)(r % H);
    r /= H;
    const int c = (int)(r % C), b = (int)(r / C);
    const int ch = c * 4 + (yy & 1) * 2 + (xx & 1);
    y[i] = o[((((long)b * G + (ch >> 3)) * H2 + (yy >> 1)) * W2 + (xx >> 1)) * 8 + (ch & 7)];
  }
}

// in / out: C8 fp32; Gin = input channel groups (even: chunks of 2), Gout = output groups actually stored.
// PRE / PSO (split-f16 only): the input arrives / the output leaves in the P8 form (k_bx_pack_in) -- the PRODUCING layer's epilogue
// splits its fp32 results into the two binary16 planes (the same split2_f16 of the same fp32 values the consumer's split pass
// performed: bit-identical operands), the consumer's LDS-DMA lands them directly as its operand tile (pieces ordered plane by plane),
// two tiles alternating: no landing buffer, no split pass, one workgroup barrier less per 16-channel chunk.
// TH: rows of the workgroup's tile (16: eight waves, one workgroup per CU; 8: four waves and half the LDS -- two workgroups per CU --
// for launches whose 16-row tiles would leave CUs idle: 4 x 1 x 320 x 320 is 200 tiles of 16 rows on 256 CUs)
#ifdef DPX_WN_TRACE
__device__ unsigned long long dpx_bx_trace_buf[8 * 64];
#define DPX_BX_STAMP(i)                                                                                                                         \
  do {                                                                                                                                         \
    if (MT == 3 && MODE == 3 && Gin > 2 && (tid & 63) == 0 && blockIdx.x == 300 && blockIdx.y == 0 && (i) < 64) dpx_bx_trace_buf[(tid >> 6) * 64 + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define DPX_BX_STAMP(i) ((void)0)
#endif
template <int MT, bool RELU, int MODE, bool PRE = false, bool PSO = false, int TH = BX_TH>
// mask (nullable; C8, Gout groups): the stored value is zeroed where mask <= 0 -- the ReLU derivative of the backward-data pass
// ([a_l > 0] from the saved forward activation), applied in the producing layer's epilogue.
__global__ void __launch_bounds__(TH * 32, TH == 16 ? 1 : 2) k_conv3x3_bf16(const float* __restrict__ in, float* __restrict__ out, const char* __restrict__ wpk, int Gin,
                                                        int Gout, int H, int W, int tiles_x, const float* __restrict__ mask) {
  static_assert(MODE == 3 || (!PRE && !PSO), "pre-split operand planes exist for the split-f16 arithmetic only");
  constexpr int M32 = MT * 32, NPW = bx_planes(MODE), TAPB = bx_tap_bytes(MT, NPW), SLOTB = bx_slot_bytes(MT, NPW);
  constexpr int NPL = MODE == 1 ? 1 : (MODE == 3 ? 2 : 3);            // operand planes in use (the packed layouts always have room for three)
  static_assert(TH == 16 || (TH == 8 && !PRE && !PSO), "tile rows");
  // the tile geometry (TH = 16: the BX_* constants above)
  constexpr int NWV = TH / 2, NT = NWV * 64, ROWS = TH + 2, UNITS = 2 * ROWS * BX_COLS, PIECES = 2 * UNITS;
  constexpr int LAND_BYTES = ((PIECES + 63) / 64) * 1024, PLANE_BYTES = UNITS * 16, NPI = (PIECES + NT - 1) / NT;
  constexpr int TILE_BYTES = (TH == 16 ? 3 : NPL) * PLANE_BYTES;
  HIP_DYNAMIC_SHARED(char, smem_bx)
  char* land = smem_bx;
  char* tile = PRE ? smem_bx : smem_bx + LAND_BYTES;               // PRE: two tiles of two planes, [c & 1], LAND_BYTES apart (the
  char* ring = PRE ? smem_bx + 2 * LAND_BYTES : tile + TILE_BYTES;      // last DMA instruction of a tile is a whole KB: 768 bytes of slack)
  const int tid = threadIdx.x, lane = tid & 63;
  DPX_BX_STAMP(0);
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int y0 = ty * TH, x0 = tx * BX_TW;
  const int n = lane & 31, kg = lane >> 5;
  const int chunks = Gin / 2;
  const float* bias = (const float*)(wpk + (size_t)chunks * 3 * SLOTB);
  const float* zero_block = bias + M32;                               // 64 zero bytes behind the bias
  const float* inb = in + (size_t)b * Gin * H * W * 8;

  // per-lane source offset (floats, relative to the chunk's first group) of DMA piece k; ~0u = outside the image.  MT = 3 (192 accumulators)
  // recomputes it for every chunk (~20 integer instructions per piece against 162 matrix instructions per chunk) instead of keeping NPI registers
  // -- with them the split-f16 forms spilled 4 - 6 registers; the smaller forms keep the table.
  constexpr bool POFF_TABLE = MT < 3;
  auto piece_off = [&](int k, int tid_) {
    const int q = k * NT + tid_;
    const int u = PRE ? (q < UNITS ? q : q - UNITS) : q >> 1, half = PRE ? (q < UNITS ? 0 : 1) : q & 1;      // PRE: plane by plane
    const int g = u / (ROWS * BX_COLS), rem = u - g * (ROWS * BX_COLS);
    const int row = rem / BX_COLS, col = rem - row * BX_COLS;
    const int yy = y0 + row - 1, xx = x0 + col - 1;
    const bool ok = q < PIECES && yy >= 0 && yy < H && xx >= 0 && xx < W;
    return ok ? (unsigned)(((size_t)g * H * W + (size_t)yy * W + xx) * 8 + half * 4) : ~0u;
  };
  unsigned poff[POFF_TABLE ? NPI : 1];
  if constexpr (POFF_TABLE) {
#pragma unroll
    for (int k = 0; k < NPI; ++k) poff[k] = piece_off(k, tid);
  }
  auto issue_act = [&](int c) {
    if (DPX_BX_DBG & 8) return;
    const float* cb = inb + (size_t)(2 * c) * H * W * 8;
    int tid_c = tid;
    if constexpr (!POFF_TABLE) DPX_OPAQUE(tid_c);                     // (per chunk: the offsets must not be hoisted out of the chunk loop)
#pragma unroll
    for (int k = 0; k < NPI; ++k) {
      if (k * NT + wv * 64 < PIECES) {                            // wave-uniform: whole 1 KB instructions
        unsigned po;
        if constexpr (POFF_TABLE) po = poff[k];
        else po = piece_off(k, tid_c);
        const float* src = (po != ~0u && !(DPX_BX_DBG & 32)) ? cb + po : zero_block;
        if (DPX_BX_DBG & 64) src = inb + (tid & 63) * 4;
        dpx_glds16(src, (PRE ? smem_bx + (c & 1) * LAND_BYTES : land) + (k * NT + wv * 64) * 16);
      }
    }
  };
  auto issue_w = [&](int slot_idx) {                                  // global slot index = chunk * 3 + tap group
    if (DPX_BX_DBG & 8) return;
    const char* src = wpk + (size_t)slot_idx * SLOTB + lane * 16;
    char* dst = ring + (slot_idx & 1) * SLOTB;
    for (int i = wv; i < SLOTB / 1024; i += NWV) dpx_glds16(src + i * 1024, dst + i * 1024);
  };

  f32x16 acc[MT][2];
  f32x16 accx[MODE == 3 ? MT : 1][2];                                 // split-f16: the two cross terms, scaled by 2^11
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        acc[mt][r][i] = 0.f;
        if (MODE == 3) accx[MODE == 3 ? mt : 0][r][i] = 0.f;
      }
  float f16_max = 0.f;                                                // split-f16: largest |operand| this thread has split

  const int nslots = chunks * 3;
  if (!(DPX_BX_DBG & 4)) {
    issue_act(0);
    issue_w(0);
  }
  for (int c = 0; c < chunks; ++c) {
    // ---- the chunk's activations: landed -> split into bf16 planes ------------------------------------------------------
    DPX_BX_STAMP(1 + c * 8);
    if (!(DPX_BX_DBG & 4)) {
    if (!(DPX_BX_DBG & 16)) dpx_wait_vm<0>();
    DPX_BX_STAMP(2 + c * 8);
    DPX_LDS_BARRIER();                                                // landing buffer complete; everybody is done with the old tile
    }
    DPX_BX_STAMP(3 + c * 8);
    if constexpr (PRE) {
      tile = smem_bx + (c & 1) * LAND_BYTES;                       // landed as the operand planes themselves; the other tile is free:
      if (c + 1 < chunks) issue_act(c + 1);                           // every wave has left chunk c - 1 (barrier above)
    }
    for (int u = tid; u < ((PRE || (DPX_BX_DBG & 2)) ? 0 : UNITS); u += NT) {
      const float4 lo4 = *(const float4*)(land + u * 32), hi4 = *(const float4*)(land + u * 32 + 16);
      const float v[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
      if constexpr (MODE == 3) {
        unsigned hw[4], lw[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) split2_f16_pair(v[2 * j], v[2 * j + 1], hw[j], lw[j]);
        *(uint4*)(tile + u * 16) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
        *(uint4*)(tile + PLANE_BYTES + u * 16) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
        const float m8 = fmaxf(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))),
                               fmaxf(fmaxf(fabsf(v[4]), fabsf(v[5])), fmaxf(fabsf(v[6]), fabsf(v[7]))));
        f16_max = fmaxf(f16_max, m8);
        continue;
      }
      unsigned h[8], m[8], l[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if constexpr (MODE == 1) {
          h[j] = bf16_rne(v[j]);
          m[j] = l[j] = 0u;
        } else {
          split3(v[j], h[j], m[j], l[j]);
        }
      }
      *(uint4*)(tile + u * 16) = make_uint4(pack_hi16(h[0], h[1]), pack_hi16(h[2], h[3]), pack_hi16(h[4], h[5]), pack_hi16(h[6], h[7]));
      if constexpr (MODE != 1)
        *(uint4*)(tile + PLANE_BYTES + u * 16) = make_uint4(pack_hi16(m[0], m[1]), pack_hi16(m[2], m[3]), pack_hi16(m[4], m[5]), pack_hi16(m[6], m[7]));
      if constexpr (MODE == 6)
        *(uint4*)(tile + 2 * PLANE_BYTES + u * 16) = make_uint4(pack_hi16(l[0], l[1]), pack_hi16(l[2], l[3]), pack_hi16(l[4], l[5]), pack_hi16(l[6], l[7]));
    }
    DPX_BX_STAMP(4 + c * 8);
    if (!PRE && !(DPX_BX_DBG & 4)) {
    DPX_LDS_BARRIER();                                                // tile ready, landing buffer free
    if (c + 1 < chunks) issue_act(c + 1);
    }
    DPX_BX_STAMP(5 + c * 8);
    // ---- three slots of three taps ----------------------------------------------------------------------------------------
    for (int tg = 0; tg < 3; ++tg) {
      const int s = c * 3 + tg;
      if (tg > 0 && !(DPX_BX_DBG & 4)) {                               // (tg == 0: the barrier pair above already covered slot s)
        if (!(DPX_BX_DBG & 16)) dpx_wait_vm<0>();
        DPX_LDS_BARRIER();                                            // slot s landed; slot s - 1 no longer read by anybody
      }
      if (s + 1 < nslots && !(DPX_BX_DBG & 4)) issue_w(s + 1);
      DPX_BX_STAMP(6 + tg + c * 8);
      const char* wslot = ring + (s & 1) * SLOTB;
#pragma unroll
      for (int t3 = 0; t3 < 3; ++t3) {
        const int tap = tg * 3 + t3, dy = tap / 3, dx = tap - dy * 3;
        uint4 bf[2][NPL];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int u = (DPX_BX_DBG & 1) ? (kg * ROWS + 2 * wv + r) * BX_COLS + n : (kg * ROWS + 2 * wv + r + dy) * BX_COLS + n + dx;
#pragma unroll
          for (int p = 0; p < NPL; ++p) bf[r][p] = *(const uint4*)(tile + p * PLANE_BYTES + u * 16);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          uint4 af[NPL];
#pragma unroll
          for (int p = 0; p < NPL; ++p) af[p] = *(const uint4*)(wslot + ((DPX_BX_DBG & 1) ? 0 : t3 * TAPB) + ((p * 2 + kg) * M32 + mt * 32 + n) * 16);
          if constexpr (MODE == 3) {
            // (an accumulator is written again four matrix instructions later: no wait for the previous result)
            accx[mt][0] = mfma_f16(af[1], bf[0][0], accx[mt][0]);     // wl ah   (x 2^11)
            accx[mt][1] = mfma_f16(af[1], bf[1][0], accx[mt][1]);
            acc[mt][0] = mfma_f16(af[0], bf[0][0], acc[mt][0]);       // wh ah
            acc[mt][1] = mfma_f16(af[0], bf[1][0], acc[mt][1]);
            accx[mt][0] = mfma_f16(af[0], bf[0][1], accx[mt][0]);     // wh al   (x 2^11)
            accx[mt][1] = mfma_f16(af[0], bf[1][1], accx[mt][1]);
          }
          if constexpr (MODE == 1) {
            acc[mt][0] = mfma_bf16(af[0], bf[0][0], acc[mt][0]);
            acc[mt][1] = mfma_bf16(af[0], bf[1][0], acc[mt][1]);
          } else if constexpr (MODE == 6) {
            // small terms first, the leading product last; the two rows alternate so that an accumulator is not written by two
            // consecutive matrix instructions
#pragma unroll
            for (int r = 0; r < 2; ++r) acc[mt][r] = mfma_bf16(af[1], bf[r][1], acc[mt][r]);    // wm am
#pragma unroll
            for (int r = 0; r < 2; ++r) acc[mt][r] = mfma_bf16(af[2], bf[r][0], acc[mt][r]);    // wl ah
#pragma unroll
            for (int r = 0; r < 2; ++r) acc[mt][r] = mfma_bf16(af[0], bf[r][2], acc[mt][r]);    // wh al
#pragma unroll
            for (int r = 0; r < 2; ++r) acc[mt][r] = mfma_bf16(af[1], bf[r][0], acc[mt][r]);    // wm ah
#pragma unroll
            for (int r = 0; r < 2; ++r) acc[mt][r] = mfma_bf16(af[0], bf[r][1], acc[mt][r]);    // wh am
#pragma unroll
            for (int r = 0; r < 2; ++r) acc[mt][r] = mfma_bf16(af[0], bf[r][0], acc[mt][r]);    // wh ah
          }
        }
      }
    }
  }
  DPX_BX_STAMP(49);
  if (MODE == 3 && !PSO && !(f16_max <= 6.0e4f)) atomicOr(&g_f16_overflow, 1u);        // (NaN counts; PSO: behind the epilogue, which splits the outputs)
  // ---- epilogue: bias, ReLU, C8 store.  D layout: col = lane & 31 (pixel), row = (i & 3) + 8 (i >> 2) + 4 (lane >> 5) (cout) ----
  const int xx = x0 + n;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int yy = y0 + 2 * wv + r;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int cg = mt * 4 + q;                                    // output channel group of this register quad
        float4 v;
        const int cl = mt * 32 + 8 * q + 4 * kg;
        v.x = acc[mt][r][4 * q + 0] + bias[cl + 0];
        v.y = acc[mt][r][4 * q + 1] + bias[cl + 1];
        v.z = acc[mt][r][4 * q + 2] + bias[cl + 2];
        v.w = acc[mt][r][4 * q + 3] + bias[cl + 3];
        if constexpr (MODE == 3) {
          constexpr float s = 1.0f / F16_LO_SCALE;
          v.x = fmaf(accx[mt][r][4 * q + 0], s, v.x);
          v.y = fmaf(accx[mt][r][4 * q + 1], s, v.y);
          v.z = fmaf(accx[mt][r][4 * q + 2], s, v.z);
          v.w = fmaf(accx[mt][r][4 * q + 3], s, v.w);
        }
        if (RELU) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
        if constexpr (PSO) {
          if (cg < Gout && yy < H && xx < W) {
            unsigned h0, l0, h1, l1;
            split2_f16_pair(v.x, v.y, h0, l0);
            split2_f16_pair(v.z, v.w, h1, l1);
            f16_max = fmaxf(f16_max, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
            char* unit = (char*)out + ((size_t)b * Gout * H * W + ((size_t)cg * H + yy) * W + xx) * 32;
            *(uint2*)(unit + 8 * kg) = make_uint2(h0, h1);
            *(uint2*)(unit + 16 + 8 * kg) = make_uint2(l0, l1);
          }
          continue;
        }
        if (cg < Gout && yy < H && xx < W) {
          const size_t o = (size_t)b * Gout * H * W * 8 + (((size_t)cg * H + yy) * W + xx) * 8 + 4 * kg;
          if (mask) {                                                 // (kernel-uniform)
            const float4 m = *(const float4*)(mask + o);
            v = make_float4(m.x > 0.f ? v.x : 0.f, m.y > 0.f ? v.y : 0.f, m.z > 0.f ? v.z : 0.f, m.w > 0.f ? v.w : 0.f);
          }
          *(float4*)(out + o) = v;
        }
      }
    }
  if (PSO && !(f16_max <= 6.0e4f)) atomicOr(&g_f16_overflow, 1u);
  DPX_BX_STAMP(50);
#ifdef DPX_WN_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  DPX_BX_STAMP(51);
#endif
}

// split-f16 inference layers with pre-split operand planes in HBM (pre: input P8, pso: output P8)
template <int MT>
static void launch_bx_p8(bool relu, bool pre, bool pso, const float* in, float* out, const char* wpk, int Gin, int Gout, int B, int H, int W,
                         hipStream_t s) {
  const int tx = (W + BX_TW - 1) / BX_TW, ty = (H + BX_TH - 1) / BX_TH;
  const size_t sh_pre = (size_t)2 * BX_LAND_BYTES + 2 * bx_slot_bytes(MT, 2), sh_std = (size_t)BX_LAND_BYTES + BX_TILE_BYTES + 2 * bx_slot_bytes(MT, 2);
  const dim3 grid(tx * ty, B);
#define DPX_BX_P8(R, PRE_, PSO_)                                                                                                             \
  do {                                                                                                                                       \
    const size_t sh = PRE_ ? sh_pre : sh_std;                                                                                                \
    static bool attr = false;                                                                                                                \
    if (!attr) {                                                                                                                             \
      hipFuncSetAttribute((const void*)k_conv3x3_bf16<MT, R, 3, PRE_, PSO_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);           \
      attr = true;                                                                                                                           \
    }                                                                                                                                        \
    DPX_LAUNCH("k_conv3x3_bf16", (k_conv3x3_bf16<MT, R, 3, PRE_, PSO_>), grid, dim3(512), sh, s, in, out, wpk, Gin, Gout, H, W, tx,            \
               (const float*)nullptr);                                                                                                       \
  } while (0)
  if (pre && pso) { if (relu) DPX_BX_P8(true, true, true); else DPX_BX_P8(false, true, true); }
  else if (pre) { if (relu) DPX_BX_P8(true, true, false); else DPX_BX_P8(false, true, false); }
  else if (pso) { if (relu) DPX_BX_P8(true, false, true); else DPX_BX_P8(false, false, true); }
#undef DPX_BX_P8
}
static void launch_bx_p8_mt(int mt, bool relu, bool pre, bool pso, const float* in, float* out, const char* wpk, int Gin, int Gout, int B, int H,
                            int W, hipStream_t s) {
  switch (mt) {
    case 1: launch_bx_p8<1>(relu, pre, pso, in, out, wpk, Gin, Gout, B, H, W, s); break;
    case 2: launch_bx_p8<2>(relu, pre, pso, in, out, wpk, Gin, Gout, B, H, W, s); break;
    default: launch_bx_p8<3>(relu, pre, pso, in, out, wpk, Gin, Gout, B, H, W, s); break;
  }
}

template <int MT, int MODE, int TH>
static void launch_bx_th(bool relu, const float* in, float* out, const char* wpk, int Gin, int Gout, int B, int H, int W, hipStream_t s,
                         const float* mask) {
  constexpr int NPL = MODE == 1 ? 1 : (MODE == 3 ? 2 : 3), UNITS = 2 * (TH + 2) * BX_COLS;
  const int tx = (W + BX_TW - 1) / BX_TW, ty = (H + TH - 1) / TH;
  const size_t sh = (size_t)((2 * UNITS + 63) / 64) * 1024 + (size_t)(TH == 16 ? 3 : NPL) * UNITS * 16 + 2 * bx_slot_bytes(MT, bx_planes(MODE));
  static bool attr[2] = {false, false};
  if (!attr[relu]) {
    if (relu) hipFuncSetAttribute((const void*)k_conv3x3_bf16<MT, true, MODE, false, false, TH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    else hipFuncSetAttribute((const void*)k_conv3x3_bf16<MT, false, MODE, false, false, TH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    attr[relu] = true;
  }
  if (relu)
    DPX_LAUNCH("k_conv3x3_bf16", (k_conv3x3_bf16<MT, true, MODE, false, false, TH>), dim3(tx * ty, B), dim3(TH * 32), sh, s, in, out, wpk, Gin, Gout, H,
               W, tx, mask);
  else
    DPX_LAUNCH("k_conv3x3_bf16", (k_conv3x3_bf16<MT, false, MODE, false, false, TH>), dim3(tx * ty, B), dim3(TH * 32), sh, s, in, out, wpk, Gin, Gout, H,
               W, tx, mask);
}
// knob conv_tile_rows = 8: 8-row tiles (two workgroups per CU) -- measured on the one launch shape they were made for (4 x 1 x 320 x 320:
// 200 tiles of 16 rows on 256 CUs) they are 2 % SLOWER (a tile is bound by its CU's matrix pipe): off by default
template <int MT, int MODE>
static void launch_bx(bool relu, const float* in, float* out, const char* wpk, int Gin, int Gout, int B, int H, int W, hipStream_t s,
                      const float* mask = nullptr) {
  if (tune(TUNE_CONV_TILE_ROWS) == 8) launch_bx_th<MT, MODE, 8>(relu, in, out, wpk, Gin, Gout, B, H, W, s, mask);
  else launch_bx_th<MT, MODE, 16>(relu, in, out, wpk, Gin, Gout, B, H, W, s, mask);
}
template <int MODE>
static void launch_bx_mt(int mt, bool relu, const float* in, float* out, const char* wpk, int Gin, int Gout, int B, int H, int W, hipStream_t s,
                         const float* mask = nullptr) {
  switch (mt) {
    case 1: launch_bx<1, MODE>(relu, in, out, wpk, Gin, Gout, B, H, W, s, mask); break;
    case 2: launch_bx<2, MODE>(relu, in, out, wpk, Gin, Gout, B, H, W, s, mask); break;
    default: launch_bx<3, MODE>(relu, in, out, wpk, Gin, Gout, B, H, W, s, mask); break;
  }
}

// ---- backward-data of the FFDNet stack on the split kernels: adjoints of the input / output stages in the C8 layout ----------------
// adjoint of k_bx_unpack_out (PixelShuffle + crop): g_last[b][ch][y2][x2] = gy[b][c][2 y2 + dy][2 x2 + dx] inside the image, else 0
// ---- the gradient scale of a split-f16 backward pass ------------------------------------------------------------------------------
// Gradients of a mean loss sit at 1e-7, far below binary16's normal range, and the backward pass is linear in them: the incoming gradient
// is multiplied by a power of two that brings max |gy| into [8, 16) (k_bx_absmax leaves the bits of max |gy| in a word, every consumer
// derives the same factor from it), and gx, d/dsigma, dW, db are multiplied by its inverse -- exact, with 2^12 of headroom upwards for
// what twelve transposed layers do to the magnitude (an operand beyond 6e4 trips dpx_ffdnet_f16_overflow as in the forward pass;
// operands below 2^-14 keep an ABSOLUTE error of 2^-35 through the scaled low part).
__global__ void __launch_bounds__(256) k_bx_absmax(const float* __restrict__ x, long n, unsigned* __restrict__ amax_bits) {
  __shared__ float sh[4];
  float m = 0.f;
  long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  for (; i + 3 < n; i += (long)gridDim.x * 1024) {
    const float4 v = *(const float4*)(x + i);
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  for (; i < n; ++i) m = fmaxf(m, fabsf(x[i]));                         // (the tail: at most three elements of one thread)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(amax_bits, __float_as_uint(fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]))));   // (non-negative floats order like their bits)
}
// 2^(BX_GS - e) with 2^(e-1) <= max |gy| < 2^e, i.e. max |gy| lands in [2^(BX_GS-1), 2^BX_GS); 1 when there is no word (split-bf16 passes), for a
// zero / denormal-sized gradient, and for inf / NaN (the range trap's case)
constexpr int BX_GS = 4;
__device__ __forceinline__ float bx_grad_scale(const unsigned* amax_bits) {
  if (!amax_bits) return 1.f;
  const int ex = (int)(*amax_bits >> 23);
  return (ex < 8 || ex >= 255) ? 1.f : __uint_as_float((unsigned)(253 + BX_GS - ex) << 23);
}
__device__ __forceinline__ float bx_grad_unscale(const unsigned* amax_bits) {
  if (!amax_bits) return 1.f;
  const int ex = (int)(*amax_bits >> 23);
  return (ex < 8 || ex >= 255) ? 1.f : __uint_as_float((unsigned)(ex + 1 - BX_GS) << 23);
}

__global__ void k_bx_pack_gout(const float* __restrict__ gy, float* __restrict__ g, int B, int C, int H, int W, int H2, int W2, int G,
                               const unsigned* __restrict__ amax_bits) {
  const long total = (long)B * G * H2 * W2 * 8;
  const float sc = bx_grad_scale(amax_bits);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int j = (int)(i % 8);
    long r = i / 8;
    const int x2 = (int)(r % W2);
    r /= W2;
    const int y2 = (int)(r % H2);
    r /= H2;
    const int gq = (int)(r % G), b = (int)(r / G);
    const int ch = gq * 8 + j;
    float v = 0.f;
    if (ch < 4 * C) {
      const int c = ch >> 2, yy = 2 * y2 + ((ch >> 1) & 1), xx = 2 * x2 + (ch & 1);
      if (yy < H && xx < W) v = gy[(((long)b * C + c) * H + yy) * W + xx] * sc;
    }
    g[i] = v;
    if (i == 0 && amax_bits) ((float*)amax_bits)[1] = bx_grad_unscale(amax_bits);       // (the weight-gradient reduction's factor: word 1 of the tail)
  }
}
// adjoint of k_bx_pack_in's image part (replicate-pad to even size + pixel-unshuffle): the padded row / column folds onto the last one
__global__ void k_bx_unpack_gin(const float* __restrict__ ga, float* __restrict__ gx, int B, int C, int H, int W, int H2, int W2, int G,
                                const unsigned* __restrict__ amax_bits) {
  const long total = (long)B * C * H * W;
  const float us = bx_grad_unscale(amax_bits);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int xx = (int)(i % W);
    long r = i / W;
    const int yy = (int)(r % H);
    r /= H;
    const int c = (int)(r % C), b = (int)(r / C);
    auto at = [&](int py, int px) {                                    // padded position (py, px) of channel c
      const int ch = c * 4 + (py & 1) * 2 + (px & 1);
      return ga[((((long)b * G + (ch >> 3)) * H2 + (py >> 1)) * W2 + (px >> 1)) * 8 + (ch & 7)];
    };
    float v = at(yy, xx);
    const bool fy = (H & 1) && yy == H - 1, fx = (W & 1) && xx == W - 1;      // the replicated row / column comes back to this pixel
    if (fy) v += at(yy + 1, xx);
    if (fx) v += at(yy, xx + 1);
    if (fy && fx) v += at(yy + 1, xx + 1);
    gx[i] = v * us;
  }
}
// d / d sigma_b = sum over the sigma-map channel (4 C) of g_a0, in a fixed order: BX_SG_SLICES workgroups per image leave the sums of their
// contiguous slices, one workgroup per image adds those up (one workgroup per image walking the whole plane took 147 us at 2 x 384 x 384)
constexpr int BX_SG_SLICES = 64;
__global__ void __launch_bounds__(256) k_bx_sigma_grad(const float* __restrict__ ga, float* __restrict__ part, int C, int H2, int W2, int G) {
  __shared__ float sh[256];
  const int b = blockIdx.y, ch = 4 * C;
  const float* base = ga + (((long)b * G + (ch >> 3)) * H2 * W2) * 8 + (ch & 7);
  const long hw = (long)H2 * W2, i0 = hw * blockIdx.x / BX_SG_SLICES, i1 = hw * (blockIdx.x + 1) / BX_SG_SLICES;
  float acc = 0.f;
  for (long i = i0 + threadIdx.x; i < i1; i += 256) acc += base[i * 8];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[b * BX_SG_SLICES + blockIdx.x] = sh[0];
}
__global__ void __launch_bounds__(BX_SG_SLICES) k_bx_sigma_grad_finish(const float* __restrict__ part, float* __restrict__ gs,
                                                                     const unsigned* __restrict__ amax_bits) {
  float v = part[blockIdx.x * BX_SG_SLICES + threadIdx.x];
#pragma unroll
  for (int o = BX_SG_SLICES / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if (threadIdx.x == 0) gs[blockIdx.x] = v * bx_grad_unscale(amax_bits);
}
// the tail of the backward workspaces: [0] the bits of max |gy| (split-f16 passes), from byte 256 on the sigma gradient's partial sums
static size_t bx_bwd_tail_bytes(int B) { return 256 + (size_t)B * BX_SG_SLICES * sizeof(float); }
static void launch_bx_sigma_grad(const float* g_a0, float* gsigma, char* tail, const unsigned* amax_bits, int B, int in_nc, int H2, int W2, int G0,
                                 hipStream_t s) {
  float* part = (float*)(tail + 256);
  DPX_LAUNCH("k_bx_sigma_grad", k_bx_sigma_grad, dim3(BX_SG_SLICES, B), dim3(256), 0, s, g_a0, part, in_nc, H2, W2, G0);
  DPX_LAUNCH("k_bx_sigma_grad_finish", k_bx_sigma_grad_finish, dim3(B), dim3(BX_SG_SLICES), 0, s, (const float*)part, gsigma, amax_bits);
}

static int bx_cin(int l, int in_nc, int nc) { return l == 0 ? 4 * in_nc + 1 : nc; }
static int bx_cout(int l, int in_nc, int nc, int nb) { return l == nb - 1 ? 4 * in_nc : nc; }
static int groups16(int c) { return 2 * ((c + 15) / 16); }           // channel groups of 8, rounded up to whole 16-channel chunks

}  // namespace dpx

#include "dpx_conv_wino_dev.h"
#ifdef DPX_WN_TRACE
extern "C" int dpx_dbg_bx_trace(unsigned long long* host, int n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(dpx::dpx_bx_trace_buf), sizeof(unsigned long long) * (n < 512 ? n : 512)) == hipSuccess ? 0 : -1;
}
#endif

namespace dpx {
// mode 4 ("split-f16 Winograd"): the first layer (13 -> nc channels: one 16-channel chunk) and layers of more than 64 output channels (their
// 192 accumulators leave the Winograd kernel's pipeline no registers) stay on the direct split-f16 kernel, every other layer runs as F(2x2, 3x3)
static bool bx_layer_is_wino(int mode, int l, int cout) { return mode == 4 && l >= 1 && cout <= 64; }
static size_t bx_layer_bytes_mode(int mode, int l, int in_nc, int nc, int nb) {
  const int cin = bx_cin(l, in_nc, nc), cout = bx_cout(l, in_nc, nc, nb);
  return bx_layer_is_wino(mode, l, cout) ? wn_layer_bytes(cin, cout) : bx_layer_bytes(cin, cout, bx_planes(mode == 4 ? 3 : mode));
}
}  // namespace dpx

using namespace dpx;

// mode: 6 = split-bf16 (fp32-accurate), 3 = split-f16 (fp32-accurate for |operands| < 6e4, half the matrix work), 1 = plain bf16 operands,
// 4 = split-f16 with the layers behind the first one as Winograd F(2x2, 3x3) (dpx_conv_wino_dev.h: 2.25 x fewer matrix instructions)
extern "C" size_t dpx_ffdnet_bf16_packed_bytes(int in_nc, int nc, int nb) {
  size_t n = 0, n4 = 0;
  for (int l = 0; l < nb; ++l) {
    n += bx_layer_bytes(bx_cin(l, in_nc, nc), bx_cout(l, in_nc, nc, nb));
    n4 += bx_layer_bytes_mode(4, l, in_nc, nc, nb);
  }
  return (n > n4 ? n : n4) + 1024;                                   // (the last KB: scratch of the packing kernels)
}

extern "C" int dpx_ffdnet_bf16_pack(void* packed, const float* const* w, const float* const* b, int in_nc, int nc, int nb, int mode,
                                    dpx_stream_t stream) {
  DPX_REQUIRE(packed && w && b && in_nc > 0 && nc > 0 && nb >= 2 && (mode == 6 || mode == 1 || mode == 3 || mode == 4), "dpx_ffdnet_bf16_pack: bad arguments");
  DPX_REQUIRE(nc <= 96 && nc % 16 == 0 && 4 * in_nc <= 96, "dpx_ffdnet_bf16_pack: layers of 16..96 channels (multiples of 16), got %d", nc);
  DPX_REQUIRE(nb <= 64, "dpx_ffdnet_bf16_pack: at most 64 layers");
  char* dst = (char*)packed;
  unsigned* umax = (unsigned*)((char*)packed + dpx_ffdnet_bf16_packed_bytes(in_nc, nc, nb) - 1024);       // one word per layer
  if (mode == 4 && hipMemsetAsync(umax, 0, 64 * sizeof(unsigned), (hipStream_t)stream) != hipSuccess) return DPX_ERR_LAUNCH;
  for (int l = 0; l < nb; ++l) {
    const int cin = bx_cin(l, in_nc, nc), cout = bx_cout(l, in_nc, nc, nb);
    DPX_REQUIRE(w[l] && b[l], "dpx_ffdnet_bf16_pack: layer %d has null weights", l);
    const size_t n = bx_layer_bytes_mode(mode, l, in_nc, nc, nb);
    if (bx_layer_is_wino(mode, l, cout)) {
      DPX_LAUNCH("k_wn_umax", k_wn_umax, dim3(grid_for((long)cin * cout * 16, 256, 256)), dim3(256), 0, (hipStream_t)stream, w[l], cin * cout, umax + l);
      DPX_LAUNCH("k_wn_pack_weights", k_wn_pack_weights, dim3(grid_for((long)(n / 2), 256, 2048)), dim3(256), 0, (hipStream_t)stream, w[l], b[l],
                 (unsigned short*)dst, cin, cout, (const unsigned*)(umax + l));
    } else {
      DPX_LAUNCH("k_bx_pack_weights", k_bx_pack_weights, dim3(grid_for((long)(n / 2), 256, 2048)), dim3(256), 0, (hipStream_t)stream, w[l], b[l],
                 (unsigned short*)dst, cin, cout, mode == 4 ? 3 : mode, 0);
    }
    dst += n;
  }
  return launch_status("dpx_ffdnet_bf16_pack");
}

extern "C" size_t dpx_ffdnet_bf16_ws_bytes(int B, int in_nc, int nc, int H, int W) {
  const size_t H2 = (H + 1) / 2, W2 = (W + 1) / 2, px = (size_t)B * H2 * W2;
  return (px * 8 * groups16(4 * in_nc + 1) + 2 * px * 8 * groups16(nc) + px * 8 * groups16(4 * in_nc)) * sizeof(float);
}

// What dpx_admm_cg_pnp_iter runs in front of the first layer instead of k_zupdate + k_bx_pack_in (identity terms, even H and W: every pixel is one
// slot of the packed tensor): d_i = x + u_i, closed-form terms v_i = prox_i(d_i), u_i = d_i - v_i, the prior's term v = d -- k_zupdate's
// arithmetic -- and d of the prior's term pixel-unshuffled with the sigma plane into the first layer's C8 input.  pred (nullable): the CG's
// `done` word -- launched ahead of the host's look at it (CgSpeculate), the pass does nothing unless the solve has converged.
struct PnpHead {
  const float* x;
  const float* sigma;
  float* a;
  const int* pred;
  int nterms, ext;
  float* v[DPX_MAX_TERMS];
  float* u[DPX_MAX_TERMS];
  const float* lam[DPX_MAX_TERMS];
  float alpha[DPX_MAX_TERMS];
  int prox[DPX_MAX_TERMS];
};
__global__ void k_pnp_head(PnpHead Q, int B, int C, int H, int W, int H2, int W2, int G) {
  if (Q.pred && Q.pred[0] == 0) return;
  const long total = (long)B * G * H2 * W2 * 8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int j = (int)(i % 8);
    long r = i / 8;
    const int x2 = (int)(r % W2);
    r /= W2;
    const int y2 = (int)(r % H2);
    r /= H2;
    const int g = (int)(r % G), b = (int)(r / G);
    const int ch = g * 8 + j;
    float out = 0.f;
    if (ch < 4 * C) {
      const int c = ch >> 2, yy = 2 * y2 + ((ch >> 1) & 1), xx = 2 * x2 + (ch & 1);
      const long off = (((long)b * C + c) * H + yy) * W + xx;
      const float xv = Q.x[off];
      for (int t = 0; t < Q.nterms; ++t) {
        const float d = xv + Q.u[t][off];
        if (t == Q.ext) {
          Q.v[t][off] = d;
          out = d;
        } else {
          const float lam = Q.lam[t] ? Q.lam[t][b] * Q.alpha[t] : 0.f;
          const float vv = prox_eval(Q.prox[t], d, lam);
          Q.v[t][off] = vv;
          Q.u[t][off] = d - vv;
        }
      }
    } else if (ch == 4 * C) {
      out = Q.sigma[b];
    }
    Q.a[i] = out;
  }
}

// What dpx_admm_cg_pnp_iter hangs behind the last layer instead of k_bx_unpack_out (identity terms only: every Psi term acts on x itself):
//   v_new = PixelShuffle(last layer), u_ext = d - v_new (d = terms[ext].v), and -- rho_next non-null -- the NEXT iteration's right-hand side
//   ktb + rho_next sum_i (v_i - u_i) written straight into the CG's start state (r = rhs, x_next = p = 0, flags / counters cleared):
//   k_bx_unpack_out + k_lincomb + k_rhs + k_cgm_start as one launch, the same arithmetic in the same order.
struct PnpTail {
  float* v_new;
  float* u_ext;
  const float* d;
  const float* ktb;        // nullable
  const float* rho_next;   // nullable: no next right-hand side
  float* x_next;
  dpx::CgStartPtrs cg;
  int nterms, ext;
  const float* v[DPX_MAX_TERMS];
  const float* u[DPX_MAX_TERMS];
};
__global__ void k_pnp_tail(const float* __restrict__ o, PnpTail Q, int B, int C, int H, int W, int H2, int W2, int G) {
  const long total = (long)B * C * H * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int xx = (int)(i % W);
    long r = i / W;
    const int yy = (int)(r % H);
    r /= H;
    const int c = (int)(r % C), b = (int)(r / C);
    const int ch = c * 4 + (yy & 1) * 2 + (xx & 1);
    const float vn = o[((((long)b * G + (ch >> 3)) * H2 + (yy >> 1)) * W2 + (xx >> 1)) * 8 + (ch & 7)];
    const float ue = Q.d[i] - vn;
    Q.v_new[i] = vn;
    Q.u_ext[i] = ue;
    if (Q.rho_next) {
      float acc = 0.f;
      for (int t = 0; t < Q.nterms; ++t) acc += (t == Q.ext) ? vn - ue : Q.v[t][i] - Q.u[t][i];
      Q.cg.r[i] = fmaf(Q.rho_next[b], acc, Q.ktb ? Q.ktb[i] : 0.f);
      Q.cg.p[i] = 0.f;
      Q.x_next[i] = 0.f;
    }
  }
  if (Q.rho_next && blockIdx.x == 0 && threadIdx.x == 0) {
    Q.cg.flags[0] = 0;
    Q.cg.flags[1] = -1;
    Q.cg.flags[2] = 0;
    Q.cg.flags[3] = 0;
    Q.cg.counters[0] = 0u;
    Q.cg.counters[1] = 0u;
  }
}

static int ffdnet_forward_bf16_impl(const float* x, float* y, const float* sigma, const void* packed, int in_nc, int nc, int nb, int mode, int B, int H,
                                    int W, void* ws, dpx_stream_t stream, const PnpTail* tail, bool packed_in = false);
extern "C" int dpx_ffdnet_forward_bf16(const float* x, float* y, const float* sigma, const void* packed, int in_nc, int nc, int nb, int mode,
                                       int B, int H, int W, void* ws, dpx_stream_t stream) {
  return ffdnet_forward_bf16_impl(x, y, sigma, packed, in_nc, nc, nb, mode, B, H, W, ws, stream, nullptr);
}
static int ffdnet_forward_bf16_impl(const float* x, float* y, const float* sigma, const void* packed, int in_nc, int nc, int nb, int mode, int B, int H,
                                    int W, void* ws, dpx_stream_t stream, const PnpTail* tail, bool packed_in) {
  DPX_REQUIRE(x && y && sigma && packed && ws, "dpx_ffdnet_forward_bf16: null pointer");
  DPX_REQUIRE(B > 0 && H > 0 && W > 0 && in_nc > 0 && nb >= 2 && nc % 16 == 0 && nc <= 96 && 4 * in_nc <= 96 &&
                  (mode == 6 || mode == 1 || mode == 3 || mode == 4),
              "dpx_ffdnet_forward_bf16: unsupported configuration (in_nc=%d nc=%d nb=%d mode=%d)", in_nc, nc, nb, mode);
  hipStream_t s = (hipStream_t)stream;
  const int H2 = (H + 1) / 2, W2 = (W + 1) / 2;
  DPX_REQUIRE((size_t)2 * H2 * W2 * 8 < ((size_t)1 << 32), "dpx_ffdnet_forward_bf16: plane %dx%d too large", H, W);
  const size_t px = (size_t)B * H2 * W2;
  const int G0 = groups16(4 * in_nc + 1), Gc = groups16(nc), GL = groups16(4 * in_nc);
  float* a0 = (float*)ws;
  float* bufA = a0 + px * 8 * G0;
  float* bufB = bufA + px * 8 * Gc;
  float* last = bufB + px * 8 * Gc;
  // split-f16 inference with the activations travelling between the layers as pre-split operand planes (P8, bit-identical results): knob
  // ffdnet_presplit = 1, OFF by default.  Measured, round 4 (8x3x1024^2 colour / 32x1x320^2 gray, alternating runs on one box): 10.16 /
  // 3.00 ms with it, 9.93 - 10.14 / 2.85 ms without -- no split pass, no landing buffer and one barrier less per chunk buy nothing
  // (the kernel runs at the matrix pipe's power-limited rate), and two 8-byte stores per lane instead of one
  // 16-byte store in the epilogue cost a little.
  const bool p8 = mode == 3 && tune(TUNE_FFDNET_PRESPLIT) == 1;
  if (!packed_in)              // (packed_in: k_pnp_head has filled a0 -- plain C8, never with ffdnet_presplit)
    DPX_LAUNCH("k_bx_pack_in", k_bx_pack_in, dim3(grid_for((long)(px * 8 * G0), 256, 8192)), dim3(256), 0, s, x, sigma, a0, B, in_nc, H, W, H2, W2, G0,
               p8 ? 1 : 0);
  const char* wl = (const char*)packed;
  const float* cur = a0;
  int gin = G0;
  for (int l = 0; l < nb; ++l) {
    const int cin = bx_cin(l, in_nc, nc), cout = bx_cout(l, in_nc, nc, nb);
    const bool lastl = l == nb - 1;
    float* dst = lastl ? last : ((l & 1) ? bufB : bufA);
    const int gout = lastl ? GL : Gc;
    if (p8) launch_bx_p8_mt((cout + 31) / 32, !lastl, true, !lastl, cur, dst, wl, gin, gout, B, H2, W2, s);
    else if (bx_layer_is_wino(mode, l, cout)) launch_wino_mt((cout + 31) / 32, !lastl, cur, dst, wl, gin, gout, B, H2, W2, s);
    else if (mode == 1) launch_bx_mt<1>((cout + 31) / 32, !lastl, cur, dst, wl, gin, gout, B, H2, W2, s);
    else if (mode == 3 || mode == 4) launch_bx_mt<3>((cout + 31) / 32, !lastl, cur, dst, wl, gin, gout, B, H2, W2, s);
    else launch_bx_mt<6>((cout + 31) / 32, !lastl, cur, dst, wl, gin, gout, B, H2, W2, s);
    wl += bx_layer_bytes_mode(mode, l, in_nc, nc, nb);
    cur = dst;
    gin = gout;
  }
  if (tail)
    DPX_LAUNCH("k_pnp_tail", k_pnp_tail, dim3(grid_for((long)B * in_nc * H * W, 256, 8192)), dim3(256), 0, s, (const float*)last, *tail, B, in_nc, H, W,
               H2, W2, GL);
  else
    DPX_LAUNCH("k_bx_unpack_out", k_bx_unpack_out, dim3(grid_for((long)B * in_nc * H * W, 256, 8192)), dim3(256), 0, s, last, y, B, in_nc, H, W, H2,
               W2, GL);
  return launch_status("dpx_ffdnet_forward_bf16");
}

// ---- reverse mode on the split kernels (frozen weights: gradients w.r.t. the image and sigma; the reference differentiates
// network_ffdnet.py:54-68 with autograd).  Forward pass that KEEPS every layer's output (C8, the backward pass's ReLU masks), weights
// of the backward-data layers (flipped / transposed, split like the forward ones, no bias), and the backward pass: the same
// k_conv3x3_bf16 on those weights with [a_l > 0] applied in the epilogue.  Gradients can be tiny (1e-7 of an MSE loss): the backward
// pass always runs split-bf16 (mode 6, fp32's range); the forward pass may be any fp32-accurate mode.
extern "C" size_t dpx_ffdnet_bf16_acts_bytes(int B, int in_nc, int nc, int nb, int H, int W) {
  const size_t H2 = (H + 1) / 2, W2 = (W + 1) / 2, px = (size_t)B * H2 * W2;
  return (px * 8 * groups16(4 * in_nc + 1) + (size_t)(nb - 1) * px * 8 * groups16(nc) + px * 8 * groups16(4 * in_nc)) * sizeof(float);
}

extern "C" int dpx_ffdnet_forward_bf16_save(const float* x, float* y, const float* sigma, const void* packed, int in_nc, int nc, int nb, int mode,
                                            int B, int H, int W, void* acts, dpx_stream_t stream) {
  DPX_REQUIRE(x && y && sigma && packed && acts, "dpx_ffdnet_forward_bf16_save: null pointer");
  DPX_REQUIRE(B > 0 && H > 0 && W > 0 && in_nc > 0 && nb >= 2 && nc % 16 == 0 && nc <= 96 && 4 * in_nc <= 96 && (mode == 6 || mode == 3),
              "dpx_ffdnet_forward_bf16_save: unsupported configuration (in_nc=%d nc=%d nb=%d mode=%d)", in_nc, nc, nb, mode);
  hipStream_t s = (hipStream_t)stream;
  const int H2 = (H + 1) / 2, W2 = (W + 1) / 2;
  DPX_REQUIRE((size_t)2 * H2 * W2 * 8 < ((size_t)1 << 32), "dpx_ffdnet_forward_bf16_save: plane %dx%d too large", H, W);
  const size_t px = (size_t)B * H2 * W2;
  const int G0 = groups16(4 * in_nc + 1), Gc = groups16(nc), GL = groups16(4 * in_nc);
  float* a0 = (float*)acts;
  float* hidden = a0 + px * 8 * G0;                                   // layer l's output (l < nb - 1) at hidden + l px 8 Gc
  float* last = hidden + (size_t)(nb - 1) * px * 8 * Gc;
  DPX_LAUNCH("k_bx_pack_in", k_bx_pack_in, dim3(grid_for((long)(px * 8 * G0), 256, 8192)), dim3(256), 0, s, x, sigma, a0, B, in_nc, H, W, H2, W2, G0, 0);
  const char* wl = (const char*)packed;
  const float* cur = a0;
  int gin = G0;
  for (int l = 0; l < nb; ++l) {
    const int cin = bx_cin(l, in_nc, nc), cout = bx_cout(l, in_nc, nc, nb);
    const bool lastl = l == nb - 1;
    float* dst = lastl ? last : hidden + (size_t)l * px * 8 * Gc;
    const int gout = lastl ? GL : Gc;
    if (mode == 3) launch_bx_mt<3>((cout + 31) / 32, !lastl, cur, dst, wl, gin, gout, B, H2, W2, s);
    else launch_bx_mt<6>((cout + 31) / 32, !lastl, cur, dst, wl, gin, gout, B, H2, W2, s);
    wl += bx_layer_bytes(cin, cout, bx_planes(mode));
    cur = dst;
    gin = gout;
  }
  DPX_LAUNCH("k_bx_unpack_out", k_bx_unpack_out, dim3(grid_for((long)B * in_nc * H * W, 256, 8192)), dim3(256), 0, s, last, y, B, in_nc, H, W, H2,
             W2, GL);
  return launch_status("dpx_ffdnet_forward_bf16_save");
}

extern "C" size_t dpx_ffdnet_bf16_packed_transposed_bytes(int in_nc, int nc, int nb) {
  size_t n = 0;
  for (int l = 0; l < nb; ++l) n += bx_layer_bytes(bx_cout(l, in_nc, nc, nb), bx_cin(l, in_nc, nc));
  return n + 1024;
}

// packed_T: the nb backward-data layers in FORWARD order (layer l: bx_cout(l) -> bx_cin(l) channels), zero bias; mode 6: split-bf16 planes,
// mode 3: split-f16 planes (a weight beyond the binary16 range trips dpx_ffdnet_f16_overflow)
extern "C" int dpx_ffdnet_bf16_pack_transposed(void* packed_T, const float* const* w, int in_nc, int nc, int nb, int mode, dpx_stream_t stream) {
  DPX_REQUIRE(packed_T && w && in_nc > 0 && nc > 0 && nb >= 2 && (mode == 6 || mode == 3), "dpx_ffdnet_bf16_pack_transposed: bad arguments");
  DPX_REQUIRE(nc <= 96 && nc % 16 == 0 && 4 * in_nc <= 96, "dpx_ffdnet_bf16_pack_transposed: layers of 16..96 channels (multiples of 16), got %d", nc);
  char* dst = (char*)packed_T;
  for (int l = 0; l < nb; ++l) {
    const int cin_t = bx_cout(l, in_nc, nc, nb), cout_t = bx_cin(l, in_nc, nc);
    DPX_REQUIRE(w[l], "dpx_ffdnet_bf16_pack_transposed: layer %d has null weights", l);
    const size_t n = bx_layer_bytes(cin_t, cout_t, bx_planes(mode));
    DPX_LAUNCH("k_bx_pack_weights", k_bx_pack_weights, dim3(grid_for((long)(n / 2), 256, 2048)), dim3(256), 0, (hipStream_t)stream, w[l],
               (const float*)nullptr, (unsigned short*)dst, cin_t, cout_t, mode, 1);
    dst += n;
  }
  return launch_status("dpx_ffdnet_bf16_pack_transposed");
}

extern "C" size_t dpx_ffdnet_bf16_bwd_ws_bytes(int B, int in_nc, int nc, int H, int W) {
  const size_t H2 = (H + 1) / 2, W2 = (W + 1) / 2, px = (size_t)B * H2 * W2;
  return (px * 8 * groups16(4 * in_nc) + 2 * px * 8 * groups16(nc) + px * 8 * groups16(4 * in_nc + 1)) * sizeof(float) + bx_bwd_tail_bytes(B);
}
// the planes of a backward workspace (everything in front of its tail)
static size_t bx_bwd_plane_bytes(int B, int in_nc, int nc, int H, int W) { return dpx_ffdnet_bf16_bwd_ws_bytes(B, in_nc, nc, H, W) - bx_bwd_tail_bytes(B); }
// the backward-data layer of forward layer l; mode 3: in the split-f16 arithmetic (its operands -- the scaled gradients -- are watched by the range trap)
static void launch_bx_bwd_layer(int mode, int mt, const float* cur, float* dst, const char* wpk, int gin, int gout, int B, int H2, int W2, hipStream_t s,
                                const float* mask) {
  if (mode == 3) launch_bx_mt<3>(mt, false, cur, dst, wpk, gin, gout, B, H2, W2, s, mask);
  else launch_bx_mt<6>(mt, false, cur, dst, wpk, gin, gout, B, H2, W2, s, mask);
}
// the first launches of a backward pass: mode 3 -- the bits of max |gy| into the workspace tail's first word; then the scaled gradient in the C8 layout
static const unsigned* launch_bx_pack_gout(int mode, const float* gy, float* g_last, char* tail, int B, int in_nc, int H, int W, int H2, int W2, int GL,
                                           hipStream_t s) {
  unsigned* amax_bits = nullptr;
  if (mode == 3) {
    amax_bits = (unsigned*)tail;
    hipMemsetAsync(amax_bits, 0, sizeof(unsigned), s);
    const long n = (long)B * in_nc * H * W;
    DPX_LAUNCH("k_bx_absmax", k_bx_absmax, dim3(grid_for((n + 3) / 4, 256, 1024)), dim3(256), 0, s, gy, n, amax_bits);
  }
  const size_t px = (size_t)B * H2 * W2;
  DPX_LAUNCH("k_bx_pack_gout", k_bx_pack_gout, dim3(grid_for((long)(px * 8 * GL), 256, 8192)), dim3(256), 0, s, gy, g_last, B, in_nc, H, W, H2, W2, GL,
             (const unsigned*)amax_bits);
  return amax_bits;
}

// gx, gsigma: either may be NULL.  acts: dpx_ffdnet_forward_bf16_save's buffer.  mode: the arithmetic packed_T was packed for (6: split-bf16,
// any range; 3: split-f16 on gradients scaled by a power of two, half the matrix work -- dpx_ffdnet_f16_overflow says afterwards whether it held).
extern "C" int dpx_ffdnet_backward_bf16(const float* gy, float* gx, float* gsigma, const void* packed_T, const void* acts, int in_nc, int nc, int nb,
                                        int mode, int B, int H, int W, void* ws, dpx_stream_t stream) {
  DPX_REQUIRE(gy && packed_T && acts && ws && (gx || gsigma), "dpx_ffdnet_backward_bf16: null pointer");
  DPX_REQUIRE(B > 0 && H > 0 && W > 0 && in_nc > 0 && nb >= 2 && nc % 16 == 0 && nc <= 96 && 4 * in_nc <= 96 && (mode == 6 || mode == 3),
              "dpx_ffdnet_backward_bf16: unsupported configuration (in_nc=%d nc=%d nb=%d mode=%d)", in_nc, nc, nb, mode);
  hipStream_t s = (hipStream_t)stream;
  const int H2 = (H + 1) / 2, W2 = (W + 1) / 2;
  const size_t px = (size_t)B * H2 * W2;
  const int G0 = groups16(4 * in_nc + 1), Gc = groups16(nc), GL = groups16(4 * in_nc);
  const float* hidden = (const float*)acts + px * 8 * G0;
  float* g_last = (float*)ws;
  float* gA = g_last + px * 8 * GL;
  float* gB = gA + px * 8 * Gc;
  float* g_a0 = gB + px * 8 * Gc;
  char* tail = (char*)ws + bx_bwd_plane_bytes(B, in_nc, nc, H, W);
  const unsigned* amax_bits = launch_bx_pack_gout(mode, gy, g_last, tail, B, in_nc, H, W, H2, W2, GL, s);
  size_t off[64];
  DPX_REQUIRE(nb <= 64, "dpx_ffdnet_backward_bf16: at most 64 layers");
  size_t o = 0;
  for (int l = 0; l < nb; ++l) { off[l] = o; o += bx_layer_bytes(bx_cout(l, in_nc, nc, nb), bx_cin(l, in_nc, nc), bx_planes(mode)); }
  const float* cur = g_last;
  int gin = GL;
  for (int l = nb - 1; l >= 0; --l) {
    const int cout_t = bx_cin(l, in_nc, nc);
    float* dst = (l == 0) ? g_a0 : (((nb - 1 - l) & 1) ? gB : gA);
    const int gout = (l == 0) ? G0 : Gc;
    // the output of backward layer l is the gradient w.r.t. a_l, the (post-ReLU) output of forward layer l - 1: stored already
    // multiplied by [a_l > 0], ready to be the next layer's plain input
    const float* mask = (l >= 1) ? hidden + (size_t)(l - 1) * px * 8 * Gc : nullptr;
    launch_bx_bwd_layer(mode, (cout_t + 31) / 32, cur, dst, (const char*)packed_T + off[l], gin, gout, B, H2, W2, s, mask);
    cur = dst;
    gin = gout;
  }
  if (gx)
    DPX_LAUNCH("k_bx_unpack_gin", k_bx_unpack_gin, dim3(grid_for((long)B * in_nc * H * W, 256, 8192)), dim3(256), 0, s, (const float*)g_a0, gx, B, in_nc, H,
               W, H2, W2, G0, amax_bits);
  if (gsigma) launch_bx_sigma_grad(g_a0, gsigma, tail, amax_bits, B, in_nc, H2, W2, G0, s);
  return launch_status("dpx_ffdnet_backward_bf16");
}

// ---- the same backward pass WITH the weight / bias gradients (deep_prior(trainable=True) on the split kernels) -------------------------
// Forward and backward-data stay on the split kernels (C8 planes); the weight-gradient GEMM of a layer reads the same planes -- the gradient
// w.r.t. its pre-activation output and its saved input -- in the arithmetic of the backward pass (k_wgrad_c8, dpx_wgrad_c8.hip).  (Round 4 fed
// planar copies of both operands to a register kernel, k_wgrad_bf16x3: 338 us per 96 -> 96 layer at 2 x 384 x 384 against 157 now.)
namespace dpx {
size_t wgrad_c8_ws_floats(int cout_max, int cin_max);                // dpx_wgrad_c8.hip
void launch_wgrad_c8(int mode, const float* G, const float* A, float* gw, float* gb, int Cout, int Cin_w, int Gg, int Ga, int B, int H, int W,
                     float* ws, const float* mul, hipStream_t s);
}

extern "C" size_t dpx_ffdnet_bf16_bwd_w_ws_bytes(int B, int in_nc, int nc, int H, int W) {
  return dpx_ffdnet_bf16_bwd_ws_bytes(B, in_nc, nc, H, W) +
         wgrad_c8_ws_floats(nc > 4 * in_nc ? nc : 4 * in_nc, nc > 4 * in_nc + 1 ? nc : 4 * in_nc + 1) * sizeof(float);
}

// gw[l] [cout_l][cin_l][9], gb[l] [cout_l] (entries may be NULL: that layer's gradients are not wanted); gx, gsigma may be NULL
extern "C" int dpx_ffdnet_backward_bf16_w(const float* gy, float* gx, float* gsigma, float* const* gw, float* const* gb, const void* packed_T,
                                          const void* acts, int in_nc, int nc, int nb, int mode, int B, int H, int W, void* ws, dpx_stream_t stream) {
  DPX_REQUIRE(gy && packed_T && acts && ws && gw && gb, "dpx_ffdnet_backward_bf16_w: null pointer");
  DPX_REQUIRE(B > 0 && H > 0 && W > 0 && in_nc > 0 && nb >= 2 && nb <= 64 && nc % 16 == 0 && nc <= 96 && 4 * in_nc <= 96 && (mode == 6 || mode == 3),
              "dpx_ffdnet_backward_bf16_w: unsupported configuration (in_nc=%d nc=%d nb=%d mode=%d)", in_nc, nc, nb, mode);
  hipStream_t s = (hipStream_t)stream;
  const int H2 = (H + 1) / 2, W2 = (W + 1) / 2;
  DPX_REQUIRE((size_t)12 * H2 * W2 * 32 < ((size_t)1 << 32), "dpx_ffdnet_backward_bf16_w: plane %dx%d too large", H, W);
  const size_t px = (size_t)B * H2 * W2;
  const int G0 = groups16(4 * in_nc + 1), Gc = groups16(nc), GL = groups16(4 * in_nc);
  const float* a0 = (const float*)acts;
  const float* hidden = a0 + px * 8 * G0;
  float* g_last = (float*)ws;
  float* gA = g_last + px * 8 * GL;
  float* gB = gA + px * 8 * Gc;
  float* g_a0 = gB + px * 8 * Gc;
  char* tail = (char*)ws + bx_bwd_plane_bytes(B, in_nc, nc, H, W);
  float* wg_ws = (float*)((char*)ws + dpx_ffdnet_bf16_bwd_ws_bytes(B, in_nc, nc, H, W));
  const unsigned* amax_bits = launch_bx_pack_gout(mode, gy, g_last, tail, B, in_nc, H, W, H2, W2, GL, s);
  size_t off[64];
  size_t o = 0;
  for (int l = 0; l < nb; ++l) { off[l] = o; o += bx_layer_bytes(bx_cout(l, in_nc, nc, nb), bx_cin(l, in_nc, nc), bx_planes(mode)); }
  const float* cur = g_last;
  int gin = GL;
  const bool need_data = gx || gsigma;
  for (int l = nb - 1; l >= 0; --l) {
    const int cout_f = bx_cout(l, in_nc, nc, nb), cin_f = bx_cin(l, in_nc, nc);     // the FORWARD layer's channel counts
    if (gw[l]) {
      DPX_REQUIRE(gb[l], "dpx_ffdnet_backward_bf16_w: weight and bias gradients of layer %d come together", l);
      // `cur`: the gradient w.r.t. forward layer l's pre-activation output (the ReLU mask was applied by backward layer l + 1's epilogue)
      const float* a_l = (l == 0) ? a0 : hidden + (size_t)(l - 1) * px * 8 * Gc;
      const int ga = (l == 0) ? G0 : Gc;
      // (the sums leave multiplied by the inverse gradient scale of a split-f16 pass: word 1 of the tail)
      launch_wgrad_c8(mode, cur, a_l, gw[l], gb[l], cout_f, cin_f, gin, ga, B, H2, W2, wg_ws, amax_bits ? (const float*)amax_bits + 1 : nullptr, s);
    }
    if (l == 0 && !need_data) break;
    float* dst = (l == 0) ? g_a0 : (((nb - 1 - l) & 1) ? gB : gA);
    const int gout = (l == 0) ? G0 : Gc;
    const float* mask = (l >= 1) ? hidden + (size_t)(l - 1) * px * 8 * Gc : nullptr;
    launch_bx_bwd_layer(mode, (cin_f + 31) / 32, cur, dst, (const char*)packed_T + off[l], gin, gout, B, H2, W2, s, mask);
    cur = dst;
    gin = gout;
  }
  if (gx)
    DPX_LAUNCH("k_bx_unpack_gin", k_bx_unpack_gin, dim3(grid_for((long)B * in_nc * H * W, 256, 8192)), dim3(256), 0, s, (const float*)g_a0, gx, B, in_nc, H,
               W, H2, W2, G0, amax_bits);
  if (gsigma) launch_bx_sigma_grad(g_a0, gsigma, tail, amax_bits, B, in_nc, H2, W2, G0, s);
  return launch_status("dpx_ffdnet_backward_bf16_w");
}

// ---------------------------------------------------------------------------------------------------------------------
// One plug-and-play ADMM iteration (config 3) without returning to the host language: algo/admm.py:49-59 with a deep_prior
// z-update (proxfn/pnp/prior.py:73-86) --
//   rhs = rho sum_i K_i^T (v_i - u_i)  ->  x = Fourier solve (+ fp64 data spectrum)  ->  closed-form terms: v_i, u_i;  the
//   prior's term `ext`: d = x + u (left in terms[ext].v by the z stage), v = FFDNet(d, sigma), u = d - v.
// The denoiser runs in `mode` 6 (split-bf16) / 1 (bf16) on dpx_ffdnet_forward_bf16 or 0 on the f32-input kernel; a gray network
// (in_nc = 1) sees every band as an image ([B, C, H, W] -> [B C, 1, H, W]; sigma then has B C entries).  v_new: where the
// denoised image goes (the caller swaps it with terms[ext].v for the next iteration).
// ---------------------------------------------------------------------------------------------------------------------
extern "C" int dpx_admm_pnp_iter(float* x, float* rhs, const dpx_term* terms, int nterms, int ext, float* v_new, const float* rho,
                                 const float* sigma, const void* spec_add, const void* dd, float eps, const void* packed, int in_nc, int nc,
                                 int nb, int mode, int B, int C, int H, int W, const void* table, void* spectrum_ws, void* ffd_ws,
                                 dpx_stream_t stream) {
  DPX_REQUIRE(x && rhs && terms && v_new && rho && sigma && dd && packed && table && spectrum_ws && ffd_ws, "dpx_admm_pnp_iter: null pointer");
  DPX_REQUIRE(nterms >= 1 && nterms <= DPX_MAX_TERMS && ext >= 0 && ext < nterms && terms[ext].linop == DPX_LIN_IDENTITY,
              "dpx_admm_pnp_iter: the prior must be a term on x itself");
  DPX_REQUIRE((in_nc == C) || (in_nc == 1), "dpx_admm_pnp_iter: a %d-channel network on %d-channel images", in_nc, C);
  int rc = dpx_admm_rhs(rhs, nullptr, rho, terms, nterms, B, C, H, W, stream);
  if (rc) return rc;
  rc = dpx_fourier_solve(rhs, x, spec_add, dd, rho, eps, B, C, H, W, table, spectrum_ws, stream);
  if (rc) return rc;
  rc = dpx_admm_zupdate(x, terms, nterms, B, C, H, W, stream);
  if (rc) return rc;
  const float* d = terms[ext].v;
  const int Bn = in_nc == C ? B : B * C;
  if (mode == 0) rc = dpx_ffdnet_forward(d, v_new, sigma, packed, in_nc, nc, nb, Bn, H, W, ffd_ws, stream);
  else rc = dpx_ffdnet_forward_bf16(d, v_new, sigma, packed, in_nc, nc, nb, mode, Bn, H, W, ffd_ws, stream);
  if (rc) return rc;
  const float* xs[2] = {d, v_new};
  const float cf[2] = {1.f, -1.f};
  return dpx_lincomb(terms[ext].u, 2, xs, cf, nullptr, B, (long)C * H * W, stream);     // u = d - v
}

// ---------------------------------------------------------------------------------------------------------------------
// One plug-and-play iteration whose x-update is the masked-Fourier CG solve (config 4: ADMM / LinearizedADMM on compressed-sensing MRI,
// algo/admm.py:49-59 / 78-100 with least_squares.solve_cg, proxfn/sum_square.py:158-197) without returning to the host language:
//   rhs = Ktb + rho sum_i (v_i - u_i)  ->  x = dpx_cg_masked_fft(rhs)  ->  closed-form terms: v_i, u_i;  the prior's term `ext`:
//   d = x + u (left in terms[ext].v by the z stage), v = FFDNet(d, sigma), u = d - v.
// Between the CG solve's last look at its stop flag and the denoiser's first launch the host runs this function's few lines instead of
// the host language's operator layers (44 us of idle stream per iteration of the 4 x 1 x 320^2 shard, profiles/r5_c4_timeline.txt).
// Single-channel images (C = 1: the fused CG's planes).  Returns the CG exit iteration (>= 0) or a negative status.
// ---------------------------------------------------------------------------------------------------------------------
extern "C" int dpx_admm_cg_pnp_iter(float* x, float* rhs, const float* ktb, const dpx_term* terms, int nterms, int ext, float* v_new, const float* rho,
                                    const float* sigma, const float* mask, int mask_images, float n_identity, float rtol, int max_iters,
                                    const void* packed, int in_nc, int nc, int nb, int mode, int B, int H, int W, const void* table, void* cg_ws,
                                    void* ffd_ws, const float* rho_next, float* x_next, int rhs_ready, int cg_hint, dpx_stream_t stream) {
  DPX_REQUIRE(x && rhs && terms && v_new && rho && sigma && mask && packed && table && cg_ws && ffd_ws, "dpx_admm_cg_pnp_iter: null pointer");
  DPX_REQUIRE(nterms >= 1 && nterms <= DPX_MAX_TERMS && ext >= 0 && ext < nterms && terms[ext].linop == DPX_LIN_IDENTITY,
              "dpx_admm_cg_pnp_iter: the prior must be a term on x itself");
  DPX_REQUIRE(in_nc == 1, "dpx_admm_cg_pnp_iter: a %d-channel network on single-channel images", in_nc);
  // the folded tail (and with it a prepared next right-hand side) needs the split kernels' C8 output, identity terms and the fused CG branch
  bool fold = mode != 0 && dpx::cg_masked_fft_is_fused(B) && tune(TUNE_PNP_CG_NO_FOLD) == 0;
  for (int t = 0; t < nterms; ++t) fold = fold && terms[t].linop == DPX_LIN_IDENTITY && terms[t].v && terms[t].u && !terms[t].u_out;
  DPX_REQUIRE(!rhs_ready || fold, "dpx_admm_cg_pnp_iter: rhs_ready without the folded tail (see dpx_admm_cg_pnp_iter_folds)");
  DPX_REQUIRE(!rho_next || (x_next && x_next != x), "dpx_admm_cg_pnp_iter: rho_next needs a second iterate buffer");
  int rc;
  if (!rhs_ready) {
    rc = dpx_admm_rhs(rhs, ktb, rho, terms, nterms, B, 1, H, W, stream);
    if (rc) return rc;
  }
  // the folded head (z / dual stage + the first layer's input as one pass) additionally needs even planes and the plain C8 input
  const bool fold_head = fold && H % 2 == 0 && W % 2 == 0 && !(mode == 3 && tune(TUNE_FFDNET_PRESPLIT) == 1);
  struct HeadCtx {
    PnpHead Q;
    int B, H, W, G0;
  } hc;
  if (fold_head) {
    hc.B = B; hc.H = H; hc.W = W; hc.G0 = groups16(4 * in_nc + 1);
    hc.Q.x = x; hc.Q.sigma = sigma; hc.Q.a = (float*)ffd_ws; hc.Q.pred = nullptr; hc.Q.nterms = nterms; hc.Q.ext = ext;
    for (int t = 0; t < DPX_MAX_TERMS; ++t) {
      const bool on = t < nterms;
      hc.Q.v[t] = on ? terms[t].v : nullptr;
      hc.Q.u[t] = on ? terms[t].u : nullptr;
      hc.Q.lam[t] = on ? terms[t].lam : nullptr;
      hc.Q.alpha[t] = on ? terms[t].alpha : 0.f;
      hc.Q.prox[t] = on ? terms[t].prox : 0;
    }
  }
  auto launch_head = +[](void* ctx, const int* pred, dpx_stream_t st) -> int {
    HeadCtx& h = *(HeadCtx*)ctx;
    PnpHead Q = h.Q;
    Q.pred = pred;
    const int H2 = h.H / 2, W2 = h.W / 2;
    DPX_LAUNCH("k_pnp_head", k_pnp_head, dim3(grid_for((long)h.B * h.G0 * H2 * W2 * 8, 256, 8192)), dim3(256), 0, (hipStream_t)st, Q, h.B, 1, h.H, h.W, H2, W2,
               h.G0);
    return launch_status("dpx_admm_cg_pnp_iter");
  };
  // (the head goes into the stream right behind the stop test of the iteration the previous solve ended at, predicated on that test: the
  //  host's look at the flag and its next launches overlap with it instead of leaving the stream idle -- knob pnp_cg_no_fold = 2: off)
  dpx::CgSpeculate spec{launch_head, &hc};
  spec.hint = cg_hint;
  const bool speculate = fold_head && tune(TUNE_PNP_CG_NO_FOLD) != 2;
  const int n_cg = dpx::cg_masked_fft_run(x, rhs, mask, mask_images, rho, n_identity, rtol, max_iters, B, H, W, table, cg_ws, rhs_ready != 0,
                                          speculate ? &spec : nullptr, stream);
  if (n_cg < 0) return n_cg;
  if (fold_head) {
    if (!spec.valid) {
      rc = launch_head(&hc, nullptr, stream);
      if (rc) return rc;
    }
  } else {
    rc = dpx_admm_zupdate(x, terms, nterms, B, 1, H, W, stream);
    if (rc) return rc;
  }
  const float* d = terms[ext].v;
  if (fold) {
    PnpTail Q;
    Q.v_new = v_new;
    Q.u_ext = terms[ext].u;
    Q.d = d;
    Q.ktb = ktb;
    Q.rho_next = rho_next;
    Q.x_next = x_next;
    Q.cg = dpx::cg_masked_fft_start_ptrs(cg_ws, B, H, W, mask_images);
    Q.nterms = nterms;
    Q.ext = ext;
    for (int t = 0; t < DPX_MAX_TERMS; ++t) {
      Q.v[t] = t < nterms ? terms[t].v : nullptr;
      Q.u[t] = t < nterms ? terms[t].u : nullptr;
    }
    rc = ffdnet_forward_bf16_impl(d, v_new, sigma, packed, in_nc, nc, nb, mode, B, H, W, ffd_ws, stream, &Q, fold_head);
    return rc ? rc : n_cg;
  }
  DPX_REQUIRE(!rho_next, "dpx_admm_cg_pnp_iter: rho_next without the folded tail (see dpx_admm_cg_pnp_iter_folds)");
  if (mode == 0) rc = dpx_ffdnet_forward(d, v_new, sigma, packed, in_nc, nc, nb, B, H, W, ffd_ws, stream);
  else rc = dpx_ffdnet_forward_bf16(d, v_new, sigma, packed, in_nc, nc, nb, mode, B, H, W, ffd_ws, stream);
  if (rc) return rc;
  const float* xs[2] = {d, v_new};
  const float cf[2] = {1.f, -1.f};
  rc = dpx_lincomb(terms[ext].u, 2, xs, cf, nullptr, B, (long)H * W, stream);     // u = d - v
  return rc ? rc : n_cg;
}
// 1 when dpx_admm_cg_pnp_iter would fold its tail for this (mode, B) -- only then may a caller pass rho_next / rhs_ready
extern "C" int dpx_admm_cg_pnp_iter_folds(int mode, int B) {
  return mode != 0 && dpx::cg_masked_fft_is_fused(B) && tune(TUNE_PNP_CG_NO_FOLD) == 0 ? 1 : 0;
}

// 1 if a split-f16 layer (mode 3) has met an operand outside the binary16 range since the last reset (results of that call are then
// invalid: rerun in mode 6).  Synchronises the device.
extern "C" int dpx_ffdnet_f16_overflow(int reset) {
  unsigned v = 0u;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(dpx::g_f16_overflow), sizeof(v)) != hipSuccess) return DPX_ERR_LAUNCH;
  if (reset && v) {
    const unsigned z = 0u;
    hipMemcpyToSymbol(HIP_SYMBOL(dpx::g_f16_overflow), &z, sizeof(z));
  }
  return v ? 1 : 0;
}
