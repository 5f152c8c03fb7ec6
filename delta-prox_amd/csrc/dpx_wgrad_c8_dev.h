// Weight / bias gradients of a 3x3 layer straight from the C8 planes the split kernels read and write (included by dpx_conv_bf16.hip).
//
//   dW[co][ci][dy][dx] = sum_{b, y, x} G[b][co][y][x] * A[b][ci][y + dy - 1][x + dx - 1]   (zero outside the image),   db[co] = sum G[b][co][y][x]
//
// a GEMM whose K axis is the PIXELS: v_mfma_f32_32x32x16_{f16,bf16} wants, per lane, 8 consecutive K of ONE channel, and the C8 layout
// [B][C/8][H][W][8] keeps 8 CHANNELS of one pixel together.  Round 4's kernel (k_wgrad_bf16x3) therefore read planar copies of both
// operands (k_bx_c8_to_planar: two extra plane passes per layer), every wave split its own operands (each element of G was split by three
// waves, each of A by three), and it ran one wave per SIMD at 48 % of its roofline.  Here:
//   * a workgroup (8 waves, one per CU, persistent) owns ALL MT x NT x 9 accumulator tiles of the layer -- 81 tiles of 32 x 32 at 96 -> 96
//     channels, 10 - 11 per wave -- and walks DOWN 32-pixel column strips (K = 32 pixels per step); every element of G and A is fetched
//     once per workgroup (LDS-DMA of whole 1 KB group rows, nothing through registers), split ONCE, and transposed on its way into the
//     operand planes: a thread takes a pixel PAIR of one 8-channel group and writes one dword (two pixels of one channel) per channel and
//     plane -- planes [plane][channel][40 pixels] of 16-bit elements, 80 bytes between channels (conflict-free 16-byte reads);
//   * the operand of tap (dy, dx) is the input row y + dy - 1 (a ring of three rows in LDS: one new row per step) read at pixels + dx:
//     ten pixels per lane, the three horizontal shifts by v_alignbit;
//   * MODE 3 (the split-f16 backward pass: G arrives scaled, dpx_conv_bf16.hip "gradient scale"): g = gh + gl' / 2^11, a = ah + al' / 2^11,
//     three products into ONE accumulator -- gh ah + (gh 2^-6)(al' 2^-5) + (gl' 2^-6)(ah 2^-5): the cross terms' factor 2^-11 is spread over
//     both operands by packed multiplications (exact above the subnormal range; below it the error is 2^-25 absolute on a term that is
//     2^-11 of the product).  MODE 6: three exact bf16 planes each, six products (any range).
//   * partial sums per workgroup, finished by k_wgrad_reduce in a fixed order (bit-reproducible run to run).
// Tiles are dealt to the waves as whole (mt, nt, dy) triples (three dx taps share the operand reads) plus single left-over tiles, so that the
// two waves of every SIMD carry 20 or 21 of the 81 tiles.
#pragma once

// Tuning probes (wrong results by design; tools/build_variant.sh <name> -DDPX_WC_DBG=<bits>): 1 no matrix phase, 2 no split pass, 4 no partial-sum
// stores, 8 no LDS-DMA
#ifndef DPX_WC_DBG
#define DPX_WC_DBG 0
#endif

namespace dpx {

// Tuning aid (tools/build_variant_one.sh wc_trace dpx_wgrad_c8 -DDPX_WC_TRACE; never in the shipped library): the waves of workgroup 40 stamp the shader
// clock along their steps 8 .. 11; tools/wgrad_trace.py prints the timeline of the last launch.
#ifdef DPX_WC_TRACE
__device__ unsigned long long dpx_wc_trace_buf[8 * 64];
#define DPX_WC_STAMP(i)                                                                                                                       \
  do {                                                                                                                                       \
    const long st_ = c - c_begin - 8;                                                                                                        \
    if (MT == 3 && NT == 3 && lane == 0 && blockIdx.x == 40 && st_ >= 0 && st_ < 4) dpx_wc_trace_buf[wv * 64 + st_ * 8 + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define DPX_WC_STAMP(i) ((void)0)
#endif

constexpr int WC_WT = 32;                                  // pixels of a gradient row per step (two matrix K-steps of 16)
constexpr int WC_ROWB = 80;                                // bytes between the channels of a plane row: 40 x 16 bit (A uses 34, G 32)
constexpr int WC_NW = 8;                                   // waves per workgroup

template <int MT, int NT, int MODE>
struct WcGeom {
  static constexpr int NPL = MODE == 3 ? 2 : 3, CoP = MT * 32, CiP = NT * 32;
  static constexpr int GP = NPL * CoP * WC_ROWB, AP = NPL * CiP * WC_ROWB;                 // G planes; one slot of the A ring
  static constexpr int OFF_A = GP, OFF_RG = OFF_A + 3 * AP, OFF_RA = OFF_RG + (CoP / 8) * 1024, OFF_RE = OFF_RA + (CiP / 8) * 1024;
  static constexpr int OFF_BS = OFF_RE + 1024;                                              // the bias gradient's running sums: 8 floats per pair-thread of G
  static constexpr int LDS_BYTES = OFF_BS + (CoP / 8) * 16 * 32;
  static constexpr int NTRIP = 3 * NT * MT, BASE = NTRIP / WC_NW, REM = NTRIP % WC_NW;    // whole (mt, nt, dy) triples per wave; left-over triples
  static constexpr int NEX = (3 * REM + WC_NW - 1) / WC_NW, NACC = 3 * BASE + NEX;          // left-over single tiles per wave; accumulators per wave
};

typedef _Float16 wc_f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 wc_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned wc_scale2(unsigned a, float s) {       // both binary16 halves of a dword times a power of two
  wc_f16x2 v = __builtin_bit_cast(wc_f16x2, a);
  v = v * (_Float16)s;
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ uint4 wc_scale8(uint4 a, float s) {
  wc_f16x8 v = __builtin_bit_cast(wc_f16x8, a);
  v = v * (_Float16)s;
  return __builtin_bit_cast(uint4, v);
}
// pixels DX .. DX + 7 of ten packed 16-bit pixels (five dwords, pixel 0 in the low half of p[0])
template <int DX>
__device__ __forceinline__ uint4 wc_shift(const unsigned (&p)[5]) {
  if constexpr (DX == 0) return make_uint4(p[0], p[1], p[2], p[3]);
  else if constexpr (DX == 2) return make_uint4(p[1], p[2], p[3], p[4]);
  else return make_uint4((p[0] >> 16) | (p[1] << 16), (p[1] >> 16) | (p[2] << 16), (p[2] >> 16) | (p[3] << 16), (p[3] >> 16) | (p[4] << 16));   // (v_alignbit)
}

// G: C8 [B][Gg][H][W][8] (the gradient w.r.t. the layer's pre-activation output), A: C8 [B][Ga][H][W][8] (the layer's input);
// part: [gridDim.x][MT NT 9 tiles][1024] (the accumulators' layout, see the epilogue), part_b: [gridDim.x][MT 32][2].  Channels beyond 8 Gg / 8 Ga count as zero.
template <int MT, int NT, int MODE>
__global__ void __launch_bounds__(WC_NW * 64, 1) k_wgrad_c8(const float* __restrict__ G, const float* __restrict__ A, float* __restrict__ part,
                                                            float* __restrict__ part_b, int Gg, int Ga, int B, int H, int W, int nstrips, long nchunks,
                                                            unsigned* __restrict__ f16_flag) {
  typedef WcGeom<MT, NT, MODE> GM;
  constexpr int NPL = GM::NPL, CoP = GM::CoP, CiP = GM::CiP, BASE = GM::BASE, REM = GM::REM, NEX = GM::NEX, NACC = GM::NACC;
  HIP_DYNAMIC_SHARED(char, smem_wc)
  char* const gpl = smem_wc;                                // G planes [plane][co][40 px]
  char* const apl = smem_wc + GM::OFF_A;                    // A ring: 3 slots of [plane][ci][40 px]; stored pixel s <-> image column x0 + s - 1
  char* const raw_g = smem_wc + GM::OFF_RG;                 // landing areas (fp32, C8 order): [group][32 px][8]
  char* const raw_a = smem_wc + GM::OFF_RA;
  char* const raw_e = smem_wc + GM::OFF_RE;                 // the input row's two outer pixels: [group][left, right][8]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, kg = lane >> 5;

  // ---- the planes of channels no group covers (and the pad pixels) stay zero
  for (int i = tid * 16; i < GM::OFF_RG; i += WC_NW * 64 * 16) *(uint4*)(smem_wc + i) = make_uint4(0u, 0u, 0u, 0u);

  // ---- a thread's item of the split pass: a pixel pair of one 8-channel group of the gradient row (16 pairs per group) or of the input
  // row (17 pairs: stored pixels 0 .. 33).  Recomputed from the thread index in front of every split pass (a dozen integer instructions
  // against five registers held across the matrix phase, whose 176 accumulators leave none to spare).
  const int n_gi = Gg * 16, n_ai = Ga * 17;
  struct Item {
    bool g_item, a_item;
    int g, pp, src0, src1, dst_off;                         // group, pair; the pair's two pixels in the landing areas; its dwords' offset in a plane
  };
  auto item_of = [&](int tid_) {
    Item it;
    it.g_item = tid_ < n_gi;
    it.a_item = !it.g_item && tid_ - n_gi < n_ai;
    if (it.g_item) {
      it.g = tid_ >> 4;
      it.pp = tid_ & 15;
      it.src0 = GM::OFF_RG + (it.g * 32 + 2 * it.pp) * 32;
      it.src1 = it.src0 + 32;
    } else {
      const int a = it.a_item ? tid_ - n_gi : 0;
      it.g = a / 17;
      it.pp = a - it.g * 17;
      const int s0 = 2 * it.pp, s1 = s0 + 1;               // stored pixels; s = 0: the left outer pixel, s = 33: the right one, else main pixel s - 1
      it.src0 = s0 == 0 ? GM::OFF_RE + (it.g * 2) * 32 : GM::OFF_RA + (it.g * 32 + s0 - 1) * 32;
      it.src1 = s1 == 33 ? GM::OFF_RE + (it.g * 2 + 1) * 32 : GM::OFF_RA + (it.g * 32 + s1 - 1) * 32;
    }
    it.dst_off = (it.g * 8) * WC_ROWB + it.pp * 4;
    return it;
  };
  const bool g_item = tid < n_gi;
  float f16_max = 0.f;
  // the bias gradient's running sums live in LDS (a private 32-byte slot per pair-thread of G: 8 registers less across the matrix phase)
  float* const bslot = (float*)(smem_wc + GM::OFF_BS + (g_item ? tid : 0) * 32);
  if (g_item) {
    *(float4*)bslot = make_float4(0.f, 0.f, 0.f, 0.f);
    *(float4*)(bslot + 4) = make_float4(0.f, 0.f, 0.f, 0.f);
  }

  // ---- this wave's tiles: triples t = wv BASE + i (i < BASE; BASE == MT: the MT output blocks of ONE (nt, dy) -- one read and one set of
  // shifts of the input operand serve all of them) and left-over tiles j = wv + 8 e of the triples 8 BASE ..; t = (nt 3 + dy) MT + mt
  auto decode = [&](int t, int& mt, int& nt, int& dy) {
    const int grp = t / MT;
    mt = t - grp * MT;
    nt = grp / 3;
    dy = grp - nt * 3;
  };
  int tr_g[BASE > 0 ? BASE : 1], tr_a[BASE > 0 ? BASE : 1], tr_dy[BASE > 0 ? BASE : 1];
#pragma unroll
  for (int i = 0; i < BASE; ++i) {
    int mt, nt;
    decode(wv * BASE + i, mt, nt, tr_dy[i]);
    tr_g[i] = mt * 32 * WC_ROWB;
    tr_a[i] = nt * 32 * WC_ROWB;
  }
  int ex_g[NEX > 0 ? NEX : 1], ex_a[NEX > 0 ? NEX : 1], ex_dy[NEX > 0 ? NEX : 1], ex_dx[NEX > 0 ? NEX : 1];   // ex_dx < 0: no such tile
#pragma unroll
  for (int e = 0; e < NEX; ++e) {
    const int j = wv + WC_NW * e;
    const bool on = j < 3 * REM;
    const int jj = on ? j : 0;
    int mt, nt;
    decode(WC_NW * BASE + jj / 3, mt, nt, ex_dy[e]);
    ex_dx[e] = on ? jj - (jj / 3) * 3 : -1;
    ex_g[e] = mt * 32 * WC_ROWB;
    ex_a[e] = nt * 32 * WC_ROWB;
  }
  const int lane_off = n * WC_ROWB + kg * 16;              // this lane's channel row and its 8 (+ 2) pixels of a 16-pixel K-step

  f32x16 acc[NACC];
#pragma unroll
  for (int u = 0; u < NACC; ++u)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[u][i] = 0.f;

  // ---- the walk: chunk c = (b nstrips + xs) H + y, down a strip
  const long c_begin = nchunks * blockIdx.x / gridDim.x, c_end = nchunks * (blockIdx.x + 1) / gridDim.x;
  int y = (int)(c_begin % H), xs = (int)((c_begin / H) % nstrips), b = (int)(c_begin / ((long)H * nstrips));
  const size_t plane8 = (size_t)H * W * 8;

  // LDS-DMA of one step's rows into the landing areas: whole 1 KB group rows (32 pixels x 32 bytes), dealt to the waves; pixels / rows outside
  // the image are fetched from the clamped position and zeroed by the split pass
  auto issue = [&](bool with_g, int b_, int xs_, int yg, int ya) {
    const int x0 = xs_ * WC_WT;
    const int px = lane >> 1, half = lane & 1;
    const int xc = min(x0 + px, W - 1);
    const unsigned voff = (unsigned)((xc * 8 + half * 4) * 4);
    const int yac = min(max(ya, 0), H - 1);
    const int n_i = (with_g ? Gg : 0) + Ga + 1;
    for (int k = wv; k < ((DPX_WC_DBG & 8) ? 0 : n_i); k += WC_NW) {
      const int ka = with_g ? k - Gg : k;
      if (ka < 0) {
        dpx_glds16_s(G + ((size_t)b_ * Gg + k) * plane8 + (size_t)yg * W * 8, voff, raw_g + k * 1024);
      } else if (ka < Ga) {
        dpx_glds16_s(A + ((size_t)b_ * Ga + ka) * plane8 + (size_t)yac * W * 8, voff, raw_a + ka * 1024);
      } else {
        const int g = min(lane >> 2, Ga - 1), e = (lane >> 1) & 1;
        const int x = e ? min(x0 + WC_WT, W - 1) : max(x0 - 1, 0);
        dpx_glds16_s(A + (size_t)b_ * Ga * plane8, (unsigned)((((size_t)g * H + yac) * W + x) * 32 + half * 16), raw_e);
      }
    }
  };

  // the split pass of one landed row: fp32 pixel pair x 8 channels -> one dword per channel and plane
  auto split_item = [&](const Item& it, char* planes, int plane_bytes, bool ok0, bool ok1, bool is_g) {
    const float4 p0a = *(const float4*)(smem_wc + it.src0), p0b = *(const float4*)(smem_wc + it.src0 + 16);
    const float4 p1a = *(const float4*)(smem_wc + it.src1), p1b = *(const float4*)(smem_wc + it.src1 + 16);
    const float v0[8] = {p0a.x, p0a.y, p0a.z, p0a.w, p0b.x, p0b.y, p0b.z, p0b.w};
    const float v1[8] = {p1a.x, p1a.y, p1a.z, p1a.w, p1b.x, p1b.y, p1b.z, p1b.w};
    if (is_g) {
      float4 ba = *(const float4*)bslot, bb = *(const float4*)(bslot + 4);
      ba.x += (ok0 ? v0[0] : 0.f) + (ok1 ? v1[0] : 0.f);
      ba.y += (ok0 ? v0[1] : 0.f) + (ok1 ? v1[1] : 0.f);
      ba.z += (ok0 ? v0[2] : 0.f) + (ok1 ? v1[2] : 0.f);
      ba.w += (ok0 ? v0[3] : 0.f) + (ok1 ? v1[3] : 0.f);
      bb.x += (ok0 ? v0[4] : 0.f) + (ok1 ? v1[4] : 0.f);
      bb.y += (ok0 ? v0[5] : 0.f) + (ok1 ? v1[5] : 0.f);
      bb.z += (ok0 ? v0[6] : 0.f) + (ok1 ? v1[6] : 0.f);
      bb.w += (ok0 ? v0[7] : 0.f) + (ok1 ? v1[7] : 0.f);
      *(float4*)bslot = ba;
      *(float4*)(bslot + 4) = bb;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float a0 = ok0 ? v0[j] : 0.f, a1 = ok1 ? v1[j] : 0.f;
      char* d = planes + it.dst_off + j * WC_ROWB;
      if constexpr (MODE == 3) {
        unsigned hw, lw;
        split2_f16_pair(a0, a1, hw, lw);
        *(unsigned*)d = hw;
        *(unsigned*)(d + plane_bytes) = lw;
        f16_max = fmaxf(f16_max, fmaxf(fabsf(a0), fabsf(a1)));
      } else {
        unsigned h0, m0, l0, h1, m1, l1;
        split3(a0, h0, m0, l0);
        split3(a1, h1, m1, l1);
        *(unsigned*)d = pack_hi16(h0, h1);
        *(unsigned*)(d + plane_bytes) = pack_hi16(m0, m1);
        *(unsigned*)(d + 2 * plane_bytes) = pack_hi16(l0, l1);
      }
    }
  };
  auto split_a = [&](const Item& it, int xs_, int ya, int slot) {
    if (!it.a_item) return;
    const int x0 = xs_ * WC_WT, xa = x0 + 2 * it.pp - 1;    // image column of stored pixel 2 pp
    const bool row_ok = ya >= 0 && ya < H;
    split_item(it, apl + slot * GM::AP, CiP * WC_ROWB, row_ok && xa >= 0 && xa < W, row_ok && xa + 1 < W, false);
  };
  auto split_g = [&](const Item& it, int xs_) {
    if (!it.g_item) return;
    const int xg = xs_ * WC_WT + 2 * it.pp;
    split_item(it, gpl, CoP * WC_ROWB, xg < W, xg + 1 < W, true);
  };
  auto my_item = [&]() {
    int tid_c = tid;
    DPX_OPAQUE(tid_c);                                      // (per split pass: the geometry must not be hoisted out of the walk)
    return item_of(tid_c);
  };
  // an input row on its own (the two rows above a strip's first step)
  auto stage_row = [&](int b_, int xs_, int ya, int slot) {
    issue(false, b_, xs_, 0, ya);
    dpx_wait_vm<0>();
    DPX_LDS_BARRIER();
    split_a(my_item(), xs_, ya, slot);
    DPX_LDS_BARRIER();
  };

  // one tile: the three (six) products of its K-step
  auto tile_mma = [&](f32x16& d, const uint4 (&gq)[3], const uint4 (&aq)[3]) {
    if constexpr (MODE == 3) {                               // gq: gh, gh 2^-6, gl' 2^-6;  aq: ah, ah 2^-5, al' 2^-5
      d = mfma_f16(gq[1], aq[2], d);
      d = mfma_f16(gq[2], aq[1], d);
      d = mfma_f16(gq[0], aq[0], d);
    } else {                                                 // small terms first, the leading product last (as the forward layers' MODE = 6)
      d = mfma_bf16(gq[1], aq[1], d);
      d = mfma_bf16(gq[2], aq[0], d);
      d = mfma_bf16(gq[0], aq[2], d);
      d = mfma_bf16(gq[1], aq[0], d);
      d = mfma_bf16(gq[0], aq[1], d);
      d = mfma_bf16(gq[0], aq[0], d);
    }
  };
  auto load_g = [&](int goff, int ks, uint4 (&gq)[3]) {
    const char* p = gpl + goff + lane_off + ks * 32;
    if constexpr (MODE == 3) {
      gq[0] = *(const uint4*)p;
      gq[1] = wc_scale8(gq[0], 0.015625f);
      gq[2] = wc_scale8(*(const uint4*)(p + CoP * WC_ROWB), 0.015625f);
    } else {
#pragma unroll
      for (int q = 0; q < 3; ++q) gq[q] = *(const uint4*)(p + q * CoP * WC_ROWB);
    }
  };
  // ten pixels per plane of the lane's input channel; MODE 3: [ah, ah 2^-5, al' 2^-5]
  auto load_a = [&](int aoff, int slot, int ks, unsigned (&ar)[3][5]) {
    const char* p = apl + slot * GM::AP + aoff + lane_off + ks * 32;
    unsigned raw[NPL][5];
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
      const uint4 v = *(const uint4*)(p + q * CiP * WC_ROWB);
      raw[q][0] = v.x;
      raw[q][1] = v.y;
      raw[q][2] = v.z;
      raw[q][3] = v.w;
      raw[q][4] = *(const unsigned*)(p + q * CiP * WC_ROWB + 16);
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      if constexpr (MODE == 3) {
        ar[0][k] = raw[0][k];
        ar[1][k] = wc_scale2(raw[0][k], 0.03125f);
        ar[2][k] = wc_scale2(raw[1][k], 0.03125f);
      } else {
        ar[0][k] = raw[0][k];
        ar[1][k] = raw[1][k];
        ar[2][k] = raw[NPL - 1][k];
      }
    }
  };
  auto shifted = [&](auto dxc, const unsigned (&ar)[3][5], uint4 (&aq)[3]) {
    constexpr int DX = decltype(dxc)::value;
#pragma unroll
    for (int q = 0; q < 3; ++q) aq[q] = wc_shift<DX>(ar[q]);
  };

  int s0 = 0, s1 = 1, s2 = 2;                               // ring slots of the input rows y - 1, y, y + 1
  bool fresh = true;
  for (long c = c_begin; c < c_end; ++c) {
    if (fresh) {                                            // the first step of this workgroup / of a strip: rows y - 1 and y come on their own
      stage_row(b, xs, y - 1, s0);
      stage_row(b, xs, y, s1);
      issue(true, b, xs, y, y + 1);
    }
    DPX_WC_STAMP(0);
    dpx_wait_vm<0>();
    DPX_WC_STAMP(1);
    DPX_LDS_BARRIER();                                      // this step's rows have landed; everybody is done with the previous step's planes
    DPX_WC_STAMP(2);
    if (!(DPX_WC_DBG & 2)) {
      const Item it = my_item();
      split_g(it, xs);
      split_a(it, xs, y + 1, s2);
    }
    DPX_WC_STAMP(3);
    DPX_LDS_BARRIER();                                      // planes ready, landing areas free
    DPX_WC_STAMP(4);
    const int r0 = s0, r1 = s1, r2 = s2;
    auto slot_of = [&](int dy) { return dy == 0 ? r0 : (dy == 1 ? r1 : r2); };   // (selects: an indexed array would live in scratch memory)
    // the next step's position; its rows travel under this step's matrix instructions
    int yn = y + 1, xn = xs, bn = b;
    fresh = false;
    if (yn == H) {
      yn = 0;
      fresh = true;
      if (++xn == nstrips) {
        xn = 0;
        ++bn;
      }
    }
    if (c + 1 < c_end && !fresh) issue(true, bn, xn, yn, yn + 1);
    DPX_WC_STAMP(5);
#pragma unroll
    for (int ks = 0; ks < ((DPX_WC_DBG & 1) ? 0 : 2); ++ks) {
      if constexpr (BASE == MT && MT > 1) {
        unsigned ar[3][5];
        uint4 a0[3], a1[3], a2[3];
        load_a(tr_a[0], slot_of(tr_dy[0]), ks, ar);
        shifted(std::integral_constant<int, 0>{}, ar, a0);
        shifted(std::integral_constant<int, 1>{}, ar, a1);
        shifted(std::integral_constant<int, 2>{}, ar, a2);
#pragma unroll
        for (int i = 0; i < BASE; ++i) {
          uint4 gq[3];
          load_g(tr_g[i], ks, gq);
          tile_mma(acc[3 * i + 0], gq, a0);
          tile_mma(acc[3 * i + 1], gq, a1);
          tile_mma(acc[3 * i + 2], gq, a2);
        }
      } else {
#pragma unroll
        for (int i = 0; i < BASE; ++i) {
          unsigned ar[3][5];
          uint4 gq[3], aq[3];
          load_a(tr_a[i], slot_of(tr_dy[i]), ks, ar);
          load_g(tr_g[i], ks, gq);
          shifted(std::integral_constant<int, 0>{}, ar, aq);
          tile_mma(acc[3 * i + 0], gq, aq);
          shifted(std::integral_constant<int, 1>{}, ar, aq);
          tile_mma(acc[3 * i + 1], gq, aq);
          shifted(std::integral_constant<int, 2>{}, ar, aq);
          tile_mma(acc[3 * i + 2], gq, aq);
        }
      }
#pragma unroll
      for (int e = 0; e < NEX; ++e) {
        if (ex_dx[e] >= 0) {                                // (wave-uniform)
          unsigned ar[3][5];
          uint4 gq[3], aq[3];
          load_a(ex_a[e], slot_of(ex_dy[e]), ks, ar);
          load_g(ex_g[e], ks, gq);
          if (ex_dx[e] == 0) shifted(std::integral_constant<int, 0>{}, ar, aq);
          else if (ex_dx[e] == 1) shifted(std::integral_constant<int, 1>{}, ar, aq);
          else shifted(std::integral_constant<int, 2>{}, ar, aq);
          tile_mma(acc[3 * BASE + e], gq, aq);
        }
      }
    }
    DPX_WC_STAMP(6);
    y = yn;
    xs = xn;
    b = bn;
    const int t = s0;
    s0 = s1;
    s1 = s2;
    s2 = t;
  }
  if (MODE == 3 && !(f16_max <= 6.0e4f)) atomicOr(f16_flag, 1u);        // (NaN counts: dpx_ffdnet_f16_overflow)

  // ---- this workgroup's partial sums, tile by tile in the accumulators' own order: [tile (mt NT + nt) 9 + tap][i >> 2][lane][i & 3] -- 1 KB per
  // store instruction (the reduction, k_wgrad_c8_reduce, knows the D layout: col = lane & 31 (ci), row = (i & 3) + 8 (i >> 2) + 4 (lane >> 5) (co))
  float* dst = part + (size_t)blockIdx.x * CoP * CiP * 9;
  auto store_tile = [&](const f32x16& d, int mt, int nt, int tap) {
    float* t = dst + (size_t)((mt * NT + nt) * 9 + tap) * 1024 + lane * 4;
#pragma unroll
    for (int q = 0; q < ((DPX_WC_DBG & 4) ? 1 : 4); ++q) *(float4*)(t + q * 256) = make_float4(d[4 * q], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3]);
  };
#pragma unroll
  for (int i = 0; i < BASE; ++i) {
    int mt, nt, dy;
    decode(wv * BASE + i, mt, nt, dy);
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) store_tile(acc[3 * i + dx], mt, nt, dy * 3 + dx);
  }
#pragma unroll
  for (int e = 0; e < NEX; ++e)
    if (ex_dx[e] >= 0) {
      int mt, nt, dy;
      decode(WC_NW * BASE + (wv + WC_NW * e) / 3, mt, nt, dy);
      store_tile(acc[3 * BASE + e], mt, nt, dy * 3 + ex_dx[e]);
    }
  // bias gradient: the 16 pair-threads of a group are 16 neighbouring lanes
  if (tid < ((n_gi + 63) & ~63)) {
    const float4 ba = *(const float4*)bslot, bb = *(const float4*)(bslot + 4);
    const float bs[8] = {ba.x, ba.y, ba.z, ba.w, bb.x, bb.y, bb.z, bb.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = g_item ? bs[j] : 0.f;
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o);
      if (g_item && (tid & 15) == 0) {
        float* pb = part_b + ((size_t)blockIdx.x * CoP + (tid >> 4) * 8 + j) * 2;
        pb[0] = v;
        pb[1] = 0.f;
      }
    }
  }
}

}  // namespace dpx
