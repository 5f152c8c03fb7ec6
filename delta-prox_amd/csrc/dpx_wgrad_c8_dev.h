// Weight / bias gradients of a 3x3 layer straight from the C8 planes the split kernels read and write (instantiated by dpx_wgrad_c8.hip).
//
//   dW[co][ci][dy][dx] = sum_{b, y, x} G[b][co][y][x] * A[b][ci][y + dy - 1][x + dx - 1]   (zero outside the image),   db[co] = sum G[b][co][y][x]
//
// a GEMM whose K axis is the PIXELS: v_mfma_f32_32x32x16_{f16,bf16} wants, per lane, 8 consecutive K of ONE channel, and the C8 layout
// [B][C/8][H][W][8] keeps 8 CHANNELS of one pixel together.  Round 4's kernel (k_wgrad_bf16x3) therefore read planar copies of both
// operands (k_bx_c8_to_planar: two extra plane passes per layer), every wave split its own operands (each element of G was split by three
// waves, each of A by three), and it ran one wave per SIMD at 48 % of its roofline: 268 + 70 us per 96 -> 96 layer at 2 x 384 x 384.  Here:
//   * a workgroup (8 waves, one per CU, persistent) owns ALL MT x NT x 9 accumulator tiles of the layer -- 81 tiles of 32 x 32 at 96 -> 96
//     channels, 10 - 11 per wave, 176 accumulator registers -- and walks DOWN 32-pixel column strips (K = 32 pixels per step);
//   * every element of G and A is fetched once per workgroup by LDS-DMA (whole 1 KB group rows: 32 pixels x 8 channels, nothing through
//     registers), split ONCE, and transposed on its way into the operand planes: a lane takes a pixel PAIR of one 8-channel group and
//     writes one dword (two pixels of one channel) per channel and plane -- planes [plane][channel][pixels] of 16-bit elements, 80 bytes
//     between channels (conflict-free 16-byte reads).  The group rows of a step are dealt to the waves, and a wave splits what it fetched
//     itself: no barrier between a fetch and its split pass, one workgroup barrier per step;
//   * the operand of tap (dy, dx) is the input row y + dy - 1 (a ring of four rows in LDS: one new row per step) read at pixels + dx:
//     ten pixels per lane, the three horizontal shifts by v_alignbit, shared by the three output blocks a wave owns;
//   * MODE 3 (the split-f16 backward pass: G arrives scaled, dpx_conv_bf16.hip "gradient scale"): g = gh + gl' / 2^11, a = ah + al' / 2^11,
//     three products into ONE accumulator -- gh ah + (gh 2^-6)(al' 2^-5) + (gl' 2^-6)(ah 2^-5): the cross terms' factor 2^-11 is spread over
//     both operands by packed multiplications (exact above the subnormal range; below it the error is 2^-25 absolute on a term that is
//     2^-11 of the product).  MODE 6: three exact bf16 planes each, six products (any range);
//   * partial sums per workgroup in the accumulators' own layout (1 KB per store instruction), finished by k_wgrad_c8_reduce in a fixed order
//     (bit-reproducible run to run).
// Tiles are dealt to the waves as whole (mt, nt, dy) triples (three dx taps share the operand reads) plus single left-over tiles, so that the
// two waves of every SIMD carry 20 or 21 of the 81 tiles.  Measured (MI355X, 2 x 96 x 384 x 384): 141 us per launch inside the training step
// (197 us on random data, whose matrix instructions draw more power: 248 TFLOP/s fp32-equivalent), split-bf16 266 us; DESIGN.md section 8.
#pragma once

// Tuning probes (wrong results by design; tools/build_variant.sh <name> -DDPX_WC_DBG=<bits>): 1 no matrix phase, 2 no split pass, 4 no partial-sum
// stores, 8 no LDS-DMA
#ifndef DPX_WC_DBG
#define DPX_WC_DBG 0
#endif

namespace dpx {

// Tuning aid (tools/build_variant_one.sh wc_trace dpx_wgrad_c8 -DDPX_WC_TRACE; never in the shipped library): the waves of workgroup 40 stamp the shader
// clock along their jobs 8 .. 11; tools/wgrad_trace.py prints the timeline of the last launch.
#ifdef DPX_WC_TRACE
__device__ unsigned long long dpx_wc_trace_buf[8 * 64];
#define DPX_WC_STAMP(i)                                                                                                                       \
  do {                                                                                                                                       \
    const long st_ = c - 8;                                                                                                        \
    if (MT == 3 && NT == 3 && lane == 0 && blockIdx.x == 40 && st_ >= 0 && st_ < 4) dpx_wc_trace_buf[wv * 64 + st_ * 8 + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define DPX_WC_STAMP(i) ((void)0)
#endif

constexpr int WC_WT = 32;                                  // pixels of a gradient row per step (two matrix K-steps of 16)
constexpr int WC_NW = 8;                                   // waves per workgroup

template <int MT, int NT, int MODE>
struct WcGeom {
  static constexpr int NPL = MODE == 3 ? 2 : 3, CoP = MT * 32, CiP = NT * 32;
  // bytes between the channels of a plane row (34 pixels of the input row, 32 of the gradient row, 16 bit each): 80 = 20 dwords, conflict-free
  // 16-byte reads; the three-plane arithmetic takes 72 (a few two-way conflicts) so that its six plane sets still fit the CU's 160 KB
  static constexpr int ROWB = NPL == 2 ? 80 : 72;
  static constexpr int GP = NPL * CoP * ROWB, AP = NPL * CiP * ROWB;                         // one set of G planes; one slot of the A ring
  static constexpr int OFF_A = 2 * GP, OFF_RG = OFF_A + 4 * AP, OFF_RA = OFF_RG + (CoP / 8) * 1024, OFF_RE = OFF_RA + (CiP / 8) * 1024;
  static constexpr int OFF_BS = OFF_RE + WC_NW * 256;                                       // the bias gradient's running sums: 8 floats per pixel pair of a G group row
  static constexpr int LDS_BYTES = OFF_BS + (CoP / 8) * 16 * 32;
  static constexpr int NTRIP = 3 * NT * MT, BASE = NTRIP / WC_NW, REM = NTRIP % WC_NW;    // whole (mt, nt, dy) triples per wave; left-over triples
  static constexpr int NEX = (3 * REM + WC_NW - 1) / WC_NW, NACC = 3 * BASE + NEX;          // left-over single tiles per wave; accumulators per wave
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
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
// part: [gridDim.x][MT NT 9 tiles][1024] (the accumulators' layout, see the epilogue), part_b: [gridDim.x][MT 32][2].  Channels beyond 8 Gg / 8 Ga
// count as zero.
template <int MT, int NT, int MODE>
__global__ void __launch_bounds__(WC_NW * 64, 1) k_wgrad_c8(const float* __restrict__ G, const float* __restrict__ A, float* __restrict__ part,
                                                            float* __restrict__ part_b, int Gg, int Ga, int B, int H, int W, int nstrips, long nchunks,
                                                            unsigned* __restrict__ f16_flag) {
  typedef WcGeom<MT, NT, MODE> GM;
  constexpr int NPL = GM::NPL, CoP = GM::CoP, CiP = GM::CiP, BASE = GM::BASE, REM = GM::REM, NEX = GM::NEX, NACC = GM::NACC, ROWB = GM::ROWB;
  HIP_DYNAMIC_SHARED(char, smem_wc)
  char* const gpl = smem_wc;                                // two sets of G planes [plane][co][pixels]
  char* const apl = smem_wc + GM::OFF_A;                    // A ring: 4 slots of [plane][ci][pixels]; stored pixel s <-> image column x0 + s - 1
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, kg = lane >> 5;

  // ---- the planes of channels no group covers (and the pad pixels) stay zero; the bias sums start at zero
  for (int i = tid * 16; i < GM::OFF_RG; i += WC_NW * 64 * 16) *(uint4*)(smem_wc + i) = make_uint4(0u, 0u, 0u, 0u);
  for (int i = tid * 16; i < GM::LDS_BYTES - GM::OFF_BS; i += WC_NW * 64 * 16) *(uint4*)(smem_wc + GM::OFF_BS + i) = make_uint4(0u, 0u, 0u, 0u);

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
    tr_g[i] = mt * 32 * ROWB;
    tr_a[i] = nt * 32 * ROWB;
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
    ex_g[e] = mt * 32 * ROWB;
    ex_a[e] = nt * 32 * ROWB;
  }
  const int lane_off = n * ROWB + kg * 16;                  // this lane's channel row and its 8 (+ 2) pixels of a 16-pixel K-step

  f32x16 acc[NACC];
#pragma unroll
  for (int u = 0; u < NACC; ++u)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[u][i] = 0.f;
  float f16_max = 0.f;

  // ---- the walk.  Real steps c = (b nstrips + xs) H + y go down a strip: step y stages the gradient row y and the input row y + 1 and multiplies
  // against the input rows y - 1, y, y + 1.  In front of a workgroup's first step and of every strip's first row come two staging-only jobs
  // (the input rows y - 1 and y), so that the pipeline below never needs a special case: job J stages ONE input row into ring slot J & 3 (and,
  // if real, one gradient row into plane set J & 1) and a real job reads the slots J - 2, J - 1, J.
  struct Job {
    int b, xs, ya, yg;                                      // image, strip, input row (-1 .. H: outside = zeros), gradient row (< 0: none)
    bool real, valid;
  };
  const long c_begin = nchunks * blockIdx.x / gridDim.x, c_end = nchunks * (blockIdx.x + 1) / gridDim.x;
  int it_y = (int)(c_begin % H), it_xs = (int)((c_begin / H) % nstrips), it_b = (int)(c_begin / ((long)H * nstrips)), it_warm = 2;
  long it_left = c_end - c_begin;
  auto next_job = [&]() {
    Job j;
    j.b = it_b;
    j.xs = it_xs;
    j.valid = it_left > 0;
    j.real = false;
    j.yg = -1;
    j.ya = 0;
    if (!j.valid) return j;
    if (it_warm == 2) {
      j.ya = it_y - 1;
      it_warm = 1;
    } else if (it_warm == 1) {
      j.ya = it_y;
      it_warm = 0;
    } else {
      j.ya = it_y + 1;
      j.yg = it_y;
      j.real = true;
      --it_left;
      if (++it_y == H) {
        it_y = 0;
        it_warm = 2;
        if (++it_xs == nstrips) {
          it_xs = 0;
          ++it_b;
        }
      }
    }
    return j;
  };

  // ---- staging, wave by wave: wave w owns the group rows k = w, w + 8, w + 16 of a job (k < Gg: group k of the gradient row, then the groups
  // of the input row) -- it fetches them (LDS-DMA, 1 KB = 32 pixels x 8 channels each; the input row's two outer pixels of its groups by one
  // more instruction), and it alone reads them back: no workgroup barrier between a fetch and its split pass.  Pixels / rows outside the image
  // are fetched from the clamped position and zeroed by the split pass.
  const size_t plane8 = (size_t)H * W * 8;
  // piece p of this wave's share of job jb: p = 0 .. 2 the group rows k = wv + 8 p, p = 3 the outer pixels of its input groups
  auto issue_piece = [&](const Job& jb, int p) {
    if ((DPX_WC_DBG & 8) || !jb.valid) return;
    char* const rawb = smem_wc;
    const int x0 = jb.xs * WC_WT;
    const int yac = min(max(jb.ya, 0), H - 1);
    if (p < 3) {
      const int k = wv + WC_NW * p;
      const int xc = min(x0 + (lane >> 1), W - 1);
      const unsigned voff = (unsigned)((xc * 8 + (lane & 1) * 4) * 4);
      if (k < Gg) {
        if (jb.yg >= 0) dpx_glds16_s(G + ((size_t)jb.b * Gg + k) * plane8 + (size_t)jb.yg * W * 8, voff, rawb + GM::OFF_RG + k * 1024);
      } else if (k < Gg + Ga) {
        dpx_glds16_s(A + ((size_t)jb.b * Ga + (k - Gg)) * plane8 + (size_t)yac * W * 8, voff, rawb + GM::OFF_RA + (k - Gg) * 1024);
      }
    } else if (wv + 2 * WC_NW >= Gg && wv < Gg + Ga) {      // (one of k = wv, wv + 8, wv + 16 is an input group)
      if (lane < 16) {                                      // lanes 4 j .. 4 j + 3: [left, right] outer pixel of piece j's group, two halves each
        const int kk = wv + WC_NW * (lane >> 2), ka = min(max(kk - Gg, 0), Ga - 1);
        const int x = (lane & 2) ? min(x0 + WC_WT, W - 1) : max(x0 - 1, 0);
        const unsigned vo = (((unsigned)ka * (unsigned)H + (unsigned)yac) * (unsigned)W + (unsigned)x) * 32u + (lane & 1) * 16u;
        dpx_glds16_s(A + (size_t)jb.b * Ga * plane8, vo, rawb + GM::OFF_RE + wv * 256);
      }
    }
  };
  auto issue = [&](const Job& jb) {
#pragma unroll
    for (int p = 0; p < 4; ++p) issue_piece(jb, p);
  };
  // the split pass of this wave's landed group rows: lane = (piece j, pixel pair pp); fp32 pixel pair x 8 channels -> one dword per channel and plane
  auto split_own = [&](const Job& jb, int J) {
    if (DPX_WC_DBG & 2) return;
    int lane_c = lane;
    DPX_OPAQUE(lane_c);                                     // (the item geometry is recomputed per pass: a dozen integer instructions against registers
    const int j = lane_c / 17, pp = lane_c - j * 17;        //  held across the matrix phase, whose 176 accumulators leave none to spare)
    const int k = wv + WC_NW * j;
    const bool is_g = k < Gg, is_a = !is_g && k < Gg + Ga;
    if (!((is_g && pp < 16 && jb.yg >= 0) || is_a)) return;
    const int x0 = jb.xs * WC_WT;
    const char* const rawb = smem_wc;
    int src0, src1, dst;
    bool ok0, ok1;
    if (is_g) {
      src0 = GM::OFF_RG + k * 1024 + pp * 64;
      src1 = src0 + 32;
      dst = (J & 1) * GM::GP + (k * 8) * ROWB + pp * 4;
      const int xg = x0 + 2 * pp;
      ok0 = xg < W;
      ok1 = xg + 1 < W;
    } else {
      const int ka = k - Gg, main = GM::OFF_RA + ka * 1024, edge = GM::OFF_RE + wv * 256 + j * 64;
      src0 = pp == 0 ? edge : main + (2 * pp - 1) * 32;    // stored pixels 2 pp, 2 pp + 1; s = 0: the left outer pixel, s = 33: the right one, else main pixel s - 1
      src1 = pp == 16 ? edge + 32 : main + (2 * pp) * 32;
      dst = GM::OFF_A + (J & 3) * GM::AP + (ka * 8) * ROWB + pp * 4;
      const int xa = x0 + 2 * pp - 1;
      const bool row_ok = jb.ya >= 0 && jb.ya < H;
      ok0 = row_ok && xa >= 0 && xa < W;
      ok1 = row_ok && xa + 1 < W;
    }
    const int plane_bytes = (is_g ? CoP : CiP) * ROWB;
    const float4 p0a = *(const float4*)(rawb + src0), p0b = *(const float4*)(rawb + src0 + 16);
    const float4 p1a = *(const float4*)(rawb + src1), p1b = *(const float4*)(rawb + src1 + 16);
    float v0[8] = {p0a.x, p0a.y, p0a.z, p0a.w, p0b.x, p0b.y, p0b.z, p0b.w};
    float v1[8] = {p1a.x, p1a.y, p1a.z, p1a.w, p1b.x, p1b.y, p1b.z, p1b.w};
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      v0[c] = ok0 ? v0[c] : 0.f;
      v1[c] = ok1 ? v1[c] : 0.f;
    }
    if (is_g) {                                             // the bias gradient's running sums: a private 32-byte slot per (group, pixel pair)
      float* bs = (float*)(smem_wc + GM::OFF_BS + (k * 16 + pp) * 32);
      float4 ba = *(const float4*)bs, bb = *(const float4*)(bs + 4);
      ba.x += v0[0] + v1[0];
      ba.y += v0[1] + v1[1];
      ba.z += v0[2] + v1[2];
      ba.w += v0[3] + v1[3];
      bb.x += v0[4] + v1[4];
      bb.y += v0[5] + v1[5];
      bb.z += v0[6] + v1[6];
      bb.w += v0[7] + v1[7];
      *(float4*)bs = ba;
      *(float4*)(bs + 4) = bb;
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      char* d = smem_wc + dst + c * ROWB;
      if constexpr (MODE == 3) {
        unsigned hw, lw;
        split2_f16_pair(v0[c], v1[c], hw, lw);
        *(unsigned*)d = hw;
        *(unsigned*)(d + plane_bytes) = lw;
        f16_max = fmaxf(f16_max, fmaxf(fabsf(v0[c]), fabsf(v1[c])));
      } else {
        unsigned h0, m0, l0, h1, m1, l1;
        split3(v0[c], h0, m0, l0);
        split3(v1[c], h1, m1, l1);
        *(unsigned*)d = pack_hi16(h0, h1);
        *(unsigned*)(d + plane_bytes) = pack_hi16(m0, m1);
        *(unsigned*)(d + 2 * plane_bytes) = pack_hi16(l0, l1);
      }
    }
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
  auto read16 = [&](const char* p) {                       // 16 bytes of a plane row (rows 72 bytes apart are 8 modulo 16: two 8-byte reads)
    if constexpr (ROWB % 16 == 0) return *(const uint4*)p;
    else {
      const uint2 v = *(const uint2*)p, w = *(const uint2*)(p + 8);
      return make_uint4(v.x, v.y, w.x, w.y);
    }
  };
  auto load_g = [&](const char* gset, int goff, int ks, uint4 (&gq)[3]) {
    const char* p = gset + goff + lane_off + ks * 32;
    if constexpr (MODE == 3) {
      gq[0] = read16(p);
      gq[1] = wc_scale8(gq[0], 0.015625f);
      gq[2] = wc_scale8(read16(p + CoP * ROWB), 0.015625f);
    } else {
#pragma unroll
      for (int q = 0; q < 3; ++q) gq[q] = read16(p + q * CoP * ROWB);
    }
  };
  // ten pixels per plane of the lane's input channel; MODE 3: [ah, ah 2^-5, al' 2^-5]
  auto load_a = [&](int aoff, int slot, int ks, unsigned (&ar)[3][5]) {
    const char* p = apl + slot * GM::AP + aoff + lane_off + ks * 32;
    unsigned raw[NPL][5];
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
      const uint4 v = read16(p + q * CiP * ROWB);
      raw[q][0] = v.x;
      raw[q][1] = v.y;
      raw[q][2] = v.z;
      raw[q][3] = v.w;
      raw[q][4] = *(const unsigned*)(p + q * CiP * ROWB + 16);
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
  // the matrix phase of real job J: input rows y - 1, y, y + 1 in ring slots J - 2, J - 1, J; gradient row in plane set J & 1
  auto matrix_phase = [&](int J) {
    const char* gset = gpl + (J & 1) * GM::GP;
    const int r0 = (J - 2) & 3, r1 = (J - 1) & 3, r2 = J & 3;
    auto slot_of = [&](int dy) { return dy == 0 ? r0 : (dy == 1 ? r1 : r2); };   // (selects: an indexed array would live in scratch memory)
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
          load_g(gset, tr_g[i], ks, gq);
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
          load_g(gset, tr_g[i], ks, gq);
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
          load_g(gset, ex_g[e], ks, gq);
          if (ex_dx[e] == 0) shifted(std::integral_constant<int, 0>{}, ar, aq);
          else if (ex_dx[e] == 1) shifted(std::integral_constant<int, 1>{}, ar, aq);
          else shifted(std::integral_constant<int, 2>{}, ar, aq);
          tile_mma(acc[3 * BASE + e], gq, aq);
        }
      }
    }
  };

  // ---- the pipeline: job J + 1 is split and job J + 2 sent for, then job J multiplies.  One barrier per job.  Measured per job at 96 -> 96
  // channels (tools/wgrad_trace.py; shader cycles): split pass ~2000, the LDS-DMA issue of the 25 pieces ~1300 (the CU takes ~50 cycles per KB beside
  // matrix work), matrix phase ~4800 for 3936 cycles of matrix instructions per SIMD, barrier skew ~900.  Measured and dropped (193 - 210 us per
  // launch, all of them: the three parts do not overlap, whatever the order): one wave of a SIMD staging while the other multiplies (a wave on
  // its own keeps the matrix pipe 41 % busy, two together 80 %), the LDS-DMA pieces issued between the tiles of the matrix phase (it grows by
  // what the issue took), two sets of landing areas with the LDS-DMA issued in front of the split pass, the cross terms' scaled operands kept
  // as third planes instead of the packed multiplications per use.
  Job j0 = next_job(), j1 = next_job(), j2 = next_job();
  issue(j0);
  dpx_wait_vm<0>();
  DPX_LDS_BARRIER();                                        // (the zero fill above is complete)
  split_own(j0, 0);
  dpx_wait_lds();
  issue(j1);
  DPX_LDS_BARRIER();
  for (int J = 0; j0.valid; ++J) {
    const long c = J;
    DPX_WC_STAMP(0);
    if (j1.valid) {
      dpx_wait_vm<0>();                                     // this wave's rows of job J + 1 have landed
      split_own(j1, J + 1);
    }
    dpx_wait_lds();                                         // ... and have been read: their landing areas take job J + 2
    issue(j2);
    DPX_WC_STAMP(1);
    if (j0.real) matrix_phase(J);
    DPX_WC_STAMP(2);
    DPX_LDS_BARRIER();
    DPX_WC_STAMP(3);
    j0 = j1;
    j1 = j2;
    j2 = next_job();
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
  // bias gradient of channel tid: its group's 16 pair sums, in a fixed order
  if (tid < Gg * 8) {
    const float* bs = (const float*)(smem_wc + GM::OFF_BS) + (tid >> 3) * 16 * 8 + (tid & 7);
    float v = 0.f;
#pragma unroll
    for (int pp = 0; pp < 16; ++pp) v += bs[pp * 8];
    float* pb = part_b + ((size_t)blockIdx.x * CoP + tid) * 2;
    pb[0] = v;
    pb[1] = 0.f;
  }
}

}  // namespace dpx
