// FFDNet denoiser = the deep_prior z-update (reference dprox/proxfn/pnp/prior.py:73-86 ->
// denoisers/models/network_ffdnet.py:54-68, conv stack built by basicblock.py:61-98, pixel-unshuffle :104-126).
//
//   k_ffd_pack_in   : replicate-pad to even size, pixel-unshuffle(2) (channel = c*4 + dy*2 + dx), append the
//                     per-image sigma map as the last channel                      network_ffdnet.py:56-63
//   k_conv3x3_mfma  : 3x3 / pad 1 convolution + bias (+ ReLU) as an implicit GEMM on the exact-fp32 matrix cores
//                     (v_mfma_f32_32x32x2_f32: D[32x32] += A[32x2] B[2x32], bitwise an fmaf chain), one workgroup =
//                     all output channels x (8 rows x 32 columns) pixels; K runs over (input-channel pair, tap).
//   k_ffd_unpack_out: PixelShuffle(2) + crop                                      network_ffdnet.py:65-67
//
// GEMM view per layer: M = Cout (1..3 tiles of 32), N = pixels, K = 9 * Cin.  Per K-step of 2 (two input channels,
// one tap) a wave reads MT weight fragments and 2 pixel fragments from LDS (one float per lane each, conflict-free:
// weights are pre-packed [ci/2][tap][ci&1][cout] so cout is the fast axis, pixels are the fast axis of the staged
// input tile) and issues 2*MT MFMAs.  FLOP roofline: 157.3 TFLOP/s fp32 MFMA.
#include "dpx_common.h"

namespace dpx {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int FFD_TW = 32;        // tile width  (pixels, = MFMA N)
constexpr int FFD_TH = 8;         // default tile height (rows): 4 waves x NR = 2 rows (NR = 4: 16 rows)
constexpr int FFD_CK = 8;         // input channels staged per chunk
// staged input tile of a dilation-DIL 3x3 convolution: apron of DIL pixels on every side
template <int DIL, int NR = 2> struct FfdTile {
  static constexpr int TH = 4 * NR;                                    // output rows per workgroup
  static constexpr int ROWS = TH + 2 * DIL;                           // 10 / 12 / 14 / 16 (NR = 2)
  static constexpr int USED = FFD_TW + 2 * DIL;                       // 34 / 36 / 38 / 40 columns read
  static constexpr int LDW = DIL == 1 ? 36 : (DIL == 4 ? 42 : 40);       // row pitch: ROWS * LDW mod 64 = 40 / 32 / 48 / 32 keeps the
                                                                      // two channel halves of a K-step on (mostly) different banks
};

static inline int pad_even(int c) { return (c + 1) & ~1; }
static inline int mtiles(int cout) { return (cout + 31) / 32; }

// packed layer: weights [Cin_pad/2][9][2][MT*32], bias [MT*32], then FFD_ZPAD zero floats (the "zero word" LDS-DMA lanes
// outside the image read from)
constexpr int FFD_ZPAD = 4;
static inline size_t layer_floats(int cin, int cout) {
  return (size_t)(pad_even(cin) / 2) * 9 * 2 * mtiles(cout) * 32 + (size_t)mtiles(cout) * 32 + FFD_ZPAD;
}

// transposed != 0: the layer of the backward-data pass, conv with W'[co'][ci'][tap] = W[ci'][co'][8 - tap] (w is still the
// forward tensor [cout' ... ] = [cin][cout]-swapped view: w has shape [cin][cout][3][3] in the primed names), zero bias
__global__ void k_ffd_pack_weights(const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ dst, int cin, int cout,
                                   int transposed) {
  const int MT32 = ((cout + 31) / 32) * 32, pairs = ((cin + 1) & ~1) / 2;
  const long nw = (long)pairs * 9 * 2 * MT32;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nw + MT32 + FFD_ZPAD; i += (long)gridDim.x * blockDim.x) {
    if (i >= nw + MT32) {
      dst[i] = 0.f;
    } else if (i < nw) {
      const int co = (int)(i % MT32);
      long r = i / MT32;
      const int half = (int)(r % 2);
      r /= 2;
      const int tap = (int)(r % 9), cp = (int)(r / 9);
      const int ci = 2 * cp + half;
      if (!transposed) dst[i] = (co < cout && ci < cin) ? w[((long)co * cin + ci) * 9 + tap] : 0.f;
      else dst[i] = (co < cout && ci < cin) ? w[((long)ci * cout + co) * 9 + (8 - tap)] : 0.f;
    } else {
      const int co = (int)(i - nw);
      dst[i] = (co < cout && !transposed) ? b[co] : 0.f;
    }
  }
}

// x [B][C][H][W] -> a [B][Cp][H2][W2] with Cp = pad_even(4C+1); channel 4C = sigma, channel 4C+1 (if any) = 0
__global__ void k_ffd_pack_in(const float* __restrict__ x, const float* __restrict__ sigma, float* __restrict__ a, int B, int C,
                              int H, int W, int H2, int W2, int Cp) {
  const long total = (long)B * Cp * H2 * W2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x2 = (int)(i % W2);
    long r = i / W2;
    const int y2 = (int)(r % H2);
    r /= H2;
    const int ch = (int)(r % Cp), b = (int)(r / Cp);
    float v;
    if (ch < 4 * C) {
      const int c = ch >> 2, dy = (ch >> 1) & 1, dx = ch & 1;
      const int yy = min(2 * y2 + dy, H - 1), xx = min(2 * x2 + dx, W - 1);       // ReplicationPad2d (bottom / right)
      v = x[(((long)b * C + c) * H + yy) * W + xx];
    } else {
      v = (ch == 4 * C) ? sigma[b] : 0.f;
    }
    a[i] = v;
  }
}

// o [B][4C][H2][W2] -> y [B][C][H][W]  (PixelShuffle(2): out[c][2y+dy][2x+dx] = in[c*4 + dy*2 + dx][y][x], then crop)
__global__ void k_ffd_unpack_out(const float* __restrict__ o, float* __restrict__ y, int B, int C, int H, int W, int H2, int W2) {
  const long total = (long)B * C * H * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int xx = (int)(i % W);
    long r = i / W;
    const int yy = (int)(r % H);
    r /= H;
    const int c = (int)(r % C), b = (int)(r / C);
    const int ch = c * 4 + (yy & 1) * 2 + (xx & 1);
    y[i] = o[(((long)b * 4 * C + ch) * H2 + (yy >> 1)) * W2 + (xx >> 1)];
  }
}

// ---- backward helpers ------------------------------------------------------------------------------------------
// adjoint of k_ffd_unpack_out: g_o[b][c*4 + dy*2 + dx][y2][x2] = gy[b][c][2 y2 + dy][2 x2 + dx] (0 outside the crop)
__global__ void k_ffd_pack_gout(const float* __restrict__ gy, float* __restrict__ go, int B, int C, int H, int W, int H2, int W2) {
  const long total = (long)B * 4 * C * H2 * W2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x2 = (int)(i % W2);
    long r = i / W2;
    const int y2 = (int)(r % H2);
    r /= H2;
    const int ch = (int)(r % (4 * C)), b = (int)(r / (4 * C));
    const int c = ch >> 2, yy = 2 * y2 + ((ch >> 1) & 1), xx = 2 * x2 + (ch & 1);
    go[i] = (yy < H && xx < W) ? gy[(((long)b * C + c) * H + yy) * W + xx] : 0.f;
  }
}

// adjoint of k_ffd_pack_in w.r.t. x: the replicate padding of an odd size makes the last row / column receive two terms
__global__ void k_ffd_unpack_gin(const float* __restrict__ ga, float* __restrict__ gx, int B, int C, int H, int W, int H2, int W2, int Cp) {
  const long total = (long)B * C * H * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int xx = (int)(i % W);
    long r = i / W;
    const int yy = (int)(r % H);
    r /= H;
    const int c = (int)(r % C), b = (int)(r / C);
    const int y2 = yy >> 1, x2 = xx >> 1;
    const int dy0 = yy & 1, dx0 = xx & 1;
    const int ndy = (yy == H - 1 && (H & 1)) ? 2 : 1, ndx = (xx == W - 1 && (W & 1)) ? 2 : 1;   // padded duplicates
    float acc = 0.f;
    for (int a = 0; a < ndy; ++a)
      for (int e = 0; e < ndx; ++e) {
        const int dy = ndy == 2 ? a : dy0, dx = ndx == 2 ? e : dx0;
        acc += ga[(((long)b * Cp + c * 4 + dy * 2 + dx) * H2 + y2) * W2 + x2];
      }
    gx[i] = acc;
  }
}

// gsigma[b] = sum over the sigma-map channel of g_a0 (one block per image, fixed summation order)
__global__ void __launch_bounds__(256) k_ffd_sigma_grad(const float* __restrict__ ga, float* __restrict__ gs, int C, int H2, int W2, int Cp) {
  __shared__ float sh[256];
  const int b = blockIdx.x;
  const float* pl = ga + ((size_t)b * Cp + 4 * C) * H2 * W2;
  float acc = 0.f;
  for (long i = threadIdx.x; i < (long)H2 * W2; i += 256) acc += pl[i];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) gs[b] = sh[0];
}

// NP channel pairs x NTAP taps (9: 3x3 kernel, 1: 1x1 kernel = centre tap only), fully unrolled; the LDS fragments of
// step k+1 are fetched while step k multiplies
template <int MT, int NP, int NTAP = 9, int DIL = 1, int NR = 2>
__device__ __forceinline__ void mfma_chunk(f32x16 (&acc)[MT][NR], const float* __restrict__ sin_b, const float* __restrict__ sw_b) {
  constexpr int M32 = MT * 32, NS = NP * NTAP;
  constexpr int FFD_LDW = FfdTile<DIL, NR>::LDW, FFD_ROWS = FfdTile<DIL, NR>::ROWS;
  constexpr int T0 = NTAP == 9 ? 0 : 4;                       // first tap index in the 3x3 stencil (1x1: the centre)
  float a_cur[MT], b_cur[NR];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) a_cur[mt] = sw_b[mt * 32];
#pragma unroll
  for (int r = 0; r < NR; ++r) b_cur[r] = sin_b[((T0 / 3) * DIL + r) * FFD_LDW + (T0 % 3) * DIL];
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    float a_nxt[MT], b_nxt[NR];
    if (k + 1 < NS) {
      const int cp = (k + 1) / NTAP, tw = (k + 1) % NTAP, tap = T0 + tw, dy = tap / 3, dx = tap % 3;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) a_nxt[mt] = sw_b[(cp * NTAP + tw) * 2 * M32 + mt * 32];
#pragma unroll
      for (int r = 0; r < NR; ++r) b_nxt[r] = sin_b[(2 * cp * FFD_ROWS + dy * DIL + r) * FFD_LDW + dx * DIL];
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int r = 0; r < NR; ++r) acc[mt][r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[mt], b_cur[r], acc[mt][r], 0, 0, 0);
    }
    if (k + 1 < NS) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) a_cur[mt] = a_nxt[mt];
#pragma unroll
      for (int r = 0; r < NR; ++r) b_cur[r] = b_nxt[r];
    }
  }
}

// in [B][Cin][H2][W2] (Cin even), out [B][Cout][H2][W2]; wpk = packed layer (see layer_floats)
// MASKED: the result is stored as out[.] * [mask[.] > 0] (backward through the ReLU that produced `mask`, the activation
// at the OUTPUT position of this transposed layer, fused into the epilogue)
// NTAP = 9: 3x3 / pad 1; NTAP = 1: 1x1 (stride-2 and transposed 2x2 convolutions are 1x1 convolutions around a
// space-to-depth / depth-to-space rearrangement).  blockIdx.z selects a block of MT*32 output channels (layers wider than
// 96 channels: DRUNet), each with its own packed weight block.  RES: out = conv + res (residual blocks).
// NR = output rows per wave (2 or 4): with 2 output-channel tiles (64-channel layers: gray FFDNet, DRUNet blocks) NR = 4
// doubles the matrix-core work per weight fragment read and per staged halo row.
// M16 (MT = 1, at most 16 output channels, Cin a multiple of 4: the denoisers' last layer): the 16x16x4 matrix-core tile --
// 16 output channels x 16 pixels x 4 input channels per instruction at the same rate -- so that a 12- or 4-channel layer does
// not pay for 32 output rows.  Lane = (pixel or channel lane & 15, input channel lane >> 4 of the 4-group).
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MT, bool RELU, bool MASKED = false, int NTAP = 9, bool RES = false, int DIL = 1, int NR = 2, bool M16 = false>
// (MT = 3: 96 output channels x NR rows of fp32 accumulators per wave -- one wave per SIMD is what the register file holds; declared as such)
__global__ void __launch_bounds__(256, (MT >= 3 ? 1 : 2)) k_conv3x3_mfma(const float* __restrict__ in, float* __restrict__ out,
                                                       const float* __restrict__ wpk, int Cin, int Cout, int H2, int W2, int tiles_x,
                                                       const float* __restrict__ mask, float slope) {
  // RELU epilogue: max(v, 0) + slope * min(v, 0) -- slope = 0: ReLU, 0.2: the U-Net's LeakyReLU (exact for either sign)
  constexpr int M32 = MT * 32;
  constexpr int FFD_LDW = FfdTile<DIL, NR>::LDW, FFD_ROWS = FfdTile<DIL, NR>::ROWS, FFD_USED = FfdTile<DIL, NR>::USED;
  constexpr int FFD_TH = FfdTile<DIL, NR>::TH;
  __shared__ float s_in[2 * FFD_CK * FFD_ROWS * FFD_LDW];                              // 2 x [ch][row][col]
  __shared__ __attribute__((aligned(16))) float s_w[2 * (FFD_CK / 2) * NTAP * 2 * M32];  // 2 x [pair][tap][half][cout]
  const int co0 = blockIdx.z * M32;                                                    // first output channel of this block
  wpk += (size_t)blockIdx.z * ((size_t)(Cin / 2) * NTAP * 2 * M32 + M32 + FFD_ZPAD);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int y0 = ty * FFD_TH, x0 = tx * FFD_TW;
  const int j = lane & 31, half = lane >> 5;
  const float* inb = in + (size_t)b * Cin * H2 * W2;
  const float* maskb = (MASKED || RES) ? mask + (size_t)b * Cout * H2 * W2 : nullptr;   // RES: `mask` carries the residual input
  const float* bias = wpk + (size_t)(Cin / 2) * NTAP * 2 * M32;

  static_assert(!M16 || (MT == 1 && NTAP == 9 && !MASKED && !RES), "M16: plain 3x3 layers with one 32-channel weight block");
  f32x16 acc[M16 ? 1 : MT][M16 ? 1 : NR];
  f32x4 acc16[M16 ? NR : 1][2];
#pragma unroll
  for (int mt = 0; mt < (M16 ? 1 : MT); ++mt)
#pragma unroll
    for (int nt = 0; nt < (M16 ? 1 : NR); ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
#pragma unroll
  for (int nt = 0; nt < (M16 ? NR : 1); ++nt)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc16[nt][h][r] = 0.f;

  // Software pipeline over chunks of FFD_CK input channels, two LDS buffers, one barrier per chunk.
  //  * forward layers: the next chunk is fetched by LDS-DMA (no VGPRs, no LDS-write pass, ~20 issue slots per wave and
  //    chunk instead of ~450 staging instructions that competed with the MFMA issue: 63 % -> 8x % of the MFMA peak).
  //    The weight chunk is a contiguous block of the packed layer (1 KB pieces, 16 B per lane); the input tile is
  //    fetched dword-wise (rows start at arbitrary alignments), lanes outside the image read the layer's zero word.
  constexpr int NPI = (FFD_CK * FFD_ROWS * FFD_LDW + 63) / 64;             // dword pieces of one input chunk (45)
  constexpr int NPW = (NPI + 3) / 4;                                       // per wave
  constexpr int NWP = ((FFD_CK / 2) * NTAP * 2 * M32) / 256;               // 1 KB pieces of one weight chunk (27 / 18 / 9; 3 / 2 / 1)
  static_assert(((FFD_CK / 2) * NTAP * 2 * M32) % 256 == 0, "weight chunk must be whole 1 KB pieces");
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  unsigned ioff[NPW];                      // (channel << 28) | element offset inside the chunk's channel block; ~0u = zero word
  {
#pragma unroll
    for (int k = 0; k < NPW; ++k) {
      const int e = (wv + 4 * k) * 64 + lane;
      const int ch = e / (FFD_ROWS * FFD_LDW), r = (e / FFD_LDW) % FFD_ROWS, col = e % FFD_LDW;
      const int yy = y0 + r - DIL, xx = x0 + col - DIL;
      const bool ok = (wv + 4 * k) < NPI && ch < FFD_CK && col < FFD_USED && yy >= 0 && yy < H2 && xx >= 0 && xx < W2;
      ioff[k] = ok ? (((unsigned)ch << 28) | (unsigned)((ch * H2 + yy) * W2 + xx)) : ~0u;
    }
  }
  const float* zero_word = bias + M32;
  auto issue = [&](int c0, int buf) {
    {
      const int nch = min(FFD_CK, Cin - c0);
      const float* cb = inb + (size_t)c0 * H2 * W2;
      float* si = s_in + buf * (FFD_CK * FFD_ROWS * FFD_LDW);
#pragma unroll
      for (int k = 0; k < NPW; ++k) {
        if (wv + 4 * k < NPI) {
          const unsigned pk = ioff[k];
          const bool ok = pk != ~0u && (int)(pk >> 28) < nch;
          const float* src = ok ? cb + (pk & 0x0fffffffu) : zero_word;
          dpx_glds4(src, si + (wv + 4 * k) * 64);
        }
      }
      const float* wsrc = wpk + (size_t)(c0 / 2) * NTAP * 2 * M32 + lane * 4;
      float* sw = s_w + buf * ((FFD_CK / 2) * NTAP * 2 * M32);
      // only the pieces this chunk's channels need (a partial chunk would otherwise read past the packed layer; the last
      // piece may still over-read by < 1 KB: the blobs carry that much slack)
      const int npieces = ((nch / 2) * NTAP * 2 * M32 + 255) / 256;
      for (int i = wv; i < npieces; i += 4) dpx_glds16(wsrc + i * 256, sw + i * 256);
    }
  };
  issue(0, 0);
  dpx_wait_vm<0>();
  __syncthreads();
  int buf = 0;
  for (int c0 = 0; c0 < Cin; c0 += FFD_CK, buf ^= 1) {
    const int nch = min(FFD_CK, Cin - c0);                     // even
    const bool more = c0 + FFD_CK < Cin;
    if (more) issue(c0 + FFD_CK, buf ^ 1);
    const float* sin_b = s_in + buf * (FFD_CK * FFD_ROWS * FFD_LDW) + (half * FFD_ROWS + NR * wave) * FFD_LDW + j;
    const float* sw_b = s_w + buf * ((FFD_CK / 2) * NTAP * 2 * M32) + half * M32 + j;
    if constexpr (M16) {
      const int k4 = lane >> 4, i16 = lane & 15;
      const float* sin16 = s_in + buf * (FFD_CK * FFD_ROWS * FFD_LDW) + (k4 * FFD_ROWS + NR * wave) * FFD_LDW + i16;
      const float* sw16 = s_w + buf * ((FFD_CK / 2) * NTAP * 2 * M32) + ((k4 >> 1) * NTAP * 2 + (k4 & 1)) * M32 + i16;
      for (int cp = 0; cp < nch / 2; cp += 2) {              // 4 input channels per step
        const float* sb = sin16 + 2 * cp * FFD_ROWS * FFD_LDW;
        const float* sa = sw16 + cp * NTAP * 2 * M32;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const float av = sa[tap * 2 * M32];
#pragma unroll
          for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const float bv = sb[((tap / 3) * DIL + r) * FFD_LDW + (tap % 3) * DIL + 16 * h];
              acc16[r][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc16[r][h], 0, 0, 0);
            }
        }
      }
    } else {
      for (int cp = 0; cp < nch / 2; ++cp) mfma_chunk<MT, 1, NTAP, DIL, NR>(acc, sin_b + 2 * cp * FFD_ROWS * FFD_LDW, sw_b + cp * NTAP * 2 * M32);
    }
    if (more) dpx_wait_vm<0>();
    __syncthreads();
  }
  // ---- epilogue: bias, ReLU, store.  C/D layout: col = lane & 31 (pixel), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (cout) ----
  float* outb = out + (size_t)b * Cout * H2 * W2;
  if constexpr (M16) {                     // 16x16 C/D layout: col = lane & 15 (pixel), row = 4 * (lane >> 4) + r (cout)
#pragma unroll
    for (int nt = 0; nt < NR; ++nt) {
      const int yy = y0 + NR * wave + nt;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int xq = x0 + 16 * h + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int cl = 4 * (lane >> 4) + r, co = co0 + cl;
          float v = acc16[nt][h][r] + bias[cl];
          if (RELU) v = fmaxf(v, 0.f) + slope * fminf(v, 0.f);
          if (co < Cout && yy < H2 && xq < W2) outb[((size_t)co * H2 + yy) * W2 + xq] = v;
        }
      }
    }
    return;
  }
  const int xx = x0 + j;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NR; ++nt) {
      const int yy = y0 + NR * wave + nt;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cl = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, co = co0 + cl;
        float v = acc[mt][nt][r] + bias[cl];
        if (RELU) v = fmaxf(v, 0.f) + slope * fminf(v, 0.f);
        if (co < Cout && yy < H2 && xx < W2) {
          const size_t idx = ((size_t)co * H2 + yy) * W2 + xx;
          if (MASKED) v = maskb[idx] > 0.f ? v : 0.f;
          if (RES) v += maskb[idx];
          outb[idx] = v;
        }
      }
    }
}

template <int MT>
static void launch_conv(bool relu, const float* in, float* out, const float* wpk, int Cin, int Cout, int B, int H2, int W2, hipStream_t s,
                        const float* mask = nullptr) {
  constexpr int NR = 2;        // rows per wave (NR = 4 measured: +2 % on 1024^2 gray planes, -5..-13 % on 320^2 planes and DRUNet's deep levels)
  const int tx = (W2 + FFD_TW - 1) / FFD_TW, ty = (H2 + 4 * NR - 1) / (4 * NR);
  if (mask) {
    DPX_LAUNCH("k_conv3x3_mfma_bwd", (k_conv3x3_mfma<MT, false, true, 9, false, 1, NR>), dim3(tx * ty, B), dim3(256), 0, s, in, out, wpk, Cin,
               Cout, H2, W2, tx, mask, 0.f);
    return;
  }
  if constexpr (MT == 1) {
    if (!relu && Cout <= 16 && Cin % 4 == 0) {               // the last layer: 16-wide matrix-core tile
      DPX_LAUNCH("k_conv3x3_mfma16", (k_conv3x3_mfma<1, false, false, 9, false, 1, NR, true>), dim3(tx * ty, B), dim3(256), 0, s, in, out, wpk,
                 Cin, Cout, H2, W2, tx, (const float*)nullptr, 0.f);
      return;
    }
  }
  if (relu)
    DPX_LAUNCH("k_conv3x3_mfma", (k_conv3x3_mfma<MT, true, false, 9, false, 1, NR>), dim3(tx * ty, B), dim3(256), 0, s, in, out, wpk, Cin, Cout,
               H2, W2, tx, (const float*)nullptr, 0.f);
  else
    DPX_LAUNCH("k_conv3x3_mfma", (k_conv3x3_mfma<MT, false, false, 9, false, 1, NR>), dim3(tx * ty, B), dim3(256), 0, s, in, out, wpk, Cin, Cout,
               H2, W2, tx, (const float*)nullptr, 0.f);
}

static int layer_cin(int l, int in_nc, int nc) { return l == 0 ? 4 * in_nc + 1 : nc; }
static int layer_cout(int l, int in_nc, int nc, int nb) { return l == nb - 1 ? 4 * in_nc : nc; }

}  // namespace dpx

using namespace dpx;

extern "C" size_t dpx_ffdnet_packed_bytes(int in_nc, int nc, int nb) {
  size_t n = 0;
  for (int l = 0; l < nb; ++l) n += layer_floats(layer_cin(l, in_nc, nc), layer_cout(l, in_nc, nc, nb));
  return n * sizeof(float) + 1024;       // slack: the last 1 KB LDS-DMA piece of a partial channel chunk may over-read
}

extern "C" int dpx_ffdnet_pack(void* packed, const float* const* w, const float* const* b, int in_nc, int nc, int nb,
                               dpx_stream_t stream) {
  DPX_REQUIRE(packed && w && b && in_nc > 0 && nc > 0 && nb >= 2, "dpx_ffdnet_pack: bad arguments");
  DPX_REQUIRE(nc <= 96 && 4 * in_nc <= 96, "dpx_ffdnet_pack: at most 96 channels per layer are built");
  float* dst = (float*)packed;
  for (int l = 0; l < nb; ++l) {
    const int cin = layer_cin(l, in_nc, nc), cout = layer_cout(l, in_nc, nc, nb);
    DPX_REQUIRE(w[l] && b[l], "dpx_ffdnet_pack: layer %d has null weights", l);
    const size_t n = layer_floats(cin, cout);
    DPX_LAUNCH("k_ffd_pack_weights", k_ffd_pack_weights, dim3(grid_for((long)n, 256, 1024)), dim3(256), 0, (hipStream_t)stream, w[l],
               b[l], dst, cin, cout, 0);
    dst += n;
  }
  return launch_status("dpx_ffdnet_pack");
}

extern "C" size_t dpx_ffdnet_ws_bytes(int B, int in_nc, int nc, int H, int W) {
  const size_t H2 = (H + 1) / 2, W2 = (W + 1) / 2;
  const size_t px = (size_t)B * H2 * W2;
  return (px * pad_even(4 * in_nc + 1) + 2 * px * nc + px * 4 * in_nc) * sizeof(float);
}

extern "C" int dpx_ffdnet_forward(const float* x, float* y, const float* sigma, const void* packed, int in_nc, int nc, int nb,
                                  int B, int H, int W, void* ws, dpx_stream_t stream) {
  DPX_REQUIRE(x && y && sigma && packed && ws, "dpx_ffdnet_forward: null pointer");
  DPX_REQUIRE((size_t)FFD_CK * ((H + 1) / 2) * ((W + 1) / 2) < ((size_t)1 << 28), "dpx_ffdnet_forward: plane %dx%d too large for the 28-bit staging offsets", H, W);
  DPX_REQUIRE(B > 0 && H > 0 && W > 0 && in_nc > 0 && nb >= 2 && nc % 2 == 0 && nc <= 96 && 4 * in_nc <= 96,
              "dpx_ffdnet_forward: unsupported configuration (in_nc=%d nc=%d nb=%d)", in_nc, nc, nb);
  hipStream_t s = (hipStream_t)stream;
  const int H2 = (H + 1) / 2, W2 = (W + 1) / 2, Cp = pad_even(4 * in_nc + 1);
  const size_t px = (size_t)B * H2 * W2;
  float* a0 = (float*)ws;
  float* bufA = a0 + px * Cp;
  float* bufB = bufA + px * nc;
  float* last = bufB + px * nc;
  DPX_LAUNCH("k_ffd_pack_in", k_ffd_pack_in, dim3(grid_for((long)(px * Cp), 256, 8192)), dim3(256), 0, s, x, sigma, a0, B, in_nc, H, W,
             H2, W2, Cp);
  const float* wl = (const float*)packed;
  const float* cur = a0;
  for (int l = 0; l < nb; ++l) {
    const int cin = layer_cin(l, in_nc, nc), cout = layer_cout(l, in_nc, nc, nb);
    float* dst = (l == nb - 1) ? last : ((l & 1) ? bufB : bufA);
    const bool relu = l != nb - 1;
    switch (mtiles(cout)) {
      case 1: launch_conv<1>(relu, cur, dst, wl, pad_even(cin), cout, B, H2, W2, s); break;
      case 2: launch_conv<2>(relu, cur, dst, wl, pad_even(cin), cout, B, H2, W2, s); break;
      default: launch_conv<3>(relu, cur, dst, wl, pad_even(cin), cout, B, H2, W2, s); break;
    }
    wl += layer_floats(cin, cout);
    cur = dst;
  }
  DPX_LAUNCH("k_ffd_unpack_out", k_ffd_unpack_out, dim3(grid_for((long)B * in_nc * H * W, 256, 8192)), dim3(256), 0, s, last, y, B,
             in_nc, H, W, H2, W2);
  return launch_status("dpx_ffdnet_forward");
}

// ---- weight gradients -------------------------------------------------------------------------------------------------
// dW[co][ci][tap] = sum_{b,y,x} G[b][co][y][x] * A[b][ci][y+dy-1][x+dx-1]  (tap = dy*3+dx),  db[co] = sum G[b][co][y][x]
// as a GEMM on the exact-fp32 matrix cores with the pixels as the K dimension: per MFMA (32x32x2) the D rows are 32 output
// channels, the D columns 32 input channels, K = 2 pixels; one wave owns one 32-channel slice of Cout, all 9 taps
// (9 accumulators) and the workgroup's block of 32 input channels.  A workgroup (MT waves) walks over pixel tiles
// (4 rows x 32 columns) in a fixed order and keeps the accumulators in registers; per-workgroup partial sums go to a
// workspace and a second kernel adds them in a fixed order (deterministic, no atomics).
// LDS: G tile [pixel][co] pitch 97 and activation tile [(row, col)][ci] pitch 33 -- the MFMA operands (lanes = channels)
// and the transposing stores (lanes = pixels) are both bank-conflict free with these odd pitches.
constexpr int WG_TH = 4, WG_TW = 32, WG_PX = WG_TH * WG_TW, WG_GP = 97, WG_AP = 33;

// NTAP = 9: 3x3 stencil with dilation DIL (apron of DIL pixels), 2*MT waves per workgroup: wave w owns output-channel slice
// w % MT and taps [0,5) (w < MT) or [5,9) (w >= MT), so that twice as many waves overlap the (register-staged, transposing)
// tile loads with the matrix-core work.  NTAP = 1: 1x1 convolution (the strided / transposed 2x2 convolutions of the U-Net
// around space-to-depth / depth-to-space), MT waves.  The launch covers output channels [co0, co0 + MT*32) of Cout.
template <int MT, int NTAP = 9, int DIL = 1>
__global__ void __launch_bounds__(MT * 128) k_conv3x3_wgrad(const float* __restrict__ G, const float* __restrict__ A, float* __restrict__ part,
                                                             float* __restrict__ part_b, int Cout, int co0, int Cin, int B, int H2, int W2,
                                                             int tiles_x, int tiles_y) {
  constexpr int PAD = NTAP == 9 ? DIL : 0, AW = WG_TW + 2 * PAD, AH = WG_TH + 2 * PAD, NACC = NTAP == 9 ? 5 : 1;
  __shared__ float s_g[WG_PX * WG_GP];
  __shared__ float s_a[AH * AW * WG_AP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = NTAP == 9 ? MT * 128 : MT * 64;
  const int j = lane & 31, kk = lane >> 5;
  const int mtile = wave % MT, t0 = (wave < MT) ? 0 : 5, nt = NTAP == 9 ? ((wave < MT) ? 5 : 4) : 1;
  const int cb = blockIdx.y;                            // block of 32 input channels
  f32x16 acc[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;
  const int ntiles = B * tiles_x * tiles_y;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int b = tile / (tiles_x * tiles_y), tr = tile - b * tiles_x * tiles_y;
    const int ty = tr / tiles_x, tx = tr - ty * tiles_x;
    const int y0 = ty * WG_TH, x0 = tx * WG_TW;
    __syncthreads();                                    // previous tile's operands are no longer needed
    // G tile: [pixel][co], zero outside the image / beyond Cout.  Loads are issued in batches before their LDS
    // writes so that each batch costs one memory round trip.
    constexpr int NG_E = MT * 32 * WG_PX, NA_E = 32 * AH * AW, BATCH = 16;
    for (int i0 = tid; i0 < NG_E; i0 += nthr * BATCH) {
      float vals[BATCH];
#pragma unroll
      for (int e = 0; e < BATCH; ++e) {
        const int i = i0 + e * nthr;
        const int x = i % WG_TW, y = (i / WG_TW) % WG_TH, co = co0 + i / WG_PX;
        const int yy = y0 + y, xx = x0 + x;
        vals[e] = (i < NG_E && co < Cout && yy < H2 && xx < W2) ? G[(((size_t)b * Cout + co) * H2 + yy) * W2 + xx] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < BATCH; ++e) {
        const int i = i0 + e * nthr;
        const int x = i % WG_TW, y = (i / WG_TW) % WG_TH, co = i / WG_PX;
        if (i < NG_E) s_g[(y * WG_TW + x) * WG_GP + co] = vals[e];
      }
    }
    // activation tile with a PAD-pixel apron: [(row, col)][ci]
    for (int i0 = tid; i0 < NA_E; i0 += nthr * BATCH) {
      float vals[BATCH];
#pragma unroll
      for (int e = 0; e < BATCH; ++e) {
        const int i = i0 + e * nthr;
        const int x = i % AW, y = (i / AW) % AH, c = i / (AH * AW);
        const int yy = y0 + y - PAD, xx = x0 + x - PAD, ci = cb * 32 + c;
        vals[e] = (i < NA_E && ci < Cin && yy >= 0 && yy < H2 && xx >= 0 && xx < W2) ? A[(((size_t)b * Cin + ci) * H2 + yy) * W2 + xx] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < BATCH; ++e) {
        const int i = i0 + e * nthr;
        const int x = i % AW, y = (i / AW) % AH, c = i / (AH * AW);
        if (i < NA_E) s_a[(y * AW + x) * WG_AP + c] = vals[e];
      }
    }
    __syncthreads();
    const float* ga = s_g + mtile * 32 + j;
#pragma unroll 4
    for (int s = 0; s < WG_PX / 2; ++s) {
      const int p = 2 * s + kk, py = p / WG_TW, px = p % WG_TW;
      const float av = ga[p * WG_GP];
      bsum += av;
      const float* ap = s_a + (py * AW + px) * WG_AP + j;
#pragma unroll
      for (int t = 0; t < NACC; ++t) {
        if (t < nt) {
          const int tap = t0 + t;
          const float bv = NTAP == 9 ? ap[((tap / 3) * DIL * AW + (tap % 3) * DIL) * WG_AP] : ap[0];
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
        }
      }
    }
  }
  // partial sums: part[blockIdx.x][co][ci][tap] over the padded channel counts (MT*32 x gridDim.y*32)
  const int CiP = gridDim.y * 32, CoP = MT * 32;
  float* pp = part + (size_t)blockIdx.x * CoP * CiP * NTAP;
#pragma unroll
  for (int t = 0; t < NACC; ++t)
    if (t < nt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = mtile * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk, ci = cb * 32 + j;
        pp[((size_t)co * CiP + ci) * NTAP + t0 + t] = acc[t][r];
      }
    }
  if (cb == 0 && wave < MT) part_b[((size_t)blockIdx.x * CoP + mtile * 32 + j) * 2 + kk] = bsum;
}

// gw[co0 + co][ci][tap] (co < CoN) = sum over the NG partial slices, in a fixed order
// mul (device, nullable): every sum leaves multiplied by *mul (the inverse gradient scale of a split-f16 backward pass: a power of two)
__global__ void __launch_bounds__(256) k_wgrad_reduce(const float* __restrict__ part, const float* __restrict__ part_b, float* __restrict__ gw,
                                                      float* __restrict__ gb, int NG, int CoN, int co0, int Cin, int CoP, int CiP, int ntap,
                                                      const float* __restrict__ mul) {
  __shared__ float shb[256];
  const float ms = mul ? *mul : 1.f;
  const long nw = (long)CoN * Cin * ntap;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nw; i += (long)gridDim.x * blockDim.x) {
    const int t = (int)(i % ntap), ci = (int)((i / ntap) % Cin), co = (int)(i / ((long)ntap * Cin));
    // eight interleaved running sums (slice g goes to sum g % 8), joined in a fixed order: the same bits run to run, eight loads in flight
    const float* p0 = part + ((size_t)co * CiP + ci) * ntap + t;
    const size_t gs = (size_t)CoP * CiP * ntap;
    float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int g = 0;
    for (; g + 8 <= NG; g += 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) a8[j] += p0[(size_t)(g + j) * gs];
    }
    for (int j = 0; g < NG; ++g, ++j) a8[j] += p0[(size_t)g * gs];
    if (gw) gw[(long)co0 * Cin * ntap + i] = (((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]))) * ms;
  }
  // the bias gradients: one workgroup per output channel, a thread per slice, a fixed tree over the threads (one thread walking the NG
  // slices of a channel was the launch's critical path: 168 dependent round trips = 63 us behind a 15-us weight pass)
  if (!gb) return;
  for (int co = blockIdx.x; co < CoN; co += gridDim.x) {
    float acc = 0.f;
    for (int g = threadIdx.x; g < NG; g += 256) acc += part_b[((size_t)g * CoP + co) * 2] + part_b[((size_t)g * CoP + co) * 2 + 1];
    __syncthreads();
    shb[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if ((int)threadIdx.x < w) shb[threadIdx.x] += shb[threadIdx.x + w];
      __syncthreads();
    }
    if (threadIdx.x == 0) gb[co0 + co] = shb[0] * ms;
  }
}

constexpr int WGRAD_NG = 168;      // persistent workgroups per input-channel block (FFDNet layers)
static size_t wgrad_ws_floats(int nc, int in_nc) {
  const int cop = mtiles(nc > 4 * in_nc ? nc : 4 * in_nc) * 32, cip = ((pad_even(nc > 4 * in_nc + 1 ? nc : 4 * in_nc + 1) + 31) / 32) * 32;
  return (size_t)WGRAD_NG * cop * cip * 9 + (size_t)WGRAD_NG * cop * 2;
}

template <int NTAP, int DIL>
static void launch_wgrad_mt(int MT, dim3 grid, hipStream_t s, const float* G, const float* A, float* part, float* part_b, int Cout, int co0,
                            int Cin_a, int B, int H2, int W2, int tx, int ty) {
  constexpr int TPW = NTAP == 9 ? 128 : 64;
  switch (MT) {
    case 1: DPX_LAUNCH("k_conv3x3_wgrad", (k_conv3x3_wgrad<1, NTAP, DIL>), grid, dim3(TPW), 0, s, G, A, part, part_b, Cout, co0, Cin_a, B, H2, W2, tx, ty); break;
    case 2: DPX_LAUNCH("k_conv3x3_wgrad", (k_conv3x3_wgrad<2, NTAP, DIL>), grid, dim3(2 * TPW), 0, s, G, A, part, part_b, Cout, co0, Cin_a, B, H2, W2, tx, ty); break;
    default: DPX_LAUNCH("k_conv3x3_wgrad", (k_conv3x3_wgrad<3, NTAP, DIL>), grid, dim3(3 * TPW), 0, s, G, A, part, part_b, Cout, co0, Cin_a, B, H2, W2, tx, ty); break;
  }
}

// persistent workgroups per input-channel block of the generic layer: about WGRAD_NG workgroups in total
static int wgrad_ng(int cin, long ntiles) {
  const int CB = (cin + 31) / 32;
  int ng = WGRAD_NG / CB;
  if (ng < 8) ng = 8;
  return (long)ng > ntiles ? (int)ntiles : ng;
}

// G: [B][Cout][H2][W2] gradient w.r.t. the layer's pre-activation output; A: [B][Cin_a][H2][W2] the layer's input;
// gw: [Cout][Cin_w][ntap].  Output channels are processed in blocks of <= 96 that reuse the workspace.
static void launch_wgrad(const float* G, const float* A, float* gw, float* gb, int Cout, int Cin_w, int Cin_a, int B, int H2, int W2,
                         float* ws, hipStream_t s, int ntap = 9, int dil = 1, int ng_max = WGRAD_NG) {
  const int tx = (W2 + WG_TW - 1) / WG_TW, ty = (H2 + WG_TH - 1) / WG_TH;
  const int CB = (Cin_w + 31) / 32, CiP = CB * 32;
  int NG = ng_max;
  if (NG > B * tx * ty) NG = B * tx * ty;
  for (int co0 = 0; co0 < Cout; co0 += 96) {
    const int CoN = Cout - co0 < 96 ? Cout - co0 : 96, MT = mtiles(CoN), CoP = MT * 32;
    float* part = ws;
    float* part_b = ws + (size_t)NG * CoP * CiP * ntap;
    const dim3 grid(NG, CB);
    if (ntap == 1) launch_wgrad_mt<1, 1>(MT, grid, s, G, A, part, part_b, Cout, co0, Cin_a, B, H2, W2, tx, ty);
    else if (dil == 1) launch_wgrad_mt<9, 1>(MT, grid, s, G, A, part, part_b, Cout, co0, Cin_a, B, H2, W2, tx, ty);
    else if (dil == 2) launch_wgrad_mt<9, 2>(MT, grid, s, G, A, part, part_b, Cout, co0, Cin_a, B, H2, W2, tx, ty);
    else if (dil == 3) launch_wgrad_mt<9, 3>(MT, grid, s, G, A, part, part_b, Cout, co0, Cin_a, B, H2, W2, tx, ty);
    else launch_wgrad_mt<9, 4>(MT, grid, s, G, A, part, part_b, Cout, co0, Cin_a, B, H2, W2, tx, ty);
    DPX_LAUNCH("k_wgrad_reduce", k_wgrad_reduce, dim3(grid_for((long)CoN * Cin_w * ntap + CoN, 256, 1024)), dim3(256), 0, s, (const float*)part,
               (const float*)part_b, gw, gb, NG, CoN, co0, Cin_w, CoP, CiP, ntap, (const float*)nullptr);
  }
}

// (for the split-arithmetic stack's training variant, dpx_conv_bf16.hip: the same weight-gradient kernels on planar copies of its C8 planes)
namespace dpx {
size_t ffd_wgrad_ws_floats(int nc, int in_nc) { return wgrad_ws_floats(nc, in_nc); }
void ffd_launch_wgrad(const float* G, const float* A, float* gw, float* gb, int Cout, int Cin_w, int Cin_a, int B, int H2, int W2, float* ws,
                      hipStream_t s) {
  launch_wgrad(G, A, gw, gb, Cout, Cin_w, Cin_a, B, H2, W2, ws, s);
}
void ffd_launch_wgrad_reduce(const float* part, const float* part_b, float* gw, float* gb, int NG, int CoN, int co0, int Cin, int CoP, int CiP,
                             const float* mul, hipStream_t s) {
  DPX_LAUNCH("k_wgrad_reduce", k_wgrad_reduce, dim3(grid_for((long)CoN * Cin * 9 + CoN, 256, 1024)), dim3(256), 0, s, part, part_b, gw, gb, NG, CoN, co0,
             Cin, CoP, CiP, 9, mul);
}
}  // namespace dpx

// ---- training variants: forward that keeps every layer's output, backward-data through the whole stack ----------------
extern "C" size_t dpx_ffdnet_acts_bytes(int B, int in_nc, int nc, int nb, int H, int W) {
  const size_t H2 = (H + 1) / 2, W2 = (W + 1) / 2, px = (size_t)B * H2 * W2;
  return (px * pad_even(4 * in_nc + 1) + (size_t)(nb - 1) * px * nc + px * 4 * in_nc) * sizeof(float);
}

extern "C" int dpx_ffdnet_forward_save(const float* x, float* y, const float* sigma, const void* packed, int in_nc, int nc, int nb,
                                       int B, int H, int W, void* acts, dpx_stream_t stream) {
  DPX_REQUIRE(x && y && sigma && packed && acts, "dpx_ffdnet_forward_save: null pointer");
  DPX_REQUIRE((size_t)FFD_CK * ((H + 1) / 2) * ((W + 1) / 2) < ((size_t)1 << 28), "dpx_ffdnet_forward_save: plane %dx%d too large for the 28-bit staging offsets", H, W);
  DPX_REQUIRE(B > 0 && H > 0 && W > 0 && in_nc > 0 && nb >= 2 && nc % 2 == 0 && nc <= 96 && 4 * in_nc <= 96,
              "dpx_ffdnet_forward_save: unsupported configuration (in_nc=%d nc=%d nb=%d)", in_nc, nc, nb);
  hipStream_t s = (hipStream_t)stream;
  const int H2 = (H + 1) / 2, W2 = (W + 1) / 2, Cp = pad_even(4 * in_nc + 1);
  const size_t px = (size_t)B * H2 * W2;
  float* a0 = (float*)acts;
  float* hidden = a0 + px * Cp;                          // nb-1 buffers of px*nc
  float* last = hidden + (size_t)(nb - 1) * px * nc;
  DPX_LAUNCH("k_ffd_pack_in", k_ffd_pack_in, dim3(grid_for((long)(px * Cp), 256, 8192)), dim3(256), 0, s, x, sigma, a0, B, in_nc, H, W,
             H2, W2, Cp);
  const float* wl = (const float*)packed;
  const float* cur = a0;
  for (int l = 0; l < nb; ++l) {
    const int cin = layer_cin(l, in_nc, nc), cout = layer_cout(l, in_nc, nc, nb);
    float* dst = (l == nb - 1) ? last : hidden + (size_t)l * px * nc;
    const bool relu = l != nb - 1;
    switch (mtiles(cout)) {
      case 1: launch_conv<1>(relu, cur, dst, wl, pad_even(cin), cout, B, H2, W2, s); break;
      case 2: launch_conv<2>(relu, cur, dst, wl, pad_even(cin), cout, B, H2, W2, s); break;
      default: launch_conv<3>(relu, cur, dst, wl, pad_even(cin), cout, B, H2, W2, s); break;
    }
    wl += layer_floats(cin, cout);
    cur = dst;
  }
  DPX_LAUNCH("k_ffd_unpack_out", k_ffd_unpack_out, dim3(grid_for((long)B * in_nc * H * W, 256, 8192)), dim3(256), 0, s, last, y, B,
             in_nc, H, W, H2, W2);
  return launch_status("dpx_ffdnet_forward_save");
}

// transposed layer l: input channels = forward cout (padded even), output channels = forward cin
static size_t layer_floats_T(int l, int in_nc, int nc, int nb) { return layer_floats(layer_cout(l, in_nc, nc, nb), layer_cin(l, in_nc, nc)); }

extern "C" size_t dpx_ffdnet_packed_transposed_bytes(int in_nc, int nc, int nb) {
  size_t n = 0;
  for (int l = 0; l < nb; ++l) n += layer_floats_T(l, in_nc, nc, nb);
  return n * sizeof(float) + 1024;
}

extern "C" int dpx_ffdnet_pack_transposed(void* packed_T, const float* const* w, int in_nc, int nc, int nb, dpx_stream_t stream) {
  DPX_REQUIRE(packed_T && w && in_nc > 0 && nc > 0 && nb >= 2, "dpx_ffdnet_pack_transposed: bad arguments");
  DPX_REQUIRE(nc % 2 == 0 && nc <= 96 && 4 * in_nc <= 96 && (4 * in_nc) % 2 == 0, "dpx_ffdnet_pack_transposed: unsupported channel counts");
  float* dst = (float*)packed_T;
  for (int l = 0; l < nb; ++l) {
    const int cin_f = layer_cin(l, in_nc, nc), cout_f = layer_cout(l, in_nc, nc, nb);
    DPX_REQUIRE(w[l], "dpx_ffdnet_pack_transposed: layer %d has null weights", l);
    const size_t n = layer_floats(cout_f, cin_f);
    DPX_LAUNCH("k_ffd_pack_weights", k_ffd_pack_weights, dim3(grid_for((long)n, 256, 1024)), dim3(256), 0, (hipStream_t)stream, w[l],
               (const float*)nullptr, dst, cout_f, cin_f, 1);
    dst += n;
  }
  return launch_status("dpx_ffdnet_pack_transposed");
}

extern "C" size_t dpx_ffdnet_bwd_ws_bytes(int B, int in_nc, int nc, int H, int W) {
  const size_t H2 = (H + 1) / 2, W2 = (W + 1) / 2, px = (size_t)B * H2 * W2;
  return (px * 4 * in_nc + 2 * px * nc + px * pad_even(4 * in_nc + 1) + wgrad_ws_floats(nc, in_nc)) * sizeof(float);
}

extern "C" int dpx_ffdnet_backward(const float* gy, float* gx, float* gsigma, float* const* gw, float* const* gb, const void* packed_T,
                                   const void* acts, int in_nc, int nc, int nb, int B, int H, int W, void* ws, dpx_stream_t stream) {
  DPX_REQUIRE(gy && packed_T && acts && ws && (gx || gsigma || gw), "dpx_ffdnet_backward: null pointer");
  DPX_REQUIRE(!gw == !gb, "dpx_ffdnet_backward: weight and bias gradients come together");
  DPX_REQUIRE((size_t)FFD_CK * ((H + 1) / 2) * ((W + 1) / 2) < ((size_t)1 << 28), "dpx_ffdnet_backward: plane %dx%d too large for the 28-bit staging offsets", H, W);
  DPX_REQUIRE(B > 0 && H > 0 && W > 0 && in_nc > 0 && nb >= 2 && nc % 2 == 0 && nc <= 96 && 4 * in_nc <= 96,
              "dpx_ffdnet_backward: unsupported configuration (in_nc=%d nc=%d nb=%d)", in_nc, nc, nb);
  hipStream_t s = (hipStream_t)stream;
  const int H2 = (H + 1) / 2, W2 = (W + 1) / 2, Cp = pad_even(4 * in_nc + 1);
  const size_t px = (size_t)B * H2 * W2;
  const float* a0 = (const float*)acts;
  const float* hidden = a0 + px * Cp;
  float* g_last = (float*)ws;
  float* gA = g_last + px * 4 * in_nc;
  float* gB = gA + px * nc;
  float* g_a0 = gB + px * nc;
  float* wg_ws = g_a0 + px * Cp;
  DPX_LAUNCH("k_ffd_pack_gout", k_ffd_pack_gout, dim3(grid_for((long)(px * 4 * in_nc), 256, 8192)), dim3(256), 0, s, gy, g_last, B, in_nc,
             H, W, H2, W2);
  // offsets of the transposed layers inside packed_T (stored in forward order)
  const float* wt = (const float*)packed_T;
  size_t off[64];
  DPX_REQUIRE(nb <= 64, "dpx_ffdnet_backward: at most 64 layers");
  size_t o = 0;
  for (int l = 0; l < nb; ++l) { off[l] = o; o += layer_floats_T(l, in_nc, nc, nb); }
  const float* cur = g_last;
  const bool need_data = gx || gsigma;
  for (int l = nb - 1; l >= 0; --l) {
    const int cin_t = layer_cout(l, in_nc, nc, nb), cout_t = layer_cin(l, in_nc, nc);   // transposed layer: cin_t -> cout_t channels
    if (gw && gw[l]) {
      // `cur` is the gradient w.r.t. forward layer l's pre-activation output (ReLU mask already applied by the epilogue
      // of transposed layer l+1); the layer's input is a_l
      const float* a_l = (l == 0) ? a0 : hidden + (size_t)(l - 1) * px * nc;
      launch_wgrad(cur, a_l, gw[l], gb[l], cin_t, cout_t, (l == 0) ? Cp : cout_t, B, H2, W2, wg_ws, s);
    }
    if (l == 0 && !need_data) break;
    float* dst = (l == 0) ? g_a0 : (((nb - 1 - l) & 1) ? gB : gA);
    // the output of transposed layer l is the gradient w.r.t. a_l, the (post-ReLU) output of forward layer l-1: it is
    // stored already multiplied by [a_l > 0] (saved activation hidden[l-1]), ready to be the next layer's plain input
    const float* mask = (l >= 1) ? hidden + (size_t)(l - 1) * px * nc : nullptr;
    // layer 0 writes Cp channels (13 -> 14 padded: the pad channel's packed weights are zero)
    const int cout_w = (l == 0) ? Cp : cout_t;
    switch (mtiles(cout_t)) {
      case 1: launch_conv<1>(false, cur, dst, wt + off[l], pad_even(cin_t), cout_w, B, H2, W2, s, mask); break;
      case 2: launch_conv<2>(false, cur, dst, wt + off[l], pad_even(cin_t), cout_w, B, H2, W2, s, mask); break;
      default: launch_conv<3>(false, cur, dst, wt + off[l], pad_even(cin_t), cout_w, B, H2, W2, s, mask); break;
    }
    cur = dst;
  }
  if (gx)
    DPX_LAUNCH("k_ffd_unpack_gin", k_ffd_unpack_gin, dim3(grid_for((long)B * in_nc * H * W, 256, 8192)), dim3(256), 0, s, g_a0, gx, B,
               in_nc, H, W, H2, W2, Cp);
  if (gsigma) DPX_LAUNCH("k_ffd_sigma_grad", k_ffd_sigma_grad, dim3(B), dim3(256), 0, s, g_a0, gsigma, in_nc, H2, W2, Cp);
  return launch_status("dpx_ffdnet_backward");
}

// ---- generic convolution layers on the same kernel (DRUNet-style residual U-Nets behind deep_prior) ----------------------
namespace dpx {
static int conv_block_width(int cout) {                  // output channels per workgroup (blockIdx.z blocks)
  if (cout <= 96) return mtiles(cout) * 32;
  return (cout % 64 == 0) ? 64 : 96;
}
static size_t conv_block_floats(int cin, int m32, int taps) { return (size_t)(pad_even(cin) / 2) * taps * 2 * m32 + m32 + FFD_ZPAD; }

// w [cout][cin][taps] -> blocks of m32 output channels, each [cin/2][taps][2][m32] + bias[m32] + zero pad
__global__ void k_conv_pack(const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ dst, int cin, int cout, int taps,
                            int m32, int nblk) {
  const long per = (long)(((cin + 1) & ~1) / 2) * taps * 2 * m32, blk = per + m32 + FFD_ZPAD;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < blk * nblk; i += (long)gridDim.x * blockDim.x) {
    const int kb = (int)(i / blk);
    const long r0 = i - (long)kb * blk;
    float v = 0.f;
    if (r0 < per) {
      const int cl = (int)(r0 % m32);
      long r = r0 / m32;
      const int half = (int)(r % 2);
      r /= 2;
      const int tap = (int)(r % taps), cp = (int)(r / taps);
      const int ci = 2 * cp + half, co = kb * m32 + cl;
      if (co < cout && ci < cin) v = w[((long)co * cin + ci) * taps + tap];
    } else if (r0 < per + m32) {
      const int co = kb * m32 + (int)(r0 - per);
      if (b && co < cout) v = b[co];
    }
    dst[i] = v;
  }
}

__global__ void k_space_to_depth(const float* __restrict__ x, float* __restrict__ y, int B, int C, int H, int W) {
  const int H2 = H / 2, W2 = W / 2;
  const long total = (long)B * C * H * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x2 = (int)(i % W2);
    long r = i / W2;
    const int y2 = (int)(r % H2);
    r /= H2;
    const int ch = (int)(r % (4 * C)), b = (int)(r / (4 * C));
    const int c = ch >> 2, dy = (ch >> 1) & 1, dx = ch & 1;
    y[i] = x[(((long)b * C + c) * H + 2 * y2 + dy) * W + 2 * x2 + dx];
  }
}
__global__ void k_depth_to_space(const float* __restrict__ x, float* __restrict__ y, int B, int C, int H, int W) {
  const int Ho = 2 * H, Wo = 2 * W;
  const long total = (long)B * C * Ho * Wo;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int xx = (int)(i % Wo);
    long r = i / Wo;
    const int yy = (int)(r % Ho);
    r /= Ho;
    const int c = (int)(r % C), b = (int)(r / C);
    y[i] = x[(((long)b * 4 * C + c * 4 + (yy & 1) * 2 + (xx & 1)) * H + (yy >> 1)) * W + (xx >> 1)];
  }
}

template <int MT, int NTAP, int DIL = 1>
static void launch_conv_generic(int relu, const float* in, float* out, const float* wpk, const float* res, int Cin, int Cout, int nblk, int B,
                                int H, int W, hipStream_t s, float slope = 0.f) {
  constexpr int NR = 2;
  const int tx = (W + FFD_TW - 1) / FFD_TW, ty = (H + 4 * NR - 1) / (4 * NR);
  const dim3 grid(tx * ty, B, nblk);
  if constexpr (DIL == 1) {
    if (res) {
      DPX_LAUNCH("k_conv_mfma", (k_conv3x3_mfma<MT, false, false, NTAP, true, 1, NR>), grid, dim3(256), 0, s, in, out, wpk, Cin, Cout, H, W, tx, res, 0.f);
      return;
    }
  }
  if (relu)
    DPX_LAUNCH("k_conv_mfma", (k_conv3x3_mfma<MT, true, false, NTAP, false, DIL, NR>), grid, dim3(256), 0, s, in, out, wpk, Cin, Cout, H, W, tx, (const float*)nullptr, slope);
  else
    DPX_LAUNCH("k_conv_mfma", (k_conv3x3_mfma<MT, false, false, NTAP, false, DIL, NR>), grid, dim3(256), 0, s, in, out, wpk, Cin, Cout, H, W, tx, (const float*)nullptr, 0.f);
}
template <int MT>
static void launch_conv_dilated(int dil, int relu, const float* in, float* out, const float* wpk, int Cin, int Cout, int nblk, int B, int H,
                                int W, hipStream_t s) {
  switch (dil) {
    case 2: launch_conv_generic<MT, 9, 2>(relu, in, out, wpk, nullptr, Cin, Cout, nblk, B, H, W, s); break;
    case 3: launch_conv_generic<MT, 9, 3>(relu, in, out, wpk, nullptr, Cin, Cout, nblk, B, H, W, s); break;
    default: launch_conv_generic<MT, 9, 4>(relu, in, out, wpk, nullptr, Cin, Cout, nblk, B, H, W, s); break;
  }
}
}  // namespace dpx

extern "C" size_t dpx_conv_packed_bytes(int cin, int cout, int taps) {
  if (cin <= 0 || cout <= 0 || (taps != 9 && taps != 1)) return 0;
  const int m32 = conv_block_width(cout), nblk = (cout + m32 - 1) / m32;
  return (conv_block_floats(cin, m32, taps) * nblk) * sizeof(float) + 1024;
}

extern "C" int dpx_conv_pack(void* packed, const float* w, const float* b, int cin, int cout, int taps, dpx_stream_t stream) {
  DPX_REQUIRE(packed && w && cin > 0 && cout > 0 && (taps == 9 || taps == 1), "dpx_conv_pack: bad arguments (taps must be 9 or 1)");
  const int m32 = conv_block_width(cout), nblk = (cout + m32 - 1) / m32;
  const long n = (long)conv_block_floats(cin, m32, taps) * nblk;
  DPX_LAUNCH("k_conv_pack", k_conv_pack, dim3(grid_for(n, 256, 4096)), dim3(256), 0, (hipStream_t)stream, w, b, (float*)packed, cin, cout, taps,
             m32, nblk);
  return launch_status("dpx_conv_pack");
}

static int conv2d_impl(const float* in, float* out, const void* packed, const float* res, int relu, float slope, int cin, int cout, int taps,
                       int dilation, int B, int H, int W, dpx_stream_t stream) {
  DPX_REQUIRE(in && out && packed && B > 0 && H > 0 && W > 0 && cin > 0 && cout > 0, "dpx_conv2d: bad arguments");
  DPX_REQUIRE(dilation >= 1 && dilation <= 4, "dpx_conv2d: dilation must be 1..4 (padding = dilation), got %d", dilation);
  DPX_REQUIRE(dilation == 1 || (taps == 9 && !res), "dpx_conv2d: dilated layers are 3x3 without a fused residual");
  DPX_REQUIRE(taps == 9 || taps == 1, "dpx_conv2d: taps must be 9 (3x3, pad 1) or 1 (1x1)");
  DPX_REQUIRE(cin % 2 == 0, "dpx_conv2d: the input must have an even number of channels (pad with a zero channel), got %d", cin);
  DPX_REQUIRE(!(res && relu), "dpx_conv2d: residual add and ReLU are not combined (ResBlock: conv-ReLU-conv + x)");
  DPX_REQUIRE((size_t)FFD_CK * H * W < ((size_t)1 << 28), "dpx_conv2d: plane %dx%d too large for the 28-bit staging offsets", H, W);
  const int m32 = conv_block_width(cout), nblk = (cout + m32 - 1) / m32;
  hipStream_t s = (hipStream_t)stream;
  const float* wp = (const float*)packed;
  if (dilation > 1) {
    switch (m32 / 32) {
      case 1: launch_conv_dilated<1>(dilation, relu, in, out, wp, cin, cout, nblk, B, H, W, s); break;
      case 2: launch_conv_dilated<2>(dilation, relu, in, out, wp, cin, cout, nblk, B, H, W, s); break;
      default: launch_conv_dilated<3>(dilation, relu, in, out, wp, cin, cout, nblk, B, H, W, s); break;
    }
  } else if (taps == 9) {
    switch (m32 / 32) {
      case 1: launch_conv_generic<1, 9>(relu, in, out, wp, res, cin, cout, nblk, B, H, W, s, slope); break;
      case 2: launch_conv_generic<2, 9>(relu, in, out, wp, res, cin, cout, nblk, B, H, W, s, slope); break;
      default: launch_conv_generic<3, 9>(relu, in, out, wp, res, cin, cout, nblk, B, H, W, s, slope); break;
    }
  } else {
    switch (m32 / 32) {
      case 1: launch_conv_generic<1, 1>(relu, in, out, wp, res, cin, cout, nblk, B, H, W, s); break;
      case 2: launch_conv_generic<2, 1>(relu, in, out, wp, res, cin, cout, nblk, B, H, W, s); break;
      default: launch_conv_generic<3, 1>(relu, in, out, wp, res, cin, cout, nblk, B, H, W, s); break;
    }
  }
  return launch_status("dpx_conv2d");
}

extern "C" int dpx_conv2d(const float* in, float* out, const void* packed, const float* res, int relu, int cin, int cout, int taps,
                          int dilation, int B, int H, int W, dpx_stream_t stream) {
  return conv2d_impl(in, out, packed, res, relu, 0.f, cin, cout, taps, dilation, B, H, W, stream);
}

extern "C" int dpx_conv2d_leaky(const float* in, float* out, const void* packed, float neg_slope, int cin, int cout, int taps, int B, int H,
                                int W, dpx_stream_t stream) {
  DPX_REQUIRE(taps == 9, "dpx_conv2d_leaky: 3x3 layers only");
  return conv2d_impl(in, out, packed, nullptr, 1, neg_slope, cin, cout, taps, 1, B, H, W, stream);
}

extern "C" int dpx_space_to_depth(const float* x, float* y, int B, int C, int H, int W, dpx_stream_t stream) {
  DPX_REQUIRE(x && y && B > 0 && C > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "dpx_space_to_depth: even H, W required");
  DPX_LAUNCH("k_space_to_depth", k_space_to_depth, dim3(grid_for((long)B * C * H * W, 256, 8192)), dim3(256), 0, (hipStream_t)stream, x, y, B, C,
             H, W);
  return launch_status("dpx_space_to_depth");
}

extern "C" int dpx_depth_to_space(const float* x, float* y, int B, int C, int H, int W, dpx_stream_t stream) {
  DPX_REQUIRE(x && y && B > 0 && C > 0 && H > 0 && W > 0, "dpx_depth_to_space: bad arguments");
  DPX_LAUNCH("k_depth_to_space", k_depth_to_space, dim3(grid_for((long)B * C * 4 * H * W, 256, 8192)), dim3(256), 0, (hipStream_t)stream, x, y,
             B, C, H, W);
  return launch_status("dpx_depth_to_space");
}

// ---- weight / bias gradients of one generic layer (trainable U-Net / IRCNN denoisers inside `unroll`) ----------------
static size_t conv_wgrad_floats(int cin, int cout, int taps, int B, int H, int W) {
  const long ntiles = (long)B * ((W + WG_TW - 1) / WG_TW) * ((H + WG_TH - 1) / WG_TH);
  const int NG = wgrad_ng(cin, ntiles), CoP = mtiles(cout < 96 ? cout : 96) * 32, CiP = ((cin + 31) / 32) * 32;
  return (size_t)NG * CoP * CiP * taps + (size_t)NG * CoP * 2;
}
extern "C" size_t dpx_conv2d_wgrad_ws_bytes(int cin, int cout, int taps, int B, int H, int W) {
  return conv_wgrad_floats(cin, cout, taps, B, H, W) * sizeof(float);
}

extern "C" int dpx_conv2d_wgrad(const float* g, const float* a, float* gw, float* gb, int cin, int cout, int taps, int dilation, int B, int H,
                                int W, void* ws, dpx_stream_t stream) {
  DPX_REQUIRE(g && a && ws && (gw || gb), "dpx_conv2d_wgrad: null pointer");
  DPX_REQUIRE(cin > 0 && cout > 0 && B > 0 && H > 0 && W > 0 && (taps == 9 || taps == 1), "dpx_conv2d_wgrad: bad arguments");
  DPX_REQUIRE(dilation >= 1 && dilation <= 4 && (taps == 9 || dilation == 1), "dpx_conv2d_wgrad: dilation 1..4 (3x3 only)");
  const long ntiles = (long)B * ((W + WG_TW - 1) / WG_TW) * ((H + WG_TH - 1) / WG_TH);
  launch_wgrad(g, a, gw, gb, cout, cin, cin, B, H, W, (float*)ws, (hipStream_t)stream, taps, dilation, wgrad_ng(cin, ntiles));
  return launch_status("dpx_conv2d_wgrad");
}
