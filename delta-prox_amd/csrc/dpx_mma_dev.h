// Operand splits and matrix-instruction wrappers of the split-arithmetic kernels (dpx_conv_bf16.hip, dpx_wgrad_c8.hip):
// "split-bf16" x = hi + mid + lo (three exact bf16 parts, six products) and "split-f16" x = hi + lo / 2^11 (two binary16 parts, three products).
#pragma once
#include "dpx_common.h"

namespace dpx {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// exact three-way split by truncation; every part is returned as fp32 bits whose low 16 bits are zero
__device__ __forceinline__ void split3(float x, unsigned& h, unsigned& m, unsigned& l) {
  h = __float_as_uint(x) & 0xffff0000u;
  const float r1 = x - __uint_as_float(h);
  m = __float_as_uint(r1) & 0xffff0000u;
  l = __float_as_uint(r1 - __uint_as_float(m)) & 0xffff0000u;
}
__device__ __forceinline__ unsigned pack_hi16(unsigned lo_elem, unsigned hi_elem) { return (hi_elem & 0xffff0000u) | (lo_elem >> 16); }
// round-to-nearest-even bf16 (MODE = 1, plain bf16 operands), as fp32 bits with a zero low half
__device__ __forceinline__ unsigned bf16_rne(float x) {
  const unsigned u = __float_as_uint(x);
  return (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
}

// MODE = 3 ("split-f16"): x = hi + lo / 2048 with hi = half(x) (11 significant bits, round to nearest) and lo = half((x - hi) * 2048)
// (the next 11 bits, scaled by 2^11 so that it stays in the normal range of binary16).  Three products  wh ah | wh al | wl ah
// (dropped: wl al <= 2^-22 |w a|); the two cross terms go to their own accumulator, which enters the result times 2^-11.
// Measured on the FFDNet stack against float64: 9.7e-8 (the f32-input instruction: 1.1e-7, split-bf16: 5.8e-8).  Operands must stay
// below the binary16 range (6.5e4): the split pass counts the values that do not (dpx_ffdnet_f16_overflow).
// Both parts as fp32-style words whose upper 16 bits hold the binary16 pattern (so that pack_hi16 packs them like the bf16 parts).
constexpr float F16_LO_SCALE = 2048.f;
__device__ __forceinline__ unsigned f16_word(float x) {
  const _Float16 h = (_Float16)x;
  unsigned short b;
  __builtin_memcpy(&b, &h, 2);
  return (unsigned)b << 16;
}
__device__ __forceinline__ float f16_word_value(unsigned w) {
  const unsigned short b = (unsigned short)(w >> 16);
  _Float16 h;
  __builtin_memcpy(&h, &b, 2);
  return (float)h;
}
__device__ __forceinline__ void split2_f16(float x, unsigned& h, unsigned& l) {
  h = f16_word(x);
  l = f16_word((x - f16_word_value(h)) * F16_LO_SCALE);
}
// two neighbouring elements at once, as the packed dwords of the operand tile (element 0 in the low half): packed conversions
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2_f16_pair(float a, float b, unsigned& hw, unsigned& lw) {
  const f32x2_t x = {a, b};
  const f16x2_t h = __builtin_convertvector(x, f16x2_t);
  const f32x2_t r = (x - __builtin_convertvector(h, f32x2_t)) * F16_LO_SCALE;
  const f16x2_t l = __builtin_convertvector(r, f16x2_t);
  __builtin_memcpy(&hw, &h, 4);
  __builtin_memcpy(&lw, &l, 4);
}

#ifdef DPX_EMULATED
__device__ inline f32x16 mfma_bf16(uint4 a, uint4 b, f32x16 c) { return emul_mfma_32x32x16_bf16(a, b, c); }
__device__ inline f32x16 mfma_f16(uint4 a, uint4 b, f32x16 c) { return emul_mfma_32x32x16_f16(a, b, c); }
#else
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mfma_f16(uint4 a, uint4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mfma_bf16(uint4 a, uint4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
#endif

}  // namespace dpx
