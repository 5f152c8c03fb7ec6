// Internal helpers shared by the gfx950 kernels of libdpx_hip.so (not part of the C ABI).
#pragma once
#ifndef DPX_COLS_PACK0
#define DPX_COLS_PACK0 1
#endif
#include <hip/hip_runtime.h>
#ifndef DPX_EMULATED
#include <hip/hip_ext.h>
#endif

#include <cstdarg>
#include <cstdio>

#include "../../include/dpx.h"

// Hides a value's provenance from the optimiser (zero instructions): used where common-subexpression
// elimination across kernel phases would keep dozens of addresses alive and spill them.
// DPX_LDS_BARRIER: workgroup barrier that only waits for this wave's LDS traffic (lgkmcnt), not for its
// outstanding global loads / stores (vmcnt) the way __syncthreads() does -- prefetched loads and posted stores
// stay in flight across it.  Only LDS data may be exchanged through it.
#ifdef DPX_EMULATED
#define DPX_OPAQUE(x) ((void)(x))
#define DPX_OPAQUE_AFTER(x, dep) ((void)(x))
#define DPX_LDS_BARRIER() __syncthreads()
#else
#define DPX_OPAQUE(x) asm volatile("" : "+v"(x))
// ... and not before `dep` has been computed (an address derived from x cannot be hoisted across the code producing dep)
#define DPX_OPAQUE_AFTER(x, dep) asm volatile("" : "+v"(x) : "v"(dep))
#define DPX_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif

// ---- LDS-DMA (global -> LDS without touching VGPRs) and hand-counted waits -----------------------------
// dpx_glds16 / dpx_glds4: every lane names its own 16 / 4 source bytes; the wave's 64 pieces land lane-linearly at
// lds_wave_base (+ lane * size).  hipcc does not see these loads: the kernel counts them itself with
// dpx_wait_vm<N>() ("all but the N most recent vector-memory operations of this wave have completed"; loads,
// LDS-DMA and stores retire in issue order on gfx9-family vmcnt) before it reads the landed bytes, and calls
// dpx_wait_lds() (this wave's LDS reads have returned) before it re-targets a staging area.
#ifdef DPX_EMULATED
#include <cstring>
template <int NT = 0> __device__ inline void dpx_glds16(const void* g, void* lds_wave_base) { std::memcpy((char*)lds_wave_base + (threadIdx.x & 63) * 16, g, 16); }
template <int NT = 0> __device__ inline void dpx_glds4(const void* g, void* lds_wave_base) { std::memcpy((char*)lds_wave_base + (threadIdx.x & 63) * 4, g, 4); }
// (wave-uniform 64-bit base + per-lane 32-bit byte offset)
__device__ inline void dpx_glds16_s(const void* sbase, unsigned voff, void* lds_wave_base) { std::memcpy((char*)lds_wave_base + (threadIdx.x & 63) * 16, (const char*)sbase + voff, 16); }
template <int N> __device__ inline void dpx_wait_vm() { __builtin_amdgcn_wave_barrier(); }
__device__ inline void dpx_wait_lds() { __builtin_amdgcn_wave_barrier(); }
#else
// Cache policy of streamed accesses.  NT = 1 marks a load `nt` (data read once by one CU: do not keep it in L2 ahead of
// lines that are re-read); stores: ST = 0 plain, 1 `sc1` (write-through), 2 `nt`.  Measured on the config-2 iteration
// (DESIGN.md section 3): `nt` on the row kernel's LDS-DMA streams -11 %, `nt` on the column kernel's stores -7 %.
template <int NT = 0> __device__ __forceinline__ void dpx_glds16(const void* g, void* lds_wave_base) {
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds_wave_base);
  unsigned keep;
  if constexpr (NT)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
}
template <int NT = 0> __device__ __forceinline__ void dpx_glds4(const void* g, void* lds_wave_base) {
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds_wave_base);
  unsigned keep;
  if constexpr (NT)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
}
// the same with a wave-uniform 64-bit base in scalar registers and a per-lane 32-bit byte offset: one address register per lane instead of two
__device__ __forceinline__ void dpx_glds16_s(const void* sbase, unsigned voff, void* lds_wave_base) {
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds_wave_base);
  const unsigned long long b = (unsigned long long)(size_t)sbase;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
  const unsigned long long sb = ((unsigned long long)hi << 32) | lo;
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sb), "s"(dst) : "memory");
}
template <int N> __device__ __forceinline__ void dpx_wait_vm() {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void dpx_wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
#endif

// 1 / x with the hardware reciprocal (1 ulp): the per-frequency scale / denominator of the power-of-two x-update.  The
// correctly rounded division costs ~10 VALU instructions per frequency (v_div_scale / v_rcp / 4 x v_fma / v_div_fmas /
// v_div_fixup); its extra 0.5 ulp is far below the fp32 transform round-off around it.
#ifdef DPX_EMULATED
#define DPX_RCP(x) (1.0f / (x))
#else
#define DPX_RCP(x) __builtin_amdgcn_rcpf(x)
#endif

namespace dpx {

// ---- error reporting (thread-local last error, int status across the ABI) -------------------
void set_error(const char* fmt, ...);
int launch_status(const char* what);   // hipGetLastError() -> DPX_OK / DPX_ERR_LAUNCH

// ---- dpx_cg_masked_fft's start state, for a producer of the right-hand side that writes it itself (dpx_cg.hip) -------------
struct CgStartPtrs {
  float* r;             // residual (= b at the start), [B][H W]
  float* p;             // direction (= 0 at the start)
  int* flags;           // 4 control words: (0, -1, 0, 0) at the start
  unsigned* counters;   // two arrival counters: 0 at the start
};
bool cg_masked_fft_is_fused(int B);
CgStartPtrs cg_masked_fft_start_ptrs(void* ws, int B, int H, int W, int mask_images);
// Work that follows the solve, issued BEFORE the host has seen the stop flag: at the iteration the previous solve of this thread exited at, right
// behind that iteration's stop test, `launch` is called once with the device address of the solve's `done` word -- what it launches must do
// nothing unless *done_flag != 0.  valid: the solve ended exactly there (the launch ran for real); otherwise the caller issues that work again,
// unconditionally, behind the solve.
struct CgSpeculate {
  int (*launch)(void* ctx, const int* done_flag, dpx_stream_t stream);
  void* ctx;
  bool launched = false, valid = false;
  int hint = -1;        // >= 0: the iteration this solve is expected to end at (instead of the thread's previous solve's)
};
int cg_masked_fft_run(float* x, const float* b, const float* mask, int mask_images, const float* rho, float n_identity, float rtol, int max_iters,
                      int B, int H, int W, const void* table, void* ws, bool started, CgSpeculate* spec, dpx_stream_t stream);

// ---- tuning knobs (dpx_tune_set / dpx_tune_get of the C ABI; dpx_core.hip holds the table, include/dpx.h documents it) ----------
enum Tune {
  TUNE_CG_FUSED_MAX_B, TUNE_CG_SPLIT_UPDATE, TUNE_CG_UNFUSED, TUNE_COMM_ALLGATHER_RING, TUNE_ITER_ROWS,
  TUNE_ITER_BAND, TUNE_ITER_R, TUNE_DS_RPB, TUNE_DS_ROW_THREADS,
  TUNE_DS_COL_THREADS, TUNE_CG_ROWS_PER_WG, TUNE_CG_COLS_PER_WG, TUNE_CG_GRAM_SMALL, TUNE_CG_NO_HINT, TUNE_UNROLL_BWD_STAGED, TUNE_UNROLL_BWD_FOLD_FINISH, TUNE_FFDNET_PRESPLIT, TUNE_CG_WAVE_FFT, TUNE_CONV_TILE_ROWS, TUNE_UNROLL_BWD_BAND, TUNE_GENERIC_INTERLEAVED, TUNE_ITER_BAND_MIN_ROWS, TUNE_ITER_PAR_MAX_ROWS, TUNE_UNROLL_BWD_PAR_MAX_ROWS, TUNE_CG_EVENT_WAIT, TUNE_PNP_CG_NO_FOLD, TUNE_COUNT
};
int tune(Tune k);

// ---- optional per-kernel timing (bench.py's roofline leg) ------------------------------------------------
// With timing on, a launch goes through hipExtLaunchKernelGGL with a start / stop event pair attached to the kernel's
// own dispatch packet: their elapsed time is the kernel's execution time (the dispatch timestamps rocprofv3's kernel
// trace reports), not the stream-level interval between two separately recorded events, which also contains the
// two event packets' own dispatch latency (~5-8 us per launch on this stack).
bool timing_events(const char* name, hipEvent_t* start, hipEvent_t* stop);
#define DPX_LAUNCH(name, kernel, grid, block, shmem, stream, ...)                                         \
  do {                                                                                                     \
    hipEvent_t dpx_e0_, dpx_e1_;                                                                           \
    if (::dpx::timing_events(name, &dpx_e0_, &dpx_e1_))                                                    \
      hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, dpx_e0_, dpx_e1_, 0, __VA_ARGS__);         \
    else                                                                                                   \
      hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                                 \
  } while (0)

#define DPX_REQUIRE(cond, ...)          \
  do {                                  \
    if (!(cond)) {                      \
      ::dpx::set_error(__VA_ARGS__);    \
      return DPX_ERR_ARG;               \
    }                                   \
  } while (0)

// ---- streamed global accesses (data written once for the NEXT kernel / read once) -----------------------------------
typedef float dpx_v2f __attribute__((ext_vector_type(2)));
template <int ST> __device__ __forceinline__ void st_stream(float2* p, float2 v) {
#ifdef DPX_EMULATED
  *p = v;
#else
  if constexpr (ST == 1)
    __hip_atomic_store((unsigned long long*)p, __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else if constexpr (ST == 2)
    __builtin_nontemporal_store(__builtin_bit_cast(dpx_v2f, v), (dpx_v2f*)p);
  else
    *p = v;
#endif
}
template <int NT> __device__ __forceinline__ float2 ld_stream(const float2* p) {
#ifdef DPX_EMULATED
  return *p;
#else
  if constexpr (NT) return __builtin_bit_cast(float2, __builtin_nontemporal_load((const dpx_v2f*)p));
  else return *p;
#endif
}

// ---- bf16 history planes of the unrolled iteration (round to nearest even on the way in, exact on the way out) --------------------
__device__ __forceinline__ unsigned dpx_bf16_bits(float x) {
  const unsigned u = __float_as_uint(x);
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
// element pair `idx` of a plane that is fp32 (float2) or bf16 (two 16-bit values in one dword)
__device__ __forceinline__ void dpx_emit_pair(float* plane, int bf16, size_t idx, float2 v) {
  if (bf16) ((unsigned*)plane)[idx] = dpx_bf16_bits(v.x) | (dpx_bf16_bits(v.y) << 16);
  else ((float2*)plane)[idx] = v;
}
// element i of a saved plane that is fp32 or bf16
__device__ __forceinline__ float dpx_hist_load(const float* plane, int bf16, long i) {
  return bf16 ? __uint_as_float((unsigned)((const unsigned short*)plane)[i] << 16) : plane[i];
}

// four consecutive elements (i a multiple of 4) of such a plane
__device__ __forceinline__ float4 dpx_hist_load4(const float* plane, int bf16, long i) {
  if (bf16) {
    const uint2 u = *(const uint2*)((const unsigned short*)plane + i);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
  }
  return *(const float4*)(plane + i);
}

// ---- complex arithmetic on float2 ---------------------------------------------------------------
// gfx950 has packed-fp32 VOP3P instructions (v_pk_add / v_pk_mul / v_pk_fma_f32: one issue slot, both halves of a 64-bit register
// pair) whose per-source op_sel / op_sel_hi (which half feeds the low / the high result) and neg_lo / neg_hi modifiers make a complex
// product two instructions and a butterfly with a +-i rotation one -- hipcc finds the plain packed add / sub but builds every swapped
// or half-negated operand with v_mov / v_pk_mov and falls back to scalar v_mul / v_fma for most products (k_cols_p2: 325 v_mov and
// 420 scalar flops of 1870 vector instructions per wave).  DPX_PK_ASM = 1 writes these few primitives as VOP3P by hand; the C forms
// below them are the same arithmetic in the same rounding order (product of the first components rounded, then one fma), used by the
// host emulator and by -DDPX_PK_ASM=0 builds.
#ifndef DPX_PK_ASM
#ifdef DPX_EMULATED
#define DPX_PK_ASM 0
#else
#define DPX_PK_ASM 1
#endif
#endif
#if DPX_PK_ASM
typedef float dpx_pk2 __attribute__((ext_vector_type(2)));
#define DPX_PK(x) __builtin_bit_cast(dpx_pk2, x)
#define DPX_F2(x) __builtin_bit_cast(float2, x)
#endif
// a * b:   t = (a.x b.x, a.y b.x);  (t.x - a.y b.y, t.y + a.x b.y)
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
#if DPX_PK_ASM
  dpx_pk2 t, d;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(DPX_PK(a)), "v"(DPX_PK(b)));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(d) : "v"(DPX_PK(a)), "v"(DPX_PK(b)), "v"(t));
  return DPX_F2(d);
#else
  return make_float2(fmaf(-a.y, b.y, a.x * b.x), fmaf(a.x, b.y, a.y * b.x));
#endif
}
// a * conj(b):   (t.x + a.y b.y, t.y - a.x b.y)
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) {
#if DPX_PK_ASM
  dpx_pk2 t, d;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(DPX_PK(a)), "v"(DPX_PK(b)));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(d) : "v"(DPX_PK(a)), "v"(DPX_PK(b)), "v"(t));
  return DPX_F2(d);
#else
  return make_float2(fmaf(a.y, b.y, a.x * b.x), fmaf(-a.x, b.y, a.y * b.x));
#endif
}
// a * (cr + i ci) with a compile-time constant held in a scalar register pair (rotations inside the register butterflies)
__device__ __forceinline__ float2 cmul_const(float2 a, float cr, float ci) {
#if DPX_PK_ASM
  dpx_pk2 t, d;
  const dpx_pk2 k = {cr, ci};
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(DPX_PK(a)), "s"(k));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(d) : "v"(DPX_PK(a)), "s"(k), "v"(t));
  return DPX_F2(d);
#else
  return make_float2(fmaf(-a.y, ci, a.x * cr), fmaf(a.x, ci, a.y * cr));
#endif
}
// a + i b = (a.x - b.y, a.y + b.x)   and   a - i b = (a.x + b.y, a.y - b.x): one instruction each
__device__ __forceinline__ float2 cadd_ib(float2 a, float2 b) {
#if DPX_PK_ASM
  dpx_pk2 d;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(DPX_PK(a)), "v"(DPX_PK(b)));
  return DPX_F2(d);
#else
  return make_float2(a.x - b.y, a.y + b.x);
#endif
}
__device__ __forceinline__ float2 csub_ib(float2 a, float2 b) {
#if DPX_PK_ASM
  dpx_pk2 d;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(DPX_PK(a)), "v"(DPX_PK(b)));
  return DPX_F2(d);
#else
  return make_float2(a.x + b.y, a.y - b.x);
#endif
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }
__device__ __forceinline__ float2 cscale(float2 a, float s) { return make_float2(a.x * s, a.y * s); }
// multiply by -i (DIR = -1, forward) or +i (DIR = +1, inverse)
template <int DIR> __device__ __forceinline__ float2 cmul_i(float2 a) {
  return DIR < 0 ? make_float2(a.y, -a.x) : make_float2(-a.y, a.x);
}
// a + w b  and  a - w b  with  w = -i (DIR < 0, forward) / +i (DIR > 0, inverse): the rotation folded into the butterfly
template <int DIR> __device__ __forceinline__ float2 cadd_rot(float2 a, float2 b) { return DIR < 0 ? csub_ib(a, b) : cadd_ib(a, b); }
template <int DIR> __device__ __forceinline__ float2 csub_rot(float2 a, float2 b) { return DIR < 0 ? cadd_ib(a, b) : csub_ib(a, b); }

// ---- mixed-radix plan for one 1-D complex transform (passed by value to kernels) -----------
struct Plan1D {
  int n;
  int nf;
  int radix[20];
};
Plan1D make_plan(int n);

// half-spectrum geometry of an H x W real plane: Ws stored columns; for even W the Nyquist column
// is packed into the imaginary part of column 0 (both are real-valued after the row transform).
static inline int spec_cols(int W) { return (W + 1) / 2; }

// ---- spectral layouts --------------------------------------------------------------------------
// A half spectrum / spectral table of one plane has a main part of H x Ws entries and (even W) a side
// part of H entries for the Nyquist column l = W/2.  Generic sizes: main is row-major [H][Ws] and the
// transform packs the Nyquist bins into the imaginary part of column 0.  Power-of-two planes: main is
// COLUMN-TILE-MAJOR, [Ws/8][H][8] -- each 8-column tile is contiguous, so every stream of the column
// kernel is a linear 512-byte-per-wave-instruction access -- and the Nyquist bins live in the side part.
bool pow2_path_available(int H, int W);
#ifndef DPX_SPEC_TILE
#define DPX_SPEC_TILE 16           // columns per spectrum tile of the power-of-two layout (4, 8 or 16): 16 = one 128-byte line per row
#endif
constexpr int SPEC_TILE = DPX_SPEC_TILE;
__host__ __device__ __forceinline__ size_t spec_main_index(int tiled, int H, int Ws, int k, int l) {
  return tiled ? ((size_t)(l / SPEC_TILE) * H + k) * SPEC_TILE + (l % SPEC_TILE) : (size_t)k * Ws + l;
}
// table = all planes' main parts [C][H*Ws] followed by all side parts [C][H]
// Power-of-two planes, DPX_COLS_PACK0 = 1: the column kernel carries a plane's Nyquist column through its transforms as the
// imaginary part of the DC column (dpx_fft_pow2.hip); the iteration-invariant data spectrum (dpx_data_spectrum) is stored that
// way already -- column 0 of its main part holds A + iB, its side part is unused.  Spectra handed between the row and column
// kernels and the tables keep the side array.
static inline size_t table_elems(int C, int H, int W) { return (size_t)C * H * spec_cols(W) + (size_t)C * H; }

// twiddle table: float2[W] (e^{-2 pi i t / W}) followed by float2[H]
static inline const float2* tw_rows(const void* table) { return (const float2*)table; }
static inline const float2* tw_cols(const void* table, int W) { return (const float2*)table + W; }

// pointwise operator applied between the forward and inverse column transforms
enum SpecOp { OP_MUL = 0, OP_MULCONJ = 1, OP_SOLVE = 2 };
struct SpecArgs {
  const float2* otf;   // OP_MUL / OP_MULCONJ
  const float2* dd;    // OP_SOLVE: interleaved denominators (d0 + c0, d1 + c1); den = dd.x + rho_b * dd.y + eps
  const float* rho;    // OP_SOLVE device [B]
  const float2* add;   // OP_SOLVE (nullable): per-plane data spectrum added before the division
  float eps;
  float eps_num;       // OP_SOLVE: added to the numerator's real part (= eps in the x-update, 0 in its adjoint / backward)
  float scale;         // 1/(H*W)
};

int spectral_apply(const float* x, float* y, int op, const SpecArgs& a, int B, int C, int H, int W,
                   const void* table, void* ws, hipStream_t stream);

// Stencil passes: the hardware deals workgroup i to XCD i % 8, so neighbouring rows of a plane -- handled by neighbouring workgroups --
// land in eight different L2s and every halo row crosses the fabric again (k_zupdate_rhs, 8 x 3 x 1000 x 1000, FETCH_SIZE: 582 MB for 288 MB
// of operands = x thrice, u_0 twice).  xcd_block() renumbers the workgroups of a launch whose size is a multiple of 8 so that every XCD
// walks ONE contiguous eighth of the data: vertical neighbours are then in flight on the same XCD at the same time and hit its L2.
__device__ __forceinline__ unsigned xcd_block() {
  const unsigned nb = gridDim.x, b = blockIdx.x;
  return (nb & 7u) ? b : (b & 7u) * (nb >> 3) + (b >> 3);
}
static inline int grid_for8(long n, int block, int cap = 256 * 8) {      // grid_for, rounded up to a multiple of 8 (xcd_block)
  long g = (n + block - 1) / block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)((g + 7) & ~7L);
}
static inline int grid_for(long n, int block, int cap = 256 * 8) {
  long g = (n + block - 1) / block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace dpx
