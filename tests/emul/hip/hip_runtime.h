// TEST INFRASTRUCTURE ONLY.
//
// A tiny functional SIMT emulator: lets the *unchanged* HIP kernel sources under
// delta-prox_amd/csrc/ be compiled for the host (clang++ -x c++ -I tests/emul) and executed
// with one ucontext fiber per GPU thread, so that indexing / barrier / wave-shuffle / MFMA-layout
// mistakes are caught on the CPU-only build container before GPU minutes are spent.
// It shadows <hip/hip_runtime.h>; nothing in the product includes or links it.
//
// Supported: hipLaunchKernelGGL, threadIdx/blockIdx/blockDim/gridDim, __shared__ (static),
// HIP_DYNAMIC_SHARED, __syncthreads, wave64 __shfl*/__ballot/__all/__any, readfirstlane,
// f32 MFMA builtins (32x32x2, 16x16x4) with the gfx950 lane layouts, atomics (serial), float2/4.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#define DPX_EMULATED 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define HIP_DYNAMIC_SHARED(type, var) type* var = reinterpret_cast<type*>(::emul::dyn_smem());

struct dim3 {
  unsigned x, y, z;
  constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct double2 { double x, y; };
struct int2 { int x, y; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
static inline float2 make_float2(float x, float y) { return {x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return {x, y, z, w}; }
static inline double2 make_double2(double x, double y) { return {x, y}; }
static inline int2 make_int2(int x, int y) { return {x, y}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return {x, y}; }

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
#define hipErrorNotReady 600
static inline hipError_t hipStreamQuery(hipStream_t) { return 0; }
static inline hipError_t hipGetLastError() { return 0; }
static inline hipError_t hipPeekAtLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emul"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
#define hipMemcpyDeviceToDevice 3
#define hipMemcpyHostToDevice 1
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }
#define HIP_SYMBOL(x) (&(x))
static inline hipError_t hipMemcpyFromSymbol(void* d, const void* sym, size_t n) { memcpy(d, sym, n); return 0; }
static inline hipError_t hipMemcpyToSymbol(void* sym, const void* s, size_t n) { memcpy(sym, s, n); return 0; }
static inline hipError_t hipGetSymbolAddress(void** p, const void* sym) { *p = (void*)sym; return 0; }

namespace emul {
struct Idx { unsigned x, y, z; };
struct Fiber;
Fiber* cur();
const Idx& tid();
const Idx& bid();
const Idx& bdim();
const Idx& gdim();
int lane();
void launch(const std::function<void()>& body, dim3 grid, dim3 block, size_t shmem);
void sync_block();
void sync_wave();
char* dyn_smem();
uint64_t* wave_slot(int lane, int which);   // 2 x 8-byte exchange slots per lane of the current wave
bool lane_alive(int lane);                  // lane exists in the current wave and has not exited

template <class T> inline T xchg(T v, int src_lane) {
  static_assert(sizeof(T) <= 8, "shuffle payload");
  uint64_t raw = 0;
  memcpy(&raw, &v, sizeof(T));
  *wave_slot(lane(), 0) = raw;
  sync_wave();
  int s = src_lane & 63;
  uint64_t got = lane_alive(s) ? *wave_slot(s, 0) : raw;
  sync_wave();
  T r;
  memcpy(&r, &got, sizeof(T));
  return r;
}
inline unsigned long long ballot(int pred) {
  *wave_slot(lane(), 0) = pred ? 1 : 0;
  sync_wave();
  unsigned long long m = 0;
  for (int l = 0; l < 64; ++l)
    if (lane_alive(l) && *wave_slot(l, 0)) m |= 1ull << l;
  sync_wave();
  return m;
}
}  // namespace emul

#define threadIdx (::emul::tid())
#define blockIdx (::emul::bid())
#define blockDim (::emul::bdim())
#define gridDim (::emul::gdim())
#define warpSize 64

static inline void __syncthreads() { ::emul::sync_block(); }
static inline void __threadfence() {}
static inline void __threadfence_block() {}
static inline void __threadfence_system() {}

template <class T> static inline T __shfl(T v, int src, int width = 64) {
  int l = ::emul::lane();
  return ::emul::xchg(v, (l & ~(width - 1)) | (src & (width - 1)));
}
template <class T> static inline T __shfl_xor(T v, int mask, int width = 64) {
  int l = ::emul::lane();
  int s = l ^ mask;
  if ((s & ~(width - 1)) != (l & ~(width - 1))) s = l;
  return ::emul::xchg(v, s);
}
template <class T> static inline T __shfl_down(T v, unsigned d, int width = 64) {
  int l = ::emul::lane();
  int s = l + (int)d;
  if ((s & ~(width - 1)) != (l & ~(width - 1))) s = l;
  return ::emul::xchg(v, s);
}
template <class T> static inline T __shfl_up(T v, unsigned d, int width = 64) {
  int l = ::emul::lane();
  int s = l - (int)d;
  if (s < 0 || (s & ~(width - 1)) != (l & ~(width - 1))) s = l;
  return ::emul::xchg(v, s);
}
static inline unsigned long long __ballot(int p) { return ::emul::ballot(p); }
static inline int __all(int p) { unsigned long long m = ::emul::ballot(!p); return m == 0; }
static inline int __any(int p) { return ::emul::ballot(p) != 0; }
template <class T> static inline T emul_readfirstlane(T v) {
  unsigned long long m = ::emul::ballot(1);
  return ::emul::xchg(v, __builtin_ctzll(m));
}
#define __builtin_amdgcn_readfirstlane(x) emul_readfirstlane(x)
#define __builtin_amdgcn_s_barrier() ::emul::sync_block()
#define __builtin_amdgcn_wave_barrier() ::emul::sync_wave()
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_s_sleep(x) ((void)0)

// ---- f32 MFMA (gfx950 layouts, cdna_hip_programming.md section 3) -------------------------------
typedef float emul_f32x16 __attribute__((ext_vector_type(16)));
typedef float emul_f32x4 __attribute__((ext_vector_type(4)));
static inline emul_f32x16 emul_mfma_32x32x2(float a, float b, emul_f32x16 c) {
  int l = ::emul::lane();
  memcpy(::emul::wave_slot(l, 0), &a, 4);
  memcpy(::emul::wave_slot(l, 1), &b, 4);
  ::emul::sync_wave();
  int j = l & 31, h = l >> 5;
  for (int r = 0; r < 16; ++r) {
    int i = (r & 3) + 8 * (r >> 2) + 4 * h;
    float acc = c[r];
    for (int k = 0; k < 2; ++k) {
      float av, bv;
      memcpy(&av, ::emul::wave_slot(k * 32 + i, 0), 4);   // A[i][k] lives in lane k*32+i
      memcpy(&bv, ::emul::wave_slot(k * 32 + j, 1), 4);   // B[k][j] lives in lane k*32+j
      acc = fmaf(av, bv, acc);
    }
    c[r] = acc;
  }
  ::emul::sync_wave();
  return c;
}
static inline emul_f32x4 emul_mfma_16x16x4(float a, float b, emul_f32x4 c) {
  int l = ::emul::lane();
  memcpy(::emul::wave_slot(l, 0), &a, 4);
  memcpy(::emul::wave_slot(l, 1), &b, 4);
  ::emul::sync_wave();
  int j = l & 15, q = l >> 4;
  for (int r = 0; r < 4; ++r) {
    int i = q * 4 + r;
    float acc = c[r];
    for (int k = 0; k < 4; ++k) {
      float av, bv;
      memcpy(&av, ::emul::wave_slot(k * 16 + i, 0), 4);
      memcpy(&bv, ::emul::wave_slot(k * 16 + j, 1), 4);
      acc = fmaf(av, bv, acc);
    }
    c[r] = acc;
  }
  ::emul::sync_wave();
  return c;
}
// v_mfma_f32_32x32x16_bf16: lane l holds A[m = l % 32][k = 8 (l / 32) .. + 8) and B[k = 8 (l / 32) .. + 8)][n = l % 32] as 8 bf16
// (4 dwords, element 2i in the low half of dword i); D as for 32x32x2.  Exchanged through the 8-byte wave slots in two rounds.
static inline float emul_bf16_to_f32(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
static inline emul_f32x16 emul_mfma_32x32x16_bf16(uint4 a, uint4 b, emul_f32x16 c) {
  int l = ::emul::lane();
  int j = l & 31, h = l >> 5;
  const unsigned av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
  for (int round = 0; round < 2; ++round) {
    memcpy(::emul::wave_slot(l, 0), &av[2 * round], 8);
    memcpy(::emul::wave_slot(l, 1), &bv[2 * round], 8);
    ::emul::sync_wave();
    for (int r = 0; r < 16; ++r) {
      int i = (r & 3) + 8 * (r >> 2) + 4 * h;
      float acc = c[r];
      for (int kg = 0; kg < 2; ++kg) {
        unsigned short ae[4], be[4];
        memcpy(ae, ::emul::wave_slot(kg * 32 + i, 0), 8);
        memcpy(be, ::emul::wave_slot(kg * 32 + j, 1), 8);
        for (int e = 0; e < 4; ++e) acc = fmaf(emul_bf16_to_f32(ae[e]), emul_bf16_to_f32(be[e]), acc);
      }
      c[r] = acc;
    }
    ::emul::sync_wave();
  }
  return c;
}
// v_mfma_f32_32x32x16_f16: the same operand layout with IEEE half elements
static inline float emul_f16_to_f32(unsigned short h) { _Float16 v; memcpy(&v, &h, 2); return (float)v; }
static inline emul_f32x16 emul_mfma_32x32x16_f16(uint4 a, uint4 b, emul_f32x16 c) {
  int l = ::emul::lane();
  int j = l & 31, h = l >> 5;
  const unsigned av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
  for (int round = 0; round < 2; ++round) {
    memcpy(::emul::wave_slot(l, 0), &av[2 * round], 8);
    memcpy(::emul::wave_slot(l, 1), &bv[2 * round], 8);
    ::emul::sync_wave();
    for (int r = 0; r < 16; ++r) {
      int i = (r & 3) + 8 * (r >> 2) + 4 * h;
      float acc = c[r];
      for (int kg = 0; kg < 2; ++kg) {
        unsigned short ae[4], be[4];
        memcpy(ae, ::emul::wave_slot(kg * 32 + i, 0), 8);
        memcpy(be, ::emul::wave_slot(kg * 32 + j, 1), 8);
        for (int e = 0; e < 4; ++e) acc = fmaf(emul_f16_to_f32(ae[e]), emul_f16_to_f32(be[e]), acc);
      }
      c[r] = acc;
    }
    ::emul::sync_wave();
  }
  return c;
}
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) emul_mfma_32x32x2(a, b, c)
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) emul_mfma_16x16x4(a, b, c)

// ---- atomics (fibers are serial) ------------------------------------------------------------------
template <class T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <class T> static inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> static inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }

static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }

// ---- math ---------------------------------------------------------------------------------------
static inline void sincospi(double x, double* s, double* c) {
  double r = std::fmod(x, 2.0);
  if (r < 0) r += 2.0;
  // exact values at multiples of 1/2
  if (r == 0.0) { *s = 0; *c = 1; return; }
  if (r == 0.5) { *s = 1; *c = 0; return; }
  if (r == 1.0) { *s = 0; *c = -1; return; }
  if (r == 1.5) { *s = -1; *c = 0; return; }
  *s = std::sin(M_PI * r);
  *c = std::cos(M_PI * r);
}
static inline void sincospif(float x, float* s, float* c) { double ds, dc; sincospi((double)x, &ds, &dc); *s = (float)ds; *c = (float)dc; }
static inline double cospi(double x) { double s, c; sincospi(x, &s, &c); return c; }
static inline float rsqrtf(float x) { return 1.0f / std::sqrt(x); }
static inline float __fdividef(float a, float b) { return a / b; }
static inline float __frcp_rn(float a) { return 1.0f / a; }

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  ::emul::launch([=]() { kernel(__VA_ARGS__); }, dim3(grid), dim3(block), (size_t)(shmem))

// ---- misc runtime shims ---------------------------------------------------------------------------
#define hipFuncAttributeMaxDynamicSharedMemorySize 8
static inline hipError_t hipFuncSetAttribute(const void*, int, int) { return 0; }
static inline int __mul24(int a, int b) { return a * b; }
template <class T> static inline T min(T a, T b) { return a < b ? a : b; }
template <class T> static inline T max(T a, T b) { return a > b ? a : b; }

// ---- events (timing is meaningless under emulation) --------------------------------------------
typedef int hipEvent_t;
#define hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, e0, e1, flags, ...) \
  hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__)
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = 0; return 0; }
#define hipEventDisableTiming 2
#define hipHostMallocDefault 0
#define hipHostMallocMapped 2
#define hipHostMallocCoherent 0x40000000
#define __HIP_MEMORY_SCOPE_SYSTEM 5
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n(p, v, order)
#define hipMemcpyDeviceToHost 2
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, int) { *e = 0; return 0; }
static inline hipError_t hipHostMalloc(void** p, size_t n, int) { *p = malloc(n); return *p ? 0 : 1; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return 0; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return 0; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return 0; }
static inline hipError_t hipHostFree(void* p) { free(p); return 0; }
static inline hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned) { *d = h; return 0; }
