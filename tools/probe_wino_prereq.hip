// Stand-alone probes behind the Winograd F(2x2, 3x3) layer kernel.  hipcc --offload-arch=gfx950 -O3.
//  (1) does v_mfma_f32_32x32x16_f16 honour binary16 SUBNORMAL operands (the unscaled low part of a split value can be one)?
//  (2) what LDS-DMA stream rate does a CU sustain from an L2-resident block every CU reads (the transformed weights of a layer:
//      16 positions x 96 x 96 x 2 planes x 2 B = 590 KB), alone and beside a matrix-instruction stream?
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

__global__ void k_subnormal(float* out, float aval, float bval) {
  const int lane = threadIdx.x;
  f16x8_t a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = (_Float16)0.f;
    b[j] = (_Float16)0.f;
  }
  if (lane < 32) {                                       // k-group 0, element 0: k = 0
    a[0] = (_Float16)aval;
    b[0] = (_Float16)bval;
  }
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  if (lane == 0) out[0] = c[0];
}

__device__ __forceinline__ void glds16(const void* g, void* lds_wave_base) {
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds_wave_base);
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
}

// one workgroup of 8 waves per CU; a ring of 3 slots of SLOT bytes; every step lands one slot (SLOT / 8 KB per wave), waits for the slot
// issued two steps before, meets at a barrier, and (MF > 0) issues MF matrix instructions on registers + reads RD 16-byte fragments of the slot
template <int SLOT, int MF, int RD>
__global__ void __launch_bounds__(512, 1) k_stream(const char* __restrict__ src, size_t src_bytes, int steps, float* sink) {
  extern __shared__ char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  constexpr int PPW = SLOT / 1024 / 8;                    // 1 KB pieces per wave and slot
  f32x16 acc[4];
  for (int k = 0; k < 4; ++k)
    for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
  size_t off = 0;
  auto issue = [&](int s) {
    char* dst = smem + (s % 3) * SLOT;
    for (int p = 0; p < PPW; ++p) {
      const int piece = wv * PPW + p;
      glds16(src + off + (size_t)piece * 1024 + lane * 16, dst + piece * 1024);
    }
    off += SLOT;
    if (off + SLOT > src_bytes) off = 0;
  };
  issue(0);
  issue(1);
  for (int s = 0; s < steps; ++s) {
    if (PPW == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (PPW == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    issue(s + 2);
    const char* slot = smem + (s % 3) * SLOT;
    if (MF > 0) {
      f16x8_t fr[RD > 0 ? RD : 1];
      for (int r = 0; r < RD; ++r) fr[r] = *(const f16x8_t*)(slot + ((r * 512 + tid) * 16) % SLOT);
      for (int m = 0; m < MF; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[m % (RD > 0 ? RD : 1)], fr[(m + 1) % (RD > 0 ? RD : 1)], acc[m & 3], 0, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float t = 0.f;
  for (int k = 0; k < 4; ++k)
    for (int i = 0; i < 16; ++i) t += acc[k][i];
  if (t == 123.456f) sink[0] = t;
}

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e = (x);                                                            \
    if (e != hipSuccess) {                                                         \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

template <int SLOT, int MF, int RD>
static void run_stream(const char* src, size_t bytes, float* sink, const char* what) {
  const int steps = 2000, grid = 256;
  CK(hipFuncSetAttribute((const void*)k_stream<SLOT, MF, RD>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * SLOT));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_stream<SLOT, MF, RD>), dim3(grid), dim3(512), 3 * SLOT, 0, src, bytes, steps, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double per_cu = (double)SLOT * steps / (ms * 1e-3);
    if (rep == 2)
      printf("%-58s slot %6d B  %7.3f ms  %6.1f GB/s per CU = %5.1f B/clk at 2.4 GHz, chip %5.2f TB/s, step %5.0f ns%s\n", what, SLOT, ms, per_cu * 1e-9,
             per_cu / 2.4e9, per_cu * grid * 1e-12, ms * 1e6 / steps, MF ? "" : "");
  }
}

int main() {
  float* out;
  CK(hipMalloc(&out, 64));
  const float vals[4] = {9.5367431640625e-07f /* 2^-20 */, 5.9604644775390625e-08f /* 2^-24 */, 3.0517578125e-05f /* 2^-15 */, 6.103515625e-05f /* 2^-14 normal */};
  for (int i = 0; i < 4; ++i) {
    float h;
    hipLaunchKernelGGL(k_subnormal, dim3(1), dim3(64), 0, 0, out, vals[i], 1.0f);
    CK(hipMemcpy(&h, out, 4, hipMemcpyDeviceToHost));
    printf("mfma f16: A = %.10e (%s) x B = 1     -> %.10e  %s\n", vals[i], i < 3 ? "subnormal" : "normal", h, h == vals[i] ? "kept" : "FLUSHED / wrong");
    hipLaunchKernelGGL(k_subnormal, dim3(1), dim3(64), 0, 0, out, 1.0f, vals[i]);
    CK(hipMemcpy(&h, out, 4, hipMemcpyDeviceToHost));
    printf("mfma f16: A = 1 x B = %.10e (%s)     -> %.10e  %s\n", vals[i], i < 3 ? "subnormal" : "normal", h, h == vals[i] ? "kept" : "FLUSHED / wrong");
  }
  hipLaunchKernelGGL(k_subnormal, dim3(1), dim3(64), 0, 0, out, vals[0], vals[0] * 1024.f * 1024.f);     // subnormal x 1: product 2^-20
  float h;
  CK(hipMemcpy(&h, out, 4, hipMemcpyDeviceToHost));
  printf("mfma f16: 2^-20 x 1 -> %.10e\n", h);

  const size_t bytes = 16 * 96 * 96 * 2 * 2;             // 589 824
  char* src;
  CK(hipMalloc(&src, bytes + 65536));
  CK(hipMemset(src, 0, bytes + 65536));
  run_stream<24576, 0, 0>(src, bytes, out, "stream only, slot = 1 nu (24.5 KB), 3 pieces per wave");
  run_stream<49152, 0, 0>(src, bytes, out, "stream only, slot = 2 nu (49 KB), 6 pieces per wave");
  run_stream<24576, 9, 8>(src, bytes, out, "stream + 9 MFMA + 8 ds_read_b128 per wave and step");
  run_stream<24576, 18, 8>(src, bytes, out, "stream + 18 MFMA + 8 ds_read_b128 per wave and step");
  run_stream<49152, 18, 16>(src, bytes, out, "stream (49 KB) + 18 MFMA + 16 ds_read_b128 per wave and step");
  run_stream<49152, 36, 16>(src, bytes, out, "stream (49 KB) + 36 MFMA + 16 ds_read_b128 per wave and step");
  return 0;
}
