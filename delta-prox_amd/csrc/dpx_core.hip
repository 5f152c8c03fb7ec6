// Error plumbing + version of libdpx_hip.so.
#include "dpx_common.h"

#include <ctime>
#include <string>
#include <vector>

namespace dpx {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int launch_status(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: kernel launch failed: %s", what, hipGetErrorString(e));
    return DPX_ERR_LAUNCH;
  }
  return DPX_OK;
}
}  // namespace dpx

namespace dpx {
// ---- per-kernel timing -----------------------------------------------------------------------
struct TimedLaunch { int name; hipEvent_t a, b; };
static bool g_timing = false;
static std::vector<std::string> g_names;
static std::vector<TimedLaunch> g_launches;
static std::vector<hipEvent_t> g_free;

static hipEvent_t get_event() {
  if (!g_free.empty()) { hipEvent_t e = g_free.back(); g_free.pop_back(); return e; }
  hipEvent_t e;
  hipEventCreate(&e);
  return e;
}
bool timing_events(const char* name, hipEvent_t* start, hipEvent_t* stop) {
  if (!g_timing) return false;
  int id = -1;
  for (size_t i = 0; i < g_names.size(); ++i) if (g_names[i] == name) { id = (int)i; break; }
  if (id < 0) { g_names.push_back(name); id = (int)g_names.size() - 1; }
  TimedLaunch t{id, get_event(), get_event()};
  g_launches.push_back(t);
  *start = t.a;
  *stop = t.b;
  return true;
}
}  // namespace dpx

extern "C" int dpx_timing_enable(int on) {
  dpx::g_timing = on != 0;
  return DPX_OK;
}
// writes "name count total_ms\n" lines; resets the log.  Synchronises on the recorded events.
extern "C" int dpx_timing_report(char* buf, size_t cap) {
  using namespace dpx;
  std::vector<double> tot(g_names.size(), 0.0);
  std::vector<long> cnt(g_names.size(), 0);
  for (auto& t : g_launches) {
    hipEventSynchronize(t.b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, t.a, t.b);
    tot[t.name] += ms;
    cnt[t.name] += 1;
    g_free.push_back(t.a);
    g_free.push_back(t.b);
  }
  g_launches.clear();
  size_t off = 0;
  if (buf && cap) buf[0] = 0;
  for (size_t i = 0; i < g_names.size(); ++i) {
    if (!cnt[i] || !buf) continue;
    int n = snprintf(buf + off, off < cap ? cap - off : 0, "%s %ld %.6f\n", g_names[i].c_str(), cnt[i], tot[i]);
    if (n < 0 || off + (size_t)n >= cap) break;
    off += (size_t)n;
  }
  return DPX_OK;
}

// ---- do two streams run concurrently? ---------------------------------------------------------
// HIP maps streams onto a few hardware queues (GPU_MAX_HW_QUEUES, default 4), round-robin in creation order; two streams on ONE
// queue execute strictly one after the other.  The sub-batch chains of the two-kernel iteration need two streams that overlap -- with
// RCCL initialised in the process, the caller's stream and the first side stream were found on the same queue (5500 -> 4300 it/s on
// config 2).  The probe: a single-wave kernel that waits `us` microseconds on the constant 100 MHz clock, launched on both streams;
// the host clocks both completions.  Returns 1 (overlap), 0 (serialised), < 0 on error.  Drains both streams first.
namespace dpx {
__global__ void k_spin(long long ticks) {
#ifndef DPX_EMULATED
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
#endif
}
}  // namespace dpx
extern "C" int dpx_streams_concurrent(dpx_stream_t a, dpx_stream_t b) {
#ifdef DPX_EMULATED
  return 1;
#else
  if (a == b) return 0;
  hipStream_t sa = (hipStream_t)a, sb = (hipStream_t)b;
  const int us = 400;
  if (hipStreamSynchronize(sa) != hipSuccess || hipStreamSynchronize(sb) != hipSuccess) {
    dpx::set_error("dpx_streams_concurrent: hipStreamSynchronize failed");
    return DPX_ERR_LAUNCH;
  }
  // one untimed round first (code object load, queue creation), then the timed one
  double dt = 0.0;
  for (int round = 0; round < 2; ++round) {
    timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    hipLaunchKernelGGL(dpx::k_spin, dim3(1), dim3(64), 0, sa, (long long)us * 100);
    hipLaunchKernelGGL(dpx::k_spin, dim3(1), dim3(64), 0, sb, (long long)us * 100);
    if (hipStreamSynchronize(sa) != hipSuccess || hipStreamSynchronize(sb) != hipSuccess) {
      dpx::set_error("dpx_streams_concurrent: probe kernels failed (%s)", hipGetErrorString(hipGetLastError()));
      return DPX_ERR_LAUNCH;
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    dt = (t1.tv_sec - t0.tv_sec) * 1e6 + (t1.tv_nsec - t0.tv_nsec) * 1e-3;
  }
  return dt < 1.6 * us ? 1 : 0;
#endif
}

extern "C" int dpx_version(void) { return 100; }
extern "C" const char* dpx_last_error(void) { return dpx::g_err; }
