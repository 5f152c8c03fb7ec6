// Error plumbing + version of libdpx_hip.so.
#include "dpx_common.h"

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <mutex>
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

// ---- tuning knobs -----------------------------------------------------------------------------------------------------------
// One registry for every switch that selects between kernel variants / launch geometries (A/B timing, tests that must reach a
// particular branch).  Part of the C ABI (dpx_tune_set / dpx_tune_get / dpx_tune_name, include/dpx.h holds the table): settable per
// process at run time, readable back, so a test can drive both branches of a dispatcher in one process.  The environment variable
// named next to a knob only supplies its INITIAL value (read once, at the first access of the registry).
namespace dpx {
namespace {
struct KnobDef { const char* name; const char* env; int def; const char* env_word; };   // env_word: a string value of `env` that means 1 / 2
const KnobDef kKnobs[TUNE_COUNT] = {
    {"cg_fused_max_b", "DPX_CG_FUSED_MAX_B", 8, nullptr},
    {"cg_split_update", "DPX_CG_SPLIT_UPDATE", 0, nullptr},
    {"cg_unfused", "DPX_CG_UNFUSED", 0, nullptr},
    {"comm_allgather_ring", "DPX_COMM_ALLGATHER", 0, "ring"},
    {"iter_rows", "DPX_ITER_ROWS", 0, "seq,lockstep,par"},
    {"iter_band", "DPX_ITER_BAND", 0, nullptr},
    {"iter_r", "DPX_ITER_R", 0, nullptr},
    {"ds_rpb", "DPX_DS_RPB", 0, nullptr},
    {"ds_row_threads", "DPX_DS_ROW_THREADS", 0, nullptr},
    {"ds_col_threads", "DPX_DS_COL_THREADS", 0, nullptr},
    {"cg_rows_per_wg", "DPX_CG_ROWS_PER_WG", 0, nullptr},
    {"cg_cols_per_wg", "DPX_CG_COLS_PER_WG", 0, nullptr},
    {"cg_gram_small", "DPX_CG_GRAM_SMALL", 0, nullptr},
    {"cg_no_hint", "DPX_CG_NO_HINT", 0, nullptr},
    {"unroll_bwd_staged", "DPX_UNROLL_BWD_STAGED", 0, nullptr},
    {"unroll_bwd_fold_finish", "DPX_UNROLL_BWD_FOLD_FINISH", 0, nullptr},
    {"ffdnet_presplit", "DPX_FFDNET_PRESPLIT", 0, nullptr},
    {"cg_wave_fft", "DPX_CG_WAVE_FFT", 0, nullptr},
    {"conv_tile_rows", "DPX_CONV_TILE_ROWS", 0, nullptr},
    {"unroll_bwd_band", "DPX_UNROLL_BWD_BAND", 0, nullptr},
    {"generic_interleaved", "DPX_GENERIC_INTERLEAVED", 1, nullptr},
    {"iter_band_min_rows", "DPX_ITER_BAND_MIN_ROWS", 0, nullptr},
    {"iter_par_max_rows", "DPX_ITER_PAR_MAX_ROWS", 0, nullptr},
    {"unroll_bwd_par_max_rows", "DPX_UNROLL_BWD_PAR_MAX_ROWS", 0, nullptr},
    {"cg_event_wait", "DPX_CG_EVENT_WAIT", 0, nullptr},
    {"pnp_cg_no_fold", "DPX_PNP_CG_NO_FOLD", 0, nullptr},
};
std::atomic<int> g_knob[TUNE_COUNT];
std::once_flag g_knob_once;
void knobs_init() {
  for (int i = 0; i < TUNE_COUNT; ++i) {
    int v = kKnobs[i].def;
    const char* e = getenv(kKnobs[i].env);
    if (e) {
      if (kKnobs[i].env_word) {                       // "word" -> 1, "w1,w2" -> 1 / 2
        v = 0;
        const char* w = kKnobs[i].env_word;
        int idx = 1;
        while (*w) {
          const char* c = strchr(w, ',');
          const size_t len = c ? (size_t)(c - w) : strlen(w);
          if (strlen(e) == len && !strncmp(e, w, len)) v = idx;
          ++idx;
          w += len + (c ? 1 : 0);
        }
      } else {
        v = *e ? atoi(e) : 1;                         // a flag-style variable set to the empty string counts as on
        if (v == 0 && *e && (e[0] < '0' || e[0] > '9') && e[0] != '-') v = 1;
      }
    }
    g_knob[i].store(v, std::memory_order_relaxed);
  }
}
}  // namespace
int tune(Tune k) {
  std::call_once(g_knob_once, knobs_init);
  return g_knob[k].load(std::memory_order_relaxed);
}
}  // namespace dpx

extern "C" int dpx_tune_count(void) { return dpx::TUNE_COUNT; }
extern "C" const char* dpx_tune_name(int i) { return i >= 0 && i < dpx::TUNE_COUNT ? dpx::kKnobs[i].name : nullptr; }
static int knob_index(const char* name) {
  if (!name) return -1;
  for (int i = 0; i < dpx::TUNE_COUNT; ++i)
    if (!strcmp(name, dpx::kKnobs[i].name)) return i;
  return -1;
}
extern "C" int dpx_tune_set(const char* name, int value) {
  const int i = knob_index(name);
  DPX_REQUIRE(i >= 0, "dpx_tune_set: unknown knob '%s'", name ? name : "(null)");
  std::call_once(dpx::g_knob_once, dpx::knobs_init);
  dpx::g_knob[i].store(value, std::memory_order_relaxed);
  return DPX_OK;
}
extern "C" int dpx_tune_get(const char* name, int* value) {
  const int i = knob_index(name);
  DPX_REQUIRE(i >= 0 && value, "dpx_tune_get: unknown knob '%s'", name ? name : "(null)");
  *value = dpx::tune((dpx::Tune)i);
  return DPX_OK;
}
// the CG solver's switches as one typed call (a negative argument leaves that switch as it is)
extern "C" int dpx_cg_config(int fused_max_b, int split_update, int unfused) {
  DPX_REQUIRE(fused_max_b <= 32, "dpx_cg_config: the fused CG iteration holds at most 32 systems (got %d)", fused_max_b);
  if (fused_max_b >= 0) dpx_tune_set("cg_fused_max_b", fused_max_b);
  if (split_update >= 0) dpx_tune_set("cg_split_update", split_update ? 1 : 0);
  if (unfused >= 0) dpx_tune_set("cg_unfused", unfused ? 1 : 0);
  return DPX_OK;
}

extern "C" int dpx_version(void) { return 102; }
extern "C" const char* dpx_last_error(void) { return dpx::g_err; }
