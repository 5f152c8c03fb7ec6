// Error plumbing + version of libdpx_hip.so.
#include "dpx_common.h"

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

extern "C" int dpx_version(void) { return 100; }
extern "C" const char* dpx_last_error(void) { return dpx::g_err; }
