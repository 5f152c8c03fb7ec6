// FFDNet conv stack (deep_prior z-update) -- reference dprox/proxfn/pnp/denoisers/models/network_ffdnet.py:54-68.
// Placeholder entry points until the MFMA kernel lands: they fail loudly (no fallback).
#include "dpx_common.h"
using namespace dpx;

extern "C" size_t dpx_ffdnet_packed_bytes(int, int, int) { return 0; }
extern "C" int dpx_ffdnet_pack(void*, const float* const*, const float* const*, int, int, int, dpx_stream_t) {
  set_error("dpx_ffdnet_pack: FFDNet kernels not built yet");
  return DPX_ERR_UNSUPPORTED;
}
extern "C" size_t dpx_ffdnet_ws_bytes(int, int, int, int, int) { return 0; }
extern "C" int dpx_ffdnet_forward(const float*, float*, const float*, const void*, int, int, int, int, int, int, void*,
                                  dpx_stream_t) {
  set_error("dpx_ffdnet_forward: FFDNet kernels not built yet");
  return DPX_ERR_UNSUPPORTED;
}
