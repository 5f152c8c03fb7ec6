// Register-radix kernels for power-of-two planes (256/512/1024/2048): placeholder until the tuned
// kernels land; the generic LDS path of dpx_fft.hip serves every size meanwhile.
#include "dpx_common.h"

namespace dpx {
bool pow2_path_available(int H, int W) { (void)H; (void)W; return false; }
int spectral_apply_pow2(const float*, float*, int, const SpecArgs&, int, int, int, int, const void*, void*, hipStream_t) {
  set_error("pow2 path not built");
  return DPX_ERR_UNSUPPORTED;
}
}  // namespace dpx
