#!/usr/bin/env python
"""GPU box: per-kernel times (dispatch-attached HIP events) of config 4's shard solve; DPX_CG_UNFUSED=1 for the step-by-step CG."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "delta-prox_amd")]
import torch
import dprox as dp, synthetic
from dprox import _backend as be
from dprox.contrib import masked_fft
from dprox.linalg import LinearSolveConfig
from dprox.proxfn.pnp.denoisers import FFDNetDenoiser
from dprox.utils import ifft2
dev = torch.device("cuda", 0)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4
gt4, mask, y = synthetic.csmri_case(nb, 320, 320, seed=2023)
mask_d, y_d = torch.from_numpy(mask).to(dev), torch.from_numpy(y).to(dev)
x = dp.Variable()
fns = dp.sum_squares(masked_fft(x, mask_d), y_d) + dp.nonneg(x) + dp.deep_prior(x, denoiser=FFDNetDenoiser(synthetic.ffdnet_weights(11, 1, 1, 64, 15)))
s = dp.compile(fns, method="ladmm", device=dev, linear_solve_config=LinearSolveConfig(rtol=1e-6, max_iters=100))
x0 = ifft2(y_d).real.contiguous()
L = be.lib()
def report():
    buf = ctypes.create_string_buffer(1 << 16)
    L.call("dpx_timing_report", buf, len(buf))
    return {ln.split()[0]: (int(ln.split()[1]), float(ln.split()[2])) for ln in buf.value.decode().splitlines() if ln.split()}
with torch.no_grad():
    for _ in range(2):
        s.solve(x0=x0, rhos=0.5, lams=0.03, max_iter=10)
    torch.cuda.synchronize()
    L.call("dpx_timing_enable", 1); report()
    s.solve(x0=x0, rhos=0.5, lams=0.03, max_iter=10)
    torch.cuda.synchronize()
    r = report(); L.call("dpx_timing_enable", 0)
tot = sum(t for _, t in r.values())
print("UNFUSED" if os.environ.get("DPX_CG_UNFUSED") else "fused", f"total kernel time {tot:.3f} ms per 10 outer iterations")
for k, (c, t) in sorted(r.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:28s} x{c:4d}  avg {1e3*t/c:7.1f} us  total {t:7.3f} ms")
