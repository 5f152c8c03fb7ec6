#!/usr/bin/env python
"""GPU box, variant library built with -DDPX_PAR_TRACE (tools/build_variant.sh par_trace -DDPX_PAR_TRACE; run with DPX_LIB=...): the phase
timeline of the LAST k_gram_small_test launch of a config-4 shard solve (4 x 1 x 320 x 320) -- 100 MHz stamps of thread 0 of every workgroup."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "delta-prox_amd")]
import numpy as np, torch
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
cdll = be.lib().cdll
cdll.dpx_dbg_gram_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
names = ["entry", "first elements back, flag looked at", "element loop done", "partials stored", "ticket taken (last workgroup only from here)",
         "partials of all workgroups added up (G)", "stop test + beta done", "host word stored"]
with torch.no_grad():
    for rep in range(3):
        s.solve(x0=x0, rhos=0.5, lams=0.03, max_iter=6)
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * (256 * 8))()
        assert cdll.dpx_dbg_gram_trace(buf, 256 * 8) == 0
        t = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8).astype(np.float64)
        live = t[:, 0] > 0
        t = t[live]
        t0 = t[:, 0].min()
        us = (t - t0) / 100.0
        print(f"run {rep}: {int(live.sum())} workgroups; us since the first workgroup's entry: mean / min / max")
        for i, nm in enumerate(names):
            col = us[:, i][t[:, i] >= t0] if i < 5 else us[:, i][t[:, i] >= t[:, 4].max() - 1]
            if col.size:
                print(f"   {nm:48s} {col.mean():7.2f} {col.min():7.2f} {col.max():7.2f}   (n={col.size})")
