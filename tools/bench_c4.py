#!/usr/bin/env python
"""GPU box: config 4 (CS-MRI, LADMM + CG + nonneg + FFDNet-gray) per outer iteration, for a list of batch sizes (default 4 32);
DPX_CG_UNFUSED=1 keeps the step-by-step CG sequence (A/B)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "delta-prox_amd")]
import torch
import dprox as dp, synthetic
from dprox.contrib import masked_fft
from dprox.linalg import LinearSolveConfig
from dprox.proxfn.pnp.denoisers import FFDNetDenoiser
from dprox.utils import ifft2
dev = torch.device("cuda", 0)
for nb in [int(a) for a in sys.argv[1:]] or [4, 32]:
    gt4, mask, y = synthetic.csmri_case(nb, 320, 320, seed=2023)
    mask_d, y_d = torch.from_numpy(mask).to(dev), torch.from_numpy(y).to(dev)
    x = dp.Variable()
    fns = dp.sum_squares(masked_fft(x, mask_d), y_d) + dp.nonneg(x) + dp.deep_prior(x, denoiser=FFDNetDenoiser(synthetic.ffdnet_weights(11, 1, 1, 64, 15)))
    s = dp.compile(fns, method="ladmm", device=dev, linear_solve_config=LinearSolveConfig(rtol=1e-6, max_iters=100))
    x0 = ifft2(y_d).real.contiguous()
    with torch.no_grad():
        for _ in range(3):
            s.solve(x0=x0, rhos=0.5, lams=0.03, max_iter=10)
        best = None
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(3):
                out = s.solve(x0=x0, rhos=0.5, lams=0.03, max_iter=10)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
            best = dt if best is None else min(best, dt)
    print(f"config 4, {nb} x 1 x 320 x 320: {best*1e3:.3f} ms per outer iteration, CG iterations {[int(n) for n in s.least_square.cg_iters[-10:]]}, "
          f"checksum {float(out.double().sum()):.6f}", "UNFUSED" if os.environ.get("DPX_CG_UNFUSED") else "fused")
