#!/usr/bin/env python
"""GPU box: config 4's shard (4 x 1 x 320 x 320) -- wall clock of solve(max_iter = n) for several n (intercept = per-solve host work, slope = one
outer iteration) and a cProfile of ten 10-iteration solves (where the per-solve work is)."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "delta-prox_amd")]
import torch
import dprox as dp, synthetic
from dprox.contrib import masked_fft
from dprox.linalg import LinearSolveConfig
from dprox.proxfn.pnp.denoisers import FFDNetDenoiser
from dprox.utils import ifft2
dev = torch.device("cuda", 0)
gt4, mask, y = synthetic.csmri_case(4, 320, 320, seed=2023)
mask_d, y_d = torch.from_numpy(mask).to(dev), torch.from_numpy(y).to(dev)
x = dp.Variable()
fns = dp.sum_squares(masked_fft(x, mask_d), y_d) + dp.nonneg(x) + dp.deep_prior(x, denoiser=FFDNetDenoiser(synthetic.ffdnet_weights(11, 1, 1, 64, 15)))
s = dp.compile(fns, method="ladmm", device=dev, linear_solve_config=LinearSolveConfig(rtol=1e-6, max_iters=100))
x0 = ifft2(y_d).real.contiguous()
with torch.no_grad():
    for _ in range(3):
        s.solve(x0=x0, rhos=0.5, lams=0.03, max_iter=10)
    res = {}
    for n in (1, 2, 5, 10, 20, 40):
        best = None
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            s.solve(x0=x0, rhos=0.5, lams=0.03, max_iter=n)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        res[n] = best
        print(f"solve(max_iter={n:2d}): {best * 1e3:7.3f} ms  = {best / n * 1e3:.3f} ms per iteration")
    slope = (res[40] - res[10]) / 30
    print(f"slope {slope * 1e3:.4f} ms per outer iteration, intercept {1e3 * (res[10] - 10 * slope):.3f} ms per solve")
    pr = cProfile.Profile(); pr.enable()
    for _ in range(10):
        s.solve(x0=x0, rhos=0.5, lams=0.03, max_iter=10)
    torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)
