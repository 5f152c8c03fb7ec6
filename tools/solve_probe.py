#!/usr/bin/env python
"""GPU box: wall clock of back-to-back solve(max_iter=50) calls on config 2 (hot GPU), chains on / off."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "delta-prox_amd")]
import torch
import bench
import dprox as dp, synthetic
dev = torch.device("cuda")
solver, xvar, b, gt, psf = bench.make_problem(dp, synthetic, 0, dev)
for chains in ("2", "1", "2"):
    os.environ["DPX_CHAINS"] = chains
    for _ in range(3):
        solver.solve(x0=b, rhos=0.1, lams=0.005, max_iter=50)
    torch.cuda.synchronize()
    ts = []
    for _ in range(6):
        t0 = time.perf_counter()
        solver.solve(x0=b, rhos=0.1, lams=0.005, max_iter=50)
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    st = solver.initialize(b)
    _, rhos, lams, _ = solver.defaults(b, 0.1, 0.005, 50)
    rd, ld = rhos.to(dev), {k: v.to(dev) for k, v in lams.items()}
    torch.cuda.synchronize()
    t0 = time.perf_counter(); solver.iters(st, rd, ld, 50); torch.cuda.synchronize(); ti = 1e3 * (time.perf_counter() - t0)
    print(f"chains {chains}: solve(50) ms {[round(t, 2) for t in ts]}   initialize + iters(50): {ti:.2f} ms")
