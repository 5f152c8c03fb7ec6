"""Host-side cost of one 20-iteration solve of bench.py's problem: issue time vs GPU time, when the first kernel is launched,
and a cProfile of Algorithm.iters (GPU only)."""
import os, sys, time, cProfile, pstats
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "delta-prox_amd")]
import torch
import bench
import dprox as dp, synthetic
device = torch.device("cuda", 0)
solver, xvar, b, gt, psf = bench.make_problem(dp, synthetic, 0, device)
solver.solve(x0=b, rhos=bench.RHO, lams=bench.LAM, max_iter=5)
x0, rhos, lams, _ = solver.defaults(b, bench.RHO, bench.LAM, 20)
rhos = rhos.to(device); lams = {k: v.to(device) for k, v in lams.items()}
for rep in range(3):
    state = solver.initialize(b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    state = solver.iters(state, rhos, lams, 20)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"host issue {1e3*(t1-t0):.3f} ms, total {1e3*(t2-t0):.3f} ms")
state = solver.initialize(b)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
state = solver.iters(state, rhos, lams, 20)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)

# ---- where the host time goes before the first kernel of the solve is launched
from dprox import _ops as ops
marks = {}
for name in ("admm_seed_rows", "admm_run"):
    real = getattr(ops, name)
    def wrap(*a, _r=real, _n=name, **k):
        marks.setdefault(_n + "_in", time.perf_counter())
        out = _r(*a, **k)
        marks.setdefault(_n + "_out", time.perf_counter())
        return out
    setattr(ops, name, wrap)
for rep in range(3):
    marks.clear()
    state = solver.initialize(b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    state = solver.iters(state, rhos, lams, 20)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("us since t0:", {k: round(1e6 * (v - t0)) for k, v in marks.items()}, "iters returns", round(1e6 * (t1 - t0)), "gpu done", round(1e6 * (t2 - t0)))

state = solver.initialize(b)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
state = solver.iters(state, rhos, lams, 20)
pr.disable(); torch.cuda.synchronize()
ents = sorted(pr.getstats(), key=lambda e: -e.totaltime)[:40]
for e in ents:
    code = e.code if isinstance(e.code, str) else f"{os.path.basename(e.code.co_filename)}:{e.code.co_firstlineno}({e.code.co_name})"
    print(f"{e.totaltime * 1e6:8.1f} us total {e.inlinetime * 1e6:8.1f} us inline  x{e.callcount:<4d} {code[:90]}")
