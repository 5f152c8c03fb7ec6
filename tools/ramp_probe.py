import os, sys, time
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "delta-prox_amd")]
import torch
import bench
import dprox as dp, synthetic
device = torch.device("cuda", 0)
solver, xvar, b, gt, psf = bench.make_problem(dp, synthetic, 0, device)
x0, rhos, lams, _ = solver.defaults(b, bench.RHO, bench.LAM, 20)
rhos = rhos.to(device); lams = {k: v.to(device) for k, v in lams.items()}
def timed():
    state = solver.initialize(b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    solver.iters(state, rhos, lams, 20)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0)
solver.solve(x0=b, rhos=bench.RHO, lams=bench.LAM, max_iter=5)
print("A: warm 5 ->", [round(timed(), 3) for _ in range(4)])
time.sleep(2.0)
solver.solve(x0=b, rhos=bench.RHO, lams=bench.LAM, max_iter=5)
print("B: idle 2 s, warm 5 ->", [round(timed(), 3) for _ in range(2)])
time.sleep(2.0)
solver.solve(x0=b, rhos=bench.RHO, lams=bench.LAM, max_iter=250)
solver.solve(x0=b, rhos=bench.RHO, lams=bench.LAM, max_iter=5)
print("C: idle 2 s, 250 it + warm 5 ->", [round(timed(), 3) for _ in range(2)])
time.sleep(2.0)
solver.solve(x0=b, rhos=bench.RHO, lams=bench.LAM, max_iter=50)
torch.cuda.synchronize(); time.sleep(0.005)
solver.solve(x0=b, rhos=bench.RHO, lams=bench.LAM, max_iter=5)
print("D: idle 2 s, 50 it, 5 ms idle, warm 5 ->", [round(timed(), 3) for _ in range(2)])
