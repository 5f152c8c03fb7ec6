import sys, os, time
sys.path[:0]=[os.environ["GRAFT_REPO_ROOT"], os.path.join(os.environ["GRAFT_REPO_ROOT"],"delta-prox_amd")]
import torch, numpy as np, cProfile, pstats
import bench
import dprox as dp, synthetic
dev=torch.device("cuda")
solver, xvar, b, gt, psf = bench.make_problem(dp, synthetic, 0, dev)
solver.solve(x0=b, rhos=0.1, lams=0.005, max_iter=5)
x0, rhos, lams, _ = solver.defaults(b, 0.1, 0.005, 50)
rhos=rhos.to(dev); lams={k:v.to(dev) for k,v in lams.items()}
for rep in range(3):
    state=solver.initialize(b); torch.cuda.synchronize()
    t0=time.perf_counter(); solver.iters(state, rhos, lams, int(os.environ.get("K", "2"))); t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    print(f"host return after {1e3*(t1-t0):.3f} ms, GPU done after {1e3*(t2-t0):.3f} ms")
state=solver.initialize(b); torch.cuda.synchronize()
pr=cProfile.Profile(); pr.enable(); solver.iters(state, rhos, lams, int(os.environ.get("K", "2"))); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
