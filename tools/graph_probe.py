"""tuning probe (GPU box): does a captured graph of the C-side ADMM loop shorten the kernel-to-kernel gaps?
config 2 (8x3x1024^2, 50 iterations): plain dpx_admm_run vs the same call captured once and replayed."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "delta-prox_amd")]
import synthetic
import dprox as dp
from dprox import _ops as ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
gt, b, psf = synthetic.deconv_case(B, 3, 1024, 1024, seed=2023)
dev = torch.device("cuda:0")
bt = torch.from_numpy(b).to(dev)
x = dp.Variable()
fns = dp.sum_squares(dp.conv(x, psf) - bt) + dp.norm1(dp.grad(x, dim=0)) + dp.norm1(dp.grad(x, dim=1))
solver = dp.compile(fns, method="admm", device=dev)
orig = ops.admm_run
def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n
def spy(*a):                                        # (inside the solve: its buffers are alive here)
    plain = timed(lambda: orig(*a))
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        orig(*a); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            orig(*a)
        graph = timed(g.replay)
    torch.cuda.synchronize()
    print(f"B={B}: plain {plain / 50 * 1e6:.1f} us/iteration, graph replay {graph / 50 * 1e6:.1f} us/iteration")
    return orig(*a)
ops.admm_run = spy
solver.solve(x0=bt, rhos=0.1, lams=0.005, max_iter=50)
