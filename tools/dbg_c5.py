import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "delta-prox_amd")]
import numpy as np, torch
import dprox as dp, synthetic
dev = torch.device("cuda")
def timed(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): out = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
gt, b, psf = synthetic.deconv_case(4, 3, 512, 512, seed=2023)
bt, gtt = torch.from_numpy(b).to(dev), torch.from_numpy(gt).to(dev)
def build():
    x = dp.Variable()
    n0, n1 = dp.norm1(dp.grad(x, dim=0)), dp.norm1(dp.grad(x, dim=1))
    s = dp.compile(dp.sum_squares(dp.conv(x, psf) - bt) + n0 + n1, method="admm", device=dev)
    s = dp.specialize(s, method="unroll", device=dev, max_iter=10)
    prm = [torch.full((10,), v, requires_grad=True, device=dev) for v in (0.1, 0.005, 0.005)]
    def step():
        for p_ in prm: p_.grad = None
        o = s.solve(x0=bt, rhos=prm[0], lams={n0: prm[1], n1: prm[2]})
        loss = ((o - gtt) ** 2).mean(); loss.backward(); return loss
    return step
print("fresh process:", timed(build()) * 1e3, "ms")
# after a big allocation pattern like the other configs
big = [torch.empty(8, 96, 512, 512, device=dev) for _ in range(4)]; del big
print("after big allocs:", timed(build()) * 1e3, "ms")
from dprox import _backend as be
be.lib().call("dpx_timing_enable", 1); be.lib().call("dpx_timing_enable", 0)
print("after timing toggle:", timed(build()) * 1e3, "ms")
import bench
out = bench.extra_configs(dp, synthetic, dev)
print({k: v.get("ms_per_step") for k, v in out.items() if "config5" in k})
print("after extra_configs:", timed(build()) * 1e3, "ms")
