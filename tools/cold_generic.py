#!/usr/bin/env python
"""GPU box: cold vs warm solve(max_iter=50) on a plane off the register-radix path (default 8x3x1000x1000) with the per-kernel listing of the cold one."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "delta-prox_amd")]
import torch
import dprox as dp, synthetic
from dprox import _backend as be
B, C, H, W = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "8x3x1000x1000").split("x"))
L = be.lib()
def report():
    buf = ctypes.create_string_buffer(1 << 16)
    L.call("dpx_timing_report", buf, len(buf))
    return {ln.split()[0]: (int(ln.split()[1]), float(ln.split()[2])) for ln in buf.value.decode().splitlines() if ln.split()}
def make(seed):
    gt, b, psf = synthetic.deconv_case(B, C, H, W, seed=seed)
    bt = torch.from_numpy(b).cuda()
    x = dp.Variable()
    s = dp.compile(dp.sum_squares(dp.conv(x, psf) - bt) + dp.norm1(dp.grad(x, dim=0)) + dp.norm1(dp.grad(x, dim=1)), method="admm", device="cuda")
    return s, bt
s, bt = make(1)
s.solve(x0=bt, rhos=0.1, lams=0.005, max_iter=50); torch.cuda.synchronize()
for rep in range(2):
    s, bt = make(2 + rep)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s.solve(x0=bt, rhos=0.1, lams=0.005, max_iter=50); torch.cuda.synchronize()
    cold = time.perf_counter() - t0
    t0 = time.perf_counter()
    s.solve(x0=bt, rhos=0.1, lams=0.005, max_iter=50); torch.cuda.synchronize()
    warm = time.perf_counter() - t0
    print(f"{B}x{C}x{H}x{W}: cold solve(50) {cold * 1e3:.2f} ms, warm {warm * 1e3:.2f} ms")
s, bt = make(9)
L.call("dpx_timing_enable", 1); report()
s.solve(x0=bt, rhos=0.1, lams=0.005, max_iter=50); torch.cuda.synchronize()
r = report(); L.call("dpx_timing_enable", 0)
for k, (c, t) in sorted(r.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:24s} x{c:4d}  avg {1e3 * t / c:8.1f} us  total {t:8.3f} ms")
