#!/usr/bin/env python
"""GPU box: per-kernel event times of the ADMM TV-deconvolution iteration on one plane size (default 8x3x1000x1000: the size-generic path)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "delta-prox_amd")]
import torch
import dprox as dp, synthetic
from dprox import _backend as be
B, C, H, W = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "8x3x1000x1000").split("x"))
gt, b, psf = synthetic.deconv_case(B, C, H, W, seed=1)
bt = torch.from_numpy(b).cuda()
x = dp.Variable()
s = dp.compile(dp.sum_squares(dp.conv(x, psf) - bt) + dp.norm1(dp.grad(x, dim=0)) + dp.norm1(dp.grad(x, dim=1)), method="admm", device="cuda")
L = be.lib()
def report():
    buf = ctypes.create_string_buffer(1 << 16)
    L.call("dpx_timing_report", buf, len(buf))
    return {ln.split()[0]: (int(ln.split()[1]), float(ln.split()[2])) for ln in buf.value.decode().splitlines() if ln.split()}
for _ in range(2):
    s.solve(x0=bt, rhos=0.1, lams=0.005, max_iter=10)
torch.cuda.synchronize()
L.call("dpx_timing_enable", 1); report()
s.solve(x0=bt, rhos=0.1, lams=0.005, max_iter=20)
torch.cuda.synchronize()
r = report(); L.call("dpx_timing_enable", 0)
tot = sum(t for _, t in r.values())
print(f"{B}x{C}x{H}x{W}: kernels {tot / 20:.4f} ms per iteration (20 iterations), path {s.last_path}")
for k, (c, t) in sorted(r.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:24s} x{c:4d}  avg {1e3 * t / c:8.1f} us  total {t:8.3f} ms  = {1e9 * t / c / (B * C * H * W):6.2f} ps/pixel")
