#!/usr/bin/env python
"""GPU box: the ADMM TV-deconvolution iteration on a FEW planes (default 1x3x1024x1024: what one rank of an 8-way batch-sharded
config 2 runs) under the launch-geometry knobs of its two kernels -- per-kernel event times, wall clock per iteration (difference of
a 60- and a 20-iteration solve) and a bit-identity check of the iterate against the default setting.

    python tools/small_shard_probe.py [BxCxHxW] [setting ...]        setting = knob=value[,knob=value...]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "delta-prox_amd")]
import torch
import dprox as dp, synthetic
from dprox import _backend as be

shape = sys.argv[1] if len(sys.argv) > 1 and "x" in sys.argv[1] else "1x3x1024x1024"
B, C, H, W = (int(v) for v in shape.split("x"))
settings = [a for a in sys.argv[1:] if "=" in a or a == "default"]
if not settings:
    settings = ["default", "iter_band=64", "iter_band=128", "iter_band=256", "iter_band=512,iter_band_min_rows=2", "iter_band=1024,iter_band_min_rows=1",
                "iter_rows=2,iter_r=2", "iter_rows=2,iter_r=6", "iter_rows=2,iter_r=16"]
gt, b, psf = synthetic.deconv_case(B, C, H, W, seed=1)
bt = torch.from_numpy(b).cuda()
x = dp.Variable()
fns = dp.sum_squares(dp.conv(x, psf) - bt) + dp.norm1(dp.grad(x, dim=0)) + dp.norm1(dp.grad(x, dim=1))
if "nonneg" in sys.argv:                                  # TV + nonneg: three terms (the 8-wave forms of the row-parallel kernel)
    fns = fns + dp.nonneg(x)
    print("objective: TV + nonneg (three terms)")
s = dp.compile(fns, method="admm", device="cuda")
L = be.lib()


def report():
    buf = ctypes.create_string_buffer(1 << 16)
    L.call("dpx_timing_report", buf, len(buf))
    return {ln.split()[0]: (int(ln.split()[1]), float(ln.split()[2])) for ln in buf.value.decode().splitlines() if ln.split()}


def wall(n):
    best = None
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.solve(x0=bt, rhos=0.1, lams=0.005, max_iter=n)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return best


ref = None
print(f"{shape}: path after the first solve:", end=" ")
s.solve(x0=bt, rhos=0.1, lams=0.005, max_iter=3)
print(s.last_path)
for st in settings:
    kn = {} if st == "default" else {k: int(v) for k, v in (kv.split("=") for kv in st.split(","))}
    with be.tuned(**kn):
        out = s.solve(x0=bt, rhos=0.1, lams=0.005, max_iter=10).clone()
        torch.cuda.synchronize()
        if ref is None:
            ref = out
        same = bool(torch.equal(out, ref))
        rel = float((out - ref).norm() / ref.norm())
        s.solve(x0=bt, rhos=0.1, lams=0.005, max_iter=60)
        per_it = (wall(60) - wall(20)) / 40
        L.call("dpx_timing_enable", 1); report()
        s.solve(x0=bt, rhos=0.1, lams=0.005, max_iter=20)
        torch.cuda.synchronize()
        r = report(); L.call("dpx_timing_enable", 0)
    ks = "  ".join(f"{k} {1e3 * t / c:.1f}us" for k, (c, t) in sorted(r.items(), key=lambda kv: -kv[1][1]) if c >= 19)
    print(f"{st:48s} {per_it * 1e6:7.2f} us/it   bit-identical {same} (rel {rel:.1e})   {ks}", flush=True)
