#!/usr/bin/env python
"""Tuning probe (GPU box): ms per iteration of config 2 (8x3x1024^2, 200 steps) for several (sub-batch chains, bands per plane)
settings of the two-kernel iteration; also a few other shapes with chains on / off.  usage: concurrent_probe.py [quick]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "delta-prox_amd")); sys.path.insert(0, ROOT)
import dprox as dp
import synthetic
from dprox import _backend as be

dev = torch.device("cuda", 0)
N = int(os.environ.get("ITERS", "200"))
psf = synthetic.point_spread_function(15, 5.0)


def problem(B, C, H, W):
    rng = np.random.RandomState(2023)
    gt = torch.from_numpy(synthetic.synth(rng, B, C, H, W)).to(dev)
    b = (dp.conv(dp.Variable(), psf).to(dev).forward(gt) + torch.from_numpy((rng.randn(B, C, H, W) * (2.0 / 255.0)).astype(np.float32)).to(dev)).contiguous()
    x = dp.Variable()
    s = dp.compile(dp.sum_squares(dp.conv(x, psf) - b) + dp.norm1(dp.grad(x, dim=0)) + dp.norm1(dp.grad(x, dim=1)), method="admm", device=dev)
    _, rhos, lams, _ = s.defaults(b, 0.1, 0.005, N)
    return s, b, rhos.to(dev), {k: v.to(dev) for k, v in lams.items()}


def run(s, b, rs, ls, chains, bands, n=N):
    os.environ["DPX_CHAINS"] = str(chains)
    be.lib().call("dpx_admm_iter_config", 1 if bands else 0, bands)
    s.iters(s.initialize(b), rs[..., :4].contiguous(), {k: v[..., :4].contiguous() for k, v in ls.items()}, 4)
    best = None
    for rep in range(3):
        st = s.initialize(b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = s.iters(st, rs, ls, n)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    be.lib().call("dpx_admm_iter_config", 0, 0)
    return 1e3 * best / n, out[0]


s, b, rs, ls = problem(8, 3, 1024, 1024)
ref_ms, xref = run(s, b, rs, ls, 1, 0)
print(f"8x3x1024x1024  1 chain, auto bands : {ref_ms:.4f} ms/it")
for chains, bands in ((2, 0), (2, 64), (2, 128), (2, 256), (3, 0), (3, 64), (4, 0), (4, 64)):
    ms, x = run(s, b, rs, ls, chains, bands)
    print(f"8x3x1024x1024  {chains} chains, {bands or 'auto':>4} bands : {ms:.4f} ms/it   max|x - x_ref| = {float((x - xref).abs().max()):.1e}")
del s, b, rs, ls, xref
if len(sys.argv) < 2:
    for shape in ((2, 3, 1024, 1024), (4, 3, 1024, 1024), (8, 3, 512, 512), (16, 3, 512, 512), (4, 1, 1024, 1024), (8, 1, 512, 512), (4, 3, 512, 1024), (16, 1, 256, 256)):
        s, b, rs, ls = problem(*shape)
        a, _ = run(s, b, rs, ls, 1, 0)
        c, _ = run(s, b, rs, ls, 2, 0)
        px = shape[0] * shape[1] * shape[2] * shape[3]
        print(f"{shape}: 1 chain {a:.4f}  2 chains {c:.4f} ms/it  ({px / 2**20:.1f} Mpixel, ratio {c / a:.3f})")
        del s, b, rs, ls
os.environ.pop("DPX_CHAINS", None)
