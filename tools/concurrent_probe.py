#!/usr/bin/env python
"""Tuning probe (GPU box): config 2's 8 images as ONE batch-8 solve vs. the same images as 2 / 4 independent sub-batch solves running
concurrently on separate HIP streams (the images of a batch never exchange data: the chains have no common synchronisation point, so
one chain's column kernel can run beside another chain's row kernel).  Prints ms per iteration of the whole batch for each split."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "delta-prox_amd")); sys.path.insert(0, ROOT)
import dprox as dp
import synthetic
from dprox import _backend as be

dev = torch.device("cuda", 0)
B, C, H, W = 8, 3, 1024, 1024
N = int(os.environ.get("ITERS", "200"))
rng = np.random.RandomState(2023)
gt = torch.from_numpy(synthetic.synth(rng, B, C, H, W)).to(dev)
psf = synthetic.point_spread_function(15, 5.0)
b = (dp.conv(dp.Variable(), psf).to(dev).forward(gt) + torch.from_numpy((rng.randn(B, C, H, W) * (2.0 / 255.0)).astype(np.float32)).to(dev)).contiguous()


def make(bs):
    x = dp.Variable()
    return dp.compile(dp.sum_squares(dp.conv(x, psf) - bs) + dp.norm1(dp.grad(x, dim=0)) + dp.norm1(dp.grad(x, dim=1)), method="admm", device=dev)


def run(nsplit, bands):
    be.lib().call("dpx_admm_iter_config", 1 if bands else 0, bands)
    parts = [b[i * (B // nsplit):(i + 1) * (B // nsplit)].contiguous() for i in range(nsplit)]
    solvers = [make(p) for p in parts]
    streams = [torch.cuda.Stream() for _ in range(nsplit)] if nsplit > 1 else [torch.cuda.current_stream()]
    sched = []
    for s, p in zip(solvers, parts):
        _, rhos, lams, _ = s.defaults(p, 0.1, 0.005, N)
        sched.append((rhos.to(dev), {k: v.to(dev) for k, v in lams.items()}))
        s.solve(x0=p, rhos=0.1, lams=0.005, max_iter=5)              # tables, data spectrum
    best = None
    for rep in range(3):
        states = [s.initialize(p) for s, p in zip(solvers, parts)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = []
        for s, st, (rs, ls), strm in zip(solvers, states, sched, streams):
            with torch.cuda.stream(strm):
                outs.append(s.iters(st, rs, ls, N))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    x = torch.cat([o[0] for o in outs], dim=0)
    return 1e3 * best / N, x


ref_ms, xref = run(1, 0)
print(f"1 x batch 8 (default bands)      : {ref_ms:.4f} ms/it")
for nsplit, bands in ((2, 128), (2, 256), (4, 128), (4, 256), (8, 128), (1, 128)):
    ms, x = run(nsplit, bands)
    print(f"{nsplit} x batch {B // nsplit}, {bands:3d} bands per plane : {ms:.4f} ms/it   max|x - x_ref| = {float((x - xref).abs().max()):.2e}")
be.lib().call("dpx_admm_iter_config", 0, 0)
