#!/usr/bin/env python
"""GPU box: does a launch of a few planes gain from running its planes as independent chains on separate streams?  A B x 1 x H x W problem
(B single-plane images: what plane-granularity chains of a 1 x B x H x W image would launch) as 1 chain and as B chains (DPX_CHAINS):
wall clock per iteration (difference of a 60- and a 20-iteration solve) and bit-identity.   usage: plane_chain_probe.py [BxHxW ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "delta-prox_amd")]
import torch
import dprox as dp, synthetic

shapes = [a for a in sys.argv[1:] if "x" in a] or ["3x1024x1024", "3x768x1024", "3x512x512", "6x1024x1024"]
for shp in shapes:
    B, H, W = (int(v) for v in shp.split("x"))
    gt, b, psf = synthetic.deconv_case(B, 1, H, W, seed=1)
    bt = torch.from_numpy(b).cuda()
    x = dp.Variable()
    s = dp.compile(dp.sum_squares(dp.conv(x, psf) - bt) + dp.norm1(dp.grad(x, dim=0)) + dp.norm1(dp.grad(x, dim=1)), method="admm", device="cuda")

    issue = {}

    def wall(n):
        best = None
        for _ in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            s.solve(x0=bt, rhos=0.1, lams=0.005, max_iter=n)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if best is None or dt < best:
                best, issue[n] = dt, t1 - t0          # (issue: until solve() returned -- every launch handed to the runtime)
        return best

    ref = None
    for chains in (1, 2, B):
        os.environ["DPX_CHAINS"] = str(chains)
        out = s.solve(x0=bt, rhos=0.1, lams=0.005, max_iter=10).clone()
        ref = out if ref is None else ref
        s.solve(x0=bt, rhos=0.1, lams=0.005, max_iter=60)
        per_it = (wall(100) - wall(20)) / 80
        print(f"{shp}: {chains} chain(s)  {per_it * 1e6:7.2f} us/it   (host issue {(issue[100] - issue[20]) / 80 * 1e6:6.2f} us/it)   bit-identical {bool(torch.equal(out, ref))}", flush=True)
    os.environ.pop("DPX_CHAINS", None)
