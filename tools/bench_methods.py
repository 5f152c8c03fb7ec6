#!/usr/bin/env python
"""Iterations/s of every compile() method on the config-2 problem (8x3x1024x1024 TV deconvolution; another BxCxHxW as argument), GPU only."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "delta-prox_amd")]
import torch
import dprox as dp
import synthetic
dev = torch.device("cuda")
shape = tuple(int(v) for v in sys.argv[1].split("x")) if len(sys.argv) > 1 else (8, 3, 1024, 1024)
gt, b, psf = synthetic.deconv_case(*shape, seed=2023)
print("x".join(map(str, shape)))
bt = torch.from_numpy(b).to(dev)
for method in ("admm", "admm_vxu", "ladmm", "hqs", "pc", "pgd"):
    x = dp.Variable()
    if method == "pgd":
        fns = dp.sum_squares(dp.conv(x, psf) - bt) + dp.norm1(x)
    else:
        fns = dp.sum_squares(dp.conv(x, psf) - bt) + dp.norm1(dp.grad(x, dim=0)) + dp.norm1(dp.grad(x, dim=1))
    s = dp.compile(fns, method=method, device=dev)
    s.solve(x0=bt, rhos=0.1, lams=0.005, max_iter=3); torch.cuda.synchronize()
    t0 = time.perf_counter(); out = s.solve(x0=bt, rhos=0.1, lams=0.005, max_iter=30); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 30
    print(f"{method:9s} {dt*1e3:7.3f} ms/it = {1/dt:7.1f} it/s   path={getattr(s, 'last_path', '-')}")
