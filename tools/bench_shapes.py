#!/usr/bin/env python
"""ADMM TV-deconvolution ms/iteration on plane sizes around the reference's example images (face 768x1024, ascent 512x512), GPU only."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "delta-prox_amd")]
import torch
import dprox as dp
import synthetic
dev = torch.device("cuda")
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(8, 3, 1024, 1024), (8, 3, 768, 1024), (8, 3, 1000, 1000), (8, 3, 720, 1280), (1, 3, 768, 1024), (1, 3, 512, 512)]
for (B, C, H, W) in shapes:
    gt, b, psf = synthetic.deconv_case(B, C, H, W, seed=1)
    bt = torch.from_numpy(b).to(dev)
    x = dp.Variable()
    fns = dp.sum_squares(dp.conv(x, psf) - bt) + dp.norm1(dp.grad(x, dim=0)) + dp.norm1(dp.grad(x, dim=1))
    s = dp.compile(fns, method="admm", device=dev)

    def run(n):
        s.solve(x0=bt, rhos=0.1, lams=0.005, max_iter=n); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); s.solve(x0=bt, rhos=0.1, lams=0.005, max_iter=n); torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return best
    a, c = run(10), run(60)
    per = (c - a) / 50
    print(f"{B}x{C}x{H}x{W}: {per * 1e3:7.4f} ms/it = {per * 1e12 / (B * C * H * W):6.2f} ps/pixel   path={getattr(s, 'last_path', '-')}")
