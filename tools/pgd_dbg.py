import os, sys
sys.path[:0] = ["/root/repo", "/root/repo/delta-prox_amd"]
import torch, numpy as np
import dprox as dp, synthetic
dev = "cuda"
B, C, H, W = 6, 2, 512, 512
gt, b0, psf = synthetic.deconv_case(B, C, H, W, seed=17 + B + H)
b = torch.from_numpy(b0).to(dev)
iters = 23
rhos = torch.linspace(0.5, 0.2, iters)[None, :] * torch.linspace(1.0, 1.5, B)[:, None]
def run(nch):
    os.environ["DPX_CHAINS"] = str(nch)
    x = dp.Variable()
    s = dp.compile(dp.sum_squares(dp.conv(x, psf) - b) + dp.norm1(x) * 0.5, method="pgd", device=dev)
    return s.solve(x0=b, rhos=rhos * 0.5, lams=0.01, max_iter=iters)
ref = run(1)
for nch in (2, 3, 3, 6, 3):
    got = run(nch)
    d = (got - ref).abs().amax(dim=(1, 2, 3))
    print(nch, [float(v) for v in d])
