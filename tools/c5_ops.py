#!/usr/bin/env python
"""GPU box: the torch operators of one config-5 training step in issue order (torch.profiler), to see which glue ops surround the two C calls"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "delta-prox_amd")]
import torch
from torch.profiler import profile, ProfilerActivity
import dprox as dp, synthetic
dev = torch.device("cuda", 0)
gt, b, psf = synthetic.deconv_case(4, 3, 512, 512, seed=2023)
bt, gtt = torch.from_numpy(b).to(dev), torch.from_numpy(gt).to(dev)
x = dp.Variable()
n0, n1 = dp.norm1(dp.grad(x, dim=0)), dp.norm1(dp.grad(x, dim=1))
s = dp.compile(dp.sum_squares(dp.conv(x, psf) - bt) + n0 + n1, method="admm", device=dev)
s = dp.specialize(s, method="unroll", device=dev, max_iter=10)
prm = [torch.full((10,), v, requires_grad=True, device=dev) for v in (0.1, 0.005, 0.005)]
def step():
    for p in prm:
        p.grad = None
    o = s.solve(x0=bt, rhos=prm[0], lams={n0: prm[1], n1: prm[2]})
    loss = ((o - gtt) ** 2).mean()
    loss.backward()
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=False) as prof:
    step()
    torch.cuda.synchronize()
evs = sorted([e for e in prof.events() if e.cpu_parent is None or e.cpu_parent.name.startswith(("_UnrolledClosed", "autograd::engine"))], key=lambda e: e.time_range.start)
t0 = evs[0].time_range.start
for e in evs:
    if e.name.startswith(("aten::", "_Unrolled", "hip", "Memcpy", "autograd")) or "Backward" in e.name:
        print(f"{(e.time_range.start - t0):8.0f} us  {e.cpu_time_total:7.0f}  {e.name}  {[list(s) for s in (e.input_shapes or [])][:3]}")
