#!/usr/bin/env python
"""GPU box: config 5's training step (4 x 3 x 512 x 512, ADMM unrolled x10, MSE, backward w.r.t. the schedules), 40 steps per setting of the
backward knob, alternating in one process: wall time per step, and the host's issue time per step (no synchronize inside the loop)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "delta-prox_amd")]
import torch
import dprox as dp, synthetic
from dprox import _backend as be
dev = torch.device("cuda", 0)
dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
gt, b, psf = synthetic.deconv_case(4, 3, 512, 512, seed=2023)
bt, gtt = torch.from_numpy(b).to(dev), torch.from_numpy(gt).to(dev)
x = dp.Variable()
n0, n1 = dp.norm1(dp.grad(x, dim=0)), dp.norm1(dp.grad(x, dim=1))
s = dp.compile(dp.sum_squares(dp.conv(x, psf) - bt) + n0 + n1, method="admm", device=dev)
s = dp.specialize(s, method="unroll", device=dev, max_iter=10, **({} if dtype == "f32" else {"dtype": "bf16"}))
prm = [torch.full((10,), v, requires_grad=True, device=dev) for v in (0.1, 0.005, 0.005)]
def step():
    for p in prm:
        p.grad = None
    o = s.solve(x0=bt, rhos=prm[0], lams={n0: prm[1], n1: prm[2]})
    loss = ((o - gtt) ** 2).mean()
    loss.backward()
    return loss
def run(n):
    step(); step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t2 - t0) / n * 1e3, (t1 - t0) / n * 1e3
for rnd in range(3):
    for name, knobs in (("two-kernel backward", {}), ("... lock-step bands (k_bwd_rows)", dict(unroll_bwd_par_max_rows=-1)),
                        ("... forward on the streaming kernel too", dict(unroll_bwd_par_max_rows=-1, iter_par_max_rows=-1)),
                        ("image-domain fused stage", dict(unroll_bwd_staged=2)), ("staged", dict(unroll_bwd_staged=1))):
        with be.tuned(**knobs):
            w, h = run(40)
        print(f"{dtype} {name:42s} {w:.3f} ms per step (host issue {h:.3f} ms)")
