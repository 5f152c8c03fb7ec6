#!/usr/bin/env python
"""Host-side cost per backend call (tiny tensors, GPU only): what bounds the launch-bound configurations (1, 4)."""
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "delta-prox_amd")]
import torch
from dprox import _ops as ops
dev = torch.device("cuda")
a = torch.rand(2, 1, 16, 16, device=dev); b = torch.rand_like(a)
def t(fn, n=2000):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print(f"lincomb  {t(lambda: ops.lincomb([(1.0, a), (0.5, b)])):6.1f} us/call")
print(f"bdot     {t(lambda: ops.bdot(a, b)):6.1f} us/call")
print(f"grad     {t(lambda: ops.grad(a, 0)):6.1f} us/call")
print(f"torch add{t(lambda: torch.add(a, b)):6.1f} us/call")
pr = cProfile.Profile(); pr.enable()
for _ in range(500): ops.lincomb([(1.0, a), (0.5, b)])
pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(12)
