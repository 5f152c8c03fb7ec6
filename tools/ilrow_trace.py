#!/usr/bin/env python
"""GPU box, variant library built with -DDPX_PAR_TRACE (DPX_LIB=...): phase timeline of the one-wave workgroups of the LAST k_rows_c2r_il launch of an
ADMM solve on size-generic planes (default 8x3x1000x1000) -- 100 MHz stamps of the first 4096 workgroups."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "delta-prox_amd")]
import numpy as np, torch
import dprox as dp, synthetic
from dprox import _backend as be
B, C, H, W = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "8x3x1000x1000").split("x"))
gt, b, psf = synthetic.deconv_case(B, C, H, W, seed=1)
bt = torch.from_numpy(b).cuda()
x = dp.Variable()
s = dp.compile(dp.sum_squares(dp.conv(x, psf) - bt) + dp.norm1(dp.grad(x, dim=0)) + dp.norm1(dp.grad(x, dim=1)), method="admm", device="cuda")
cdll = be.lib().cdll
cdll.dpx_dbg_ilrow_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
names = ["entry", "spectrum row untangled into shared memory", "pass 1", "pass 2", "pass 3", "pass 4", "(transform done)", "row stored"]
for rep in range(2):
    s.solve(x0=bt, rhos=0.1, lams=0.005, max_iter=8)
    torch.cuda.synchronize()
    n = 4096 * 8
    buf = (ctypes.c_ulonglong * n)()
    assert cdll.dpx_dbg_ilrow_trace(buf, n) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8).astype(np.float64)
    t = t[t[:, 0] > 0]
    us = (t - t[:, 0].min()) / 100.0
    dur = us[:, 7] - us[:, 0]
    print(f"run {rep}: {len(t)} workgroups stamped; a one-wave workgroup lives {dur.mean():.2f} us (min {dur.min():.2f}, max {dur.max():.2f}); the 4096 of them span {us[:, 7].max():.1f} us")
    prev = 0
    for i in range(1, 8):
        if (t[:, i] > 0).all():
            d = us[:, i] - us[:, prev]
            print(f"   {names[prev]:44s} -> {names[i]:44s} {d.mean():6.2f} us (min {d.min():5.2f}, max {d.max():5.2f})")
            prev = i
