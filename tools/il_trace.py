#!/usr/bin/env python
"""GPU box, variant library built with -DDPX_PAR_TRACE (DPX_LIB=...): phase timeline of the LAST k_cols_il launch of an ADMM solve on size-generic
planes (default 8x3x1000x1000) -- 100 MHz stamps of thread 0 of the first 2048 workgroups."""
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
cdll.dpx_dbg_il_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
names = ["entry", "tile in shared memory", "forward transform done", "operator done (tables + data spectrum)", "inverse transform done", "stores issued"]
for rep in range(2):
    s.solve(x0=bt, rhos=0.1, lams=0.005, max_iter=8)
    torch.cuda.synchronize()
    n = 2048 * 8
    buf = (ctypes.c_ulonglong * n)()
    assert cdll.dpx_dbg_il_trace(buf, n) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(2048, 8).astype(np.float64)
    live = t[:, 0] > 0
    t = t[live]
    t0 = t[:, 0].min()
    us = (t - t0) / 100.0
    dur = us[:, 5] - us[:, 0]
    print(f"run {rep}: {int(live.sum())} workgroups stamped; launch span {us[:, 5].max():.1f} us; a workgroup lives {dur.mean():.2f} us (min {dur.min():.2f}, max {dur.max():.2f})")
    for i in range(1, 6):
        d = us[:, i] - us[:, i - 1]
        print(f"   {names[i - 1]:40s} -> {names[i]:40s} {d.mean():6.2f} us  (min {d.min():5.2f}, max {d.max():5.2f})")
    order = np.argsort(us[:, 0])
    starts = us[order, 0]
    print("   workgroup starts (us), every 128th:", " ".join(f"{v:.1f}" for v in starts[::128]))
