#!/usr/bin/env python
"""GPU box, variant library built with -DDPX_PAR_TRACE (tools/build_variant.sh par_trace -DDPX_PAR_TRACE; run with DPX_LIB=...): the phase
timeline of the row-parallel kernel's last launch -- 100 MHz real-time stamps of every wave of the first 256 workgroups."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "delta-prox_amd")]
import numpy as np, torch
import dprox as dp, synthetic
from dprox import _backend as be
B, C, H, W = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1x3x1024x1024").split("x"))
gt, b, psf = synthetic.deconv_case(B, C, H, W, seed=1)
bt = torch.from_numpy(b).cuda()
x = dp.Variable()
s = dp.compile(dp.sum_squares(dp.conv(x, psf) - bt) + dp.norm1(dp.grad(x, dim=0)) + dp.norm1(dp.grad(x, dim=1)), method="admm", device="cuda")
cdll = be.lib().cdll
for rep in range(3):
    s.solve(x0=bt, rhos=0.1, lams=0.005, max_iter=12, return_full_states=True)      # (full states: the last launch is a regular emitting pass)
    torch.cuda.synchronize()
    n = 256 * 16 * 10
    buf = (ctypes.c_ulonglong * n)()
    cdll.dpx_dbg_par_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert cdll.dpx_dbg_par_trace(buf, n) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(256, 16, 10).astype(np.float64)
    nblk = min(256, B * C * ((H + 13) // 14))
    t = t[:nblk]
    live = t[..., 0] > 0
    t0 = t[..., 0][live].min()
    us = (t - t0) / 100.0
    names = ["entry", "twiddles+barrier", "DMA landed", "inverse FFT done", "barrier 1", "phase B done", "barrier 2", "forward FFT done", "stores issued"]
    print(f"run {rep}: {nblk} workgroups; us since the first wave's entry: mean / min / max over waves")
    names.append("(all prologue loads back)")
    for i, nm in enumerate(names):
        col = us[..., i][live & (t[..., i] > 0)]
        if col.size:
            print(f"   {nm:20s} {col.mean():7.2f} {col.min():7.2f} {col.max():7.2f}   (n={col.size})")

    n = 256 * 16 * 10
    cdll.dpx_dbg_cols_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert cdll.dpx_dbg_cols_trace(buf, n) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(256, 16, 10).astype(np.float64)
    nblk = min(256, B * C * (W // 2 // 8))
    t = t[:nblk, :8]
    live = t[..., 0] > 0
    us = (t - t[..., 0][live].min()) / 100.0
    print(f"   k_cols_p2: {nblk} workgroups")
    for i, nm in enumerate(["entry", "tile loaded", "forward FFT done (+ table / data spectrum requested)", "operator done", "barrier", "inverse FFT done", "stores issued", "stores acknowledged", "  (hook entry: forward pass C inputs read)", "  (hook exit: table DMA + data-spectrum loads issued)"]):
        col = us[..., i][live]
        print(f"   {nm:54s} {col.mean():7.2f} {col.min():7.2f} {col.max():7.2f}")
