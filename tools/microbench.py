#!/usr/bin/env python
"""Per-kernel timing of the spectral pipeline at the config-2 shape (tuning aid, GPU only)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "delta-prox_amd")]
import torch
from dprox import _backend as be, _ops as ops
import synthetic

def report(tag):
    buf = ctypes.create_string_buffer(1 << 16)
    be.lib().call("dpx_timing_report", buf, len(buf))
    out = {l.split()[0]: round(1e3 * float(l.split()[2]) / int(l.split()[1]), 1) for l in buf.value.decode().splitlines()}
    print(tag, out, flush=True)

B, C, H, W = 8, 3, 1024, 1024
dev = torch.device("cuda")
x = torch.rand(B, C, H, W, device=dev)
psf = synthetic.point_spread_function(15, 5.0)
otf = ops.make_otf(psf, C, H, W, dev)
d0 = ops.new_diag(C, H, W, dev); ops.accumulate_diag(d0, psf, 1.0, C, H, W)
d1 = ops.new_diag(C, H, W, dev)
import numpy as np
for dim in (0, 1):
    k = np.array([1, -1], dtype=np.float64).reshape((2, 1, 1) if dim == 0 else (1, 2, 1))
    ops.accumulate_diag(d1, k, 1.0, C, H, W)
rho = torch.full((B,), 0.1, device=dev)
FK = ops.data_spectrum(x, otf, conj=True)
out = torch.empty_like(x)
for _ in range(3):
    ops.fourier_solve(x, d0, d1, 0.0, 0.0, rho, out=out, spec_add=FK); ops.fft_conv(x, otf, out=out)
torch.cuda.synchronize()
be.lib().call("dpx_timing_enable", 1)
report("drop")
for _ in range(20): ops.fourier_solve(x, d0, d1, 0.0, 0.0, rho, out=out, spec_add=FK)
torch.cuda.synchronize(); report("solve+add ")
for _ in range(20): ops.fourier_solve(x, d0, d1, 0.0, 0.0, rho, out=out)
torch.cuda.synchronize(); report("solve      ")
for _ in range(20): ops.fft_conv(x, otf, out=out)
torch.cuda.synchronize(); report("conv (MUL) ")
