#!/usr/bin/env python
"""GPU box: times the one-off setup kernels of a cold config-2 solve (fp64 data spectrum, OTF tables) with the library's
per-kernel timers.  Geometry knobs of the data-spectrum kernels come from the environment (DPX_DS_RPB, DPX_DS_ROW_THREADS,
DPX_DS_COL_THREADS): run once per setting."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "delta-prox_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
import dprox as dp, synthetic
from dprox import _backend as be, _ops as ops

dev = torch.device("cuda", 0)
B, C, H, W = 8, 3, 1024, 1024
b = torch.rand(B, C, H, W, device=dev)
psf = synthetic.point_spread_function(15, 5.0)
otf = ops.make_otf(psf, C, H, W, dev)
L = be.lib()

def report():
    buf = ctypes.create_string_buffer(1 << 16)
    L.call("dpx_timing_report", buf, len(buf))
    return {ln.split()[0]: (int(ln.split()[1]), float(ln.split()[2])) for ln in buf.value.decode().splitlines() if ln.split()}

for _ in range(2):
    ops.data_spectrum(b, otf, conj=True)
torch.cuda.synchronize()
L.call("dpx_timing_enable", 1); report()
for _ in range(5):
    ops.data_spectrum(b, otf, conj=True)
    ops.make_otf(psf, C, H, W, dev)
torch.cuda.synchronize()
r = report(); L.call("dpx_timing_enable", 0)
print({k: round(1e3 * t / c, 1) for k, (c, t) in r.items()}, {k: os.environ[k] for k in os.environ if k.startswith("DPX_DS")})
