#!/usr/bin/env python
"""GPU box, variant library built with -DDPX_WC_TRACE (tools/build_variant_one.sh wc_trace dpx_wgrad_c8 -DDPX_WC_TRACE; run with DPX_LIB=...):
shader-clock timeline of k_wgrad_c8<3, 3, mode> -- the eight waves of workgroup 40 along four jobs (a job = one 32-pixel step of the walk).  python tools/wgrad_trace.py [mode]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "delta-prox_amd")]
import numpy as np, torch
from dprox import _backend as be
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B, C, H, W = 2, 96, 384, 384
L = be.lib()
G = 12
g = torch.randn(B, G, H, W, 8, device="cuda") * 3
a = torch.relu(torch.randn(B, G, H, W, 8, device="cuda"))
gw, gb = torch.empty(C, C, 3, 3, device="cuda"), torch.empty(C, device="cuda")
ws = torch.empty(L.query("dpx_conv3x3_wgrad_c8_ws_bytes", C, C), dtype=torch.uint8, device="cuda")
cdll = L.cdll
cdll.dpx_dbg_wc_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
names = ["top", "split pass + LDS-DMA issue done", "matrix phase done", "barrier passed"]
for rep in range(3):
    L.call("dpx_conv3x3_wgrad_c8", be.ptr(g), be.ptr(a), be.ptr(gw), be.ptr(gb), C, C, G, G, mode, None, B, H, W, be.ptr(ws), be.stream())
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 512)()
    assert cdll.dpx_dbg_wc_trace(buf, 512) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(8, 64).astype(np.float64)
    t0 = t[:, 0].min()
    print(f"run {rep} (mode {mode}): shader-clock cycles since the first wave reached job 8 of workgroup 40; columns: waves 0..7")
    for st in range(4):
        for i, nm in enumerate(names):
            print(f"  job {8 + st} {nm:32s} " + " ".join(f"{int(v - t0):7d}" for v in t[:, st * 8 + i]))
