#!/usr/bin/env python
"""GPU box, variant library built with -DDPX_WN_TRACE (tools/build_variant.sh wn_trace -DDPX_WN_TRACE; run with DPX_LIB=...): shader-clock timeline
of the Winograd layer kernel -- the eight waves of workgroup 0 along their second tile.  python tools/wino_trace.py [gray|color]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "delta-prox_amd")]
import numpy as np, torch
from dprox import _backend as be
from dprox.proxfn.pnp.denoisers import FFDNetColorDenoiser, FFDNetDenoiser
import synthetic as O
which = sys.argv[1] if len(sys.argv) > 1 else "gray"
dev = torch.device("cuda")
if which == "direct":
    den = FFDNetColorDenoiser(O.ffdnet_weights(7)).to(dev); x = torch.rand(8, 3, 1024, 1024, device=dev)
    den.model.compute_mode = "f16x2"
    sig = torch.full((8,), 0.05, device=dev)
    cdll = be.lib().cdll
    cdll.dpx_dbg_bx_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
    with torch.no_grad():
        for rep in range(2):
            den.denoise(x, sig); torch.cuda.synchronize()
            buf = (ctypes.c_ulonglong * 512)()
            assert cdll.dpx_dbg_bx_trace(buf, 512) == 0
            t = np.frombuffer(buf, dtype=np.uint64).reshape(8, 64).astype(np.float64)
            t0 = t[:, 0].min()
            names = {0: "entry", 49: "main loop done", 50: "stores issued", 51: "stores acknowledged"}
            for c in range(6):
                names.update({1 + c * 8: f"c{c} top", 2 + c * 8: f"c{c} act landed", 3 + c * 8: f"c{c} barrier", 4 + c * 8: f"c{c} split done", 5 + c * 8: f"c{c} barrier2",
                              6 + c * 8: f"c{c} tg0", 7 + c * 8: f"c{c} tg1", 8 + c * 8: f"c{c} tg2"})
            print(f"run {rep}: direct split-f16 kernel, workgroup 300 of image 0 (96 -> 96 layer): cycles since entry, waves 0..7")
            for i in sorted(names):
                if t[:, i].max() > 0:
                    print(f"  {names[i]:22s} " + " ".join(f"{int(v - t0):7d}" for v in t[:, i]))
    sys.exit(0)
if which == "gray":
    den = FFDNetDenoiser(O.ffdnet_weights(11, 1, 1, 64, 15)).to(dev); x = torch.rand(32, 1, 320, 320, device=dev); nchunk = 4; nmt = 2
else:
    den = FFDNetColorDenoiser(O.ffdnet_weights(7)).to(dev); x = torch.rand(8, 3, 1024, 1024, device=dev); nchunk = 6; nmt = 3
den.model.compute_mode = "f16x2w"
sig = torch.full((x.shape[0],), 0.05, device=dev)
cdll = be.lib().cdll
cdll.dpx_dbg_wn_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
with torch.no_grad():
    for rep in range(3):
        # stop behind a middle layer: run the whole network, the buffer holds the LAST Winograd launch (the last layer, MT = 1) -- so trace a
        # network whose last traced launch is a middle layer: truncate via nb is not possible; instead read after each forward and accept MT of the last layer
        den.denoise(x, sig)
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 512)()
        assert cdll.dpx_dbg_wn_trace(buf, 512) == 0
        t = np.frombuffer(buf, dtype=np.uint64).reshape(8, 64).astype(np.float64)
        t0 = t[:, 0].min()
        print(f"run {rep}: cycles since the first wave entered the tile (rows: waves 0..7)")
        names = {0: "tile entry", 49: "main loop done", 62: "tile done"}
        for c in range(nchunk):
            names.update({1 + c * 8: f"c{c} top", 2 + c * 8: f"c{c} land waited", 3 + c * 8: f"c{c} barrier", 4 + c * 8: f"c{c} unit0 built + A", 5 + c * 8: f"c{c} u0", 6 + c * 8: f"c{c} u1",
                          7 + c * 8: f"c{c} u2", 8 + c * 8: f"c{c} u3"})
        for m in range(3):
            names.update({50 + m * 4: f"ep{m} barrier", 51 + m * 4: f"ep{m} Z written", 52 + m * 4: f"ep{m} barrier2"})
        for i in sorted(names):
            if t[:, i].max() > 0:
                print(f"  {names[i]:22s} " + " ".join(f"{int(v - t0):7d}" for v in t[:, i]))
