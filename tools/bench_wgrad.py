#!/usr/bin/env python
"""GPU box: dpx_conv3x3_wgrad_c8 (k_wgrad_c8 + k_wgrad_reduce) on its own at the training bench's layer size (2 x 96 x 384 x 384 by default):
per-kernel event times of both arithmetic modes.   python tools/bench_wgrad.py [B C H W] [--reps N]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "delta-prox_amd")]
import torch
from dprox import _backend as be
args = [a for a in sys.argv[1:] if not a.startswith("--")]
B, C, H, W = (int(v) for v in args[:4]) if len(args) >= 4 else (2, 96, 384, 384)
reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 20
L = be.lib()
G = 2 * ((C + 15) // 16)
torch.manual_seed(0)
g = torch.randn(B, G, H, W, 8, device="cuda") * 3
a = torch.relu(torch.randn(B, G, H, W, 8, device="cuda"))
gw, gb = torch.empty(C, C, 3, 3, device="cuda"), torch.empty(C, device="cuda")
ws = torch.empty(L.query("dpx_conv3x3_wgrad_c8_ws_bytes", C, C), dtype=torch.uint8, device="cuda")
buf = ctypes.create_string_buffer(1 << 16)
for mode in (3, 6):
    for _ in range(3):
        L.call("dpx_conv3x3_wgrad_c8", be.ptr(g), be.ptr(a), be.ptr(gw), be.ptr(gb), C, C, G, G, mode, None, B, H, W, be.ptr(ws), be.stream())
    torch.cuda.synchronize()
    L.call("dpx_timing_enable", 1)
    L.call("dpx_timing_report", buf, len(buf))
    for _ in range(reps):
        L.call("dpx_conv3x3_wgrad_c8", be.ptr(g), be.ptr(a), be.ptr(gw), be.ptr(gb), C, C, G, G, mode, None, B, H, W, be.ptr(ws), be.stream())
    torch.cuda.synchronize()
    L.call("dpx_timing_report", buf, len(buf))
    L.call("dpx_timing_enable", 0)
    flop = 2.0 * B * H * W * C * C * 9
    for line in buf.value.decode().splitlines():
        name, cnt, tot = line.split()
        us = float(tot) / int(cnt) * 1e3
        extra = f"   {flop / us * 1e-6:7.1f} TFLOP/s fp32-equivalent" if name == "k_wgrad_c8" else ""
        print(f"{B}x{C}x{H}x{W} mode {mode}: {name:18s} {us:8.1f} us{extra}")
