#!/usr/bin/env python
"""FFDNet-color forward at the config-3 shape: time, TFLOP/s vs the 157.3 TFLOP/s fp32-MFMA roofline (GPU only)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "delta-prox_amd")]
import torch
from dprox import _backend as be
from dprox.proxfn.pnp.denoisers import FFDNetColorDenoiser
import synthetic as O
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda")
den = FFDNetColorDenoiser(O.ffdnet_weights(7)).to(dev)
x = torch.rand(B, 3, 1024, 1024, device=dev)
sig = torch.full((B,), 0.05, device=dev)
for _ in range(2): y = den.denoise(x, sig)
torch.cuda.synchronize()
be.lib().call("dpx_timing_enable", 1)
buf = ctypes.create_string_buffer(1 << 16); be.lib().call("dpx_timing_report", buf, len(buf))
t0 = time.perf_counter(); n = 3
for _ in range(n): y = den.denoise(x, sig)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
be.lib().call("dpx_timing_report", buf, len(buf)); print(buf.value.decode())
flop = 2 * 9 * (13 * 96 + 10 * 96 * 96 + 96 * 12) * 512 * 512 * B
print(f"FFDNet-color B={B}: {dt*1e3:.2f} ms  {flop/dt/1e12:.1f} TFLOP/s  ({flop/dt/157.3e12*100:.1f}% of fp32 MFMA peak)")
# training forward (keeps activations) + backward-data pass (gradients w.r.t. image and sigma), frozen weights
den.requires_grad_(False)
Bt = min(B, 4)
xt = torch.rand(Bt, 3, 512, 512, device=dev, requires_grad=True)
st = torch.full((Bt,), 0.05, device=dev, requires_grad=True)
def step():
    xt.grad = None; st.grad = None
    y = den.denoise(xt, st)
    y.sum().backward()
for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): step()
torch.cuda.synchronize(); dtb = (time.perf_counter() - t0) / 5
flop_t = 2 * 9 * (13 * 96 + 10 * 96 * 96 + 96 * 12) * 256 * 256 * Bt
print(f"FFDNet-color train fwd+bwd-data B={Bt} 512x512: {dtb*1e3:.2f} ms  {2*flop_t/dtb/1e12:.1f} TFLOP/s over the two passes")
den.requires_grad_(True)
def step_w():
    den.zero_grad()
    y = den.denoise(xt.detach(), st.detach())
    y.sum().backward()
for _ in range(2): step_w()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): step_w()
torch.cuda.synchronize(); dtw = (time.perf_counter() - t0) / 5
print(f"FFDNet-color train fwd + bwd-data + weight grads B={Bt} 512x512: {dtw*1e3:.2f} ms  {3*flop_t/dtw/1e12:.1f} TFLOP/s over the three passes")
