import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "delta-prox_amd")]
import torch
from dprox.proxfn.pnp.denoisers import FFDNetDenoiser
import synthetic as S
dev = torch.device("cuda")
den = FFDNetDenoiser(S.ffdnet_weights(11, 1, 1, 64, 15)).to(dev)
for shape in ((32, 1, 320, 320), (8, 1, 1024, 1024)):
    x = torch.rand(*shape, device=dev); sig = torch.full((shape[0],), 0.05, device=dev)
    with torch.no_grad():
        for _ in range(2): y = den.denoise(x, sig)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): y = den.denoise(x, sig)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    flop = 2 * 9 * (5 * 64 + 13 * 64 * 64 + 64 * 4) * (shape[2] // 2) * (shape[3] // 2) * shape[0]
    print(f"FFDNet-gray {shape}: {dt*1e3:.2f} ms {flop/dt/1e12:.1f} TFLOP/s ({flop/dt/157.3e12*100:.1f}%)")
