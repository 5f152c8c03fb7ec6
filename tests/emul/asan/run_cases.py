"""TEST INFRASTRUCTURE: parity cases against the ASan build of the emulated kernels (see run_asan.sh)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path[:0] = [os.path.join(ROOT, 'tests'), ROOT, os.path.join(ROOT, 'delta-prox_amd')]
from dprox import _backend as be, _ops
be._inject_for_tests(be.Library(os.environ["DPX_ASAN_LIB"]), host_pointers=True)
_ops.clear_caches()
import parity_cases as pc
import torch
which = sys.argv[1:]
cases = {
 "conv2d": lambda: pc.case_conv2d_generic("cpu"),
 "ffdnet": lambda: pc.case_ffdnet("cpu", which=("odd","gray")),
 "config1": lambda: pc.case_admm_tv_config1("cpu"),
 "small": lambda: pc.case_admm_tv_small("cpu", True),
 "linops": lambda: [pc.case_linops("cpu", t) for t in "abc"],
 "csmri": lambda: pc.case_csmri("cpu", solve=False),
 "sisr": lambda: pc.case_sisr("cpu", solve=False),
 "doe": lambda: pc.case_conv_doe("cpu"),
 "grads": lambda: pc.case_unrolled_grads("cpu"),
 "cg": lambda: pc.case_cg("cpu", 4),
 "ffbwd": lambda: pc.case_ffdnet_grads("cpu", which=("even",)),
 "ladmm": lambda: pc.case_ladmm_cg("cpu"),                    # device-side CG, dpx_cg_masked_fft, fused split-CG loop, gray bf16x3 denoiser
 "other": lambda: pc.case_other_algorithms("cpu"),            # dpx_split_rhs / dpx_pc_dual
 "bf16hist": lambda: pc.case_unrolled_grads_bf16("cpu"),
 "lsolve": lambda: pc.case_linear_solve_grad("cpu"),
 "pgd": lambda: pc.case_pgd_pow2("cpu", tiny=True),                                                      # dpx_pgd_run
 "h768": lambda: pc.case_h768("cpu", tiny=True),                                                          # fft_reg_x3 columns
 "sizes": lambda: pc.case_other_plane_sizes("cpu", sizes=((384, 256), (256, 768)), channels=1),          # ... and rows (RowMap)
 "hqs": lambda: pc.case_hqs_pow2("cpu"),                                                                  # DPX_TERM_NO_DUAL
}
def unet_layers():
    import test_emul_kernels as t
    t.test_unet_layer_kernels_vs_torch(); t.test_leaky_conv_layer_vs_torch()
cases["unet"] = unet_layers
def ffd_modes():
    """split-bf16 / bf16 convolution path on the odd-sized colour fixture (tile edges, zero-block DMA lanes)"""
    from conftest import load_golden, rel_l2
    import synthetic
    from dprox.proxfn.pnp.denoisers import FFDNetColorDenoiser
    g = load_golden("g8_ffdnet")
    col = FFDNetColorDenoiser(synthetic.ffdnet_weights(7))
    for mode, tol in (("bf16x3", 1e-5), ("bf16", 1e-2)):
        col.model.compute_mode = mode
        with torch.no_grad():
            out = col.denoise(torch.from_numpy(g["odd_x"]), torch.tensor(0.02))
        assert rel_l2(out.numpy(), g["odd_s0.02"]) <= tol
cases["ffmodes"] = ffd_modes
def pow2_terms():
    """two-kernel iteration (LDS-DMA row / column kernels) at 256x256 with 1, 3 and 4 Psi terms, uneven band partition"""
    import dprox as dp, synthetic
    gt, b, psf = synthetic.deconv_case(2, 1, 256, 256, seed=9)
    bt = torch.from_numpy(b)
    for terms in ("h+l1", "hw+nn", "hw+nn+l1"):
        x = dp.Variable()
        fns = dp.sum_squares(dp.conv(x, psf) - bt)
        if "h" in terms.split("+")[0]: fns = fns + dp.norm1(dp.grad(x, dim=0))
        if "w" in terms.split("+")[0]: fns = fns + dp.norm1(dp.grad(x, dim=1))
        if "nn" in terms: fns = fns + dp.nonneg(x)
        if "l1" in terms: fns = fns + dp.norm1(x) * 0.5
        s_ = dp.compile(fns, method="admm", device="cpu")
        out = s_.solve(x0=bt, rhos=0.3, lams=0.01, max_iter=2)
        assert s_.last_path == "fused" and torch.isfinite(out).all()


cases["pow2"] = pow2_terms
for w in which:
    cases[w](); print("OK", w, flush=True)
