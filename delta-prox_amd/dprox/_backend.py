"""ctypes binding of ``libdpx_hip.so`` (C ABI declared in ``include/dpx.h``).

The HIP library is the product: there is no CPU implementation behind these calls and no
fallback.  ``lib()`` raises if the shared object has not been built (``python __graft_entry__.py``),
and every op refuses tensors that do not live on a HIP device.

``_inject_for_tests`` exists for ``tests/emul`` only: it swaps in the *same kernel sources*
compiled for the host by the SIMT emulator so that indexing can be checked on a GPU-less box.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_long, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# DPX_LIB selects another *HIP* build of the same library (kernel tuning experiments); never a non-HIP substitute
# Where the library lives: inside the package for an installed wheel (dprox/lib/, setup.py's build hook), next to the package in the
# source tree (delta-prox_amd/lib/, __graft_entry__.build()).
_IN_PKG = os.path.join(_HERE, "lib", "libdpx_hip.so")
LIB_PATH = os.environ.get("DPX_LIB") or (_IN_PKG if os.path.exists(_IN_PKG) else os.path.join(os.path.dirname(_HERE), "lib", "libdpx_hip.so"))

PROX_NORM1, PROX_NONNEG, PROX_SUMSQ, PROX_EXTERNAL = 0, 1, 2, 3
LIN_IDENTITY, LIN_GRAD_H, LIN_GRAD_W = 0, 1, 2
TERM_NO_DUAL = 1          # dpx_term.reserved flags (include/dpx.h)
TERM_U_ZERO = 2
TERM_VXU = 4
MAX_TERMS = 4


class DpxError(RuntimeError):
    pass


class F16RangeError(DpxError):
    """an operand of the split-f16 FFDNet arithmetic left the binary16 range: the result of that solve / denoise() call is invalid
    (the callers re-run on the split-bf16 arithmetic, see ``f16_fallback``)"""


class Term(ctypes.Structure):
    """``dpx_term`` of include/dpx.h."""
    _fields_ = [("linop", c_int32), ("prox", c_int32), ("alpha", c_float), ("reserved", c_int32),
                ("lam", c_void_p), ("v", c_void_p), ("u", c_void_p), ("u_out", c_void_p)]


class Chain(ctypes.Structure):
    """``dpx_chain`` of include/dpx.h."""
    _fields_ = [("spec_a", c_void_p), ("spec_b", c_void_p), ("spec_add", c_void_p), ("terms", POINTER(Term)), ("rho_tab", c_void_p),
                ("lam_tabs", POINTER(c_void_p)), ("x_out", c_void_p), ("B", c_int32), ("seed", c_int32), ("stream", c_void_p), ("seed_x0", c_void_p)]


class BwdTerm(ctypes.Structure):
    """``dpx_bwd_term`` of include/dpx.h."""
    _fields_ = [("linop", c_int32), ("prox", c_int32), ("alpha", c_float), ("reserved", c_int32),
                ("lam", c_void_p), ("v", c_void_p), ("gv", c_void_p), ("gu_new", c_void_p), ("gu", c_void_p)]


# name -> (restype, argtypes); every symbol include/dpx.h declares
SIGNATURES = {
    "dpx_version": (c_int, []),
    "dpx_last_error": (c_char_p, []),
    "dpx_timing_enable": (c_int, [c_int]),
    "dpx_timing_report": (c_int, [c_char_p, c_size_t]),
    "dpx_tune_count": (c_int, []),
    "dpx_tune_name": (c_char_p, [c_int]),
    "dpx_tune_set": (c_int, [c_char_p, c_int]),
    "dpx_tune_get": (c_int, [c_char_p, POINTER(c_int)]),
    "dpx_cg_config": (c_int, [c_int, c_int, c_int]),
    "dpx_fft_table_bytes": (c_size_t, [c_int, c_int]),
    "dpx_fft_table_init": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "dpx_spectrum_bytes": (c_size_t, [c_int, c_int, c_int]),
    "dpx_otf_bytes": (c_size_t, [c_int, c_int, c_int]),
    "dpx_diag_bytes": (c_size_t, [c_int, c_int, c_int]),
    "dpx_psf2otf": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_int, c_void_p]),
    "dpx_fft_conv": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "dpx_pgd_supported": (c_int, [c_int, c_int, c_int]),
    "dpx_pgd_run": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_float, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p,
                            c_void_p, c_void_p]),
    "dpx_data_spectrum_ws_bytes": (c_size_t, [c_int, c_int, c_int]),
    "dpx_data_spectrum": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dpx_table_to_full": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "dpx_table_from_full": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "dpx_otf_from_full": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "dpx_denominator_bytes": (c_size_t, [c_int, c_int, c_int]),
    "dpx_denominator_pack": (c_int, [c_void_p, c_float, c_void_p, c_float, c_void_p, c_int, c_int, c_int, c_void_p]),
    "dpx_fourier_solve": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                  c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "dpx_cfft2": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dpx_grad": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dpx_lincomb": (c_int, [c_void_p, c_int, POINTER(c_void_p), POINTER(c_float), POINTER(c_void_p), c_int, c_long, c_void_p]),
    "dpx_mul": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_long, c_int, c_void_p]),
    "dpx_wss_prox": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_long, c_void_p]),
    "dpx_mul_color": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_long, c_void_p]),
    "dpx_upsample_zero": (c_int, [c_void_p, c_void_p, c_int, c_long, c_int, c_int, c_void_p]),
    "dpx_cplx_mul": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_long, c_int, c_void_p]),
    "dpx_sisr_update": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_float, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dpx_cplx_scale": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_long, c_int, c_void_p]),
    "dpx_csmri_update": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_float, c_int, c_long, c_void_p]),
    "dpx_cplx_lincomb": (c_int, [c_void_p, c_int, c_int, POINTER(c_void_p), POINTER(c_int), POINTER(c_float), c_long, c_void_p]),
    "dpx_bdot": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_long, c_void_p, c_void_p]),
    "dpx_bdot_ws_bytes": (c_size_t, [c_int, c_long]),
    "dpx_bgram": (c_int, [c_void_p, c_void_p, c_int, c_long, c_void_p, c_void_p]),
    "dpx_bdot_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_long, c_void_p, c_void_p]),
    "dpx_bdot_f64_ws_bytes": (c_size_t, [c_int, c_long]),
    "dpx_bgram_f64": (c_int, [c_void_p, c_void_p, c_int, c_long, c_void_p, c_void_p]),
    "dpx_lincomb_f64": (c_int, [c_void_p, c_int, POINTER(c_void_p), POINTER(c_double), POINTER(c_void_p), c_int, c_long, c_void_p]),
    "dpx_absmax": (c_int, [c_void_p, c_void_p, c_long, c_int, c_void_p, c_void_p]),
    "dpx_ffdnet_f16_overflow": (c_int, [c_int]),
    "dpx_otf_grad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dpx_comm_use_library": (c_int, [c_char_p]),
    "dpx_comm_unique_id": (c_int, [c_void_p]),
    "dpx_comm_init": (c_int, [POINTER(c_void_p), c_void_p, c_int, c_int]),
    "dpx_comm_destroy": (c_int, [c_void_p]),
    "dpx_comm_rank": (c_int, [c_void_p]),
    "dpx_comm_world": (c_int, [c_void_p]),
    "dpx_comm_broadcast": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "dpx_comm_allgather": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dpx_comm_scatter": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "dpx_zero": (c_int, [c_void_p, c_size_t, c_void_p]),
    "dpx_cg_state_bytes": (c_size_t, [c_int]),
    "dpx_cg_init": (c_int, [c_void_p, c_void_p, c_float, c_int, c_void_p]),
    "dpx_cg_test": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "dpx_cg_direction": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_long, c_void_p]),
    "dpx_cg_update": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_long, c_void_p]),
    "dpx_cg_masked_fft_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "dpx_cg_masked_fft_supported": (c_int, [c_int, c_int, c_int]),
    "dpx_cg_masked_fft": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_float, c_float, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "dpx_prox": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int, c_long, c_void_p]),
    "dpx_prox_bwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int, c_long, c_void_p]),
    "dpx_fourier_apply_inv": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "dpx_admm_rhs": (c_int, [c_void_p, c_void_p, c_void_p, POINTER(Term), c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dpx_admm_zupdate": (c_int, [c_void_p, POINTER(Term), c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dpx_admm_zupdate_rhs": (c_int, [c_void_p, POINTER(Term), c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dpx_admm_bwd_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "dpx_admm_zupdate_bwd": (c_int, [c_void_p, POINTER(BwdTerm), c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dpx_admm_solve_rho_grad": (c_int, [c_void_p, c_void_p, POINTER(c_int), c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dpx_admm_rhs_bwd": (c_int, [c_void_p, c_void_p, c_void_p, POINTER(c_int), c_int, POINTER(c_void_p), POINTER(c_void_p), c_void_p,
                                 c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dpx_split_rhs": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, POINTER(Term), c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dpx_pc_dual": (c_int, [c_void_p, POINTER(Term), c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dpx_admm_iter_config": (c_int, [c_int, c_int]),
    "dpx_admm_iter_share": (c_int, [c_int]),
    "dpx_admm_iter_bands": (c_int, [c_int, c_int, c_int]),
    "dpx_admm_iter_supported": (c_int, [c_int, c_int, POINTER(Term), c_int]),
    "dpx_admm_seed_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dpx_admm_seed_rows_fresh": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dpx_rfft_rows": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dpx_admm_iter_cols": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dpx_admm_iter_rows": (c_int, [c_void_p, c_void_p, POINTER(Term), c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                   c_void_p, c_void_p]),
    "dpx_admm_run": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, POINTER(Term), c_int, c_void_p, POINTER(c_void_p), c_float,
                             c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dpx_admm_run_chains": (c_int, [POINTER(Chain), c_int, c_void_p, c_int, c_float, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dpx_streams_concurrent": (c_int, [c_void_p, c_void_p]),
    "dpx_stream_fork": (c_int, [c_void_p, POINTER(c_void_p), c_int]),
    "dpx_stream_join": (c_int, [c_void_p, POINTER(c_void_p), c_int]),
    "dpx_admm_unrolled_hist_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "dpx_admm_unrolled_forward": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int), POINTER(c_int), POINTER(c_float), c_int,
                                          c_void_p, POINTER(c_void_p), c_int, c_void_p, c_void_p, c_float, c_int, c_int, c_int, c_int, c_void_p,
                                          c_void_p, c_void_p, c_void_p]),
    "dpx_admm_rhs_fresh": (c_int, [c_void_p, c_void_p, c_void_p, POINTER(c_int), c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dpx_admm_unrolled_bwd_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "dpx_admm_unrolled_backward": (c_int, [c_void_p, c_void_p, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), c_void_p,
                                           c_void_p, POINTER(c_void_p), POINTER(c_void_p), c_int, POINTER(c_int), POINTER(c_int), POINTER(c_float),
                                           c_int, c_void_p, POINTER(c_void_p), c_int, c_void_p, c_float, c_int, c_int, c_int, c_int, c_void_p,
                                           c_void_p, c_void_p, c_void_p]),
    "dpx_admm_unrolled_hist_bytes_bf16": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "dpx_admm_unrolled_work_bytes_bf16": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "dpx_admm_unrolled_forward_bf16": (c_int, [c_void_p, c_void_p, c_void_p, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p),
                                               POINTER(c_int), POINTER(c_int), POINTER(c_float), c_int, c_void_p, POINTER(c_void_p), c_int, c_void_p,
                                               c_void_p, c_float, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dpx_admm_unrolled_bwd_ws_bytes_bf16": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "dpx_admm_unrolled_backward_bf16": (c_int, [c_void_p, c_void_p, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), c_void_p,
                                           c_void_p, POINTER(c_void_p), POINTER(c_void_p), c_int, POINTER(c_int), POINTER(c_int), POINTER(c_float),
                                           c_int, c_void_p, POINTER(c_void_p), c_int, c_void_p, c_float, c_int, c_int, c_int, c_int, c_void_p,
                                           c_void_p, c_void_p, c_void_p]),
    "dpx_ffdnet_packed_bytes": (c_size_t, [c_int, c_int, c_int]),
    "dpx_ffdnet_pack": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_void_p), c_int, c_int, c_int, c_void_p]),
    "dpx_ffdnet_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "dpx_ffdnet_bf16_packed_bytes": (c_size_t, [c_int, c_int, c_int]),
    "dpx_ffdnet_bf16_pack": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_void_p), c_int, c_int, c_int, c_int, c_void_p]),
    "dpx_ffdnet_bf16_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "dpx_ffdnet_forward_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dpx_ffdnet_bf16_acts_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "dpx_ffdnet_forward_bf16_save": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dpx_ffdnet_bf16_packed_transposed_bytes": (c_size_t, [c_int, c_int, c_int]),
    "dpx_ffdnet_bf16_pack_transposed": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dpx_ffdnet_bf16_bwd_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "dpx_ffdnet_backward_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                 c_void_p]),
    "dpx_ffdnet_bf16_bwd_w_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "dpx_conv3x3_wgrad_c8_ws_bytes": (c_size_t, [c_int, c_int]),
    "dpx_conv3x3_wgrad_c8": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p,
                                     c_void_p]),
    "dpx_ffdnet_backward_bf16_w": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                           c_int, c_void_p, c_void_p]),
    "dpx_admm_pnp_iter": (c_int, [c_void_p, c_void_p, POINTER(Term), c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p,
                                  c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dpx_admm_cg_pnp_iter": (c_int, [c_void_p, c_void_p, c_void_p, POINTER(Term), c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float,
                                     c_float, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_int, c_int, c_void_p]),
    "dpx_admm_cg_pnp_iter_folds": (c_int, [c_int, c_int]),
    "dpx_conv_packed_bytes": (c_size_t, [c_int, c_int, c_int]),
    "dpx_conv_pack": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "dpx_conv2d": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dpx_conv2d_wgrad_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "dpx_conv2d_wgrad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dpx_conv2d_leaky": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dpx_maxpool2": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dpx_maxpool2_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dpx_copy_channels": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dpx_upsample2_into": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dpx_upsample2_into_bwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dpx_leaky_relu_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_long, c_float, c_void_p]),
    "dpx_space_to_depth": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dpx_depth_to_space": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dpx_ffdnet_acts_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "dpx_ffdnet_forward_save": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dpx_ffdnet_packed_transposed_bytes": (c_size_t, [c_int, c_int, c_int]),
    "dpx_ffdnet_pack_transposed": (c_int, [c_void_p, POINTER(c_void_p), c_int, c_int, c_int, c_void_p]),
    "dpx_ffdnet_bwd_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "dpx_ffdnet_backward": (c_int, [c_void_p, c_void_p, c_void_p, POINTER(c_void_p), POINTER(c_void_p), c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dpx_ffdnet_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
}


class Library:
    def __init__(self, path):
        self.path = path
        self.cdll = ctypes.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(self.cdll, name)       # AttributeError = symbol missing from the build
            fn.restype, fn.argtypes = res, args

    def call(self, name, *args):
        rc = getattr(self.cdll, name)(*args)
        if rc != 0:
            raise DpxError(f"{name} failed ({rc}): {self.cdll.dpx_last_error().decode()}")

    def query(self, name, *args):
        return getattr(self.cdll, name)(*args)


_lib = None
_host_pointers = False      # True only under tests/emul


def lib() -> Library:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DpxError(f"{LIB_PATH} is missing: build the HIP extension first (python __graft_entry__.py). "
                           "This backend has no CPU or PyTorch fallback.")
        _lib = Library(LIB_PATH)
    return _lib


def _inject_for_tests(library, host_pointers):
    """tests/emul only."""
    global _lib, _host_pointers
    _lib, _host_pointers = library, host_pointers


def host_mode():
    return _host_pointers


# ---- range trap of the split-f16 denoiser arithmetic (FFDNet.compute_mode = "f16x2" / "f16x2w") -----------------------------------
# FFDNet.compute_mode -> `mode` of the C ABI (dpx_ffdnet_forward_bf16, dpx_admm_pnp_iter, dpx_admm_cg_pnp_iter)
FFDNET_MODES = {"f32": 0, "bf16x3": 6, "bf16": 1, "f16x2": 3, "f16x2w": 4}
F16_MODES = ("f16x2", "f16x2w")                    # the modes whose operands must stay inside the binary16 range
_f16_pending = False
_solve_depth = 0
_solve_epoch = 0


def solve_epoch():
    """a number that identifies the outermost solve in progress, None outside of one (host-side caches of things that cannot be
    watched -- a NumPy observation -- are re-validated once per solve, and on every use outside of a solve)"""
    return _solve_epoch if _solve_depth > 0 else None


def note_f16_launch():
    global _f16_pending
    _f16_pending = True


def check_f16_range(where):
    """after a solve / a direct denoiser call that ran split-f16 layers: fail loudly if an operand left the binary16 range"""
    global _f16_pending
    if not _f16_pending or _solve_depth > 0:
        return
    _f16_pending = False
    if lib().query("dpx_ffdnet_f16_overflow", 1):
        raise F16RangeError(f"{where}: an activation or weight left the binary16 range (|x| > 6e4) in the split-f16 arithmetic of the "
                            "FFDNet layers -- that result is invalid.  Set `<denoiser>.model.compute_mode = 'bf16x3'` (any range, "
                            "~1.4x slower) and rerun")


def f16_fallback(modules, where, stacklevel=3):
    """The networks among ``modules`` that ran split-f16 are switched to split-bf16 (fp32's range, six products instead of three)
    for good and the caller re-runs; returns False when there is nothing to switch (or a network asks to raise instead:
    ``model.f16_fallback = 'raise'``), in which case the caller re-raises."""
    import warnings
    nets = [m for m in modules if getattr(m, "compute_mode", None) in F16_MODES]
    if not nets or any(getattr(m, "f16_fallback", "bf16x3") == "raise" for m in nets):
        return False
    for m in nets:
        m.compute_mode = "bf16x3"
    warnings.warn(f"{where}: an operand left the binary16 range of the split-f16 FFDNet arithmetic; re-running on split-bf16 "
                  "(compute_mode = 'bf16x3', any range, ~1.4x slower) -- the network keeps that mode", RuntimeWarning, stacklevel=stacklevel)
    return True


_bwd_f16_nets = []                                  # networks whose split-f16 backward pass launched in the autograd pass in progress


def note_f16_backward(net):
    """A split-f16 backward pass of ``net`` (dpx_ffdnet_backward_bf16[_w], mode 3) has been launched inside the running autograd pass: when
    that pass ends the range trap is read ONCE (one synchronisation per backward pass, none per layer); if a scaled gradient left the
    binary16 range the gradients are invalid -- the networks concerned fall back to the split-bf16 backward arithmetic for good and
    F16RangeError comes out of ``loss.backward()`` (dp.train's loop repeats the step; anybody else's loop should)."""
    import torch
    if not _bwd_f16_nets:
        torch.autograd.Variable._execution_engine.queue_callback(_check_f16_backward)
    if not any(m is net for m in _bwd_f16_nets):
        _bwd_f16_nets.append(net)


def _check_f16_backward():
    nets = list(_bwd_f16_nets)
    del _bwd_f16_nets[:]
    if lib().query("dpx_ffdnet_f16_overflow", 1):
        for m in nets:
            m.backward_mode = "bf16x3"
        raise F16RangeError("backward: an operand (a gradient scaled to max |g| in [8, 16), a weight, or an activation of a forward pass not "
                            "checked since) left the binary16 range in the split-f16 arithmetic of the FFDNet layers -- the gradients of this "
                            "backward pass are invalid.  The network's backward pass now runs split-bf16 (`backward_mode = 'bf16x3'`): repeat "
                            "the step")


class solve_scope:
    """defers the range check to the end of the outermost solve (one device synchronisation per solve instead of one per layer)"""

    def __init__(self, where):
        self.where = where

    def __enter__(self):
        global _solve_depth, _solve_epoch
        if _solve_depth == 0:
            _solve_epoch += 1
        _solve_depth += 1

    def __exit__(self, et, ev, tb):
        global _solve_depth
        _solve_depth -= 1
        if et is None:
            check_f16_range(self.where)
        return False


# ---- tuning knobs of the library (include/dpx.h "tuning knobs"): per process, take effect at the next call ---------------------
def tune_get(name):
    v = c_int(0)
    lib().call("dpx_tune_get", name.encode(), ctypes.byref(v))
    return v.value


def tune_set(name, value):
    lib().call("dpx_tune_set", name.encode(), int(value))


def tune_names():
    L = lib()
    return [L.query("dpx_tune_name", i).decode() for i in range(L.query("dpx_tune_count"))]


class tuned:
    """``with tuned(cg_unfused=1): ...`` -- sets knobs for the block and restores the previous values (not thread-safe: the
    registry is per process)"""

    def __init__(self, **knobs):
        self.knobs, self.old = knobs, {}

    def __enter__(self):
        for k, v in self.knobs.items():
            self.old[k] = tune_get(k)
            tune_set(k, v)
        return self

    def __exit__(self, et, ev, tb):
        for k, v in self.old.items():
            tune_set(k, v)
        return False


def host_mode_skip_fast_cg():
    """DPX_GENERIC_CG=1 keeps the x-update on the generic cg() loop (A/B timing and tests of both paths)"""
    return bool(os.environ.get("DPX_GENERIC_CG"))


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """the hipStream_t PyTorch would launch on now (raw handle: torch.cuda.current_stream() builds a Stream object per call,
    ~4 us of the ~11 us a backend call costs on the host)"""
    if _host_pointers:
        return None
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def device_guard(device):
    """context manager making `device` the current HIP device (no-op under tests/emul and for an unspecific / CPU device)"""
    import contextlib
    device = torch.device(device) if not isinstance(device, torch.device) else device
    if _host_pointers or device.type != "cuda":
        return contextlib.nullcontext()
    return torch.cuda.device(device)


def require(t: torch.Tensor, dtype=torch.float32, what="tensor"):
    if not isinstance(t, torch.Tensor):
        raise DpxError(f"{what}: expected a torch.Tensor, got {type(t)}")
    if not _host_pointers and not t.is_cuda:
        raise DpxError(f"{what} lives on {t.device}: the MI355X backend only runs on HIP devices "
                       "(no CPU fallback); pass device='cuda'")
    if not _host_pointers and t.device.index != torch.cuda.current_device():
        raise DpxError(f"{what} lives on {t.device} but the current HIP device is cuda:{torch.cuda.current_device()}: the backend "
                       f"launches on the current device's stream -- wrap the call in `with torch.cuda.device({t.device.index}):` "
                       "(Algorithm.solve does this for its own device)")
    if dtype is not None and t.dtype != dtype:
        raise DpxError(f"{what}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise DpxError(f"{what} must be contiguous")
    return t


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def default_device():
    if _host_pointers:
        return torch.device("cpu")
    if not torch.cuda.is_available():
        raise DpxError("no HIP device visible: the MI355X backend has no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())
