"""CPU-only: the unchanged HIP kernel sources compiled for the host by the SIMT emulator (tests/emul)
run the same parity cases as the GPU tests (small fixtures only).  Catches indexing / barrier / layout
bugs without a GPU; the authoritative numerics check is tests/test_gpu_parity.py on a real MI355X."""
import pytest

import emul_util


@pytest.fixture(scope="module", autouse=True)
def _emulated():
    emul_util.use_emulator()
    yield


import parity_cases as pc  # noqa: E402

DEV = "cpu"


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_linops(tag):
    pc.case_linops(DEV, tag)


def test_prox():
    pc.case_prox(DEV)


def test_solve_direct():
    pc.case_solve_direct(DEV)


def test_admm_tv_small_fused():
    pc.case_admm_tv_small(DEV, True)


def test_admm_tv_config1_pow2_path():
    pc.case_admm_tv_config1(DEV)          # 256x256: register-radix power-of-two kernels


def test_admm_tv_misc():
    pc.case_admm_tv_misc(DEV)


def test_pgd():
    pc.case_pgd(DEV)


def test_column_length_768():
    pc.case_h768(DEV, tiny=True)


def test_768_wide_rows_on_the_two_kernel_iteration():
    pc.case_w768_two_kernel(DEV, B=1, methods=("admm",))


def test_merged_z_rhs():
    pc.case_merged_z_rhs(DEV)


def test_generic_interleaved():
    pc.case_generic_interleaved(DEV, sizes=((1, 3, 45, 35), (2, 1, 30, 44), (1, 1, 52, 26), (1, 1, 63, 28), (1, 1, 24, 34), (1, 1, 1100, 24)),
                                oracle_sizes=((1, 3, 45, 35),))


def test_other_plane_sizes():
    pc.case_other_plane_sizes(DEV, sizes=((384, 256), (256, 768)), channels=1)


def test_hqs_two_kernel():
    pc.case_hqs_pow2(DEV)


def test_fresh_state_shortcut_matches_the_general_seed():
    pc.case_fresh_state(DEV)


def test_solve_returns_x_alone_from_an_x_only_last_pass():
    pc.case_solve_x_only(DEV, iters=3, configs=(("admm", 0), ("admm", 30), ("hqs", 0)))


def test_sub_batch_chains_are_bit_identical_to_one_chain():
    pc.case_sub_batch_chains(DEV, shapes=((3, 1, 256, 256),), iters=3, methods=("admm",), nchs=(2,), twice=False)      # (a 1- and a 2-image chain)


def test_hqs_no_dual_row_kernel():
    pc.case_hqs_nodual_kernel(DEV, iters=3, nterms_list=(2, 4))


def test_admm_vxu_two_kernel():
    pc.case_vxu_two_kernel(DEV, iters=3, nterms_list=(3,))


def test_pgd_streaming_row_kernel():
    pc.case_pgd_streaming_rows(DEV)


def test_pgd_pow2_fused():
    pc.case_pgd_pow2(DEV, tiny=True)


def test_known_answers():
    pc.case_known_answers(DEV)


@pytest.mark.parametrize("B", [1, 4])
def test_cg(B):
    pc.case_cg(DEV, B)


def test_numpy_observation_edited_in_place_is_seen():
    pc.case_numpy_observation_edits(DEV)


def test_cg_both_branches_of_the_fused_call():
    pc.case_cg_branches(DEV, quick=True)


def test_cg_matvec_with_one_wave_transforms_320():
    pc.case_cg_wave_fft(DEV, sizes=(320,), B=1, iters=3)


def test_cg_masked_fft_odd_and_per_image_masks():
    pc.case_cg_masked_fft_shapes(DEV)


def test_dense_systems_cg_cg2_pcg_and_implicit_gradients():
    pc.case_dense_krylov(DEV)


def test_doe_psf_gradient_through_the_unrolled_solver():
    pc.case_doe_psf_grad(DEV)
    pc.case_doe_op_autograd(DEV)


def test_builtin_linear_nodes_under_autograd():
    pc.case_linop_autograd(DEV)


def test_ffdnet_split_f16_and_its_range_trap():
    pc.case_ffdnet_f16_split(DEV, tiny=True)


def test_ffdnet_winograd_layers():
    pc.case_ffdnet_winograd(DEV, tiny=True)


def test_linear_solve_implicit_backward():
    pc.case_linear_solve_grad(DEV)


def test_ffdnet_mfma_layout():
    pc.case_ffdnet(DEV, which=("batch",))       # 2x3x16x24: the f32 MFMA lane layouts + pack / unpack, emulated


def test_other_algorithms():
    pc.case_other_algorithms(DEV)


def test_plug_and_play_cg_loop_forms_are_bit_identical():
    pc.case_split_cg_loop_forms(DEV, B=1, H=32, W=32, iters=3)          # (the emulator's share; the GPU suite runs 2 x 48 x 48 and config 4's shard)


def test_csmri_custom_admm():
    pc.case_csmri(DEV, solve=False)          # the 4-iteration solve with the 15-layer gray FFDNet runs on the GPU only


def test_conv2d_generic():
    pc.case_conv2d_generic(DEV)


def test_tiny_shapes():
    pc.case_tiny_shapes(DEV)


def test_conv_doe():
    pc.case_conv_doe(DEV)


def test_sisr_super_resolution():
    pc.case_sisr(DEV, solve=False)


def test_mosaic_joint_demosaic_deconv():
    pc.case_mosaic_jd(DEV, solve=False)      # the ADMM + CG + FFDNet solve runs on the GPU only


def test_unrolled_backward_fused_stage_matches_the_staged_loop():
    # (32 x 48 planes: off the two-kernel backward iteration -- its row-kernel choice is exercised at 256 x 256 below)
    pc.case_unrolled_bwd_fused_vs_staged(DEV, modes=[m for m in pc.UNROLL_BWD_MODES if m[0] != "lock-step bands"], term_sets=("tv+nn", "nn+l1"))


def test_unrolled_backward_two_kernel_iteration_256():
    # 2 x 2 planes of 256 x 256, bands of 14 and 5 rows (the last band of a plane shorter / a band inside one lock-step round)
    modes = [m for m in pc.UNROLL_BWD_MODES if m[0] in ("default", "lock-step bands", "staged")]
    pc.case_unrolled_bwd_fused_vs_staged(DEV, shape=(1, 2, 256, 256), K=3, term_sets=("tv+nn",), dtypes=("f32",), modes=modes)
    pc.case_unrolled_bwd_fused_vs_staged(DEV, shape=(1, 1, 256, 256), K=2, term_sets=("nn+l1",), dtypes=("bf16",), modes=modes, band=5)


def test_unrolled_gradients():
    pc.case_unrolled_grads(DEV)


def test_unrolled_gradients_bf16_history():
    pc.case_unrolled_grads_bf16(DEV)


def test_unrolled_solver_learned_params():
    pc.case_unrolled_solver(DEV)


def test_train_driver_and_resume(tmp_path):
    pc.case_train(DEV, str(tmp_path))


@pytest.mark.skipif(not __import__("os").environ.get("DPX_EMUL_SLOW"), reason="~4 min on the emulator (runs on the GPU in test_gpu_parity); DPX_EMUL_SLOW=1 enables it")
def test_ffdnet_weight_gradients():
    pc.case_ffdnet_weight_grads(DEV)


def test_wgrad_c8_kernel():
    pc.case_wgrad_c8(DEV, tiny=True)


def test_ffdnet_split_backward():
    pc.case_ffdnet_split_backward(DEV, tiny=True)


@pytest.mark.skipif(not __import__("os").environ.get("DPX_EMUL_SLOW"), reason="~1 min on the emulator (runs on the GPU in test_gpu_parity; the split-kernel backward below stays); DPX_EMUL_SLOW=1 enables it")
def test_ffdnet_backward():
    pc.case_ffdnet_grads(DEV, which=("even",))      # one small image: the emulator runs MFMA layers at ~1 s each


def test_adjoint_dot_product():
    pc.case_adjoint_dot(DEV, shape=(1, 3, 24, 20))


def test_unet_layer_kernels_vs_torch():
    """the U-Net's non-convolution layers (dpx_unet.hip) against ATen on the CPU: MaxPool2d(2) with odd sizes and ties,
    bilinear x2 (align_corners=True) + zero pad + concat, their adjoints (dot-product test + autograd), LeakyReLU backward"""
    import torch
    import torch.nn.functional as F
    from dprox import _ops as ops
    torch.manual_seed(3)
    x = torch.rand(2, 3, 9, 11)
    x[0, 0, 0, 0] = x[0, 0, 0, 1] = 0.99                      # a tie: the gradient goes to the first maximum
    assert torch.equal(ops.maxpool2(x), F.max_pool2d(x, 2))
    xr = x.clone().requires_grad_(True)
    gy = torch.randn(2, 3, 4, 5)
    (F.max_pool2d(xr, 2) * gy).sum().backward()
    assert torch.equal(ops.maxpool2_bwd(x, gy), xr.grad)
    for (h, w), (H, W) in (((4, 5), (9, 11)), ((1, 3), (2, 6)), ((6, 6), (12, 13))):
        low, skip = torch.rand(2, 3, h, w), torch.rand(2, 2, H, W)
        up = F.interpolate(low, scale_factor=2, mode="bilinear", align_corners=True)
        dy, dx = H - 2 * h, W - 2 * w
        ref = torch.cat([skip, F.pad(up, (dx // 2, dx - dx // 2, dy // 2, dy - dy // 2))], dim=1)
        got = ops.concat_skip_upsampled(skip, low)
        assert float((got - ref).abs().max()) <= 2e-7, float((got - ref).abs().max())
        g = torch.randn_like(ref)
        gskip, glow = ops.concat_skip_upsampled_bwd(g, 2, (h, w))
        lr = low.clone().requires_grad_(True)
        upr = F.pad(F.interpolate(lr, scale_factor=2, mode="bilinear", align_corners=True), (dx // 2, dx - dx // 2, dy // 2, dy - dy // 2))
        (upr * g[:, 2:]).sum().backward()
        assert torch.equal(gskip, g[:, :2]) and float((glow - lr.grad).abs().max()) <= 2e-6
    y, g = torch.randn(2, 3, 5, 7), torch.randn(2, 3, 5, 7)
    assert torch.equal(ops.leaky_relu_bwd(y, g, 0.2), torch.where(y > 0, g, 0.2 * g))


def test_leaky_conv_layer_vs_torch():
    """dpx_conv2d_leaky (bias + LeakyReLU(0.2) epilogue on the MFMA kernel) against F.conv2d on a small layer"""
    import torch
    import torch.nn.functional as F
    from dprox import _ops as ops
    torch.manual_seed(4)
    x, w, b = torch.randn(1, 2, 9, 13), torch.randn(32, 2, 3, 3) * 0.3, torch.randn(32) * 0.1
    blob = ops.conv_pack(w.reshape(32, 2, 9).contiguous(), b, 9)
    ref = F.leaky_relu(F.conv2d(x, w, b, padding=1), 0.2)
    got = ops.conv2d_leaky(x, blob, 32, 0.2)
    assert float((got - ref).abs().max()) <= 1e-5 * float(ref.abs().max())


def test_row_parallel_kernel_is_bit_identical_to_the_streaming_kernel():
    pc.case_row_parallel_kernel(DEV, shapes=((1, 1, 256, 256),), iters=2, methods=("admm",), nterms_list=(3,), hfirst=(True, False))
    pc.case_row_parallel_kernel(DEV, shapes=((1, 1, 256, 256),), iters=2, methods=("hqs", "admm_vxu"), nterms_list=(2,), hfirst=(True,))


def test_merged_loop_keeps_the_callers_duals():
    pc.case_merged_loop_keeps_the_callers_duals(DEV)
