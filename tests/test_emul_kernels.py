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


def test_known_answers():
    pc.case_known_answers(DEV)


@pytest.mark.parametrize("B", [1, 4])
def test_cg(B):
    pc.case_cg(DEV, B)


def test_ffdnet_mfma_layout():
    pc.case_ffdnet(DEV, which=("batch",))       # 2x3x16x24: the f32 MFMA lane layouts + pack / unpack, emulated


def test_other_algorithms():
    pc.case_other_algorithms(DEV)


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


def test_unrolled_gradients():
    pc.case_unrolled_grads(DEV)


def test_unrolled_solver_learned_params():
    pc.case_unrolled_solver(DEV)


@pytest.mark.skipif(not __import__("os").environ.get("DPX_EMUL_SLOW"), reason="~4 min on the emulator (runs on the GPU in test_gpu_parity); DPX_EMUL_SLOW=1 enables it")
def test_ffdnet_weight_gradients():
    pc.case_ffdnet_weight_grads(DEV)


def test_ffdnet_backward():
    pc.case_ffdnet_grads(DEV, which=("even",))      # one small image: the emulator runs MFMA layers at ~1 s each


def test_adjoint_dot_product():
    pc.case_adjoint_dot(DEV, shape=(1, 3, 24, 20))
