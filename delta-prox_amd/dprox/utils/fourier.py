"""Centred orthonormal 2-D FFT pair (reference dprox/utils/misc.py:164-193): the building block of
user-defined CS-MRI operators (mask * fft2(x)).  One fused pass of hand-written kernels per direction
(``dpx_cfft2``: the ifftshift / fftshift index rotations and the 1/sqrt(HW) factor are folded into the
loads and stores of the row / column transforms)."""
from .. import _ops as ops


def fft2(x):
    return ops.cfft2(x, inverse=False, centred=True, ortho=True)


def ifft2(x):
    return ops.cfft2(x, inverse=True, centred=True, ortho=True)
