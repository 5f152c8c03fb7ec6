"""Centred orthonormal 2-D FFT pair (reference dprox/utils/misc.py:164-193): the building block of
user-defined CS-MRI operators (mask * fft2(x)).  Complex c2c on device tensors through torch.fft
(rocFFT); the real-image solver hot path uses the hand-written kernels of libdpx_hip instead."""
import torch


def fft2(x):
    x = torch.fft.ifftshift(x, dim=(-2, -1))
    x = torch.fft.fft2(x, norm="ortho")
    return torch.fft.fftshift(x, dim=(-2, -1))


def ifft2(x):
    x = torch.fft.ifftshift(x, dim=(-2, -1))
    x = torch.fft.ifft2(x, norm="ortho")
    return torch.fft.fftshift(x, dim=(-2, -1))
