"""``conv`` and ``grad``: circular convolution by a fixed kernel and its gradient special case
(reference dprox/linop/conv.py:15-56, dprox/linop/grad.py:8-23).

* the OTF is produced on the device by a direct fp64 DFT of the kernel (``dpx_psf2otf``) and cached per
  input shape like ``conv._FB``;
* forward / adjoint are one r2c -> multiply -> c2r pipeline of hand-written kernels (``dpx_fft_conv``);
* ``grad(dim=0|1)`` is the 2-tap circular stencil x[n+1]-x[n] / y[n-1]-y[n] that the reference evaluates
  with two full complex FFTs; ``grad(dim=2)`` keeps the reference's behaviour (psf2otf also transforms
  the channel axis, which turns it into a per-channel scale) by going through the OTF path.
"""
import numpy as np
import torch

from .. import _ops as ops
from ..utils import to_ndarray
from .node import LinOp


class conv(LinOp):
    """Circular convolution of the input with a kernel (2-D, or HWC)."""

    def __init__(self, arg, kernel):
        self.kernel = to_ndarray(kernel)
        self.cache = {}
        super().__init__([arg])

    def _tables(self, shape, device):
        key = (tuple(shape), str(device))
        if key not in self.cache:
            _, C, H, W = shape
            otf = ops.make_otf(self.kernel, C, H, W, device)
            self.cache[key] = otf
        return self.cache[key]

    def forward(self, input, **kwargs):
        return ops.fft_conv(input, self._tables(input.shape, input.device), conj=False)

    def adjoint(self, input, **kwargs):
        return ops.fft_conv(input, self._tables(input.shape, input.device), conj=True)

    def is_diag(self, freq=False):
        return freq and self.input_nodes[0].is_diag(freq)

    def get_diag(self, x, freq=False):
        """|OTF|^2 as a full [1,C,H,W] array (conv.py:46-53); the solver itself accumulates the
        half-spectrum table directly (``accumulate_diag``)."""
        assert freq
        _, C, H, W = x.shape
        d = ops.new_diag(C, H, W, self.device)
        ops.accumulate_diag(d, self.kernel, 1.0, C, H, W)
        return ops.diag_to_full(d, C, H, W).to(self.device)

    def accumulate_diag(self, diag, weight, C, H, W):
        return ops.accumulate_diag(diag, self.kernel, weight, C, H, W)

    def norm_bound(self, input_mags):
        return float(np.max(np.abs(self.kernel))) * input_mags[0]


class grad(conv):
    """gradient along dim 0 (height), 1 (width) or 2 (channel)"""

    def __init__(self, arg, dim=1):
        if dim not in (0, 1, 2):
            raise ValueError("dim must be 0(Height) or 1(Width) or 2 (Channel)")
        D = np.array([1, -1], dtype=np.int64).reshape(1, 1, 2)
        D = np.swapaxes(D, dim, -1)
        self.dim = dim
        super().__init__(arg, kernel=torch.from_numpy(np.ascontiguousarray(D)))

    def forward(self, input, **kwargs):
        if self.dim == 2:
            return conv.forward(self, input)
        return ops.grad(input, self.dim, adjoint=False)

    def adjoint(self, input, **kwargs):
        if self.dim == 2:
            return conv.adjoint(self, input)
        return ops.grad(input, self.dim, adjoint=True)

    def norm_bound(self, input_mags):
        return 2.0 * input_mags[0]
