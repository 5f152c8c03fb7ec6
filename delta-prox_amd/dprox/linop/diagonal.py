"""Image-domain diagonal operators: ``mosaic`` (Bayer colour-filter-array mask, reference
dprox/linop/subsample.py:8-99) and ``mul_elementwise`` (reference dprox/linop/mul.py:46-73).
Forward and adjoint are one masked / weighted multiply on the device (``dpx_mul``)."""
import numpy as np
import torch

from .. import _ops as ops
from ..utils import to_torch_tensor
from .leaf import Placeholder
from .node import LinOp


def masks_CFA_Bayer(shape, pattern="RGGB"):
    """boolean R, G, B masks of an H x W sensor (subsample.py:33-39)"""
    channels = {c: np.zeros(shape) for c in "RGB"}
    for channel, (y, x) in zip(pattern, [(0, 0), (0, 1), (1, 0), (1, 1)]):
        channels[channel][y::2, x::2] = 1
    return tuple(channels[c].astype(bool) for c in "RGB")


class mosaic(LinOp):
    def __init__(self, arg):
        super().__init__([arg])
        self.cache = {}

    def _mask(self, shape, device=None):
        hw = tuple(shape[-2:])
        key = (hw, str(device))
        if key not in self.cache:
            R_m, G_m, B_m = masks_CFA_Bayer(hw)
            m = np.stack([R_m, G_m, B_m], axis=0)[None].astype("float32")          # [1,3,H,W]
            t = torch.from_numpy(m)
            self.cache[key] = t.to(device) if device is not None else t
        return self.cache[key]

    def forward(self, input, **kwargs):
        return ops.mul(input.contiguous(), self._mask(input.shape, input.device))

    def adjoint(self, input, **kwargs):
        return self.forward(input)

    def is_gram_diag(self, freq=False):
        return self.is_self_diag(freq) and self.input_nodes[0].is_diag(freq)

    def is_self_diag(self, freq=False):
        return not freq

    def get_diag(self, x, freq=False):
        assert not freq
        return self._mask(x.shape, self.device)

    def norm_bound(self, input_mags):
        return input_mags[0]


class mul_elementwise(LinOp):
    def __init__(self, arg, w):
        super().__init__([arg])
        self._w = w
        if isinstance(w, Placeholder):
            self._w.change(lambda val: setattr(self, "w", val))
            self.w = None
        else:
            self.w = to_torch_tensor(w, batch=True).float()

    def forward(self, x, **kwargs):
        return ops.mul(x.contiguous(), self.w.to(x.device))

    def adjoint(self, x, **kwargs):
        return self.forward(x)

    def is_diag(self, freq=False):
        return not freq and self.input_nodes[0].is_diag(freq)

    def get_diag(self, x, freq=False):
        if not freq:
            return self.w.to(x.device)
        return None


class mul_color(LinOp):
    """channel mixing by a spectral response function srf [C, C2] (reference dprox/linop/mul.py:13-43):
    forward = srf.T @ x over the channel axis, adjoint = srf @ x (``dpx_mul_color``).  Like the reference, the SRF is a 2-D
    tensor (given directly or through a Placeholder)."""

    def __init__(self, arg, srf):
        super().__init__([arg])
        self._srf = srf
        if isinstance(srf, Placeholder):
            self.srf = None
            self._srf.change(lambda val: setattr(self, "srf", val))
        else:
            self.srf = torch.as_tensor(srf).float()
        if self.srf is not None and self.srf.ndim != 2:
            raise ValueError("mul_color: srf must be a 2-D [C, C2] array")

    def forward(self, x, **kwargs):
        return ops.mul_color(x.contiguous(), self.srf, transpose=False)

    def adjoint(self, x, **kwargs):
        return ops.mul_color(x.contiguous(), self.srf, transpose=True)
