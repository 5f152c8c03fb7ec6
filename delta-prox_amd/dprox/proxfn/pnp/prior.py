"""``deep_prior`` -- plug-and-play prior: the prox is a denoiser call with sigma = lam
(reference dprox/proxfn/pnp/prior.py:14-89)."""
import copy
import os

import torch
import torch.nn as nn

from ... import _ops as ops
from ...utils import safe_sqrt
from ..core import ProxFn
from .denoisers import DRUNetDenoiser, FFDNetColorDenoiser, FFDNetDenoiser, IRCNNDenoiser

CACHE_DIR = os.path.join(os.path.expanduser("~"), ".cache", "dprox")


def get_denoiser(type):
    """pretrained weights are read from the reference's cache layout (~/.cache/dprox/pnp_denoisers/*.pth);
    nothing is downloaded."""
    table = {"ffdnet": ("ffdnet_gray.pth", FFDNetDenoiser), "ffdnet_color": ("ffdnet_color.pth", FFDNetColorDenoiser),
             "drunet": ("drunet_gray.pth", lambda p: DRUNetDenoiser(1, p)), "drunet_color": ("drunet_color.pth", lambda p: DRUNetDenoiser(3, p)),
             "ircnn": ("ircnn_gray.pth", lambda p: IRCNNDenoiser(1, p))}
    if type not in table:
        raise ValueError(f"denoiser {type!r} is not built for the MI355X backend (have {sorted(table)}); "
                         "pass a Denoiser instance instead")
    fname, cls = table[type]
    path = os.path.join(CACHE_DIR, "pnp_denoisers", fname)
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} not found: place the checkpoint there or pass a Denoiser instance "
                                "(deep_prior(x, denoiser=FFDNetColorDenoiser(weights)))")
    return cls(path)


class deep_prior(ProxFn):
    def __init__(self, linop, denoiser="ffdnet", x8=False, clamp=False, trainable=False, unroll_step=None, sqrt=False):
        super().__init__(linop)
        self.name = denoiser
        self.denoiser = get_denoiser(denoiser) if isinstance(denoiser, str) else denoiser
        if x8:
            raise NotImplementedError("x8 test-time augmentation is outside the MI355X hot path")
        self.x8, self.clamp, self.sqrt = x8, clamp, sqrt
        if not trainable:
            self.denoiser.eval()
            self.denoiser.requires_grad_(False)
        self.unroll = unroll_step is not None
        if self.unroll:
            self.denoisers = nn.ModuleList([copy.deepcopy(self.denoiser) for _ in range(unroll_step)])

    def eval(self, v=None):
        if v is None:
            return super().eval()
        raise NotImplementedError("deep prior cannot be explictly evaluated")

    def _prox(self, v: torch.Tensor, lam: torch.Tensor):
        sigma = safe_sqrt(lam) if self.sqrt else lam
        if self.clamp:
            v = v.clamp(0, 1)
        if torch.is_complex(v):
            v = ops.clincomb([(1.0, v)], out_complex=False)          # v.real (prior.py:79)
        inp = v.unsqueeze(1) if v.ndim == 3 else v
        den = self.denoisers[self.step] if self.unroll else self.denoiser
        out = den.denoise(inp.contiguous(), sigma)
        return out.type_as(v).reshape(*v.shape)

    def __repr__(self):
        return f'deep_prior(denoiser="{self.name}", unroll={self.unroll})'
