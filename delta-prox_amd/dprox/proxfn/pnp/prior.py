"""``deep_prior`` -- plug-and-play prior: the prox is a denoiser call with sigma = lam
(reference dprox/proxfn/pnp/prior.py:14-89)."""
import copy
import os

import torch
import torch.nn as nn

from ... import _ops as ops
from ...utils import safe_sqrt
from ..core import ProxFn
from .denoisers import Augment, DRUNetDenoiser, FFDNetColorDenoiser, FFDNetDenoiser, IRCNNDenoiser, UNetDenoiser

CACHE_DIR = os.path.join(os.path.expanduser("~"), ".cache", "dprox")


def get_denoiser(type):
    """pretrained weights are read from the reference's cache layout (~/.cache/dprox/pnp_denoisers/*.pth);
    nothing is downloaded."""
    table = {"ffdnet": ("ffdnet_gray.pth", FFDNetDenoiser), "ffdnet_color": ("ffdnet_color.pth", FFDNetColorDenoiser),
             "drunet": ("drunet_gray.pth", lambda p: DRUNetDenoiser(1, p)), "drunet_color": ("drunet_color.pth", lambda p: DRUNetDenoiser(3, p)),
             "ircnn": ("ircnn_gray.pth", lambda p: IRCNNDenoiser(1, p)), "unet": ("unet-nm.pt", UNetDenoiser)}
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
    """Plug-and-play prior ``g(K x)`` whose proximal operator is one call of a denoiser at noise level ``sigma = lam``
    (``sqrt=True``: ``sigma = sqrt(lam)``).

    Same constructor as the reference (prior.py:40-63).  Behavioural notes for the HIP backend: ``x8`` wraps the denoiser in
    ``Augment`` (one dihedral transform per call, cycling); a complex iterate is reduced to its real part by ``dpx_cplx_lincomb`` (prior.py:79); with
    ``unroll_step=k`` the prior owns k independent copies of the denoiser and uses copy ``self.step`` (set by the solver
    before every iteration)."""

    def __init__(self, linop, denoiser="ffdnet", x8=False, clamp=False, trainable=False, unroll_step=None, sqrt=False):
        super().__init__(linop)
        self.name = denoiser
        self.x8, self.clamp, self.sqrt = bool(x8), bool(clamp), bool(sqrt)
        self.denoiser = self._resolve(denoiser, trainable, self.x8)
        self.unroll = unroll_step is not None
        if self.unroll:
            self.denoisers = nn.ModuleList(copy.deepcopy(self.denoiser) for _ in range(int(unroll_step)))

    @staticmethod
    def _resolve(denoiser, trainable, x8=False):
        """a registry name or a ready ``Denoiser``; frozen (eval mode, no parameter gradients) unless ``trainable``
        (prior.py:57-60)"""
        net = get_denoiser(denoiser) if isinstance(denoiser, str) else denoiser
        if x8:
            net = Augment(net)                               # prior.py:55-56
        if not trainable:
            net.eval()
            net.requires_grad_(False)
        else:
            net.requires_grad_(True)                       # the HIP modules create their parameters frozen
        return net

    def _reload(self, shape=None):
        if self.x8:
            self.denoiser.reset()

    def _active_denoiser(self):
        return self.denoisers[self.step] if self.unroll else self.denoiser

    def _noise_level(self, lam):
        return safe_sqrt(lam) if self.sqrt else lam

    def eval(self, v=None):
        if v is not None:
            raise NotImplementedError("deep prior cannot be explictly evaluated")
        return super().eval()                            # nn.Module.eval()

    def _prox(self, v: torch.Tensor, lam: torch.Tensor):
        """v: [N, C, H, W] or [N, H, W] (treated as one channel); lam: 0-d or [N]"""
        shape = tuple(v.shape)
        if torch.is_complex(v):
            v = ops.clincomb([(1.0, v)], out_complex=False)
        if self.clamp:
            v = v.clamp(0, 1)
        batch = v if v.ndim == 4 else v.unsqueeze(1)
        cleaned = self._active_denoiser().denoise(batch.contiguous(), self._noise_level(lam))
        return cleaned.type_as(v).reshape(shape)

    def __repr__(self):
        return f'deep_prior(denoiser="{self.name}", unroll={self.unroll})'
