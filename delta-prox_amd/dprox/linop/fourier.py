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
        key = (tuple(shape[1:]), str(device))              # (the table of a plane set: shared by every batch size)
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


def _doe_padded(psf, shape):
    """psf2otf2's spatial part (conv.py:59-78): zero-pad the [1,C,fh,fw] PSF to the image size with the reference's
    split of the padding (computed from the HEIGHT difference for both axes), then ifftshift over ALL dims -- which
    also rotates the channel axis (C = 3: by one), a quirk kept for parity."""
    import torch.nn.functional as F
    _, _, fh, fw = psf.shape
    H = shape[2]
    if H != fh:
        pad = (H - fh) / 2
        if (H - fh) % 2 != 0:
            pt = pl = int(np.ceil(pad))
            pb = pr = int(np.floor(pad))
        else:
            pt = pl = int(pad) + 1
            pb = pr = int(pad) - 1
        psf = F.pad(psf, [pl, pr, pt, pb], mode="constant")
    if tuple(psf.shape[-2:]) != tuple(shape[-2:]):
        raise ValueError(f"conv_doe: a {fh}x{fw} PSF padded like the reference gives {tuple(psf.shape[-2:])}, not the image size "
                         f"{tuple(shape[-2:])} (the reference needs H - fh == W - fw)")
    return torch.fft.ifftshift(psf)


class _FullOtf(torch.autograd.Function):
    """O = fft2(P) of the padded, shifted PSF P [1,C,H,W] (unnormalised; conv_doe's psf2otf2, linop/conv.py:59-78) as an autograd
    node: the x-updates of all iterations hand their dL/dO back to it, one adjoint transform turns the sum into dL/dP."""

    @staticmethod
    def forward(ctx, P):
        return ops.cfft2(P.contiguous(), inverse=False, centred=False, ortho=False)

    @staticmethod
    def backward(ctx, G):
        H, W = int(G.shape[-2]), int(G.shape[-1])
        # dL/dP[n] = Re sum_k G_k e^{+i theta_kn}  (G = dL/dRe O + i dL/dIm O): the unnormalised inverse transform
        g = ops.cfft2(G.contiguous(), inverse=True, centred=False, ortho=False)
        return ops.lincomb([(float(H * W), torch.view_as_real(g)[..., 0].contiguous())])



class _DoeConv(torch.autograd.Function):
    """y = F^-1(op(O) F(x)) through dpx_fft_conv with gradients w.r.t. the image and the (full) OTF -- conv_doe.forward / adjoint
    when the PSF or the image takes part in autograd"""

    @staticmethod
    def forward(ctx, img, O, tables, conj):
        ctx.tables, ctx.conj = tables, conj
        ctx.save_for_backward(img, O)
        return ops.fft_conv(img.contiguous(), tables, conj=conj)

    @staticmethod
    def backward(ctx, g):
        img, O = ctx.saved_tensors
        g = g.contiguous()
        g_img = ops.fft_conv(g, ctx.tables, conj=not ctx.conj) if ctx.needs_input_grad[0] else None
        g_O = None
        if ctx.needs_input_grad[1]:
            G = ops.cfft2(g, inverse=False, centred=False, ortho=False)
            X = ops.cfft2(img.contiguous(), inverse=False, centred=False, ortho=False)
            g_O = ops.otf_grad(G, None, X, O) if ctx.conj else ops.otf_grad(X, None, G, O)
        return g_img, g_O, None, None


class conv_doe(LinOp):
    """Circular convolution with a PSF given as a tensor / Placeholder [1,C,fh,fw] whose OTF is rebuilt on the device
    whenever the PSF changes (reference dprox/linop/conv.py:81-156; end-to-end optics, README.md:93-116).
    The transform of the padded PSF is ``dpx_cfft2``; forward / adjoint are the same ``dpx_fft_conv`` pipeline as ``conv``.
    ``circular=False`` (conv.py:100-108,121-129): the image is zero-padded to 2H x 2H, convolved circularly at that size and
    cropped back; ``get_diag`` stays the circular |OTF|^2 of the unpadded size, as in the reference."""

    def __init__(self, arg, psf, circular=True):
        super().__init__([arg])
        from .leaf import Placeholder
        self._psf = psf
        self.circular = bool(circular)
        self.cache = {}
        self._psf_gen = 0                                    # counts assignments of the placeholder: (generation, tensor version)
        if isinstance(psf, Placeholder):                     # identifies a PSF value (an address can be recycled by the allocator)
            self.psf = None

            def assign(val):
                self.psf = val
                self._psf_gen += 1
            self._psf.change(assign)
        else:
            from ..utils import to_torch_tensor
            self.psf = to_torch_tensor(psf, batch=True).float()

    def _full_otf(self, shape, device):
        psf = self.psf
        if psf is None:
            raise ValueError("conv_doe: the PSF placeholder has no value yet")
        ver = (self._psf_gen, id(psf), psf._version)       # (id: `op.psf = other_tensor` assigned directly, not through the Placeholder)
        if self.cache.get("version") != ver:
            self.cache = {"version": ver}                    # a new PSF value invalidates every size's OTF
        key = (tuple(shape[1:]), str(device))
        if key not in self.cache:
            _, C, H, W = shape
            P = _doe_padded(psf.detach().float().to(device), shape).expand(1, C, H, W).contiguous()
            full = ops.cfft2(P, inverse=False, centred=False, ortho=False)
            self.cache[key] = (full, ops.otf_from_full(full, C, H, W))
        return self.cache[key]

    def _tables(self, shape, device):
        return self._full_otf(shape, device)[1]

    def _own_tables_version(self):
        return None if self.psf is None else (self._psf_gen, id(self.psf), self.psf._version)

    def _convolve(self, img, conj):
        if self.circular:
            tables = self._tables(img.shape, img.device)
            psf = self.psf
            if torch.is_grad_enabled() and (img.requires_grad or (isinstance(psf, torch.Tensor) and psf.requires_grad)):
                if isinstance(psf, torch.Tensor) and psf.requires_grad:
                    C, H, W = img.shape[-3:]
                    O = _FullOtf.apply(_doe_padded(psf.float().to(img.device), img.shape).expand(1, C, H, W).contiguous())
                else:
                    O = self._full_otf(img.shape, img.device)[0]
                return _DoeConv.apply(img, O, tables, bool(conj))
            return ops.fft_conv(img, tables, conj=conj)
        import torch.nn.functional as F
        H, W = img.shape[-2:]
        side = 2 * H                                         # both axes are padded to twice the HEIGHT (conv.py:102-104)
        pt, pb = int(np.ceil((side - H) / 2)), int(np.floor((side - H) / 2))
        pl, pr = int(np.ceil((side - W) / 2)), int(np.floor((side - W) / 2))
        if min(pt, pb, pl, pr) < 1:
            raise ValueError(f"conv_doe(circular=False): a {H}x{W} image cannot be padded to {side}x{side} on every side")
        big = F.pad(img, [pl, pr, pt, pb], mode="constant").contiguous()
        out = ops.fft_conv(big, self._tables(big.shape, big.device), conj=conj)
        return out[:, :, pt:-pb, pl:-pr].contiguous()

    def forward(self, input, **kwargs):
        return self._convolve(input, False)

    def adjoint(self, input, **kwargs):
        return self._convolve(input, True)

    def is_diag(self, freq=False):
        return freq and self.input_nodes[0].is_diag(freq)

    def get_diag(self, x, freq=False):
        assert freq
        full, _ = self._full_otf(x.shape, x.device)
        return ops.clincomb([(1.0, ops.cplx_mul(full, full, conj_a=True))], out_complex=False)      # |OTF|^2, [1,C,H,W]

    def accumulate_diag(self, diag, weight, C, H, W):
        tab = ops.diag_from_full(self.get_diag(torch.empty(1, C, H, W, device=diag.device), True), C, H, W, diag.device)
        return ops.lincomb([(1.0, diag.reshape(1, -1)), (float(weight), tab.reshape(1, -1))]).reshape(-1)

    def norm_bound(self, input_mags):
        return float(self.psf.abs().max()) * input_mags[0]
