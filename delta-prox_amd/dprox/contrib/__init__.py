"""Small helpers of the reference's ``dprox.contrib`` that the hot-path examples use
(reference dprox/contrib/restoration.py:14-45)."""
import numpy as np


def fspecial_gaussian(hsize, sigma):
    """MATLAB fspecial('gaussian', hsize, sigma)"""
    r = (hsize - 1.0) / 2.0
    ax = np.arange(-r, r + 1)
    xx, yy = np.meshgrid(ax, ax)
    h = np.exp(-(xx * xx + yy * yy) / (2.0 * sigma * sigma))
    h[h < np.finfo(float).eps * h.max()] = 0
    s = h.sum()
    return h / s if s != 0 else h


def sample(name="face", return_tensor=True):
    """the reference's sample images (contrib/restoration.py:8-18: SciPy's ``face`` / ``ascent``) as float [0, 1] data -- taken from
    whichever SciPy entry point still ships them; this package bundles no image data"""
    data = None
    for mod in ("scipy.datasets", "scipy.misc"):
        try:
            data = getattr(__import__(mod, fromlist=[name]), name)()
            break
        except (ImportError, AttributeError, OSError):         # (scipy.datasets needs `pooch` and a download)
            continue
    if data is None:
        raise FileNotFoundError(f"sample image {name!r}: neither scipy.datasets nor scipy.misc provides it in this environment")
    s = np.asarray(data).astype("float32") / 255
    if return_tensor:
        from ..utils import to_torch_tensor
        s = to_torch_tensor(s, batch=True).float()
    return s


def point_spread_function(ksize, sigma):
    return np.expand_dims(fspecial_gaussian(ksize, sigma), axis=2).astype("float32")


def blurring(img, psf):
    """circular blur of an NCHW tensor with ``psf`` through the backend's own conv operator"""
    from ..linop import Variable, conv
    op = conv(Variable(), psf).to(img.device)
    return op.forward(img.contiguous().float())


def mosaicing(img):
    """Bayer (RGGB) mosaic of an NCHW RGB tensor through the backend's own mask multiply (reference
    dprox/contrib/restoration.py:71-97)"""
    from ..linop import Variable, mosaic
    op = mosaic(Variable())
    return op.forward(img.contiguous().float())


def masked_fft(arg, mask):
    """Subsampled centred orthonormal Fourier operator  A x = mask * fft2(x)  with adjoint  A^H y = real(ifft2(mask * y))
    as a LinOp whose every pass is a HIP kernel (``dpx_cfft2`` + ``dpx_cplx_scale``): the CS-MRI forward model of
    config 4.  The reference has no such class -- its users write it with eager torch ops through the LinOp plugin
    protocol (``class MaskedFFT(LinOp)`` in the tests); this is the same operator without PyTorch arithmetic."""
    from .. import _ops as ops
    from ..linop import LinOp

    class _MaskedFFT(LinOp):
        is_masked_fft = True            # lets least_squares route a CG x-update over this operator to dpx_cg_masked_fft

        def __init__(self, a, m):
            super().__init__([a])
            self.mask = m

        def _plane_mask(self, ref):
            """the mask as one plane per image or per batch: a broadcastable mask (a column profile [1,1,1,W], a row profile) is
            expanded once to [.., H, W] -- the kernels take whole planes"""
            import torch
            m = self.mask
            H, W = int(ref.shape[-2]), int(ref.shape[-1])
            if isinstance(m, torch.Tensor) and m.ndim >= 2 and tuple(m.shape[-2:]) != (H, W):
                m = m.expand(*m.shape[:-2], H, W).contiguous()
                self.mask = m
            return m

        def forward(self, x, **kw):
            return ops.cplx_scale(ops.cfft2(x, inverse=False, centred=True, ortho=True), self._plane_mask(x))

        def adjoint(self, y, **kw):
            z = ops.cfft2(ops.cplx_scale(y.contiguous(), self._plane_mask(y)), inverse=True, centred=True, ortho=True)
            return ops.clincomb([(1.0, z)], out_complex=False)

    return _MaskedFFT(arg, mask)


from . import csmri                                                             # noqa: E402,F401  (contrib.csmri.*, reference contrib/csmri.py)
