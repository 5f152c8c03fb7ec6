"""Synthetic inputs shared by bench.py, the tests and the golden-vector generator.

NumPy only (no torch, no reference, no oracle).  Definitions follow SURVEY.md section 8(d):
ground truth = clipped sum of 6 random cosines + 8 random rectangles, PSF = 15x15 Gaussian
(sigma 5), observation = circular blur + N(0, (2/255)^2) noise, all from one
``numpy.random.RandomState`` (draw order matters).
"""
import numpy as np
import torch


def fspecial_gaussian(hsize=15, sigma=5.0):
    """MATLAB fspecial('gaussian') -- restates reference dprox/contrib/restoration.py:34-45 (float64)."""
    r = (hsize - 1.0) / 2.0
    ax = np.arange(-r, r + 1)
    xx, yy = np.meshgrid(ax, ax)
    h = np.exp(-(xx * xx + yy * yy) / (2.0 * sigma * sigma))
    h[h < np.finfo(float).eps * h.max()] = 0
    s = h.sum()
    return h / s if s != 0 else h


def point_spread_function(ksize=15, sigma=5.0):
    """HxWx1 float32 PSF -- reference dprox/contrib/restoration.py:21-22."""
    return fspecial_gaussian(ksize, sigma)[:, :, None].astype("float32")


def synth(rng, B, C, H, W):
    """SURVEY.md Appendix B ground-truth generator, NCHW float32 in [0, 1]."""
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    out = np.zeros((B, C, H, W), np.float32)
    for b in range(B):
        for c in range(C):
            img = np.zeros((H, W))
            for _ in range(6):
                f, g = rng.randint(0, 9), rng.randint(0, 9)
                a = rng.rand()
                ph = rng.rand() * 2 * np.pi
                img += a * np.cos(2 * np.pi * (f * yy / H + g * xx / W) + ph)
            img = img / 3 + 0.5
            for _ in range(8):
                y0, x0 = rng.randint(0, H), rng.randint(0, W)
                h, w = rng.randint(H // 16, H // 3), rng.randint(W // 16, W // 3)
                img[y0:y0 + h, x0:x0 + w] += rng.uniform(-0.3, 0.3)
            out[b, c] = np.clip(img, 0, 1)
    return out


def synth_detail(rng, B, C, H, W, base=256):
    """The same generator with its detail scaled to the plane: ``synth`` draws cosines of at most 8 cycles per IMAGE and rectangles
    of 1/16 .. 1/3 of the image whatever H is, so at 1024 x 1024 a 15 x 15 blur hardly touches it (blurred input 36.8 dB, nothing
    for a deconvolution to win).  Here the cosines have up to 8 cycles per `base` pixels (frequencies x H/base) and the rectangles
    keep their size in pixels (1/16 .. 1/3 of `base`, (H/base)(W/base) x 8 of them): the local statistics of a 256 x 256 `synth`
    image at any plane size.  H = W = base draws the same kind of image as ``synth`` (not the same numbers)."""
    sy, sx = max(H // base, 1), max(W // base, 1)
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    out = np.zeros((B, C, H, W), np.float32)
    for b in range(B):
        for c in range(C):
            img = np.zeros((H, W))
            for _ in range(6):
                f, g = rng.randint(0, 9) * sy, rng.randint(0, 9) * sx
                a = rng.rand()
                ph = rng.rand() * 2 * np.pi
                img += a * np.cos(2 * np.pi * (f * yy / H + g * xx / W) + ph)
            img = img / 3 + 0.5
            for _ in range(8 * sy * sx):
                y0, x0 = rng.randint(0, H), rng.randint(0, W)
                h, w = rng.randint(base // 16, base // 3), rng.randint(base // 16, base // 3)
                img[y0:y0 + h, x0:x0 + w] += rng.uniform(-0.3, 0.3)
            out[b, c] = np.clip(img, 0, 1)
    return out


def circular_blur(img, psf2d):
    """Circular (wrap-around) convolution of NCHW ``img`` with a centred 2-D PSF, float64 FFT internally.
    Equivalent to scipy.ndimage.convolve(mode='wrap') as used by the reference's ``blurring``
    (dprox/contrib/restoration.py:25-31)."""
    H, W = img.shape[-2:]
    kh, kw = psf2d.shape
    pad = np.zeros((H, W))
    pad[:kh, :kw] = psf2d
    pad = np.roll(pad, (-(kh // 2), -(kw // 2)), axis=(0, 1))
    otf = np.fft.rfft2(pad)
    return np.fft.irfft2(np.fft.rfft2(img.astype(np.float64), axes=(-2, -1)) * otf, s=(H, W), axes=(-2, -1))


def deconv_case(B, C, H, W, seed=2023, noise=2.0 / 255.0, ksize=15, ksigma=5.0):
    """(gt, b, psf): the config-1/2 deconvolution inputs; ``x0 = b``."""
    rng = np.random.RandomState(seed)
    gt = synth(rng, B, C, H, W)
    psf = point_spread_function(ksize, ksigma)
    b = circular_blur(gt, psf[:, :, 0].astype(np.float64))
    b = (b + rng.randn(B, C, H, W) * noise).astype(np.float32)
    return gt, b, psf


def csmri_case(B, H, W, seed=2023, rate=0.25, center=32, noise=0.01):
    """(gt, mask, y): config-4 style CS-MRI inputs.  gt real NCHW (C=1), mask {0,1} float32 [1,1,H,W],
    y = mask * centred-orthonormal-FFT(gt) + complex noise (complex64)."""
    rng = np.random.RandomState(seed)
    gt = synth(rng, B, 1, H, W)
    mask = (rng.rand(H, W) < rate).astype(np.float32)
    c0, c1 = H // 2 - center // 2, W // 2 - center // 2
    mask[c0:c0 + center, c1:c1 + center] = 1.0
    k = np.fft.fftshift(np.fft.fft2(np.fft.ifftshift(gt.astype(np.float64), axes=(-2, -1)), norm="ortho"), axes=(-2, -1))
    nz = (rng.randn(B, 1, H, W) + 1j * rng.randn(B, 1, H, W)) * noise
    y = (mask[None, None] * (k + nz)).astype(np.complex64)
    return gt, mask[None, None], y


# ---- seeded stand-ins for pretrained denoiser checkpoints (no network in the build / test environments) --------------------
def ffdnet_weights(seed=7, in_nc=3, out_nc=3, nc=96, nb=12, gain=0.5):
    """Seeded stand-in for the (un-downloadable) pretrained checkpoint: scaled He-normal weights,
    small biases, drawn from ``numpy.random.RandomState(seed)`` (platform-stable).  Layer shapes are
    FFDNet's -- reference network_ffdnet.py:43-47: (in_nc*4+1 -> nc), (nb-2) x (nc -> nc), (nc -> out_nc*4)."""
    rng = np.random.RandomState(seed)
    chans = [in_nc * 4 + 1] + [nc] * (nb - 1) + [out_nc * 4]
    layers = []
    for cin, cout in zip(chans[:-1], chans[1:]):
        w = (rng.standard_normal((cout, cin, 3, 3)) * gain * np.sqrt(2.0 / (cin * 9))).astype(np.float32)
        b = (rng.standard_normal((cout,)) * 0.01).astype(np.float32)
        layers.append((w, b))
    return layers


def drunet_weights(seed=21, in_nc=4, out_nc=3, nc=(64, 128, 256, 512), nb=4, gain=0.4):
    """Seeded weights in the reference's state-dict layout (models/network_unet.py:67-104: no biases): He-normal x gain
    (the checkpoints cannot be downloaded here)."""
    rng = np.random.RandomState(seed)
    sd = {}

    def conv_w(name, co, ci, k):
        sd[name] = torch.from_numpy((rng.randn(co, ci, k, k) * gain * np.sqrt(2.0 / (ci * k * k))).astype(np.float32))

    conv_w("m_head.weight", nc[0], in_nc, 3)
    for lvl in range(3):
        for i in range(nb):
            conv_w(f"m_down{lvl + 1}.{i}.res.0.weight", nc[lvl], nc[lvl], 3)
            conv_w(f"m_down{lvl + 1}.{i}.res.2.weight", nc[lvl], nc[lvl], 3)
        conv_w(f"m_down{lvl + 1}.{nb}.weight", nc[lvl + 1], nc[lvl], 2)
    for i in range(nb):
        conv_w(f"m_body.{i}.res.0.weight", nc[3], nc[3], 3)
        conv_w(f"m_body.{i}.res.2.weight", nc[3], nc[3], 3)
    for lvl in (3, 2, 1):
        w = (rng.randn(nc[lvl], nc[lvl - 1], 2, 2) * gain * np.sqrt(2.0 / (nc[lvl] * 4))).astype(np.float32)   # ConvTranspose2d: [in, out, 2, 2]
        sd[f"m_up{lvl}.0.weight"] = torch.from_numpy(w)
        for i in range(nb):
            conv_w(f"m_up{lvl}.{i + 1}.res.0.weight", nc[lvl - 1], nc[lvl - 1], 3)
            conv_w(f"m_up{lvl}.{i + 1}.res.2.weight", nc[lvl - 1], nc[lvl - 1], 3)
    conv_w("m_tail.weight", out_nc, nc[0], 3)
    return sd


def ircnn_weights(seed=31, in_nc=1, out_nc=1, nc=64, gain=0.5):
    """Seeded IRCNN state dict (reference models/network_dncnn.py:94-109: seven biased 3x3 convolutions, dilations
    1,2,3,4,3,2,1; Conv2d modules at the even indices of `model`)."""
    rng = np.random.RandomState(seed)
    chans = [in_nc] + [nc] * 6 + [out_nc]
    sd = {}
    for i, (ci, co) in enumerate(zip(chans[:-1], chans[1:])):
        sd[f"model.{2 * i}.weight"] = torch.from_numpy((rng.randn(co, ci, 3, 3) * gain * np.sqrt(2.0 / (ci * 9))).astype(np.float32))
        sd[f"model.{2 * i}.bias"] = torch.from_numpy((rng.randn(co) * 0.01).astype(np.float32))
    return sd


def unet_weights(seed=41, in_ch=2, out_ch=1, gain=0.7):
    """Seeded state dict of the reference's U-Net denoiser (models/unet/unet.py:34-135: ConvBlocks of three biased 3x3
    convolutions + LeakyReLU(0.2), widths 32..512, bilinear up-sampling, 1x1 output convolution), He-normal x gain."""
    rng = np.random.RandomState(seed)
    sd = {}

    def block(prefix, ci, co):
        for i in range(3):
            cin = ci if i == 0 else co
            sd[f"{prefix}.conv-{i}.conv2d.weight"] = torch.from_numpy((rng.randn(co, cin, 3, 3) * gain * np.sqrt(2.0 / (cin * 9))).astype(np.float32))
            sd[f"{prefix}.conv-{i}.conv2d.bias"] = torch.from_numpy((rng.randn(co) * 0.01).astype(np.float32))

    block("inc.conv", in_ch, 32)
    for k, (ci, co) in enumerate(((32, 64), (64, 128), (128, 256), (256, 512))):
        block(f"down{k + 1}.mpconv.1", ci, co)
    for k, (ci, co) in enumerate(((512 + 256, 256), (256 + 128, 128), (128 + 64, 64), (64 + 32, 32))):
        block(f"up{k + 1}.conv", ci, co)
    sd["outc.conv.weight"] = torch.from_numpy((rng.randn(out_ch, 32, 1, 1) * gain * np.sqrt(2.0 / 32)).astype(np.float32))
    sd["outc.conv.bias"] = torch.from_numpy((rng.randn(out_ch) * 0.01).astype(np.float32))
    return sd
