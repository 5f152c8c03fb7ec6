"""PSNR with data range 1 (restates reference dprox/utils/metrics.py:68-70 without skimage)."""
import numpy as np
import torch


def _np(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


def mse(a, b):
    return float(np.mean((_np(a).astype(np.float64) - _np(b).astype(np.float64)) ** 2))


def psnr(output, target, data_range=1.0):
    m = mse(output, target)
    return float("inf") if m == 0 else float(10.0 * np.log10(data_range ** 2 / m))


def psnr_per_image(output, target):
    o, t = _np(output).astype(np.float64), _np(target).astype(np.float64)
    m = ((o - t) ** 2).reshape(o.shape[0], -1).mean(axis=1)
    return 10.0 * np.log10(1.0 / m)
