"""Hyper-parameter schedules (reference dprox/algo/tune/dpir.py:13-39)."""
import numpy as np
import torch


def log_descent(upper, lower, iter=24, sigma=0.255 / 255, w=1.0, lam=0.23, sqrt=False):
    """sigma_t log-spaced from ``upper`` to ``lower`` (/255), rho_t = lam * sigma^2 / sigma_t^2;
    returns (rhos, sigmas) as float32 tensors, sigmas squared unless ``sqrt``."""
    s_log = np.logspace(np.log10(upper), np.log10(lower), iter).astype(np.float32)
    s_lin = np.linspace(upper, lower, iter).astype(np.float32)
    sig = (s_log * w + s_lin * (1 - w)) / 255.0
    rhos = [lam * (sigma ** 2) / (s ** 2) for s in sig]
    if not sqrt:
        sig = list(sig ** 2)
    return torch.tensor(rhos).float(), torch.tensor(np.asarray(sig)).float()
