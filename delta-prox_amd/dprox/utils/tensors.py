"""Host-side tensor conversion helpers (reference dprox/utils/misc.py:42-161)."""
import random

import numpy as np
import torch

from .containar import is_dp_tensor, tensor


def seed_everything(seed):
    np.random.seed(seed)
    torch.manual_seed(seed)
    random.seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def to_nn_parameter(*tensors, requires_grad=False):
    ps = [torch.nn.Parameter(t, requires_grad=requires_grad) for t in tensors]
    return ps[0] if len(ps) == 1 else ps


def _hwc_like(shape):
    return len(shape) == 3 and shape[2] in (1, 3)


def batchify(out):
    """HWC (C in {1,3}) -> 1CHW; anything else just gains a leading batch dim (misc.py:42-59)."""
    if is_dp_tensor(out):
        return out
    if _hwc_like(out.shape):
        out = out.permute(2, 0, 1)
    return out.unsqueeze(0)


def to_torch_tensor(x, batch=False):
    """numpy / list / tensor -> tagged torch tensor; with ``batch`` HWC arrays become NCHW (misc.py:62-96)."""
    if is_dp_tensor(x):
        return x
    if isinstance(x, torch.Tensor):
        out = x
    elif isinstance(x, np.ndarray):
        out = tensor(x.copy())
    else:
        out = tensor(x)
    if batch:
        if _hwc_like(out.shape):
            out = out.permute(2, 0, 1)
        if out.ndim < 4:
            out = out.unsqueeze(0)
    out.is_dp_tensor = True
    return out


def debatchify(out, squeeze):
    """BCHW -> CHW -> HWC -> HW on numpy arrays (misc.py:99-124)."""
    if out.ndim == 4:
        out = out.squeeze(0)
    if out.ndim == 3:
        if out.shape[0] in (1, 3):
            out = out.transpose(1, 2, 0)
        if out.shape[2] == 1 and squeeze:
            out = out.squeeze(2)
    return out


def to_ndarray(x, debatch=False, squeeze=False):
    """(misc.py:127-155) tensors keep their dtype, ndarrays become float32, the rest goes through np.array."""
    if isinstance(x, torch.Tensor):
        out = x.detach().cpu().numpy()
    elif isinstance(x, np.ndarray):
        out = x.astype("float32")
    else:
        out = np.array(x)
    return debatchify(out, squeeze) if debatch else out


def safe_sqrt(x, eps=1e-8):
    return torch.sqrt(torch.clamp(x, min=eps))
