"""``dp.tensor`` / ``dp.array`` tagging (reference dprox/utils/containar.py:5-61): a tagged tensor is
assumed to be batched NCHW already and is never re-batchified by ``to_torch_tensor``."""
import numpy as np
import torch


def is_dp_array(x):
    return getattr(x, "is_dp_array", False) is True


def is_dp_tensor(x):
    return getattr(x, "is_dp_tensor", False) is True


class _TaggedArray(np.ndarray):
    is_dp_array = True


def array(*args, **kwargs):
    return np.array(*args, **kwargs).view(_TaggedArray)


def tensor(*args, **kwargs):
    out = torch.tensor(*args, **kwargs)
    out.is_dp_tensor = True
    return out
