"""Module path of the reference's helper collection (dprox/utils/misc.py:10-193): the same functions, defined in this package's
``tensors`` / ``fourier`` modules."""
from .fourier import fft2, ifft2                                                                  # noqa: F401
from .tensors import (batchify, debatchify, safe_sqrt, seed_everything, to_ndarray, to_nn_parameter,   # noqa: F401
                      to_torch_tensor)
