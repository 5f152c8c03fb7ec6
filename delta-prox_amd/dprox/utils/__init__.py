from .containar import array, is_dp_array, is_dp_tensor, tensor
from .fourier import fft2, ifft2
from .metrics import mse, psnr, psnr_per_image
from .tensors import (batchify, debatchify, safe_sqrt, seed_everything, to_ndarray, to_nn_parameter,
                      to_torch_tensor)
from . import misc                                                              # noqa: E402,F401  (reference module path)
