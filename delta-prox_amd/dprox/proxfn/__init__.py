from .core import ProxFn
from .fast import csmri, sisr
from .pnp import Denoiser, Denoiser2D, DRUNetDenoiser, FFDNetColorDenoiser, FFDNetDenoiser, UNetDenoiser, deep_prior
from .quadratic import ext_sum_squares, least_squares, sum_squares, weighted_sum_squares
from .simple import nonneg, norm1, norm2, soft_threshold
