"""MI355X-native backend for the ADMM / proximal-gradient hot path of Delta-Prox.

Drop-in for the reference's user surface -- ``Variable, Placeholder, conv, grad, sum_squares, norm1,
nonneg, deep_prior, Problem, compile, specialize, LinOp, ProxFn, LinOpFactory, linalg`` -- with every
per-iteration arithmetic pass executed by hand-written gfx950 HIP kernels (``lib/libdpx_hip.so``,
C ABI in ``include/dpx.h``).  There is no CPU or PyTorch fallback.
"""
from . import contrib, linalg, utils
from .algo import *      # noqa: F401,F403
from .linop import *     # noqa: F401,F403
from .proxfn import *    # noqa: F401,F403
from .utils.containar import array, tensor

__version__ = "0.1.3+mi355x.1"
