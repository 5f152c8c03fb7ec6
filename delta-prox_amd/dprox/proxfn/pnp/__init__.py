from .denoisers import Denoiser, Denoiser2D, FFDNet, FFDNetColorDenoiser, FFDNetDenoiser
from .prior import deep_prior, get_denoiser
