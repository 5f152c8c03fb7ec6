from .denoisers import Denoiser, Denoiser2D, DRUNetDenoiser, FFDNet, FFDNetColorDenoiser, FFDNetDenoiser, IRCNN, IRCNNDenoiser, UNet, UNetDenoiser, UNetRes
from .prior import deep_prior, get_denoiser
