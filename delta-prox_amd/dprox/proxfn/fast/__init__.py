"""Closed-form data terms (reference dprox/proxfn/fast/): the ones on the hot path of the reference's own pipelines."""
from .csmri import csmri
from .sr import sisr
