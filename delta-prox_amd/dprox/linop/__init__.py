from .arith import copy, scale, split, sum, vstack
from .blackbox import BlackBox, LinOpFactory
from .diagonal import masks_CFA_Bayer, mosaic, mul_color, mul_elementwise
from .fourier import conv, conv_doe, grad
from .graph import CompGraph, adjoint, eval, gram, validate
from .leaf import Constant, Placeholder, Variable
from .node import LinOp
