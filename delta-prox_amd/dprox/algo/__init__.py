from .api import SOLVERS, UNROLL_BF16, Problem, UnrolledSolver, build_unrolled_solver, compile, specialize
from .driver import Algorithm
from .gradient import ProximalGradientDescent
from .splitting import ADMM, HQS, ADMM_vxu, LinearizedADMM, PockChambolle
from .tune.dpir import log_descent
from .training import TrainLoop, train
