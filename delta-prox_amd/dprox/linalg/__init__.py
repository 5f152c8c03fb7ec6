from . import solve
from .implicit import LinearSolve, LinearSolveConfig, linear_solve
