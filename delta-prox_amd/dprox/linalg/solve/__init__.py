from .krylov import bdot, cg, expand, ravel

__all__ = available_solvers = ["cg"]

SOLVERS = {"cg": cg}
