from .krylov import bdot, cg, cg2, expand, pcg, ravel

# (reference dprox/linalg/solve/__init__.py:1-22 also lists plss / plssw / minres: outside the hot path, not built)
__all__ = available_solvers = ["cg", "cg2", "pcg"]

SOLVERS = {"cg": cg, "cg2": cg2, "pcg": pcg}
