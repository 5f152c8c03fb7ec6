import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "delta-prox_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def rel_l2(a, b):
    a = np.asarray(a).astype(np.complex128 if np.iscomplexobj(a) or np.iscomplexobj(b) else np.float64)
    b = np.asarray(b).astype(a.dtype)
    return float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-30))


def assert_close(a, b, rel=1e-5, what=""):
    """SURVEY 8(a) acceptance: rel-L2 <= rel and max-abs <= rel * max|ref|."""
    a = np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    r = rel_l2(a, b)
    wide = np.complex128 if np.iscomplexobj(a) or np.iscomplexobj(b) else np.float64
    m = float(np.max(np.abs(a.astype(wide) - b.astype(wide)))) if a.size else 0.0
    scale = float(np.max(np.abs(b))) if b.size else 0.0
    assert r <= rel, f"{what}: rel-L2 {r:.3e} > {rel:.1e}"
    assert m <= rel * max(scale, 1e-30) * 4 + 1e-12, f"{what}: max-abs {m:.3e} vs {rel:.1e}*{scale:.3e}"


@pytest.fixture
def golden():
    return load_golden
