import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def peak_rel(a, b):
    """max|a-b| / max|b| -- the parity metric of SURVEY.md section 8d"""
    a = np.asarray(a)
    b = np.asarray(b)
    den = np.abs(b).max()
    return float(np.abs(a - b).max() / (den if den > 0 else 1.0))


def l2_rel(a, b):
    a = np.asarray(a, np.complex128 if np.iscomplexobj(a) or np.iscomplexobj(b) else np.float64)
    b = np.asarray(b, a.dtype)
    den = np.linalg.norm(b.ravel())
    return float(np.linalg.norm((a - b).ravel()) / (den if den > 0 else 1.0))


def assert_parity(got, want, tol=1e-5, what=""):
    """north_star tolerance: 1e-5 relative, taken peak-relative and L2-relative
    per output tensor (element-wise relative error is meaningless at near-empty
    bins: the reference itself is 1e-4 off float64 there)"""
    assert np.shape(got) == np.shape(want), f"{what}: shape {np.shape(got)} vs {np.shape(want)}"
    assert np.all(np.isfinite(got)), f"{what}: non-finite values"
    p, l = peak_rel(got, want), l2_rel(got, want)
    assert p <= tol and l <= tol, f"{what}: peak-rel {p:.3e}, l2-rel {l:.3e} > {tol}"


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def have_ref():
    from oracle import ref
    return ref.available()
