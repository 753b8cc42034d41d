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


def parity_log(what, measured, bar, kind="peak/l2", extra=None):
    """AFX_PARITY_LOG=<file>: one JSON line per parity decision (label, measured error, bar) -- the
    source of the table of non-default bars in DESIGN.md section 2 (tools/parity_table.py)"""
    path = os.environ.get("AFX_PARITY_LOG")
    if not path:
        return
    import json
    rec = {"what": str(what), "measured": float(measured), "bar": float(bar), "kind": kind,
           "test": os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0]}
    if extra:
        rec.update(extra)
    with open(path, "a") as f:
        f.write(json.dumps(rec) + "\n")


# AFX_HOSTSTUB=1 (tests/test_hoststub.py): the suite runs against the sanitizer build of the host objects with a
# stand-in device layer -- results are meaningless there, only the host code paths matter
HOSTSTUB = os.environ.get("AFX_HOSTSTUB") == "1"
if HOSTSTUB:
    sys.modules["torch"] = None  # no device: tests that need torch tensors end with ImportError (and torch does not
    #                              survive an LD_PRELOADed sanitizer runtime)


# AFX_EMULATED=1 (tests/test_emulated_kernels.py): the suite runs against the library whose kernels are the device code
# compiled for the host (tests/emu) -- no torch, but results are real: parity assertions stay on
EMULATED = os.environ.get("AFX_EMULATED") == "1"
if EMULATED:
    sys.modules["torch"] = None


def assert_parity(got, want, tol=1e-5, what=""):
    """north_star tolerance: 1e-5 relative, taken peak-relative and L2-relative
    per output tensor (element-wise relative error is meaningless at near-empty
    bins: the reference itself is 1e-4 off float64 there)"""
    assert np.shape(got) == np.shape(want), f"{what}: shape {np.shape(got)} vs {np.shape(want)}"
    if HOSTSTUB:
        return
    assert np.all(np.isfinite(got)), f"{what}: non-finite values"
    p, l = peak_rel(got, want), l2_rel(got, want)
    parity_log(what, max(p, l), tol)
    assert p <= tol and l <= tol, f"{what}: peak-rel {p:.3e}, l2-rel {l:.3e} > {tol}"


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def have_ref():
    from oracle import ref
    return ref.available()


def assert_istft_parity(got, want, gain_norm, what=""):
    """inverse STFT: every output sample is (sum of inverse-FFT samples x w^e) / (sum w^(e+1)), so the
    float32 rounding of the inverse FFT (~3e-7 of the frame peak) reaches it multiplied by the
    condition number gain / normaliser -- about 1 inside the signal, up to 1e5 at the few edge
    samples where the window sum is near the reference's 1e-6 clamp.  Bar: 1e-5 of the peak where
    that number is <= 30, 3e-7 x condition number elsewhere (any float32 implementation, the
    reference included, is that far from the exact result there)"""
    if HOSTSTUB:
        return
    gain, norm = gain_norm
    cond = np.asarray(gain) / np.asarray(norm)
    scale = np.abs(want).max()
    assert np.shape(got) == np.shape(want) and np.all(np.isfinite(got)), what
    d = np.abs(np.asarray(got, np.float64) - want) / scale
    tol = np.maximum(1e-5, 3e-7 * cond)
    bad = d > tol
    parity_log(what, float((d / tol).max()) * 1e-5, 1e-5, "istft: worst error / its bar, scaled to 1e-5",
               {"worst_where_cond_le_30": float(d[cond <= 30].max()) if (cond <= 30).any() else 0.0,
                "max_cond": float(cond.max())})
    assert not bad.any(), f"{what}: {int(bad.sum())} samples over tolerance, worst {d[bad].max():.3e} (cond {cond[bad].max():.1f})"
    assert (cond <= 30).mean() > 0.9 or len(cond) < 4096, "tolerance relaxed on too many samples"
