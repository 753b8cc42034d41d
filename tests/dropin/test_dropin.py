"""Drop-in boundary: the reference's OWN unmodified `python/audioflux` wrapper loads
libaudioflux_mi355x.so through `audioflux.fftlib.set_fft_lib(lib_ext='mi355x')`
(python/audioflux/fftlib.py:88-129, base.py:4-8) and every wrapper flow on the path returns what
the stock library (the compiled reference installed as lib/libaudioflux.so) returns.

The flows run in a fresh interpreter (tests/dropin/flows.py): the wrapper, ctypes and the two
libraries only -- no torch, no audioflux_amd.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.conftest import l2_rel, parity_log, peak_rel

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import flows  # noqa: E402  (imports nothing but numpy at module level)

FLOW_NAMES = [f.__name__[5:] for f in flows.FLOWS]


def _have_inputs():
    return (os.path.exists(flows.STOCK) and os.path.exists(flows.PRODUCT)
            and (os.path.exists(flows.WRAPPER_ZIP) or os.path.isdir("/root/reference/python/audioflux")))


needs_inputs = pytest.mark.skipif(
    not _have_inputs(), reason="needs oracle/_ref (make -C oracle) and the built product library")


def _run(tmp, mode):
    out = os.path.join(tmp, f"flows_{mode}.npz")
    env = dict(os.environ)
    env["AFX_HIP_RUNTIME"] = "system"
    res = subprocess.run([sys.executable, os.path.join(HERE, "flows.py"), os.path.join(tmp, "pkg"), out, mode],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500, env=env,
                         cwd=tmp)
    assert res.returncode == 0, f"flows.py {mode} died (rc {res.returncode}):\n{res.stdout[-4000:]}"
    data = np.load(out)
    return data, json.loads(str(data["meta"])), res.stdout


# ---------------------------------------------------------------------------------------------
# CPU: the wrapper imports, selects the product library, resolves everything, never crashes
@needs_inputs
def test_wrapper_resolves_every_symbol_and_survives_no_device(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("device present: the -2 path is not reachable (covered by the gpu flows)")
    _, meta, log = _run(str(tmp_path), "cpu")
    assert os.path.realpath(meta["lib"]) == os.path.realpath(flows.PRODUCT)
    assert meta["symbols"] >= 85
    assert meta["missing"] == [], f"wrapper looks up symbols the library does not export: {meta['missing']}"
    assert meta["survived"]
    calls = {c[0]: c for c in meta["calls"]}
    # constructors return (the wrapper ignores the -2 status), compute calls return zero-filled outputs
    for k in ("BFT()", "XXCC()", "CWT()", "CQT()", "Cepstrogram()", "STFT()", "MelSpectrogram()", "PWT()",
              "WSST()", "Synsq()", "Reassign()", "bft.bft", "xxcc.xxcc", "cwt.cwt", "cqt.cqt", "stft.stft"):
        assert calls[k][1] == "ok", calls[k]
    assert "no usable MI355X" in log  # the failure is reported on stderr, not swallowed


@needs_inputs
def test_symbol_table_matches_headers(tmp_path):
    """every symbol the wrapper binds for the path is declared in include/*.h"""
    flows.stage(str(tmp_path))
    names = flows.wrapper_symbols(str(tmp_path))
    decl = ""
    for root, _, files in os.walk(os.path.join(ROOT, "include")):
        for f in files:
            with open(os.path.join(root, f)) as fh:
                decl += fh.read()
    missing = [n for n in names if n + "(" not in decl.replace(" (", "(")]
    assert missing == [], missing


# ---------------------------------------------------------------------------------------------
# GPU: stock vs product through the same wrapper
@pytest.fixture(scope="module")
def both(tmp_path_factory):
    tmp = str(tmp_path_factory.mktemp("dropin"))
    data, meta, log = _run(tmp, "gpu")
    return data, meta, log


# bars other than plain 1e-5 peak + L2 (DESIGN section 2 table lists them with the measured values)
TOL = {}  # every flow output meets plain 1e-5 (measured worst: cepstrogram details 5.7e-6)
# outputs that are a scatter onto ROUNDED coordinates (WSST / synsq / reassign): a coefficient whose
# float32 coordinate sits on a .5 boundary lands in the neighbouring cell in any implementation whose
# transform is not bit-identical; the explained-difference proof is tests/test_{wsst,synsq,reassign}_gpu.py.
# Here: all but a small fraction of cells agree at 1e-5 and the displaced mass is small.
SCATTER = {("wsst", "wsst"), ("synsq", "synsq"), ("reassign", "reassign")}
INTEGER = {"bin", "T", "fft_length"}


@pytest.mark.gpu
@needs_inputs
def test_libraries_selected(both):
    _, meta, _ = both
    assert os.path.realpath(meta["stock_lib"]) == os.path.realpath(flows.STOCK)
    assert os.path.realpath(meta["mi355x_lib"]) == os.path.realpath(flows.PRODUCT)
    assert os.path.realpath(meta["object_lib"]) == os.path.realpath(flows.PRODUCT)
    assert meta["errors"] == {}, meta["errors"]


@pytest.mark.gpu
@needs_inputs
@pytest.mark.parametrize("flow", FLOW_NAMES)
def test_flow_matches_stock(both, flow):
    data, meta, _ = both
    assert f"stock/{flow}" not in meta["errors"] and f"mi355x/{flow}" not in meta["errors"], meta["errors"]
    keys = [k.split("/", 2)[2] for k in data.files if k.startswith(f"stock/{flow}/")]
    assert keys, f"flow {flow} produced nothing"
    report = []
    for k in keys:
        want, got = data[f"stock/{flow}/{k}"], data[f"mi355x/{flow}/{k}"]
        assert got.shape == want.shape and got.dtype == want.dtype, (flow, k, got.shape, want.shape)
        assert np.all(np.isfinite(got)), (flow, k)
        if k in INTEGER:
            assert np.array_equal(got, want), (flow, k)
            continue
        if (flow, k) in SCATTER:
            peak = np.abs(want).max()
            off = np.abs(got - want) > 1e-5 * peak
            moved = np.abs(got - want).sum() / np.abs(want).sum()
            report.append(f"{k}: {off.mean():.2e} of cells differ, displaced mass {moved:.2e}")
            parity_log(f"dropin {flow}/{k}", float(off.mean()), 1e-3, "fraction of cells beyond 1e-5 (scatter onto rounded coordinates)",
                       {"displaced_mass": float(moved), "displaced_mass_bar": 1e-4})
            assert off.mean() <= 1e-3 and moved <= 1e-4, (flow, k, off.mean(), moved)  # measured <= 2.5e-4 / 2.1e-5
            continue
        if (flow, k) == ("stft", "istft"):
            # edge samples divide by a window sum near the reference's clamp (tests/conftest.py); interior at 1e-5
            want, got = want[1024:-1024], got[1024:-1024]
        tol = TOL.get((flow, k), 1e-5)
        p, l = peak_rel(got, want), l2_rel(got, want)
        report.append(f"{k}: peak {p:.2e} l2 {l:.2e}")
        parity_log(f"dropin {flow}/{k}", max(p, l), tol)
        assert p <= tol and l <= tol, f"{flow}/{k}: peak-rel {p:.3e}, l2-rel {l:.3e} > {tol}"
    print(flow, "; ".join(report))
