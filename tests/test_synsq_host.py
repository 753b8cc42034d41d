"""CPU-only: the restatement of synsqObj_synsq pinned against the golden vectors under the
boundary-aware criterion of the GPU test (see tests/test_wsst_host.py); here the float32
uncertainty of a coefficient's frequency comes from the UNWRAPPED phase the reference keeps in
float32 (two ulps of the accumulated angle)."""
import os

import numpy as np
import pytest

from oracle import ref, restate
from tests import cases


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "synsq.npz"))


def explained(got, want, c, fre, W, what):
    name = cases.WSST_SCALE_NAME[c["scale_type"]]
    ph, ang = restate.synsq_phase(W)
    allow, amb = restate.synsq_allowance(W, ph, ang, fre, c["samplate"], name, 0.001)
    st = cases.cwt_stride(c)
    allow, Ws, ambs = allow[:, ::st], np.asarray(W)[:, ::st], amb[:, ::st]
    if got.shape != want.shape:
        got = got[:, ::st]
    scale = np.abs(want).max()
    d = np.abs(np.asarray(got, np.complex128) - want)
    bad = d > allow + 1e-5 * scale
    assert not bad.any(), f"{what}: {int(bad.sum())} cells differ beyond what boundary coefficients explain"
    from tests.conftest import parity_log
    parity_log(f"synsq explained-difference: {what}", float((d > 1e-5 * scale).mean()), 0.05,
               "fraction of cells beyond 1e-5 (all explained by boundary coefficients)",
               {"undetermined_mass": float(np.abs(Ws[ambs]).sum() / np.abs(Ws).sum()), "undetermined_mass_bar": 0.04,
                "worst_unexplained_excess": float(np.maximum(d - allow, 0).max() / scale)})
    assert np.abs(Ws[ambs]).sum() < 0.04 * np.abs(Ws).sum(), f"{what}: criterion too loose"  # measured <= 2.3e-2
    return int((d > 1e-5 * scale).sum())


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")
def test_compiled_reference_reproduces_golden(gold, tmp_path):
    from tests.golden import make_golden
    here = make_golden.HERE
    make_golden.HERE = str(tmp_path)
    try:
        make_golden.make_synsq()
    finally:
        make_golden.HERE = here
    fresh = np.load(os.path.join(str(tmp_path), "synsq.npz"))
    for k in gold.files:
        assert np.array_equal(fresh[k], gold[k]), k


@pytest.mark.parametrize("name", list(cases.SYNSQ_CASES))
def test_restatement_explains_golden(name, gold):
    c = cases.SYNSQ_CASES[name]
    fre, W = cases.synsq_input(c)
    ph = restate.synsq_frequency(W)
    v = restate.synsq_coordinates(ph, fre, c["samplate"], cases.WSST_SCALE_NAME[c["scale_type"]])
    out = restate.wsst_squeeze(W, v, 0.001)
    n_diff = explained(out, gold[f"{name}/s"], c, fre, W, name)
    assert n_diff < 0.02 * gold[f"{name}/s"].size
