"""GPU parity of the cepstrogram object against the reference's golden vectors."""
import os

import numpy as np
import pytest

import audioflux_amd as af
from tests import cases
from tests.conftest import assert_parity

pytestmark = pytest.mark.gpu
# cepstrum / envelope: 1e-5.  details: the reference's own float32 result is 0.7-1.1e-5
# (peak-relative) away from a float64 evaluation of the same formulas on these inputs
# (tests/test_oracle.py), so two float32 implementations can differ by ~sqrt(2) of that.
TOL = {"cep": 1e-5, "env": 1e-5, "det": 3e-5}


@pytest.mark.parametrize("name", list(cases.CEPS_CASES))
def test_cepstrogram_matches_golden(name, golden_dir):
    gold = np.load(os.path.join(golden_dir, "cepstrogram.npz"))
    c = cases.CEPS_CASES[name]
    o = af.Cepstrogram(radix2_exp=c["radix2_exp"], window_type=af.WindowType(c["window_type"]),
                       slide_length=c["slide_length"])
    x = cases.make_input(c["x"], 16000)
    outs = o.cepstrogram(x, cep_num=c["cep_num"])
    for k, got in zip(("cep", "env", "det"), outs):
        assert_parity(got.T, gold[f"{name}/{k}"], TOL[k], f"{name}/{k}")


def test_envelope_plus_details_reconstruct_log_spectrum():
    """size-independent property: the two lifters partition the cepstrum (index N-cepNum is in
    both), so envelope + details = ln|S|^2 + the doubled term's cosine"""
    n, hop, q = 1024, 256, 12
    x = cases.noise(77, 8000)
    o = af.Cepstrogram(radix2_exp=10, window_type=af.WindowType.HANN, slide_length=hop)
    cep, env, det = [a.T for a in o.cepstrogram(x, cep_num=q)]
    from oracle import restate
    fr = restate.frames_of(x, n, hop) * restate.fft_window(1, n)[None, :]
    logS = np.log(np.maximum(np.abs(np.fft.rfft(fr, axis=1)) ** 2, 1e-16))
    k = np.arange(n // 2 + 1)
    # c is even, so c[N-q] = c[q] = cep[:, q]
    extra = cep[:, q:q + 1] * np.cos(2 * np.pi * k * (n - q) / n)[None, :]
    assert_parity(env + det, logS + extra, 1e-4, "lifter partition")
