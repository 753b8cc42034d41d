"""GPU parity of the constant-Q transform object (cqt, chroma, cqcc) against the
reference's golden vectors."""
import os

import numpy as np
import pytest

import audioflux_amd as af
from tests import cases
from tests.conftest import assert_parity

pytestmark = pytest.mark.gpu
TOL = 1e-5


def make(c):
    return af.CQT(num=c["num"], samplate=c["samplate"], low_fre=c["min_fre"],
                  bin_per_octave=c["bin_per_octave"], window_type=af.WindowType(c["window_type"]),
                  slide_length=c.get("slide_length"),
                  normal_type=af.SpectralFilterBankNormalType(c["normal_type"]),
                  is_scale=bool(c["is_scale"]))


@pytest.mark.parametrize("name", list(cases.CQT_CASES))
def test_cqt_chroma_cqcc_match_golden(name, golden_dir):
    gold = np.load(os.path.join(golden_dir, "cqt.npz"))
    c = cases.CQT_CASES[name]
    o = make(c)
    assert o.fft_length == int(gold[f"{name}/fft"][0])
    assert np.array_equal(o.get_fre_band_arr(), gold[f"{name}/fre"])
    x = cases.make_input(c["x"], c["samplate"])
    q = o.cqt(x)  # (num, T)
    want = gold[f"{name}/re"] + 1j * gold[f"{name}/im"]
    assert_parity(q.T, want, TOL, f"{name}/cqt")
    if c["bin_per_octave"] == 12:
        for cname, (cn, dt, nt) in cases.CQT_CHROMA.items():
            ch = o.chroma(q, cn, af.SpectralDataType(dt), af.ChromaDataNormalType(nt))
            # MIN normalisation divides by the smallest chroma value of the frame, which
            # amplifies the (1e-6 level) relative error of that one value
            tol = 4.5e-5 if cname == "six_min" else TOL  # divides by the frame MINIMUM; measured 2.2e-5
            assert_parity(ch.T, gold[f"{name}/chroma_{cname}"], tol, f"{name}/chroma_{cname}")
    cc = o.cqcc(np.abs(q), 13)
    assert_parity(cc.T, gold[f"{name}/cqcc"], TOL, f"{name}/cqcc")
    # cqhc / deconv on the reference's own magnitudes (isolates them from the cqt error)
    mag = np.abs(want).astype(np.float32).T          # (num, T)
    assert_parity(o.cqhc(mag, 20).T, gold[f"{name}/cqhc"], TOL, f"{name}/cqhc")
    tone, pitch = o.deconv(mag)
    assert_parity(tone.T, gold[f"{name}/timbre"], TOL, f"{name}/timbre")
    assert_parity(pitch.T, gold[f"{name}/pitch"], 1e-5, f"{name}/pitch")


def test_cqt_tone_lands_on_its_bin():
    """domain property: a 329.63 Hz sine peaks at CQT bin 40 = E4, chroma bin 4"""
    sr = 32000
    t = np.arange(sr) / sr
    x = np.sin(2 * np.pi * 329.63 * t).astype(np.float32)
    o = af.CQT(num=84, samplate=sr)
    q = o.cqt(x)
    assert int(np.abs(q).mean(axis=1).argmax()) == 40
    ch = o.chroma(q)
    assert int(ch.mean(axis=1).argmax()) == 4
    assert abs(float(ch.max()) - 1.0) < 1e-6  # MAX-normalised


def test_cqt_linearity():
    o = af.CQT(num=48, samplate=16000)
    a, b = cases.noise(90, 6000), cases.noise(91, 6000)
    qa, qb, qs = o.cqt(a), o.cqt(b), o.cqt((a + 2 * b).astype(np.float32))
    assert_parity(qs, qa + 2 * qb, 2e-6, "linearity")
