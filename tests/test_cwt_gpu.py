"""GPU parity of the continuous wavelet transform object against the reference's
golden vectors (all eight wavelet families, padded and unpadded, 2^3 .. 2^16 samples)."""
import os

import numpy as np
import pytest

import audioflux_amd as af
from tests import cases
from tests.conftest import assert_parity

pytestmark = pytest.mark.gpu
TOL = 1e-5


def make(c):
    return af.CWT(num=c["num"], radix2_exp=c["radix2_exp"], samplate=c["samplate"],
                  low_fre=c.get("low_fre"), high_fre=c.get("high_fre"),
                  bin_per_octave=c.get("bin_per_octave", 12),
                  wavelet_type=af.WaveletContinueType(c["wavelet_type"]),
                  scale_type=af.SpectralFilterBankScaleType(c["scale_type"]),
                  gamma=c.get("gamma"), beta=c.get("beta"), is_padding=bool(c["is_padding"]))


@pytest.mark.parametrize("name", list(cases.CWT_CASES))
def test_cwt_matches_golden(name, golden_dir):
    gold = np.load(os.path.join(golden_dir, "cwt.npz"))
    c = cases.CWT_CASES[name]
    o = make(c)
    assert np.array_equal(o.get_fre_band_arr(), gold[f"{name}/fre"])
    assert np.array_equal(o.get_bin_band_arr(), gold[f"{name}/bin"])
    x = cases.make_input((c["x"][0], c["x"][1], 1 << c["radix2_exp"]), c["samplate"])
    st = cases.cwt_stride(c)
    w = o.cwt(x)[::-1, ::st]  # wrapper returns ascending frequency; the C layout is descending
    assert_parity(w, gold[f"{name}/re"] + 1j * gold[f"{name}/im"], TOL, name)
    if f"{name}/det_re" in gold.files:
        o.enable_det(True)
        d = o.cwt_det(x)[::-1, ::st]
        assert_parity(d, gold[f"{name}/det_re"] + 1j * gold[f"{name}/det_im"], TOL, name + "/det")


def test_cwt_linearity_and_reuse():
    o = af.CWT(num=30, radix2_exp=11, samplate=16000, wavelet_type=af.WaveletContinueType.MORLET)
    a, b = cases.noise(95, 2048), cases.noise(96, 2048)
    wa, wb, ws = o.cwt(a), o.cwt(b), o.cwt((a - 3 * b).astype(np.float32))
    assert_parity(ws, wa - 3 * wb, 2e-6, "linearity")
    # short input is zero-padded by the wrapper, long input truncated (utils/util.py:98-111)
    assert o.cwt(a[:1000]).shape == (30, 2048) and o.cwt(np.concatenate([a, b])).shape == (30, 2048)
