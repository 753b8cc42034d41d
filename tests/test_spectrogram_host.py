"""CPU-only checks of the spectrogram object: parameter resolution of spectrogramObj_new (through a
device-free hook) against the compiled reference's getters, the STFT-chroma bank bit for bit,
the fixtures reproduced by the compiled reference, and a numpy restatement of the scales that are
new relative to the BFT path (linear slice + phase, STFT-chroma, log-chroma fold + normalisation)."""
import ctypes as C
import os

import numpy as np
import pytest

import audioflux_amd as af
from oracle import ref, restate
from tests import cases
from tests.conftest import assert_parity

fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int)


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "spectrogram.npz"))


def _pi(v):
    return None if v is None else C.pointer(C.c_int(int(v)))


def _pf(v):
    return None if v is None else C.pointer(C.c_float(float(v)))


def plan_of(c):
    L = af.get_lib()
    fn = L.afx_test_spectrogram_plan
    fn.restype = C.c_int
    fn.argtypes = [C.c_int, ip, fp, fp, ip, ip, ip, ip, ip, fp]
    plan = np.zeros(10, np.float32)
    st = fn(c.get("num", 0), _pi(c.get("samplate")), _pf(c.get("low_fre")), _pf(c.get("high_fre")),
            _pi(c.get("bin_per_octave")), _pi(c.get("radix2_exp")), _pi(c.get("window_type")),
            _pi(c.get("slide_length")), _pi(c.get("scale_type")), plan.ctypes.data_as(fp))
    return st, plan


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")
def test_compiled_reference_reproduces_golden(gold, tmp_path):
    from tests.golden import make_golden
    here = make_golden.HERE
    make_golden.HERE = str(tmp_path)
    try:
        make_golden.make_spectrogram()
    finally:
        make_golden.HERE = here
    fresh = np.load(os.path.join(str(tmp_path), "spectrogram.npz"))
    assert sorted(fresh.files) == sorted(gold.files)
    for k in gold.files:
        assert np.array_equal(fresh[k], gold[k]), k


@pytest.mark.parametrize("name", list(cases.SPEC_CASES))
def test_parameter_resolution_matches_golden(name, gold):
    c = cases.SPEC_CASES[name]
    st, plan = plan_of(c)
    assert st == 0
    assert int(plan[0]) == int(gold[f"{name}/num"][0])
    n = 1 << c["radix2_exp"]
    assert int(plan[9]) == c.get("slide_length", n // 4)
    if c["scale_type"] == cases.SCALE["linear"]:
        # band arrays of the linear scale: slice of linspace(0, sr/2, F) from lowIndex
        assert np.array_equal(gold[f"{name}/bin"], np.arange(int(plan[3]), int(plan[3]) + int(plan[0])))
        assert int(plan[4]) - int(plan[3]) + 1 == int(plan[0])


def test_status_codes_without_device():
    assert plan_of(dict(num=128, radix2_exp=31, scale_type=2))[0] == -100
    assert plan_of(dict(num=1, radix2_exp=10, scale_type=2))[0] == -1           # num < 2
    assert plan_of(dict(num=4000, radix2_exp=10, scale_type=2))[0] == -1        # num > N/2+1
    # octave: 10 octaves from C1 do not fit under 8 kHz Nyquist (spectrogram_algorithm.c:484-487)
    assert plan_of(dict(num=120, samplate=16000, low_fre=32.703, radix2_exp=12, bin_per_octave=12, scale_type=5))[0] == -1
    st, plan = plan_of(dict(num=7, samplate=32000, radix2_exp=11, scale_type=8))   # chroma: num -> 12
    assert st == 0 and plan[0] == 12 and plan[6] == 1025
    st, plan = plan_of(dict(num=5, samplate=32000, radix2_exp=11, bin_per_octave=24, scale_type=9))
    assert st == 0 and plan[0] == 12                                              # 24 % 5 != 0 -> 12
    st, plan = plan_of(dict(num=12, samplate=32000, radix2_exp=11, bin_per_octave=7, scale_type=9))
    assert plan[5] == 12                                                          # bpo not a multiple of 12
    st, plan = plan_of(dict(num=0, samplate=32000, radix2_exp=11, low_fre=9000.0, high_fre=100.0, scale_type=0))
    assert st == 0 and plan[1] == 0 and plan[2] == 16000 and plan[0] == 1025      # high < low -> full range
    lib = af.get_lib()
    lib.spectrogramObj_free.argtypes = [C.c_void_p]
    lib.spectrogramObj_free(None)   # NULL-safe
    lib.stftObj_free.argtypes = [C.c_void_p]
    lib.stftObj_free(None)


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("num,n_fft,sr", [(12, 4096, 32000), (12, 2048, 16000), (24, 1024, 44100), (36, 512, 8000)])
def test_stft_chroma_bank_bit_identical(num, n_fft, sr):
    L, R = af.get_lib(), ref.lib()
    L.afx_chroma_stft_bank.restype = fp
    L.afx_chroma_stft_bank.argtypes = [C.c_int, C.c_int, C.c_int]
    R.chroma_stftFilterBank.restype = None
    R.chroma_stftFilterBank.argtypes = [C.c_int, C.c_int, C.c_int, fp, fp, fp]
    f = n_fft // 2 + 1
    want = np.zeros((num, f), np.float32)
    R.chroma_stftFilterBank(num, n_fft, sr, None, None, want.ctypes.data_as(fp))
    got = np.ctypeslib.as_array(L.afx_chroma_stft_bank(num, n_fft, sr), (num, f))
    assert np.array_equal(got, want)


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("sr,n_fft,lo,hi", [(16000, 512, None, None), (16000, 1024, 1000.0, 5000.0), (44100, 2048, 50.0, 20000.0)])
def test_linear_band_arrays_match_reference(sr, n_fft, lo, hi):
    r = n_fft.bit_length() - 1
    o = ref.RefSpectrogram(0, samplate=sr, low_fre=lo, high_fre=hi, radix2_exp=r, scale_type=0)
    st, plan = plan_of(dict(num=0, samplate=sr, low_fre=lo, high_fre=hi, radix2_exp=r, scale_type=0))
    assert st == 0 and o.status == 0 and int(plan[0]) == o.num
    assert np.array_equal(o.bin_band(), np.arange(int(plan[3]), int(plan[4]) + 1))
    grid = np.linspace(0, sr / 2, n_fft // 2 + 1).astype(np.float32)
    assert np.abs(o.fre_band() - grid[int(plan[3]):int(plan[4]) + 1]).max() <= 1e-3


def _restate(c, x):
    """numpy restatement of the scales the BFT tests do not cover (spectrogram_algorithm.c:1037-1340)"""
    n, sr = 1 << c["radix2_exp"], c["samplate"]
    hop = c.get("slide_length", n // 4)
    S = restate.stft(x, n, hop, c["window_type"])
    P = np.abs(S) ** 2
    norm = c.get("norm", 1.0)
    st, plan = plan_of(c)
    lo, hi = int(plan[3]), int(plan[4])
    mag = c["data_type"] == 1
    V = np.sqrt(P) if mag else (P ** norm if norm != 1 else P)
    if c["scale_type"] == cases.SCALE["linear"]:
        out = V[:, lo:hi + 1]
        return out ** norm if mag and norm != 1 else out
    # STFT-chroma: bank bit-identical to the reference (test above); zero outside [lo, hi]
    L = af.get_lib()
    L.afx_chroma_stft_bank.restype = fp
    L.afx_chroma_stft_bank.argtypes = [C.c_int, C.c_int, C.c_int]
    num = int(plan[0])
    bank = np.ctypeslib.as_array(L.afx_chroma_stft_bank(num, n, sr), (num, n // 2 + 1)).astype(np.float64)
    V = V.copy()
    V[:, :lo] = 0
    V[:, hi + 1:] = 0
    out = V @ bank.T
    if mag and norm != 1:
        out = out ** norm
    t = c.get("chroma_norm", 1)
    a = np.abs(out)
    d = {0: None, 1: a.max(1), 2: a.min(1), 3: np.sqrt((a * a).sum(1)), 4: a.sum(1)}[t]
    if d is not None:
        d = np.where(d == 0, 1.0, d)
        out = out / d[:, None]
    return out


@pytest.mark.parametrize("name", ["linear_default", "linear_slice_mag_norm", "chroma12_power",
                                  "chroma24_mag_range_p2", "chroma_none_norm"])
def test_restatement_matches_golden(name, gold):
    c = cases.SPEC_CASES[name]
    x = cases.make_input(c["x"], c["samplate"])
    assert_parity(_restate(c, x), gold[f"{name}/spec"], 1e-5, name)
    if c.get("phase"):
        n = 1 << c["radix2_exp"]
        st, plan = plan_of(c)
        S = restate.stft(x, n, c.get("slide_length", n // 4), c["window_type"])[:, int(plan[3]):int(plan[4]) + 1]
        want = gold[f"{name}/phase"]
        ph = np.arctan2(S.imag, np.maximum(S.real, 1e-16))
        # the phase of a bin is conditioned by 1/|S|: compare where |S| is not lost in the FFT rounding
        ok = np.abs(S) > 1e-3 * np.abs(S).max()
        assert np.abs(ph[ok] - want[ok]).max() < 2e-3 and ok.mean() > 0.2
