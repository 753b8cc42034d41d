"""CPU-only: the frequency-domain wavelet bank the CWT object uploads is bit-identical to
the reference's cwt_filterBank for every wavelet family and frequency scale."""
import ctypes as C

import numpy as np
import pytest

import audioflux_amd as af
from oracle import ref

fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int)
pytestmark = pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")
DEFAULTS = {0: (3, 20), 1: (6, 2), 2: (5, 0.6), 3: (4, 2), 4: (2, 2), 5: (2, 2), 6: (5, 2), 7: (4, 2)}


@pytest.mark.parametrize("wavelet", range(8))
@pytest.mark.parametrize("scale", [0, 1, 2, 3, 4, 5, 6])
def test_wavelet_bank_bit_exact(wavelet, scale):
    R, N = ref.lib(), af.get_lib()
    sig = [C.c_int] * 5 + [C.c_float, C.c_float, C.c_int, C.c_float, C.c_float, C.c_int, fp, fp, ip]
    R.cwt_filterBank.restype = None
    R.cwt_filterBank.argtypes = sig
    N.afx_cwt_bank_host.restype = C.c_int
    N.afx_cwt_bank_host.argtypes = sig
    g, b = DEFAULTS[wavelet]
    for num, d, pad, sr, lo, hi in ((20, 512, 256, 32000, 100.0, 12000.0), (12, 256, 0, 16000, 65.406, 7000.0)):
        if scale in (5,):  # octave: num semitone steps above lo must stay below Nyquist
            hi = sr / 2.0
        L = d + 2 * pad
        a = np.zeros((num, L), np.float32)
        m = np.zeros((num, L), np.float32)
        fa, fm = np.zeros(num + 2, np.float32), np.zeros(num + 2, np.float32)
        ba, bm = np.zeros(num + 2, np.int32), np.zeros(num + 2, np.int32)
        R.cwt_filterBank(num, d, sr, pad, wavelet, g, b, scale, lo, hi, 12, a.ctypes.data_as(fp),
                         fa.ctypes.data_as(fp), ba.ctypes.data_as(ip))
        N.afx_cwt_bank_host(num, d, sr, pad, wavelet, g, b, scale, lo, hi, 12, m.ctypes.data_as(fp),
                            fm.ctypes.data_as(fp), bm.ctypes.data_as(ip))
        assert np.array_equal(fa[:num], fm[:num]) and np.array_equal(ba[:num], bm[:num]), (wavelet, scale)
        assert np.array_equal(a, m, equal_nan=True), (wavelet, scale, np.nanmax(np.abs(a - m)))


def _bank(num, r, sr, wavelet, lo, pad=True):
    N = af.get_lib()
    sig = [C.c_int] * 5 + [C.c_float, C.c_float, C.c_int, C.c_float, C.c_float, C.c_int, fp, fp, ip]
    N.afx_cwt_bank_host.restype = C.c_int
    N.afx_cwt_bank_host.argtypes = sig
    d = 1 << r
    p = d // 2 if pad else 0
    L = d + 2 * p
    m = np.zeros((num, L), np.float32)
    fm, bm = np.zeros(num + 2, np.float32), np.zeros(num + 2, np.int32)
    g, b = DEFAULTS[wavelet]
    N.afx_cwt_bank_host(num, d, sr, p, wavelet, g, b, 5, lo, sr / 2.0, 12, m.ctypes.data_as(fp),
                        fm.ctypes.data_as(fp), bm.ctypes.data_as(ip))
    return m


CLASS_ROWS = (2, 4, 8, 16, 20, 24, 32)


def _classify(sup, max_r):
    N = af.get_lib()
    num = len(sup) // 2
    order = np.full(num, -1, np.int32)
    n_wide = C.c_int(-1)
    n_narrow = (C.c_int * 7)(*([-1] * 7))
    N.afx_cwt_classify_host.restype = None
    N.afx_cwt_classify_host.argtypes = [ip, C.c_int, C.c_int, ip, ip, C.c_int * 7]
    N.afx_cwt_classify_host(sup.ctypes.data_as(ip), num, max_r, order.ctypes.data_as(ip), C.byref(n_wide), n_narrow)
    return order, n_wide.value, list(n_narrow)


def test_narrow_band_scale_plan_cfg4():
    """BASELINE cfg 4 (morlet, 84 scales, L = 2^17): the support table covers every non-zero of
    every wavelet, and the execution order is a permutation whose classes bound the support."""
    N = af.get_lib()
    num, r1, L = 84, 8, 1 << 17
    bank = _bank(num, 16, 44100, 1, 32.703)
    sup = np.zeros(2 * num, np.int32)
    N.afx_cwt_support_host.restype = None
    N.afx_cwt_support_host.argtypes = [fp, C.c_int, C.c_longlong, C.c_int, ip]
    N.afx_cwt_support_host(bank.ctypes.data_as(fp), num, L, r1, sup.ctypes.data_as(ip))
    t = bank.reshape(num, L >> r1, 1 << r1)  # [scale][k2][k1]
    for i in range(num):
        rows = np.nonzero(t[i].any(axis=1))[0]
        assert sup[2 * i] == rows.min() and sup[2 * i + 1] == rows.max() + 1
    width = sup[1::2] - sup[0::2]
    # rows run from the highest centre frequency down: supports shrink monotonically
    assert np.all(np.diff(width) <= 0) and width[-1] <= 2 < width[0]
    for max_r in (0, 1, 2, 4, 8, 16, 20, 24, 32, 64):
        order, n_wide, n_narrow = _classify(sup, max_r)
        assert sorted(order.tolist()) == list(range(num))
        assert n_wide + sum(n_narrow) == num
        eff = min(max_r, 32) if max_r >= 2 else 0
        assert np.all(width[order[:n_wide]] > eff)
        base = n_wide
        for cls, n in enumerate(n_narrow):  # classes R = 2, 4, 8, 16 and the two-block classes 20, 24, 32
            w = width[order[base:base + n]]
            assert np.all(w <= CLASS_ROWS[cls]) and np.all(w <= eff)
            if cls:
                assert np.all(w > CLASS_ROWS[cls - 1])
            assert np.all(np.diff(order[base:base + n]) > 0)
            base += n
    assert _classify(sup, 32)[2][4:] == [4, 3, 5]  # widths 17-20 (scales 36-39), 21-24, 25-32
    assert _classify(sup, 0)[1] == num and _classify(sup, 8)[1] < num


@pytest.mark.parametrize("wavelet", range(8))
def test_narrow_band_plan_every_family(wavelet):
    """the reference's wavelets are all one-sided (zero for omega <= 0): one low octave at
    L = 2^17 is narrow-band in every family, and the classes bound the true support"""
    N = af.get_lib()
    num, L = 12, 1 << 17
    bank = _bank(num, 16, 44100, wavelet, 32.703)
    assert not bank[:, L // 2 + 1:].any()
    sup = np.zeros(2 * num, np.int32)
    N.afx_cwt_support_host.restype = None
    N.afx_cwt_support_host.argtypes = [fp, C.c_int, C.c_longlong, C.c_int, ip]
    N.afx_cwt_support_host(bank.ctypes.data_as(fp), num, L, 8, sup.ctypes.data_as(ip))
    width = sup[1::2] - sup[0::2]
    order, n_wide, n_narrow = _classify(sup, 16)
    assert sorted(order.tolist()) == list(range(num)) and n_wide + sum(n_narrow) == num
    assert n_wide == int((width > 16).sum())
    k = np.arange(L)
    for i in range(num):  # nothing outside rows [lo, lo + R) of the class the scale was put in
        pos = order.tolist().index(i)
        if pos < n_wide:
            continue
        cls = int(np.searchsorted(np.cumsum(n_narrow), pos - n_wide, side="right"))
        lo = min(int(sup[2 * i]), 512 - CLASS_ROWS[cls])
        inside = ((k >> 8) >= lo) & ((k >> 8) < lo + CLASS_ROWS[cls])
        assert not bank[i][~inside].any()
