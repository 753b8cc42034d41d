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
