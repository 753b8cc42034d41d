#!/usr/bin/env python3
"""k_stft_band_512 (n_fft 512: the 256-point transform as 4 x 4 x 4 x 4 in four registers per lane, three conflict-free
transposes, natural-order last stage) as emulated device code through the product's own BFT object and dispatcher, against the
numpy restatement of the reference (oracle/restate.py, float64): every tap variant (mel-128 / -80 / -40 at 16 kHz, mel-26 at
44.1 kHz), hop 128 (register re-use) and 160 / 101 (whole frames, odd starts), power / magnitude / norm exponent, complex
results, two clips an odd number of floats apart.  AFX_LIB = the library tests/test_emulated_kernels.py builds."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import restate  # noqa: E402

lib = C.CDLL(os.environ["AFX_LIB"])
vp, fp = C.c_void_p, C.POINTER(C.c_float)
lib.bftObj_calTimeLength.restype = C.c_int
lib.bftObj_fusedPlanKind.restype = C.c_int
lib.afx_emulated_launches.restype = C.c_int
lib.afx_emulated_launches.argtypes = [C.c_char_p]


def rel(got, want):
    got, want = np.asarray(got, np.complex128), np.asarray(want, np.complex128)
    return max(np.abs(got - want).max() / np.abs(want).max(), np.linalg.norm((got - want).ravel()) / np.linalg.norm(want.ravel()))


rng = np.random.default_rng(11)
launched = 0
for num, sr, hop, dt, rt, norm in ((128, 16000, 128, 0, 1, 1.0), (80, 16000, 160, 1, 1, 1.0), (40, 16000, 128, 0, 1, 0.5),
                                   (26, 44100, 101, 1, 1, 2.0), (128, 16000, 128, 1, 0, 1.0), (40, 16000, 160, 0, 0, 1.0)):
    n, stride = 512 + 9 * hop + 3, 512 + 9 * hop + 8
    x = (0.1 * rng.standard_normal(2 * stride)).astype(np.float32)
    h = vp()
    st = lib.bftObj_new(C.byref(h), num, 9, C.byref(C.c_int(sr)), C.byref(C.c_float(0.0)), C.byref(C.c_float(sr / 2)), None,
                        C.byref(C.c_int(1)), C.byref(C.c_int(hop)), C.byref(C.c_int(2)), C.byref(C.c_int(0)), C.byref(C.c_int(0)),
                        C.byref(C.c_int(dt)), None, None)
    assert st == 0, st
    assert lib.bftObj_fusedPlanKind(h) == 301, lib.bftObj_fusedPlanKind(h)
    lib.bftObj_setResultType(h, rt)
    if norm != 1.0:
        lib.bftObj_setDataNormValue(h, C.c_float(norm))
    T = lib.bftObj_calTimeLength(h, n)
    re, im = np.zeros((2, T, num), np.float32), np.zeros((2, T, num), np.float32)
    before = lib.afx_emulated_launches(b"k_stft_band_512")
    stream = (C.c_char * 8)()
    st = lib.bftObj_bftBatchDevice(h, x.ctypes.data_as(fp), 2, n, C.c_longlong(stride), re.ctypes.data_as(fp),
                                   im.ctypes.data_as(fp) if rt == 0 else None, C.cast(stream, vp))
    assert st == 0, st
    launched += lib.afx_emulated_launches(b"k_stft_band_512") - before
    bank, _, _ = restate.mel_bank(num, 512, sr, 0.0, sr / 2)
    for b in range(2):
        want = restate.bft(x[b * stride:b * stride + n], bank, 512, hop, data_type="power" if dt == 0 else "mag", result_type=rt,
                           norm_value=norm)
        got = re[b] + 1j * im[b] if rt == 0 else re[b]
        e = rel(got, want)
        print(f"mel-{num} @ {sr} hop {hop} dt {dt} rt {rt} norm {norm} clip {b}: {e:.2e}", flush=True)
        assert np.all(np.isfinite(got)) and e <= 1e-5, e
    lib.bftObj_free(h)
assert launched == 6, launched
print(f"emulated k_stft_band_512 {launched}")
print("OK")
