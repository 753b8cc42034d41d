#!/usr/bin/env python3
"""k_stft_band_512 (n_fft 512: the 256-point transform as 4 x 4 x 4 x 4 in four registers per lane, three conflict-free
transposes, natural-order last stage) as emulated device code through the product's own BFT object and dispatcher, against the
numpy restatement of the reference (oracle/restate.py, float64): every tap variant (mel-128 / -80 / -40 at 16 kHz, mel-26 at
44.1 kHz), hop 128 (register re-use) and 160 / 101 (whole frames, odd starts), power / magnitude / norm exponent, complex
results, two clips an odd number of floats apart; then the row-SEGMENT plans of the n_fft 512 and n_fft 1024 kernels (mel-13 /
-20 / -26: rows longer than the tap variants, afx_bandplan_build_split).  AFX_LIB = the library tests/test_emulated_kernels.py builds."""
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
launched = {b"k_stft_band_512": 0, b"k_stft_band_1k": 0}
# (radix2Exp, bands, sample rate, hop, data type, result type, norm exponent, expected plan kind): 301 / 302 = n_fft 512 whole rows /
# row segments, 102 = n_fft 1024 row segments (mel-13 / -20: rows of 85 - 170 bins against variants of at most 72 taps)
CASES = ((9, 128, 16000, 128, 0, 1, 1.0, 301), (9, 80, 16000, 160, 1, 1, 1.0, 301), (9, 40, 16000, 128, 0, 1, 0.5, 301),
         (9, 26, 44100, 101, 1, 1, 2.0, 301), (9, 128, 16000, 128, 1, 0, 1.0, 301), (9, 40, 16000, 160, 0, 0, 1.0, 301),
         (9, 13, 16000, 128, 0, 1, 1.0, 302), (9, 13, 16000, 100, 1, 1, 2.0, 302), (9, 20, 44100, 128, 0, 0, 1.0, 302),
         (10, 13, 16000, 256, 0, 1, 0.5, 102), (10, 20, 16000, 200, 1, 1, 2.0, 102), (10, 26, 22050, 256, 1, 0, 1.0, 102))
for r, num, sr, hop, dt, rt, norm, kind in CASES:
    N = 1 << r
    kern = b"k_stft_band_512" if r == 9 else b"k_stft_band_1k"
    n, stride = N + 9 * hop + 3, N + 9 * hop + 8
    x = (0.1 * rng.standard_normal(2 * stride)).astype(np.float32)
    h = vp()
    st = lib.bftObj_new(C.byref(h), num, r, C.byref(C.c_int(sr)), C.byref(C.c_float(0.0)), C.byref(C.c_float(sr / 2)), None,
                        C.byref(C.c_int(1)), C.byref(C.c_int(hop)), C.byref(C.c_int(2)), C.byref(C.c_int(0)), C.byref(C.c_int(0)),
                        C.byref(C.c_int(dt)), None, None)
    assert st == 0, st
    assert lib.bftObj_fusedPlanKind(h) == kind, (lib.bftObj_fusedPlanKind(h), kind)
    lib.bftObj_setResultType(h, rt)
    if norm != 1.0:
        lib.bftObj_setDataNormValue(h, C.c_float(norm))
    T = lib.bftObj_calTimeLength(h, n)
    re, im = np.zeros((2, T, num), np.float32), np.zeros((2, T, num), np.float32)
    before = lib.afx_emulated_launches(kern)
    stream = (C.c_char * 8)()
    st = lib.bftObj_bftBatchDevice(h, x.ctypes.data_as(fp), 2, n, C.c_longlong(stride), re.ctypes.data_as(fp),
                                   im.ctypes.data_as(fp) if rt == 0 else None, C.cast(stream, vp))
    assert st == 0, st
    launched[kern] += lib.afx_emulated_launches(kern) - before
    bank, _, _ = restate.mel_bank(num, N, sr, 0.0, sr / 2)
    for b in range(2):
        want = restate.bft(x[b * stride:b * stride + n], bank, N, hop, data_type="power" if dt == 0 else "mag", result_type=rt,
                           norm_value=norm)
        got = re[b] + 1j * im[b] if rt == 0 else re[b]
        e = rel(got, want)
        print(f"n_fft {N} mel-{num} @ {sr} hop {hop} dt {dt} rt {rt} norm {norm} kind {kind} clip {b}: {e:.2e}", flush=True)
        # complex results sum the spectrum itself over a band: S alternates in sign from bin to bin and a smooth band of 100+ weights
        # cancels to far below max|S| -- what is left carries the float32 error of S; the reference's own distance from float64
        # reaches 3e-5 of the output peak on such banks (tests/test_bft_gpu.py has the same bar for the segment plans at 2048)
        assert np.all(np.isfinite(got)) and e <= (3e-5 if rt == 0 and kind in (102, 302) else 1e-5), e
    lib.bftObj_free(h)
assert launched[b"k_stft_band_512"] == 9 and launched[b"k_stft_band_1k"] == 3, launched
print(f"emulated k_stft_band_512 {launched[b'k_stft_band_512']}, k_stft_band_1k {launched[b'k_stft_band_1k']}")
print("OK")
