#!/usr/bin/env python3
"""Golden BFT cases through the fused STFT -> filter-bank kernels as emulated device code (tests/emu): n_fft 2048 real
and complex (k_stft_mel_v2), n_fft 1024 (k_stft_band_1k), n_fft 4096 (k_stft_band_4k2), with the
product's own dispatcher and launchers.  Usage: emulated_bft_cases.py <case> ...; a case the dispatcher sends to the
size-generic kernels (not emulated) is reported and fails.  AFX_LIB = the library tests/test_emulated_kernels.py builds."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import cases  # noqa: E402

lib = C.CDLL(os.environ["AFX_LIB"])
vp, fp = C.c_void_p, C.POINTER(C.c_float)
lib.bftObj_calTimeLength.restype = C.c_int
lib.afx_emulated_launches.restype = C.c_int
lib.afx_emulated_launches.argtypes = [C.c_char_p]
KERNELS = (b"k_stft_mel_v2", b"k_stft_band_1k", b"k_stft_band_4k2")


def rel(got, want):
    got, want = np.asarray(got, np.complex128), np.asarray(want, np.complex128)
    return max(np.abs(got - want).max() / np.abs(want).max(), np.linalg.norm((got - want).ravel()) / np.linalg.norm(want.ravel()))


def opt(c, key, ctype):
    return C.byref(ctype(c[key])) if key in c else None


def main():
    gold = np.load(os.path.join(ROOT, "tests", "golden", "bft.npz"))
    for name in sys.argv[1:]:
        c = cases.BFT_CASES[name]
        x = cases.make_input(c["x"], c["samplate"]).astype(np.float32)
        n = len(x)
        h = vp()
        st = lib.bftObj_new(C.byref(h), c["num"], c["radix2_exp"], opt(c, "samplate", C.c_int), opt(c, "low_fre", C.c_float),
                            opt(c, "high_fre", C.c_float), opt(c, "bin_per_octave", C.c_int), opt(c, "window_type", C.c_int),
                            opt(c, "slide_length", C.c_int), opt(c, "scale_type", C.c_int), opt(c, "style_type", C.c_int),
                            opt(c, "normal_type", C.c_int), opt(c, "data_type", C.c_int), None, opt(c, "is_temporal", C.c_int))
        assert st == 0, (name, st)
        lib.bftObj_setResultType(h, c["result_type"])
        T = lib.bftObj_calTimeLength(h, n)
        want = gold[f"{name}/re"]
        cplx = c["result_type"] == 0
        if cplx:
            want = want + 1j * gold[f"{name}/im"]
        assert want.shape == (T, c["num"]), (name, want.shape, T)
        before = [lib.afx_emulated_launches(k) for k in KERNELS]
        re, im = np.zeros((T, c["num"]), np.float32), np.zeros((T, c["num"]), np.float32)
        stream = (C.c_char * 8)()
        st = lib.bftObj_bftBatchDevice(h, x.ctypes.data_as(fp), 1, n, C.c_longlong(n), re.ctypes.data_as(fp),
                                       im.ctypes.data_as(fp) if cplx else None, C.cast(stream, vp))
        assert st == 0, (name, st)
        ran = [k.decode() for k, b in zip(KERNELS, before) if lib.afx_emulated_launches(k) > b]
        got = re + 1j * im if cplx else re
        e = rel(got, want)
        print(f"{name}: {e:.2e} via {ran or 'NO EMULATED KERNEL'}", flush=True)
        assert ran and np.all(np.isfinite(got)) and e <= 1e-5, name
        lib.bftObj_free(h)
    print("OK")


if __name__ == "__main__":
    main()
