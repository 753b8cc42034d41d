#!/usr/bin/env python3
"""The headline kernel k_stft_mel_v2 (afx_melfused2.hip: STFT -> mel-128 -> log10 -> DCT-II in one launch) as emulated
device code (tests/emu): BASELINE cfg 1 -- the reference's golden mel spectrogram and MFCC -- through the C host code,
the kernel's real launcher and the kernel itself, one host thread per lane.  AFX_LIB = the library
tests/test_emulated_kernels.py builds."""
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


def rel(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return max(np.abs(got - want).max() / np.abs(want).max(), np.linalg.norm((got - want).ravel()) / np.linalg.norm(want.ravel()))


def check(what, got, want, tol=1e-5):
    e = rel(got, want)
    print(f"{what}: {e:.2e} (bar {tol:.0e})", flush=True)
    assert np.all(np.isfinite(got)) and e <= tol, what


def I(v):
    return C.byref(C.c_int(v))


def F(v):
    return C.byref(C.c_float(v))


def p(a):
    return a.ctypes.data_as(fp)


def main():
    c = cases.BFT_CASES["cfg1_mel_power"]
    gold = np.load(os.path.join(ROOT, "tests", "golden", "bft.npz"))["cfg1_mel_power/re"]
    gold_cc = np.load(os.path.join(ROOT, "tests", "golden", "xxcc.npz"))["mfcc13_log/cc"]
    x = cases.make_input(c["x"], c["samplate"]).astype(np.float32)
    n = len(x)
    h, xx = vp(), vp()
    st = lib.bftObj_new(C.byref(h), c["num"], c["radix2_exp"], I(c["samplate"]), F(c["low_fre"]), F(c["high_fre"]), None, I(c["window_type"]),
                        I(c["slide_length"]), I(c["scale_type"]), I(c["style_type"]), I(c["normal_type"]), I(c["data_type"]), None, None)
    assert st == 0, st
    assert lib.xxccObj_new(C.byref(xx), c["num"]) == 0
    lib.bftObj_setResultType(h, 1)
    T = lib.bftObj_calTimeLength(h, n)
    assert gold.shape == (T, c["num"])
    stream = (C.c_char * 8)()
    sp = C.cast(stream, vp)
    # two clips (the clip and the clip halved) with an odd row stride; mel and MFCC from ONE launch
    stride, batch = n + 3, 2
    xb = np.zeros(batch * stride, np.float32)
    xb[:n], xb[stride:stride + n] = x, 0.5 * x
    mel, cc = np.zeros((batch, T, c["num"]), np.float32), np.zeros((batch, T, 13), np.float32)
    st = lib.afx_bftXxccBatchDevice(h, xx, p(xb), batch, n, C.c_longlong(stride), 13, None, p(mel), p(cc), sp)
    assert st == 0, st
    check("cfg 1 mel (clip 0)", mel[0], gold)
    check("cfg 1 mel (clip 1, half amplitude)", mel[1], 0.25 * gold)
    check("cfg 1 MFCC-13 (clip 0)", cc[0], gold_cc)
    mel2 = np.zeros((batch, T, c["num"]), np.float32)
    assert lib.bftObj_bftBatchDevice(h, p(xb), batch, n, C.c_longlong(stride), p(mel2), None, sp) == 0
    assert np.array_equal(mel2, mel), "mel alone differs from mel beside the cepstra"
    print("launches: emulated k_stft_mel_v2 %d" % lib.afx_emulated_launches(b"k_stft_mel_v2"))
    assert lib.afx_emulated_launches(b"k_stft_mel_v2") == 2
    lib.xxccObj_free(xx)
    lib.bftObj_free(h)
    print("OK")


if __name__ == "__main__":
    main()
