#!/usr/bin/env python3
"""mel + MFCC in ONE launch at every fused size (round 6: afx_ccblock.h inside k_stft_band_512 / _1k / _4k2 and the general form of
k_stft_mel_v2) as emulated device code through afx_bftXxccBatchDevice, against the numpy restatement of the reference
(oracle/restate.py: bank, STFT, rectification, DCT-II): whole-row and row-segment plans, num 128 / 64 / 40 / 20, log and cube root,
hop N / 4 (register re-use) and an odd hop, frame counts that leave partial 16-row blocks.  AFX_LIB = the library
tests/test_emulated_kernels.py builds."""
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
lib.afx_bftXxccOneLaunchCount.restype = C.c_longlong


def rel(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return max(np.abs(got - want).max() / np.abs(want).max(), np.linalg.norm((got - want).ravel()) / np.linalg.norm(want.ravel()))


only = [int(v) for v in sys.argv[1:]]  # radix2 exponents (default: all)
LONG = os.environ.get("AFX_EMU_CUS") == "1"  # one "CU": 32 waves -- runs of 38 / 41 frames per wave (whole 16-row blocks behind each other)
rng = np.random.default_rng(66)
# (radix2Exp, bands, hop, frames, ccNum, rectify)
CASES = ((9, 128, 128, 37, 13, 0), (9, 40, 101, 20, 16, 1), (9, 20, 128, 18, 5, 0),
         (10, 128, 256, 35, 13, 0), (10, 64, 256, 18, 16, 1), (10, 20, 200, 17, 13, 0),
         (11, 64, 512, 19, 13, 1), (11, 40, 512, 18, 7, 0), (11, 128, 300, 17, 13, 1),
         (12, 128, 1024, 18, 13, 0), (12, 40, 1024, 17, 16, 1))
if LONG:
    CASES = ((9, 128, 128, 600, 13, 0), (10, 128, 256, 650, 13, 0), (10, 40, 256, 600, 13, 1), (9, 64, 128, 600, 16, 0))
ran = 0
for r, num, hop, frames, cc_num, rect in CASES:
    if only and r not in only:
        continue
    N = 1 << r
    n, stride = N + (frames - 1) * hop + 3, N + (frames - 1) * hop + 8
    x = (0.1 * rng.standard_normal(2 * stride)).astype(np.float32)
    x[: n // 3] = 0.0  # silent frames: the floor of the log, 0^(1/3)
    h, xx = vp(), vp()
    st = lib.bftObj_new(C.byref(h), num, r, C.byref(C.c_int(16000)), C.byref(C.c_float(0.0)), C.byref(C.c_float(8000.0)), None,
                        C.byref(C.c_int(1)), C.byref(C.c_int(hop)), C.byref(C.c_int(2)), C.byref(C.c_int(0)), C.byref(C.c_int(0)),
                        C.byref(C.c_int(0)), None, None)
    assert st == 0, st
    assert lib.xxccObj_new(C.byref(xx), num) == 0
    lib.bftObj_setResultType(h, 1)
    kind = lib.bftObj_fusedPlanKind(h)
    assert kind, (r, num)
    T = lib.bftObj_calTimeLength(h, n)
    assert T == frames, (T, frames)
    mel, cc = np.zeros((2, T, num), np.float32), np.zeros((2, T, cc_num), np.float32)
    stream = (C.c_char * 8)()
    before = lib.afx_bftXxccOneLaunchCount()
    st = lib.afx_bftXxccBatchDevice(h, xx, x.ctypes.data_as(fp), 2, n, C.c_longlong(stride), cc_num, C.byref(C.c_int(rect)),
                                    mel.ctypes.data_as(fp), cc.ctypes.data_as(fp), C.cast(stream, vp))
    assert st == 0, st
    assert lib.afx_bftXxccOneLaunchCount() == before + 1, f"n_fft {N} mel-{num}: plan kind {kind} took two launches"
    bank, _, _ = restate.mel_bank(num, N, 16000, 0.0, 8000.0)
    for b in range(2):
        wmel = restate.bft(x[b * stride:b * stride + n], bank, N, hop)
        wcc = restate.xxcc(wmel, cc_num, "log" if rect == 0 else "cbrt")
        em, ec = rel(mel[b], wmel), rel(cc[b], wcc)
        print(f"n_fft {N} mel-{num} hop {hop} T {T} cc {cc_num} rect {rect} kind {kind} clip {b}: mel {em:.2e} mfcc {ec:.2e}", flush=True)
        assert np.all(np.isfinite(cc[b])) and em <= 1e-5 and ec <= 1e-5, (em, ec)
    lib.xxccObj_free(xx)
    lib.bftObj_free(h)
    ran += 1
print(f"one-launch mel + MFCC cases: {ran}")
print("OK")
