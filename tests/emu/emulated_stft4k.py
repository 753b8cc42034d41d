#!/usr/bin/env python3
"""afxk_stft4k / afxk_stft1k / afxk_stft512 (k_stft_band_4k2 / _1k / _512 <STFT>: the bank kernels' transforms storing their
spectrum through scalar-register bases) as emulated device code against numpy's float64 FFT: the full layout of stftObj_stft
with its conjugate mirrors (hop N / 4: register re-use; two clips an odd number of floats apart), a range-checked complex
slice on a pitched output (odd hop, whole-frame fetches) and the mapped stores (power / magnitude of a slice).  AFX_LIB = the
library tests/test_emulated_kernels.py builds."""
import ctypes as C
import os
import sys

import numpy as np

from emulated_stft4k_args import AfxStftArgs, fp

lib = C.CDLL(os.environ["AFX_LIB"])

for fn in ("afxk_stft4k", "afxk_stft1k", "afxk_stft512"):
    getattr(lib, fn).restype = C.c_int
    getattr(lib, fn).argtypes = [C.POINTER(AfxStftArgs), C.c_void_p]
lib.afx_emulated_launches.restype = C.c_int
lib.afx_emulated_launches.argtypes = [C.c_char_p]


def ptr(a):
    return a.ctypes.data_as(fp)


def run(fn, r, x, stride, batch, n, hop, win, mode, lo, count, pitch):
    N = 1 << r
    t = (n - N) // hop + 1
    re = np.full((batch * t, pitch), np.nan, np.float32)
    im = np.full((batch * t, pitch), np.nan, np.float32)
    a = AfxStftArgs()
    a.x, a.clipStride, a.batch, a.dataLength, a.timeLength, a.radix2Exp, a.hop = ptr(x), stride, batch, n, t, r, hop
    a.window, a.mode, a.normValue, a.binLo, a.binCount, a.outPitch = ptr(win), mode, 1.0, lo, count, pitch
    a.outRe, a.outIm = ptr(re), ptr(im)
    st = fn(C.byref(a), None)
    assert st == 0, st
    return t, re, im


def want(N, x, stride, batch, n, hop, win):
    t = (n - N) // hop + 1
    out = np.empty((batch * t, N), np.complex128)
    for b in range(batch):
        for i in range(t):
            out[b * t + i] = np.fft.fft(x[b * stride + i * hop: b * stride + i * hop + N].astype(np.float64) * win.astype(np.float64))
    return out


def rel(got, ref):
    return float(np.abs(got - ref).max() / np.abs(ref).max())


rng = np.random.default_rng(7)
for r, name, kern in ((12, "afxk_stft4k", b"k_stft_band_4k2"), (10, "afxk_stft1k", b"k_stft_band_1k"), (9, "afxk_stft512", b"k_stft_band_512")):
    fn, N, H = getattr(lib, name), 1 << r, (1 << r) // 2
    win = np.ascontiguousarray((0.5 - 0.5 * np.cos(2 * np.pi * np.arange(N) / N)).astype(np.float32))
    before = lib.afx_emulated_launches(kern)
    hop4, hopx = N // 4, (N * 700) // 4096 + 1

    # 1. stftObj_stft's layout: all N bins, two clips an odd number of floats apart, hop N / 4 (register re-use)
    n, stride = N + 5 * hop4, 2 * N + 5 * hop4 + 1
    x = (0.1 * rng.standard_normal(2 * stride)).astype(np.float32)
    t, re, im = run(fn, r, x, stride, 2, n, hop4, win, 0, 0, N, N)
    e = rel(re + 1j * im, want(N, x, stride, 2, n, hop4, win))
    assert e < 2e-6 and not np.isnan(re).any() and not np.isnan(im).any(), e
    assert np.array_equal(re[:, 1:H], re[:, :H:-1]) and np.array_equal(im[:, 1:H], -im[:, :H:-1])
    print(f"n_fft {N}: full spectrum, 2 x {t} frames, hop {hop4}: {e:.2e} of the peak; mirrors are exact conjugates")

    # 2. a complex slice on a pitched output, an odd hop (frames fetched whole, starts on any sample)
    lo, count = N // 40, (3 * N) // 4 - 3
    pitch = count + 8
    x = (0.1 * rng.standard_normal(N + 3 * hopx + 11)).astype(np.float32)
    t, re, im = run(fn, r, x, 0, 1, x.size, hopx, win, 0, lo, count, pitch)
    e = rel(re[:, :count] + 1j * im[:, :count], want(N, x, 0, 1, x.size, hopx, win)[:, lo:lo + count])
    assert e < 2e-6 and np.isnan(re[:, count:]).all() and np.isnan(im[:, count:]).all(), e   # nothing outside the slice is touched
    print(f"n_fft {N}: bins {lo} .. {lo + count - 1} on rows of {pitch}, hop {hopx}, {t} frames: {e:.2e}")

    # 3. mapped stores: power (1) and magnitude (2) of bins 3 .. N / 2
    for mode, f in ((1, lambda z: np.abs(z) ** 2), (2, np.abs)):
        t, re, im = run(fn, r, x, 0, 1, x.size, hopx, win, mode, 3, H - 2, H - 2)
        e = rel(re, f(want(N, x, 0, 1, x.size, hopx, win)[:, 3:H + 1]))
        assert e < 4e-6 and np.isnan(im).all(), e   # the imaginary plane belongs to the complex modes
        print(f"n_fft {N}: mode {mode}, bins 3 .. {H}: {e:.2e}")

    # 4. not its case: a frame that leaves the clip -> the caller's size-generic kernel
    a = AfxStftArgs()
    a.x, a.clipStride, a.batch, a.dataLength, a.timeLength, a.radix2Exp, a.hop = ptr(x), 0, 1, N, 2, r, N // 8
    a.window, a.binCount, a.outRe, a.outIm = ptr(win), N, ptr(re), ptr(im)
    assert fn(C.byref(a), None) == -4   # AFX_ERR_UNSUPPORTED
    n_l = lib.afx_emulated_launches(kern) - before
    assert n_l == 4, n_l
    print(f"emulated {kern.decode()} {n_l}")
print("OK")
