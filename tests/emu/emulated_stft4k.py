#!/usr/bin/env python3
"""afxk_stft4k (k_stft_band_4k2<STFT>: the n_fft 4096 bank kernel's transform storing its spectrum through scalar-register
bases) as emulated device code against numpy's float64 FFT: the full 4096-bin layout of stftObj_stft with its conjugate
mirrors (hop 1024: register re-use; two clips an odd number of floats apart), a range-checked complex slice on a pitched
output (hop 700, whole-frame fetches) and the mapped stores (power / magnitude of a slice).  AFX_LIB = the library
tests/test_emulated_kernels.py builds."""
import ctypes as C
import os
import sys

import numpy as np

lib = C.CDLL(os.environ["AFX_LIB"])
fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int)


class AfxStftArgs(C.Structure):  # audioflux_amd/csrc/hip/afx_device.h
    _fields_ = [("x", fp), ("clipStride", C.c_longlong), ("batch", C.c_int), ("dataLength", C.c_int), ("timeLength", C.c_int),
                ("radix2Exp", C.c_int), ("hop", C.c_int), ("window", fp), ("twiddle", fp), ("mode", C.c_int),
                ("normValue", C.c_float), ("binLo", C.c_int), ("binCount", C.c_int), ("outPitch", C.c_longlong),
                ("outRe", fp), ("outIm", fp), ("energy", fp), ("rms", fp), ("zcr", fp), ("padLeft", C.c_int),
                ("bandStart", ip), ("bandLen", ip), ("bandOff", ip), ("bandW", fp), ("bandNum", C.c_int),
                ("bandPost", C.c_int), ("bandPostArg", C.c_float), ("fullSpectrum", C.c_int), ("padMode", C.c_int),
                ("padValueL", C.c_float), ("padValueR", C.c_float)]


lib.afxk_stft4k.restype = C.c_int
lib.afxk_stft4k.argtypes = [C.POINTER(AfxStftArgs), C.c_void_p]
lib.afx_emulated_launches.restype = C.c_int
lib.afx_emulated_launches.argtypes = [C.c_char_p]
N = 4096


def ptr(a):
    return a.ctypes.data_as(fp)


def run(x, stride, batch, n, hop, win, mode, lo, count, pitch):
    t = (n - N) // hop + 1
    re = np.full((batch * t, pitch), np.nan, np.float32)
    im = np.full((batch * t, pitch), np.nan, np.float32)
    a = AfxStftArgs()
    a.x, a.clipStride, a.batch, a.dataLength, a.timeLength, a.radix2Exp, a.hop = ptr(x), stride, batch, n, t, 12, hop
    a.window, a.mode, a.normValue, a.binLo, a.binCount, a.outPitch = ptr(win), mode, 1.0, lo, count, pitch
    a.outRe, a.outIm = ptr(re), ptr(im)
    st = lib.afxk_stft4k(C.byref(a), None)
    assert st == 0, st
    return t, re, im


def want(x, stride, batch, n, hop, win):
    t = (n - N) // hop + 1
    out = np.empty((batch * t, N), np.complex128)
    for b in range(batch):
        for i in range(t):
            out[b * t + i] = np.fft.fft(x[b * stride + i * hop: b * stride + i * hop + N].astype(np.float64) * win.astype(np.float64))
    return out


def rel(got, ref):
    return float(np.abs(got - ref).max() / np.abs(ref).max())


rng = np.random.default_rng(7)
win = (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(N) / N)).astype(np.float32)
before = lib.afx_emulated_launches(b"k_stft_band_4k2")

# 1. stftObj_stft's layout: all 4096 bins, two clips 9217 floats apart, hop 1024
n, stride = 4096 + 5 * 1024, 9217
x = (0.1 * rng.standard_normal(2 * stride)).astype(np.float32)
t, re, im = run(x, stride, 2, n, 1024, win, 0, 0, N, N)
w = want(x, stride, 2, n, 1024, win)
e = rel(re + 1j * im, w)
assert e < 2e-6 and not np.isnan(re).any() and not np.isnan(im).any(), e
assert np.array_equal(re[:, 1:2048], re[:, :2048:-1]) and np.array_equal(im[:, 1:2048], -im[:, :2048:-1])
print(f"full spectrum, 2 x {t} frames, hop 1024: {e:.2e} of the peak; mirrors are exact conjugates")

# 2. a complex slice on a pitched output, hop 700 (frames fetched whole, starts on any sample)
lo, count, pitch = 100, 3000, 3008
x = (0.1 * rng.standard_normal(4096 + 3 * 700 + 11)).astype(np.float32)
t, re, im = run(x, 0, 1, x.size, 700, win, 0, lo, count, pitch)
w = want(x, 0, 1, x.size, 700, win)[:, lo:lo + count]
e = rel(re[:, :count] + 1j * im[:, :count], w)
assert e < 2e-6 and np.isnan(re[:, count:]).all() and np.isnan(im[:, count:]).all(), e   # nothing outside the slice is touched
print(f"bins {lo} .. {lo + count - 1} on rows of {pitch}, hop 700, {t} frames: {e:.2e}")

# 3. mapped stores: power (1) and magnitude (2) of bins 3 .. 2048
for mode, f in ((1, lambda z: np.abs(z) ** 2), (2, np.abs)):
    t, re, im = run(x, 0, 1, x.size, 700, win, mode, 3, 2046, 2046)
    e = rel(re, f(want(x, 0, 1, x.size, 700, win)[:, 3:2049]))
    assert e < 4e-6 and np.isnan(im).all(), e   # the imaginary plane belongs to the complex modes
    print(f"mode {mode}, bins 3 .. 2048: {e:.2e}")

# 4. not its case: a frame that leaves the clip -> the caller's size-generic kernel
a = AfxStftArgs()
a.x, a.clipStride, a.batch, a.dataLength, a.timeLength, a.radix2Exp, a.hop = ptr(x), 0, 1, 4096, 2, 12, 512
a.window, a.binCount, a.outRe, a.outIm = ptr(win), N, ptr(re), ptr(im)
assert lib.afxk_stft4k(C.byref(a), None) == -4   # AFX_ERR_UNSUPPORTED
n_l = lib.afx_emulated_launches(b"k_stft_band_4k2") - before
assert n_l == 4, n_l
print(f"emulated k_stft_band_4k2 {n_l}")
print("OK")
