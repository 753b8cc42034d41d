#!/usr/bin/env python3
"""afxk_stft2k (k_stft_mel_v2 <STFT>: the headline kernel's n_fft 2048 transform storing its mapped spectrum row -- the producer
of the dense-bank route's [T, F] rows) as emulated device code against numpy's float64 FFT: the dense route's layout (all 1025
bins on rows of 1028 floats, 16-byte stores, zero pad; hop N / 4 with the register re-use, two clips an odd number of floats
apart), a slice of bins on a packed output with an odd hop (dword stores), magnitude and power-law maps, and the cases it
hands back.  AFX_LIB = the library tests/test_emulated_kernels.py builds."""
import ctypes as C
import os

import numpy as np

from emulated_stft4k_args import AfxStftArgs, fp

lib = C.CDLL(os.environ["AFX_LIB"])
lib.afxk_stft2k.restype = C.c_int
lib.afxk_stft2k.argtypes = [C.POINTER(AfxStftArgs), C.c_void_p]
N, H = 2048, 1024


def ptr(a):
    return a.ctypes.data_as(fp)


def aligned(n, align=16):
    raw = np.zeros(n + align, np.float32)
    off = (-raw.ctypes.data // 4) % (align // 4)
    return raw[off:off + n]


def run(x, stride, batch, n, hop, win, mode, norm, lo, count, pitch):
    t = (n - N) // hop + 1
    out = aligned(batch * t * pitch).reshape(batch * t, pitch)
    out[:] = np.nan
    a = AfxStftArgs()
    a.x, a.clipStride, a.batch, a.dataLength, a.timeLength, a.radix2Exp, a.hop = ptr(x), stride, batch, n, t, 11, hop
    a.window, a.mode, a.normValue, a.binLo, a.binCount, a.outPitch, a.outRe = ptr(win), mode, norm, lo, count, pitch, ptr(out)
    st = lib.afxk_stft2k(C.byref(a), None)
    assert st == 0, st
    return t, out


def power(x, stride, batch, n, hop, win):
    t = (n - N) // hop + 1
    out = np.empty((batch * t, H + 1))
    for b in range(batch):
        for i in range(t):
            s = np.fft.rfft(x[b * stride + i * hop: b * stride + i * hop + N].astype(np.float64) * win.astype(np.float64))
            out[b * t + i] = np.abs(s) ** 2
    return out


def rel(got, ref):
    return float(np.abs(got - ref).max() / np.abs(ref).max())


rng = np.random.default_rng(11)
win = aligned(N, 8)
win[:] = (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(N) / N)).astype(np.float32)

# 1. the dense route's rows: 1025 bins at a pitch of 1028, 16-byte stores with a zero pad; hop 512 (register re-use), 2 clips
n, stride = N + 40 * 512, N + 40 * 512 + 7
x = (0.1 * rng.standard_normal(2 * stride)).astype(np.float32)
t, out = run(x, stride, 2, n, 512, win, 1, 1.0, 0, 1025, 1028)
e = rel(out[:, :1025], power(x, stride, 2, n, 512, win))
assert e < 2e-6 and np.all(out[:, 1025:] == 0.0), e
print(f"power rows, 2 x {t} frames, hop 512, pitch 1028: {e:.2e} of the peak; pad words are zeros")
# (the same on a packed, unaligned output: dword stores, nothing behind bin 1024)
for hop in (256, 1024):
    t, out = run(x, stride, 2, N + 6 * hop, hop, win, 1, 1.0, 0, 1025, 1025)
    e = rel(out, power(x, stride, 2, N + 6 * hop, hop, win))
    assert e < 2e-6, e
    print(f"power rows, hop {hop}, packed: {e:.2e}")

# 2. a slice of bins, odd hop (whole-frame fetches from any sample), magnitude / power-law maps
x1 = (0.1 * rng.standard_normal(N + 5 * 301 + 3)).astype(np.float32)
ref = power(x1, 0, 1, x1.size, 301, win)
t, out = run(x1, 0, 1, x1.size, 301, win, 2, 1.0, 17, 700, 708)
e = rel(out[:, :700], np.sqrt(ref[:, 17:717]))
assert e < 2e-6 and np.isnan(out[:, 700:]).all(), e
print(f"magnitude, bins 17 .. 716 on rows of 708, hop 301, {t} frames: {e:.2e}")
t, out = run(x1, 0, 1, x1.size, 301, win, 4, 0.3, 1, 1024, 1024)
e = rel(out, ref[:, 1:1025] ** float(np.float32(0.3)))
assert e < 4e-6, e
print(f"power law 0.3, bins 1 .. 1024: {e:.2e}")

# 3. not its cases: complex results, a frame that leaves the clip, another transform size
a = AfxStftArgs()
a.x, a.clipStride, a.batch, a.dataLength, a.timeLength, a.radix2Exp, a.hop = ptr(x1), 0, 1, x1.size, 2, 11, 301
a.window, a.binCount, a.outRe, a.mode = ptr(win), 1025, ptr(out), 0
assert lib.afxk_stft2k(C.byref(a), None) == -4   # AFX_ERR_UNSUPPORTED
a.mode, a.dataLength = 1, N
assert lib.afxk_stft2k(C.byref(a), None) == -4
a.dataLength, a.radix2Exp = x1.size, 10
assert lib.afxk_stft2k(C.byref(a), None) == -4
print("OK")
