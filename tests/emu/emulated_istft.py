#!/usr/bin/env python3
"""The one-launch inverse STFT (afxk_istft_fused: k_istft_w256 / _wsmall<512, 1024> / _w2048 / _w4096) and k_stft_256 as emulated device
code against numpy in float64: spectra of random frames (and, for the inverse's Hermitian-part rule, spectra with a non-Hermitian
perturbation -- the real part of the complex inverse keeps the Hermitian part only, stft_algorithm.c:304-409), a clip long enough for
two runs of frames per wave so that the frames before a run are transformed again for their tails, an output buffer that is not zero
(the reference adds onto dataArr), a hop that divides N and one that does not.  AFX_LIB = the library tests/test_emulated_kernels.py
builds."""
import ctypes as C
import os

import numpy as np

from emulated_stft4k_args import AfxStftArgs, fp

lib = C.CDLL(os.environ["AFX_LIB"])


class AfxIstftArgs(C.Structure):  # audioflux_amd/csrc/hip/afx_device.h
    _fields_ = [("re", fp), ("im", fp), ("batch", C.c_int), ("timeLength", C.c_int), ("radix2Exp", C.c_int), ("hop", C.c_int),
                ("twiddle", fp), ("win1", fp), ("win2", fp), ("frames", fp), ("out", fp), ("outStride", C.c_longlong)]


lib.afxk_istft_fused.restype = C.c_int
lib.afxk_istft_fused.argtypes = [C.POINTER(AfxIstftArgs), C.c_void_p]
lib.afxk_stft256.restype = C.c_int
lib.afxk_stft256.argtypes = [C.POINTER(AfxStftArgs), C.c_void_p]


def aligned(n, align=16):
    raw = np.zeros(n + align, np.float32)
    off = (-raw.ctypes.data // 4) % (align // 4)
    return raw[off:off + n]


def ptr(a):
    return a.ctypes.data_as(fp)


def istft_case(r, hop, T, batch, rng, perturb):
    N = 1 << r
    w = (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(N) / N))
    win1, win2 = aligned(N), aligned(N)
    win1[:], win2[:] = w.astype(np.float32), (w * w).astype(np.float32)
    n = (T - 1) * hop + N
    re, im = aligned(batch * T * N), aligned(batch * T * N)
    spec = np.fft.fft(rng.standard_normal((batch, T, N)), axis=2)
    if perturb:  # a non-Hermitian part: must not reach the output
        spec = spec + 0.3 * (rng.standard_normal(spec.shape) + 1j * rng.standard_normal(spec.shape))
    re[:], im[:] = spec.real.astype(np.float32).ravel(), spec.imag.astype(np.float32).ravel()
    out = aligned(batch * (n + 5))
    init = (0.01 * rng.standard_normal(out.size)).astype(np.float32)
    out[:] = init
    a = AfxIstftArgs()
    a.re, a.im, a.batch, a.timeLength, a.radix2Exp, a.hop = ptr(re), ptr(im), batch, T, r, hop
    a.win1, a.win2, a.out, a.outStride = ptr(win1), ptr(win2), ptr(out), n + 5
    st = lib.afxk_istft_fused(C.byref(a), None)
    assert st == 0, st
    s64 = re.astype(np.float64).reshape(batch, T, N) + 1j * im.astype(np.float64).reshape(batch, T, N)
    frames = np.fft.ifft(s64, axis=2).real * win1.astype(np.float64)
    worst = 0.0
    for b in range(batch):
        acc = init[b * (n + 5): b * (n + 5) + n].astype(np.float64)
        nrm = np.zeros(n)
        for i in range(T):
            acc[i * hop: i * hop + N] += frames[b, i]
            nrm[i * hop: i * hop + N] += win2.astype(np.float64)
        want = acc / np.where(nrm < 1e-6, 1.0, nrm)
        got = out[b * (n + 5): b * (n + 5) + n].astype(np.float64)
        cond = nrm > 1e-3  # (window sums near the clamp amplify float32 rounding: tests/conftest.py::assert_istft_parity)
        worst = max(worst, np.abs(got - want)[cond].max() / np.abs(want)[cond].max())
        assert np.array_equal(out[b * (n + 5) + n: (b + 1) * (n + 5)], init[b * (n + 5) + n: (b + 1) * (n + 5)]), "wrote behind the clip"
    print(f"istft n_fft {N} hop {hop}, {batch} x {T} frames{' + non-Hermitian part' if perturb else ''}: {worst:.2e} of the peak", flush=True)
    assert worst < 3e-6, worst


def stft256_case(rng):
    N, hop = 256, 64
    n, stride = N + 9 * hop, N + 9 * hop + 3
    x = (0.1 * rng.standard_normal(2 * stride)).astype(np.float32)
    win = aligned(N)
    win[:] = (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(N) / N)).astype(np.float32)
    T = (n - N) // hop + 1  # 10: an even count; then 9 (the clip's last frame rides alone)
    for t_use in (T, T - 1):
        re, im = np.full((2 * t_use, N), np.nan, np.float32), np.full((2 * t_use, N), np.nan, np.float32)
        a = AfxStftArgs()
        a.x, a.clipStride, a.batch, a.dataLength, a.timeLength, a.radix2Exp, a.hop = ptr(x), stride, 2, n, t_use, 8, hop
        a.window, a.mode, a.normValue, a.binLo, a.binCount, a.outPitch, a.outRe, a.outIm = ptr(win), 0, 1.0, 0, N, N, ptr(re), ptr(im)
        assert lib.afxk_stft256(C.byref(a), None) == 0
        want = np.stack([np.fft.fft(x[b * stride + i * hop: b * stride + i * hop + N].astype(np.float64) * win) for b in range(2) for i in range(t_use)])
        e = np.abs(re + 1j * im - want).max() / np.abs(want).max()
        print(f"stft n_fft 256 hop {hop}, 2 x {t_use} frames, all 256 bins: {e:.2e} of the peak", flush=True)
        assert e < 2e-6 and not np.isnan(re).any(), e
    # a slice of bins with the power map on a pitched output
    out = np.full((2 * T, 140), np.nan, np.float32)
    a.timeLength, a.mode, a.binLo, a.binCount, a.outPitch, a.outRe = T, 1, 3, 129, 140, ptr(out)
    assert lib.afxk_stft256(C.byref(a), None) == 0
    want = np.stack([np.abs(np.fft.fft(x[b * stride + i * hop: b * stride + i * hop + N].astype(np.float64) * win)[3:132]) ** 2 for b in range(2) for i in range(T)])
    e = np.abs(out[:, :129] - want).max() / want.max()
    assert e < 4e-6 and np.isnan(out[:, 129:]).all(), e
    print(f"stft n_fft 256 power of bins 3 .. 131 on rows of 140: {e:.2e}", flush=True)


rng = np.random.default_rng(21)
stft256_case(rng)
# (two runs per clip need more than 64 / 32 frames: kept to the smaller sizes -- a lane is a host thread here)
istft_case(8, 64, 70, 2, rng, False)
istft_case(8, 100, 9, 1, rng, True)
istft_case(9, 128, 40, 2, rng, True)
istft_case(10, 300, 7, 1, rng, False)
istft_case(11, 512, 36, 1, rng, True)
istft_case(12, 1024, 5, 1, rng, False)
print("OK")
