"""numpy prototype of k_cepstrogram_w2048 (afx_cepstrogram.hip): the lane-level data flow between
the four wave transforms -- spectrum layout of afx_wavefft2048.h (validated on its own by
tools/proto_fft1024.py), natural-order row, even extension, lifter selects, output mapping --
checked against the restatement of the reference (oracle/restate.py::cepstrogram)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import restate

N, F = 2048, 1025
lane = np.arange(64)


def rfft2048(v):
    """v[lane, n1] = (s[2n], s[2n+1]) as complex, n = 64 n1 + lane -> Bins in the kernel's layout"""
    s = np.zeros(N)
    for n1 in range(16):
        n = 64 * n1 + lane
        s[2 * n], s[2 * n + 1] = v[:, n1].real, v[:, n1].imag
    S = np.fft.rfft(s)
    x = np.zeros((64, 2, 4), complex)
    y = np.zeros((64, 2, 4), complex)
    for sb in range(2):
        for j in range(4):
            k = lane + 64 * sb + 256 * j
            x[:, sb, j] = S[k]
            y[:, sb, j] = np.conj(S[1024 - k])
    xc = np.array([S[128], S[384]])
    yc = np.conj(np.array([S[896], S[640]]))
    return x, y, xc, yc


def to_row(b, val):
    x, y, xc, yc = b
    row = np.full(1025, np.nan)
    for sb in range(2):
        for j in range(4):
            k = lane + 64 * sb + 256 * j
            row[k] = val(x[:, sb, j])
            row[1024 - k] = val(y[:, sb, j])
    row[128], row[896], row[384], row[640] = val(xc[0]), val(yc[0]), val(xc[1]), val(yc[1])
    assert not np.isnan(row).any()
    return row


def cepstrogram_frame(frame_windowed, q):
    v = np.stack([frame_windowed[2 * (64 * n1 + lane)] + 1j * frame_windowed[2 * (64 * n1 + lane) + 1]
                  for n1 in range(16)], axis=1)
    row = to_row(rfft2048(v), lambda z: np.log(np.maximum(np.abs(z) ** 2, 1e-16)))
    idx = lambda m: np.where(m <= 1024, m, N - m)
    v = np.stack([row[idx(2 * (64 * n1 + lane))] + 1j * row[idx(2 * (64 * n1 + lane) + 1)] for n1 in range(16)], axis=1)
    b = rfft2048(v)
    out1 = to_row(b, lambda z: np.real(z) / N)
    row = out1
    vl = np.zeros((64, 16), complex)
    vd = np.zeros((64, 16), complex)
    for n1 in range(16):
        m = 2 * (64 * n1 + lane)
        c0, c1 = row[idx(m)], row[idx(m + 1)]
        l0, l1 = (m <= q) | (m >= N - q), (m + 1 <= q) | (m + 1 >= N - q)
        d0, d1 = (m >= q + 1) & (m <= N - q), (m + 1 >= q + 1) & (m + 1 <= N - q)
        vl[:, n1] = np.where(l0, c0, 0) + 1j * np.where(l1, c1, 0)
        vd[:, n1] = np.where(d0, c0, 0) + 1j * np.where(d1, c1, 0)
    out2 = to_row(rfft2048(vl), np.real)
    out3 = to_row(rfft2048(vd), np.real)
    return out1, out2, out3


rng = np.random.default_rng(5)
x = 0.1 * rng.standard_normal(2048 + 3 * 512)
for q in (0, 1, 4, 37, 1022):
    want = restate.cepstrogram(x.astype(np.float32), N, 512, q, window_type=1)
    w = restate.fft_window(1, N)
    for t in range(4):
        got = cepstrogram_frame(x.astype(np.float32)[t * 512: t * 512 + N] * w, q)
        for name, g, r in zip(("cepstrum", "envelope", "details"), got, want):
            err = np.abs(g - r[t]).max() / np.abs(r[t]).max()
            assert err < 1e-9, (q, t, name, err)
print("cepstrogram wave prototype OK")
