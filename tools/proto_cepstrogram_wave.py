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


# ---------------------------------------------------------------------------------------------
# closed-form lifters (cepNum <= 16) and the N = 4096 kernel (even / odd split, combine4096)
def lifters_direct(c, q, w, Lk):
    z = np.ones_like(w)
    acc = np.zeros(w.shape)
    for m in range(1, q + 1):
        z = z * w
        acc = acc + c[m] * z.real
    env = c[0] + 2 * acc
    det = Lk - env + (c[q] * z.real if q >= 1 else 0.0)
    return env, det


def direct_2048(frame_windowed, q):
    v = np.stack([frame_windowed[2 * (64 * n1 + lane)] + 1j * frame_windowed[2 * (64 * n1 + lane) + 1]
                  for n1 in range(16)], axis=1)
    x, y, xc, yc = rfft2048(v)
    logp = lambda z: np.log(np.maximum(np.abs(z) ** 2, 1e-16))
    L = to_row((x, y, xc, yc), logp)
    idx = lambda m: np.where(m <= 1024, m, N - m)
    v = np.stack([L[idx(2 * (64 * n1 + lane))] + 1j * L[idx(2 * (64 * n1 + lane) + 1)] for n1 in range(16)], axis=1)
    c = to_row(rfft2048(v), lambda z: np.real(z) / N)
    out2, out3 = np.full(1025, np.nan), np.full(1025, np.nan)
    W = np.exp(-2j * np.pi * np.arange(1024) / N)  # 2 * tw3
    for sb in range(2):
        for j in range(4):
            k = lane + 64 * sb + 256 * j
            t = W[k]
            for kk, w in ((k, t), (1024 - k, -np.conj(t))):
                e, d = lifters_direct(c, q, w, L[kk])
                out2[kk], out3[kk] = e, d
    for kb in (128, 384):
        t = W[kb]
        for kk, w in ((kb, t), (1024 - kb, -np.conj(t))):
            e, d = lifters_direct(c, q, np.array([w]), L[kk:kk + 1])
            out2[kk], out3[kk] = e[0], d[0]
    return c, out2, out3


def bin4096(slot, ln):
    p, r = slot >> 2, slot & 3
    kp = ln + 64 * (p >> 2) + 256 * (p & 3) if p < 8 else 128 + 256 * (p - 8)
    return [kp, 2048 - kp, 1024 - kp, 1024 + kp][r], kp


def combine4096(be, bo):
    """-> X[slot][lane] (true spectrum values), slots as in the kernel"""
    w4 = np.exp(-2j * np.pi * np.arange(1025) / 4096)
    X = np.zeros((40, 64), complex)
    xe, ye, xce, yce = be
    xo, yo, xco, yco = bo

    def position(slot, kp, xE, xO, yE, yO):
        wk, wp = w4[kp], w4[1024 - kp]
        t, u = xO * wk, yO * np.conj(wp)
        X[slot + 0] = xE + t
        X[slot + 1] = np.conj(xE - t)
        X[slot + 2] = np.conj(yE + u)
        X[slot + 3] = yE - u
    for sb in range(2):
        for j in range(4):
            position(4 * (4 * sb + j), lane + 64 * sb + 256 * j, xe[:, sb, j], xo[:, sb, j], ye[:, sb, j], yo[:, sb, j])
    for i in range(2):
        position(32 + 4 * i, 128 + 256 * i, xce[i], xco[i], yce[i], yco[i])
    return X


def rfft4096_slots(s):
    """s[4096] real -> slot values; even / odd samples through rfft2048 in the kernel's packing"""
    n = np.stack([64 * n1 + lane for n1 in range(16)], axis=1)  # [lane, n1]
    ve = s[4 * n] + 1j * s[4 * n + 2]
    vo = s[4 * n + 1] + 1j * s[4 * n + 3]
    return combine4096(rfft2048(ve), rfft2048(vo))


def slots_to_row(X, val):
    row = np.full(2049, np.nan)
    for slot in range(40):
        for ln in (range(64) if slot < 32 else (0,)):
            row[bin4096(slot, ln)[0]] = val(X[slot, ln])
    assert not np.isnan(row).any()
    return row


def direct_4096(frame_windowed, q):
    N4 = 4096
    X = rfft4096_slots(frame_windowed)
    assert np.allclose(slots_to_row(X, lambda z: z.real) + 1j * slots_to_row(X, lambda z: z.imag),
                       np.fft.rfft(frame_windowed))
    L = slots_to_row(X, lambda z: np.log(max(abs(z) ** 2, 1e-16)))
    m = np.arange(N4)
    Lfull = L[np.where(m <= 2048, m, N4 - m)]
    C = rfft4096_slots(Lfull)
    c = slots_to_row(C, lambda z: z.real / N4)
    w4 = np.exp(-2j * np.pi * np.arange(1025) / 4096)
    out2, out3 = np.full(2049, np.nan), np.full(2049, np.nan)
    for slot in range(40):
        for ln in (range(64) if slot < 32 else (0,)):
            k, kp = bin4096(slot, ln)
            r = slot & 3
            wk, wp = w4[kp], w4[1024 - kp]
            w = [wk, -np.conj(wk), wp, -np.conj(wp)][r]
            assert abs(w - np.exp(-2j * np.pi * k / N4)) < 1e-12
            e, d = lifters_direct(c, q, np.array([w]), np.array([L[k]]))
            out2[k], out3[k] = e[0], d[0]
    return c, out2, out3


for q in (0, 1, 4, 16):
    want = restate.cepstrogram(x.astype(np.float32), N, 512, q, window_type=1)
    w = restate.fft_window(1, N)
    for t in range(2):
        got = direct_2048(x.astype(np.float32)[t * 512: t * 512 + N] * w, q)
        for name, g, r in zip(("cepstrum", "envelope", "details"), got, want):
            err = np.abs(g - r[t]).max() / np.abs(r[t]).max()
            assert err < 1e-9, ("direct 2048", q, t, name, err)
x4 = 0.1 * rng.standard_normal(4096 + 2 * 1024)
for q in (0, 1, 4, 16):
    want = restate.cepstrogram(x4.astype(np.float32), 4096, 1024, q, window_type=1)
    w = restate.fft_window(1, 4096)
    for t in range(2):
        got = direct_4096(x4.astype(np.float32)[t * 1024: t * 1024 + 4096] * w, q)
        for name, g, r in zip(("cepstrum", "envelope", "details"), got, want):
            err = np.abs(g - r[t]).max() / np.abs(r[t]).max()
            assert err < 1e-9, ("direct 4096", q, t, name, err)
print("closed-form lifters and N = 4096 prototype OK")
