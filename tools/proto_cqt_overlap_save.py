#!/usr/bin/env python3
"""VERDICT round 4 item 5, decided on a prototype: the three lowest octaves of the CQT ladder (levels 4-6: hop 8 / 4 / 2
under a 512-tap image, i.e. every level sample enters 64-256 frames) as OVERLAP-SAVE in float32 -- one real transform of
a 2048-sample block of the level signal, per bin the product with the 2048-point spectrum of the bin's 512-tap image,
the spectrum folded to every hop-th lag, one complex inverse transform of 2048 / hop points -- against the direct
evaluation (what k_cqt_pyramid computes: one 512-tap product per frame), both from the same float32 level signals and
both against the float64 restatement (oracle/restate.py).  Kill criteria (decided before writing this): the tensor bar
1e-5 and the per-frame bars of tests/test_realaudio_gpu.py::test_cqt_chroma_f16_octave_kernels -- per frame
max|err| / max|frame| <= max(1e-5, the reference's own distance from float64) -- on the hard clips.
usage: proto_cqt_overlap_save.py            (numpy >= 2: float32 transforms stay float32)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import restate  # noqa: E402
from tests import cases  # noqa: E402

SR, NUM, BPO, N = 32000, 84, 12, 512
f32, c64 = np.float32, np.complex64


def level_signals(x):
    """float64 and float32 level signals (the float32 chain rounds every decimation like the device and the reference)"""
    xs64, xs32 = [np.asarray(x, np.float64)], [np.asarray(x, f32)]
    for _ in range(6):
        xs64.append(restate.decimate2(xs64[-1]))
        xs32.append(restate.decimate2(xs32[-1].astype(np.float64)).astype(f32))
    return xs64, xs32


def images():
    """G[j, n] = sum_k K[j, k] e^{-2 pi i k n / N}: the time image of the thresholded spectral kernel"""
    fre, n, lens, K = restate.cqt_plan(NUM, SR, 32.703, BPO, 1, "area")
    assert n == N
    k = np.arange(N // 2 + 1)
    G = K @ np.exp(-2j * np.pi * np.outer(k, np.arange(N)) / N)
    return G, lens


def valid_part(xl, hop):
    """stft_algorithm.c:838-843 through restate.cqt: the level signal is cut to whole hops"""
    frames = len(xl) // hop + 1
    return xl[:len(xl) - (len(xl) % hop if frames > 1 else 0)]


def frames_direct(xl, hop, T, G, dtype):
    xp = np.concatenate([np.zeros(N // 2), valid_part(xl, hop), np.zeros(2 * N + hop * T)])
    idx = np.arange(N)[None, :] + hop * np.arange(T)[:, None]
    if dtype is np.float64:
        return xp[idx] @ G.T
    # float32 evaluation of the same sums (float32 operands, float32 accumulation in blocks of 16 taps like the matrix core)
    A = xp.astype(f32)[idx]
    Gr, Gi = G.real.astype(f32), G.imag.astype(f32)
    acc_r = np.zeros((T, G.shape[0]), f32)
    acc_i = np.zeros((T, G.shape[0]), f32)
    for k0 in range(0, N, 16):
        acc_r += (A[:, k0:k0 + 16].astype(np.float64) @ Gr[:, k0:k0 + 16].T.astype(np.float64)).astype(f32)
        acc_i += (A[:, k0:k0 + 16].astype(np.float64) @ Gi[:, k0:k0 + 16].T.astype(np.float64)).astype(f32)
    return acc_r + 1j * acc_i


def frames_overlap_save(xl, hop, T, G, B=2048):
    """float32: block spectrum x image spectrum, folded to every hop-th lag, inverse transform of B / hop points"""
    xp = np.concatenate([np.zeros(N // 2), valid_part(xl, hop), np.zeros(2 * N + hop * T + B)]).astype(f32)
    H = np.zeros((G.shape[0], B), np.complex128)
    grev = np.zeros((G.shape[0], B), np.complex128)
    grev[:, 0] = G[:, 0]
    grev[:, B - np.arange(1, N)] = G[:, 1:]   # y[lag] = sum_n x[lag + n] G[n]  = (x * grev)[lag], grev[m] = G[-m]
    H = np.fft.fft(grev, axis=1).astype(c64)  # image spectra: built once in double, stored in float32
    per = (B - N) // hop + 1                  # valid frames per block
    out = np.zeros((T, G.shape[0]), c64)
    M = B // hop
    for t0 in range(0, T, per):
        blk = xp[t0 * hop:t0 * hop + B]
        X = np.fft.fft(blk.astype(c64))       # (the device would run the real transform; same rounding class)
        assert X.dtype == c64
        Y = X[None, :] * H                    # complex64
        Yf = Y.reshape(G.shape[0], hop, M).sum(axis=1, dtype=c64) / f32(hop)
        y = np.fft.ifft(Yf, axis=1)
        assert y.dtype == c64
        cnt = min(per, T - t0)
        out[t0:t0 + cnt] = y[:, :cnt].T
    return out


def main():
    G, lens = images()
    print("clip                level  hop | tensor err: direct f32, overlap-save f32 | per-frame err (own peak, all 84 bins): "
          "direct, overlap-save, the compiled reference")
    try:
        from oracle import ref
        have_ref = ref.available()
    except Exception:
        have_ref = False
    worst = {}
    for name in ("level_step", "silence_then_signal", "clicks", "dc_offset"):
        x = cases.hard_clip(name)
        F = restate.cqt(x.astype(np.float64), NUM, SR, 32.703, BPO, 1, "area")
        T = F.shape[0]
        R = None
        if have_ref:
            r = ref.RefCQT(num=NUM, samplate=SR, min_fre=32.703, bin_per_octave=BPO, normal_type=1)
            rre, rim = r.cqt(x)
            R = rre + 1j * rim
        xs64, xs32 = level_signals(x)
        pk_frame = np.abs(F).max(axis=1)
        live = pk_frame > 1e-6 * pk_frame.max()
        for lvl in (4, 5, 6):
            o = 6 - lvl
            hop = 128 >> lvl
            scale = np.sqrt(2.0 ** lvl) / np.sqrt(lens[o * BPO:(o + 1) * BPO])[None, :]
            want = F[:, o * BPO:(o + 1) * BPO]
            d32 = frames_direct(xs32[lvl].astype(np.float64), hop, T, G, f32) * scale
            os32 = frames_overlap_save(xs32[lvl], hop, T, G) * scale
            chk = frames_direct(xs64[lvl], hop, T, G, np.float64) * scale
            assert np.abs(chk - want).max() <= 1e-9 * np.abs(F).max(), "the image does not reproduce the restatement"
            tens = [np.abs(v - want).max() / np.abs(F).max() for v in (d32, os32)]
            pf = [float((np.abs(v - want).max(axis=1)[live] / pk_frame[live]).max()) for v in (d32, os32)]
            pr = float((np.abs(R[:, o * BPO:(o + 1) * BPO] - want).max(axis=1)[live] / pk_frame[live]).max()) if R is not None else float("nan")
            print(f"{name:20s} {lvl}     {hop:2d}  | {tens[0]:.2e}  {tens[1]:.2e} | {pf[0]:.2e}  {pf[1]:.2e}  {pr:.2e}")
            worst[name] = max(worst.get(name, 0.0), pf[1] / max(1e-5, pr if pr == pr else 1e-5))
    print("\nper-frame error of overlap-save over its bar max(1e-5, the reference's own distance), worst level per clip:")
    for k, v in worst.items():
        print(f"  {k:20s} {v:8.1f} x the bar  -> {'PASS' if v <= 1.0 else 'KILL'}")


main()
