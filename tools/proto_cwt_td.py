"""numpy model of the time-domain ("Toeplitz") evaluation of the WIDE CWT scales (round 3, afx_cwt_td.hip):
a wavelet that is wide in frequency is short in time, so W_j[n] = sum_t g_j[t] xp[pad + n - t] with
g_j = IFFT(psi_j) truncated to |t| <= Kh_j -- no four-step intermediate.  The product runs on the f16 matrix cores
with (hi, lo) binary16 words of both operands (the formulation of afx_cqt_f16.hip): rows = output positions 8a + p,
columns = (scale, re|im, phase p), K = taps; x scaled by 2^e per tile, g by 2^s per column.
Checks truncation + split against the float64 restatement and the compiled reference on BASELINE cfg 4.
    python tools/proto_cwt_td.py [thr]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref, restate  # noqa: E402

f32 = np.float32


def words(a):
    a = np.asarray(a, f32)
    hi = a.astype(np.float16)
    lo = (a - hi.astype(f32)).astype(np.float16)
    return hi.astype(f32), lo.astype(f32)


def main():
    thr = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-7
    sr, num, r = 44100, 84, 16
    D = 1 << r
    L, pad = 2 * D, D // 2
    o = ref.RefCWT(num=num, radix2_exp=r, samplate=sr, low_fre=32.703, bin_per_octave=12, wavelet_type=1, scale_type=5, is_padding=1)
    x = (0.1 * np.random.default_rng(3).standard_normal(D)).astype(f32)
    x[D // 2:] *= 1e-3  # a level step inside the chunk
    rre, rim = o.cwt(x)
    R = rre + 1j * rim
    fre = np.asarray(o.fre_band(), np.float64)[::-1]
    F = restate.cwt(x.astype(np.float64), fre, sr, "morlet", 6.0, 2.0, True)
    # the reference's own bank rows (float32), natural bin order -> time kernels in double
    w = 2 * np.pi * np.arange(L) / L
    w[L // 2 + 1:] = -w[1:L - L // 2][::-1]
    s = 6.0 / (2 * np.pi * fre / sr)
    xp = np.concatenate([x[:pad][::-1], x, x[D - pad:][::-1]]).astype(f32)
    worst = 0.0
    for j in (0, 13, 27, 39):
        sw = s[j] * w
        psi = np.where(sw > 0, 2 * np.exp(-(sw - 6.0) ** 2 / 2.0), 0.0).astype(f32).astype(np.float64)
        g = np.fft.ifft(psi)                       # g[t mod L]
        a = np.abs(g)
        idx = np.nonzero(a > thr * a.max())[0]
        t = np.where(idx > L // 2, idx - L, idx)
        Kh = int(np.abs(t).max())
        taps = np.concatenate([g[L - Kh:], g[:Kh + 1]])  # g[-Kh .. Kh]
        h = taps[::-1]                                   # y[n] = sum_u h[u] xp[pad + n - Kh + u]
        # exact (float64) truncated correlation
        seg = xp[pad - Kh: pad + D + Kh].astype(np.float64)
        yt = np.convolve(seg, taps, mode="valid")       # sum_t g[t] seg[n + Kh - t + ...]
        # split-f16 product, tiles of 256 outputs (32 rows x 8 phases), per-tile exponent, per-column exponent
        hr, hi_ = h.real.astype(f32), h.imag.astype(f32)
        y = np.zeros(D, complex)
        for part, hv in ((0, hr), (1, hi_)):
            pk = np.abs(hv).max()
            sc = 14 - np.frexp(pk)[1]
            gh, gl = words(np.ldexp(hv, sc))
            for n0 in range(0, D, 256):
                win = xp[pad + n0 - Kh: pad + n0 + 256 + Kh]
                wp = np.abs(win).max()
                e = min(13 - (int(np.frexp(wp)[1]) - 1), 126) if wp >= 2.0 ** -126 else 0
                xh, xl = words(np.ldexp(win, e))
                idx2 = np.arange(len(h))[None, :] + np.arange(256)[:, None]
                acc = (xh[idx2] @ gh).astype(f32) + ((xh[idx2] @ gl).astype(f32) + (xl[idx2] @ gh).astype(f32))
                v = np.ldexp(acc.astype(np.float64), -e - sc)
                y[n0:n0 + 256] += v if part == 0 else 1j * v
        pk = np.abs(F[j]).max()
        blk = lambda z: np.abs(z).reshape(-1, 512).max(axis=1)
        pf = lambda A, B: (blk(A - B) / np.maximum(blk(B), 1e-300))[blk(B) > 1e-6 * blk(B).max()].max()
        print(f"scale {j:2d} f {fre[j]:7.1f} Hz taps {2 * Kh + 1:5d}: truncation vs f64 {np.abs(yt - F[j]).max() / pk:.2e}; "
              f"split-f16 vs f64 {np.abs(y - F[j]).max() / pk:.2e} (per 512-block {pf(y, F[j]):.2e}); "
              f"reference vs f64 {np.abs(R[j] - F[j]).max() / pk:.2e} (per block {pf(R[j], F[j]):.2e}); "
              f"split-f16 vs reference {np.abs(y - R[j]).max() / np.abs(R[j]).max():.2e}")
        worst = max(worst, np.abs(y - R[j]).max() / np.abs(R[j]).max())
    assert worst < 5e-6, worst
    print("OK")


if __name__ == "__main__":
    main()
