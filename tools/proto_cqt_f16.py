#!/usr/bin/env python3
"""Index algebra of k_cqt_octave_f16 (audioflux_amd/csrc/hip/afx_cqt_f16.hip), lane by lane in numpy:

  * the wave's window stores (one float4 group per lane and trip, every shifted copy) followed by the A-fragment
    ds_read_b128 of every step return, for lane (i = lane & 31, g = lane >> 5), the samples
    (i hop + 16 ks + 8 g .. + 7) of the tile's window -- for every hop of the default octave ladder;
  * reads are 16-byte aligned, stores as aligned as the kernel assumes, nothing leaves the wave's region,
    pads are never read;
  * every read is bank-conflict free under the gfx950 ds_read_b128 lane groups (MI355X_MICROARCH.md LDS table);
  * the host's fragment order of the image (afx_cqt.c: afx_cqt_time_kernel_f16) pairs A element e of lane half g
    with the same k as B element e of lane half g, and the (hi, lo) f16 split with power-of-two scaling meets
    float32-level accuracy (the three-product sum against a float64 product).

  * the transposed 12-bin epilogue (accumulator layout -> LDS -> four 12-byte stores per lane) and the index algebra
    of k_cqt_decimate's even / odd staging (afx_cqt.hip) are restated lane by lane as well.

Exits non-zero on any mismatch; prints OK."""
import sys

import numpy as np

N, KS = 512, 32
G128 = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
        [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
G128 = G128 + [[l + 32 for l in g] for g in G128]


class Cfg:
    def __init__(self, H):
        self.H = H
        self.COPIES = 1 if H >= 8 else 8 // H
        self.PAD = H >= 16
        self.S = 31 * H + N
        self.NV = (self.S + 255) // 256
        self.MARGIN = 16
        self.RAW = self.MARGIN + 2 * (self.S + 8) + (16 * (self.S // H + 1) if self.PAD else 0)
        if self.COPIES == 1:
            self.CS = (self.RAW + 15) & ~15
        else:
            self.CS = ((self.RAW + 255) & ~255) + (64 if self.COPIES == 4 else 128)
        self.PART = self.COPIES * self.CS
        self.WAVE_BYTES = max(2 * self.PART, 4096)  # the transposed epilogue reuses the region (4 KB)
        self.B_BYTES = 2 * KS * 64 * 16

    def at(self, s, c):
        return self.MARGIN + 2 * (s - c * self.H) + (16 * (s // self.H) if self.PAD else 0)

    def step(self, ks):
        return 32 * ks + (16 * ((16 * ks) // self.H) if self.PAD else 0)


def conflicts(addr, groups):
    worst = 1
    for g in groups:
        banks = {}
        for l in g:
            for d in range(4):
                a = addr[l] + 4 * d
                banks.setdefault((a // 4) % 64, set()).add(a)
        worst = max(worst, max(len(v) for v in banks.values()))
    return worst


def check_hop(H):
    c = Cfg(H)
    assert c.CS % 16 == 0 and c.PART % 16 == 0
    waves = min(8, (160 * 1024 - c.B_BYTES) // c.WAVE_BYTES)
    if waves >= 4:
        waves &= ~3
    assert waves >= 4, (H, waves)
    assert c.B_BYTES + waves * c.WAVE_BYTES <= 160 * 1024
    # the window as int16 "sample ids" (hi plane; the lo plane is the same map + PART)
    lds = np.full(c.PART, -1, np.int64)  # one entry per f16 word of the two planes
    for u in range(c.NV):
        for lane in range(64):
            s = 4 * (lane + 64 * u)
            if s >= c.S:
                continue
            base = c.MARGIN + 2 * s + (16 * (s // H) if c.PAD else 0)
            for cp in range(c.COPIES):
                d = cp * c.CS + base - 2 * cp * H
                assert d >= cp * c.CS, "store below its copy"
                assert d + 8 <= (cp + 1) * c.CS, "store beyond its copy"
                assert d % (8 if (2 * cp * H) % 8 == 0 else 4) == 0, "store alignment"
                for q in range(4):
                    w = d // 2 + q
                    assert lds[w] in (-1, s + q)
                    lds[w] = s + q
                    lds[w + c.PART // 2] = s + q
    worst = 1
    for ks in range(KS):
        addr = np.zeros(64, np.int64)
        for lane in range(64):
            i, g = lane & 31, lane >> 5
            cp = i % c.COPIES
            a = cp * c.CS + c.at(i * H + 8 * g, cp) + c.step(ks)
            assert a % 16 == 0, ("fragment alignment", H, lane, ks, a)
            assert a + 16 <= c.PART
            addr[lane] = a
            want = i * H + 16 * ks + 8 * g + np.arange(8)
            got = lds[a // 2:a // 2 + 8]
            assert np.array_equal(got, want), (H, lane, ks, got, want)
            got_lo = lds[(a + c.PART) // 2:(a + c.PART) // 2 + 8]
            assert np.array_equal(got_lo, want)
        worst = max(worst, conflicts(addr, G128))
    assert worst == 1, f"hop {H}: {worst}-way bank conflict on the A fragments"
    # B fragments: lane-contiguous 16-byte words
    assert conflicts(np.arange(64) * 16, G128) == 1
    return waves, c.WAVE_BYTES


def f16_split(a):
    hi = a.astype(np.float16)
    lo = (a - hi.astype(np.float32)).astype(np.float16)
    return hi, lo


def check_numerics():
    rng = np.random.default_rng(5)
    n = np.arange(N)
    G = np.zeros((N, 32), np.float32)
    for j in range(12):
        L = N - 17 * j
        w = np.zeros(N)
        w[(N - L) // 2:(N - L) // 2 + L] = np.hanning(L) / L
        k = w * np.exp(2j * np.pi * 0.25 * 2 ** (j / 12) * (n - N / 2))
        K = np.fft.fft(k) / N
        K[np.abs(K) < 0.01 * np.abs(K).max()] = 0
        g = np.fft.fft(K)
        G[:, j], G[:, 12 + j] = g.real, g.imag
    # host fragment order: [word][ks][lane][e] <- k = 16 ks + 8 (lane >> 5) + e, column lane & 31
    sj = np.zeros(32, np.int64)
    frag = np.zeros((2, KS, 64, 8), np.float16)
    for j in range(32):
        pk = np.abs(G[:, j]).max()
        sj[j] = 14 - np.frexp(pk)[1] if pk > 0 else 0
        v = np.ldexp(G[:, j], sj[j]).astype(np.float32)
        assert pk == 0 or 2.0 ** 13 <= np.abs(v).max() < 2.0 ** 14
        hi, lo = f16_split(v)
        assert np.all(np.isfinite(hi.astype(np.float32)))
        for k in range(N):
            frag[0, k // 16, 32 * ((k % 16) // 8) + j, k % 8] = hi[k]
            frag[1, k // 16, 32 * ((k % 16) // 8) + j, k % 8] = lo[k]
    worst = 0.0
    for name, x in (("noise", rng.standard_normal(40 * 128 + N)), ("quiet", 1e-6 * rng.standard_normal(40 * 128 + N)),
                    ("tone", np.sin(2 * np.pi * 0.3 * np.arange(40 * 128 + N))), ("loud", 3e7 * rng.standard_normal(40 * 128 + N))):
        x = x.astype(np.float32)
        X = np.stack([x[t * 128:t * 128 + N] for t in range(32)])
        ref = X.astype(np.float64) @ G.astype(np.float64)
        pe = np.frexp(np.abs(X).max())[1] - 1
        e = min(13 - pe, 126)
        xh, xl = f16_split((X * np.float32(2.0 ** e)).astype(np.float32))
        acc = np.zeros((3, 32, 32), np.float32)
        for ks in range(KS):
            for g in range(2):  # lane half g of A pairs with lane half g of B, element by element
                a_h = xh[:, 16 * ks + 8 * g:16 * ks + 8 * g + 8].astype(np.float32)
                a_l = xl[:, 16 * ks + 8 * g:16 * ks + 8 * g + 8].astype(np.float32)
                b_h = frag[0, ks, 32 * g:32 * g + 32].astype(np.float32).T  # [8][32]
                b_l = frag[1, ks, 32 * g:32 * g + 32].astype(np.float32).T
                acc[0] += a_h @ b_h
                acc[1] += a_h @ b_l
                acc[2] += a_l @ b_h
        out = (acc[0] + (acc[1] + acc[2])) * np.float32(2.0 ** -e) * np.ldexp(np.float32(1), -sj)[None, :]
        err = np.abs(out - ref).max() / np.abs(ref).max()
        worst = max(worst, err)
        assert err < 1e-6, (name, err)
    return worst


def check_epilogue():
    """R12 epilogue: accumulator layout -> LDS [frame][plane][piece][4] -> four 12-byte stores per lane"""
    rng = np.random.default_rng(2)
    D = rng.standard_normal((32, 32))  # tile result [frame][column], columns 0..11 re, 12..23 im, 24..31 padding
    lds = np.full(4096 // 4, np.nan)
    for lane in range(64):
        i, g = lane & 31, lane >> 5
        jj = i if i < 12 else i - 12
        if i < 24:
            base = (64 if i >= 12 else 0) + (jj // 3) * 16 + (jj % 3) * 4
        else:
            base = (i - 24) * 16 + 12  # padding columns land in the unused fourth word of a piece
        base += 4 * g * 128
        for r in range(16):
            row = (r & 3) + 8 * (r >> 2) + 4 * g  # D layout of v_mfma_f32_32x32x*: col = lane & 31
            a = base + ((r & 3) + 8 * (r >> 2)) * 128
            assert a % 4 == 0 and a + 4 <= 4096
            if i < 24:
                assert np.isnan(lds[a // 4]), "two lanes write one word"
            lds[a // 4] = D[row, i]
    rows_bytes = 84 * 4
    seen = set()
    for q in range(4):
        for lane in range(64):
            a = (lane >> 2) * 128 + (lane & 3) * 16 + (q >> 1) * 2048 + (q & 1) * 64
            v = lds[a // 4: a // 4 + 3]
            frame, plane, piece = 16 * (q >> 1) + (lane >> 2), q & 1, lane & 3
            want = D[frame, plane * 12 + 3 * piece: plane * 12 + 3 * piece + 3]
            assert np.array_equal(v, want), (q, lane)
            # byte offset inside the clip's plane (colBase = 0, t0 = 0): voff12 + (q >> 1) 16 rowBytes
            off = (lane >> 2) * rows_bytes + 3 * (lane & 3) * 4 + (q >> 1) * 16 * rows_bytes
            assert off == frame * rows_bytes + 3 * piece * 4
            seen.add((frame, plane, piece))
    assert len(seen) == 32 * 2 * 4
    # k_cqt_all_f16 (afx_cqt_all.hip): lane f < 32 takes frame f's 12 (re, im) pairs from the same image for the
    # chroma accumulator: four 16-byte reads per plane at f 128 + 16 p (re) and f 128 + 64 + 16 p (im)
    for f in range(32):
        for p in range(4):
            re = lds[(f * 128 + 16 * p) // 4: (f * 128 + 16 * p) // 4 + 3]
            im = lds[(f * 128 + 64 + 16 * p) // 4: (f * 128 + 64 + 16 * p) // 4 + 3]
            assert np.array_equal(re, D[f, 3 * p: 3 * p + 3]) and np.array_equal(im, D[f, 12 + 3 * p: 12 + 3 * p + 3])


def check_decimator():
    """k_cqt_decimate (afx_cqt.hip): staging XE[k] = x[2 (i0-15+k)], XO[k] = x[2 (i0-16+k) + 1] by 16-byte groups,
    operands E[q + 15 -/+ .], O[q + 16 -/+ .] -> the reference's tap order"""
    rng = np.random.default_rng(3)
    h = rng.standard_normal(32)
    DEC_OUT, DEC_LDS = 1024, 1060
    for src_len, blocks in ((5000, 3), (2049, 2), (40, 1)):
        x = rng.standard_normal(src_len)
        dst_len = src_len // 2
        at = lambda s: x[s] if 0 <= s < src_len else 0.0
        for blk in range(blocks):
            i0 = blk * DEC_OUT
            XE, XO = np.full(DEC_LDS, np.nan), np.full(DEC_LDS, np.nan)
            for q in range(0, (DEC_LDS - 2) // 2 + 1):
                if q == 0:
                    s0 = 2 * (i0 - 15)
                    XE[0], XO[0], XO[1] = at(s0), at(s0 - 1), at(s0 + 1)
                else:
                    k = 2 * q - 1
                    s = 2 * (i0 - 15 + k)
                    assert s % 4 == 0  # one 16-byte load
                    v = [at(s), at(s + 1), at(s + 2), at(s + 3)]
                    XE[k], XE[k + 1] = v[0], v[2]
                    if k + 1 < DEC_LDS:
                        XO[k + 1] = v[1]
                    if k + 2 < DEC_LDS:
                        XO[k + 2] = v[3]
            for tid in (0, 1, 7, 255):
                E, O = XE[4 * tid: 4 * tid + 36], XO[4 * tid: 4 * tid + 36]
                for q in range(4):
                    i = i0 + 4 * tid + q
                    if i >= dst_len:
                        continue
                    acc = 0.0
                    for j in range(32):
                        acc += h[j] * (O[q + 16 - (j + 1) // 2] if j & 1 else E[q + 15 - j // 2])
                    for j in range(1, 32):
                        acc += h[j] * (O[q + 16 + (j - 1) // 2] if j & 1 else E[q + 15 + j // 2])
                    want = sum(h[j] * at(2 * i - j) for j in range(32)) + sum(h[j] * at(2 * i + j) for j in range(1, 32))
                    assert np.isfinite(acc) and abs(acc - want) < 1e-12, (src_len, blk, tid, q)


def main():
    check_epilogue()
    print("transposed epilogue: every (frame, plane, 3-bin piece) leaves exactly once with the right values")
    check_decimator()
    print("decimator staging / operand indices reproduce y[i] = sum h_j x[2i - j] + sum h_j x[2i + j]")
    for H in (128, 64, 32, 16, 8, 4, 2):
        waves, wb = check_hop(H)
        print(f"hop {H:3d}: {waves} waves per CU, {wb} bytes of window planes per wave, fragments conflict-free")
    w = check_numerics()
    print(f"(hi, lo) f16 split, three products, float32 accumulation: worst peak-relative error {w:.2e}")
    print("OK")
    return 0


if __name__ == "__main__":
    sys.exit(main())
