#!/usr/bin/env python3
"""k_gemm_nt128_bf16x3 (audioflux_amd/csrc/hip/afx_gemm_bf16.hip) in numpy:

  * the loader's stores (thread tid: rows tid >> 2 and + 64, four k at 8 (tid & 3) bytes, row pitch 48 bytes) followed
    by the fragment ds_read_b128 of lane (i = lane & 31, g = lane >> 5) at row 48 + 16 g return the eight k values
    8 g .. 8 g + 7 of row 64 w + 32 t + i -- what v_mfma_f32_32x32x16_bf16 pairs between its A and B operands;
  * those reads are bank-conflict free under the gfx950 ds_read_b128 lane groups;
  * three bf16 words per operand (round to nearest even, exact remainders) and the six kept products, accumulated in
    float32, reproduce a float64 product of power-spectrum-like data spanning ten decades per row to the
    ELEMENTWISE accuracy of a float32 GEMM.

Exits non-zero on any mismatch; prints OK."""
import sys

import numpy as np

ROW, TM = 48, 128
PLANE = TM * ROW
G128 = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
        [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
G128 = G128 + [[l + 32 for l in g] for g in G128]


def bf16(a):
    a = np.asarray(a, np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    return ((u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000).astype(np.uint32).view(np.float32)


def check_layout():
    ids = np.full(PLANE // 2, -1, np.int64)  # one entry per bf16 word of a plane: id = row * 16 + k
    for tid in range(256):
        lrow, kq = tid >> 2, tid & 3
        for p in range(2):
            row = lrow + 64 * p
            off = row * ROW + 8 * kq
            assert off % 8 == 0
            for c in range(4):
                assert ids[off // 2 + c] == -1
                ids[off // 2 + c] = row * 16 + 4 * kq + c
    for w in range(2):            # quadrant row / column block of the wave
        for t in range(2):
            addr = np.zeros(64, np.int64)
            for lane in range(64):
                i, g = lane & 31, lane >> 5
                a = (64 * w) * ROW + i * ROW + 16 * g + 32 * t * ROW
                assert a % 16 == 0
                addr[lane] = a
                got = ids[a // 2: a // 2 + 8]
                want = (64 * w + 32 * t + i) * 16 + 8 * g + np.arange(8)
                assert np.array_equal(got, want), (w, t, lane)
            for grp in G128:
                banks = {}
                for l in grp:
                    for d in range(4):
                        banks.setdefault(((addr[l] + 4 * d) // 4) % 64, set()).add(addr[l] + 4 * d)
                assert max(len(v) for v in banks.values()) == 1, "bank conflict on a fragment read"


def check_numerics():
    rng = np.random.default_rng(0)
    F, NB, T = 1025, 128, 64
    x = rng.standard_normal((T, 2048)).astype(np.float32) * np.logspace(-3, 0, 2048)[None, :].astype(np.float32)
    S = (np.abs(np.fft.rfft(x * np.hanning(2048), axis=1)) ** 2).astype(np.float32)
    S = S * np.logspace(0, -9, F)[None, :].astype(np.float32)
    k = np.arange(F)[None, :]
    c = np.linspace(5, 1000, NB)[:, None]
    B = (1 / (1 + ((k - c) / (0.1 * c + 3)) ** 2) ** 2).astype(np.float32)
    ref = S.astype(np.float64) @ B.astype(np.float64).T

    def mm(a, b):
        out = np.zeros((T, NB), np.float32)
        for k0 in range(0, F, 16):
            out += (a[:, k0:k0 + 16] @ b[:, k0:k0 + 16].T).astype(np.float32)
        return out

    def words(a):
        h = bf16(a)
        m = bf16(a - h)
        return h, m, bf16(a - h - m)
    sh, sm, sl = words(S)
    bh, bm, bl = words(B)
    six = mm(sl, bh) + mm(sh, bl) + mm(sm, bm) + mm(sm, bh) + mm(sh, bm) + mm(sh, bh)
    f32 = mm(S, B)
    e6 = (np.abs(six - ref) / np.abs(ref)).max()
    e32 = (np.abs(f32 - ref) / np.abs(ref)).max()
    assert e6 < 2e-6 and e6 < 2 * e32, (e6, e32)
    return e6, e32


def main():
    check_layout()
    print("loader stores -> fragment reads: right rows and k, 16-byte aligned, conflict-free")
    e6, e32 = check_numerics()
    print(f"three bf16 words, six products: elementwise relative error {e6:.2e} (float32 GEMM {e32:.2e})")
    print("OK")
    return 0


if __name__ == "__main__":
    sys.exit(main())
