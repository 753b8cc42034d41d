"""numpy prototype of the round-2 LDS layouts of the fused kernel (afx_melfused2.hip):
every index formula, the lane-0 re-mapping that removes the redundant "centre" butterflies,
and an LDS bank-conflict check per instruction class, validated before the HIP transcription.

Changes against tools/proto_fft1024.py (round 1):
  * window / W_1024 tables in pair layout [(n1>>1)][lane][n1&1]: one ds_read_b128 = two rows
  * exchange 1: row pitch 72 float2, writer lane l = 4 m1 + m2 stores at column
    8 (m1>>1) + 2 m2 + (m1&1), so the reader (k1, m2) takes (m1, m1+1) with one ds_read_b128
  * exchange 2: image V[q][m2] (m2 fastest), the two 16-byte halves of a row swapped when
    bit 3 of q is set (conflict-free ds_read_b128 without padding: 8192 bytes)
  * final stage: 513 conjugate pairs on 512 slots -- lane 0's mirror side reads q = 128
    (the self-mirrored base) instead of q = 0, its slots 2, 3 take the pairs (128, 896),
    (384, 640), and bin 512 is |Z[512]|^2 directly; no lane computes a pair twice
  * W_2048 table per lane: tw3L[s][lane][m]; lane 0's s = 0 row is {W^0, W^256, W^128, W^384}
"""
import numpy as np

N, M = 2048, 1024
rng = np.random.default_rng(0)
x = rng.standard_normal(N)
z = x[0::2] + 1j * x[1::2]
lane = np.arange(64)
ref = np.fft.rfft(x)

# ds_read_b128 lane groups (MI355X_MICROARCH.md LDS table); b64 reads: 2 x 32; b64 writes: 4 x 16
G128 = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
        [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
G128 = G128 + [[l + 32 for l in g] for g in G128]
G64R = [list(range(0, 32)), list(range(32, 64))]
G64W = [list(range(16 * g, 16 * g + 16)) for g in range(4)]


def conflicts(addr, width, groups, nbanks):
    """worst multiplicity of distinct addresses on one bank within a lane group (1 = conflict-free)"""
    worst = 1
    for g in groups:
        banks = {}
        for l in g:
            for d in range(width // 4):
                b = ((addr[l] + 4 * d) // 4) % nbanks
                banks.setdefault(b, set()).add(addr[l] + 4 * d)
        worst = max(worst, max(len(v) for v in banks.values()))
    return worst


def dft4(p):
    s0, s1, s2, s3 = p[0] + p[2], p[0] - p[2], p[1] + p[3], p[1] - p[3]
    return [s0 + s2, s1 - 1j * s3, s0 - s2, s1 + 1j * s3]


def dft16(a):
    return np.fft.fft(a, axis=-1)


# ---- pass 1 + exchange 1 -------------------------------------------------------------------
a = np.stack([z[64 * n1 + lane] for n1 in range(16)], axis=1)
Y = dft16(a) * np.exp(-2j * np.pi * np.outer(lane, np.arange(16)) / M)
P1 = 72  # float2 per row
ex1 = np.full(16 * P1, np.nan + 0j)
m1w, m2w = lane >> 2, lane & 3
posL = 8 * (m1w >> 1) + 2 * m2w + (m1w & 1)
for k in range(16):
    ex1[k * P1 + posL] = Y[:, k]
    assert conflicts((k * P1 + posL) * 8, 8, G64W, 32) == 1
k1, m2 = lane >> 2, lane & 3
b = np.zeros((64, 16), complex)
for jj in range(8):
    at = k1 * P1 + 8 * jj + 2 * m2  # float2 index of a 16-byte pair
    assert np.all(at % 2 == 0)
    assert conflicts(at * 8, 16, G128, 64) == 1, ("ex1 read", jj)
    b[:, 2 * jj] = ex1[at]
    b[:, 2 * jj + 1] = ex1[at + 1]
assert not np.isnan(b).any()

# ---- pass 2 + exchange 2: V[q][m2], halves swapped when bit 3 of q ------------------------
V = dft16(b) * np.exp(-2j * np.pi * np.outer(m2, np.arange(16)) / 64)


def vaddr(q, m):  # float2 index of V[q][m]
    return 4 * q + 2 * ((m >> 1) ^ ((q >> 3) & 1)) + (m & 1)


ex2 = np.full(1024, np.nan + 0j)
for j1 in range(16):
    q = k1 + 16 * j1
    at = vaddr(q, m2)
    base = vaddr(k1, m2)
    assert np.all(at == base + 64 * j1)  # constant stride 64 float2 per j1: ds_write2_b64 offsets 0,64,128,192
    assert conflicts(at * 8, 8, G64W, 32) == 1
    ex2[at] = V[:, j1]
assert not np.isnan(ex2).any()

# ---- final radix-4 + real split, 512 slots for 513 pairs ----------------------------------
W2 = np.exp(-2j * np.pi * np.arange(1024) / N)
X = np.full(M + 1, np.nan + 0j)
written = np.zeros(M + 1, int)


def read_row(qv):
    """two ds_read_b128 per lane: first = 32 q + 16 bit3(q), second = 32 q + 16 (1 - bit3(q))"""
    b3 = (qv >> 3) & 1
    lo = 4 * qv + 2 * b3        # holds m = 0, 1
    hi = 4 * qv + 2 * (1 - b3)  # holds m = 2, 3
    assert conflicts(lo * 8, 16, G128, 64) == 1 and conflicts(hi * 8, 16, G128, 64) == 1, "ex2 read"
    return [ex2[lo], ex2[lo + 1], ex2[hi], ex2[hi + 1]]


# constant tables (round 5): the W_64 rows -- lane (k1, m2) reads row m2 at 16 j, four distinct addresses per lane group -- are
# conflict-free at a row pitch of 144 bytes; the natural 128 puts rows m2 and m2 + 2 on the same banks (every read two-way:
# the 6.4 % of the kernel's LDS cycles SQ_LDS_BANK_CONFLICT reported through round 4).  Window / W_1024 rows: 16 bytes per lane.
for j in range(8):
    assert conflicts(144 * (lane & 3) + 16 * j, 16, G128, 64) == 1, "W_64 rows at pitch 144"
    assert conflicts(128 * (lane & 3) + 16 * j, 16, G128, 64) == 2, "W_64 rows at pitch 128 (the round-2 layout)"
    assert conflicts(16 * lane + 1024 * j, 16, G128, 64) == 1, "window / W_1024 rows"

tw3L = np.zeros((2, 64, 4), complex)  # 0.5 W_2048^k of the slot's P-bin
pbin = np.zeros((2, 64, 4), int)
for s in range(2):
    for ln in range(64):
        for j in range(4):
            pbin[s, ln, j] = ln + 64 * s + 256 * j
pbin[0, 0, 2], pbin[0, 0, 3] = 128, 384
tw3L = 0.5 * W2[pbin]

for s in range(2):
    q = lane + 64 * s
    qm = (256 - q) & 255
    if s == 0:
        qm = np.where(lane == 0, 128, qm)
    za = dft4(read_row(q))    # Z[q + 256 j]
    zb = dft4(read_row(qm))   # Z[qm + 256 j]
    for j in range(4):
        A = za[j]
        B = zb[3 - j]
        if s == 0:  # lane 0: slots (za0, za0), (za1, za3), (zb0, zb3), (zb1, zb2)
            A0 = [za[0], za[1], zb[0], zb[1]][j]
            B0 = [za[0], za[3], zb[3], zb[2]][j]
            A = np.where(lane == 0, A0, A)
            B = np.where(lane == 0, B0, B)
        w = tw3L[s, :, j]
        e2 = A + np.conj(B)
        d = A - np.conj(B)
        wo = w * (-1j * d)
        xk = 0.5 * e2 + wo
        yk = 0.5 * e2 - wo
        kb = pbin[s, :, j]
        X[kb] = xk
        X[M - kb] = np.conj(yk)
        written[kb] += 1
        written[M - kb] += 1
    if s == 0:
        X[512] = np.conj(za[2][0])  # lane 0: Z[512]
        written[512] += 1
assert not np.isnan(X).any()
assert np.all(written == 1), np.nonzero(written != 1)
assert np.allclose(X, ref), np.abs(X - ref).max()

# ---- power-row write addressing: ds_write2st64_b32 (two dwords 64-dword units apart) ------
# P: general lane bins k + 256 j from base k (s = 0: k = lane; s = 1: k = 64 + lane via offset +1 unit)
aP01 = lane            # float index; offsets 0, 4 units (x64 floats): bins k, k + 256
aP23 = np.where(lane == 0, 128, lane + 512)  # offsets 0, 4: (k+512, k+768) | (128, 384)
aQ01 = 768 - lane      # offsets 0 (slot 1), 4 (slot 0)
aQ23 = np.where(lane == 0, 640, 256 - lane)  # offsets 0 (slot 3), 4 (slot 2)
aQs1 = 192 - lane      # s = 1: offsets 0, 4, 8, 12 for slots 3, 2, 1, 0
for j in range(4):
    assert np.all((aP01 if j < 2 else aP23) + 256 * (j & 1) == pbin[0, :, j])
    assert np.all(aP01 + 64 + 256 * j == pbin[1, :, j])
    assert np.all((aQ01 + 256 * (1 - j) if j < 2 else aQ23 + 256 * (3 - j)) == M - pbin[0, :, j])
    assert np.all(aQs1 + 256 * (3 - j) == M - pbin[1, :, j])
print("v2 layouts OK: max err", np.abs(X - ref).max())
