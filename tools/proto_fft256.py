"""numpy prototype of the n_fft 512 fused kernel's transform (afx_melfused512.hip): 512 real samples = 256 complex points in
FOUR registers per lane, 256 = 4 x 4 x 4 x 4: radix-4 in registers, three transposes through LDS (rows of 5 float2), the last one
lands in natural order -- lane L holds Z[L + 64 q2] -- so only the mirror partners of the real-input split come from LDS.  Every
index formula and an LDS bank-conflict check per instruction class (lane groups as in tools/proto_fft1024_v2.py).  The row
numbering of each transpose is the one permutation of the three base-4 digits that leaves every store free of conflicts
(searched: pitch 5; p1 = 16 d0 + 4 b + c, p2 = 16 d0 + q0 + 4 c, p3 = d0 + 4 q0 + 16 q1).

  n = 64 r + l,  l = 16 a + 4 b + c;   k = d0 + 4 q0 + 16 q1 + 64 q2
  stage 1  lane l = (a, b, c), over r           y[d0] = radix4(z)[d0] W_256^(l d0)         -> row 16 d0 + 4 b + c, column a
  stage 2  lane 16 d0 + 4 b + c, over a         u[q0] = radix4(y)[q0] W_64^((4 b + c) q0)  -> row 16 d0 + q0 + 4 c, column b
  stage 3  lane 16 d0 + q0 + 4 c, over b        v[q1] = radix4(u)[q1] W_16^(c q1)          -> row d0 + 4 q0 + 16 q1, column c
  stage 4  lane L = d0 + 4 q0 + 16 q1, over c   Z[L + 64 q2]
  split    X[k], X[256 - k] from (Z[k], Z[256 - k]), k = L + 64 j, j = 0, 1: Z[k] is register j, Z[256 - k] comes from the
           natural-order image at (64 - L) + 64 (3 - j)  (L = 0: Z[0] itself and, j = 1, Z[192]); bin 128 pairs with itself
"""
import numpy as np

G64R = [list(range(0, 32)), list(range(32, 64))]           # ds_read_b64: 2 x 32 lanes, 64 banks
G64W = [list(range(16 * g, 16 * g + 16)) for g in range(4)]  # ds_write_b64: 4 x 16 lanes, 32 banks


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


N, M, RP = 512, 256, 5
rng = np.random.default_rng(1)
x = rng.standard_normal(N)
z = x[0::2] + 1j * x[1::2]
lane = np.arange(64)
ref = np.fft.rfft(x)
W = lambda n, e: np.exp(-2j * np.pi * e / n)
radix4 = lambda v: np.fft.fft(v, axis=-1)
ex = np.full(64 * RP, np.nan + 0j)
worst = {}

# stage 1: lane l = 16 a + 4 b + c
a, b, c = lane >> 4, (lane >> 2) & 3, lane & 3
y = radix4(np.stack([z[64 * r + lane] for r in range(4)], axis=1)) * W(256, np.outer(lane, np.arange(4)))
for d0 in range(4):
    idx = (16 * d0 + 4 * b + c) * RP + a
    ex[idx] = y[:, d0]
    worst["ex1 write"] = max(worst.get("ex1 write", 1), conflicts(8 * idx, 8, G64W, 32))
rows = lambda: np.stack([ex[lane * RP + i] for i in range(4)], axis=1)
worst["row read"] = max(conflicts(8 * (lane * RP + i), 8, G64R, 64) for i in range(4))
# stage 2: lane 16 d0 + 4 b + c
d0, b, c = lane >> 4, (lane >> 2) & 3, lane & 3
u = radix4(rows()) * W(64, np.outer(4 * b + c, np.arange(4)))
for q0 in range(4):
    idx = (16 * d0 + q0 + 4 * c) * RP + b
    ex[idx] = u[:, q0]
    worst["ex2 write"] = max(worst.get("ex2 write", 1), conflicts(8 * idx, 8, G64W, 32))
# stage 3: lane 16 d0 + q0 + 4 c
d0, c, q0 = lane >> 4, (lane >> 2) & 3, lane & 3
t = radix4(rows()) * W(16, np.outer(c, np.arange(4)))
for q1 in range(4):
    idx = (d0 + 4 * q0 + 16 * q1) * RP + c
    ex[idx] = t[:, q1]
    worst["ex3 write"] = max(worst.get("ex3 write", 1), conflicts(8 * idx, 8, G64W, 32))
# stage 4: lane L = d0 + 4 q0 + 16 q1 holds Z[L + 64 q2]
Zr = radix4(rows())
assert np.allclose(Zr, np.fft.fft(z).reshape(4, 64).T)
img = np.full(256, np.nan + 0j)
for q2 in range(4):
    img[lane + 64 * q2] = Zr[:, q2]
    worst["image write"] = max(worst.get("image write", 1), conflicts(8 * (lane + 64 * q2), 8, G64W, 32))
# split
X = np.zeros(257, complex)
for j in range(2):
    k = lane + 64 * j
    mi = np.where(lane == 0, (256 - 64 * j) & 255, (64 - lane) + 64 * (3 - j))   # index of Z[256 - k]
    worst["mirror read"] = max(worst.get("mirror read", 1), conflicts(8 * mi, 8, G64R, 64))
    A, B = Zr[:, j], img[mi]
    E, O = (A + np.conj(B)) / 2, (A - np.conj(B)) / 2j
    X[k] = E + W(512, k) * O
    X[256 - k] = np.conj(E - W(512, k) * O)
Zm = img[128]
X[128] = (Zm + np.conj(Zm)) / 2 + W(512, 128) * (Zm - np.conj(Zm)) / 2j
assert np.allclose(X, ref)
print("256-point transform and split exact; worst bank multiplicity per class:", worst)
assert all(v == 1 for v in worst.values()), worst
print("OK")
