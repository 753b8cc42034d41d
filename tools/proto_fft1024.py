"""numpy prototype of the register/LDS FFT decomposition used by the fused kernel
(afx_melfused.hip): validates every index formula before it is transcribed to HIP.
One wave (64 lanes) transforms one real frame of 2048 samples via a 1024-point
complex FFT: radix-16 in registers -> twiddle -> LDS transpose -> radix-16 in registers
-> twiddle -> radix-4 across each lane quad -> real-input split."""
import numpy as np

N = 2048
M = N // 2
rng = np.random.default_rng(0)
x = rng.standard_normal(N)
z = x[0::2] + 1j * x[1::2]
lane = np.arange(64)


def dft16_rows(a):
    """a[..., 16] -> DFT over last axis, radix-4 x radix-4 as in the kernel"""
    W16 = np.exp(-2j * np.pi * np.arange(16) / 16)
    t = np.zeros_like(a)
    # step 1: for each b, 4-pt DFT over a-index (inputs x[4a'+b])
    for b in range(4):
        p0, p1, p2, p3 = a[..., b], a[..., 4 + b], a[..., 8 + b], a[..., 12 + b]
        s0, s1, s2, s3 = p0 + p2, p0 - p2, p1 + p3, p1 - p3
        o = [s0 + s2, s1 - 1j * s3, s0 - s2, s1 + 1j * s3]
        for c in range(4):
            t[..., 4 * b + c] = o[c] * W16[(b * c) % 16]  # t[b][c]
    out = np.zeros_like(a)
    for c in range(4):
        p0, p1, p2, p3 = t[..., c], t[..., 4 + c], t[..., 8 + c], t[..., 12 + c]
        s0, s1, s2, s3 = p0 + p2, p0 - p2, p1 + p3, p1 - p3
        o = [s0 + s2, s1 - 1j * s3, s0 - s2, s1 + 1j * s3]
        for d in range(4):
            out[..., c + 4 * d] = o[d]
    return out


assert np.allclose(dft16_rows(np.eye(16, dtype=complex)), np.fft.fft(np.eye(16), axis=1))

# pass 1: lane n2 holds a[n1] = z[64 n1 + n2]
a = np.stack([z[64 * n1 + lane] for n1 in range(16)], axis=1)  # [lane, n1]
Y = dft16_rows(a)  # [lane=n2, k1]
tw1 = np.exp(-2j * np.pi * np.outer(lane, np.arange(16)) / M)  # W_1024^{n2 k1}
Y = Y * tw1
# LDS exchange, pitch 68 float2 per k1 row
PITCH = 68
lds = np.zeros(16 * PITCH, complex)
for k1 in range(16):
    lds[k1 * PITCH + lane] = Y[:, k1]
# pass 2: lane l: k1 = l>>2, m2 = l&3 reads b[m1] = lds[k1*68 + 4 m1 + m2]
k1 = lane >> 2
m2 = lane & 3
b = np.stack([lds[k1 * PITCH + 4 * m1 + m2] for m1 in range(16)], axis=1)
V = dft16_rows(b)  # [lane, j1]
tw2 = np.exp(-2j * np.pi * np.outer(m2, np.arange(16)) / 64)  # W_64^{m2 j1}
V = V * tw2
# quad radix-4 across lanes via xor-2 then xor-1 butterflies
o = V[lane ^ 2]
r = np.where((m2 & 2)[:, None] != 0, o - V, V + o)
o = r[lane ^ 1]
U = np.empty_like(r)
sel = m2[:, None]
U = np.where(sel == 0, r + o, np.where(sel == 1, o - r, np.where(sel == 2, r - 1j * o, o + 1j * r)))
j2 = ((lane & 1) << 1) | ((lane >> 1) & 1)
Zk = np.zeros(M, complex)
for j1 in range(16):
    Zk[k1 + 16 * j1 + 256 * j2] = U[:, j1]
assert np.allclose(Zk, np.fft.fft(z)), np.abs(Zk - np.fft.fft(z)).max()

# natural-order LDS image with 4-float2 pad per 256: phys(k) = k + 4*(k>>8)
phys = lambda k: k + 4 * (k >> 8)
zl = np.zeros(1040, complex)
for j1 in range(16):
    k = k1 + 16 * j1 + 256 * j2
    zl[phys(k)] = U[:, j1]
# real split: lane l, i = 0..7: k = l + 64 i, partner (1024-k)&1023
X = np.zeros(M + 1, complex)
for i in range(8):
    k = lane + 64 * i
    kp = (M - k) & (M - 1)
    A, B = zl[phys(k)], zl[phys(kp)]
    E = 0.5 * (A + np.conj(B))
    O = -0.5j * (A - np.conj(B))
    W = np.exp(-2j * np.pi * k / N)
    X[k] = E + W * O
    X[M - k] = np.conj(E - W * O)
X[512] = np.conj(zl[phys(512)])
ref = np.fft.rfft(x)
assert np.allclose(X, ref), np.abs(X - ref).max()
print("prototype OK: max err", np.abs(X - ref).max())

# ---------------------------------------------------------------------------------------
# v4 final stage: the cross-quad radix-4 is folded into the real-input split.  After pass 2
# and the W_64 twiddle, lane (k1, m2) writes V_m2[q = k1 + 16 j1] to the LDS image
# img[m2*260 + q]; then every lane takes bases q = lane, 64 + lane (and all lanes q = 128),
# runs the radix-4 for base q and its mirror 256 - q in registers and forms the pairs.
def dft4(p):
    s0, s1, s2, s3 = p[0] + p[2], p[0] - p[2], p[1] + p[3], p[1] - p[3]
    return [s0 + s2, s1 - 1j * s3, s0 - s2, s1 + 1j * s3]

img = np.zeros(4 * 260, complex)
for j1 in range(16):
    img[m2 * 260 + k1 + 16 * j1] = V[:, j1]
W2 = np.exp(-2j * np.pi * np.arange(1024) / N)
X4 = np.full(M + 1, np.nan + 0j)
def pair(k, A, B):
    E = 0.5 * (A + np.conj(B)); O = -0.5j * (A - np.conj(B))
    X4[k] = E + W2[k] * O
    X4[M - k] = np.conj(E - W2[k] * O)
for ln in range(64):
    for s in range(2):
        q = ln + 64 * s
        qp = (256 - q) & 255
        Za = dft4([img[m * 260 + q] for m in range(4)])
        Zb = dft4([img[m * 260 + qp] for m in range(4)])
        for j in range(4):
            B = Zb[(4 - j) & 3] if q == 0 else Zb[3 - j]
            pair(q + 256 * j, Za[j], B)
Zc = dft4([img[m * 260 + 128] for m in range(4)])
for j in range(2):
    pair(128 + 256 * j, Zc[j], Zc[3 - j])
assert not np.isnan(X4).any()
assert np.allclose(X4, ref), np.abs(X4 - ref).max()
print("v4 final stage OK: max err", np.abs(X4 - ref).max())
