"""Index algebra of the wave-level FFTs used by the fast CWT inverse (afx_cwt.hip):
  fft512: 64 lanes x 8 registers, radix 8 x 8 x 8, two LDS exchanges
  fft256: 16 lanes x 16 registers, radix 16 x 16, one exchange
validated against numpy.fft."""
import numpy as np

rng = np.random.default_rng(0)


def fft512(x):
    W = lambda n, m: np.exp(-2j * np.pi * m / n)
    # regs[a][l] = x[64 a + l]
    r = x.reshape(8, 64).copy()
    # stage 1: DFT-8 over a -> d0 ; twiddle W512^(l d0)
    y = np.zeros((8, 64), complex)
    for d0 in range(8):
        for l in range(64):
            y[d0, l] = sum(r[a, l] * W(8, a * d0) for a in range(8)) * W(512, l * d0)
    # exchange 1: lane l = 8 b + c holds reg d0  ->  lane (d0, c) = d0 + 8 c?? holds reg b
    # choose lane' = 8 d0 + c
    z = np.zeros((8, 64), complex)   # z[b][lane']
    for d0 in range(8):
        for b in range(8):
            for c in range(8):
                z[b, 8 * d0 + c] = y[d0, 8 * b + c]
    # stage 2: DFT-8 over b -> d1 ; twiddle W64^(c d1)
    u = np.zeros((8, 64), complex)   # u[d1][lane' = 8 d0 + c]
    for lp in range(64):
        c = lp & 7
        for d1 in range(8):
            u[d1, lp] = sum(z[b, lp] * W(8, b * d1) for b in range(8)) * W(64, c * d1)
    # exchange 2: lane' (d0, c) reg d1 -> lane'' = d0 + 8 d1 holds reg c
    v = np.zeros((8, 64), complex)
    for d0 in range(8):
        for c in range(8):
            for d1 in range(8):
                v[c, d0 + 8 * d1] = u[d1, 8 * d0 + c]
    # stage 3: DFT-8 over c -> d2 ; X[d0 + 8 d1 + 64 d2] at lane'' = d0 + 8 d1, reg d2
    X = np.zeros(512, complex)
    for lam in range(64):
        for d2 in range(8):
            X[lam + 64 * d2] = sum(v[c, lam] * W(8, c * d2) for c in range(8))
    return X


def fft256(x):
    W = lambda n, m: np.exp(-2j * np.pi * m / n)
    # lane g (16), regs a: x[16 a + g]
    r = x.reshape(16, 16)           # r[a][g]
    y = np.zeros((16, 16), complex)  # y[p][g]
    for p in range(16):
        for g in range(16):
            y[p, g] = sum(r[a, g] * W(16, a * p) for a in range(16)) * W(256, g * p)
    # exchange: lane g reg p -> lane p reg g
    X = np.zeros(256, complex)
    for p in range(16):
        for q in range(16):
            X[p + 16 * q] = sum(y[p, g] * W(16, g * q) for g in range(16))
    return X


x = rng.standard_normal(512) + 1j * rng.standard_normal(512)
e512 = np.abs(fft512(x) - np.fft.fft(x)).max()
print("fft512 err", e512)
assert e512 < 1e-10
x = rng.standard_normal(256) + 1j * rng.standard_normal(256)
e256 = np.abs(fft256(x) - np.fft.fft(x)).max()
print("fft256 err", e256)
assert e256 < 1e-10
print("prototype OK")
