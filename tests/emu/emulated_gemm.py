#!/usr/bin/env python3
"""k_gemm_nt128_bf16x3 (audioflux_amd/csrc/hip/afx_gemm_bf16.hip: written without hardware access) on the CPU: its
device code compiled for the host (tests/emu) computes C = A B^T for power-spectrum-like operands spanning ten decades
per row, odd sizes (row / column / k tails, pitched operands), with and without the power-law epilogue, and is compared
ELEMENTWISE with a float64 product.  AFX_LIB = the library tests/test_emulated_kernels.py builds."""
import ctypes as C
import os
import sys

import numpy as np

lib = C.CDLL(os.environ["AFX_LIB"])
fp = C.POINTER(C.c_float)
lib.afxk_gemm_nt128_bf16.restype = C.c_int
lib.afxk_gemm_nt128_bf16.argtypes = [fp, C.c_longlong, fp, C.c_int, fp, C.c_longlong, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_float,
                                     C.c_void_p]


def aligned(shape):
    raw = np.zeros(int(np.prod(shape)) + 16, np.float32)
    off = (-raw.ctypes.data // 4) % 4
    return raw[off:off + int(np.prod(shape))].reshape(shape)


def run(M, N, K, lda, ldb, ldc, post, arg, seed):
    rng = np.random.default_rng(seed)
    A, B, Cm = aligned((M, lda)), aligned((N, ldb)), aligned((M, ldc))
    A[:, :K] = (rng.standard_normal((M, K)) ** 2 * 10.0 ** rng.uniform(-5, 5, (M, K))).astype(np.float32)
    A[:, K:] = np.nan  # the pitch padding must never reach a result
    B[:, :K] = np.abs(rng.standard_normal((N, K))).astype(np.float32) * (rng.uniform(0, 1, (N, K)) < 0.7)
    B[:, K:] = np.nan
    Cm[:] = -1.0
    st = lib.afxk_gemm_nt128_bf16(A.ctypes.data_as(fp), lda, B.ctypes.data_as(fp), ldb, Cm.ctypes.data_as(fp), ldc, M, N, K, post, arg, None)
    assert st == 0, st
    want = A[:, :K].astype(np.float64) @ B[:, :K].astype(np.float64).T
    if post:
        want = want ** float(np.float32(arg))
    got = Cm[:, :N].astype(np.float64)
    assert np.all(np.isfinite(got)), "non-finite results (padding read?)"
    err = np.abs(got - want) / np.maximum(np.abs(want), 1e-300)
    assert np.all(Cm[:, N:] == -1.0), "wrote past the N columns"
    print(f"M {M} N {N} K {K} pitches {lda}/{ldb}/{ldc} post {post}: elementwise relative error max {err.max():.2e} mean {err.mean():.2e}",
          flush=True)
    assert err.max() <= 4e-6, err.max()


def main():
    AFX_MAP_POW = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    run(300, 130, 77, 80, 80, 136, 0, 0.0, 1)          # every tail: rows, columns, k (77 = 4 x 16 + 13), pitched rows
    run(128, 128, 64, 64, 64, 128, 0, 0.0, 2)          # exactly one tile, no tails
    run(130, 40, 1025, 1028, 1028, 40, AFX_MAP_POW, 0.5, 3)  # the dense filter-bank shape: K = n_fft / 2 + 1, power-law epilogue
    print("OK")


if __name__ == "__main__":
    main()
