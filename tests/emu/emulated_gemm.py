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


lib.afxk_gemm_bank_prepare.restype = C.c_int
lib.afxk_gemm_bank_prepare.argtypes = [fp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_void_p]
lib.afxk_gemm_nt_bank.restype = C.c_int
lib.afxk_gemm_nt_bank.argtypes = [fp, C.c_longlong, C.c_void_p, C.c_int, C.c_int, fp, C.c_longlong, C.c_longlong, C.c_int, C.c_float, C.c_void_p]
lib.afxdev_free.argtypes = [C.c_void_p]


def run_bank(M, N, K, lda, ldb, ldc, post, arg, seed, signed=False):
    """the prepared-bank form (k_bank_split + k_gemm_bank_bf16x3): same operands, same bar; `signed`: A carries both signs
    (the real / imaginary planes of a complex spectrum go through the same product)"""
    rng = np.random.default_rng(seed)
    A, B, Cm = aligned((M, lda)), aligned((N, ldb)), aligned((M, ldc))
    A[:, :K] = (rng.standard_normal((M, K)) ** 2 * 10.0 ** rng.uniform(-5, 5, (M, K))).astype(np.float32)
    if signed:
        A[:, :K] *= rng.choice([-1.0, 1.0], (M, K)).astype(np.float32)
    A[:, K:] = np.nan
    B[:, :K] = np.abs(rng.standard_normal((N, K))).astype(np.float32) * (rng.uniform(0, 1, (N, K)) < 0.7)
    B[:, K:] = np.nan
    Cm[:] = -1.0
    img = C.c_void_p()
    st = lib.afxk_gemm_bank_prepare(B.ctypes.data_as(fp), ldb, N, K, C.byref(img), None)
    assert st == 0 and img.value, st
    st = lib.afxk_gemm_nt_bank(A.ctypes.data_as(fp), lda, img, N, K, Cm.ctypes.data_as(fp), ldc, M, post, arg, None)
    assert st == 0, st
    lib.afxdev_free(img)
    want = A[:, :K].astype(np.float64) @ B[:, :K].astype(np.float64).T
    got = Cm[:, :N].astype(np.float64)
    assert np.all(np.isfinite(got)), "non-finite results (padding read?)"
    if signed:  # sums with cancellation: against the sum of magnitudes
        scale = np.abs(A[:, :K]).astype(np.float64) @ B[:, :K].astype(np.float64).T
    else:
        if post:
            want = want ** float(np.float32(arg))
        scale = np.abs(want)
    err = np.abs(got - want) / np.maximum(scale, 1e-300)
    assert np.all(Cm[:, N:] == -1.0), "wrote past the N columns"
    print(f"bank form M {M} N {N} K {K} pitches {lda}/{ldb}/{ldc} post {post} signed {signed}: elementwise relative error max {err.max():.2e} "
          f"mean {err.mean():.2e}", flush=True)
    assert err.max() <= 4e-6, err.max()


def main():
    AFX_MAP_POW = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    run(300, 130, 77, 80, 80, 136, 0, 0.0, 1)          # every tail: rows, columns, k (77 = 4 x 16 + 13), pitched rows
    run(128, 128, 64, 64, 64, 128, 0, 0.0, 2)          # exactly one tile, no tails
    run(130, 40, 1025, 1028, 1028, 40, AFX_MAP_POW, 0.5, 3)  # the dense filter-bank shape: K = n_fft / 2 + 1, power-law epilogue
    run_bank(300, 130, 77, 80, 80, 136, 0, 0.0, 1)     # two column tiles, row / column / k tails (five k-steps: the ring wraps once)
    run_bank(128, 128, 64, 64, 64, 128, 0, 0.0, 2)     # four k-steps: exactly the ring
    run_bank(130, 40, 1025, 1028, 1028, 40, AFX_MAP_POW, 0.5, 3)  # the dense filter-bank shape, 65 k-steps, one word in the last
    run_bank(70, 128, 33, 36, 36, 128, 0, 0.0, 4, signed=True)    # three k-steps (fewer than the ring), signed rows
    run_bank(5, 12, 7, 8, 8, 12, 0, 0.0, 5)            # one k-step
    print("OK")


if __name__ == "__main__":
    main()
