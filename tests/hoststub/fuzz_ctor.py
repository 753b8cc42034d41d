#!/usr/bin/env python3
"""Random constructor arguments (valid, borderline and invalid) for every object of the C API, through raw ctypes,
against the sanitizer build of the host code (tests/test_hoststub.py sets AFX_LIB and preloads the ASan runtime).
A constructor either returns 0 and a usable handle -- one small compute call and free follow -- or a non-zero status
(negative; 1 for the reference's "scaleType is error") and no handle; anything else (a crash, a sanitizer report) fails.  The reference validates little
and reads out of bounds on several of these combinations; this library must refuse them."""
import ctypes as C
import os
import sys

import numpy as np

lib = C.CDLL(os.environ["AFX_LIB"])
rng = np.random.default_rng(int(os.environ.get("AFX_FUZZ_SEED", "1")))
P = C.POINTER


VERBOSE = os.environ.get("AFX_FUZZ_VERBOSE")  # print every argument drawn (to reproduce a finding by hand)


def opt(val, ctype):
    """an optional pointer argument: NULL one time in three"""
    if rng.integers(0, 3) == 0:
        if VERBOSE:
            print("  opt NULL", flush=True)
        return None
    if VERBOSE:
        print("  opt", val, flush=True)
    return C.byref(ctype(val))


def pick(*vals):
    v = vals[rng.integers(0, len(vals))]
    if VERBOSE:
        print("  pick", v, flush=True)
    return v


def irange(lo, hi):
    return int(rng.integers(lo, hi + 1))


counts = {"ok": 0, "refused": 0}


def status(st, handle):
    assert st <= 1, f"status {st}"  # 1: the reference's own code for a scale type out of range (kept: bft / cwt / pwt)
    if st == 0:
        assert handle.value, "status 0 without a handle"
        counts["ok"] += 1
        return True
    counts["refused"] += 1
    return False


def f32(n):
    return (0.1 * rng.standard_normal(max(n, 1))).astype(np.float32)


def vp(a):
    return a.ctypes.data_as(C.c_void_p)


def fuzz_bft():
    h = C.c_void_p()
    num, r2 = pick(0, 1, 2, 13, 40, 128, 129, 600, 5000), pick(0, 1, 5, 8, 10, 11, 12, 15, 31)
    st = lib.bftObj_new(C.byref(h), num, r2, opt(pick(0, 8000, 16000, 44100), C.c_int), opt(pick(-5.0, 0.0, 27.5, 300.0), C.c_float),
                        opt(pick(-1.0, 0.0, 4000.0, 8000.0, 1e6), C.c_float), opt(pick(0, 3, 12, 24, 49), C.c_int),
                        opt(irange(-1, 13), C.c_int), opt(pick(-3, 0, 1, 128, 512, 5000), C.c_int), opt(irange(-1, 8), C.c_int),
                        opt(irange(-1, 12), C.c_int), opt(irange(-1, 3), C.c_int), opt(irange(-1, 3), C.c_int),
                        opt(irange(0, 1), C.c_int), opt(irange(0, 1), C.c_int))
    if status(st, h):
        n = pick(1, 100, 3000, 20000)
        lib.bftObj_calTimeLength.restype = C.c_int
        T = lib.bftObj_calTimeLength(h, n)
        assert 0 <= T < 10 ** 7
        x = f32(n)
        re, im = np.zeros((max(T, 1), max(num, 1)), np.float32), np.zeros((max(T, 1), max(num, 1)), np.float32)
        lib.bftObj_bft(h, x.ctypes.data_as(C.c_void_p), n, re.ctypes.data_as(C.c_void_p), im.ctypes.data_as(C.c_void_p))
        lib.bftObj_free(h)


def fuzz_cqt():
    h = C.c_void_p()
    num = pick(0, 1, 12, 13, 48, 84, 96, 500)
    st = lib.cqtObj_newWith(C.byref(h), num, opt(pick(0, 8000, 32000, 44100), C.c_int), opt(pick(-1.0, 0.0, 32.703, 500.0, 1e5), C.c_float),
                            opt(pick(0, 1, 12, 24, 36, 100), C.c_int), opt(pick(-1.0, 0.0, 1.0, 8.0), C.c_float),
                            opt(pick(0.0, 5.0), C.c_float), opt(pick(-1.0, 0.0, 0.01, 2.0), C.c_float), opt(irange(-1, 13), C.c_int),
                            opt(pick(-1, 0, 1, 128, 4096), C.c_int), opt(irange(0, 1), C.c_int), opt(irange(-1, 3), C.c_int),
                            opt(irange(0, 1), C.c_int))
    if status(st, h):
        n = pick(1, 700, 9000)
        lib.cqtObj_calTimeLength.restype = C.c_int
        T = lib.cqtObj_calTimeLength(h, n)
        assert 0 <= T < 10 ** 7
        x = f32(n)
        re, im = np.zeros((max(T, 1), num), np.float32), np.zeros((max(T, 1), num), np.float32)
        lib.cqtObj_cqt(h, x.ctypes.data_as(C.c_void_p), n, re.ctypes.data_as(C.c_void_p), im.ctypes.data_as(C.c_void_p))
        ch = np.zeros((max(T, 1), 12), np.float32)
        lib.cqtObj_chroma(h, opt(pick(0, 5, 6, 12, 13), C.c_int), None, None, re.ctypes.data_as(C.c_void_p),
                          im.ctypes.data_as(C.c_void_p), ch.ctypes.data_as(C.c_void_p))
        lib.cqtObj_free(h)


def big(num, r2):
    """a valid but huge plan (the host builds a [num, 2^r2] bank): minutes of honest work, not a target here"""
    return 1 <= r2 <= 30 and num >= 2 and num * (1 << r2) > 1 << 24


def fuzz_cwt():
    h = C.c_void_p()
    num, r2 = pick(0, 1, 2, 12, 84, 300, 5000), pick(0, 1, 6, 10, 12, 13, 14, 17, 20, 31)
    if big(num, r2):
        return
    st = lib.cwtObj_new(C.byref(h), num, r2, opt(pick(0, 16000, 44100), C.c_int), opt(pick(-1.0, 0.0, 32.703, 1e5), C.c_float),
                        opt(pick(-1.0, 0.0, 8000.0, 1e6), C.c_float), opt(pick(0, 3, 12, 49), C.c_int), opt(irange(-1, 9), C.c_int),
                        opt(irange(-1, 8), C.c_int), opt(pick(-1.0, 0.0, 3.0, 6.0), C.c_float), opt(pick(-1.0, 0.0, 2.0, 20.0), C.c_float),
                        opt(irange(0, 1), C.c_int))
    if status(st, h):
        L = 1 << r2
        if L * num <= 1 << 22:
            x = f32(L)
            re, im = np.zeros((num, L), np.float32), np.zeros((num, L), np.float32)
            lib.cwtObj_cwt(h, x.ctypes.data_as(C.c_void_p), re.ctypes.data_as(C.c_void_p), im.ctypes.data_as(C.c_void_p))
        lib.cwtObj_free(h)


def fuzz_stft():
    h = C.c_void_p()
    r2 = pick(0, 1, 4, 9, 11, 14, 15, 31)
    st = lib.stftObj_new(C.byref(h), r2, opt(irange(-1, 13), C.c_int), opt(pick(-1, 0, 1, 100, 512, 100000), C.c_int), opt(irange(0, 1), C.c_int))
    if status(st, h):
        n = pick(1, 50, 5000)
        lib.stftObj_calTimeLength.restype = C.c_int
        T = lib.stftObj_calTimeLength(h, n)
        assert -1 <= T < 10 ** 7
        if T > 0 and T * (1 << r2) <= 1 << 22:
            x = f32(n)
            re, im = np.zeros((T, 1 << r2), np.float32), np.zeros((T, 1 << r2), np.float32)
            lib.stftObj_stft(h, x.ctypes.data_as(C.c_void_p), n, re.ctypes.data_as(C.c_void_p), im.ctypes.data_as(C.c_void_p))
        lib.stftObj_free(h)


def fuzz_misc():
    h = C.c_void_p()
    xn = pick(-1, 0, 1, 13, 128, 100000)
    st = lib.xxccObj_new(C.byref(h), xn)
    if status(st, h):
        if xn <= 128:
            T, cc = pick(1, 7, 300), pick(1, 13, xn, xn + 1)
            lib.xxccObj_setTimeLength(h, T)
            m, out = np.abs(f32(T * xn)) + 1e-3, np.zeros(T * max(cc, 1), np.float32)
            lib.xxccObj_xxcc(h, vp(m), cc, opt(irange(-1, 3), C.c_int), vp(out))
        lib.xxccObj_free(h)
    h = C.c_void_p()
    cr = pick(0, 1, 8, 11, 12, 16, 31)
    st = lib.cepstrogramObj_new(C.byref(h), cr, opt(irange(-1, 13), C.c_int), opt(pick(-1, 0, 1, 256), C.c_int))
    if status(st, h):
        n = pick(1, 300, 5000)
        lib.cepstrogramObj_calTimeLength.restype = C.c_int
        T = lib.cepstrogramObj_calTimeLength(h, n)
        assert 0 <= T < 10 ** 7
        F = (1 << cr) // 2 + 1
        if T * F <= 1 << 22:
            x = f32(n)
            o1, o2, o3 = (np.zeros(max(T, 1) * F, np.float32) for _ in range(3))
            lib.cepstrogramObj_cepstrogram(h, pick(-1, 0, 1, 4, F, F + 5), vp(x), n, vp(o1), vp(o2) if pick(0, 1) else None,
                                           vp(o3) if pick(0, 1) else None)
        lib.cepstrogramObj_free(h)
    h = C.c_void_p()
    st = lib.spectrogramObj_new(C.byref(h), pick(0, 1, 40, 128, 3000), opt(pick(0, 16000, 32000), C.c_int), opt(pick(-1.0, 0.0, 27.5), C.c_float),
                                opt(pick(0.0, 8000.0, 1e6), C.c_float), opt(pick(0, 12, 36), C.c_int), opt(pick(0, 5, 10, 12, 17), C.c_int),
                                opt(irange(-1, 13), C.c_int), opt(pick(-1, 0, 128, 99999), C.c_int), opt(irange(0, 1), C.c_int),
                                opt(irange(-1, 3), C.c_int), opt(irange(-1, 11), C.c_int), opt(irange(-1, 12), C.c_int), opt(irange(-1, 3), C.c_int))
    if status(st, h):
        lib.spectrogramObj_getBandNum.restype = C.c_int
        lib.spectrogramObj_calTimeLength.restype = C.c_int
        n = pick(1, 600, 20000)
        bn, T = lib.spectrogramObj_getBandNum(h), lib.spectrogramObj_calTimeLength(h, n)
        assert 0 <= bn < 10 ** 6 and 0 <= T < 10 ** 7
        if T * max(bn, 1) <= 1 << 22:
            x = f32(n)
            sp, ph = np.zeros(max(T * bn, 1), np.float32), np.zeros(max(T * bn, 1), np.float32)
            lib.spectrogramObj_spectrogram(h, vp(x), n, vp(sp), vp(ph) if pick(0, 1) else None)
        lib.spectrogramObj_free(h)
    h = C.c_void_p()
    pn, pr = pick(0, 2, 84, 3000), pick(0, 6, 12, 14, 20)
    if big(pn, pr):
        pn = 2
    st = lib.pwtObj_new(C.byref(h), pn, pr, opt(pick(0, 32000), C.c_int), opt(pick(-1.0, 32.703), C.c_float),
                        opt(pick(0.0, 8000.0), C.c_float), opt(pick(0, 12), C.c_int), opt(irange(-1, 8), C.c_int), opt(irange(-1, 12), C.c_int),
                        opt(irange(-1, 3), C.c_int), opt(irange(0, 1), C.c_int))
    if status(st, h):
        L = 1 << pr
        if L * pn <= 1 << 22:
            x, re, im = f32(L), np.zeros(pn * L, np.float32), np.zeros(pn * L, np.float32)
            lib.pwtObj_pwt(h, vp(x), vp(re), vp(im))
        lib.pwtObj_free(h)
    h = C.c_void_p()
    wn, wr = pick(0, 2, 84, 3000), pick(0, 6, 12, 14, 20)
    if big(wn, wr):
        wn = 2
    st = lib.wsstObj_new(C.byref(h), wn, wr, opt(pick(0, 32000), C.c_int), opt(pick(-1.0, 32.703), C.c_float),
                         opt(pick(0.0, 8000.0), C.c_float), opt(pick(0, 12), C.c_int), opt(irange(-1, 9), C.c_int), opt(irange(-1, 8), C.c_int),
                         opt(pick(0.0, 6.0), C.c_float), opt(pick(0.0, 2.0), C.c_float), opt(pick(-1.0, 0.0, 0.001), C.c_float), opt(irange(0, 1), C.c_int))
    if status(st, h):
        L = 1 << wr
        if L * wn <= 1 << 20:
            x = f32(L)
            a, b, c, d = (np.zeros(wn * L, np.float32) for _ in range(4))
            two = pick(0, 1)
            lib.wsstObj_wsst(h, vp(x), vp(a), vp(b), vp(c) if two else None, vp(d) if two else None)
        lib.wsstObj_free(h)
    h = C.c_void_p()
    rr = pick(0, 1, 9, 11, 15, 31)
    st = lib.reassignObj_new(C.byref(h), rr, opt(pick(0, 16000), C.c_int), opt(irange(-1, 13), C.c_int),
                             opt(pick(-1, 0, 64, 99999), C.c_int), opt(irange(-1, 4), C.c_int), opt(pick(-1.0, 0.0, 0.001), C.c_float),
                             opt(irange(0, 1), C.c_int), opt(irange(0, 1), C.c_int))
    if status(st, h):
        n = pick(1, 3000, 20000)
        lib.reassignObj_calTimeLength.restype = C.c_int
        T = lib.reassignObj_calTimeLength(h, n)
        assert 0 <= T < 10 ** 7
        F = (1 << (rr if 1 < rr < 31 else 12)) // 2 + 1  # radix2Exp <= 1 means "default" to this constructor
        if T * F <= 1 << 22:
            lib.reassignObj_setResultType(h, pick(0, 1))
            x = f32(n)
            a, b, c, d = (np.zeros(max(T, 1) * F, np.float32) for _ in range(4))
            two = pick(0, 1)
            lib.reassignObj_reassign(h, vp(x), n, vp(a), vp(b), vp(c) if two else None, vp(d) if two else None)
        lib.reassignObj_free(h)
    h = C.c_void_p()
    sn, sr2 = pick(0, 1, 84, 100000), pick(0, 1, 12, 20, 31)
    st = lib.synsqObj_new(C.byref(h), sn, sr2, opt(pick(0, 32000), C.c_int), opt(irange(-1, 3), C.c_int),
                          opt(pick(-1.0, 0.0, 0.001), C.c_float))
    if status(st, h):
        L = 1 << sr2
        if L * sn <= 1 << 20:
            fre = np.sort(np.abs(f32(sn)) * 1e4 + 20).astype(np.float32)
            a, b, c, d = (f32(sn * L) for _ in range(4))
            lib.synsqObj_synsq(h, vp(fre), irange(0, 8), vp(a), vp(b), vp(c), vp(d))
        lib.synsqObj_free(h)


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    for _ in range(rounds):
        fuzz_bft()
        fuzz_cqt()
        fuzz_cwt()
        fuzz_stft()
        fuzz_misc()
    print(f"constructed {counts['ok']}, refused {counts['refused']}")
    assert counts["ok"] > rounds and counts["refused"] > rounds
    print("OK")


if __name__ == "__main__":
    main()
