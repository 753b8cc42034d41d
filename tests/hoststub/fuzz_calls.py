#!/usr/bin/env python3
"""Compute-call arguments at their edges (zero / one-sample / shorter-than-a-frame inputs, batch 0, a clip stride
shorter than the clip, NULL outputs) on ordinary objects, through raw ctypes against the sanitizer build of the host
code.  Every call must return (a status <= 0 where the API has one) without a crash or a sanitizer report.  (Clip strides
shorter than the clip are legal for the framed transforms -- overlapping windows of one long signal.)"""
import ctypes as C
import os
import sys

import numpy as np

lib = C.CDLL(os.environ["AFX_LIB"])
rng = np.random.default_rng(int(os.environ.get("AFX_FUZZ_SEED", "1")))
vp = C.c_void_p


def pick(*vals):
    return vals[rng.integers(0, len(vals))]


def arr(n):
    return np.zeros(max(int(n), 1), np.float32)


def ptr(a, null_ok=True):
    if null_ok and rng.integers(0, 8) == 0:
        return None
    return a.ctypes.data_as(vp)


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    stream = (C.c_char * 8)()
    sp = C.cast(stream, vp)
    lib.bftObj_calTimeLength.restype = C.c_int
    lib.stftObj_calTimeLength.restype = C.c_int
    lib.cqtObj_calTimeLength.restype = C.c_int
    refused = done = 0
    for _ in range(rounds):
        r2 = pick(8, 10, 11)
        N = 1 << r2
        num = pick(13, 40, 128)
        hop = pick(1, N // 4, N, N + 3, 3 * N)
        # ---- BFT
        h = vp()
        assert lib.bftObj_new(C.byref(h), num, r2, C.byref(C.c_int(16000)), None, None, None, None, C.byref(C.c_int(hop)), None, None,
                              None, None, None, C.byref(C.c_int(pick(0, 1)))) == 0
        lib.bftObj_setResultType(h, pick(0, 1))
        n = pick(0, 1, 5, N - 1, N, N + 1, 3 * N + 7)
        T = max(lib.bftObj_calTimeLength(h, n), 0)
        x, re, im = arr(n), arr(T * num), arr(T * num)
        lib.bftObj_bft(h, ptr(x), n, ptr(re), ptr(im))
        batch, stride = pick(0, 1, 3), pick(n - 1, n, n + 7)
        xb, reb, imb = arr(batch * max(stride, n)), arr(batch * T * num), arr(batch * T * num)
        st = lib.bftObj_bftBatchDevice(h, ptr(xb), batch, n, C.c_longlong(stride), ptr(reb), ptr(imb, False), sp)
        assert st <= 0
        refused += st < 0
        done += st == 0
        lib.bftObj_free(h)
        # ---- STFT with padding / streaming switches
        s = vp()
        assert lib.stftObj_new(C.byref(s), r2, None, C.byref(C.c_int(hop)), C.byref(C.c_int(pick(0, 1)))) == 0
        lib.stftObj_enablePadding(s, pick(0, 1))
        lib.stftObj_setPadding(s, C.byref(C.c_int(pick(0, 1, 2))), C.byref(C.c_int(pick(0, 1, 2))), None, None)
        for _call in range(6):  # a stream: the kept tail (also the "negative" tail of hop > fftLength) carries over
            n = pick(0, 1, 7, hop, N - 1, N, N + hop - 1, 2 * N + 5, 4 * N + 1)
            T = max(lib.stftObj_calTimeLength(s, n), 0)
            x, re, im = arr(n), arr(T * N), arr(T * N)
            lib.stftObj_stft(s, ptr(x), n, ptr(re), ptr(im))
        lib.stftObj_free(s)
        # ---- CQT
        q = vp()
        assert lib.cqtObj_new(C.byref(q), 84, 44100, C.c_float(32.703), None) == 0
        n = pick(0, 1, 100, 511, 513, 5000)
        T = max(lib.cqtObj_calTimeLength(q, n), 0)
        x, re, im = arr(n), arr(T * 84), arr(T * 84)
        lib.cqtObj_cqt(q, ptr(x), n, ptr(re), ptr(im))
        batch, stride = pick(0, 1, 2), pick(n - 1, n, n + 3)
        xb, reb, imb = arr(batch * max(stride, n)), arr(batch * T * 84), arr(batch * T * 84)
        st = lib.cqtObj_cqtBatchDevice(q, ptr(xb), batch, n, C.c_longlong(stride), ptr(reb), ptr(imb), sp)
        assert st <= 0
        if batch > 0 and (n <= 0 or stride < n):  # the CQT calls refuse clips that overlap or are empty
            assert st < 0, (batch, n, stride, st)
        ch = arr(batch * T * 12)
        st = lib.cqtObj_cqtChromaBatchDevice(q, ptr(xb), batch, n, C.c_longlong(stride), ptr(reb), ptr(imb), None, None, None,
                                             ptr(ch), sp)
        assert st <= 0
        st = lib.cqtObj_chromaBatchDevice(q, None, None, None, ptr(reb), ptr(imb), C.c_longlong(batch * T), ptr(ch), sp)
        assert st <= 0
        lib.cqtObj_free(q)
        # ---- XXCC
        c = vp()
        assert lib.xxccObj_new(C.byref(c), num) == 0
        rows, cc = pick(0, 1, 7), pick(0, 1, 13, num, num + 1)
        m, out = arr(rows * num), arr(rows * max(cc, 1))
        st = lib.xxccObj_xxccDevice(c, ptr(m), C.c_longlong(rows), cc, None, ptr(out), sp)
        assert st <= 0
        lib.xxccObj_setTimeLength(c, rows)
        lib.xxccObj_xxcc(c, ptr(m), cc, None, ptr(out))
        lib.xxccObj_free(c)
        # ---- cepstrogram: cepNum at and beyond its range, inputs shorter than a frame
        ce = vp()
        assert lib.cepstrogramObj_new(C.byref(ce), r2, None, C.byref(C.c_int(hop))) == 0
        lib.cepstrogramObj_calTimeLength.restype = C.c_int
        n = pick(0, 1, N - 1, N, 3 * N + 1)
        T = max(lib.cepstrogramObj_calTimeLength(ce, n), 0)
        F = N // 2 + 1
        x, o1, o2, o3 = arr(n), arr(T * F), arr(T * F), arr(T * F)
        lib.cepstrogramObj_cepstrogram(ce, pick(-1, 0, 1, 4, F, F + 5), ptr(x), n, ptr(o1), ptr(o2), ptr(o3))
        lib.cepstrogramObj_free(ce)
        # ---- mel spectrogram object as a stream of chunks
        sg = vp()
        cont = C.c_int(pick(0, 1))
        assert lib.spectrogramObj_newMel(C.byref(sg), 40, 16000, r2, C.byref(cont)) == 0
        lib.spectrogramObj_calTimeLength.restype = C.c_int
        for _call in range(4):
            n = pick(0, 1, N // 4, N - 1, N, 2 * N + 3)
            T = max(lib.spectrogramObj_calTimeLength(sg, n), 0)
            x, m, ph = arr(n), arr(T * 40), arr(T * 40)
            lib.spectrogramObj_spectrogram(sg, ptr(x), n, ptr(m), ptr(ph))
        lib.spectrogramObj_free(sg)
        # ---- reassignment, every reassign type, inputs around one frame
        ra = vp()
        assert lib.reassignObj_new(C.byref(ra), r2, None, None, C.byref(C.c_int(min(hop, N))), C.byref(C.c_int(pick(0, 1, 2, 3))),
                                   None, C.byref(C.c_int(pick(0, 1))), None) == 0
        lib.reassignObj_calTimeLength.restype = C.c_int
        lib.reassignObj_setResultType(ra, pick(0, 1))
        n = pick(0, 1, N - 1, N, N + 1, 2 * N + 9)
        T = max(lib.reassignObj_calTimeLength(ra, n), 0)
        x = arr(n)
        b1, b2, b3, b4 = (arr(T * F) for _ in range(4))
        lib.reassignObj_reassign(ra, ptr(x), n, ptr(b1), ptr(b2), ptr(b3), ptr(b4))
        lib.reassignObj_free(ra)
        # ---- CWT at the in-LDS size with NULL outputs now and then
        cw = vp()
        cr2 = pick(6, 10, 12)
        assert lib.cwtObj_new(C.byref(cw), 24, cr2, None, None, None, None, None, None, None, None, C.byref(C.c_int(pick(0, 1)))) == 0
        x, a1, a2 = arr(1 << cr2), arr(24 << cr2), arr(24 << cr2)
        lib.cwtObj_cwt(cw, ptr(x), ptr(a1), ptr(a2))
        lib.cwtObj_free(cw)
    print(f"calls accepted {done}, refused {refused}")
    print("OK")


if __name__ == "__main__":
    main()
