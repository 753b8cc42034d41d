#!/usr/bin/env python3
"""The CQT host code against the reference's golden vectors, with tests/hoststub/cqt_functional.c standing in for the
kernels (double-precision loops that do what afx_device.h says each launcher does).  Environment switches select the
launch path whose glue is exercised (default per-octave f16 arguments, AFX_CQT_F32, AFX_CQT_CHUNK,
AFX_NO_FUSED for the spectral-kernel arguments).  Raw ctypes, no torch; AFX_LIB = the library built by
tests/test_hoststub.py.  Prints one line per comparison and OK at the end."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import cases  # noqa: E402

lib = C.CDLL(os.environ["AFX_LIB"])
vp, fp = C.c_void_p, C.POINTER(C.c_float)
lib.cqtObj_calTimeLength.restype = C.c_int


def rel(got, want):
    got, want = np.asarray(got, np.complex128), np.asarray(want, np.complex128)
    peak = np.abs(got - want).max() / max(np.abs(want).max(), 1e-300)
    l2 = np.linalg.norm((got - want).ravel()) / max(np.linalg.norm(want.ravel()), 1e-300)
    return max(peak, l2)


def check(what, got, want, tol):
    e = rel(got, want)
    print(f"{what}: {e:.2e} (bar {tol:.0e})")
    assert np.all(np.isfinite(got)) and e <= tol, what


def p(a):
    return a.ctypes.data_as(fp)


def run(name, gold):
    c = cases.CQT_CASES[name]
    x = cases.make_input(c["x"], c["samplate"]).astype(np.float32)
    n, num = len(x), c["num"]
    h = vp()
    st = lib.cqtObj_newWith(C.byref(h), num, C.byref(C.c_int(c["samplate"])), C.byref(C.c_float(c["min_fre"])),
                            C.byref(C.c_int(c["bin_per_octave"])), None, None, None, C.byref(C.c_int(c["window_type"])),
                            C.byref(C.c_int(c["slide_length"])) if "slide_length" in c else None, None,
                            C.byref(C.c_int(c["normal_type"])), C.byref(C.c_int(c["is_scale"])))
    assert st == 0, st
    T = lib.cqtObj_calTimeLength(h, n)
    want = gold[f"{name}/re"] + 1j * gold[f"{name}/im"]
    assert want.shape == (T, num)
    # ---- one clip through the host-pointer call
    re, im = np.zeros((T, num), np.float32), np.zeros((T, num), np.float32)
    lib.cqtObj_cqt(h, p(x), n, p(re), p(im))
    check(f"{name} cqtObj_cqt", re + 1j * im, want, 1e-5)
    # ---- three clips (the clip, the clip halved, the clip negated) with a row stride that is not a multiple of
    #      four, CQT and chroma in one call; the passes (AFX_CQT_CHUNK) must put every clip where it belongs
    stride, batch = n + 5, 3
    xb = np.zeros(batch * stride, np.float32)
    gains = (1.0, 0.5, -1.0)
    for b, g in enumerate(gains):
        xb[b * stride:b * stride + n] = g * x
    stream = (C.c_char * 8)()
    sp = C.cast(stream, vp)
    reb, imb = np.zeros((batch, T, num), np.float32), np.zeros((batch, T, num), np.float32)
    assert lib.cqtObj_cqtBatchDevice(h, p(xb), batch, n, C.c_longlong(stride), p(reb), p(imb), sp) == 0
    for b, g in enumerate(gains):
        check(f"{name} cqtBatchDevice clip {b}", reb[b] + 1j * imb[b], g * want, 1e-5)
    if c["bin_per_octave"] == 12:
        for cname, (cn, dt, nt) in cases.CQT_CHROMA.items():
            ch = np.zeros((batch, T, cn), np.float32)
            reb[:], imb[:] = 0, 0
            st = lib.cqtObj_cqtChromaBatchDevice(h, p(xb), batch, n, C.c_longlong(stride), p(reb), p(imb), C.byref(C.c_int(cn)),
                                                 C.byref(C.c_int(dt)), C.byref(C.c_int(nt)), p(ch), sp)
            assert st == 0, st
            tol = 5e-5 if cname == "six_min" else 1e-5  # divides by the frame MINIMUM: ill-conditioned (tests/test_oracle.py)
            g = gold[f"{name}/chroma_{cname}"]
            for b, gain in enumerate(gains):
                # normalised rows do not see the gain; un-normalised ones scale with |gain| (mag) or gain^2 (power)
                scale = 1.0 if nt != 0 else (abs(gain) if dt == 1 else gain * gain)
                check(f"{name} chroma {cname} clip {b}", ch[b], scale * g, tol)
            check(f"{name} cqt beside chroma {cname}", reb[2] + 1j * imb[2], -want, 1e-5)
    lib.cqtObj_free(h)
    n = (C.c_int * 4).in_dll(lib, "afx_functional_launches")
    print(f"{name} launches: octave_f16 {n[0]} octave_f32 {n[1]} chroma {n[3]}")
    for i in range(4):
        n[i] = 0


def main():
    gold = np.load(os.path.join(ROOT, "tests", "golden", "cqt.npz"))
    for name in sys.argv[1:]:
        run(name, gold)
    print("OK")


if __name__ == "__main__":
    main()
