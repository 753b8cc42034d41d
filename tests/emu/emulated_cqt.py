#!/usr/bin/env python3
"""The DEVICE code of the f16 CQT kernels on the CPU (tests/emu): golden CQT / chroma vectors through the C host code,
the real launchers and the CQT kernels compiled for the host and run one thread per lane.
AFX_LIB = the library tests/test_emulated_kernels.py builds; AFX_CQT_F32=1 selects the float32 matrix-core octave kernels,
AFX_CQT_PYRAMID=0 the per-octave launches instead of the one-launch ladder (k_cqt_pyramid).
Prints one line per comparison, the launches seen, and OK."""
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
lib.afx_emulated_launches.restype = C.c_int
lib.afx_emulated_launches.argtypes = [C.c_char_p]


def rel(got, want):
    got, want = np.asarray(got, np.complex128), np.asarray(want, np.complex128)
    peak = np.abs(got - want).max() / max(np.abs(want).max(), 1e-300)
    l2 = np.linalg.norm((got - want).ravel()) / max(np.linalg.norm(want.ravel()), 1e-300)
    return max(peak, l2)


def check(what, got, want, tol):
    e = rel(got, want)
    print(f"{what}: {e:.2e} (bar {tol:.0e})", flush=True)
    assert np.all(np.isfinite(got)) and e <= tol, what


def p(a):
    return a.ctypes.data_as(fp)


def stream_pieces():
    """isContinue = 1: pieces of a signal through the emulated kernels against the float64 restatement of the
    reference's tail rule (oracle/restate.py: CqtStream) -- the rightPad framing of every octave kernel"""
    from oracle import restate
    hop = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    x = (0.1 * np.random.default_rng(9).standard_normal(9000)).astype(np.float32)
    h = vp()
    st = lib.cqtObj_newWith(C.byref(h), 84, C.byref(C.c_int(32000)), None, None, None, None, None, None,
                            C.byref(C.c_int(hop)) if hop else None, C.byref(C.c_int(1)), None, None)
    assert st == 0, st
    want = restate.CqtStream(num=84, samplate=32000, normal="none", hop=hop or None)
    pos = 0
    for n in (4000, 300, 3000):
        seg = np.ascontiguousarray(x[pos:pos + n])
        pos += n
        T = lib.cqtObj_calTimeLength(h, n)
        w = want.cqt(seg)
        assert T == w.shape[0], (T, w.shape)
        re, im = np.zeros((T, 84), np.float32), np.zeros((T, 84), np.float32)
        lib.cqtObj_cqt(h, p(seg), n, p(re), p(im))
        if T:
            err = np.abs((re + 1j * im) - w).max(axis=0) / np.abs(w).max()
            print("   per-octave worst error:", [f"{err[12 * o:12 * o + 12].max():.1e}" for o in range(7)], flush=True)
            check(f"stream +{n} samples -> {T} frames", re + 1j * im, w, 1e-5)
    lib.cqtObj_free(h)
    print("OK")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "stream":
        return stream_pieces()
    name = sys.argv[1] if len(sys.argv) > 1 else "c84_32k_area"
    gold = np.load(os.path.join(ROOT, "tests", "golden", "cqt.npz"))
    c = cases.CQT_CASES[name]
    x = cases.make_input(c["x"], c["samplate"]).astype(np.float32)
    n, num = len(x), c["num"]
    h = vp()
    st = lib.cqtObj_newWith(C.byref(h), num, C.byref(C.c_int(c["samplate"])), C.byref(C.c_float(c["min_fre"])),
                            C.byref(C.c_int(c["bin_per_octave"])), None, None, None, C.byref(C.c_int(c["window_type"])), None, None,
                            C.byref(C.c_int(c["normal_type"])), C.byref(C.c_int(c["is_scale"])))
    assert st == 0, st
    T = lib.cqtObj_calTimeLength(h, n)
    want = gold[f"{name}/re"] + 1j * gold[f"{name}/im"]
    re, im = np.zeros((T, num), np.float32), np.zeros((T, num), np.float32)
    lib.cqtObj_cqt(h, p(x), n, p(re), p(im))
    check(f"{name} cqtObj_cqt", re + 1j * im, want, 1e-5)
    # two clips (the clip, the clip negated and halved) with an odd row stride, CQT + chroma in one call
    stride, batch, gains = n + 5, 2, (1.0, -0.5)
    xb = np.zeros(batch * stride, np.float32)
    for b, g in enumerate(gains):
        xb[b * stride:b * stride + n] = g * x
    stream = (C.c_char * 8)()
    sp = C.cast(stream, vp)
    reb, imb = np.zeros((batch, T, num), np.float32), np.zeros((batch, T, num), np.float32)
    for cname in sys.argv[2:] or ["power_max"]:
        cn, dt, nt = cases.CQT_CHROMA[cname]
        ch = np.zeros((batch, T, cn), np.float32)
        st = lib.cqtObj_cqtChromaBatchDevice(h, p(xb), batch, n, C.c_longlong(stride), p(reb), p(imb), C.byref(C.c_int(cn)),
                                             C.byref(C.c_int(dt)), C.byref(C.c_int(nt)), p(ch), sp)
        assert st == 0, st
        g = gold[f"{name}/chroma_{cname}"]
        for b, gain in enumerate(gains):
            check(f"{name} clip {b} beside chroma {cname}", reb[b] + 1j * imb[b], gain * want, 1e-5)
            scale = 1.0 if nt != 0 else (abs(gain) if dt == 1 else gain * gain)
            check(f"{name} chroma {cname} clip {b}", ch[b], scale * g, 5e-5 if cname == "six_min" else 1e-5)
    lib.cqtObj_free(h)
    try:  # tests/hoststub/cqt_functional.c: [.., f32 octave, .., chroma] (absent when every CQT kernel is emulated)
        fn = list((C.c_int * 4).in_dll(lib, "afx_functional_launches"))
    except ValueError:
        fn = [0, 0, 0, 0]
    print("launches: emulated octave_f16 %d; contract-level octave_f32 %d, chroma %d" % (
        lib.afx_emulated_launches(b"k_cqt_octave_f16"), fn[1], fn[3]))
    print("          emulated pyramid %d" % lib.afx_emulated_launches(b"k_cqt_pyramid"))
    print("          emulated decimate %d, chroma %d, chroma_scan %d, octave_mfma (f32) %d" % (
        lib.afx_emulated_launches(b"k_cqt_decimate"), lib.afx_emulated_launches(b"k_cqt_chroma") - lib.afx_emulated_launches(b"k_cqt_chroma_scan"),
        lib.afx_emulated_launches(b"k_cqt_chroma_scan"), lib.afx_emulated_launches(b"k_cqt_octave_mfma")))
    print("OK")


if __name__ == "__main__":
    main()
