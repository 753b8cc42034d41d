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


def run_stream(name):
    """isContinue = 1 (cqt_algorithm.c:345-456): a signal fed in pieces -- shorter than a frame, shorter than a hop,
    long, a few samples -- against the float64 restatement of the reference's tail handling (oracle/restate.py:
    CqtStream, pinned against the compiled reference in tests/test_oracle.py); the batch entry points refuse."""
    from oracle import restate
    name, _, hop = name.partition("@")  # "<case>@<hop>": the case's plan with another hop
    c = cases.CQT_CASES[name]
    sr, num = c["samplate"], c["num"]
    x = (0.1 * np.random.default_rng(41).standard_normal(60000)).astype(np.float32)
    h = vp()
    hop = int(hop) if hop else c.get("slide_length")
    st = lib.cqtObj_newWith(C.byref(h), num, C.byref(C.c_int(sr)), C.byref(C.c_float(c["min_fre"])),
                            C.byref(C.c_int(c["bin_per_octave"])), None, None, None, C.byref(C.c_int(c["window_type"])),
                            C.byref(C.c_int(hop)) if hop else None, C.byref(C.c_int(1)),
                            C.byref(C.c_int(c["normal_type"])), C.byref(C.c_int(c["is_scale"])))
    assert st == 0, st
    normal = {0: "none", 1: "area", 2: "bandwidth"}[c["normal_type"]]
    want = restate.CqtStream(num=num, samplate=sr, min_fre=float(np.float32(c["min_fre"])), bpo=c["bin_per_octave"], window_type=c["window_type"],
                             normal=normal, hop=hop, is_scale=bool(c["is_scale"]))
    pos, frames = 0, 0
    for n in (300, 4000, 100, 130, 9000, 511, 12000, 7, 1, 2000, 20000):
        seg = np.ascontiguousarray(x[pos:pos + n])
        pos += n
        T = lib.cqtObj_calTimeLength(h, n)
        w = want.cqt(seg)
        assert T == w.shape[0], (n, T, w.shape)
        re, im = np.full((max(T, 1), num), 7.0, np.float32), np.full((max(T, 1), num), 7.0, np.float32)
        lib.cqtObj_cqt(h, p(seg), n, p(re), p(im))
        if T:
            check(f"{name} stream +{n} samples -> {T} frames", re[:T] + 1j * im[:T], w, 1e-5)
        else:
            assert (re == 7.0).all() and (im == 7.0).all(), "no frame: the outputs must stay untouched"
        frames += T
    assert frames == (pos - want.n) // want.hop + 1, (frames, pos)  # the pieces add up to the one-shot framing
    stream = (C.c_char * 8)()
    out = np.zeros((1, 4, num), np.float32)
    assert lib.cqtObj_cqtBatchDevice(h, p(x), 1, 4000, C.c_longlong(4000), p(out), p(out), C.cast(stream, vp)) != 0
    lib.cqtObj_free(h)
    n4 = (C.c_int * 4).in_dll(lib, "afx_functional_launches")
    for i in range(4):
        n4[i] = 0


def main():
    gold = np.load(os.path.join(ROOT, "tests", "golden", "cqt.npz"))
    for name in sys.argv[1:]:
        if name.startswith("stream:"):
            run_stream(name[7:])
        else:
            run(name, gold)
    print("OK")


if __name__ == "__main__":
    main()
