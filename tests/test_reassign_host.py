"""CPU-only: the three analysis windows of the reassignment object bit for bit against the reference's
own helper functions, the float64 restatement pinned against the golden vectors under the
boundary-aware criterion the GPU test uses, the fixtures reproduced by the compiled reference."""
import ctypes as C
import os

import numpy as np
import pytest

import audioflux_amd as af
from oracle import ref, restate
from tests import cases

fp = C.POINTER(C.c_float)


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "reassign.npz"))


def restated(c, x):
    """float64 STFTs with the three windows -> (S_h, vt, vf)"""
    n, hop, sr = 1 << c["radix2_exp"], c["slide_length"], c["samplate"]
    h, dh, th = restate.reassign_windows(restate.fft_window(c["window_type"], n), n)
    if c.get("is_padding"):
        S = [restate.stft_padded(x, n, hop, w)[:, : n // 2 + 1] for w in (h, dh, th)]
    else:
        S = [restate.stft_full(x, n, hop, w)[:, : n // 2 + 1] for w in (h, dh, th)]
    vt, vf = restate.reassign_coordinates(S[0], S[1], S[2], sr, hop, c.get("thresh", 0.001),
                                          cases.RETYPE_NAME[c.get("re_type", 0)])
    return S[0], vt, vf


def explained(got, want, Sh, vt, vf, thresh, what, order=1):
    allow, amb = restate.reassign_allowance(Sh, vt, vf, thresh, order=order)
    scale = np.abs(want).max()
    d = np.abs(np.asarray(got, np.complex128) - want)
    bad = d > allow + 1e-5 * scale
    assert not bad.any(), f"{what}: {int(bad.sum())} cells differ beyond what boundary coefficients explain " \
                          f"(worst {d[bad].max() / scale:.3e})"
    frac = np.abs(Sh[amb]).sum() / np.abs(Sh).sum()
    from tests.conftest import parity_log
    parity_log(f"reassign explained-difference: {what}", float((d > 1e-5 * scale).mean()), 0.005,
               "fraction of cells beyond 1e-5 (all explained by boundary coefficients)",
               {"undetermined_mass": float(frac), "undetermined_mass_bar": 0.01,
                "worst_unexplained_excess": float(np.maximum(d - allow, 0).max() / scale)})
    assert frac < 0.01, \
        f"{what}: {frac:.4f} of the coefficient mass is undetermined -- criterion too loose"
    return int((d > 1e-5 * scale).sum())


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")
def test_compiled_reference_reproduces_golden(gold, tmp_path):
    from tests.golden import make_golden
    here = make_golden.HERE
    make_golden.HERE = str(tmp_path)
    try:
        make_golden.make_reassign()
    finally:
        make_golden.HERE = here
    fresh = np.load(os.path.join(str(tmp_path), "reassign.npz"))
    for k in gold.files:
        assert np.array_equal(fresh[k], gold[k]), k


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("wtype,r", [(1, 10), (2, 9), (3, 8), (8, 6)])
def test_analysis_windows_bit_identical(wtype, r):
    """h from window_calFFTWindow; dh = __vgradient of the wrapped extension (+1 offset); t.h = arange x h
    (reassign_algorithm.c:417-452), all through the reference's exported helpers"""
    R, L = ref.lib(), af.get_lib()
    n = 1 << r
    R.window_calFFTWindow.restype = fp
    R.window_calFFTWindow.argtypes = [C.c_int, C.c_int]
    w = np.ctypeslib.as_array(R.window_calFFTWindow(wtype, n), (n,)).copy()
    ext = np.concatenate([w[-1:], w, w[:1]]).astype(np.float32)
    grad = np.zeros(n + 2, np.float32)
    R.__vgradient.restype = None
    R.__vgradient.argtypes = [fp, C.c_int, C.c_int, fp]
    R.__vgradient(ext.ctypes.data_as(fp), n + 2, 1, grad.ctypes.data_as(fp))
    th = (np.arange(-n // 2, n // 2).astype(np.float32) * w).astype(np.float32)
    got = np.zeros((3, n), np.float32)
    L.afx_reassign_windows.restype = C.c_int
    L.afx_reassign_windows.argtypes = [C.c_int, C.c_int, fp]
    assert L.afx_reassign_windows(wtype, r, got.ctypes.data_as(fp)) == 0
    assert np.array_equal(got[0], w) and np.array_equal(got[1], grad[1:n + 1]) and np.array_equal(got[2], th)


@pytest.mark.parametrize("name", [k for k in cases.REASSIGN_CASES if k != "none_plain_stft"])
def test_restatement_explains_golden(name, gold):
    c = cases.REASSIGN_CASES[name]
    x = cases.make_input(c["x"], c["samplate"])
    Sh, vt, vf = restated(c, x)
    amp = c.get("result_type", 0) == 1
    out = restate.reassign_scatter(Sh, vt, vf, amp, c.get("order", 1))
    want = gold[f"{name}/re"] if amp else gold[f"{name}/re"] + 1j * gold[f"{name}/im"]
    n_diff = explained(out, want, Sh, vt, vf, c.get("thresh", 0.001), name, order=c.get("order", 1))
    assert n_diff < 0.005 * want.size


def test_none_type_is_the_plain_stft(gold):
    c = cases.REASSIGN_CASES["none_plain_stft"]
    x = cases.make_input(c["x"], c["samplate"])
    n = 1 << c["radix2_exp"]
    S = restate.stft(x, n, c["slide_length"], c["window_type"])
    want = gold["none_plain_stft/re"] + 1j * gold["none_plain_stft/im"]
    assert np.abs(S - want).max() <= 1e-5 * np.abs(want).max()


def test_status_codes_without_device():
    lib = af.get_lib()
    lib.reassignObj_free.argtypes = [C.c_void_p]
    lib.reassignObj_free(None)
    f = lib.reassignObj_new
    f.restype = C.c_int
    f.argtypes = [C.POINTER(C.c_void_p), C.c_int] + [C.c_void_p] * 7
    obj = C.c_void_p(None)
    if af.runtime_status() != 0:
        assert f(C.byref(obj), 10, *([None] * 7)) == -2 and not obj   # no device: no CPU fallback
    assert f(C.byref(obj), 20, *([None] * 7)) == -4 and not obj       # 2^20 exceeds the on-chip FFT limit
