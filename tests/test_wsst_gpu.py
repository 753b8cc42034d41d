"""GPU parity of the wavelet synchrosqueezed transform.  The target band of a coefficient is a
ROUNDED function of a float32 ratio, so an implementation whose CWT is not bit-identical to the
reference's can place a coefficient whose coordinate sits on a rounding boundary in the neighbouring
band.  The criterion (tests/test_wsst_host.py::explained, pinned there against the reference
itself): every cell must agree to 1e-5 of the peak EXCEPT for the summed magnitude of the
coefficients in its column whose coordinate lies within the float32 uncertainty of a boundary;
that undetermined mass must stay below 5 % of the total, and in practice only a handful of cells
differ at all."""
import os

import numpy as np
import pytest

import audioflux_amd as af
from oracle import ref, restate
from tests import cases
from tests.conftest import assert_parity
from tests.test_wsst_host import explained

pytestmark = pytest.mark.gpu


def make(c):
    return af.WSST(num=c["num"], radix2_exp=c["radix2_exp"], samplate=c["samplate"], low_fre=c.get("low_fre"),
                   high_fre=c.get("high_fre"), wavelet_type=af.WaveletContinueType(c["wavelet_type"]),
                   scale_type=af.SpectralFilterBankScaleType(c["scale_type"]), thresh=c.get("thresh", 0.001),
                   is_padding=bool(c["is_padding"]))


def coordinates(c, x, fre):
    """W, W' from the library's own CWT object (parity-checked separately) -> float64 coordinates"""
    o = af.CWT(num=c["num"], radix2_exp=c["radix2_exp"], samplate=c["samplate"], low_fre=c.get("low_fre"),
               high_fre=c.get("high_fre"), wavelet_type=af.WaveletContinueType(c["wavelet_type"]),
               scale_type=af.SpectralFilterBankScaleType(c["scale_type"]), is_padding=bool(c["is_padding"]))
    o.enable_det(True)
    W, Wd = o.cwt(x)[::-1], o.cwt_det(x)[::-1]   # back to the C row order
    return W, restate.wsst_coordinates(W, Wd, fre, c["samplate"], cases.WSST_SCALE_NAME[c["scale_type"]])


@pytest.mark.parametrize("name", list(cases.WSST_CASES))
def test_wsst_matches_golden(name, golden_dir):
    gold = np.load(os.path.join(golden_dir, "wsst.npz"))
    c = cases.WSST_CASES[name]
    o = make(c)
    assert np.array_equal(o.get_fre_band_arr(), gold[f"{name}/fre"])
    x = cases.make_input((c["x"][0], c["x"][1], 1 << c["radix2_exp"]), c["samplate"])
    s, w = o.wsst_raw(x)
    W, v = coordinates(c, x, gold[f"{name}/fre"])
    assert_parity(w, W, 1e-6, name + " cwt output")   # second output = the CWT itself
    st = cases.cwt_stride(c)
    n_diff = explained(s[:, ::st], gold[f"{name}/s"], W[:, ::st], v[:, ::st], c.get("thresh", 0.001), name)
    assert n_diff < 0.01 * gold[f"{name}/s"].size


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")
def test_wsst_full_resolution_against_compiled_reference():
    c = dict(num=84, radix2_exp=13, samplate=32000, low_fre=32.703, wavelet_type=1, scale_type=5, is_padding=1)
    x = cases.mix(410, 1 << 13, 32000)
    r = ref.RefWSST(84, 13, samplate=32000, low_fre=32.703, wavelet_type=1, scale_type=5, is_padding=1)
    want, _ = r.wsst(x)
    o = make(c)
    s, _ = o.wsst_raw(x)
    W, v = coordinates(c, x, r.fre_band())
    n_diff = explained(s, want, W, v, 0.001, "full resolution")
    assert n_diff < 0.002 * want.size
    # energy is moved, never created: the squeezed mass equals the mass of the accepted coefficients
    ok = np.isfinite(v) & (np.floor(v + 0.5) >= 0) & (np.floor(v + 0.5) < 84) & (np.abs(W) > 0.001)
    assert abs(s.sum() - W[ok].sum()) <= 1e-3 * np.abs(W[ok]).sum()


def test_wsst_at_the_cwt_headline_geometry_against_compiled_reference():
    """84 morlet scales, 2^16-sample chunk, reflect padded (L = 2^17, BASELINE cfg 4's CWT object): the plain and the
    derivative transform both run their 36 short-kernel scales in the time domain (afx_cwt_td.hip); the squeezed
    tensor against the compiled reference under the index-flip accounting of the smaller cases."""
    c = dict(num=84, radix2_exp=16, samplate=44100, low_fre=32.703, wavelet_type=1, scale_type=5, is_padding=1)
    x = cases.mix(411, 1 << 16, 44100)
    r = ref.RefWSST(84, 16, samplate=44100, low_fre=32.703, wavelet_type=1, scale_type=5, is_padding=1)
    want, _ = r.wsst(x)
    o = make(c)
    s, _ = o.wsst_raw(x)
    W, v = coordinates(c, x, r.fre_band())
    n_diff = explained(s, want, W, v, 0.001, "cfg 4 geometry")
    assert n_diff < 0.002 * want.size


def test_accumulate_semantics_and_device_batch():
    import torch
    c = cases.WSST_CASES["morlet_octave48"]
    o = make(c)
    n = 1 << c["radix2_exp"]
    xs = np.stack([cases.mix(420 + i, n, c["samplate"]) for i in range(3)])
    host = [o.wsst_raw(x) for x in xs]
    sre, sim, wre, wim = o.wsst_device(torch.from_numpy(xs).cuda(), with_cwt=True)
    torch.cuda.synchronize()
    for i in range(3):
        assert np.array_equal(sre[i].cpu().numpy(), host[i][0].real) and np.array_equal(sim[i].cpu().numpy(), host[i][0].imag)
        assert np.array_equal(wre[i].cpu().numpy(), host[i][1].real)
    # the C entry ADDS onto the caller's arrays (wsst_algorithm.c:332-333)
    import ctypes as C
    fp = C.POINTER(C.c_float)
    fn = o._lib.wsstObj_wsst
    fn.restype, fn.argtypes = None, [C.c_void_p] + [fp] * 5
    a = np.full((c["num"], n), 2.0, np.float32)
    b = np.full((c["num"], n), -1.0, np.float32)
    x0 = np.ascontiguousarray(xs[0])
    fn(o._obj, x0.ctypes.data_as(fp), a.ctypes.data_as(fp), b.ctypes.data_as(fp), None, None)
    assert np.allclose(a - 2.0, host[0][0].real, atol=1e-6) and np.allclose(b + 1.0, host[0][0].imag, atol=1e-6)
    sw, cw = o.wsst(xs)   # wrapper: leading axes are clips, CWT flipped to ascending frequency
    assert sw.shape == (3, c["num"], n) and np.array_equal(cw[1], host[1][1][::-1])
