"""GPU parity of the continuous wavelet transform object against the reference's
golden vectors (all eight wavelet families, padded and unpadded, 2^3 .. 2^16 samples)."""
import os

import numpy as np
import pytest

import audioflux_amd as af
from tests import cases
from tests.conftest import assert_parity

pytestmark = pytest.mark.gpu
TOL = 1e-5


def make(c):
    return af.CWT(num=c["num"], radix2_exp=c["radix2_exp"], samplate=c["samplate"],
                  low_fre=c.get("low_fre"), high_fre=c.get("high_fre"),
                  bin_per_octave=c.get("bin_per_octave", 12),
                  wavelet_type=af.WaveletContinueType(c["wavelet_type"]),
                  scale_type=af.SpectralFilterBankScaleType(c["scale_type"]),
                  gamma=c.get("gamma"), beta=c.get("beta"), is_padding=bool(c["is_padding"]))


@pytest.mark.parametrize("name", list(cases.CWT_CASES))
def test_cwt_matches_golden(name, golden_dir):
    gold = np.load(os.path.join(golden_dir, "cwt.npz"))
    c = cases.CWT_CASES[name]
    o = make(c)
    assert np.array_equal(o.get_fre_band_arr(), gold[f"{name}/fre"])
    assert np.array_equal(o.get_bin_band_arr(), gold[f"{name}/bin"])
    x = cases.make_input((c["x"][0], c["x"][1], 1 << c["radix2_exp"]), c["samplate"])
    st = cases.cwt_stride(c)
    w = o.cwt(x)[::-1, ::st]  # wrapper returns ascending frequency; the C layout is descending
    assert_parity(w, gold[f"{name}/re"] + 1j * gold[f"{name}/im"], TOL, name)
    if f"{name}/det_re" in gold.files:
        o.enable_det(True)
        d = o.cwt_det(x)[::-1, ::st]
        assert_parity(d, gold[f"{name}/det_re"] + 1j * gold[f"{name}/det_im"], TOL, name + "/det")


def test_cwt_linearity_and_reuse():
    o = af.CWT(num=30, radix2_exp=11, samplate=16000, wavelet_type=af.WaveletContinueType.MORLET)
    a, b = cases.noise(95, 2048), cases.noise(96, 2048)
    wa, wb, ws = o.cwt(a), o.cwt(b), o.cwt((a - 3 * b).astype(np.float32))
    assert_parity(ws, wa - 3 * wb, 2e-6, "linearity")
    # short input is zero-padded by the wrapper, long input truncated (utils/util.py:98-111)
    assert o.cwt(a[:1000]).shape == (30, 2048) and o.cwt(np.concatenate([a, b])).shape == (30, 2048)


@pytest.mark.parametrize("max_r", [2, 4, 8, 16, 20, 24, 32])
def test_cwt_narrow_band_scales_equal_two_pass(max_r, monkeypatch):
    """L = 2^17 (BASELINE cfg 4): scales whose wavelet occupies <= max_r rows of the transposed
    spectrum skip the row pass (k_cwt_inv_cols256_nb); every scale must agree with the two-pass
    result, for the transform and for its time derivative, at 1e-5 of the scale's own peak."""
    import torch

    def build(mr):
        monkeypatch.setenv("AFX_CWT_NARROW_MAX", str(mr))
        o = af.CWT(num=84, radix2_exp=16, samplate=44100, low_fre=32.703, bin_per_octave=12,
                   wavelet_type=af.WaveletContinueType.MORLET,
                   scale_type=af.SpectralFilterBankScaleType.OCTAVE, is_padding=True)
        o.enable_det(True)
        return o

    two_pass, narrow = build(0), build(max_r)
    g = torch.Generator(device="cuda").manual_seed(77)
    x = 0.1 * torch.randn((3, 1 << 16), device="cuda", generator=g)
    x[1] += torch.sin(torch.arange(1 << 16, device="cuda") * (2 * np.pi * 55.0 / 44100))  # energy in a narrow scale
    for det in (False, True):
        r0, i0 = two_pass.cwt_device(x, det=det)
        r1, i1 = narrow.cwt_device(x, det=det)
        torch.cuda.synchronize()
        w0, w1 = torch.complex(r0, i0), torch.complex(r1, i1)
        peak = w0.abs().amax(dim=2, keepdim=True)
        err = ((w1 - w0).abs() / peak).amax(dim=2)  # [chunk, scale]
        assert float(err.max()) <= 1e-5, (det, max_r, err.max(dim=0).values.cpu().numpy())
    # the one-chunk host entry point takes the same plan
    h0 = two_pass.cwt(x[1].cpu().numpy())
    h1 = narrow.cwt(x[1].cpu().numpy())
    assert_parity(h1, h0, TOL, f"host entry, max_r {max_r}")


def test_more_short_kernel_scales_than_the_time_domain_launch_takes():
    """a bank of 36 bins per octave: 150 scales between 500 Hz and 9 kHz at 32 kHz all have short time kernels -- more
    than the 96 (2 x 48 pairs) one time-domain launch takes.  The plan keeps the 96 shortest there and leaves the rest
    on the FFT path (round 3 planned all of them and every call of the object failed); the result is the reference's."""
    from oracle import ref
    if not ref.available():
        pytest.skip("compiled reference not built")
    rng = np.random.default_rng(77)
    x = (0.1 * rng.standard_normal(1 << 14)).astype(np.float32)
    kw = dict(num=150, radix2_exp=14, samplate=32000, low_fre=500.0, bin_per_octave=36)
    o = af.CWT(wavelet_type=af.WaveletContinueType.MORLET, scale_type=af.SpectralFilterBankScaleType.OCTAVE, is_padding=True, **kw)
    got = o.cwt(x)[::-1]
    r = ref.RefCWT(wavelet_type=1, scale_type=5, is_padding=1, **kw)
    rre, rim = r.cwt(x)
    assert_parity(got, rre + 1j * rim, TOL, "150 short-kernel scales")
    got2 = o.cwt(x)[::-1]  # (the object stays usable)
    assert np.array_equal(got, got2)


@pytest.mark.parametrize("td_det", ["1", "0"])
def test_derivative_transform_time_domain_scales_against_reference(td_det, monkeypatch):
    """BASELINE cfg 4's object with cwtObj_enableDet: the 36 short-kernel scales of the DERIVATIVE transform run in
    the time domain too (kernels IFFT(j w psi), cwt_algorithm.c:485-528) -- every scale against the compiled reference
    at 1e-5 of the scale's own peak, from the batched device entry and from the one-chunk host entry.  AFX_CWT_TD_DET=0
    (read by cwtObj_enableDet) keeps those scales on the two-pass path: the same bar."""
    import torch
    monkeypatch.setenv("AFX_CWT_TD_DET", td_det)
    from oracle import ref
    if not ref.available():
        pytest.skip("compiled reference not built")
    kw = dict(num=84, radix2_exp=16, samplate=44100, low_fre=32.703, bin_per_octave=12)
    o = af.CWT(wavelet_type=af.WaveletContinueType.MORLET, scale_type=af.SpectralFilterBankScaleType.OCTAVE, is_padding=True, **kw)
    o.enable_det(True)
    rng = np.random.default_rng(4242)
    n = 1 << 16
    x = (0.1 * rng.standard_normal((2, n))).astype(np.float32)
    x[1] += np.sin(np.arange(n) * (2 * np.pi * 3520.0 / 44100)).astype(np.float32)  # energy in a short-kernel scale
    r = ref.RefCWT(wavelet_type=1, scale_type=5, is_padding=1, **kw)
    re, im = o.cwt_device(torch.from_numpy(x).cuda(), det=True)
    torch.cuda.synchronize()
    got = (re.cpu().numpy() + 1j * im.cpu().numpy())
    worst = 0.0
    for c in range(2):
        rre, rim = r.cwt(x[c], det=True)
        want = rre + 1j * rim  # (the device entry and the reference: C order, row 0 = highest frequency)
        peak = np.abs(want).max(axis=1, keepdims=True)
        err = (np.abs(got[c] - want) / peak).max(axis=1)
        worst = max(worst, float(err.max()))
        assert err.max() <= 1e-5, (c, np.argmax(err), err.max())
    from tests.conftest import parity_log
    parity_log("cwt/derivative transform, per scale (36 short-kernel scales, AFX_CWT_TD_DET=%s)" % td_det, worst, 1e-5, kind="per-scale peak")
    assert_parity(o.cwt_det(x[1])[::-1], (lambda a, b: a + 1j * b)(*r.cwt(x[1], det=True)), TOL, "host entry, derivative")
