"""GPU parity of the pseudo wavelet transform object against the reference's golden vectors
(six scales, slaney / ETSI / window styles, padded and unpadded, 2^5 .. 2^16 samples), its
derivative variant, and the batched device call."""
import os

import numpy as np
import pytest

import audioflux_amd as af
from oracle import ref
from tests import cases
from tests.conftest import assert_parity

pytestmark = pytest.mark.gpu
TOL = 1e-5


def make(c):
    return af.PWT(num=c["num"], radix2_exp=c["radix2_exp"], samplate=c["samplate"], low_fre=c.get("low_fre"),
                  high_fre=c.get("high_fre"), bin_per_octave=c.get("bin_per_octave", 12),
                  scale_type=af.SpectralFilterBankScaleType(c["scale_type"]),
                  style_type=af.SpectralFilterBankStyleType(c["style_type"]),
                  normal_type=af.SpectralFilterBankNormalType(c["normal_type"]), is_padding=bool(c["is_padding"]))


@pytest.mark.parametrize("name", list(cases.PWT_CASES))
def test_pwt_matches_golden(name, golden_dir):
    gold = np.load(os.path.join(golden_dir, "pwt.npz"))
    c = cases.PWT_CASES[name]
    o = make(c)
    assert np.array_equal(o.get_fre_band_arr(), gold[f"{name}/fre"])
    assert np.array_equal(o.get_bin_band_arr(), gold[f"{name}/bin"])
    x = cases.make_input((c["x"][0], c["x"][1], 1 << c["radix2_exp"]), c["samplate"])
    st = cases.cwt_stride(c)
    assert_parity(o.pwt(x)[:, ::st], gold[f"{name}/re"] + 1j * gold[f"{name}/im"], TOL, name)
    if f"{name}/det_re" in gold.files:
        o.enable_det(True)
        assert_parity(o.pwt_det(x)[:, ::st], gold[f"{name}/det_re"] + 1j * gold[f"{name}/det_im"], TOL, name + "/det")


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")
def test_pwt_matches_compiled_reference_fresh_input():
    x = cases.noise(310, 1 << 14)
    for scale, style, pad in ((5, 0, 1), (2, 1, 0), (3, 5, 1)):
        r = ref.RefPWT(48, 14, samplate=32000, low_fre=65.406, high_fre=12000.0, scale_type=scale, style_type=style,
                       normal_type=0, is_padding=pad)
        assert r.status == 0
        re, im = r.pwt(x)
        o = af.PWT(num=48, radix2_exp=14, samplate=32000, low_fre=65.406, high_fre=12000.0,
                   scale_type=af.SpectralFilterBankScaleType(scale), style_type=af.SpectralFilterBankStyleType(style),
                   is_padding=bool(pad))
        assert_parity(o.pwt(x), re + 1j * im, TOL, f"scale{scale} style{style} pad{pad}")


def test_device_batch_and_linearity():
    import torch
    o = af.PWT(num=30, radix2_exp=11, samplate=16000, scale_type=af.SpectralFilterBankScaleType.MEL, low_fre=0.0)
    a, b = cases.noise(311, 2048), cases.noise(312, 2048)
    wa, wb, ws = o.pwt(a), o.pwt(b), o.pwt((a - 3 * b).astype(np.float32))
    assert_parity(ws, wa - 3 * wb, 2e-6, "linearity")
    re, im = o.pwt_device(torch.from_numpy(np.stack([a, b])).cuda())
    torch.cuda.synchronize()
    assert np.array_equal(re[1].cpu().numpy(), wb.real) and np.array_equal(im[0].cpu().numpy(), wa.imag)


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("scale,normal,pad,r2", [(2, 0, 1, 12), (3, 0, 0, 13), (4, 2, 1, 12), (5, 0, 1, 14)])
def test_pwt_gammatone_style_reproduces_the_reference_layout(scale, normal, pad, r2):
    """the gammatone style in the pseudo layout (auditory_filterBank.c:509-582): the reference writes the magnitude
    responses at the half pitch and normalises / doubles at the full pitch, so row i of its bank is a window over
    the concatenated responses and the upper rows are zero.  Reproduced as it is: every row against the compiled
    reference (zero rows stay exactly zero), band arrays equal."""
    x = cases.noise(313 + scale, 1 << r2)
    r = ref.RefPWT(40, r2, samplate=32000, low_fre=100.0, high_fre=12000.0, scale_type=scale, style_type=2,
                   normal_type=normal, is_padding=pad)
    assert r.status == 0
    re, im = r.pwt(x)
    want = re + 1j * im
    o = af.PWT(num=40, radix2_exp=r2, samplate=32000, low_fre=100.0, high_fre=12000.0,
               scale_type=af.SpectralFilterBankScaleType(scale), style_type=af.SpectralFilterBankStyleType.GAMMATONE,
               normal_type=af.SpectralFilterBankNormalType(normal), is_padding=bool(pad))
    got = o.pwt(x)
    assert np.isfinite(want).all() and np.isfinite(got).all()
    live = np.abs(want).max(axis=1) > 0
    assert 10 <= live.sum() < 40 and not np.abs(got[~live]).any()  # about half of the rows carry data
    assert_parity(got, want, TOL, f"gammatone scale{scale} normal{normal} pad{pad}")
    peak = np.abs(want[live]).max(axis=1)
    assert (np.abs(got[live] - want[live]).max(axis=1) <= TOL * peak.max()).all()
    assert np.array_equal(o.get_fre_band_arr(), r.fre_band())
