"""GPU parity: the product C-ABI (through the wrapper classes that mirror the
reference's ctypes wrapper) against (a) the committed golden vectors produced by
the reference and (b) the compiled reference itself when oracle/_ref is present.
Tolerance: 1e-5 peak-relative and L2-relative per output tensor (north_star)."""
import os

import numpy as np
import pytest

import audioflux_amd as af
from oracle import ref
from tests import cases
from tests.conftest import assert_parity

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "bft.npz"))


def make_bft(c):
    kw = cases.ctor_kwargs(c)
    o = af.BFT(kw.pop("num"), radix2_exp=kw.pop("radix2_exp"),
               samplate=kw["samplate"], low_fre=kw.get("low_fre"), high_fre=kw.get("high_fre"),
               bin_per_octave=kw.get("bin_per_octave", 12),
               window_type=af.WindowType(kw["window_type"]), slide_length=kw["slide_length"],
               scale_type=af.SpectralFilterBankScaleType(kw["scale_type"]),
               style_type=af.SpectralFilterBankStyleType(kw["style_type"]),
               normal_type=af.SpectralFilterBankNormalType(kw["normal_type"]),
               data_type=af.SpectralDataType(kw["data_type"]),
               is_temporal=bool(kw.get("is_temporal", 0)))
    if "norm" in c:
        o.set_data_norm_value(c["norm"])
    return o


@pytest.mark.parametrize("name", list(cases.BFT_CASES))
def test_bft_matches_golden(name, gold):
    c = cases.BFT_CASES[name]
    o = make_bft(c)
    assert np.array_equal(o.get_fre_band_arr(), gold[f"{name}/fre"])
    assert np.array_equal(o.get_bin_band_arr(), gold[f"{name}/bin"])
    x = cases.make_input(c["x"], c["samplate"])
    got = o.bft(x, result_type=c["result_type"]).T  # wrapper returns (num, time)
    want = gold[f"{name}/re"]
    if c["result_type"] == 0:
        want = want + 1j * gold[f"{name}/im"]
    assert_parity(got, want, TOL, name)
    if c.get("is_temporal"):
        e, r, z = o.get_temporal_data()
        assert_parity(e, gold[f"{name}/energy"], TOL, "energy")
        assert_parity(r, gold[f"{name}/rms"], TOL, "rms")
        assert np.array_equal(z, gold[f"{name}/zcr"])


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("seed", [21, 22])
def test_bft_matches_compiled_reference_fresh_inputs(seed):
    """fresh seeds (not in the fixtures) straight against the reference library"""
    x = cases.noise(seed, 16000 * 3 + 123)
    # hop 512 runs the register-reuse variant of the fused kernel, hop 300 the plain one;
    # result type 0 (complex, the wrapper's default) its two-pass filter-bank stage
    for hop in (512, 300, 256, 1024):
        for rt in (1, 0):
            for dt in (0, 1):
                r = ref.RefBFT(128, 11, samplate=16000, low_fre=0.0, high_fre=8000.0, window_type=1,
                               slide_length=hop, scale_type=2, style_type=0, normal_type=0, data_type=dt)
                r.set_result_type(rt)
                re, im = r.bft(x)
                o = af.BFT(128, radix2_exp=11, samplate=16000, low_fre=0.0, high_fre=8000.0,
                           slide_length=hop, scale_type=af.SpectralFilterBankScaleType.MEL,
                           data_type=af.SpectralDataType(dt))
                got = o.bft(x, result_type=rt).T
                assert_parity(got, re if rt == 1 else re + 1j * im, TOL, f"hop{hop} rt{rt} dt{dt}")


def test_batch_equals_per_clip_loop():
    o = af.BFT(128, radix2_exp=11, samplate=16000, low_fre=0.0, high_fre=8000.0, slide_length=512,
               scale_type=af.SpectralFilterBankScaleType.MEL, data_type=af.SpectralDataType.POWER)
    xs = np.stack([cases.noise(30 + i, 20000) for i in range(5)])
    loop = np.stack([o.bft(x, result_type=1).T for x in xs])
    batch = o.bft_batch(xs, result_type=1)
    assert np.array_equal(loop, batch)  # same kernels, same order of operations
    # a caller that loops keeps its result array (out=): written in place, shape and layout checked
    keep = np.full_like(batch, np.nan)
    assert o.bft_batch(xs, result_type=1, out=keep) is keep and np.array_equal(keep, batch)
    with pytest.raises(ValueError):
        o.bft_batch(xs, result_type=1, out=np.zeros(batch.shape[::-1], np.float32))
    with pytest.raises(ValueError):
        o.bft_batch(xs, result_type=1, out=np.zeros(batch.shape, np.float64))


def test_device_resident_api_matches_host_api():
    import torch
    o = af.BFT(128, radix2_exp=11, samplate=16000, low_fre=0.0, high_fre=8000.0, slide_length=512,
               scale_type=af.SpectralFilterBankScaleType.MEL, data_type=af.SpectralDataType.POWER)
    xs = np.stack([cases.noise(40 + i, 30000) for i in range(7)])
    host = o.bft_batch(xs, result_type=1)
    xd = torch.from_numpy(xs).cuda()
    out = o.bft_device(xd)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), host)
    # strided clips (rows of a wider buffer)
    wide = torch.zeros((7, 30000 + 64), dtype=torch.float32, device="cuda")
    wide[:, :30000] = xd
    out2 = o.bft_device(wide[:, :30000])
    torch.cuda.synchronize()
    assert np.array_equal(out2.cpu().numpy(), host)


def test_short_and_degenerate_inputs():
    o = af.BFT(128, radix2_exp=11, samplate=16000, slide_length=512,
               scale_type=af.SpectralFilterBankScaleType.MEL)
    assert o.cal_time_length(2047) == 0 and o.cal_time_length(2048) == 1
    with pytest.raises(ValueError):
        o.bft(np.zeros(100, np.float32))
    z = o.bft(np.zeros(4096, np.float32), result_type=1)
    assert z.shape == (128, 5) and not z.any()
    # reuse of one object across lengths (scratch regrowth)
    for n in (5000, 60000, 9000):
        assert o.bft(cases.noise(1, n), result_type=1).shape == (128, (n - 2048) // 512 + 1)


def test_linearity_and_scaling_property():
    """size-independent property: power-mel is quadratic, magnitude-mel is linear in the input gain"""
    x = cases.noise(50, 40000)
    p = af.BFT(64, radix2_exp=10, samplate=16000, slide_length=256,
               scale_type=af.SpectralFilterBankScaleType.MEL, data_type=af.SpectralDataType.POWER)
    m = af.BFT(64, radix2_exp=10, samplate=16000, slide_length=256,
               scale_type=af.SpectralFilterBankScaleType.MEL, data_type=af.SpectralDataType.MAG)
    assert_parity(p.bft(2 * x, 1), 4 * p.bft(x, 1), 1e-6, "power gain")
    assert_parity(m.bft(2 * x, 1), 2 * m.bft(x, 1), 1e-6, "mag gain")


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("scale", [2, 3, 4])   # mel, bark, erb
def test_bft_nfft1024_fused_kernel_matches_compiled_reference(scale):
    """n_fft 1024 runs k_stft_band_1k (512-point complex FFT in 8 registers per lane):
    hop 256 (register re-use) and 200 (plain), real / complex results, power / magnitude,
    with and without a norm exponent, against the reference library."""
    x = cases.noise(60 + scale, 16000 * 2 + 77)
    for hop in (256, 200):
        for rt, dt, norm in ((1, 0, None), (1, 1, None), (0, 0, None), (0, 1, None), (1, 0, 0.5), (1, 1, 2.0)):
            r = ref.RefBFT(64 if scale != 2 else 128, 10, samplate=16000, low_fre=0.0, high_fre=8000.0,
                           window_type=1, slide_length=hop, scale_type=scale, style_type=0, normal_type=0,
                           data_type=dt)
            assert r.status == 0
            r.set_result_type(rt)
            if norm:
                r.set_norm(norm)
            re, im = r.bft(x)
            o = af.BFT(64 if scale != 2 else 128, radix2_exp=10, samplate=16000, low_fre=0.0, high_fre=8000.0,
                       slide_length=hop, scale_type=af.SpectralFilterBankScaleType(scale),
                       data_type=af.SpectralDataType(dt))
            if norm:
                o.set_data_norm_value(norm)
            got = o.bft(x, result_type=rt).T
            assert_parity(got, re if rt == 1 else re + 1j * im, TOL, f"scale{scale} hop{hop} rt{rt} dt{dt} norm{norm}")


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("scale,num,sr", [(2, 128, 16000), (2, 80, 16000), (2, 40, 16000), (2, 26, 44100), (3, 64, 16000), (4, 64, 16000)])
def test_bft_nfft512_fused_kernel_matches_compiled_reference(scale, num, sr):
    """n_fft 512 (32 ms of 16 kHz speech) runs k_stft_band_512 (256-point complex FFT as 4 x 4 x 4 x 4 in four registers
    per lane): every tap variant (mel-128: 16 + 4 ... mel-26 at 44.1 kHz: 64 + 8), hop 128 (register re-use), 160 and 101
    (plain; frames on odd samples), real / complex results, power / magnitude, norm exponent -- against the reference library;
    then the device batch call on clips an odd number of floats apart."""
    x = cases.noise(80 + scale + num, sr + 77)
    kinds = set()
    for hop in (128, 160, 101):
        for rt, dt, norm in ((1, 0, None), (1, 1, None), (0, 0, None), (0, 1, None), (1, 0, 0.5), (1, 1, 2.0)):
            r = ref.RefBFT(num, 9, samplate=sr, low_fre=0.0, high_fre=sr / 2, window_type=1, slide_length=hop,
                           scale_type=scale, style_type=0, normal_type=0, data_type=dt)
            assert r.status == 0
            r.set_result_type(rt)
            if norm:
                r.set_norm(norm)
            re, im = r.bft(x)
            o = af.BFT(num, radix2_exp=9, samplate=sr, low_fre=0.0, high_fre=sr / 2, slide_length=hop,
                       scale_type=af.SpectralFilterBankScaleType(scale), data_type=af.SpectralDataType(dt))
            kinds.add(o.fused_plan_kind())
            if norm:
                o.set_data_norm_value(norm)
            got = o.bft(x, result_type=rt).T
            assert_parity(got, re if rt == 1 else re + 1j * im, TOL, f"scale{scale} num{num} hop{hop} rt{rt} dt{dt} norm{norm}")
    assert kinds == {301}, kinds
    import torch
    n = sr + 4
    xs = np.stack([cases.noise(180 + scale + i, n + 1) for i in range(3)])
    xd = torch.from_numpy(xs).cuda()[:, :n]
    for rt in (1, 0):
        o = af.BFT(num, radix2_exp=9, samplate=sr, low_fre=0.0, high_fre=sr / 2, slide_length=128,
                   scale_type=af.SpectralFilterBankScaleType(scale), data_type=af.SpectralDataType.POWER)
        o.set_result_type(rt)
        out = o.bft_device(xd)
        torch.cuda.synchronize()
        for i in range(3):
            host = o.bft(xs[i, :n], result_type=rt).T
            dev = out[i].cpu().numpy() if rt == 1 else out[0][i].cpu().numpy() + 1j * out[1][i].cpu().numpy()
            assert np.array_equal(dev, host), (rt, i)


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("scale", [2, 3])   # mel, bark
def test_bft_nfft4096_fused_kernel_matches_compiled_reference(scale):
    """n_fft 4096 (the reference wrapper's default) runs k_stft_band_4k2: two 1024-point
    halves of the packed frame through the 16 x 16 x 4 pipeline, lane-local combine + real-input
    split.  hop 1024 (register re-use) and 900 (plain), power / magnitude / norm exponent,
    real and complex results."""
    x = cases.noise(70 + scale, 16000 * 3 + 55)
    for hop in (1024, 900, 1001):  # 1001: frames start on odd samples (8-byte loads at 4-byte alignment)
        for rt, dt, norm in ((1, 0, None), (1, 1, None), (1, 0, 0.5), (1, 1, 2.0), (0, 0, None), (0, 1, None)):
            r = ref.RefBFT(128, 12, samplate=16000, low_fre=0.0, high_fre=8000.0, window_type=1,
                           slide_length=hop, scale_type=scale, style_type=0, normal_type=0, data_type=dt)
            assert r.status == 0
            r.set_result_type(rt)
            if norm:
                r.set_norm(norm)
            re, im = r.bft(x)
            o = af.BFT(128, radix2_exp=12, samplate=16000, low_fre=0.0, high_fre=8000.0, slide_length=hop,
                       scale_type=af.SpectralFilterBankScaleType(scale), data_type=af.SpectralDataType(dt))
            if norm:
                o.set_data_norm_value(norm)
            got = o.bft(x, result_type=rt).T
            assert_parity(got, re if rt == 1 else re + 1j * im, TOL, f"scale{scale} hop{hop} rt{rt} dt{dt} norm{norm}")
    # the device batch call on clips an ODD number of samples apart (row pitch n + 1: every second clip starts at 4-byte
    # alignment), real and complex results, three clips across the wave's clip boundaries
    import torch
    n = 16000 * 2 + 4
    xs = np.stack([cases.noise(170 + scale + i, n + 1) for i in range(3)])
    xd = torch.from_numpy(xs).cuda()[:, :n]
    for rt in (1, 0):
        o = af.BFT(128, radix2_exp=12, samplate=16000, low_fre=0.0, high_fre=8000.0, slide_length=1024,
                   scale_type=af.SpectralFilterBankScaleType(scale), data_type=af.SpectralDataType.POWER)
        o.set_result_type(rt)
        out = o.bft_device(xd)
        torch.cuda.synchronize()
        r = ref.RefBFT(128, 12, samplate=16000, low_fre=0.0, high_fre=8000.0, window_type=1, slide_length=1024,
                       scale_type=scale, style_type=0, normal_type=0, data_type=0)
        r.set_result_type(rt)
        for i in range(3):
            re, im = r.bft(np.ascontiguousarray(xs[i, :n]))
            got = out[i].cpu().numpy() if rt == 1 else out[0][i].cpu().numpy() + 1j * out[1][i].cpu().numpy()
            want = re if rt == 1 else re + 1j * im
            assert_parity(got, want.reshape(got.shape) if want.shape != got.shape else want, TOL, f"batch scale{scale} rt{rt} clip{i} (odd row pitch)")


def _bank32(num, n, sr, scale):
    """the float32 bank the object uploads (bit-identical to the reference's, tests/test_host_setup.py)"""
    import ctypes as C
    lib = af.get_lib()
    fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int)
    lib.afx_auditory_bank.restype = None
    lib.afx_auditory_bank.argtypes = [C.c_int] * 6 + [C.c_float, C.c_float, C.c_int, fp, fp, ip]
    bank = np.zeros((num, n // 2 + 1), np.float32)
    fre, bins = np.zeros(num + 2, np.float32), np.zeros(num + 2, np.int32)
    lib.afx_auditory_bank(num, n, sr, scale, 0, 0, 0.0, sr / 2.0, 12, bank.ctypes.data_as(fp),
                          fre.ctypes.data_as(fp), bins.ctypes.data_as(ip))
    return bank


SPLIT_CASES = [  # (radix2_exp, scale, num, samplate): banks whose rows exceed the fused kernels' tap variants
    (11, "MEL", 40, 16000), (11, "MEL", 64, 32000), (11, "MEL", 20, 22050), (11, "BARK", 64, 16000),
    (11, "BARK", 80, 44100), (11, "ERB", 64, 22050), (11, "ERB", 40, 44100),
    (12, "MEL", 80, 32000), (12, "MEL", 40, 16000), (12, "BARK", 40, 32000), (12, "BARK", 64, 44100),
    (12, "ERB", 64, 22050),
    # n_fft 1024 / 512 (round 5): rows longer than those kernels' tap variants
    (10, "MEL", 13, 16000), (10, "MEL", 26, 22050), (10, "BARK", 24, 16000), (9, "MEL", 13, 16000), (9, "MEL", 20, 44100),
    (9, "BARK", 24, 16000),
]


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("r2,scale,num,sr", SPLIT_CASES)
def test_bft_split_plan_matches_compiled_reference(r2, scale, num, sr, monkeypatch):
    """n_fft 512 ... 4096 with long-row banks: the fused kernel runs a split plan (row segments per lane slot,
    summed in ascending bin order) instead of falling back to the size-generic kernel -- real and
    complex results, power / magnitude / norm exponent, the register-reuse (hop 512) and plain (hop 300)
    instantiations (the size-generic kernel is held to the same reference by the other transform sizes)"""
    st = getattr(af.SpectralFilterBankScaleType, scale)
    noise = cases.noise(700 + num, sr * 2 + 77)
    tonal = noise.copy()
    tonal[sr // 2:sr] += (0.3 * np.sin(np.arange(sr - sr // 2) * 0.21)).astype(np.float32)
    kw = dict(radix2_exp=r2, samplate=sr, low_fre=0.0, high_fre=sr / 2.0, scale_type=st)
    nfft = 1 << r2
    for hop in (nfft // 4, 300):
        for rt, dt, norm in ((1, 0, 1.0), (1, 1, 1.0), (1, 0, 0.5), (0, 0, 1.0), (0, 1, 1.0)):
            # complex results sum the spectrum itself: around a strong tone the band sums cancel to
            # far below max|S|, and the float32 error of S (1e-7 of max|S| -- the reference's too:
            # 2.5e-5 of the output peak for mel-20 with the tone) is all that is left; the tone
            # therefore goes through the real-result modes only
            x = tonal if rt == 1 else noise
            r = ref.RefBFT(num, r2, samplate=sr, low_fre=0.0, high_fre=sr / 2.0, window_type=1, slide_length=hop,
                           scale_type=int(st), style_type=0, normal_type=0, data_type=dt)
            r.set_result_type(rt)
            if norm != 1.0:
                r.set_norm(norm)
            re, im = r.bft(x)
            o = af.BFT(num, slide_length=hop, data_type=af.SpectralDataType(dt), **kw)
            assert o.fused_plan_kind() == {9: 302, 10: 102, 11: 2, 12: 202}[r2], (scale, num, sr)
            if norm != 1.0:
                o.set_data_norm_value(norm)
            got = o.bft(x, result_type=rt).T
            tol = TOL
            if rt == 0:
                # Complex results sum the spectrum itself (or its square) over the band.  Frames are
                # not centred at the phase origin, so S[k] alternates in sign from bin to bin and a
                # smooth band of 100-250 weights cancels to far below max|S|: what is left carries
                # the float32 error of S (1e-7 of max|S|) -- the reference's own distance from a
                # float64 evaluation reaches 3e-5 of the output peak here.  Bar: the larger of TOL
                # and 3x that distance.
                from oracle import restate
                fr = restate.frames_of(x.astype(np.float64), nfft, hop) * restate.fft_window(1, nfft)[None, :]
                S = np.fft.rfft(fr, axis=1)
                f64 = (S if dt == 1 else S * S) @ _bank32(num, nfft, sr, int(st)).astype(np.float64).T
                want = re + 1j * im
                ref_err = max(np.abs(want - f64).max() / np.abs(f64).max(),
                              np.linalg.norm(want - f64) / np.linalg.norm(f64))
                tol = max(TOL, 3.0 * ref_err)
                assert_parity(got, f64, tol, f"{scale}-{num}@{sr} hop{hop} rt{rt} dt{dt} vs float64")
            assert_parity(got, re if rt == 1 else re + 1j * im, tol, f"{scale}-{num}@{sr} hop{hop} rt{rt} dt{dt} norm{norm}")


def test_fused_plan_kinds():
    mk = lambda num, r, sr=16000, **k: af.BFT(num, radix2_exp=r, samplate=sr, low_fre=0.0, high_fre=sr / 2.0,
                                              scale_type=af.SpectralFilterBankScaleType.MEL, **k)
    assert mk(128, 11).fused_plan_kind() == 1
    assert mk(40, 11).fused_plan_kind() == 2
    assert mk(13, 11).fused_plan_kind() == 0   # rows of ~160 bins: more than four segments
    assert mk(128, 10).fused_plan_kind() == 101
    assert mk(128, 9).fused_plan_kind() == 301 and mk(40, 9).fused_plan_kind() == 301
    assert mk(128, 12).fused_plan_kind() == 201
    assert mk(80, 12, 32000).fused_plan_kind() == 202


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("hop", [512, 300])
def test_bft_linear_scale_n2048_every_result_mode(hop):
    """linear scale at n_fft 2048 = a bin slice straight from the wave-per-frame STFT kernel: power,
    magnitude, norm exponent, and the complex results S / S^2"""
    x = cases.noise(41, 16000 * 2 + 5)
    for rt, dt, norm in ((1, 0, 1.0), (1, 1, 1.0), (1, 0, 0.5), (1, 1, 2.0), (0, 0, 1.0), (0, 1, 1.0)):
        r = ref.RefBFT(100, 11, samplate=16000, low_fre=1000.0, high_fre=8000.0, window_type=1, slide_length=hop,
                       scale_type=0, style_type=0, normal_type=0, data_type=dt)
        r.set_result_type(rt)
        if norm != 1.0:
            r.set_norm(norm)
        re, im = r.bft(x)
        o = af.BFT(100, radix2_exp=11, samplate=16000, low_fre=1000.0, high_fre=8000.0, slide_length=hop,
                   scale_type=af.SpectralFilterBankScaleType.LINEAR, data_type=af.SpectralDataType(dt))
        if norm != 1.0:
            o.set_data_norm_value(norm)
        got = o.bft(x, result_type=rt).T
        assert_parity(got, re if rt == 1 else re + 1j * im, TOL, f"linear hop{hop} rt{rt} dt{dt} norm{norm}")


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("r2,hop", [(12, 1024), (12, 600), (10, 256), (10, 150), (9, 128), (9, 75)])
def test_bft_linear_scale_on_the_spectrum_kernels_every_result_mode(r2, hop):
    """linear scale at n_fft 4096 / 1024 / 512 = a bin slice from k_stft_band_4k2 / _1k / _512 <STFT> (range-checked stores,
    every AFX_SPEC_* map)"""
    x = cases.noise(43 + r2, 16000 * 3 + 5)
    num = (1 << r2) * 300 // 4096
    for rt, dt, norm in ((1, 0, 1.0), (1, 1, 1.0), (1, 0, 0.5), (1, 1, 2.0), (0, 0, 1.0), (0, 1, 1.0)):
        r = ref.RefBFT(num, r2, samplate=16000, low_fre=500.0, high_fre=8000.0, window_type=1, slide_length=hop,
                       scale_type=0, style_type=0, normal_type=0, data_type=dt)
        r.set_result_type(rt)
        if norm != 1.0:
            r.set_norm(norm)
        re, im = r.bft(x)
        o = af.BFT(num, radix2_exp=r2, samplate=16000, low_fre=500.0, high_fre=8000.0, slide_length=hop,
                   scale_type=af.SpectralFilterBankScaleType.LINEAR, data_type=af.SpectralDataType(dt))
        if norm != 1.0:
            o.set_data_norm_value(norm)
        got = o.bft(x, result_type=rt).T
        assert_parity(got, re if rt == 1 else re + 1j * im, TOL, f"linear r{r2} hop{hop} rt{rt} dt{dt} norm{norm}")


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("hop", [512, 300])
def test_temporal_features_ride_in_the_fused_kernel(hop):
    """isTemporal objects at n_fft 2048 stay on k_stft_mel_v2 (energy / rms / zcr as wave reductions,
    temporal_algorithm.c:138-144) -- whole-row and split plans, register re-use and plain fetch."""
    x = cases.tones(12, 16000 * 2 + 301, 16000) + 0.05 * cases.noise(70, 16000 * 2 + 301)
    for num, scale in ((128, 2), (64, 3)):  # mel-128: whole rows; bark-64: split plan
        r = ref.RefBFT(num, 11, samplate=16000, low_fre=0.0, high_fre=8000.0, window_type=1,
                       slide_length=hop, scale_type=scale, style_type=0, normal_type=0, data_type=0, is_temporal=1)
        r.set_result_type(1)
        re, _ = r.bft(x)
        we, wr, wz = r.temporal(re.shape[0])
        o = af.BFT(num, radix2_exp=11, samplate=16000, low_fre=0.0, high_fre=8000.0, slide_length=hop,
                   scale_type=af.SpectralFilterBankScaleType(scale), data_type=af.SpectralDataType.POWER,
                   is_temporal=True)
        assert o.fused_plan_kind() in (1, 2)
        got = o.bft(x, result_type=1).T
        e, rms, z = o.get_temporal_data()
        assert_parity(got, re, TOL, f"temporal spec num{num} hop{hop}")
        assert_parity(e, we, TOL, "energy")
        assert_parity(rms, wr, TOL, "rms")
        # a sign change decided by a product at float32 rounding may differ: at most one count per frame
        assert np.abs(z - wz).max() <= 1.0 / 2048 + 1e-9 and (z != wz).mean() <= 0.02, np.abs(z - wz).max()


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("r2", [9, 11, 12])
def test_temporal_features_beside_a_linear_bin_slice(r2):
    """linear-scale objects with isTemporal: the bins from the wave / spectrum kernels, the features from k_temporal"""
    n, hop = 1 << r2, (1 << r2) // 4
    num = n * 100 // 2048
    # (noise: single power bins of a strong tone sit 1-3e-5 from the reference, whose float32 radix-2 transform carries that
    #  error -- the mel rows of the test below average it away)
    x = cases.noise(75 + r2, 16000 * 2 + 301)
    r = ref.RefBFT(num, r2, samplate=16000, low_fre=1000.0, high_fre=8000.0, window_type=1, slide_length=hop, scale_type=0,
                   style_type=0, normal_type=0, data_type=0, is_temporal=1)
    r.set_result_type(1)
    re, _ = r.bft(x)
    we, wr, wz = r.temporal(re.shape[0])
    o = af.BFT(num, radix2_exp=r2, samplate=16000, low_fre=1000.0, high_fre=8000.0, slide_length=hop,
               scale_type=af.SpectralFilterBankScaleType.LINEAR, data_type=af.SpectralDataType.POWER, is_temporal=True)
    got = o.bft(x, result_type=1).T
    e, rms, z = o.get_temporal_data()
    assert_parity(got, re, TOL, f"linear temporal r{r2}")
    assert_parity(e, we, TOL, "energy")
    assert_parity(rms, wr, TOL, "rms")
    assert np.abs(z - wz).max() <= 1.0 / n + 1e-9 and (z != wz).mean() <= 0.02, np.abs(z - wz).max()


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("r2,rt", [(9, 1), (10, 1), (12, 1), (11, 0), (12, 0)])
def test_temporal_features_beside_the_other_fused_kernels(r2, rt):
    """isTemporal objects at n_fft 512 / 1024 / 4096 (and complex results at any size) keep their fused bank kernel: the
    energy / rms / zcr come from k_temporal, one wave per frame over the frames the bank kernel has just read (round 5;
    such objects ran the size-generic kernels for everything before) -- bank rows and features against the reference"""
    n, hop = 1 << r2, (1 << r2) // 4
    x = cases.tones(12, 16000 * 2 + 301, 16000) + 0.05 * cases.noise(71 + r2, 16000 * 2 + 301)
    r = ref.RefBFT(128, r2, samplate=16000, low_fre=0.0, high_fre=8000.0, window_type=1, slide_length=hop, scale_type=2,
                   style_type=0, normal_type=0, data_type=0, is_temporal=1)
    r.set_result_type(rt)
    re, im = r.bft(x)
    we, wr, wz = r.temporal(re.shape[0])
    o = af.BFT(128, radix2_exp=r2, samplate=16000, low_fre=0.0, high_fre=8000.0, slide_length=hop,
               scale_type=af.SpectralFilterBankScaleType.MEL, data_type=af.SpectralDataType.POWER, is_temporal=True)
    assert o.fused_plan_kind() != 0
    got = o.bft(x, result_type=rt).T
    e, rms, z = o.get_temporal_data()
    assert_parity(got, re if rt == 1 else re + 1j * im, TOL, f"temporal spec r{r2} rt{rt}")
    assert_parity(e, we, TOL, "energy")
    assert_parity(rms, wr, TOL, "rms")
    # a sign change decided by a product at float32 rounding may differ: at most one count per frame
    assert np.abs(z - wz).max() <= 1.0 / n + 1e-9 and (z != wz).mean() <= 0.02, np.abs(z - wz).max()


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("rt,dt", [(1, 0), (1, 1), (0, 0)])
def test_dense_bank_at_the_headline_shape(rt, dt):
    """Dense (gammatone) bank at n_fft 2048 / 128 bands: STFT wave kernel -> pitched [T,F] scratch ->
    128 x 128 MFMA GEMM (k_gemm_nt128), chunked; real power / magnitude and complex results against the
    reference's double-accumulating __mdot1 (flux_vector.c:55-86) on several clips (more than one chunk
    with AFX_SCRATCH_MB=1)."""
    xs = np.stack([cases.noise(80 + i, 16000 * 2 + 50) for i in range(3)])
    r = ref.RefBFT(128, 11, samplate=16000, low_fre=0.0, high_fre=8000.0, window_type=1, slide_length=512,
                   scale_type=4, style_type=2, normal_type=0, data_type=dt)
    r.set_result_type(rt)
    o = af.BFT(128, radix2_exp=11, samplate=16000, low_fre=0.0, high_fre=8000.0, slide_length=512,
               scale_type=af.SpectralFilterBankScaleType.ERB, style_type=af.SpectralFilterBankStyleType.GAMMATONE,
               data_type=af.SpectralDataType(dt))
    assert o.fused_plan_kind() == 0  # dense bank: no banded plan
    old = os.environ.get("AFX_SCRATCH_MB")
    try:
        for mb in (None, "1"):
            if mb:
                os.environ["AFX_SCRATCH_MB"] = mb
            got = o.bft_batch(xs, result_type=rt)
            for i in range(3):
                re, im = r.bft(xs[i])
                want = re if rt == 1 else re + 1j * im
                g = got[i] if rt == 1 else got[i]
                assert_parity(g, want, TOL, f"gammatone rt{rt} dt{dt} clip{i} scratch{mb}")
    finally:
        if old is None:
            os.environ.pop("AFX_SCRATCH_MB", None)
        else:
            os.environ["AFX_SCRATCH_MB"] = old


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("hop,dt,num,extra,norm", [(256, 0, 128, 51, None), (1024, 1, 128, 50, None), (300, 0, 40, 7, None), (512, 0, 12, 1, None),
                                                    (512, 0, 128, 3, 0.6), (256, 1, 40, 2, 0.7)])
def test_dense_bank_rows_from_the_headline_transform(hop, dt, num, extra, norm):
    """round 6: the dense route's two launches at the shapes around the headline one -- afxk_stft2k (k_stft_mel_v2 <STFT>) with the
    register re-use of overlapping frames at hop N / 8 and N / 2 and whole-frame fetches from any sample (hop 300, clips an odd
    number of floats apart), power and magnitude rows; k_gemm_bank_bf16x3 on banks of 128, 40 and 12 rows (column padding of the
    tile, the k tail: 1025 = 64 x 16 + 1), more than one chunk (AFX_SCRATCH_MB=1), the norm exponent on the power rows (the kernel's
    |S|^2p map) and on magnitude banks (the product's power-law epilogue) -- against the reference's double-accumulating
    __mdot1 (flux_vector.c:55-86) on the reference's own spectrum (stft_algorithm.c:717-803)"""
    xs = np.stack([cases.noise(90 + i, 16000 + extra) for i in range(3)])
    r = ref.RefBFT(num, 11, samplate=16000, low_fre=50.0, high_fre=7000.0, window_type=1, slide_length=hop,
                   scale_type=4, style_type=2, normal_type=0, data_type=dt)
    r.set_result_type(1)
    o = af.BFT(num, radix2_exp=11, samplate=16000, low_fre=50.0, high_fre=7000.0, slide_length=hop,
               scale_type=af.SpectralFilterBankScaleType.ERB, style_type=af.SpectralFilterBankStyleType.GAMMATONE,
               data_type=af.SpectralDataType(dt))
    assert o.fused_plan_kind() == 0
    if norm is not None:
        r.set_norm(norm)
        o.set_data_norm_value(norm)
    old = os.environ.get("AFX_SCRATCH_MB")
    try:
        os.environ["AFX_SCRATCH_MB"] = "1"
        got = o.bft_batch(xs, result_type=1)
    finally:
        if old is None:
            os.environ.pop("AFX_SCRATCH_MB", None)
        else:
            os.environ["AFX_SCRATCH_MB"] = old
    for i in range(3):
        re, _ = r.bft(xs[i])
        assert_parity(got[i], re, TOL, f"gammatone-{num} hop {hop} dt{dt} clip{i}")

