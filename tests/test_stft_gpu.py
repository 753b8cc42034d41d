"""GPU parity of the STFT object (stft_algorithm.h): every padding mode / position, the streaming
tail, the full-spectrum layout, the inverse (weighted / plain overlap-add, accumulate semantics)
and the batched device calls -- against the golden vectors of the reference and, when
oracle/_ref is present, the compiled reference on fresh inputs."""
import os

import numpy as np
import pytest

import audioflux_amd as af
from oracle import ref, restate
from tests import cases
from tests.conftest import assert_istft_parity, assert_parity

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "stft.npz"))


def make_stft(c, is_continue=False):
    o = af.STFT(radix2_exp=c["radix2_exp"], window_type=af.WindowType(c["window_type"]),
                slide_length=c["slide_length"], is_continue=is_continue)
    pad = cases.stft_pad_args(c)
    if pad:
        o.enable_padding(True)
        o.set_padding(af.PaddingPositionType(pad[0]), af.PaddingModeType(pad[1]),
                      0.0 if pad[2] is None else pad[2], 0.0 if pad[3] is None else pad[3])
    return o


@pytest.mark.parametrize("name", list(cases.STFT_CASES))
def test_stft_matches_golden(name, gold):
    c = cases.STFT_CASES[name]
    o = make_stft(c)
    x = cases.make_input(c["x"], 16000)
    want = gold[f"{name}/re"] + 1j * gold[f"{name}/im"]
    assert o.cal_time_length(len(x)) == want.shape[0]
    re, im = o.stft_full(x)
    assert_parity(re + 1j * im, want, TOL, name)
    # the wrapper-level result: (fft_length // 2 + 1, time)
    assert_parity(o.stft(x), want[:, : want.shape[1] // 2 + 1].T, TOL, name + " wrapper")


@pytest.mark.parametrize("name", list(cases.STFT_STREAMS))
def test_streaming_matches_golden(name, gold):
    c = cases.STFT_STREAMS[name]
    o = make_stft(c, is_continue=True)
    x = cases.noise(c["seed"], sum(c["chunks"]))
    off, rows, tl = 0, [], []
    for n in c["chunks"]:
        tl.append(o.cal_time_length(n))
        re, im = o.stft_full(x[off:off + n])
        rows.append(re + 1j * im)
        off += n
    assert np.array_equal(np.array(tl), gold[f"{name}/tl"])
    assert_parity(np.concatenate(rows), gold[f"{name}/re"] + 1j * gold[f"{name}/im"], TOL, name)


@pytest.mark.parametrize("name", list(cases.ISTFT_CASES))
def test_istft_matches_golden(name, gold):
    src, method, acc = cases.ISTFT_CASES[name]
    c = cases.STFT_CASES[src]
    n = 1 << c["radix2_exp"]
    o = make_stft(c)
    re, im = np.ascontiguousarray(gold[f"{src}/re"]), np.ascontiguousarray(gold[f"{src}/im"])
    t = re.shape[0]
    y = cases.noise(80, o.cal_data_length(t)) if acc else np.zeros(o.cal_data_length(t), np.float32)
    fn = o._lib.stftObj_istft
    fn.restype = None
    import ctypes as C
    fp = C.POINTER(C.c_float)
    fn.argtypes = [C.c_void_p, fp, fp, C.c_int, C.c_int, fp]
    fn(o._obj, re.ctypes.data_as(fp), im.ctypes.data_as(fp), t, method, y.ctypes.data_as(fp))
    gn = restate.istft_norm(t, n, c["slide_length"], restate.fft_window(c["window_type"], n), method)
    assert_istft_parity(y, gold[f"{name}/y"], gn, name)


def test_round_trip_and_wrapper_istft():
    """size-independent property: istft(stft(x)) == x on the framed span (hann, hop N/4), through the
    wrapper methods (half spectrum in, mirrored internally)"""
    o = af.STFT(radix2_exp=11, window_type=af.WindowType.HANN, slide_length=512)
    x = cases.noise(91, 16000 * 4)
    S = o.stft(x)
    assert S.shape == (1025, (len(x) - 2048) // 512 + 1) and S.dtype == np.complex64
    y = o.istft(S)
    assert y.shape == (o.cal_data_length(S.shape[1]),)
    inner = slice(2048, len(y) - 2048)
    assert np.abs(y[inner] - x[: len(y)][inner]).max() <= 1e-5 * np.abs(x).max()
    # multi-channel input: leading axes are independent clips
    xs = np.stack([x[:20000], x[20000:40000]])
    S2 = o.stft(xs)
    assert S2.shape[0] == 2 and np.array_equal(S2[1], o.stft(xs[1]))
    assert np.array_equal(o.istft(S2)[0], o.istft(S2[0]))


def test_custom_window_and_slide_length_switches():
    o = af.STFT(radix2_exp=9, window_type=af.WindowType.RECT, slide_length=128)
    w = restate.fft_window(cases.WIN["hamm"], 512).astype(np.float32)
    o.use_window_data_arr(w)
    assert np.array_equal(o.get_window_data_arr(), w)
    o.set_slide_length(100)
    x = cases.noise(92, 4000)
    re, im = o.stft_full(x)
    assert_parity(re + 1j * im, restate.stft_full(x, 512, 100, w), TOL, "custom window")
    o.set_slide_length(-5)  # ignored
    assert o.cal_time_length(4000) == (4000 - 512) // 100 + 1
    # set_padding is ignored until padding is enabled (stft_algorithm.c:189)
    o.set_padding(af.PaddingPositionType.LEFT, af.PaddingModeType.WRAP)
    o.enable_padding(True)
    re2, im2 = o.stft_full(x)
    want = restate.stft_padded(x, 512, 100, w, position="center", mode="constant")
    assert_parity(re2 + 1j * im2, want, TOL, "default padding after enable")


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("r,hop", [(11, 512), (12, 1024), (11, 300), (12, 999), (13, 3000), (5, 8), (1, 1), (2, 1), (10, 256), (10, 300), (9, 128), (9, 77),
                                   (11, 1024), (10, 512), (8, 64), (8, 100), (8, 256)])
def test_stft_istft_match_compiled_reference_fresh_inputs(r, hop):
    n = 1 << r
    x = cases.noise(100 + r, max(6 * n + 17, 50))
    for pad in (None, (0, 1), (2, 2), (1, 0)):
        rr = ref.RefSTFT(r, 1, hop)
        o = af.STFT(radix2_exp=r, window_type=af.WindowType.HANN, slide_length=hop)
        if pad:
            rr.enable_padding(1)
            rr.set_padding(pad[0], pad[1], 0.25, -0.5)
            o.enable_padding(True)
            o.set_padding(af.PaddingPositionType(pad[0]), af.PaddingModeType(pad[1]), 0.25, -0.5)
        re, im = rr.stft(x)
        gre, gim = o.stft_full(x)
        assert_parity(gre + 1j * gim, re + 1j * im, TOL, f"r{r} hop{hop} pad{pad}")
    if r >= 2:
        for method in (0, 1):
            want = rr.istft(re, im, method)
            mirror_ok = o.istft((re + 1j * im)[:, : n // 2 + 1].T.astype(np.complex64), method)
            gn = restate.istft_norm(re.shape[0], n, hop, rr.window(), method)
            assert_istft_parity(mirror_ok, want, gn, f"istft r{r} hop{hop} m{method}")


def test_device_batch_calls_match_host_calls():
    import torch
    o = af.STFT(radix2_exp=10, window_type=af.WindowType.HANN, slide_length=256)
    o.enable_padding(True)
    o.set_padding(af.PaddingPositionType.CENTER, af.PaddingModeType.REFLECT)
    xs = np.stack([cases.noise(110 + i, 9000) for i in range(5)])
    host = [o.stft_full(x) for x in xs]
    wide = torch.zeros((5, 9000 + 32), dtype=torch.float32, device="cuda")
    wide[:, :9000] = torch.from_numpy(xs).cuda()
    re, im = o.stft_device(wide[:, :9000])   # strided rows
    torch.cuda.synchronize()
    for i in range(5):
        assert np.array_equal(re[i].cpu().numpy(), host[i][0]) and np.array_equal(im[i].cpu().numpy(), host[i][1])
    y = o.istft_device(re, im, method_type=0)
    torch.cuda.synchronize()
    S = (host[2][0] + 1j * host[2][1])[:, :513].T.astype(np.complex64)
    assert np.array_equal(y[2].cpu().numpy(), o.istft(S))


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("r,hop", [(11, 512), (11, 300), (11, 1024), (10, 256), (10, 300), (9, 128), (9, 77), (12, 1024), (12, 999), (8, 64), (8, 100)])
def test_one_launch_istft_over_several_runs_matches_the_reference(r, hop):
    """round 6: k_istft_w2048 / _w4096 / _w256 / k_istft_wsmall cut a clip into runs of >= 32 frames, one wave each, and transform the frames before a
    run again for their tails: clips of ~150 frames (five runs), a batch of two, both synthesis methods, against the compiled
    reference clip by clip; hop N / 4, N / 2, hops that do not divide N"""
    import torch
    n = 1 << r
    xs = np.stack([cases.noise(300 + 7 * r + i, 150 * hop + n) for i in range(2)])
    rr = ref.RefSTFT(r, 1, hop)
    o = af.STFT(radix2_exp=r, window_type=af.WindowType.HANN, slide_length=hop)
    spec = [rr.stft(x) for x in xs]
    re = torch.from_numpy(np.stack([s[0] for s in spec])).cuda()
    im = torch.from_numpy(np.stack([s[1] for s in spec])).cuda()
    for method in (0, 1):
        y = o.istft_device(re, im, method_type=method).cpu().numpy()
        for i in range(2):
            want = rr.istft(spec[i][0], spec[i][1], method)
            gn = restate.istft_norm(spec[i][0].shape[0], n, hop, rr.window(), method)
            assert_istft_parity(y[i, :want.size], want, gn, f"one-launch istft r{r} hop{hop} m{method} clip{i}")


def test_degenerate_inputs():
    o = af.STFT(radix2_exp=8, window_type=af.WindowType.HANN, slide_length=64)
    assert o.cal_time_length(255) == 0 and o.cal_time_length(256) == 1
    re, im = o.stft_full(np.zeros(100, np.float32))   # too short: nothing written, no error
    assert re.shape == (0, 256)
    o.enable_padding(True)
    assert o.cal_time_length(1) == 1 and o.cal_time_length(0) == 0
    re, im = o.stft_full(np.ones(1, np.float32))
    w = o.get_window_data_arr()
    assert_parity(re[0], np.fft.fft(np.where(np.arange(256) == 128, w, 0)).real, 1e-6, "one sample, centre pad")
    with pytest.raises(RuntimeError):
        af.STFT(radix2_exp=31)


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("r,hop", [(11, 512), (11, 301), (12, 1024), (12, 700)])
def test_wave_kernel_every_padding_mode_and_the_generic_kernel(r, hop):
    """n_fft 2048 runs one wave per frame (k_stft_wave): interior frames by vector loads, frames touching
    the clip's ends sample by sample through the padding index map -- every position x mode, against
    the compiled reference (n_fft 4096: the size-generic kernel, same checks)"""
    n = 1 << r
    x = cases.noise(300 + r + hop, 5 * n + 123)
    for pos in (0, 1, 2):
        for mode in (0, 1, 2):
            rr = ref.RefSTFT(r, 1, hop)
            o = af.STFT(radix2_exp=r, window_type=af.WindowType.HANN, slide_length=hop)
            rr.enable_padding(1)
            rr.set_padding(pos, mode, 0.5, -0.25)
            o.enable_padding(True)
            o.set_padding(af.PaddingPositionType(pos), af.PaddingModeType(mode), 0.5, -0.25)
            re, im = rr.stft(x)
            gre, gim = o.stft_full(x)
            assert_parity(gre + 1j * gim, re + 1j * im, TOL, f"r{r} hop{hop} pos{pos} mode{mode}")


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("r,hop", [(12, 1024), (12, 700), (12, 1001), (12, 4096), (10, 256), (10, 175), (10, 1024), (9, 128), (9, 89), (9, 160)])
def test_spectrum_from_the_bank_kernels_transforms(r, hop):
    """n_fft 4096 / 1024 / 512 without padding (every frame inside the clip) run the transform of that size's bank kernel with
    its spectrum stored from registers (k_stft_band_4k2 / _1k / _512 <STFT>, afxk_stft4k / 1k / 512): all N bins of stftObj_stft
    -- conjugate mirrors above N / 2 -- against the compiled reference; hop N / 4 re-uses 3/4 of a frame from registers, the
    other hops fetch every frame whole (odd hops: frame starts on odd samples)."""
    n = 1 << r
    x = cases.noise(900 + hop + r, 9 * n + 77)
    rr = ref.RefSTFT(r, 1, hop)
    o = af.STFT(radix2_exp=r, window_type=af.WindowType.HANN, slide_length=hop)
    re, im = rr.stft(x)
    gre, gim = o.stft_full(x)
    assert gre.shape == re.shape and re.shape[0] >= 2
    assert_parity(gre + 1j * gim, re + 1j * im, TOL, f"n{n} hop{hop}")
    # mirrors are exact conjugates of what the lower half stores, bin 0 / N / 2 have none
    h = n // 2
    assert np.array_equal(gre[:, 1:h], gre[:, :h:-1]) and np.array_equal(gim[:, 1:h], -gim[:, :h:-1])


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("r", [12, 10, 9])
def test_spectrum_kernels_device_batch_with_an_odd_clip_pitch(r):
    """three clips 48 001 floats apart through the device call = each clip through the legacy call, bit for bit;
    more frames per clip than one wave's run, so runs cross clip boundaries"""
    import torch
    o = af.STFT(radix2_exp=r, window_type=af.WindowType.HAMM, slide_length=(1 << r) // 4)
    n = 48000
    base = torch.from_numpy(np.stack([cases.noise(950 + i, n + 1) for i in range(3)])).cuda()
    x = base[:, :n]
    assert x.stride(0) == n + 1
    re, im = o.stft_device(x)
    torch.cuda.synchronize()
    rr = ref.RefSTFT(r, 2, (1 << r) // 4)
    for i in range(3):
        xi = x[i].cpu().numpy()
        g1, g2 = o.stft_full(xi)
        assert np.array_equal(re[i].cpu().numpy(), g1) and np.array_equal(im[i].cpu().numpy(), g2), i
        w1, w2 = rr.stft(xi)
        assert_parity(g1 + 1j * g2, w1 + 1j * w2, TOL, f"clip {i}")


def test_batches_beyond_2_32_threads_per_launch_are_split():
    """the size-generic kernels run one workgroup per frame and HIP rejects a launch with 2^32 or more threads in one
    dimension: afxk_stft / afxk_istft split such a batch into launches of whole clips (found by the CPU launch audit,
    tests/test_hoststub.py).  Three clips of 23 M samples at n_fft 4 / hop 1 = 69 M frames of 64 threads in one call
    must equal the three clips transformed one by one, bit for bit; likewise the inverse of 2 x 8.5 M frames."""
    import torch
    o = af.STFT(radix2_exp=2, window_type=af.WindowType.RECT, slide_length=1)
    n = 23_000_000
    g = torch.Generator(device="cuda").manual_seed(9)
    x = torch.randn((3, n), device="cuda", generator=g)
    re, im = o.stft_device(x)
    torch.cuda.synchronize()
    assert re.shape == (3, n - 3, 4)
    for i in range(3):
        r1, i1 = o.stft_device(x[i:i + 1])
        torch.cuda.synchronize()
        assert torch.equal(re[i], r1[0]) and torch.equal(im[i], i1[0]), i
    # a sample by hand: frame t, bin 1 of a rectangular 4-point transform = (x0 - x2) + i (x3 - x1)
    t = 12_345_678
    w = x[2, t:t + 4].double().cpu().numpy()
    assert abs(float(re[2, t, 1]) - (w[0] - w[2])) < 1e-5 and abs(float(im[2, t, 1]) - (w[3] - w[1])) < 1e-5
    m = 8_500_000
    y = o.istft_device(re[:2, :m].contiguous(), im[:2, :m].contiguous(), method_type=0)
    torch.cuda.synchronize()
    for i in range(2):
        y1 = o.istft_device(re[i:i + 1, :m].contiguous(), im[i:i + 1, :m].contiguous(), method_type=0)
        torch.cuda.synchronize()
        assert torch.equal(y[i], y1[0]), i
