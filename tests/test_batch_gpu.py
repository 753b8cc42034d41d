"""Batched / HBM-resident entry points of include/afx_batch.h for CWT, CQT(+chroma),
cepstrogram and xxcc: identical to looping the legacy one-clip call (bit-exact, same
kernels), and within tolerance of the compiled reference on sampled clips."""
import os

import numpy as np
import pytest

import audioflux_amd as af
from oracle import ref
from tests.conftest import assert_parity

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    return torch


def test_cwt_batch_device_equals_loop_and_reference():
    torch = _torch()
    rng = np.random.default_rng(31)
    x = (0.1 * rng.standard_normal((5, 4096))).astype(np.float32)
    o = af.CWT(num=40, radix2_exp=12, samplate=16000, low_fre=40.0, bin_per_octave=8,
               wavelet_type=af.WaveletContinueType.MORLET, scale_type=af.SpectralFilterBankScaleType.OCTAVE)
    xd = torch.from_numpy(x).cuda()
    re, im = o.cwt_device(xd)
    torch.cuda.synchronize()
    got = (re.cpu().numpy() + 1j * im.cpu().numpy())[:, ::-1, :]   # -> ascending frequency
    for i in range(x.shape[0]):
        loop = o.cwt(x[i])
        assert np.array_equal(got[i], loop), f"chunk {i}: batch != loop"
    if ref.available():
        r = ref.RefCWT(num=40, radix2_exp=12, samplate=16000, low_fre=40.0, bin_per_octave=8,
                       wavelet_type=int(af.WaveletContinueType.MORLET),
                       scale_type=int(af.SpectralFilterBankScaleType.OCTAVE), is_padding=1)
        rre, rim = r.cwt(x[3])
        assert_parity(got[3][::-1], rre + 1j * rim, what="cwt batch vs reference")
    # strided input (chunks cut out of longer clips) and the d/dt variant
    long = (0.1 * rng.standard_normal((2, 3 * 4096))).astype(np.float32)
    ld = torch.from_numpy(long).cuda()
    chunks = ld.view(2, 3, 4096)[:, 1, :]                     # stride 3*4096 between chunks
    re2, im2 = o.cwt_device(chunks)
    torch.cuda.synchronize()
    assert np.array_equal((re2.cpu().numpy() + 1j * im2.cpu().numpy())[1, ::-1], o.cwt(long[1, 4096:8192]))
    o.enable_det(True)
    dre, dim = o.cwt_device(xd, det=True)
    torch.cuda.synchronize()
    assert np.array_equal((dre.cpu().numpy() + 1j * dim.cpu().numpy())[2, ::-1], o.cwt_det(x[2]))


def test_cwt_batch_host():
    import ctypes
    rng = np.random.default_rng(32)
    x = (0.1 * rng.standard_normal((3, 2048))).astype(np.float32)
    o = af.CWT(num=24, radix2_exp=11, samplate=16000, low_fre=60.0, bin_per_octave=6,
               wavelet_type=af.WaveletContinueType.MORLET)
    re = np.zeros((3, 24, 2048), np.float32)
    im = np.zeros_like(re)
    fn = o._lib.cwtObj_cwtBatch
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    assert fn(o._obj, x.ctypes.data, 3, re.ctypes.data, im.ctypes.data) == 0
    for i in range(3):
        assert np.array_equal((re[i] + 1j * im[i])[::-1], o.cwt(x[i]))


def test_cqt_batch_device_equals_loop_and_reference():
    torch = _torch()
    rng = np.random.default_rng(33)
    n = 30000
    x = (0.1 * rng.standard_normal((4, n))).astype(np.float32)
    o = af.CQT(num=84, samplate=44100, low_fre=32.703, bin_per_octave=12,
               normal_type=af.SpectralFilterBankNormalType.AREA)
    xd = torch.from_numpy(x).cuda()
    re, im = o.cqt_device(xd)
    ch = o.chroma_device(re, im)
    torch.cuda.synchronize()
    got = np.swapaxes(re.cpu().numpy() + 1j * im.cpu().numpy(), -1, -2)      # (clips, num, time)
    gch = np.swapaxes(ch.cpu().numpy(), -1, -2)
    for i in range(x.shape[0]):
        q = o.cqt(x[i])
        assert np.array_equal(got[i], q), f"clip {i}: batch != loop"
        assert np.array_equal(gch[i], o.chroma(q)), f"clip {i}: chroma batch != loop"
    if ref.available():
        r = ref.RefCQT(num=84, samplate=44100, min_fre=32.703, bin_per_octave=12, normal_type=1)
        rre, rim = r.cqt(x[2])
        assert_parity(got[2].T, rre + 1j * rim, what="cqt batch vs reference")
        assert_parity(gch[2].T, r.chroma(rre, rim), what="chroma batch vs reference")
    # the fused call, in one pass and in passes of 3 clips: the transform's bits are those of the two calls; its chroma
    # travels through the one-launch ladder as partial sums, highest octave first (the chroma kernel of the separate
    # call sums lowest first): the same terms, last-bit differences of the normalised values
    for chunk in (None, "3"):
        if chunk:
            os.environ["AFX_CQT_CHUNK"] = chunk
        try:
            re3, im3, ch3 = o.cqt_chroma_device(xd)
            re4, im4 = o.cqt_device(xd)
            torch.cuda.synchronize()
        finally:
            os.environ.pop("AFX_CQT_CHUNK", None)
        assert torch.equal(re3, re) and torch.equal(im3, im), f"fused call, chunk {chunk}"
        assert float((ch3 - ch).abs().max()) <= 2e-6, f"fused call's chroma, chunk {chunk}: {float((ch3 - ch).abs().max()):.2e}"
        assert torch.equal(re4, re) and torch.equal(im4, im), f"cqt_device in passes, chunk {chunk}"
    # ragged tail + padded row stride
    wide = torch.zeros((3, n + 77), dtype=torch.float32, device="cuda")
    wide[:, :n - 13] = xd[:3, :n - 13]
    re2, im2 = o.cqt_device(wide[:, :n - 13])
    torch.cuda.synchronize()
    assert np.array_equal(np.swapaxes(re2.cpu().numpy() + 1j * im2.cpu().numpy(), -1, -2)[1], o.cqt(x[1, :n - 13]))


def test_cqt_batch_host():
    import ctypes
    rng = np.random.default_rng(34)
    x = (0.1 * rng.standard_normal((3, 9000))).astype(np.float32)
    o = af.CQT(num=48, samplate=22050, low_fre=55.0, bin_per_octave=12)
    t = o.cal_time_length(9000)
    re = np.zeros((3, t, 48), np.float32)
    im = np.zeros_like(re)
    fn = o._lib.cqtObj_cqtBatch
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    assert fn(o._obj, x.ctypes.data, 3, 9000, re.ctypes.data, im.ctypes.data) == 0
    for i in range(3):
        assert np.array_equal(np.swapaxes(re[i] + 1j * im[i], 0, 1), o.cqt(x[i]))


def test_cepstrogram_batch_device_equals_loop():
    torch = _torch()
    rng = np.random.default_rng(35)
    x = (0.1 * rng.standard_normal((3, 12000))).astype(np.float32)
    # n_fft 256: both entry points run the size-generic kernel -- the same bits; n_fft 1024: the batched call runs the wave kernel
    # (round 6), the one-clip call the size-generic one -- two float32 evaluations of one chain
    for r2, hop in ((8, 64), (10, 256)):
        o = af.Cepstrogram(radix2_exp=r2, samplate=16000, window_type=af.WindowType.HANN, slide_length=hop)
        outs = o.cepstrogram_device(torch.from_numpy(x).cuda(), cep_num=6)
        torch.cuda.synchronize()
        for i in range(3):
            loop = o.cepstrogram(x[i], cep_num=6)
            for k in range(3):
                got = outs[k][i].cpu().numpy().T
                if r2 == 8:
                    assert np.array_equal(got, loop[k]), f"clip {i} output {k}"
                else:
                    assert_parity(got, loop[k], tol=2e-5, what=f"n_fft 1024 clip {i} output {k}: wave kernel vs size-generic kernel")


def test_xxcc_batch_host():
    import ctypes
    rng = np.random.default_rng(36)
    m = np.abs(rng.standard_normal((700, 128))).astype(np.float32)
    o = af.XXCC(128)
    out = np.zeros((700, 13), np.float32)
    fn = o._lib.xxccObj_xxccBatch
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    assert fn(o._obj, m.ctypes.data, 700, 13, None, out.ctypes.data) == 0
    assert np.array_equal(out, o.xxcc(m.T, 13).T)
    assert fn(o._obj, m.ctypes.data, 700, 200, None, out.ctypes.data) != 0


def test_cwt_register_fft_path_L131072():
    """radix2_exp 16 with padding (L = 2^17) runs the register-FFT inverse kernels
    (k_cwt_inv_rows512 / k_cwt_inv_cols256): parity with the compiled reference, plus the
    d/dt variant, plus batch == loop."""
    torch = _torch()
    rng = np.random.default_rng(41)
    x = (0.1 * rng.standard_normal((3, 65536))).astype(np.float32)
    kw = dict(num=12, radix2_exp=16, samplate=44100, low_fre=110.0, bin_per_octave=2)
    o = af.CWT(wavelet_type=af.WaveletContinueType.MORLET, scale_type=af.SpectralFilterBankScaleType.OCTAVE,
               is_padding=True, **kw)
    re, im = o.cwt_device(torch.from_numpy(x).cuda())
    torch.cuda.synchronize()
    got = re.cpu().numpy() + 1j * im.cpu().numpy()           # library order: descending frequency
    assert np.array_equal(got[1][::-1], o.cwt(x[1]))
    if ref.available():
        r = ref.RefCWT(wavelet_type=int(af.WaveletContinueType.MORLET),
                       scale_type=int(af.SpectralFilterBankScaleType.OCTAVE), is_padding=1, **kw)
        rre, rim = r.cwt(x[2])
        assert_parity(got[2], rre + 1j * rim, what="cwt L=2^17 vs reference")
        o.enable_det(True)
        dre, dim = o.cwt_device(torch.from_numpy(x[:1]).cuda(), det=True)
        torch.cuda.synchronize()
        rdr, rdi = r.cwt(x[0], det=True)
        assert_parity(dre.cpu().numpy()[0] + 1j * dim.cpu().numpy()[0], rdr + 1j * rdi, what="cwtDet L=2^17")
    # linearity at full size (size-independent property)
    y = (0.1 * rng.standard_normal(65536)).astype(np.float32)
    wa, wb = o.cwt(x[0]), o.cwt(y)
    ws = o.cwt((x[0] - 2 * y).astype(np.float32))
    assert np.abs(ws - (wa - 2 * wb)).max() <= 2e-5 * np.abs(ws).max()


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")
def test_one_launch_mel_mfcc_variants_against_reference():
    """afx_bftXxccBatchDevice on the one-launch path of k_stft_mel_v2 (cepstra of every 16 rows of a
    wave by MFMA): clip lengths whose frame counts leave partial 16-row blocks, every ccNum <= 16,
    with and without the mel output, several hops; the separate-kernel route (cubic-root
    rectification, ccNum > 16) must give reference results too."""
    torch = _torch()
    rng = np.random.default_rng(77)
    for hop, n, clips in ((512, 2048 + 512 * 36 + 5, 5), (512, 2048 + 512 * 15, 3), (300, 16000, 2), (1024, 40000, 4)):
        x = (0.1 * rng.standard_normal((clips, n))).astype(np.float32)
        x[0, : n // 2] = 0.0  # silent frames: the 1e-8 floor of the rectification
        xd = torch.from_numpy(x).cuda()
        bft = af.BFT(128, radix2_exp=11, samplate=16000, low_fre=0.0, high_fre=8000.0, slide_length=hop,
                     scale_type=af.SpectralFilterBankScaleType.MEL, data_type=af.SpectralDataType.POWER)
        bft.set_result_type(1)
        xx = af.XXCC(128)
        rb = ref.RefBFT(128, 11, samplate=16000, low_fre=0.0, high_fre=8000.0, window_type=1, slide_length=hop,
                        scale_type=2, style_type=0, normal_type=0, data_type=0)
        rb.set_result_type(1)
        rc = ref.RefXXCC(128)
        rmel = np.stack([rb.bft(x[i])[0] for i in range(clips)])
        for cc_num, want_mel, rect in ((13, True, 0), (16, False, 0), (1, True, 0), (7, False, 0), (20, True, 0), (13, True, 1)):
            mel, cc = af.mel_mfcc_device(bft, xx, xd, cc_num, rectify_type=af.CepstralRectifyType(rect),
                                         want_mel=want_mel)
            torch.cuda.synchronize()
            want = np.stack([rc.xxcc(rmel[i], cc_num, rect) for i in range(clips)])
            assert_parity(cc.cpu().numpy(), want, what=f"mfcc hop{hop} cc{cc_num} mel{want_mel} rect{rect}")
            if want_mel:
                assert_parity(mel.cpu().numpy(), rmel, what=f"mel hop{hop}")


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("r2", [9, 10, 11, 12])
def test_one_launch_mel_mfcc_at_every_fused_size(r2):
    """mel + MFCC in ONE launch at the shapes the reference defaults to -- n_fft 512 / 1024 / 2048 / 4096
    (python/audioflux/core.py:511-512: radix2_exp 12), mel banks of any num <= 128 that is a multiple of 4 incl. the split band
    plans (mel-40 / 64 / 80), log and cube-root rectification (xxcc_algorithm.c:124-137), hop N/4 and an odd hop; against
    the compiled reference at plain 1e-5, and the route is asserted (afx_bftXxccOneLaunchCount)."""
    torch = _torch()
    import ctypes
    lib = af.get_lib()
    lib.afx_bftXxccOneLaunchCount.restype = ctypes.c_longlong
    rng = np.random.default_rng(600 + r2)
    n_fft = 1 << r2
    for num, hop, clips, rects in ((128, n_fft // 4, 3, (0, 1)), (64, n_fft // 4, 2, (0,)), (40, n_fft // 4, 2, (0, 1)), (80, 3 * n_fft // 8 + 4, 2, (0,)),
                                   (20, n_fft // 4, 2, (0,))):
        n = n_fft + hop * 37 + 3
        x = (0.1 * rng.standard_normal((clips, n))).astype(np.float32)
        x[0, : n // 2] = 0.0  # silent frames: the 1e-8 floor of the log, 0^(1/3)
        xd = torch.from_numpy(x).cuda()
        bft = af.BFT(num, radix2_exp=r2, samplate=16000, low_fre=0.0, high_fre=8000.0, slide_length=hop,
                     scale_type=af.SpectralFilterBankScaleType.MEL, data_type=af.SpectralDataType.POWER)
        bft.set_result_type(1)
        xx = af.XXCC(num)
        rb = ref.RefBFT(num, r2, samplate=16000, low_fre=0.0, high_fre=8000.0, window_type=1, slide_length=hop,
                        scale_type=2, style_type=0, normal_type=0, data_type=0)
        rb.set_result_type(1)
        rc = ref.RefXXCC(num)
        rmel = np.stack([rb.bft(x[i])[0] for i in range(clips)])
        kind = lib.bftObj_fusedPlanKind(bft._obj)
        for rect in rects:
            for cc_num, want_mel in ((13, True), (16, False), (5, True)):
                if cc_num > num:
                    continue
                before = lib.afx_bftXxccOneLaunchCount()
                mel, cc = af.mel_mfcc_device(bft, xx, xd, cc_num, rectify_type=af.CepstralRectifyType(rect), want_mel=want_mel)
                torch.cuda.synchronize()
                if kind:  # every fused plan carries the cepstra
                    assert lib.afx_bftXxccOneLaunchCount() == before + 1, f"n_fft {n_fft} num {num}: plan kind {kind} took two launches"
                want = np.stack([rc.xxcc(rmel[i], cc_num, rect) for i in range(clips)])
                assert_parity(cc.cpu().numpy(), want, what=f"mfcc n_fft{n_fft} num{num} hop{hop} cc{cc_num} rect{rect} kind{kind}")
                if want_mel:
                    assert_parity(mel.cpu().numpy(), rmel, what=f"mel n_fft{n_fft} num{num} hop{hop}")
        assert kind, f"mel-{num} at n_fft {n_fft}: no fused plan"


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("r2", [9, 10, 11, 12])
def test_one_launch_mel_mfcc_edge_shapes(r2):
    """the same call at the edges of its shapes: ONE frame per clip over 70 clips (every wave's block is a partial block of one row,
    a clip boundary behind every frame), two frames, the smallest bank the block takes (num = 4) and a bank it does not take
    (num = 30: not a multiple of 4 -- the cepstra come from the second kernel, same results); ccNum = num."""
    torch = _torch()
    rng = np.random.default_rng(700 + r2)
    n_fft, hop = 1 << r2, (1 << r2) // 4
    for num, frames, clips, cc_num in ((128, 1, 70, 13), (128, 2, 5, 16), (4, 19, 3, 4), (30, 18, 2, 13), (64, 1, 1, 1)):
        n = n_fft + hop * (frames - 1)
        x = (0.1 * rng.standard_normal((clips, n))).astype(np.float32)
        xd = torch.from_numpy(x).cuda()
        bft = af.BFT(num, radix2_exp=r2, samplate=16000, low_fre=0.0, high_fre=8000.0, slide_length=hop,
                     scale_type=af.SpectralFilterBankScaleType.MEL, data_type=af.SpectralDataType.POWER)
        bft.set_result_type(1)
        xx = af.XXCC(num)
        rb = ref.RefBFT(num, r2, samplate=16000, low_fre=0.0, high_fre=8000.0, window_type=1, slide_length=hop,
                        scale_type=2, style_type=0, normal_type=0, data_type=0)
        rb.set_result_type(1)
        rc = ref.RefXXCC(num)
        rmel = np.stack([rb.bft(x[i])[0] for i in range(clips)])
        assert rmel.shape[1] == frames
        for rect in (0, 1):
            mel, cc = af.mel_mfcc_device(bft, xx, xd, cc_num, rectify_type=af.CepstralRectifyType(rect))
            torch.cuda.synchronize()
            want = np.stack([rc.xxcc(rmel[i], cc_num, rect) for i in range(clips)])
            assert_parity(cc.cpu().numpy(), want, what=f"mfcc n_fft{n_fft} num{num} frames{frames} clips{clips} cc{cc_num} rect{rect}")
            assert_parity(mel.cpu().numpy(), rmel, what=f"mel n_fft{n_fft} num{num} frames{frames}")


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("r2,clips", [(9, 64), (10, 64), (12, 48)])
def test_one_launch_mel_mfcc_long_runs_per_wave(r2, clips):
    """the same at a size where every wave walks 30-40 frames (332 000 frames at n_fft 512): whole 16-row blocks behind each other
    (afx_ccblock.h).  Every frame against the two-launch route (k_cepstrum_mfma over the same mel rows), two clips against the compiled reference."""
    torch = _torch()
    n_fft, hop = 1 << r2, (1 << r2) // 4
    n = n_fft + hop * 5196
    gen = torch.Generator(device="cuda").manual_seed(900 + r2)
    xd = 0.1 * torch.randn((clips, n), generator=gen, device="cuda", dtype=torch.float32)
    bft = af.BFT(128, radix2_exp=r2, samplate=16000, low_fre=0.0, high_fre=8000.0, slide_length=hop,
                 scale_type=af.SpectralFilterBankScaleType.MEL, data_type=af.SpectralDataType.POWER)
    bft.set_result_type(1)
    xx = af.XXCC(128)
    mel, cc = af.mel_mfcc_device(bft, xx, xd, 13)
    two = xx.xxcc_device(mel.reshape(-1, 128), 13)
    torch.cuda.synchronize()
    got, want = cc.reshape(-1, 13), two.reshape(-1, 13)
    peak = float(want.abs().max())
    assert float((got - want).abs().max()) <= 2e-6 * peak, "one launch vs bank kernel + cepstrum kernel"
    for i in (0, clips - 1):
        rmel, rcc = ref.mel_mfcc(xd[i:i + 1].cpu().numpy(), radix2_exp=r2, hop=hop)
        assert_parity(mel[i].cpu().numpy(), rmel[0], what=f"mel n_fft{n_fft} clip {i}")
        assert_parity(cc[i].cpu().numpy(), rcc[0], what=f"mfcc n_fft{n_fft} clip {i}")


def test_objects_are_usable_from_other_threads():
    """HIP's current device is per thread: an object built on one thread computes on another, and a
    thread that never touched the library builds its own (device binding per entry point)"""
    import threading
    x = (0.1 * np.random.default_rng(5).standard_normal(30000)).astype(np.float32)
    o = af.BFT(128, radix2_exp=11, samplate=16000, low_fre=0.0, high_fre=8000.0, slide_length=512,
               scale_type=af.SpectralFilterBankScaleType.MEL, data_type=af.SpectralDataType.POWER)
    want = o.bft(x, result_type=1)
    out = {}

    def worker():
        out["shared"] = o.bft(x, result_type=1)
        mine = af.BFT(128, radix2_exp=11, samplate=16000, low_fre=0.0, high_fre=8000.0, slide_length=512,
                      scale_type=af.SpectralFilterBankScaleType.MEL, data_type=af.SpectralDataType.POWER)
        out["own"] = mine.bft(x, result_type=1)
        c = af.CQT(num=84, samplate=32000)
        out["cqt"] = c.cqt(x)

    th = threading.Thread(target=worker)
    th.start()
    th.join()
    assert np.array_equal(out["shared"], want) and np.array_equal(out["own"], want)
    assert np.array_equal(out["cqt"], af.CQT(num=84, samplate=32000).cqt(x))


@pytest.mark.parametrize("ordinal", ["0", "99"])
def test_afx_device_environment_switch(ordinal):
    """AFX_DEVICE selects the default device of a process (read once, at the library's first call); an ordinal the box does
    not have falls back to device 0 instead of failing.  A fresh process each: the switch is latched."""
    import subprocess, sys
    code = ("import numpy as np, audioflux_amd as af\n"
            "x = (0.1 * np.random.default_rng(1).standard_normal(4096)).astype(np.float32)\n"
            "o = af.BFT(32, radix2_exp=10, samplate=16000, slide_length=256)\n"
            "y = o.bft(x)\n"
            "assert y.shape[0] == 32 and np.all(np.isfinite(y)) and float(np.abs(y).max()) > 0\n"
            "print('ok', y.shape)\n")
    env = dict(os.environ, AFX_DEVICE=ordinal)
    r = subprocess.run([sys.executable, "-c", code], cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-1500:]
