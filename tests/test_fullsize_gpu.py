"""Parity at BASELINE.json's full sizes.  The compiled reference cannot process whole
corpora in test time, so each configuration is checked through (a) size-independent
properties over the WHOLE batch and (b) the compiled reference on clips sampled from it.

cfg 2: 1000 x 30 s @ 16 kHz, n_fft 2048, hop 512, mel-128 + MFCC-13 (934 000 frames)
cfg 4: CWT morlet 84 scales on 2^16-sample chunks @ 44.1 kHz, padded (L = 2^17)
cfg 5: CQT 84 bins + chroma, 30 s @ 44.1 kHz clips (125 = one GPU's share of the 8-GPU run)
"""
import pytest

import audioflux_amd as af
from oracle import ref
from tests.conftest import assert_parity

pytestmark = pytest.mark.gpu


def test_cfg2_full_corpus_shift_property_and_sampled_reference():
    import torch
    clips, n, hop, t = 1000, 480000, 512, 934
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.empty((clips, n), dtype=torch.float32, device="cuda")
    x[0::2] = 0.1 * torch.randn((clips // 2, n), device="cuda", generator=g)
    # odd clips: the even clip advanced by one hop (new samples at the tail), so that
    # frame t of clip 2i+1 holds exactly the samples of frame t+1 of clip 2i
    x[1::2, : n - hop] = x[0::2, hop:]
    x[1::2, n - hop:] = 0.1 * torch.randn((clips // 2, hop), device="cuda", generator=g)
    bft = af.BFT(128, radix2_exp=11, samplate=16000, low_fre=0.0, high_fre=8000.0, slide_length=hop,
                 scale_type=af.SpectralFilterBankScaleType.MEL, data_type=af.SpectralDataType.POWER)
    bft.set_result_type(1)
    xx = af.XXCC(128)
    mel, cc = af.mel_mfcc_device(bft, xx, x, 13)
    torch.cuda.synchronize()
    assert mel.shape == (clips, t, 128) and cc.shape == (clips, t, 13)
    assert bool(torch.isfinite(mel).all()) and bool(torch.isfinite(cc).all())
    # (a) shift property, bit-exact, over the whole corpus: the same samples give the same
    #     bits wherever the frame sits in its clip (exercises the register re-use of the
    #     overlapping frames at every position)
    assert torch.equal(mel[1::2, : t - 1], mel[0::2, 1:])
    assert torch.equal(cc[1::2, : t - 1], cc[0::2, 1:])
    # (a') MFCC of the one-launch call vs the separate cepstral kernel on the mel it wrote: same DCT rows,
    #      log10 by v_log_f32 * log10(2) there, log10f here -- 1e-6 of the peak over the whole corpus
    sep = xx.xxcc_device(mel, 13)
    assert float((cc - sep).abs().amax() / sep.abs().amax()) <= 1e-6
    # (b) compiled reference on clips sampled across the batch (first, last, middle, an odd one)
    if ref.available():
        for i in (0, 999, 500, 777):
            rmel, rcc = ref.mel_mfcc(x[i:i + 1].cpu().numpy())
            assert_parity(mel[i].cpu().numpy(), rmel[0], what=f"cfg2 mel clip {i}")
            assert_parity(cc[i].cpu().numpy(), rcc[0], what=f"cfg2 mfcc clip {i}")


def test_cfg4_cwt_chunks_linearity_and_sampled_reference():
    import torch
    chunks, r, num = 48, 16, 84
    o = af.CWT(num=num, radix2_exp=r, samplate=44100, low_fre=32.703, bin_per_octave=12,
               wavelet_type=af.WaveletContinueType.MORLET,
               scale_type=af.SpectralFilterBankScaleType.OCTAVE, is_padding=True)
    g = torch.Generator(device="cuda").manual_seed(3)
    x = 0.1 * torch.randn((chunks, 1 << r), device="cuda", generator=g)
    # last third = linear combinations of the first two thirds
    k = chunks // 3
    x[2 * k:] = x[:k] - 2.0 * x[k:2 * k]
    re, im = o.cwt_device(x)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(re).all()) and bool(torch.isfinite(im).all())
    for got in (re, im):
        want = got[:k] - 2.0 * got[k:2 * k]
        err = (got[2 * k:] - want).abs().amax() / want.abs().amax()
        from tests.conftest import parity_log
        parity_log("cfg4 CWT linearity (a - 2b vs W(a) - 2 W(b))", float(err), 2e-6, "property: two float32 evaluations")
        assert float(err) <= 2e-6, f"linearity {float(err):.2e}"
    if ref.available():
        rr = ref.RefCWT(num=num, radix2_exp=r, samplate=44100, low_fre=32.703, bin_per_octave=12,
                        wavelet_type=int(af.WaveletContinueType.MORLET),
                        scale_type=int(af.SpectralFilterBankScaleType.OCTAVE), is_padding=1)
        i = 17
        rre, rim = rr.cwt(x[i].cpu().numpy())
        assert_parity(re[i].cpu().numpy() + 1j * im[i].cpu().numpy(), rre + 1j * rim, what="cfg4 chunk 17")


def test_cfg5_cqt_chroma_gpu_share_duplicates_and_sampled_reference():
    import torch
    clips, n, num = 125, 1323000, 84
    o = af.CQT(num=num, samplate=44100, low_fre=32.703, bin_per_octave=12,
               normal_type=af.SpectralFilterBankNormalType.AREA)
    g = torch.Generator(device="cuda").manual_seed(4)
    x = 0.1 * torch.randn((clips, n), device="cuda", generator=g)
    x[100:] = x[:25]                      # duplicates far apart in the batch
    re, im = o.cqt_device(x)
    ch = o.chroma_device(re, im)
    torch.cuda.synchronize()
    t = o.cal_time_length(n)
    assert re.shape == (clips, t, num) and ch.shape == (clips, t, 12)
    assert bool(torch.isfinite(re).all()) and bool(torch.isfinite(ch).all())
    assert torch.equal(re[100:], re[:25]) and torch.equal(im[100:], im[:25]) and torch.equal(ch[100:], ch[:25])
    assert float(ch.amax()) <= 1.0 + 1e-6 and float(ch.amin()) >= 0.0   # MAX-normalised power
    if ref.available():
        rr = ref.RefCQT(num=num, samplate=44100, min_fre=32.703, bin_per_octave=12, normal_type=1)
        i = 61
        rre, rim = rr.cqt(x[i].cpu().numpy())
        assert_parity(re[i].cpu().numpy() + 1j * im[i].cpu().numpy(), rre + 1j * rim, what="cfg5 cqt clip 61")
        assert_parity(ch[i].cpu().numpy(), rr.chroma(rre, rim), what="cfg5 chroma clip 61")


def test_cfg2_spectrogram_object_equals_bft_and_stft_round_trip():
    """the sibling objects at the cfg-2 scale: (a) the mel spectrogram object runs the BFT execution
    plan, so on the full 1000-clip corpus its result equals bftObj's bit for bit; (b) STFT ->
    inverse STFT (weighted overlap-add) returns the framed span of 256 whole clips to 1e-5;
    (c) centre-padded frames: clip j and clip j advanced by one hop share their interior frames."""
    import torch
    clips, n, hop, nfft = 1000, 480000, 512, 2048
    g = torch.Generator(device="cuda").manual_seed(11)
    x = 0.1 * torch.randn((clips, n), device="cuda", generator=g)
    bft = af.BFT(128, radix2_exp=11, samplate=16000, low_fre=0.0, high_fre=8000.0, slide_length=hop,
                 scale_type=af.SpectralFilterBankScaleType.MEL, data_type=af.SpectralDataType.POWER)
    bft.set_result_type(1)
    spec = af.MelSpectrogram(num=128, samplate=16000, low_fre=0.0, high_fre=8000.0, radix2_exp=11, slide_length=hop)
    a, b = bft.bft_device(x), spec.spectrogram_device(x)
    torch.cuda.synchronize()
    assert a.shape == (clips, 934, 128) and torch.equal(a, b)
    del a, b
    s = af.STFT(radix2_exp=11, window_type=af.WindowType.HANN, slide_length=hop)
    xs = x[:256]
    re, im = s.stft_device(xs)
    y = s.istft_device(re, im)
    torch.cuda.synchronize()
    m = y.shape[1]
    err = (y[:, nfft:m - nfft] - xs[:, nfft:m - nfft]).abs().amax() / xs.abs().amax()
    assert float(err) <= 1e-5, f"round trip {float(err):.2e}"
    del re, im, y
    s.enable_padding(True)
    s.set_padding(af.PaddingPositionType.CENTER, af.PaddingModeType.REFLECT)
    xa = xs[:64]
    xb = torch.empty_like(xa)
    xb[:, : n - hop] = xa[:, hop:]
    xb[:, n - hop:] = 0.05
    ra, ia = s.stft_device(xa)
    rb, ib = s.stft_device(xb)
    torch.cuda.synchronize()
    t = ra.shape[1]
    # frames whose support lies inside both clips (away from the reflected edges) are the same samples
    assert torch.equal(rb[:, 2:t - 4], ra[:, 3:t - 3]) and torch.equal(ib[:, 2:t - 4], ia[:, 3:t - 3])
