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


def test_cfg2_second_distribution_three_sines_plus_noise():
    """SURVEY 8d's second corpus for cfg 2: 1000 x 30 s of 0.3 sin 220 Hz + 0.2 sin 880 Hz + 0.1 sin 3520 Hz + 1e-3
    noise (seed 2).  Mel at the plain bar; the MFCC takes log10 of bands that hold only the 1e-3 noise floor next to
    tones 50-110 dB above it, where every float32 FFT (the reference's radix-2 one included) carries ~1e-7 of the
    frame's PEAK: bar = max(1e-5, 2 x the reference's own distance from a float64 evaluation), logged."""
    import numpy as np
    import torch
    from oracle import restate
    from tests.conftest import l2_rel, parity_log, peak_rel
    clips, n, hop, t = 1000, 480000, 512, 934
    g = torch.Generator(device="cuda").manual_seed(2)
    tt = torch.arange(n, device="cuda", dtype=torch.float64) / 16000.0
    tones = (0.3 * torch.sin(2 * np.pi * 220 * tt) + 0.2 * torch.sin(2 * np.pi * 880 * tt)
             + 0.1 * torch.sin(2 * np.pi * 3520 * tt)).to(torch.float32)
    x = tones[None, :] + 1e-3 * torch.randn((clips, n), device="cuda", generator=g)
    x[1::2] = x[0::2]                         # every clip twice: position independence over the whole corpus
    bft = af.BFT(128, radix2_exp=11, samplate=16000, low_fre=0.0, high_fre=8000.0, slide_length=hop,
                 scale_type=af.SpectralFilterBankScaleType.MEL, data_type=af.SpectralDataType.POWER)
    bft.set_result_type(1)
    xx = af.XXCC(128)
    mel, cc = af.mel_mfcc_device(bft, xx, x, 13)
    torch.cuda.synchronize()
    assert mel.shape == (clips, t, 128) and bool(torch.isfinite(mel).all()) and bool(torch.isfinite(cc).all())
    assert torch.equal(mel[1::2], mel[0::2]) and torch.equal(cc[1::2], cc[0::2])
    if ref.available():
        bank, _, _ = restate.mel_bank(128, 2048, 16000, 0.0, 8000.0)
        for i in (0, 998, 501):
            xi = x[i].cpu().numpy()
            rmel, rcc = ref.mel_mfcc(xi[None])
            fmel = restate.bft(xi.astype(np.float64), bank, 2048, 512)
            fcc = restate.xxcc(fmel)
            assert_parity(mel[i].cpu().numpy(), rmel[0], what=f"cfg2 tones mel clip {i}")
            got = cc[i].cpu().numpy()
            ref_d = max(peak_rel(rcc[0], fcc), l2_rel(rcc[0], fcc))
            for tag, other in (("reference", rcc[0]), ("float64", fcc)):
                d = max(peak_rel(got, other), l2_rel(got, other))
                bar = max(1e-5, 2.0 * ref_d)
                parity_log(f"cfg2 tones mfcc clip {i} vs {tag}", d, bar, "max(1e-5, 2 x reference-vs-float64)",
                           {"reference_vs_float64": ref_d})
                assert d <= bar, f"cfg2 tones mfcc clip {i} vs {tag}: {d:.3e} > {bar:.3e}"


def test_cfg4_whole_corpus_through_the_output_ring():
    """cfg 4 as bench.py runs it: 1000 clips x 7 chunks = 7000 chunks of 2^16 samples (the 7th zero padded from sample
    47 784 on), 32 chunks per device call into a two-slot output ring.  The second half of the corpus repeats the first
    3488 chunks (109 calls later: same ring slot, same position in its call): per-chunk checksums formed before the
    slot is overwritten must agree bit for bit, and chunks sampled at the corpus's ends, a ragged one and a repeated one
    meet the compiled reference."""
    import torch
    chunks, r, num, grp, rep = 7000, 16, 84, 32, 3488
    o = af.CWT(num=num, radix2_exp=r, samplate=44100, low_fre=32.703, bin_per_octave=12,
               wavelet_type=af.WaveletContinueType.MORLET,
               scale_type=af.SpectralFilterBankScaleType.OCTAVE, is_padding=True)
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.zeros((1000, 7 * 65536), device="cuda")
    x[:, :441000] = 0.1 * torch.randn((1000, 441000), device="cuda", generator=g)
    x = x.view(chunks, 65536)
    x[rep:2 * rep] = x[:rep].clone()
    ring = [(torch.empty((grp, num, 65536), device="cuda"), torch.empty((grp, num, 65536), device="cuda")) for _ in range(2)]
    sums = torch.empty((chunks, 2), device="cuda", dtype=torch.float64)
    picks = {0: None, 6: None, rep + 6: None, 3499: None, chunks - 1: None}   # 6, 6999: ragged (zero-padded) chunks
    k = 0
    for c0 in range(0, chunks, grp):
        n = min(grp, chunks - c0)
        re, im = ring[k & 1]
        o.cwt_device(x[c0:c0 + n], re[:n], im[:n])
        sums[c0:c0 + n, 0] = re[:n].double().sum(dim=(1, 2))
        sums[c0:c0 + n, 1] = (re[:n].double() ** 2 + im[:n].double() ** 2).sum(dim=(1, 2))
        for i in picks:
            if c0 <= i < c0 + n:
                picks[i] = (re[i - c0].cpu().numpy(), im[i - c0].cpu().numpy())
        k += 1
    torch.cuda.synchronize()
    assert bool(torch.isfinite(sums).all())
    bad = (sums[rep:2 * rep] != sums[:rep]).any(dim=1).nonzero().flatten()
    assert bad.numel() == 0, f"{bad.numel()} repeated chunks differ from their first copy, first: {bad[:8].tolist()} (mod 32: {(bad[:8] % 32).tolist()})"
    assert (picks[6][0] == picks[rep + 6][0]).all() and (picks[6][1] == picks[rep + 6][1]).all()
    if ref.available():
        rr = ref.RefCWT(num=num, radix2_exp=r, samplate=44100, low_fre=32.703, bin_per_octave=12,
                        wavelet_type=int(af.WaveletContinueType.MORLET),
                        scale_type=int(af.SpectralFilterBankScaleType.OCTAVE), is_padding=1)
        for i in (0, 6, 3499, chunks - 1):
            rre, rim = rr.cwt(x[i].cpu().numpy())
            assert_parity(picks[i][0] + 1j * picks[i][1], rre + 1j * rim, what=f"cfg4 ring chunk {i}")


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


def test_cfg5_fused_call_sampled_reference():
    """BASELINE cfg 5 through the ONE call bench.py times (cqt_chroma_device: the one-launch ladder with chroma-12 in its
    epilogue), one GPU's share of 125 clips: clips 0, 61 and 124 -- the first, a middle one and the last run of the last
    workgroup -- against the compiled reference, CQT and chroma, plain 1e-5."""
    import torch
    if not ref.available():
        pytest.skip("oracle/_ref not built")
    clips, n, num = 125, 1323000, 84
    o = af.CQT(num=num, samplate=44100, low_fre=32.703, bin_per_octave=12,
               normal_type=af.SpectralFilterBankNormalType.AREA)
    g = torch.Generator(device="cuda").manual_seed(4)
    x = 0.1 * torch.randn((clips, n), device="cuda", generator=g)
    re, im, ch = o.cqt_chroma_device(x)
    torch.cuda.synchronize()
    for i in (0, 61, 124):
        rr = ref.RefCQT(num=num, samplate=44100, min_fre=32.703, bin_per_octave=12, normal_type=1)
        rre, rim = rr.cqt(x[i].cpu().numpy())
        assert_parity(re[i].cpu().numpy() + 1j * im[i].cpu().numpy(), rre + 1j * rim, what=f"cfg5 fused call: cqt clip {i}")
        assert_parity(ch[i].cpu().numpy(), rr.chroma(rre, rim), what=f"cfg5 fused call: chroma clip {i}")


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
