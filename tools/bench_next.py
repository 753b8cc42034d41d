"""Throughput of the objects added after the SURVEY 8 (a)-(e) rows -- STFT / inverse STFT, the
spectrogram object (mel through the fused kernel, STFT-chroma, log-chroma) and PWT -- through the
batched device calls, inputs and outputs resident in HBM.  Prints one line per workload with the
algorithmic bytes (inputs read once + requested outputs written once) as GB/s."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import audioflux_amd as af


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    sr, n, hop = 16000, 2048, 512
    x = 0.1 * torch.randn((128, 30 * sr), device="cuda")
    s = af.STFT(radix2_exp=11, window_type=af.WindowType.HANN, slide_length=hop)
    s.enable_padding(True)
    s.set_padding(af.PaddingPositionType.CENTER, af.PaddingModeType.REFLECT)
    re, im = s.stft_device(x)
    frames = re.shape[0] * re.shape[1]
    ms = timed(lambda: s.stft_device(x))
    print(f"stft  n_fft {n} hop {hop} reflect pad, full spectrum: {ms:.3f} ms, {frames / ms / 1e3:.1f} M frames/s, "
          f"{frames * (4 * hop + 8 * n) / ms / 1e6:.0f} GB/s algorithmic (outputs torch.empty inside the call)")
    ms = timed(lambda: s.istft_device(re, im))
    print(f"istft n_fft {n} hop {hop} weighted overlap-add: {ms:.3f} ms, {frames / ms / 1e3:.1f} M frames/s, "
          f"{frames * (8 * n + 4 * hop) / ms / 1e6:.0f} GB/s algorithmic (output torch.zeros inside the call)")
    del re, im
    x = 0.1 * torch.randn((1000, 30 * sr), device="cuda")
    for name, cls, kw in (("mel-128 spectrogram object", af.MelSpectrogram, dict(num=128)),
                          ("bark-64 spectrogram object", af.BarkSpectrogram, dict(num=64))):
        o = cls(samplate=sr, radix2_exp=11, slide_length=hop, **kw)
        out = o.spectrogram_device(x)
        frames = out.shape[0] * out.shape[1]
        ms = timed(lambda: o.spectrogram_device(x, out=out))
        print(f"{name} n_fft {n} hop {hop}: {ms:.3f} ms, {frames / ms / 1e3:.1f} M frames/s, "
              f"{frames * (4 * hop + 4 * o.num) / ms / 1e6:.0f} GB/s algorithmic")
    xc = x[:200]
    for name, scale, kw in (("stft-chroma-12", af.SpectralFilterBankScaleType.CHROMA, dict(low_fre=0.0)),
                            ("log-chroma-12", af.SpectralFilterBankScaleType.OCTAVE_CHROMA, dict(low_fre=32.703, high_fre=4000.0))):
        o = af.Spectrogram(num=12, samplate=sr, radix2_exp=11, slide_length=hop, filter_bank_type=scale, **kw)
        out = o.spectrogram_device(xc)
        frames = out.shape[0] * out.shape[1]
        ms = timed(lambda: o.spectrogram_device(xc, out=out))
        print(f"{name} spectrogram object n_fft {n} hop {hop}: {ms:.3f} ms, {frames / ms / 1e3:.1f} M frames/s, "
              f"{frames * (4 * hop + 4 * 12) / ms / 1e6:.0f} GB/s algorithmic")
    del x, xc
    p = af.PWT(num=84, radix2_exp=16, samplate=44100, low_fre=32.703, is_padding=True)
    xp = 0.1 * torch.randn((32, 1 << 16), device="cuda")
    p.pwt_device(xp)
    ms = timed(lambda: p.pwt_device(xp), reps=3)
    print(f"pwt octave-84, 2^16-sample chunks, padded: {ms:.3f} ms / 32 chunks, {32 / ms * 1e3:.0f} chunks/s, "
          f"{32 * 65536 * (4 + 8 * 84) / ms / 1e6:.0f} GB/s algorithmic (outputs torch.empty inside the call)")


if __name__ == "__main__":
    main()
