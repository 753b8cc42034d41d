"""STFT (full complex spectrum, hop N / 4) and inverse STFT (weighted overlap-add) of 64 clips x 30 s through the batched device calls,
n_fft 256 ... 8192: ms per call, frames/s, algorithmic GB/s (inputs read once + requested outputs written once)."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch, audioflux_amd as af
x = 0.1 * torch.randn((64, 480000), device="cuda")
for r in (8, 9, 10, 11, 12, 13):
    hop = (1 << r) // 4
    o = af.STFT(radix2_exp=r, window_type=af.WindowType.HANN, slide_length=hop)
    for _ in range(4):
        re, im = o.stft_device(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        re, im = o.stft_device(x)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    frames = re.shape[0] * re.shape[1]; n = 1 << r
    print(f"stft n_fft {n} hop {hop}: {ms:.3f} ms, {frames / ms / 1e3:.1f} M frames/s, {frames * (4 * hop + 8 * n) / ms / 1e6:.0f} GB/s algorithmic")
    if r <= 12:  # (the inverse: 8 N bytes of bins in, 4 hop out; the output's zero fill by torch is inside the call)
        y = o.istft_device(re, im, method_type=0)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            y = o.istft_device(re, im, method_type=0)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"istft n_fft {n} hop {hop}: {ms:.3f} ms, {frames / ms / 1e3:.1f} M frames/s, {frames * (8 * n + 4 * hop) / ms / 1e6:.0f} GB/s algorithmic")
        del y
    del re, im
