"""STFT object through the batched device call: full complex spectrum (all n_fft bins), 64 clips x 30 s,
n_fft 2048 / hop 512 and n_fft 4096 / hop 1024; AFX_NO_STFT_WAVE=1 times the size-generic kernel"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import audioflux_amd as af

x = 0.1 * torch.randn((64, 480000), device="cuda")
for r, hop in ((11, 512), (12, 1024)):
    o = af.STFT(radix2_exp=r, window_type=af.WindowType.HANN, slide_length=hop)
    for _ in range(4):  # torch's allocator ends up with the two output pairs the loop alternates between: no hipMalloc in the timed region
        re, im = o.stft_device(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        re, im = o.stft_device(x)  # (allocates its outputs inside the timed call, from torch's cache)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    frames = re.shape[0] * re.shape[1]
    n = 1 << r
    print(f"stft n_fft {n} hop {hop}: {ms:.3f} ms, {frames / ms / 1e3:.1f} M frames/s, "
          f"{frames * (4 * hop + 8 * n) / ms / 1e6:.0f} GB/s algorithmic")
