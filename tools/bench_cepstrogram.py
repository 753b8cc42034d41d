"""Cepstrogram at the reference wrapper's defaults (radix2_exp 12, hop 1024) and at 2048/512, 1024/256, 512/128,
through the batched device call"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audioflux_amd as af
for r, hop in ((12, 1024), (11, 512), (10, 256), (9, 128)):
    o = af.Cepstrogram(radix2_exp=r, samplate=32000, window_type=af.WindowType.HANN, slide_length=hop)
    x = 0.1 * torch.randn((64, 480000), device="cuda")
    for _ in range(4):  # torch's allocator ends up with the output sets the loop alternates between: no hipMalloc in the timed region
        outs = o.cepstrogram_device(x, cep_num=4)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        outs = o.cepstrogram_device(x, cep_num=4)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    frames = outs[0].shape[0] * outs[0].shape[1]
    f = (1 << r) // 2 + 1
    print(f"cepstrogram n_fft {1 << r} hop {hop}: {ms:.3f} ms, {frames / ms / 1e3:.1f} M frames/s, "
          f"{frames * (4 * hop + 12 * f) / ms / 1e6:.0f} GB/s algorithmic (outputs torch.empty inside the call)")
