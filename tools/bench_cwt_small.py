"""CWT at the reference wrapper's default size (num 84, radix2_exp 12, padded: L = 8192)
through the batched device call"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audioflux_amd as af
r = int(sys.argv[1]) if len(sys.argv) > 1 else 12
chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 256
o = af.CWT(num=84, radix2_exp=r, samplate=32000, low_fre=32.703, bin_per_octave=12,
           wavelet_type=af.WaveletContinueType.MORLET, scale_type=af.SpectralFilterBankScaleType.OCTAVE)
x = 0.1 * torch.randn((chunks, 1 << r), device="cuda")
re = torch.empty((chunks, 84, 1 << r), device="cuda")
im = torch.empty_like(re)
o.cwt_device(x, re, im)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    o.cwt_device(x, re, im)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
samples = chunks * (1 << r)
print(f"CWT 84 scales, 2^{r}-sample chunks x {chunks}: {ms:.3f} ms, {samples / ms / 1e6:.2f} G samples/s, "
      f"{samples * 676 / ms / 1e6:.0f} GB/s algorithmic")
