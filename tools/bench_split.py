"""n_fft 2048 / hop 512 and n_fft 4096 / hop 1024 with banks whose rows exceed the fused kernel's tap variants (split band
plans; AFX_NO_FUSED=1 gives the size-generic kernels for comparison), 500 clips x 30 s @ 16 kHz, real power results"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import audioflux_amd as af

x = 0.1 * torch.randn((500, 480000), device="cuda")
for r2, scale, num, sr in ((11, "MEL", 40, 16000), (11, "MEL", 64, 16000), (11, "MEL", 80, 22050),
                           (11, "BARK", 64, 16000), (11, "ERB", 64, 16000), (11, "MEL", 64, 44100),
                           (11, "MEL", 128, 16000), (12, "MEL", 80, 32000), (12, "MEL", 40, 16000),
                           (12, "BARK", 64, 32000), (12, "ERB", 64, 22050), (12, "MEL", 128, 32000)):
    o = af.BFT(num, radix2_exp=r2, samplate=sr, low_fre=0.0, high_fre=sr / 2.0, slide_length=(1 << r2) // 4,
               scale_type=getattr(af.SpectralFilterBankScaleType, scale), data_type=af.SpectralDataType.POWER)
    o.set_result_type(1)
    t = o.cal_time_length(480000)
    out = torch.empty((500, t, num), device="cuda")
    for _ in range(2):
        o.bft_device(x, out_real=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        o.bft_device(x, out_real=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"n_fft {1 << r2} {scale}-{num} @ {sr}: plan kind {o.fused_plan_kind()}, {ms:.3f} ms, {500 * t / ms / 1e3:.1f} M frames/s")
