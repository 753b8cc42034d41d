"""mel-128 power spectrogram through the device batch call at other transform sizes:
python tools/bench_nfft.py <radix2_exp> <hop>   (AFX_NO_FUSED=1: size-generic kernels)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audioflux_amd as af
r, hop = int(sys.argv[1]), int(sys.argv[2])
x = 0.1 * torch.randn((500, 480000), device="cuda")
bft = af.BFT(128, radix2_exp=r, samplate=16000, low_fre=0.0, high_fre=8000.0, slide_length=hop,
             scale_type=af.SpectralFilterBankScaleType.MEL, data_type=af.SpectralDataType.POWER)
bft.set_result_type(1)
out = bft.bft_device(x)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.4:   # clock warm-up: sustained clocks, not the first launches' boost
    for _ in range(10):
        bft.bft_device(x, out_real=out)
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
N = 40
e0.record()
for _ in range(N):
    bft.bft_device(x, out_real=out)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / N
frames = out.shape[0] * out.shape[1]
print(f"n_fft {1 << r} hop {hop} fused={'0' if os.environ.get('AFX_NO_FUSED') else '1'}: {ms:.3f} ms, "
      f"{frames / ms / 1e3:.1f} M frames/s, {frames * (4 * hop + 512) / ms / 1e6:.0f} GB/s algorithmic (0.4 s warm-up, {N} calls)")
