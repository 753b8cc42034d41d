"""220 dispatches of each fused STFT kernel that round 5 added or rewrote, after a clock warm-up: the profiling target of
`tools/prof_cmd.sh r05_secondary "" python tools/run_round5_kernels.py` (rocprofv3 --kernel-trace --stats; summary copied to
profiles/r05_rocprofv3_secondary_trace.txt).  500 clips x 30 s @ 16 kHz for the bank kernels, 64 clips for the spectrum stores."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audioflux_amd as af

N = 220
x = 0.1 * torch.randn((500, 480000), device="cuda")
xs = x[:64]
jobs = []
for r in (9, 10, 12):
    hop = (1 << r) // 4
    b = af.BFT(128, radix2_exp=r, samplate=16000, low_fre=0.0, high_fre=8000.0, slide_length=hop,
               scale_type=af.SpectralFilterBankScaleType.MEL, data_type=af.SpectralDataType.POWER)
    b.set_result_type(1)
    out = b.bft_device(x)
    jobs.append((f"mel-128 n_fft {1 << r} hop {hop}", (lambda b=b, out=out: b.bft_device(x, out_real=out)), out.shape[0] * out.shape[1]))
for r in (9, 10, 11):
    hop = (1 << r) // 4
    b = af.BFT(128, radix2_exp=r, samplate=16000, low_fre=0.0, high_fre=8000.0, slide_length=hop,
               scale_type=af.SpectralFilterBankScaleType.MEL, data_type=af.SpectralDataType.POWER)
    b.set_result_type(0)
    re, im = b.bft_device(xs)
    jobs.append((f"mel-128 complex n_fft {1 << r} hop {hop}", (lambda b=b, re=re, im=im: b.bft_device(xs, out_real=re, out_imag=im)), re.shape[0] * re.shape[1]))
for r in (9, 10, 11, 12):
    hop = (1 << r) // 4
    o = af.STFT(radix2_exp=r, window_type=af.WindowType.HANN, slide_length=hop)
    re, im = o.stft_device(xs)
    jobs.append((f"stft n_fft {1 << r} hop {hop}", (lambda o=o: o.stft_device(xs)), re.shape[0] * re.shape[1]))
torch.cuda.synchronize()
for name, fn, frames in jobs:
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(N):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / N
    print(f"{name}: {ms:.4f} ms per call, {frames / ms / 1e3:.1f} M frames/s ({N} calls)")
