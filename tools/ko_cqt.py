#!/usr/bin/env python3
"""GPU: step time and shader clock of BASELINE cfg 5 (CQT + chroma, 125 clips) for the library named by AFX_LIB -- the
measurement behind the knock-out table of k_cqt_pyramid (make EXTRA=-DAFX_KO_CQT=<mask>, tools/gpu_ko_cqt.sh).
usage: tools/ko_cqt.py <label> [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audioflux_amd as af
from audioflux_amd.batch import ClockProbe
label = sys.argv[1] if len(sys.argv) > 1 else "shipped"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
clips, n = 125, 1323000
dev = torch.device("cuda", 0)
o = af.CQT(num=84, samplate=44100, low_fre=32.703, bin_per_octave=12, normal_type=af.SpectralFilterBankNormalType.AREA)
x = 0.1 * torch.randn((clips, n), device=dev)
T = o.cal_time_length(n)
re = torch.empty((clips, T, 84), device=dev); im = torch.empty_like(re); ch = torch.empty((clips, T, 12), device=dev)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.5:  # clock warm-up
    for _ in range(8):
        o.cqt_chroma_device(x, out_real=re, out_imag=im, out=ch)
    torch.cuda.synchronize()
probe = ClockProbe(torch, dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
probe.start()
e0.record()
for _ in range(steps):
    o.cqt_chroma_device(x, out_real=re, out_imag=im, out=ch)
e1.record()
clk = probe.stop()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / steps
cyc = ms * 1e-3 * clk["clock_mhz"] * 1e6
print(f"{label}: {ms:.4f} ms/step  clock {clk['clock_mhz']:.0f} MHz  {cyc / 1e6:.3f} M cycles/step  ({steps} steps)")
