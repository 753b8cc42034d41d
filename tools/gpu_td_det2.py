"""Determinism of the batched CWT on the device: 64 chunks as two calls, three runs, bitwise against the first run;
where rows differ, which (chunk, scale) tiles and by how much.  (Round 3: schedules that overlapped afx_cwt_td.hip
with the FFT-path kernels failed this; profiles/r03_cwt_td_schedules.txt.)   python tools/gpu_td_det2.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audioflux_amd as af
o = af.CWT(num=84, radix2_exp=16, samplate=44100, low_fre=32.703, bin_per_octave=12, wavelet_type=af.WaveletContinueType.MORLET,
           scale_type=af.SpectralFilterBankScaleType.OCTAVE, is_padding=True)
g = torch.Generator(device="cuda").manual_seed(3)
x = 0.1 * torch.randn((64, 65536), device="cuda", generator=g)
def run():
    re = torch.full((64, 84, 65536), float("nan"), device="cuda")
    im = torch.full((64, 84, 65536), float("nan"), device="cuda")
    for c0 in (0, 32):
        o.cwt_device(x[c0:c0 + 32], re[c0:c0 + 32], im[c0:c0 + 32])
    torch.cuda.synchronize()
    return re, im
gre, gim = run()
for rep in range(3):
    re, im = run()
    d = (re != gre) | (im != gim)
    print("rep", rep, "wrong elements", int(d.sum()), "scales", d.sum(dim=(0, 2)).nonzero().flatten().tolist())
    if not int(d.sum()):
        continue
    # per (chunk, scale): fraction of wrong elements, and where in time
    frac = d.float().mean(dim=2)
    cs = (frac > 0).nonzero()[:12].tolist()
    for c, s in cs:
        w = d[c, s].nonzero().flatten()
        seg = (int(w.min()), int(w.max()), int(w.numel()))
        # does the wrong data equal the golden data of another chunk (same scale)?
        best = None
        for c2 in range(64):
            if c2 == c: continue
            m = (re[c, s, w] == gre[c2, s, w]).float().mean().item()
            if best is None or m > best[1]: best = (c2, m)
        relerr = float(((re[c, s] - gre[c, s]).abs().max() / gre[c, s].abs().max()))
        print(f"   chunk {c} scale {s}: wrong n in [{seg[0]}, {seg[1]}] count {seg[2]}; max rel err {relerr:.2e}; best match with golden chunk {best[0]}: {best[1]:.2f}")
