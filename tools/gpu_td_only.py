import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audioflux_amd as af
o = af.CWT(num=84, radix2_exp=16, samplate=44100, low_fre=32.703, bin_per_octave=12, wavelet_type=af.WaveletContinueType.MORLET,
           scale_type=af.SpectralFilterBankScaleType.OCTAVE, is_padding=True)
g = torch.Generator(device="cuda").manual_seed(3)
x = 0.1 * torch.randn((64, 65536), device="cuda", generator=g)
os.environ["AFX_CWT_TDONLY"] = "1"
for rep in range(3):
    re = torch.full((64, 84, 65536), float("nan"), device="cuda")
    im = torch.full((64, 84, 65536), float("nan"), device="cuda")
    guard = torch.full((1 << 24,), float("nan"), device="cuda")
    for c0 in (0, 32):
        o.cwt_device(x[c0:c0 + 32], re[c0:c0 + 32], im[c0:c0 + 32])
    torch.cuda.synchronize()
    wr = ~torch.isnan(re)
    rows = wr.any(dim=2).any(dim=0).nonzero().flatten().tolist()
    full = wr.all(dim=2).all(dim=0).nonzero().flatten().tolist()
    print("rep", rep, "rows touched", rows, "rows fully written", len(full), "guard touched", int((~torch.isnan(guard)).sum()))
    part = [r for r in rows if r not in full]
    for r in part[:6]:
        w = wr[:, r]
        print("   partial row", r, "elements", int(w.sum()), "chunks", w.any(dim=1).nonzero().flatten().tolist()[:10])
