"""mel-128 (AFX_BENCH_NUMS: other banks) + MFCC-13 through afx_bftXxccBatchDevice at n_fft 512 .. 4096 (hop N / 4, 500 x 30 s @ 16 kHz):
the bank kernel alone and with the cepstra (round 6: one launch at every fused size and plan)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audioflux_amd as af
x = 0.1 * torch.randn((500, 480000), device="cuda")
NUMS = [int(v) for v in os.environ.get("AFX_BENCH_NUMS", "128").split(",")]  # e.g. AFX_BENCH_NUMS=128,80,64,40 (split band plans)
for r, num in [(r, n) for r in (9, 10, 11, 12) for n in NUMS]:
    hop = (1 << r) // 4
    bft = af.BFT(num, radix2_exp=r, samplate=16000, low_fre=0.0, high_fre=8000.0, slide_length=hop,
                 scale_type=af.SpectralFilterBankScaleType.MEL, data_type=af.SpectralDataType.POWER)
    bft.set_result_type(1)
    xx = af.XXCC(num)
    mel, cc = af.mel_mfcc_device(bft, xx, x, 13)
    out = bft.bft_device(x)
    torch.cuda.synchronize()
    res = []
    for what in ("mel", "mel+mfcc"):
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3:
            for _ in range(5):
                if what == "mel": bft.bft_device(x, out_real=out)
                else: af.mel_mfcc_device(bft, xx, x, 13, out_mel=mel, out_cc=cc)
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            if what == "mel": bft.bft_device(x, out_real=out)
            else: af.mel_mfcc_device(bft, xx, x, 13, out_mel=mel, out_cc=cc)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        res.append(f"{what} {ms:.3f} ms = {mel.shape[0] * mel.shape[1] / ms / 1e3:.0f} M frames/s")
    print(f"n_fft {1 << r} hop {hop} mel-{num} (plan kind {af.get_lib().bftObj_fusedPlanKind(bft._obj)}): " + "; ".join(res))
