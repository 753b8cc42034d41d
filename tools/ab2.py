"""in-process interleaved A/B of runtime switches (environment variables the library reads per
call) on BASELINE cfg 2: python tools/ab2.py "" "AFX_MEL_V1=1" ...
Every sample is a >= 0.4 s back-to-back loop (sustained clocks, not boost); columns: ms per call of
the mel kernel alone and of mel + MFCC (afx_bftXxccBatchDevice), median over the rounds."""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np
import torch
import audioflux_amd as af

cfgs = sys.argv[1:] or [""]
x = 0.1 * torch.randn((1000, 480000), device="cuda")
bft = af.BFT(128, radix2_exp=11, samplate=16000, low_fre=0.0, high_fre=8000.0, slide_length=512,
             scale_type=af.SpectralFilterBankScaleType.MEL, data_type=af.SpectralDataType.POWER)
bft.set_result_type(1)
xx = af.XXCC(128)
mel = torch.empty((1000, 934, 128), device="cuda")
cc = torch.empty((1000, 934, 13), device="cuda")


def setenv(cfg):
    for k in ("AFX_MEL_V1", "AFX_NO_FUSED_CC"):
        os.environ.pop(k, None)
    for kv in cfg.split():
        k, v = kv.split("=")
        os.environ[k] = v


def loop(fn, seconds=0.4):
    fn(); torch.cuda.synchronize()
    n = 0
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    while True:
        for _ in range(20):
            fn()
        n += 20
        if time.perf_counter() - t0 > seconds:
            break
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


res = {c: ([], []) for c in cfgs}
sums = {}
for rnd in range(5):
    for c in cfgs:
        setenv(c)
        res[c][0].append(loop(lambda: bft.bft_device(x, out_real=mel)))
        res[c][1].append(loop(lambda: af.mel_mfcc_device(bft, xx, x, 13, out_mel=mel, out_cc=cc)))
        sums[c] = (float(mel.double().sum()), float(cc.double().sum()))
for c in cfgs:
    a, b = np.median(res[c][0]), np.median(res[c][1])
    print(f"{c!r:28s} mel {a:.4f} ms ({934000 / a / 1e3:.1f} M frames/s)   mel+mfcc {b:.4f} ms ({934000 / b / 1e3:.1f} M frames/s)"
          f"   checksums {sums[c][0]:.6e} {sums[c][1]:.6e}", flush=True)
