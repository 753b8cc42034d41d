"""why is k_stft_mel_banded ~12 % slower inside bench.py's step than back to back?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audioflux_amd as af
x = 0.1 * torch.randn((1000, 480000), device="cuda")
bft = af.BFT(128, radix2_exp=11, samplate=16000, low_fre=0.0, high_fre=8000.0, slide_length=512,
             scale_type=af.SpectralFilterBankScaleType.MEL, data_type=af.SpectralDataType.POWER)
bft.set_result_type(1)
xx = af.XXCC(128)
mel = torch.empty((1000, 934, 128), device="cuda")
mel2 = torch.empty((1000, 934, 128), device="cuda")
cc = torch.empty((1000, 934, 13), device="cuda")
def timed(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn_mel(); e1.record(); fn_rest(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts), sorted(ts)[len(ts) // 2]
fn_mel = lambda: bft.bft_device(x, out_real=mel)
fn_rest = lambda: None
print("mel only, events per launch      ", timed(lambda: fn_mel()))
fn_rest = lambda: xx.xxcc_device(mel, 13, out=cc)
print("mel + cepstrum per step           ", timed(lambda: (fn_mel(), fn_rest())))
fn_rest = lambda: xx.xxcc_device(mel2, 13, out=cc)
print("mel + cepstrum on another buffer  ", timed(lambda: (fn_mel(), fn_rest())))
fn_rest = lambda: torch.cuda._sleep(200000)
print("mel + 200k-cycle sleep kernel     ", timed(lambda: (fn_mel(), fn_rest())))
