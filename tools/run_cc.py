"""a few calls of the one-launch mel + MFCC path on BASELINE cfg 2 (profiling target of tools/prof_cmd.sh)"""
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
cc = torch.empty((1000, 934, 13), device="cuda")
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    af.mel_mfcc_device(bft, xx, x, 13, out_mel=mel, out_cc=cc)
torch.cuda.synchronize()
print("done", float(cc.double().sum()))
