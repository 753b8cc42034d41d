"""complex-result BFT (bftObj_setResultType 0, the reference wrapper's default) through the
device-resident batch call: frames/s with the fused kernel and with AFX_NO_FUSED=1"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audioflux_amd as af
R = int(sys.argv[1]) if len(sys.argv) > 1 else 11
x = 0.1 * torch.randn((200, 480000), device="cuda")
bft = af.BFT(128, radix2_exp=R, samplate=16000, low_fre=0.0, high_fre=8000.0, slide_length=(1 << R) // 4,
             scale_type=af.SpectralFilterBankScaleType.MEL, data_type=af.SpectralDataType.POWER)
bft.set_result_type(0)
re, im = bft.bft_device(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    bft.bft_device(x, out_real=re, out_imag=im)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"fused={'0' if os.environ.get('AFX_NO_FUSED') else '1'} complex mel: {ms:.3f} ms, {re.shape[0]*re.shape[1]/ms/1e3:.1f} M frames/s (n_fft {1 << R}), checksum {float(re.abs().sum()):.6e} {float(im.abs().sum()):.6e}")
