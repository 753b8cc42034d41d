import os, sys
sys.path.insert(0, os.getcwd())
os.environ['AFX_DBG']='1'
import torch, audioflux_amd as af
x = 0.1*torch.randn((1000, 480000), device='cuda')
bft = af.BFT(128, radix2_exp=11, samplate=16000, low_fre=0.0, high_fre=8000.0, slide_length=512,
             scale_type=af.SpectralFilterBankScaleType.MEL, data_type=af.SpectralDataType.POWER)
bft.set_result_type(1)
out = torch.empty((1000, 934, 128), device='cuda')
for _ in range(3): bft.bft_device(x, out_real=out)
torch.cuda.synchronize()
os.environ['AFX_DBG_DUMP']='1'
bft.bft_device(x, out_real=out); torch.cuda.synchronize()
