"""Dense filter-bank route at the headline shape: gammatone-128 (ERB scale) at n_fft 2048 / hop 512,
1000 x 30 s @ 16 kHz (934 000 frames): afxk_stft2k (the headline kernel's transform storing its spectrum rows) -> pitched [T,F]
scratch -> k_gemm_bank_bf16x3 (128 x 128 tiles of v_mfma_f32_32x32x16_bf16, three bf16 words per operand, bank prepared once).
Prints frames/s; run under rocprofv3 (tools/gpu_dense.sh) for the kernel split and SQ_VALU_MFMA_BUSY_CYCLES.
   python tools/bench_dense.py [clips] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audioflux_amd as af

clips = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
x = 0.1 * torch.randn((clips, 480000), device="cuda")
o = af.BFT(128, radix2_exp=11, samplate=16000, low_fre=0.0, high_fre=8000.0, slide_length=512,
           scale_type=af.SpectralFilterBankScaleType.ERB, style_type=af.SpectralFilterBankStyleType.GAMMATONE,
           data_type=af.SpectralDataType.POWER)
o.set_result_type(1)
assert o.fused_plan_kind() == 0
T = o.cal_time_length(480000)
out = torch.empty((clips, T, 128), device="cuda")
for _ in range(2):
    o.bft_device(x, out_real=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(steps):
    o.bft_device(x, out_real=out)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / steps
frames = clips * T
flop = 2.0 * frames * 1025 * 128
print(f"dense gammatone-128 n_fft 2048: {ms:.3f} ms per {frames} frames = {frames / ms / 1e3:.1f} M frames/s; "
      f"GEMM work {flop / 1e9:.1f} GFLOP per step (time split: see the kernel trace)", flush=True)
