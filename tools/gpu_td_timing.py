"""Phase times of k_cwt_td per tile (probe build with -DAFX_TD_TIMING, loaded through AFX_LIB): s_memtime sums over
all waves of the long (MAXK 1024) and short (MAXK 384) classes.  python tools/gpu_td_timing.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audioflux_amd as af
from audioflux_amd import _lib
lib = _lib.get_lib()
td_only = af.CWT(num=36, radix2_exp=16, samplate=44100, low_fre=32.703 * 16, bin_per_octave=12, wavelet_type=af.WaveletContinueType.MORLET,
                 scale_type=af.SpectralFilterBankScaleType.OCTAVE, is_padding=True)
x = 0.1 * torch.randn((224, 65536), device="cuda")
out = td_only.cwt_device(x)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 8)()
lib.afxk_cwt_td_timing(buf)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
td_only.cwt_device(x, out[0], out[1])
e1.record()
torch.cuda.synchronize()
lib.afxk_cwt_td_timing(buf)
v = list(buf)
print("call of 224 chunks: %.3f ms" % e0.elapsed_time(e1))
for name, o in (("long class (MAXK 1024)", 0), ("short class (MAXK 384)", 4)):
    a, b, c, tot = v[o:o + 4]
    print(f"{name}: window wait + split {a / tot:.3f}, K loop {b / tot:.3f}, epilogue {c / tot:.3f} of the waves' time; sum of wave times {tot / 1e8:.3f} s at 100 MHz")
