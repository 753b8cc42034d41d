"""PCIe-inclusive rate of the host-pointer entry points: bftObj_bftBatch on
pageable numpy arrays, 200 clips x 30 s, and the legacy one-clip bftObj_bft + xxccObj_xxcc loop"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import audioflux_amd as af
clips, n = 200, 480000
x = (0.1 * np.random.default_rng(1).standard_normal((clips, n))).astype(np.float32)
bft = af.BFT(128, radix2_exp=11, samplate=16000, low_fre=0.0, high_fre=8000.0, slide_length=512,
             scale_type=af.SpectralFilterBankScaleType.MEL, data_type=af.SpectralDataType.POWER)
xx = af.XXCC(128)
bft.bft_batch(x[:4], result_type=1)
t0 = time.perf_counter()
mel = bft.bft_batch(x, result_type=1)
t1 = time.perf_counter()
frames = clips * 934
print(f"bftObj_bftBatch (host in, host out): {frames / (t1 - t0) / 1e6:.1f} M frames/s "
      f"({(x.nbytes + mel.nbytes) / (t1 - t0) / 1e9:.1f} GB/s over PCIe incl. pageable staging)")
t0 = time.perf_counter()
for i in range(20):
    m = bft.bft(x[i], result_type=1)
    c = xx.xxcc(m, 13)
t1 = time.perf_counter()
print(f"legacy one-clip loop (bftObj_bft + xxccObj_xxcc): {20 * 934 / (t1 - t0) / 1e6:.2f} M frames/s")
