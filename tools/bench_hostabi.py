"""PCIe-inclusive rate of the host-pointer entry points: bftObj_bftBatch on pageable numpy
arrays, 200 clips x 30 s -- first call (the object's device buffers and the result's pages are new), then the steady
state of a caller that loops (same buffers) -- and the legacy one-clip bftObj_bft + xxccObj_xxcc loop.
Three callers: the first call of an object, a caller that loops over the same buffers, a data loader that brings new
arrays every call (profiles/r04_hostabi.txt)."""
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
frames = clips * 934
t0 = time.perf_counter()
mel = bft.bft_batch(x, result_type=1)
t1 = time.perf_counter()
print(f"bftObj_bftBatch (host in, host out), first call: {frames / (t1 - t0) / 1e6:.1f} M frames/s "
      f"({(x.nbytes + mel.nbytes) / (t1 - t0) / 1e9:.1f} GB/s over PCIe)")
ts = []
ref = mel.copy()
for i in range(6):
    t0 = time.perf_counter()
    bft.bft_batch(x, result_type=1, out=mel)
    ts.append(time.perf_counter() - t0)
assert np.array_equal(ref, mel)
best, med = min(ts), sorted(ts)[len(ts) // 2]
print(f"bftObj_bftBatch (host in, host out), looping caller: median {frames / med / 1e6:.1f} M frames/s "
      f"({(x.nbytes + mel.nbytes) / med / 1e9:.1f} GB/s over PCIe), best {frames / best / 1e6:.1f} M frames/s")
ts = []
for i in range(5):   # a data loader: new arrays every call (their pages written once, never seen by the runtime)
    xi = x + np.float32(i)
    oi = np.empty_like(mel)
    oi[:] = 0
    t0 = time.perf_counter()
    bft.bft_batch(xi, result_type=1, out=oi)
    ts.append(time.perf_counter() - t0)
    del xi, oi
med = sorted(ts)[len(ts) // 2]
print(f"bftObj_bftBatch (host in, host out), new arrays every call: median {frames / med / 1e6:.1f} M frames/s "
      f"({(x.nbytes + mel.nbytes) / med / 1e9:.1f} GB/s over PCIe), worst {frames / max(ts) / 1e6:.1f}")
t0 = time.perf_counter()
for i in range(20):
    m = bft.bft(x[i], result_type=1)
    c = xx.xxcc(m, 13)
t1 = time.perf_counter()
print(f"legacy one-clip loop (bftObj_bft + xxccObj_xxcc): {20 * 934 / (t1 - t0) / 1e6:.2f} M frames/s")
