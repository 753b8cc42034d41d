import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import audioflux_amd as af
from oracle import restate
from tests import cases
c = cases.REASSIGN_CASES["all_order2"] if hasattr(cases, "REASSIGN_CASES") else None
x = cases.make_input(c["x"], c["samplate"])
n, hop = 256, 64
h, dh, th = restate.reassign_windows(restate.fft_window(c["window_type"], n), n)
for name, w in (("h", h), ("dh", dh), ("th", th)):
    o = af.STFT(radix2_exp=8, window_type=af.WindowType.RECT, slide_length=hop)
    o.use_window_data_arr(w.astype(np.float32))
    re, im = o.stft_full(x)
    want = restate.stft_full(x, n, hop, w.astype(np.float32))
    got = re + 1j * im
    d = np.abs(got - want)
    print(name, "peak-rel err %.3e" % (d.max() / np.abs(want).max()), "frames", got.shape, "worst frame", np.unravel_index(d.argmax(), d.shape),
          "per-frame worst rel to frame peak %.3e" % (d.max(axis=1) / np.abs(want).max(axis=1)).max())
