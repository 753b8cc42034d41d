"""AFX_STAGING=1 (afx_runtime.hip): host-pointer copies of at most 4 MB go through pinned slabs of the object's stream
-- uploads in pieces under the next piece's memcpy, downloads delivered at the stream synchronisation.  Off by default
(profiles/r05_legacy_phases.txt); the switch is read once per process, so each mode runs in a child interpreter and the
one-clip entry points of every object family must return the same bits in both."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import audioflux_amd as af
from tests import cases
x = cases.noise(11, 70000)
out = {}
b = af.BFT(128, radix2_exp=11, samplate=16000, low_fre=0.0, high_fre=8000.0, slide_length=512,
           scale_type=af.SpectralFilterBankScaleType.MEL, data_type=af.SpectralDataType.POWER)
for i in range(3):  # the slabs are re-used call after call
    out["bft%%d" %% i] = b.bft(x[i * 1000:], result_type=1)
out["bft_c"] = b.bft(x, result_type=0)
out["xxcc"] = af.XXCC(128).xxcc(out["bft0"], 13)
c = af.CQT(num=84, samplate=32000, low_fre=32.703, bin_per_octave=12)
q = c.cqt(x)
out["cqt"] = q
out["chroma"] = c.chroma(q)
w = af.CWT(num=40, radix2_exp=12, samplate=16000, low_fre=40.0, bin_per_octave=8)
out["cwt"] = w.cwt(x[:4096])
out["spec"] = af.MelSpectrogram(num=64, samplate=16000, radix2_exp=10).spectrogram(x)
np.savez(sys.argv[1], **{k: np.asarray(v) for k, v in out.items()})
print("OK")
"""


@pytest.mark.gpu
def test_staged_copies_return_the_same_bits(tmp_path):
    res = {}
    for mode in ("plain", "staged"):
        e = dict(os.environ, AFX_QUIET="1")
        e.pop("AFX_STAGING", None)
        if mode == "staged":
            e["AFX_STAGING"] = "1"
        path = str(tmp_path / f"{mode}.npz")
        r = subprocess.run([sys.executable, "-c", CHILD % ROOT, path], capture_output=True, text=True, env=e, timeout=600, cwd=ROOT)
        assert r.returncode == 0 and "OK" in r.stdout, (r.stdout + r.stderr)[-3000:]
        res[mode] = np.load(path)
    assert sorted(res["plain"].files) == sorted(res["staged"].files) and len(res["plain"].files) >= 8
    for k in res["plain"].files:
        a, b = res["plain"][k], res["staged"][k]
        assert a.shape == b.shape and np.isfinite(a).all() and np.abs(a).max() > 0, k
        assert np.array_equal(a, b), k
