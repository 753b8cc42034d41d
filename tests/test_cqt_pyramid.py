"""GPU: the level signals of the one-launch CQT ladder (k_cqt_pyramid) against the 2:1 resampler in float64.

The matrix-core resampler inside the ladder replaces /root/reference/src/dsp/resample_algorithm.c:430-521; its output never
reaches a caller (level rings in the L2), so the final CQT rows were its only witnesses.  Here the rings of the first
workgroup are copied out (afx_cqt_pyramid_rings) after a launch whose runs are cut short (AFX_CQT_PYR_TILES, read when the
object is created), so that the rings hold signal: every level against restate.decimate2 in float64 (tests/cqt_rings.py).
Child processes: the switch must be in the environment when the object is created, and no other test should see it."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _child(n, seed, tiles, slack):
    import ctypes as C

    import numpy as np

    import audioflux_amd as af
    from tests import cqt_rings
    x = cqt_rings.ladder_input(n, seed)
    o = af.CQT(num=84, samplate=44100, low_fre=32.703, bin_per_octave=12, normal_type=af.SpectralFilterBankNormalType.AREA)
    q = o.cqt(x)
    assert np.isfinite(q).all()
    fn = o._lib.afx_cqt_pyramid_rings
    fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int]
    ring = np.zeros(cqt_rings.RING_FLOATS, np.float32)
    assert fn(o._obj, ring.ctypes.data_as(C.POINTER(C.c_float)), 1) == 1, "no rings: the ladder did not run"
    res, bars = cqt_rings.check_rings(ring, x, tiles), cqt_rings.bars(x, slack)
    for k, (err, lo, hi, nz) in res.items():
        print(f"level {k}: samples [{lo}, {hi}) ({nz} inside the signal): {err:.2e} of the peak (bar {bars[k]:.2e})")
        assert nz >= 256 and err <= bars[k], (k, err, bars[k])
    print("RINGS OK")


@pytest.mark.parametrize("n,seed,tiles,slack", [
    (40000, 77, 3, 1.0),        # the input of tests/emu/emulated_cqt_rings.py: 10 tiles in runs of 3
    (900000, 5, 40, 1.15),      # 220 tiles in runs of 40 (the planner alone would take 55): every ring deep inside the clip
])
def test_level_rings_against_the_float64_resampler(n, seed, tiles, slack):
    """1e-6 of the level's peak; where six float32-grade stages add up to more (levels 3-6), not farther from float64 than
    the reference's own float32 resampler chain on the same clip (x slack: 1.0 on the input the CPU emulation pins, 1.15
    on the long clip)"""
    env = dict(os.environ, AFX_CQT_PYR_TILES=str(tiles))
    env.pop("AFX_CQT_PYRAMID", None)
    r = subprocess.run([sys.executable, "-c", f"import sys; sys.path.insert(0, {ROOT!r}); from tests import test_cqt_pyramid as t; "
                        f"t._child({n}, {seed}, {tiles}, {slack})"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "RINGS OK" in r.stdout, (r.stdout + r.stderr)[-3000:]
    from tests.conftest import parity_log
    import re
    for m in re.finditer(r"level (\d): .*?: (\S+) of the peak \(bar (\S+)\)", r.stdout):
        parity_log(f"ladder level {m.group(1)} ring vs float64 resampler (n={n})", float(m.group(2)), float(m.group(3)),
                   "max(1e-6, the reference's float32 resampler chain vs float64)")
