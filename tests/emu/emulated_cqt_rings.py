#!/usr/bin/env python3
"""k_cqt_pyramid emulated on the CPU (tests/emu): the level rings of the first workgroup against the 2:1 resampler in
float64 (tests/cqt_rings.py).  AFX_LIB = the emulated library, AFX_CQT_PYR_TILES=3 (set by the caller) cuts the clip into
runs of three tiles, so that run 0 = tiles [0, 3) ends mid-clip.  Prints one line per level and OK."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import cqt_rings  # noqa: E402

lib = C.CDLL(os.environ["AFX_LIB"])
vp, fp = C.c_void_p, C.POINTER(C.c_float)
lib.cqtObj_calTimeLength.restype = C.c_int
lib.afx_cqt_pyramid_rings.restype = C.c_int
lib.afx_cqt_pyramid_rings.argtypes = [vp, fp, C.c_int]
lib.afx_emulated_launches.restype = C.c_int
lib.afx_emulated_launches.argtypes = [C.c_char_p]


def main():
    tiles = int(os.environ["AFX_CQT_PYR_TILES"])
    x = cqt_rings.ladder_input()
    n = len(x)
    h = vp()
    st = lib.cqtObj_newWith(C.byref(h), 84, C.byref(C.c_int(44100)), C.byref(C.c_float(32.703)), None, None, None, None, None, None, None,
                            C.byref(C.c_int(1)), None)
    assert st == 0, st
    T = lib.cqtObj_calTimeLength(h, n)
    re, im = np.zeros((T, 84), np.float32), np.zeros((T, 84), np.float32)
    lib.cqtObj_cqt(h, x.ctypes.data_as(fp), n, re.ctypes.data_as(fp), im.ctypes.data_as(fp))
    assert lib.afx_emulated_launches(b"k_cqt_pyramid") == 1
    ring = np.zeros(cqt_rings.RING_FLOATS, np.float32)
    assert lib.afx_cqt_pyramid_rings(h, ring.ctypes.data_as(fp), 1) == 1
    res, bars = cqt_rings.check_rings(ring, x, tiles), cqt_rings.bars(x)
    for k, (err, lo, hi, nz) in res.items():
        print(f"level {k}: ring = samples [{lo}, {hi}) ({nz} of them inside the signal), {err:.2e} of the level's peak from the float64 "
              f"resampler (bar {bars[k]:.2e})")
        assert nz >= 256 and err <= bars[k], (k, err, bars[k])
    lib.cqtObj_free(h)
    print("OK")


if __name__ == "__main__":
    main()
