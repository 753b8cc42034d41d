"""The level signals of the one-launch CQT ladder (k_cqt_pyramid) against the 2:1 resampler in float64.

The ladder never writes its level signals anywhere a caller sees: level k + 1 is the matrix-core product of level k's
planes with a 63-tap table (what replaces /root/reference/src/dsp/resample_algorithm.c:430-521), kept in per-workgroup
rings of global memory.  `afx_cqt_pyramid_rings` (csrc/host/afx_cqt.c) copies the rings out as the last launch left them;
a ring of level k holds the LAST ring_size(k) positions its workgroup's last run wrote (sample p at p mod the size): the
blocks up to tile t1 + need_ahead(k - 1) of level k - 1, where [t0, t1) is the run (afx_cqt_f16.hip, namespace pyr).
With runs cut short (AFX_CQT_PYR_TILES, read when the object is created) the first workgroup's run ends mid-clip and its
rings hold signal, not the zeros behind a clip's end.

Used by tests/test_cqt_pyramid.py (device) and tests/emu/emulated_cqt_rings.py (the same kernel emulated on the CPU).
"""
import numpy as np

RING_FLOATS = 17408
RING_OFF = {1: 0, 2: 8192, 3: 12288, 4: 14336, 5: 15360, 6: 16384}
RING_SIZE = {1: 8192, 2: 4096, 3: 2048, 4: 1024, 5: 1024, 6: 1024}


def level_signals(x):
    """float64 level signals 0 ... 6 of one clip: level k + 1 = decimate2(level k) (oracle/restate.py, the closed form of
    the reference's 'Fast' resampler that tests/test_oracle.py pins against the compiled reference)"""
    from oracle import restate
    lv = [np.asarray(x, np.float64)]
    for _ in range(6):
        lv.append(restate.decimate2(lv[-1]))
    return lv


def check_rings(ring, x, t1, what=""):
    """ring: the RING_FLOATS floats of the workgroup whose last run was tiles [t0, t1) of clip x (t1 mid-clip).
    Returns {level: (error relative to the level's peak, window start, window end)}; the window's end is SEARCHED among
    the block boundaries near the schedule's (t1 + 9 - k blocks of 4096 >> k samples) -- agreement to 1e-6 on thousands
    of noise samples at a wrong alignment is impossible, so the search cannot make a wrong ring pass."""
    lv = level_signals(x)
    out = {}
    for k in range(1, 7):
        size, blk = RING_SIZE[k], 4096 >> k
        r = np.asarray(ring[RING_OFF[k]:RING_OFF[k] + size], np.float64)
        want_full = lv[k]
        peak = np.abs(want_full).max()
        best = None
        for hi_blk in range(max(t1, 1), t1 + 14):
            hi = hi_blk * blk
            lo = hi - size
            p = np.arange(lo, hi)
            want = np.where((p >= 0) & (p < len(want_full)), want_full[np.clip(p, 0, len(want_full) - 1)], 0.0)
            got = r[p % size]
            err = np.abs(got - want).max() / peak
            if best is None or err < best[0]:
                best = (err, lo, hi, int(np.count_nonzero(want)))
        out[k] = best
    return out


def reference_chain_errors(x):
    """the compiled reference's own float32 resampler (resampleObj, quality Fast, isScale: what cqtObj creates at
    cqt_algorithm.c:1268-1274) applied six times, each level's distance from the float64 chain relative to its peak --
    the yardstick for the deeper levels, where six float32 stages add up; None without oracle/_ref"""
    import ctypes as C
    from oracle import ref
    if not ref.available():
        return None
    lib = ref.lib()
    fp = C.POINTER(C.c_float)
    h = C.c_void_p()
    assert lib.resampleObj_new(C.byref(h), C.byref(C.c_int(2)), C.byref(C.c_int(1)), None) == 0
    lib.resampleObj_setSamplate(h, 2, 1)
    lv = level_signals(x)
    cur = np.ascontiguousarray(x, np.float32)
    out = {}
    for k in range(1, 7):
        nxt = np.zeros(len(cur) // 2 + 8, np.float32)
        lib.resampleObj_resample(h, cur.ctypes.data_as(fp), len(cur), nxt.ctypes.data_as(fp))
        cur = nxt[: len(cur) // 2].copy()
        out[k] = float(np.abs(cur - lv[k]).max() / np.abs(lv[k]).max())
    lib.resampleObj_free(h)
    return out


def bars(x, slack=1.0):
    """1e-6 of the level's peak (one application of the resampler is within it); where six stages add up to more, not
    farther from float64 than the reference's own float32 chain at that level"""
    r = reference_chain_errors(x)
    return {k: max(1e-6, slack * r[k]) if r else 1e-6 * max(1.0, k / 2.0) for k in range(1, 7)}


def ladder_input(n=40000, seed=77):
    x = (0.1 * np.random.default_rng(seed).standard_normal(n)).astype(np.float32)
    return x + (0.3 * np.sin(2 * np.pi * 110.0 / 44100 * np.arange(n))).astype(np.float32)  # energy in the low octaves too
