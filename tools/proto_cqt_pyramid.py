#!/usr/bin/env python3
"""The schedule of k_cqt_pyramid (audioflux_amd/csrc/hip/afx_cqt_f16.hip) as a model: who writes which block of which
level ring in which step, who reads it when -- every read must find a block that was written in an EARLIER step (one
s_barrier per step is the only synchronisation) and that no later write has taken the ring slot of.

Levels k = 0 ... 6, hop H = 128 >> k, window S = 31 H + 512 samples from 32 t H - 256, block b of level k = samples
[32 b H, 32 (b+1) H).  Wave of level k works on tile s - lag(k); the resampler of a tile writes block t of level k+1
(level 0: tile s - 1, by the multiplying wave); a window is requested in the step BEFORE its tile is worked on
(multiply-first waves: at the start of that step -- the same blocks).  check(t0, t1) walks a run."""
import sys

LEVELS, LEAD, DRAIN = 7, 9, 22
LAG = [1, 4, 7, 10, 13, 17, 23]
RING = [0, 8192, 4096, 2048, 1024, 1024, 1024]
ROWS_LATE = [0, 1, 1, 1, 0, 0, 0]  # levels 1-3 keep a tile's accumulators over the barrier: rows (and chroma sums) one step later


def hop(k):
    return 128 >> k


def need_back(k):
    return 9 - k


def need_ahead(k):
    return 8 - k


def blocks_of(k, lo, hi):
    """blocks of level k that hold samples lo ... hi - 1"""
    return range(lo // (32 * hop(k)), (hi - 1) // (32 * hop(k)) + 1)


def check(t0, t1, length_tiles=None):
    written = {}  # (level, slot) -> (block, step)
    reads = 0
    for s in range(t0 - LEAD, t1 + DRAIN + 1):
        pending = []
        for k in range(LEVELS):
            H = hop(k)
            t = s - LAG[k]
            oct_ = t0 <= t < t1
            dec = k < 6 and t0 - need_back(k) <= t <= t1 + need_ahead(k)
            # the window of tile t was requested in step s - 1 (levels >= 1 read their ring)
            if k >= 1 and (oct_ or dec):
                p0 = 32 * t * H - 256
                lo, hi = (p0, p0 + 31 * H + 512) if oct_ else (p0 + 224, p0 + 32 * H + 288)
                for b in blocks_of(k, lo, hi):
                    slot = b % (RING[k] // (32 * H))
                    got = written.get((k, slot))
                    assert got is not None and got[0] == b, f"step {s}: level {k} tile {t} reads block {b}, ring slot holds {got}"
                    assert got[1] <= s - 2, f"step {s}: level {k} tile {t} reads block {b} written in step {got[1]} (requested in {s - 1})"
                    reads += 1
            if dec:
                pending.append((k + 1, t))
        for k, b in pending:  # the step's writes (visible from the next step on)
            slot = b % (RING[k] // (32 * hop(k)))
            written[(k, slot)] = (b, s)
        # no write of this step may take a slot that a request of this step (for tiles worked on in s + 1) still needs
        for k in range(1, LEVELS):
            H = hop(k)
            t = s + 1 - LAG[k]
            if t0 <= t < t1 or (k < 6 and t0 - need_back(k) <= t <= t1 + need_ahead(k)):
                p0 = 32 * t * H - 256
                oct_ = t0 <= t < t1
                lo, hi = (p0, p0 + 31 * H + 512) if oct_ else (p0 + 224, p0 + 32 * H + 288)
                for b in blocks_of(k, lo, hi):
                    slot = b % (RING[k] // (32 * H))
                    got = written.get((k, slot))
                    assert got is not None and got[0] == b and got[1] <= s - 1, f"step {s}: request of level {k} tile {t}: block {b} vs slot {got}"
    # every octave tile of the run was worked on inside the step range
    for k in range(LEVELS):
        assert t0 + LAG[k] >= t0 - LEAD and t1 - 1 + LAG[k] <= t1 + DRAIN
    # chroma partial sums travel through the output rows: level k REQUESTS tile t's partials in step t + LAG[k] (before its K
    # loop) and ADDS its own in the step its rows leave -- the same step, or one later for the convert-first waves of levels
    # 1-3 (ROWS_LATE: they store a tile's rows at the start of the next step).  A request must find the add of the level
    # before it drained by a vmcnt(0) + barrier: issued at least two steps earlier; a deferred add must still fall inside the run's steps.
    for t in range(t0, t1):
        for k in range(1, LEVELS):
            add_before = t + LAG[k - 1] + ROWS_LATE[k - 1]
            assert add_before <= t + LAG[k] - 2, f"tile {t}: level {k} requests the partials in step {t + LAG[k]}, level {k - 1} adds in {add_before}"
        for k in range(LEVELS):
            assert t + LAG[k] + ROWS_LATE[k] <= t1 + DRAIN, f"tile {t}: the rows of level {k} leave after the run's last step"
    return reads


# ---- the resampler as a matrix-core product: index algebra of pyr_dec_loop / pyr_dec_store and of the tap table ----
def tap_table(taps):
    """afx_cqt_dec_table: [copy a][x] = T[x - 160 - 2a], T[d] = h[|d|] for |d| <= 31 (here in float64, unscaled)"""
    import numpy as np
    tab = np.zeros((4, 352))
    for a in range(4):
        for x in range(352):
            d = abs(x - 160 - 2 * a)
            if d <= 31:
                tab[a, x] = taps[d]
    return tab


def dec_ks(H, ct):
    cols = min(H // 2, 32)
    return (225 + 64 * ct) // 16, (2 * (cols - 1 + 32 * ct) + 287) // 16


def model_resampler():
    import numpy as np
    sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
    from oracle import restate
    taps = restate.halfband_taps()
    tab = tap_table(taps)
    rng = np.random.default_rng(1)
    for H in (128, 64, 32, 16, 8, 4):
        x = rng.standard_normal(40 * H * 32 // 8)
        want = restate.decimate2(x) * np.sqrt(0.5)  # (the kernel applies 1 / sqrt(ratio) in its multiplier)
        W = H // 2
        for t in (0, 1, 2):
            p0 = 32 * t * H - 256
            for ct in range(2 if W > 32 else 1):
                ks0, ks1 = dec_ks(H, ct)
                D = np.zeros((32, 32))  # [output row c' - 32 ct][frame]
                for lane in range(64):
                    i, g = lane & 31, lane >> 5
                    cp = 32 * ct + i  # the table operand's row: output c'
                    for ks in range(ks0, ks1 + 1):
                        x0 = 16 * ks + 8 * g - 96 - 8 * (cp >> 2)
                        assert 0 <= x0 and x0 + 7 < 352 and x0 % 8 == 0, (H, ct, lane, ks, x0)
                        tfrag = tab[cp & 3, x0:x0 + 8]
                        for frame in range(32):  # the signal operand's column: lane (frame, g) holds these eight samples
                            pos = p0 + frame * H + 16 * ks + 8 * g + np.arange(8)
                            sig = np.where((pos >= 0) & (pos < len(x)), x[np.clip(pos, 0, len(x) - 1)], 0.0)
                            D[i, frame] += tfrag @ sig
                for frame in range(32):
                    for c in range(min(W - 32 * ct, 32)):
                        idx = (32 * t + frame) * W + 32 * ct + c
                        if 0 <= idx < len(want):
                            assert abs(D[c, frame] - want[idx]) <= 1e-12 * max(1.0, abs(want[idx])), (H, t, ct, frame, c, D[c, frame], want[idx])
        # the 16 lanes of a ds_read_b128 group read 16 distinct bank quads of the table (copies 704 bytes apart)
        for grp in ([0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]):
            quads = {(((l & 3) * 704 + 2 * (-8 * (l >> 2) - 96)) // 16) % 16 for l in grp}
            assert len(quads) == 16, quads
    return True


if __name__ == "__main__":
    model_resampler()
    n = 0
    for t0, t1 in ((0, 1), (0, 3), (5, 9), (0, 162), (161, 323), (40, 41)):
        n += check(t0, t1)
    print(f"OK: resampler index algebra and table banks; schedule: {n} block reads checked")
