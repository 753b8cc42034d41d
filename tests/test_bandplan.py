"""CPU-only: the per-lane banded view of a filter bank that the fused kernel consumes
(afx_bandplan.c) reproduces the dense bank exactly, covers every row once, and gives
each 32-lane half-wave 32 distinct LDS banks (conflict-free ds_read_b32 streams)."""
import ctypes as C

import numpy as np
import pytest

import audioflux_amd as af
from oracle import restate


class Band(C.Structure):
    _fields_ = [("num", C.c_int), ("tapsA", C.c_int), ("tapsB", C.c_int),
                ("startA", C.c_int * 64), ("startB", C.c_int * 64), ("rowA", C.c_int * 64),
                ("rowB", C.c_int * 64), ("wA", C.POINTER(C.c_float)), ("wB", C.POINTER(C.c_float))]


@pytest.mark.parametrize("num,n,sr,style", [(128, 2048, 16000, "slaney"), (128, 2048, 32000, "slaney"),
                                            (80, 2048, 16000, "etsi"), (64, 2048, 16000, "slaney"),
                                            (40, 2048, 16000, "slaney"), (128, 2048, 44100, "etsi"),
                                            (13, 2048, 8000, "slaney")])
def test_bandplan_is_exact_and_conflict_free(num, n, sr, style):
    lib = af.get_lib()
    bank, _, _ = restate.mel_bank(num, n, sr, 0, sr / 2, style)
    F = n // 2 + 1
    b = Band()
    assert lib.afx_bandplan_build(bank.ctypes.data_as(C.POINTER(C.c_float)), num, F, C.byref(b)) == 0
    sA, sB = np.array(b.startA), np.array(b.startB)
    rA, rB = np.array(b.rowA), np.array(b.rowB)
    for s in (sA, sB):
        assert (s >= 0).all() and (s % 2 == 0).all()  # ds_read_b64 streams: even starts,
        for h in range(2):                             # 32 distinct bank pairs per half-wave
            assert len(set((s[h * 32:(h + 1) * 32] // 2) % 32)) == 32
    lens = [(np.nonzero(r)[0][-1] - np.nonzero(r)[0][0] + 1) if r.any() else 0 for r in bank]
    assert b.tapsA <= max(lens) + 2  # padding rows to free bank pairs costs at most one pair
    wA = np.ctypeslib.as_array(b.wA, (b.tapsA, 64))
    wB = np.ctypeslib.as_array(b.wB, (b.tapsB, 64))
    rec = np.zeros_like(bank)
    for l in range(64):
        for rows, starts, w, taps in ((rA, sA, wA, b.tapsA), (rB, sB, wB, b.tapsB)):
            if rows[l] >= 0:
                for t in range(taps):
                    if w[t, l] != 0:
                        rec[rows[l], starts[l] + t] += w[t, l]
    assert np.array_equal(rec, bank)
    covered = sorted(list(rA[rA >= 0]) + list(rB[rB >= 0]))
    assert covered == list(range(num))
    lib.afx_bandplan_free(C.byref(b))


def test_bandplan_rejects_wide_banks():
    lib = af.get_lib()
    bank = np.ones((200, 1025), np.float32)
    b = Band()
    assert lib.afx_bandplan_build(bank.ctypes.data_as(C.POINTER(C.c_float)), 200, 1025, C.byref(b)) == 1
