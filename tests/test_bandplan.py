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
                ("rowB", C.c_int * 64), ("wA", C.POINTER(C.c_float)), ("wB", C.POINTER(C.c_float)),
                ("split", C.c_int), ("segIdx", C.c_uint * 128)]


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


SCALE = {"mel": 2, "bark": 3, "erb": 4}


def _bank(num, n, sr, scale):
    lib = af.get_lib()
    fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int)
    lib.afx_auditory_bank.restype = None
    lib.afx_auditory_bank.argtypes = [C.c_int] * 6 + [C.c_float, C.c_float, C.c_int, fp, fp, ip]
    F = n // 2 + 1
    bank = np.zeros((num, F), np.float32)
    fre, bins = np.zeros(num + 2, np.float32), np.zeros(num + 2, np.int32)
    lib.afx_auditory_bank(num, n, sr, SCALE[scale], 0, 0, 0.0, sr / 2, 12, bank.ctypes.data_as(fp),
                          fre.ctypes.data_as(fp), bins.ctypes.data_as(ip))
    return bank


VARIANTS = {2048: ((48, 16), (72, 32)), 4096: ((96, 32), (128, 64), (176, 8))}  # compiled tap variants
ROW_CAP = {2048: 1104, 4096: 2176}  # floats of the kernels' zero-padded power rows (PROW_F)


@pytest.mark.parametrize("n,scale,num,sr,variant", [
    (2048, "mel", 40, 16000, (48, 16)), (2048, "mel", 64, 16000, (48, 16)), (2048, "mel", 64, 44100, (48, 16)),
    (2048, "mel", 20, 22050, (72, 32)), (2048, "bark", 64, 16000, (48, 16)), (2048, "bark", 64, 44100, (72, 32)),
    (2048, "erb", 64, 16000, (48, 16)), (2048, "erb", 40, 44100, (72, 32)), (2048, "bark", 96, 48000, (72, 32)),
    (4096, "mel", 80, 32000, (96, 32)), (4096, "mel", 40, 16000, (96, 32)), (4096, "bark", 40, 32000, (176, 8)),
    (4096, "bark", 64, 44100, (128, 64)), (4096, "erb", 64, 22050, (96, 32)),
])
def test_split_plan_is_exact_and_conflict_free(n, scale, num, sr, variant):
    """rows longer than the compiled tap variants are cut into segments (afx_bandplan_build_split):
    emulating the kernel's stage 4 (slot dot products over the zero-padded power row) + stage 5
    (a row = the sum of its <= 4 slots) reproduces bank . power, starts are even and give every
    32-lane half-wave 32 distinct LDS bank pairs, and the smallest variant that fits is chosen"""
    lib = af.get_lib()
    F, cap = n // 2 + 1, ROW_CAP[n]
    bank = _bank(num, n, sr, scale)
    fp = C.POINTER(C.c_float)
    b = Band()
    rc = lib.afx_bandplan_build(bank.ctypes.data_as(fp), num, F, C.byref(b))
    assert rc == 0 and not any(b.tapsA <= ta and b.tapsB <= tb for ta, tb in VARIANTS[n])
    lib.afx_bandplan_free(C.byref(b))
    chosen = None
    for ta, tb in VARIANTS[n]:
        b = Band()
        if lib.afx_bandplan_build_split(bank.ctypes.data_as(fp), num, F, ta, tb, cap, C.byref(b)) == 0:
            chosen = (ta, tb)
            break
    assert chosen == variant
    ta, tb = chosen
    assert b.split == 1 and b.num == num and (b.tapsA, b.tapsB) == chosen
    assert all(r == -1 for r in b.rowA) and all(r == -1 for r in b.rowB)
    sA, sB = np.array(b.startA), np.array(b.startB)
    for s, taps in ((sA, ta), (sB, tb)):
        assert (s >= 0).all() and (s % 2 == 0).all() and (s + taps <= cap).all()  # no read past the power row
        for h in range(2):
            assert len(set((s[h * 32:(h + 1) * 32] // 2) % 32)) == 32
    wA = np.ctypeslib.as_array(b.wA, (ta, 64)).astype(np.float64)
    wB = np.ctypeslib.as_array(b.wB, (tb, 64)).astype(np.float64)
    power = np.zeros(cap)
    power[:F] = np.random.default_rng(num + sr).random(F)
    part = np.zeros(129)
    for l in range(64):
        part[l] = wA[:, l] @ power[sA[l]:sA[l] + ta]
        part[64 + l] = wB[:, l] @ power[sB[l]:sB[l] + tb]
    seg = np.array(b.segIdx, dtype=np.uint32)
    got = np.array([sum(part[(seg[r] >> (8 * c)) & 255] for c in range(4)) for r in range(num)])
    assert np.allclose(got, bank.astype(np.float64) @ power[:F], rtol=1e-12, atol=1e-12)
    assert all(seg[r] == 0x80808080 for r in range(num, 128))  # rows beyond num: no slots
    # a row's slots are listed in ascending bin order and every used slot belongs to one row
    used = [int((seg[r] >> (8 * c)) & 255) for r in range(num) for c in range(4) if (seg[r] >> (8 * c)) & 255 != 128]
    assert len(used) == len(set(used))
    for r in range(num):
        starts = [(sA[u] if u < 64 else sB[u - 64]) for u in [int((seg[r] >> (8 * c)) & 255) for c in range(4)] if u != 128]
        assert starts == sorted(starts)
    lib.afx_bandplan_free(C.byref(b))


def test_split_plan_refuses_what_does_not_fit():
    lib = af.get_lib()
    fp = C.POINTER(C.c_float)
    for scale, num, sr in (("mel", 13, 16000), ("bark", 128, 44100)):
        bank = _bank(num, 2048, sr, scale)
        for ta, tb in VARIANTS[2048]:
            b = Band()
            assert lib.afx_bandplan_build_split(bank.ctypes.data_as(fp), num, 1025, ta, tb, 1104, C.byref(b)) == 1
            assert b.wA is None or not b.wA  # nothing left allocated
