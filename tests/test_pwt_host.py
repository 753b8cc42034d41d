"""CPU-only checks of the pseudo wavelet transform object: the "pseudo" (full-length) auditory bank
and band arrays bit for bit against the compiled reference, the fixtures reproduced by the
reference, the numpy restatement pinned against the fixtures, status codes without a device."""
import ctypes as C
import os

import numpy as np
import pytest

import audioflux_amd as af
from oracle import ref, restate
from tests import cases
from tests.conftest import assert_parity

fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int)


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "pwt.npz"))


def lib_bank(num, n, sr, scale, style, normal, low, high, bpo):
    L = af.get_lib()
    L.afx_pwt_bank_host.restype = C.c_int
    L.afx_pwt_bank_host.argtypes = [C.c_int, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                    C.c_int, fp, fp, ip]
    bank, fre, bins = np.zeros((num, n), np.float32), np.zeros(num + 2, np.float32), np.zeros(num + 2, np.int32)
    st = L.afx_pwt_bank_host(num, n, sr, scale, style, normal, low, high, bpo, bank.ctypes.data_as(fp),
                             fre.ctypes.data_as(fp), bins.ctypes.data_as(ip))
    assert st == 0
    return bank, fre[:num], bins[:num]


BANKS = [(84, 8192, 32000, 5, 0, 0, 32.703, 16000.0, 12), (40, 2048, 16000, 2, 0, 1, 0.0, 8000.0, 12),
         (32, 1024, 16000, 3, 1, 2, 50.0, 7000.0, 12), (20, 4096, 16000, 4, 5, 0, 100.0, 6000.0, 12),
         (16, 512, 8000, 1, 4, 0, 200.0, 3000.0, 12), (24, 2048, 44100, 6, 10, 1, 800.0, 15000.0, 12),
         (50, 1024, 16000, 0, 0, 0, 500.0, 8000.0, 12), (30, 2048, 16000, 0, 1, 0, 1000.0, 8000.0, 12)]


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("cfg", BANKS)
def test_pseudo_bank_bit_identical(cfg):
    num, n, sr, scale, style, normal, low, high, bpo = cfg
    R = ref.lib()
    R.auditory_filterBank.restype = None
    R.auditory_filterBank.argtypes = [C.c_int] * 7 + [C.c_float, C.c_float, C.c_int, fp, fp, ip]
    want, wf, wb = np.zeros((num, n), np.float32), np.zeros(num + 2, np.float32), np.zeros(num + 2, np.int32)
    R.auditory_filterBank(num, n, sr, 1, scale, style, normal, low, high, bpo, want.ctypes.data_as(fp),
                          wf.ctypes.data_as(fp), wb.ctypes.data_as(ip))
    bank, fre, bins = lib_bank(*cfg)
    assert np.array_equal(bank, want) and np.array_equal(fre, wf[:num]) and np.array_equal(bins, wb[:num])


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")
def test_compiled_reference_reproduces_golden(gold, tmp_path):
    from tests.golden import make_golden
    here = make_golden.HERE
    make_golden.HERE = str(tmp_path)
    try:
        make_golden.make_pwt()
    finally:
        make_golden.HERE = here
    fresh = np.load(os.path.join(str(tmp_path), "pwt.npz"))
    assert sorted(fresh.files) == sorted(gold.files)
    for k in gold.files:
        assert np.array_equal(fresh[k], gold[k]), k


def resolved_range(c):
    """lowFre / highFre after pwtObj_new's own revision (pwt_algorithm.c:171-190), float32;
    only the linear scale of the restated cases is revised there"""
    from numpy import float32 as f
    sr, d = c["samplate"], 1 << c["radix2_exp"]
    low, high = f(c.get("low_fre", 0.0)), f(c.get("high_fre", sr / 2))
    if c["scale_type"] == cases.SCALE["linear"]:
        det = f(sr) / f(d)
        lo = np.round(low / det)
        return f(lo * det), f((lo + c["num"] - 1) * det)
    return low, high


@pytest.mark.parametrize("name", ["mel40_area_nopad", "bark32_etsi_bw", "erb20_hann_style", "log24_gauss_big",
                                  "linear50_points", "linspace_rect_tiny"])
def test_restatement_matches_golden(name, gold):
    c = cases.PWT_CASES[name]
    d = 1 << c["radix2_exp"]
    pad = d // 2 if c["is_padding"] else 0
    low, high = resolved_range(c)
    bank, fre, bins = lib_bank(c["num"], d + 2 * pad, c["samplate"], c["scale_type"], c["style_type"],
                               c["normal_type"], float(low), float(high), c.get("bin_per_octave", 12))
    assert np.array_equal(fre, gold[f"{name}/fre"]) and np.array_equal(bins, gold[f"{name}/bin"])
    x = cases.make_input((c["x"][0], c["x"][1], d), c["samplate"])
    st = cases.cwt_stride(c)
    got = restate.pwt(x, bank, pad)[:, ::st]
    assert_parity(got, gold[f"{name}/re"] + 1j * gold[f"{name}/im"], 1e-5, name)
    if f"{name}/det_re" in gold.files:
        got = restate.pwt(x, bank, pad, det=True)[:, ::st]
        assert_parity(got, gold[f"{name}/det_re"] + 1j * gold[f"{name}/det_im"], 1e-5, name + " det")


def test_status_codes_without_device():
    lib = af.get_lib()
    f = lib.pwtObj_new
    f.restype = C.c_int
    f.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int] + [C.c_void_p] * 8
    obj = C.c_void_p(None)
    assert f(C.byref(obj), 84, 31, *([None] * 8)) == -100 and not obj
    assert f(C.byref(obj), 1, 10, *([None] * 8)) == -1 and not obj
    assert f(C.byref(obj), 5000, 10, *([None] * 8)) == -1 and not obj
    scale = C.c_int(7)
    args = [None] * 8
    args[4] = C.cast(C.pointer(scale), C.c_void_p)
    assert f(C.byref(obj), 84, 12, *args) == 1 and not obj
    style = C.c_int(2)  # gammatone: a valid plan since round 3 -- the constructor gets as far as the device
    mel = C.c_int(2)
    args = [None] * 8
    args[4], args[5] = C.cast(C.pointer(mel), C.c_void_p), C.cast(C.pointer(style), C.c_void_p)
    lib.pwtObj_free.argtypes = [C.c_void_p]
    st = f(C.byref(obj), 40, 12, *args)
    assert st in (0, -2), st  # -2: no gfx950 device on this machine
    if st == 0:
        lib.pwtObj_free(obj)
    lib.pwtObj_free(None)
