"""CPU-only checks of the STFT object: (1) the numpy restatement of the padding modes and of the
inverse is pinned against the golden vectors (and the compiled reference when present);
(2) the library's HOST logic -- the padding index map the kernel uses and the streaming
(isContinue) framing state machine -- is driven through device-free test hooks and compared
with the reference's behaviour, call by call."""
import ctypes as C
import os

import numpy as np
import pytest

import audioflux_amd as af
from oracle import ref, restate
from tests import cases
from tests.conftest import assert_istft_parity, assert_parity

fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int)
POS = {v: k for k, v in cases.POS.items()}
PADMODE = {v: k for k, v in cases.PADMODE.items()}


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "stft.npz"))


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")
def test_compiled_reference_reproduces_golden(gold, tmp_path):
    from tests.golden import make_golden
    here = make_golden.HERE
    make_golden.HERE = str(tmp_path)
    try:
        make_golden.make_stft()
    finally:
        make_golden.HERE = here
    fresh = np.load(os.path.join(str(tmp_path), "stft.npz"))
    assert sorted(fresh.files) == sorted(gold.files)
    for k in gold.files:
        assert np.array_equal(fresh[k], gold[k]), k


@pytest.mark.parametrize("name", list(cases.STFT_CASES))
def test_restatement_matches_golden_stft(name, gold):
    c = cases.STFT_CASES[name]
    n = 1 << c["radix2_exp"]
    x = cases.make_input(c["x"], 16000)
    w = restate.fft_window(c["window_type"], n)
    if "pad" in c:
        p = c["pad"]
        S = restate.stft_padded(x, n, c["slide_length"], w, position=p[0], mode=p[1],
                                value1=p[2] if len(p) > 2 else 0.0, value2=p[3] if len(p) > 3 else 0.0)
    else:
        S = restate.stft_full(x, n, c["slide_length"], w)
    assert_parity(S, gold[f"{name}/re"] + 1j * gold[f"{name}/im"], 1e-5, name)


@pytest.mark.parametrize("name", list(cases.ISTFT_CASES))
def test_restatement_matches_golden_istft(name, gold):
    src, method, acc = cases.ISTFT_CASES[name]
    c = cases.STFT_CASES[src]
    n = 1 << c["radix2_exp"]
    S = gold[f"{src}/re"] + 1j * gold[f"{src}/im"]
    init = cases.noise(80, (S.shape[0] - 1) * c["slide_length"] + n) if acc else None
    w = restate.fft_window(c["window_type"], n)
    y = restate.istft(S, n, c["slide_length"], w, method, init)
    assert_istft_parity(y, gold[f"{name}/y"], restate.istft_norm(S.shape[0], n, c["slide_length"], w, method), name)


def test_wola_inverts_the_forward_transform(gold):
    """property: weighted overlap-add of an un-modified STFT returns the framed part of the input"""
    c = cases.STFT_CASES["plain_hann_1024"]
    x = cases.make_input(c["x"], 16000)
    y = gold["wola_hann_1024/y"]
    n = 1 << c["radix2_exp"]
    inner = slice(n, len(y) - n)  # edges see fewer than N/hop windows but are still normalised
    assert np.abs(y[inner] - x[: len(y)][inner]).max() < 1e-5


def test_pad_index_map_of_the_library():
    L = af.get_lib()
    L.afx_test_pad_index.restype = C.c_longlong
    L.afx_test_pad_index.argtypes = [C.c_longlong, C.c_int, C.c_int]
    for n in (1, 2, 3, 7, 100):
        q = np.arange(-3 * n - 5, 4 * n + 5)
        for mode, name in ((2, "reflect"), (3, "wrap")):
            got = np.array([L.afx_test_pad_index(int(v), n, mode) for v in q])
            assert np.array_equal(got, restate.pad_index(q, n, name)), (n, name)


def _stream(r, hop, lens, seed):
    L = af.get_lib()
    L.afx_test_stft_stream.restype = C.c_int
    L.afx_test_stft_stream.argtypes = [C.c_int, C.c_int, C.c_int, fp, ip, C.c_int, fp, ip, ip, ip]
    n_fft = 1 << r
    x = cases.noise(seed, int(sum(lens)))
    lens_a = np.array(lens, np.int32)
    cur = np.zeros(int(sum(lens)) + n_fft * len(lens) + 16, np.float32)
    cur_l, tl, tails = (np.zeros(len(lens), np.int32) for _ in range(3))
    st = L.afx_test_stft_stream(r, hop, 0, x.ctypes.data_as(fp), lens_a.ctypes.data_as(ip), len(lens),
                                cur.ctypes.data_as(fp), cur_l.ctypes.data_as(ip), tl.ctypes.data_as(ip),
                                tails.ctypes.data_as(ip))
    assert st == 0
    return x, cur, cur_l, tl, tails


@pytest.mark.parametrize("name", list(cases.STFT_STREAMS))
def test_streaming_state_machine_matches_golden(name, gold):
    """what stftObj_stft would upload call after call ([kept tail | chunk]), framed by the
    restatement, equals the reference's streaming output; frame counts equal call by call"""
    c = cases.STFT_STREAMS[name]
    n = 1 << c["radix2_exp"]
    x, cur, cur_l, tl, tails = _stream(c["radix2_exp"], c["slide_length"], c["chunks"], c["seed"])
    assert np.array_equal(tl, gold[f"{name}/tl"])
    w = restate.fft_window(c["window_type"], n)
    pos, rows = 0, []
    for k in range(len(tl)):
        if tl[k] > 0:
            rows.append(restate.stft_full(cur[pos:pos + cur_l[k]], n, c["slide_length"], w))
            assert rows[-1].shape[0] == tl[k]
            pos += cur_l[k]
    assert_parity(np.concatenate(rows), gold[f"{name}/re"] + 1j * gold[f"{name}/im"], 1e-5, name)


def test_streaming_equals_one_shot_when_hop_divides():
    """property: for hop <= fftLength the frames of a chunked stream are the frames of the whole signal"""
    lens = [100, 5, 3, 70, 64, 1, 200, 333]
    x, cur, cur_l, tl, tails = _stream(6, 16, lens, 5)
    whole = restate.frames_of(x, 64, 16)
    pos, rows = 0, []
    for k in range(len(tl)):
        if tl[k] > 0:
            rows.append(restate.frames_of(cur[pos:pos + cur_l[k]], 64, 16))
            pos += cur_l[k]
    got = np.concatenate(rows)
    assert got.shape == whole.shape and np.array_equal(got, whole)
    assert all(0 <= t < 64 for t in tails)
