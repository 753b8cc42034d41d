"""CPU-only: the host-side plan construction of the product library (windows,
filter banks -- float32, setup-time) is BIT-IDENTICAL to the reference's, because
band edges are decided by float32 comparisons (SURVEY.md section 7 hard part 2)."""
import ctypes as C

import numpy as np
import pytest

import audioflux_amd as af
from oracle import ref

fp = C.POINTER(C.c_float)
ip = C.POINTER(C.c_int)
pytestmark = pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")


@pytest.mark.parametrize("wt", range(14))
def test_fft_windows_bit_exact(wt):
    R, N = ref.lib(), af.get_lib()
    R.window_calFFTWindow.restype = fp
    R.window_calFFTWindow.argtypes = [C.c_int, C.c_int]
    N.afx_window_fft.restype = fp
    N.afx_window_fft.argtypes = [C.c_int, C.c_int]
    for n in (2, 3, 4, 5, 8, 17, 64, 512, 2048, 4096):
        a = np.ctypeslib.as_array(R.window_calFFTWindow(wt, n), (n,)).copy()
        b = np.ctypeslib.as_array(N.afx_window_fft(wt, n), (n,)).copy()
        assert np.array_equal(a, b), (wt, n, np.abs(a - b).max())


SHAPES = [(128, 2048, 16000, 0.0, 8000.0), (40, 1024, 32000, 27.5, 16000.0),
          (64, 4096, 44100, 50.0, 20000.0), (13, 512, 8000, 100.0, 3500.0)]


@pytest.mark.parametrize("scale", [1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("style", range(11))
def test_filter_banks_bit_exact(scale, style):
    R, N = ref.lib(), af.get_lib()
    R.auditory_filterBank.restype = None
    R.auditory_filterBank.argtypes = [C.c_int] * 7 + [C.c_float, C.c_float, C.c_int, fp, fp, ip]
    N.afx_auditory_bank.restype = None
    N.afx_auditory_bank.argtypes = [C.c_int] * 6 + [C.c_float, C.c_float, C.c_int, fp, fp, ip]
    for norm in range(3):
        for num, n, sr, lo, hi in SHAPES:
            if scale in (5, 6) and lo == 0:
                lo = 27.5
            if scale == 1:  # keep the widened linspace edges inside [0, sr/2]: outside is UB in the reference
                lo, hi = 1000.0, hi - 1200.0
            if scale == 5 and num == 128:  # 128 semitones above 27.5 Hz overflow Nyquist (reference UB; bftObj_new rejects it)
                continue
            F = n // 2 + 1
            a = np.zeros(num * F + 8 * n, np.float32)
            b = np.zeros((num, F), np.float32)
            fa, fb = np.zeros(num + 8, np.float32), np.zeros(num + 8, np.float32)
            ba, bb = np.zeros(num + 8, np.int32), np.zeros(num + 8, np.int32)
            R.auditory_filterBank(num, n, sr, 0, scale, style, norm, lo, hi, 12,
                                  a.ctypes.data_as(fp), fa.ctypes.data_as(fp), ba.ctypes.data_as(ip))
            N.afx_auditory_bank(num, n, sr, scale, style, norm, lo, hi, 12,
                                b.ctypes.data_as(fp), fb.ctypes.data_as(fp), bb.ctypes.data_as(ip))
            a = a[: num * F].reshape(num, F)
            tag = (scale, style, norm, num, n, sr)
            assert np.array_equal(fa, fb) and np.array_equal(ba, bb), tag
            assert np.array_equal(a, b, equal_nan=True), (tag, np.nanmax(np.abs(a - b)))


def test_dct_and_twiddle_tables():
    N = af.get_lib()
    N.afx_dct2_matrix.restype = fp
    N.afx_dct2_matrix.argtypes = [C.c_int, C.c_int]
    for num in (13, 80, 128):
        d = np.ctypeslib.as_array(N.afx_dct2_matrix(num, num), (num, num)).astype(np.float64)
        assert np.abs(d @ d.T - np.eye(num)).max() < 1e-6  # orthonormal
    N.afx_twiddle_table.restype = fp
    N.afx_twiddle_table.argtypes = [C.c_int]
    t = np.ctypeslib.as_array(N.afx_twiddle_table(2048), (1024, 2)).astype(np.float64)
    w = np.exp(-2j * np.pi * np.arange(1024) / 2048)
    assert np.abs(t[:, 0] + 1j * t[:, 1] - w).max() < 1e-7


def test_cqt_f16_fragment_words_match_numpy():
    """afx_cqt_time_kernel_f16 (host side of afx_cqt_f16.hip): round-to-nearest-even binary16 (hi, lo) words of the
    power-of-two scaled image columns, in MFMA B-fragment order, bit for bit against numpy.float16"""
    import ctypes

    from audioflux_amd import _lib
    lib = _lib.get_lib()
    fn = lib.afx_cqt_time_kernel_f16
    fn.restype = None
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    rng = np.random.default_rng(11)
    N = 512
    G = (rng.standard_normal((N, 32)) * 10.0 ** rng.uniform(-9, -1, (1, 32))).astype(np.float32)
    G[:, 24:] = 0.0                                   # padding columns
    G[5, 3] = 0.0
    G[7, 2] = np.float32(1e-30)                       # far below the column peak: subnormal / zero words
    G[9:40, 1] *= np.float32(2.0 ** -13)              # values whose hi word is an f16 subnormal
    out = np.zeros((2, N // 16, 64, 8), np.uint16)
    cm = np.zeros(32, np.float32)
    fn(G.ctypes.data, N, out.ctypes.data, cm.ctypes.data)
    for j in range(32):
        pk = np.abs(G[:, j]).max()
        s = 14 - int(np.frexp(pk)[1]) if pk > 0 else 0
        assert cm[j] == np.ldexp(np.float32(1), -s)
        v = np.ldexp(G[:, j], s).astype(np.float32)
        hi = v.astype(np.float16)
        lo = (v - hi.astype(np.float32)).astype(np.float16)
        k = np.arange(N)
        got_hi = out[0, k // 16, 32 * ((k % 16) // 8) + j, k % 8]
        got_lo = out[1, k // 16, 32 * ((k % 16) // 8) + j, k % 8]
        # +0 / -0 words compare equal as values; everything else bit for bit
        assert np.array_equal(got_hi.view(np.float16), hi) and np.array_equal(got_lo.view(np.float16), lo), j
        nz = hi != 0
        assert np.array_equal(got_hi[nz], hi.view(np.uint16)[nz])
        # hi + lo carries the float32 value to 2^-21 of itself (2^-25 absolute in the subnormal range)
        err = np.abs(hi.astype(np.float64) + lo.astype(np.float64) - v)
        assert np.all(err <= np.maximum(np.abs(v) * 2.0 ** -21, 2.0 ** -25))


def test_cqt_pass_sizes(monkeypatch):
    """afx_cqt_pass_clips: the fewest equal passes of <= 448 MB of output; AFX_CQT_CHUNK overrides"""
    import ctypes

    from audioflux_amd import _lib
    fn = _lib.get_lib().afx_cqt_pass_clips
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_longlong, ctypes.c_int]
    monkeypatch.delenv("AFX_CQT_CHUNK", raising=False)
    row = 10336 * 84                       # BASELINE cfg 5: 30 s @ 44.1 kHz, 84 bins
    cap = int(448 * 1024 * 1024 / (8.0 * row))
    assert cap == 67
    assert fn(row, 125) == 63              # two passes, 63 + 62
    assert fn(row, 67) == 67 and fn(row, 68) == 34 and fn(row, 1) == 67
    assert fn(row, 1000) == 67             # 15 passes of <= 67
    for batch in (1, 7, 67, 68, 125, 134, 135, 1000, 32768):
        c = fn(row, batch)
        passes = -(-batch // c)
        assert c <= cap and passes == -(-batch // cap)   # as few passes as the cap allows ...
        if batch > cap:
            assert passes * c - batch < passes            # ... of equal size: short by less than one clip per pass
    assert fn(10, 5) >= 5                                # tiny rows: one pass
    monkeypatch.setenv("AFX_CQT_CHUNK", "3")
    assert fn(row, 125) == 3


def test_chroma_lists_equal_the_folding_matrix():
    """afx_chroma_lists (argument of the switched-off k_cqt_chroma_v2): per-class ascending bin lists of the 0/1 matrix
    of afx_chroma_fold -- summing a row's bins in list order is the matrix product"""
    import ctypes

    from audioflux_amd import _lib
    lib = _lib.get_lib()

    class Lists(ctypes.Structure):
        _fields_ = [("start", ctypes.c_ushort * 65), ("bins", ctypes.c_ubyte * 256)]
    fold_fn = lib.afx_chroma_fold
    fold_fn.restype = ctypes.POINTER(ctypes.c_ubyte)
    fold_fn.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float]
    lists_fn = lib.afx_chroma_lists
    lists_fn.restype = ctypes.c_int
    lists_fn.argtypes = [ctypes.POINTER(ctypes.c_ubyte), ctypes.c_int, ctypes.c_int, ctypes.POINTER(Lists)]
    rng = np.random.default_rng(3)
    for cn, num, bpo, fmin in ((12, 84, 12, 32.703), (6, 84, 12, 32.703), (12, 48, 12, 65.4), (24, 168, 24, 27.5), (12, 84, 12, 55.0)):
        fp = fold_fn(cn, num, bpo, fmin)
        fold = np.ctypeslib.as_array(fp, shape=(cn * num,)).reshape(cn, num).copy()
        L = Lists()
        assert lists_fn(fp, cn, num, ctypes.byref(L)) == 0
        start, bins = np.array(L.start[:]), np.array(L.bins[:])
        assert start[0] == 0 and np.all(np.diff(start[:cn + 1]) >= 0) and np.all(start[cn:] == start[cn])
        assert start[cn] == fold.sum()
        p = rng.standard_normal(num).astype(np.float32)
        for c in range(cn):
            mine = bins[start[c]:start[c + 1]]
            assert np.array_equal(mine, np.flatnonzero(fold[c])), (cn, num, c)
            assert np.isclose(p[mine].sum(), (fold[c] * p).sum())
    assert lists_fn(fp, 65, 84, ctypes.byref(L)) == -1 and lists_fn(None, 12, 84, ctypes.byref(L)) == -1
    # the switched-off all-octave kernel adds |Q|^2 to a per-frame accumulator octave by octave, bin by bin (class
    # of bin j from these lists): for a partition into 12 classes that is k_cqt_chroma's float32 summation order
    fp = fold_fn(12, 84, 12, 32.703)
    assert lists_fn(fp, 12, 84, ctypes.byref(L)) == 0
    start, bins = np.array(L.start[:]), np.array(L.bins[:])
    cls = np.full(84, 255)
    for c in range(12):
        for q in range(start[c], start[c + 1]):
            assert cls[bins[q]] == 255
            cls[bins[q]] = c
    assert np.all(cls < 12)
    pw = (rng.standard_normal(84).astype(np.float32)) ** 2
    per_class = np.zeros(12, np.float32)
    for c in range(12):
        v = np.float32(0)
        for q in range(start[c], start[c + 1]):
            v = np.float32(v + pw[bins[q]])
        per_class[c] = v
    acc = np.zeros(12, np.float32)
    for octave in range(7):
        for j in range(12):
            acc[cls[12 * octave + j]] = np.float32(acc[cls[12 * octave + j]] + pw[12 * octave + j])
    assert np.array_equal(acc, per_class)


def test_cqt_resampler_tap_table_matches_numpy():
    """afx_cqt_dec_table (host side of k_cqt_pyramid's resampler product): T[d] = h[|d|] 2^15 as binary16 (hi, lo) words,
    [word][copy a][x] = T[x - 160 - 2a]; bit for bit against numpy.float16, hi + lo = the tap to 2^-21 of itself, zeros
    outside |d| <= 31, and the taps are the reference resampler's (oracle/restate.py: halfband_taps)"""
    import ctypes

    from audioflux_amd import _lib
    from oracle import restate
    lib = _lib.get_lib()
    fn = lib.afx_cqt_dec_table
    fn.restype = None
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    taps = np.asarray(restate.halfband_taps(), np.float64)[:32].astype(np.float32)
    out = np.zeros((2, 4, 352), np.uint16)
    fn(taps.ctypes.data, out.ctypes.data)
    for a in range(4):
        d = np.abs(np.arange(352) - 160 - 2 * a)
        v = np.where(d <= 31, np.ldexp(taps[np.minimum(d, 31)], 15), np.float32(0)).astype(np.float32)
        hi = v.astype(np.float16)
        lo = (v - hi.astype(np.float32)).astype(np.float16)
        assert np.array_equal(out[0, a].view(np.float16), hi) and np.array_equal(out[1, a].view(np.float16), lo), a
        nz = v != 0
        assert nz.sum() == 63 and np.array_equal(out[0, a][nz], hi.view(np.uint16)[nz])
        err = np.abs(hi.astype(np.float64) + lo.astype(np.float64) - v)
        assert np.all(err <= np.abs(v) * 2.0 ** -21)
    assert 8192 <= np.ldexp(taps[0], 15) < 16384  # the centre tap puts the table's peak into [2^13, 2^14)
