#!/usr/bin/env python3
"""The DEVICE code of the time-domain CWT kernel (afx_cwt_td.hip) on the CPU (tests/emu): BASELINE cfg 4's chunk geometry
(morlet, 2^16-sample chunks, reflect padded; the 16 highest scales, to keep the emulation short) and a circular
(no padding, 2^17 samples) plan through the C host code -- the
plan builder of afx_cwt.c: double IFFT of the bank rows, truncation rule, pairing, f16 images -- the real launcher and
the kernel, one host thread per lane.  The other device kernels are stand-ins here, so only the rows the time-domain
plan owns are compared: against the compiled reference when it is present, else against the float64 restatement.
AFX_LIB = the library tests/test_emulated_kernels.py builds.  Prints one line per comparison and OK."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref, restate  # noqa: E402
from tests import cases  # noqa: E402

lib = C.CDLL(os.environ["AFX_LIB"])
vp = C.c_void_p
lib.afx_emulated_launches.restype = C.c_int
lib.afx_emulated_launches.argtypes = [C.c_char_p]
lib.cwtObj_getFreBandArr.restype = C.POINTER(C.c_float)


def run(is_padding, r, num, low_fre, clip, det=False):
    sr = 32000
    D = 1 << r
    h = vp()
    args = [C.byref(h), C.c_int(num), C.c_int(r), C.byref(C.c_int(sr)), C.byref(C.c_float(low_fre)), None,
            C.byref(C.c_int(12)), C.byref(C.c_int(1)), C.byref(C.c_int(5)), None, None, C.byref(C.c_int(is_padding))]
    st = lib.cwtObj_new(*args)
    assert st == 0, st
    x = np.resize(cases.real_audio(clip) if clip in cases.REAL_AUDIO else cases.hard_clip(clip), D).astype(np.float32)
    buf = np.zeros(D + 8, np.float32)
    xs = buf[1:1 + D]          # a chunk that starts 4 bytes off a 16-byte boundary: dword-aligned buffer loads only
    xs[:] = x
    re = np.zeros((1, num, D), np.float32)
    im = np.zeros_like(re)
    stream = (C.c_char * 8)()
    if det:
        lib.cwtObj_enableDet(h, C.c_int(1))  # the derivative bank's own time-domain plan (cwt_td_plan_det)
    before = lib.afx_emulated_launches(b"k_cwt_td")
    st = (lib.cwtObj_cwtDetBatchDevice if det else lib.cwtObj_cwtBatchDevice)(h, xs.ctypes.data_as(vp), 1, C.c_longlong(D), re.ctypes.data_as(vp), im.ctypes.data_as(vp),
                                   C.cast(stream, vp))
    assert st == 0, st
    launches = lib.afx_emulated_launches(b"k_cwt_td") - before
    assert launches >= 1, launches
    fre = np.ctypeslib.as_array(lib.cwtObj_getFreBandArr(h), (num,)).astype(np.float64)[::-1]
    lib.cwtObj_free(h)
    if ref.available():
        rr = ref.RefCWT(num=num, radix2_exp=r, samplate=sr, low_fre=low_fre, bin_per_octave=12, wavelet_type=1, scale_type=5,
                        is_padding=is_padding)
        wre, wim = rr.cwt(x, det=det)
        want, who = wre + 1j * wim, "compiled reference"
    elif det:
        print("derivative transform: no compiled reference here, skipped", flush=True)
        return
    else:
        want, who = restate.cwt(x.astype(np.float64), fre, sr, "morlet", 6.0, 2.0, bool(is_padding)), "float64 restatement"
    got = re[0] + 1j * im[0]
    # every scale of these plans has a short kernel: all rows come from the emulated kernel (a row the stand-in
    # FFT-path launchers wrote would be a constant)
    own = [j for j in range(num) if not np.all(re[0, j] == re[0, j, 0])]
    assert own == list(range(num)), own
    err = max(np.abs(got[j] - want[j]).max() / np.abs(want[j]).max() for j in own)
    print(f"pad {is_padding} 2^{r} clip {clip}{' derivative' if det else ''}: {len(own)} time-domain rows in {launches} launch(es), worst row {err:.2e} vs {who}", flush=True)
    if clip == "dc_offset":
        # a window that sits on a constant is split as x - middle (afx_cwt_td.hip): the rows must be CLOSER to the float64
        # evaluation than the float32 reference is (which is 4-6e-6 of a row's peak away from it on this clip)
        f64 = restate.cwt(x.astype(np.float64), fre, sr, "morlet", 6.0, 2.0, bool(is_padding))
        e64 = max(np.abs(got[j] - f64[j]).max() / np.abs(f64[j]).max() for j in own)
        r64 = max(np.abs(want[j] - f64[j]).max() / np.abs(f64[j]).max() for j in own)
        print(f"    against float64: {e64:.2e} (the reference: {r64:.2e})", flush=True)
        assert e64 <= 2e-6 and err <= 1e-5, (e64, err)
        return
    assert err <= 5e-6, err


def main():
    run(1, 16, 16, 1661.22, "voice")       # BASELINE cfg 4's chunk geometry (reflect padded, L = 2^17), its 16 highest scales
    run(0, 17, 6, 2793.83, "level_step")   # no padding: circular, L = 2^17 = the chunk itself
    run(1, 16, 16, 1661.22, "voice", det=True)  # cwtObj_cwtDet's scales on kernels IFFT(j w psi)
    run(1, 16, 16, 1661.22, "dc_offset")        # windows on a constant: the x - middle split and the kernel's bin-0 response
    print("OK")


if __name__ == "__main__":
    main()
