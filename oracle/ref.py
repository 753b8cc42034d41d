"""Raw ctypes driver of the compiled reference (oracle/_ref/libaudioflux_ref.so).

TEST INFRASTRUCTURE.  Calls the reference's exported C functions with the same
arguments the product library receives, so parity tests compare like with like.
Signatures: /root/reference/src/{bft,cwt,cqt,cepstrogram}_algorithm.h and
src/feature/xxcc_algorithm.h.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_PATH = os.path.join(_HERE, "_ref", "libaudioflux_ref.so")
_lib = None
fp = C.POINTER(C.c_float)
ip = C.POINTER(C.c_int)


def available():
    return os.path.exists(REF_PATH)


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(REF_PATH)
    return _lib


def _pi(v):
    return None if v is None else C.pointer(C.c_int(int(v)))


def _pf(v):
    return None if v is None else C.pointer(C.c_float(float(v)))


def _f(a):
    return a.ctypes.data_as(fp)


class RefBFT:
    def __init__(self, num, radix2_exp, samplate=None, low_fre=None, high_fre=None,
                 bin_per_octave=None, window_type=None, slide_length=None, scale_type=None,
                 style_type=None, normal_type=None, data_type=None, is_temporal=None, is_reassign=None):
        L = lib()
        self.L = L
        self.num = num
        self.obj = C.c_void_p(None)
        L.bftObj_new.restype = C.c_int
        L.bftObj_new.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, ip, fp, fp, ip, ip, ip,
                                 ip, ip, ip, ip, ip, ip]
        self.status = L.bftObj_new(C.byref(self.obj), num, radix2_exp, _pi(samplate), _pf(low_fre),
                                   _pf(high_fre), _pi(bin_per_octave), _pi(window_type),
                                   _pi(slide_length), _pi(scale_type), _pi(style_type),
                                   _pi(normal_type), _pi(data_type), _pi(is_reassign), _pi(is_temporal))
        L.bftObj_calTimeLength.argtypes = [C.c_void_p, C.c_int]
        L.bftObj_bft.restype = None
        L.bftObj_bft.argtypes = [C.c_void_p, fp, C.c_int, fp, fp]
        L.bftObj_setResultType.argtypes = [C.c_void_p, C.c_int]
        L.bftObj_setDataNormValue.argtypes = [C.c_void_p, C.c_float]
        L.bftObj_getFreBandArr.restype = fp
        L.bftObj_getFreBandArr.argtypes = [C.c_void_p]
        L.bftObj_getBinBandArr.restype = ip
        L.bftObj_getBinBandArr.argtypes = [C.c_void_p]
        L.bftObj_getTemporalData.restype = None
        L.bftObj_getTemporalData.argtypes = [C.c_void_p, C.POINTER(fp), C.POINTER(fp), C.POINTER(fp)]
        L.bftObj_free.argtypes = [C.c_void_p]
        self.result_type = 0

    def time_length(self, n):
        return self.L.bftObj_calTimeLength(self.obj, n)

    def set_result_type(self, t):
        self.L.bftObj_setResultType(self.obj, t)
        self.result_type = t

    def set_norm(self, v):
        self.L.bftObj_setDataNormValue(self.obj, v)

    def fre_band(self):
        return np.ctypeslib.as_array(self.L.bftObj_getFreBandArr(self.obj), (self.num,)).copy()

    def bin_band(self):
        return np.ctypeslib.as_array(self.L.bftObj_getBinBandArr(self.obj), (self.num,)).copy()

    def bft(self, x):
        """x[n] float32 -> (re[T,num], im[T,num])"""
        x = np.ascontiguousarray(x, np.float32)
        t = self.time_length(x.shape[0])
        re = np.zeros((t, self.num), np.float32)
        im = np.zeros((t, self.num), np.float32)
        self.L.bftObj_bft(self.obj, _f(x), x.shape[0], _f(re), _f(im))
        return re, im

    def temporal(self, t):
        e, r, z = fp(), fp(), fp()
        self.L.bftObj_getTemporalData(self.obj, C.byref(e), C.byref(r), C.byref(z))
        return tuple(np.ctypeslib.as_array(p, (t,)).copy() for p in (e, r, z))

    def __del__(self):
        if getattr(self, "obj", None):
            self.L.bftObj_free(self.obj)
            self.obj = C.c_void_p(None)


class RefXXCC:
    def __init__(self, num):
        L = lib()
        self.L = L
        self.num = num
        self.obj = C.c_void_p(None)
        L.xxccObj_new.argtypes = [C.POINTER(C.c_void_p), C.c_int]
        self.status = L.xxccObj_new(C.byref(self.obj), num)
        L.xxccObj_setTimeLength.argtypes = [C.c_void_p, C.c_int]
        L.xxccObj_setTimeLength.restype = None
        L.xxccObj_xxcc.restype = None
        L.xxccObj_xxcc.argtypes = [C.c_void_p, fp, C.c_int, ip, fp]
        L.xxccObj_xxccStandard.restype = None
        L.xxccObj_xxccStandard.argtypes = [C.c_void_p, fp, C.c_int, fp, ip, ip, ip, fp, fp, fp]
        L.xxccObj_free.argtypes = [C.c_void_p]

    def xxcc(self, m, cc_num=13, rectify=None):
        """m[T,num] -> [T,cc_num]"""
        m = np.ascontiguousarray(m, np.float32)
        t = m.shape[0]
        out = np.zeros((t, cc_num), np.float32)
        self.L.xxccObj_setTimeLength(self.obj, t)
        self.L.xxccObj_xxcc(self.obj, _f(m), cc_num, _pi(rectify), _f(out))
        return out

    def standard(self, m, energy, cc_num=13, delta_len=None, energy_type=None, rectify=None):
        m = np.ascontiguousarray(m, np.float32)
        energy = np.ascontiguousarray(energy, np.float32)
        t = m.shape[0]
        n_out = cc_num + (1 if energy_type == 1 else 0)
        outs = [np.zeros((t, n_out), np.float32) for _ in range(3)]
        self.L.xxccObj_setTimeLength(self.obj, t)
        self.L.xxccObj_xxccStandard(self.obj, _f(m), cc_num, _f(energy), _pi(delta_len),
                                    _pi(energy_type), _pi(rectify), _f(outs[0]), _f(outs[1]),
                                    _f(outs[2]))
        return outs

    def __del__(self):
        if getattr(self, "obj", None):
            self.L.xxccObj_free(self.obj)
            self.obj = C.c_void_p(None)


class RefCepstrogram:
    def __init__(self, radix2_exp, window_type=None, slide_length=None):
        L = lib()
        self.L = L
        self.n = 1 << radix2_exp
        self.obj = C.c_void_p(None)
        L.cepstrogramObj_new.argtypes = [C.POINTER(C.c_void_p), C.c_int, ip, ip]
        self.status = L.cepstrogramObj_new(C.byref(self.obj), radix2_exp, _pi(window_type), _pi(slide_length))
        L.cepstrogramObj_calTimeLength.argtypes = [C.c_void_p, C.c_int]
        L.cepstrogramObj_cepstrogram.restype = None
        L.cepstrogramObj_cepstrogram.argtypes = [C.c_void_p, C.c_int, fp, C.c_int, fp, fp, fp]
        L.cepstrogramObj_free.argtypes = [C.c_void_p]

    def cepstrogram(self, x, cep_num):
        x = np.ascontiguousarray(x, np.float32)
        t = self.L.cepstrogramObj_calTimeLength(self.obj, x.shape[0])
        f = self.n // 2 + 1
        outs = [np.zeros((t, f), np.float32) for _ in range(3)]
        self.L.cepstrogramObj_cepstrogram(self.obj, cep_num, _f(x), x.shape[0], _f(outs[0]), _f(outs[1]), _f(outs[2]))
        return outs

    def __del__(self):
        if getattr(self, "obj", None):
            self.L.cepstrogramObj_free(self.obj)
            self.obj = C.c_void_p(None)


class RefCQT:
    def __init__(self, num=84, samplate=None, min_fre=None, bin_per_octave=None, factor=None, beta=None,
                 thresh=None, window_type=None, slide_length=None, normal_type=None, is_scale=None, is_continue=None):
        L = lib()
        self.L = L
        self.num = num
        self.obj = C.c_void_p(None)
        L.cqtObj_newWith.argtypes = [C.POINTER(C.c_void_p), C.c_int, ip, fp, ip, fp, fp, fp, ip, ip, ip, ip, ip]
        self.status = L.cqtObj_newWith(C.byref(self.obj), num, _pi(samplate), _pf(min_fre),
                                       _pi(bin_per_octave), _pf(factor), _pf(beta), _pf(thresh),
                                       _pi(window_type), _pi(slide_length), _pi(is_continue), _pi(normal_type),
                                       _pi(is_scale))
        L.cqtObj_calTimeLength.argtypes = [C.c_void_p, C.c_int]
        L.cqtObj_getFFTLength.argtypes = [C.c_void_p]
        L.cqtObj_getFreBandArr.argtypes = [C.c_void_p]
        L.cqtObj_getFreBandArr.restype = fp
        L.cqtObj_cqt.restype = None
        L.cqtObj_cqt.argtypes = [C.c_void_p, fp, C.c_int, fp, fp]
        L.cqtObj_chroma.restype = None
        L.cqtObj_chroma.argtypes = [C.c_void_p, ip, ip, ip, fp, fp, fp]
        L.cqtObj_cqcc.restype = None
        L.cqtObj_cqcc.argtypes = [C.c_void_p, fp, C.c_int, ip, fp]
        L.cqtObj_cqhc.restype = None
        L.cqtObj_cqhc.argtypes = [C.c_void_p, fp, C.c_int, fp]
        L.cqtObj_deconv.restype = None
        L.cqtObj_deconv.argtypes = [C.c_void_p, fp, fp, fp]
        L.cqtObj_free.argtypes = [C.c_void_p]

    def fft_length(self):
        return self.L.cqtObj_getFFTLength(self.obj)

    def fre_band(self):
        return np.ctypeslib.as_array(self.L.cqtObj_getFreBandArr(self.obj), (self.num,)).copy()

    def cqt(self, x):
        x = np.ascontiguousarray(x, np.float32)
        t = self.L.cqtObj_calTimeLength(self.obj, x.shape[0])
        re = np.zeros((t, self.num), np.float32)
        im = np.zeros((t, self.num), np.float32)
        self.L.cqtObj_cqt(self.obj, _f(x), x.shape[0], _f(re), _f(im))
        return re, im

    def chroma(self, re, im, chroma_num=12, data_type=None, norm_type=None):
        re = np.ascontiguousarray(re, np.float32)
        im = np.ascontiguousarray(im, np.float32)
        out = np.zeros((re.shape[0], chroma_num), np.float32)
        self.L.cqtObj_chroma(self.obj, _pi(chroma_num), _pi(data_type), _pi(norm_type), _f(re), _f(im), _f(out))
        return out

    def cqcc(self, mag, cc_num=13, rectify=None):
        mag = np.ascontiguousarray(mag, np.float32)
        out = np.zeros((mag.shape[0], cc_num), np.float32)
        self.L.cqtObj_cqcc(self.obj, _f(mag), cc_num, _pi(rectify), _f(out))
        return out

    def cqhc(self, mag, hc_num=20):
        """mag [T,num] of the LAST cqt call -> [T,hc_num] (src/cqt_algorithm.c:662-711)"""
        mag = np.ascontiguousarray(mag, np.float32)
        out = np.zeros((mag.shape[0], hc_num), np.float32)
        self.L.cqtObj_cqhc(self.obj, _f(mag), hc_num, _f(out))
        return out

    def deconv(self, mag):
        """mag [T,num] -> (timbre [T,num], pitch [T,num]) (src/cqt_algorithm.c:718-781)"""
        mag = np.ascontiguousarray(mag, np.float32)
        tone = np.zeros_like(mag)
        pitch = np.zeros_like(mag)
        self.L.cqtObj_deconv(self.obj, _f(mag), _f(tone), _f(pitch))
        return tone, pitch

    def __del__(self):
        if getattr(self, "obj", None):
            self.L.cqtObj_free(self.obj)
            self.obj = C.c_void_p(None)


class RefCWT:
    def __init__(self, num, radix2_exp, samplate=None, low_fre=None, high_fre=None, bin_per_octave=None,
                 wavelet_type=None, scale_type=None, gamma=None, beta=None, is_padding=None):
        L = lib()
        self.L = L
        self.num = num
        self.n = 1 << radix2_exp
        self.obj = C.c_void_p(None)
        L.cwtObj_new.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, ip, fp, fp, ip, ip, ip, fp, fp, ip]
        self.status = L.cwtObj_new(C.byref(self.obj), num, radix2_exp, _pi(samplate), _pf(low_fre),
                                   _pf(high_fre), _pi(bin_per_octave), _pi(wavelet_type),
                                   _pi(scale_type), _pf(gamma), _pf(beta), _pi(is_padding))
        L.cwtObj_getFreBandArr.restype = fp
        L.cwtObj_getFreBandArr.argtypes = [C.c_void_p]
        L.cwtObj_getBinBandArr.restype = ip
        L.cwtObj_getBinBandArr.argtypes = [C.c_void_p]
        for f in (L.cwtObj_cwt, L.cwtObj_cwtDet):
            f.restype = None
            f.argtypes = [C.c_void_p, fp, fp, fp]
        L.cwtObj_enableDet.argtypes = [C.c_void_p, C.c_int]
        L.cwtObj_free.argtypes = [C.c_void_p]

    def fre_band(self):
        return np.ctypeslib.as_array(self.L.cwtObj_getFreBandArr(self.obj), (self.num,)).copy()

    def bin_band(self):
        return np.ctypeslib.as_array(self.L.cwtObj_getBinBandArr(self.obj), (self.num,)).copy()

    def cwt(self, x, det=False):
        x = np.ascontiguousarray(x, np.float32)
        re = np.zeros((self.num, self.n), np.float32)
        im = np.zeros((self.num, self.n), np.float32)
        if det:
            self.L.cwtObj_enableDet(self.obj, 1)
            self.L.cwtObj_cwtDet(self.obj, _f(x), _f(re), _f(im))
        else:
            self.L.cwtObj_cwt(self.obj, _f(x), _f(re), _f(im))
        return re, im

    def __del__(self):
        if getattr(self, "obj", None):
            self.L.cwtObj_free(self.obj)
            self.obj = C.c_void_p(None)


def mel_mfcc(x_clips, num=128, radix2_exp=11, samplate=16000, hop=512, cc_num=13):
    """reference mel (real power) + MFCC for each clip of x_clips[b,n]; the timed
    region of BASELINE.md section 3.  Returns (mel[b,T,num], mfcc[b,T,cc])."""
    bft = RefBFT(num, radix2_exp, samplate=samplate, low_fre=0.0, high_fre=samplate / 2.0,
                 window_type=1, slide_length=hop, scale_type=2, style_type=0, normal_type=0,
                 data_type=0)
    bft.set_result_type(1)
    cc = RefXXCC(num)
    mels, ccs = [], []
    for x in x_clips:
        re, _ = bft.bft(x)
        mels.append(re)
        ccs.append(cc.xxcc(re, cc_num, 0))
    return np.stack(mels), np.stack(ccs)


class RefSTFT:
    """src/stft_algorithm.h:16-39"""

    def __init__(self, radix2_exp, window_type=None, slide_length=None, is_continue=None):
        L = lib()
        self.L = L
        self.n = 1 << radix2_exp
        self.obj = C.c_void_p(None)
        L.stftObj_new.restype = C.c_int
        L.stftObj_new.argtypes = [C.POINTER(C.c_void_p), C.c_int, ip, ip, ip]
        self.status = L.stftObj_new(C.byref(self.obj), radix2_exp, _pi(window_type), _pi(slide_length),
                                    _pi(is_continue))
        L.stftObj_calTimeLength.argtypes = [C.c_void_p, C.c_int]
        L.stftObj_calDataLength.argtypes = [C.c_void_p, C.c_int]
        L.stftObj_enablePadding.restype = None
        L.stftObj_enablePadding.argtypes = [C.c_void_p, C.c_int]
        L.stftObj_enableContinue.restype = None
        L.stftObj_enableContinue.argtypes = [C.c_void_p, C.c_int]
        L.stftObj_setPadding.restype = None
        L.stftObj_setPadding.argtypes = [C.c_void_p, ip, ip, fp, fp]
        L.stftObj_setSlideLength.restype = None
        L.stftObj_setSlideLength.argtypes = [C.c_void_p, C.c_int]
        L.stftObj_useWindowDataArr.restype = None
        L.stftObj_useWindowDataArr.argtypes = [C.c_void_p, fp]
        L.stftObj_getWindowDataArr.restype = fp
        L.stftObj_getWindowDataArr.argtypes = [C.c_void_p]
        L.stftObj_stft.restype = None
        L.stftObj_stft.argtypes = [C.c_void_p, fp, C.c_int, fp, fp]
        L.stftObj_istft.restype = None
        L.stftObj_istft.argtypes = [C.c_void_p, fp, fp, C.c_int, C.c_int, fp]
        L.stftObj_free.argtypes = [C.c_void_p]

    def enable_padding(self, flag):
        self.L.stftObj_enablePadding(self.obj, int(flag))

    def set_padding(self, position=None, mode=None, value1=None, value2=None):
        self.L.stftObj_setPadding(self.obj, _pi(position), _pi(mode), _pf(value1), _pf(value2))

    def use_window(self, w):
        w = np.ascontiguousarray(w, np.float32)
        self.L.stftObj_useWindowDataArr(self.obj, _f(w))

    def window(self):
        return np.ctypeslib.as_array(self.L.stftObj_getWindowDataArr(self.obj), (self.n,)).copy()

    def time_length(self, n):
        return self.L.stftObj_calTimeLength(self.obj, n)

    def data_length(self, t):
        return self.L.stftObj_calDataLength(self.obj, t)

    def stft(self, x):
        """x[n] -> (re[T, N], im[T, N]); T as reported BEFORE the call (streaming tail included)"""
        x = np.ascontiguousarray(x, np.float32)
        t = self.time_length(x.shape[0])
        re = np.zeros((t, self.n), np.float32)
        im = np.zeros((t, self.n), np.float32)
        self.L.stftObj_stft(self.obj, _f(x), x.shape[0], _f(re), _f(im))
        return re, im

    def istft(self, re, im, method=0, init=None):
        re = np.ascontiguousarray(re, np.float32)
        im = np.ascontiguousarray(im, np.float32)
        n = self.data_length(re.shape[0])
        out = np.zeros(n, np.float32) if init is None else np.ascontiguousarray(init, np.float32).copy()
        self.L.stftObj_istft(self.obj, _f(re), _f(im), re.shape[0], int(method), _f(out))
        return out

    def __del__(self):
        if getattr(self, "obj", None):
            self.L.stftObj_free(self.obj)
            self.obj = C.c_void_p(None)


class RefSpectrogram:
    """src/spectrogram_algorithm.h:43-106 (core entry points)"""

    def __init__(self, num=0, samplate=None, low_fre=None, high_fre=None, bin_per_octave=None, radix2_exp=None,
                 window_type=None, slide_length=None, is_continue=None, data_type=None, scale_type=None,
                 style_type=None, normal_type=None, preset=None):
        L = lib()
        self.L = L
        self.obj = C.c_void_p(None)
        if preset:  # ("Mel", num, samplate, radix2_exp) | ("Linear"/"Chroma", samplate, radix2_exp)
            fn = getattr(L, "spectrogramObj_new" + preset[0])
            fn.restype = C.c_int
            ints = [int(v) for v in preset[1:]]
            fn.argtypes = [C.POINTER(C.c_void_p)] + [C.c_int] * len(ints) + [ip]
            self.status = fn(C.byref(self.obj), *ints, _pi(is_continue))
            self.n = 1 << ints[-1]
        else:
            L.spectrogramObj_new.restype = C.c_int
            L.spectrogramObj_new.argtypes = [C.POINTER(C.c_void_p), C.c_int, ip, fp, fp] + [ip] * 9
            self.status = L.spectrogramObj_new(C.byref(self.obj), num, _pi(samplate), _pf(low_fre), _pf(high_fre),
                                               _pi(bin_per_octave), _pi(radix2_exp), _pi(window_type),
                                               _pi(slide_length), _pi(is_continue), _pi(data_type),
                                               _pi(scale_type), _pi(style_type), _pi(normal_type))
            self.n = 1 << (12 if radix2_exp is None else radix2_exp)
        L.spectrogramObj_calTimeLength.argtypes = [C.c_void_p, C.c_int]
        L.spectrogramObj_getBandNum.argtypes = [C.c_void_p]
        L.spectrogramObj_getFreBandArr.restype = fp
        L.spectrogramObj_getFreBandArr.argtypes = [C.c_void_p]
        L.spectrogramObj_getBinBandArr.restype = ip
        L.spectrogramObj_getBinBandArr.argtypes = [C.c_void_p]
        L.spectrogramObj_setDataNormValue.restype = None
        L.spectrogramObj_setDataNormValue.argtypes = [C.c_void_p, C.c_float]
        L.spectrogramObj_setChromaDataNormalType.restype = None
        L.spectrogramObj_setChromaDataNormalType.argtypes = [C.c_void_p, C.c_int]
        L.spectrogramObj_spectrogram.restype = None
        L.spectrogramObj_spectrogram.argtypes = [C.c_void_p, fp, C.c_int, fp, fp]
        L.spectrogramObj_spectrogram1.restype = None
        L.spectrogramObj_spectrogram1.argtypes = [C.c_void_p, fp, fp, C.c_int, C.c_int, fp, fp]
        for name in ("mfcc", "bfcc", "gtcc"):
            f = getattr(L, "spectrogramObj_" + name)
            f.restype, f.argtypes = None, [C.c_void_p, fp, C.c_int, fp]
        L.spectrogramObj_xxcc.restype = None
        L.spectrogramObj_xxcc.argtypes = [C.c_void_p, fp, C.c_int, ip, fp]
        L.spectrogramObj_deconv.restype = None
        L.spectrogramObj_deconv.argtypes = [C.c_void_p, fp, fp, fp]
        L.spectrogramObj_free.argtypes = [C.c_void_p]
        self.num = L.spectrogramObj_getBandNum(self.obj) if self.status == 0 else 0

    def set_norm(self, v):
        self.L.spectrogramObj_setDataNormValue(self.obj, v)

    def set_chroma_norm(self, t):
        self.L.spectrogramObj_setChromaDataNormalType(self.obj, int(t))

    def time_length(self, n):
        return self.L.spectrogramObj_calTimeLength(self.obj, n)

    def fre_band(self, count=None):
        return np.ctypeslib.as_array(self.L.spectrogramObj_getFreBandArr(self.obj), (count or self.num,)).copy()

    def bin_band(self, count=None):
        return np.ctypeslib.as_array(self.L.spectrogramObj_getBinBandArr(self.obj), (count or self.num,)).copy()

    def spectrogram(self, x, phase=False):
        x = np.ascontiguousarray(x, np.float32)
        t = self.time_length(x.shape[0])
        out = np.zeros((t, self.num), np.float32)
        ph = np.zeros((t, self.num), np.float32) if phase else None
        self.L.spectrogramObj_spectrogram(self.obj, _f(x), x.shape[0], _f(out), _f(ph) if phase else None)
        return (out, ph) if phase else out

    def spectrogram1(self, re, im, phase=False):
        re, im = np.ascontiguousarray(re, np.float32), np.ascontiguousarray(im, np.float32)
        out = np.zeros((re.shape[0], self.num), np.float32)
        ph = np.zeros((re.shape[0], self.num), np.float32) if phase else None
        self.L.spectrogramObj_spectrogram1(self.obj, _f(re), _f(im), re.shape[0], re.shape[1], _f(out),
                                           _f(ph) if phase else None)
        return (out, ph) if phase else out

    def cc(self, kind, m, cc_num, rectify=None):
        m = np.ascontiguousarray(m, np.float32)
        out = np.zeros((m.shape[0], cc_num), np.float32)
        if kind == "xxcc":
            self.L.spectrogramObj_xxcc(self.obj, _f(m), cc_num, _pi(rectify), _f(out))
        else:
            getattr(self.L, "spectrogramObj_" + kind)(self.obj, _f(m), cc_num, _f(out))
        return out

    def deconv(self, m):
        m = np.ascontiguousarray(m, np.float32)
        a, b = np.zeros_like(m), np.zeros_like(m)
        self.L.spectrogramObj_deconv(self.obj, _f(m), _f(a), _f(b))
        return a, b

    def __del__(self):
        if getattr(self, "obj", None):
            self.L.spectrogramObj_free(self.obj)
            self.obj = C.c_void_p(None)


class RefPWT:
    """src/pwt_algorithm.h:14-31"""

    def __init__(self, num, radix2_exp, samplate=None, low_fre=None, high_fre=None, bin_per_octave=None,
                 scale_type=None, style_type=None, normal_type=None, is_padding=None):
        L = lib()
        self.L = L
        self.num = num
        self.n = 1 << radix2_exp
        self.obj = C.c_void_p(None)
        L.pwtObj_new.restype = C.c_int
        L.pwtObj_new.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, ip, fp, fp, ip, ip, ip, ip, ip]
        self.status = L.pwtObj_new(C.byref(self.obj), num, radix2_exp, _pi(samplate), _pf(low_fre), _pf(high_fre),
                                   _pi(bin_per_octave), _pi(scale_type), _pi(style_type), _pi(normal_type),
                                   _pi(is_padding))
        L.pwtObj_getFreBandArr.restype = fp
        L.pwtObj_getFreBandArr.argtypes = [C.c_void_p]
        L.pwtObj_getBinBandArr.restype = ip
        L.pwtObj_getBinBandArr.argtypes = [C.c_void_p]
        for f in (L.pwtObj_pwt, L.pwtObj_pwtDet):
            f.restype = None
            f.argtypes = [C.c_void_p, fp, fp, fp]
        L.pwtObj_enableDet.argtypes = [C.c_void_p, C.c_int]
        L.pwtObj_free.argtypes = [C.c_void_p]

    def fre_band(self):
        return np.ctypeslib.as_array(self.L.pwtObj_getFreBandArr(self.obj), (self.num,)).copy()

    def bin_band(self):
        return np.ctypeslib.as_array(self.L.pwtObj_getBinBandArr(self.obj), (self.num,)).copy()

    def pwt(self, x, det=False):
        x = np.ascontiguousarray(x, np.float32)
        re = np.zeros((self.num, self.n), np.float32)
        im = np.zeros((self.num, self.n), np.float32)
        if det:
            self.L.pwtObj_enableDet(self.obj, 1)
            self.L.pwtObj_pwtDet(self.obj, _f(x), _f(re), _f(im))
        else:
            self.L.pwtObj_pwt(self.obj, _f(x), _f(re), _f(im))
        return re, im

    def __del__(self):
        if getattr(self, "obj", None):
            self.L.pwtObj_free(self.obj)
            self.obj = C.c_void_p(None)


class RefWSST:
    """src/wsst_algorithm.h:28-49"""

    def __init__(self, num, radix2_exp, samplate=None, low_fre=None, high_fre=None, bin_per_octave=None,
                 wavelet_type=None, scale_type=None, gamma=None, beta=None, thresh=None, is_padding=None):
        L = lib()
        self.L = L
        self.num = num
        self.n = 1 << radix2_exp
        self.obj = C.c_void_p(None)
        L.wsstObj_new.restype = C.c_int
        L.wsstObj_new.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, ip, fp, fp, ip, ip, ip, fp, fp, fp, ip]
        self.status = L.wsstObj_new(C.byref(self.obj), num, radix2_exp, _pi(samplate), _pf(low_fre), _pf(high_fre),
                                    _pi(bin_per_octave), _pi(wavelet_type), _pi(scale_type), _pf(gamma), _pf(beta),
                                    _pf(thresh), _pi(is_padding))
        L.wsstObj_getFreBandArr.restype = fp
        L.wsstObj_getFreBandArr.argtypes = [C.c_void_p]
        L.wsstObj_wsst.restype = None
        L.wsstObj_wsst.argtypes = [C.c_void_p, fp, fp, fp, fp, fp]
        L.wsstObj_free.argtypes = [C.c_void_p]

    def fre_band(self):
        return np.ctypeslib.as_array(self.L.wsstObj_getFreBandArr(self.obj), (self.num,)).copy()

    def wsst(self, x):
        """-> (squeezed [num, n] complex, cwt [num, n] complex) in the C row order"""
        x = np.ascontiguousarray(x, np.float32)
        a = [np.zeros((self.num, self.n), np.float32) for _ in range(4)]
        self.L.wsstObj_wsst(self.obj, _f(x), _f(a[0]), _f(a[1]), _f(a[2]), _f(a[3]))
        return a[0] + 1j * a[1], a[2] + 1j * a[3]

    def __del__(self):
        if getattr(self, "obj", None):
            self.L.wsstObj_free(self.obj)
            self.obj = C.c_void_p(None)


class RefReassign:
    """src/reassign_algorithm.h:15-56"""

    def __init__(self, radix2_exp, samplate=None, window_type=None, slide_length=None, re_type=None, thresh=None,
                 is_padding=None):
        L = lib()
        self.L = L
        self.n = 1 << radix2_exp
        self.obj = C.c_void_p(None)
        L.reassignObj_new.restype = C.c_int
        L.reassignObj_new.argtypes = [C.POINTER(C.c_void_p), C.c_int, ip, ip, ip, ip, fp, ip, ip]
        self.status = L.reassignObj_new(C.byref(self.obj), radix2_exp, _pi(samplate), _pi(window_type),
                                        _pi(slide_length), _pi(re_type), _pf(thresh), _pi(is_padding), None)
        L.reassignObj_calTimeLength.argtypes = [C.c_void_p, C.c_int]
        L.reassignObj_setResultType.restype = None
        L.reassignObj_setResultType.argtypes = [C.c_void_p, C.c_int]
        L.reassignObj_setOrder.restype = None
        L.reassignObj_setOrder.argtypes = [C.c_void_p, C.c_int]
        L.reassignObj_reassign.restype = None
        L.reassignObj_reassign.argtypes = [C.c_void_p, fp, C.c_int, fp, fp, fp, fp]
        L.reassignObj_free.argtypes = [C.c_void_p]

    def set_result_type(self, t):
        self.L.reassignObj_setResultType(self.obj, int(t))

    def set_order(self, k):
        self.L.reassignObj_setOrder(self.obj, int(k))

    def time_length(self, n):
        return self.L.reassignObj_calTimeLength(self.obj, n)

    def reassign(self, x):
        """-> four float32 [T, F] planes: reassigned re, im, stft re, im"""
        x = np.ascontiguousarray(x, np.float32)
        t, f = self.time_length(x.shape[0]), self.n // 2 + 1
        a = [np.zeros((t, f), np.float32) for _ in range(4)]
        self.L.reassignObj_reassign(self.obj, _f(x), x.shape[0], _f(a[0]), _f(a[1]), _f(a[2]), _f(a[3]))
        return a

    def __del__(self):
        if getattr(self, "obj", None):
            self.L.reassignObj_free(self.obj)
            self.obj = C.c_void_p(None)


class RefSynsq:
    """src/synsq_algorithm.h:12-31"""

    def __init__(self, num, radix2_exp, samplate=None, order=None, thresh=None):
        L = lib()
        self.L = L
        self.num, self.n = num, 1 << radix2_exp
        self.obj = C.c_void_p(None)
        L.synsqObj_new.restype = C.c_int
        L.synsqObj_new.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, ip, ip, fp]
        self.status = L.synsqObj_new(C.byref(self.obj), num, radix2_exp, _pi(samplate), _pi(order), _pf(thresh))
        L.synsqObj_synsq.restype = None
        L.synsqObj_synsq.argtypes = [C.c_void_p, fp, C.c_int, fp, fp, fp, fp]
        L.synsqObj_free.argtypes = [C.c_void_p]

    def synsq(self, fre, scale_type, W):
        fre = np.ascontiguousarray(fre, np.float32)
        re, im = np.ascontiguousarray(W.real, np.float32), np.ascontiguousarray(W.imag, np.float32)
        a, b = np.zeros_like(re), np.zeros_like(re)
        self.L.synsqObj_synsq(self.obj, _f(fre), int(scale_type), _f(re), _f(im), _f(a), _f(b))
        return a + 1j * b

    def __del__(self):
        if getattr(self, "obj", None):
            self.L.synsqObj_free(self.obj)
            self.obj = C.c_void_p(None)
