"""Spectrogram family -- ctypes mirror of the reference wrapper classes
(python/audioflux/spectrogram.py:31-2809: SpectrogramBase, Spectrogram, MelSpectrogram,
BarkSpectrogram, ErbSpectrogram, Linear, Mel, Bark, Erb, Chroma) over libaudioflux_mi355x.so:
same constructor arguments, method names and (fre, time) result orientation.  The "deep"
scales and the spectral-descriptor methods are not part of this backend.  Extra:
`spectrogram_device` keeps clips and results in HBM."""
import ctypes
from ctypes import POINTER, c_float, c_int, c_longlong, c_void_p

import numpy as np

from . import _lib, _util
from .types import (CepstralRectifyType, ChromaDataNormalType, SpectralDataType,
                    SpectralFilterBankNormalType, SpectralFilterBankScaleType, SpectralFilterBankStyleType,
                    WindowType)

SpectralFilterBankType = SpectralFilterBankScaleType  # the reference wrapper's name for the enum
_C1 = 32.703195662574764  # note_to_hz('C1')
_LOW_ZERO = (SpectralFilterBankScaleType.LINEAR, SpectralFilterBankScaleType.LINSPACE,
             SpectralFilterBankScaleType.MEL, SpectralFilterBankScaleType.BARK, SpectralFilterBankScaleType.ERB)
_OCTAVE_LIKE = (SpectralFilterBankScaleType.OCTAVE, SpectralFilterBankScaleType.LOG,
                SpectralFilterBankScaleType.OCTAVE_CHROMA, SpectralFilterBankScaleType.DEEP,
                SpectralFilterBankScaleType.DEEP_CHROMA)


class SpectrogramBase:
    def __init__(self, num=0, samplate=32000, low_fre=None, high_fre=None, bin_per_octave=12, radix2_exp=12,
                 window_type=None, slide_length=None, data_type=SpectralDataType.POWER,
                 filter_bank_type=SpectralFilterBankScaleType.LINEAR,
                 style_type=SpectralFilterBankStyleType.SLANEY,
                 normal_type=SpectralFilterBankNormalType.NONE, is_continue=False):
        self._lib = _lib.get_lib()
        self._obj = c_void_p(None)
        if filter_bank_type in (SpectralFilterBankScaleType.OCTAVE, SpectralFilterBankScaleType.OCTAVE_CHROMA):
            if bin_per_octave not in (12, 24, 36):
                raise ValueError(f"bin_per_octave={bin_per_octave} must be 12, 24 or 36")
        if filter_bank_type == SpectralFilterBankScaleType.OCTAVE and num % bin_per_octave != 0:
            raise ValueError(f"num={num} must be an integer multiple of bin_per_octave={bin_per_octave}")
        if low_fre is None:
            low_fre = 0.0 if filter_bank_type in _LOW_ZERO else _C1
        if high_fre is None:
            high_fre = samplate / 2
        if window_type is None:
            deep = filter_bank_type in (SpectralFilterBankScaleType.DEEP, SpectralFilterBankScaleType.DEEP_CHROMA)
            window_type = WindowType.HAMM if deep else WindowType.HANN
        if filter_bank_type in _OCTAVE_LIKE and low_fre < round(_C1, 3):
            raise ValueError(f"{filter_bank_type.name} low_fre={low_fre} must be greater than or equal to 32.703")
        if low_fre < 0:
            raise ValueError(f"{filter_bank_type.name} low_fre={low_fre} must be a non-negative number")
        fft_length = 1 << radix2_exp
        if slide_length is None:
            slide_length = fft_length // 4
        self.num, self.samplate, self.low_fre, self.high_fre = num, samplate, low_fre, high_fre
        self.bin_per_octave, self.radix2_exp, self.window_type = bin_per_octave, radix2_exp, window_type
        self.slide_length, self.is_continue, self.data_type = slide_length, bool(is_continue), data_type
        self.filter_bank_type, self.style_type, self.normal_type = filter_bank_type, style_type, normal_type
        self.fft_length = fft_length

    def _new(self):
        fn = self._lib.spectrogramObj_new
        fn.restype = c_int
        fn.argtypes = [POINTER(c_void_p), c_int, POINTER(c_int), POINTER(c_float), POINTER(c_float)] + \
                      [POINTER(c_int)] * 9
        st = fn(ctypes.byref(self._obj), int(self.num), _util.opt_int(self.samplate), _util.opt_float(self.low_fre),
                _util.opt_float(self.high_fre), _util.opt_int(self.bin_per_octave), _util.opt_int(self.radix2_exp),
                _util.opt_int(int(self.window_type)), _util.opt_int(self.slide_length),
                _util.opt_int(int(self.is_continue)), _util.opt_int(int(self.data_type)),
                _util.opt_int(int(self.filter_bank_type)), _util.opt_int(int(self.style_type)),
                _util.opt_int(int(self.normal_type)))
        self._check_created(st, "spectrogramObj_new")
        if self.filter_bank_type == SpectralFilterBankScaleType.LINEAR:
            self.num = self.get_band_num()

    def _new_preset(self, name, with_num):
        fn = getattr(self._lib, name)
        fn.restype = c_int
        fn.argtypes = [POINTER(c_void_p)] + ([c_int] if with_num else []) + [c_int, c_int, POINTER(c_int)]
        args = ([int(self.num)] if with_num else []) + [int(self.samplate), int(self.radix2_exp),
                                                        _util.opt_int(int(self.is_continue))]
        self._check_created(fn(ctypes.byref(self._obj), *args), name)

    def _check_created(self, st, what):
        if st != 0 or not self._obj:
            self._obj = c_void_p(None)
            raise RuntimeError(f"{what} failed with status {st}: {_lib.last_error()}")

    # -- switches and plan queries ------------------------------------------
    def set_data_norm_value(self, norm_value):
        fn = self._lib.spectrogramObj_setDataNormValue
        fn.restype, fn.argtypes = None, [c_void_p, c_float]
        fn(self._obj, float(norm_value))

    def set_chroma_data_normal_type(self, data_norm_type):
        fn = self._lib.spectrogramObj_setChromaDataNormalType
        fn.restype, fn.argtypes = None, [c_void_p, c_int]
        fn(self._obj, int(data_norm_type))

    def cal_time_length(self, data_length):
        fn = self._lib.spectrogramObj_calTimeLength
        fn.restype, fn.argtypes = c_int, [c_void_p, c_int]
        return int(fn(self._obj, int(data_length)))

    def get_fre_band_arr(self):
        fn = self._lib.spectrogramObj_getFreBandArr
        fn.argtypes, fn.restype = [c_void_p], POINTER(c_float)
        return np.ctypeslib.as_array(fn(self._obj), (self.num,)).copy()

    def get_bin_band_arr(self):
        fn = self._lib.spectrogramObj_getBinBandArr
        fn.argtypes, fn.restype = [c_void_p], POINTER(c_int)
        return np.ctypeslib.as_array(fn(self._obj), (self.num,)).copy()

    def get_band_num(self):
        fn = self._lib.spectrogramObj_getBandNum
        fn.restype, fn.argtypes = c_int, [c_void_p]
        return int(fn(self._obj))

    def get_bin_band_length(self):
        fn = self._lib.spectrogramObj_getBinBandLength
        fn.restype, fn.argtypes = c_int, [c_void_p]
        return int(fn(self._obj))

    # -- transforms ---------------------------------------------------------
    def spectrogram(self, data_arr, is_phase_arr=False):
        """data_arr (..., n) -> (..., num, time) [, phase (..., num, time) for the linear scale]"""
        x = _util.as_f32(data_arr)
        n = x.shape[-1]
        if n < self.fft_length:
            raise ValueError(f"radix2_exp={self.radix2_exp}(fft_length={self.fft_length}) is too large for "
                             f"data_arr length={n}")
        if is_phase_arr and self.filter_bank_type != SpectralFilterBankScaleType.LINEAR:
            raise ValueError("Only LINEAR bank type has phase arr")
        clips, lead = _util.flatten_leading(x, 1)
        fn = self._lib.spectrogramObj_spectrogram
        fn = _lib.checked(fn)
        fn.restype = None
        fn.argtypes = [c_void_p, _util.c_float_p, c_int, _util.c_float_p, _util.c_float_p]
        specs, phases = [], []
        for i in range(clips.shape[0]):
            t = self.cal_time_length(n)
            spec = np.zeros((t, self.num), np.float32)
            ph = np.zeros((t, self.num), np.float32) if is_phase_arr else None
            fn(self._obj, _util.fptr(clips[i]), n, _util.fptr(spec), _util.fptr(ph) if is_phase_arr else None)
            specs.append(spec)
            phases.append(ph)
        out = np.ascontiguousarray(np.swapaxes(_util.restore_leading(np.stack(specs), lead), -1, -2))
        if is_phase_arr:
            return out, np.ascontiguousarray(np.swapaxes(_util.restore_leading(np.stack(phases), lead), -1, -2))
        return out

    def spectrogram_from_stft(self, m_real_arr, m_imag_arr, is_phase_arr=False):
        """spectrogramObj_spectrogram1: a caller-supplied STFT (time, fft_length) re / im -> (num, time)"""
        re, im = _util.as_f32(m_real_arr), _util.as_f32(m_imag_arr)
        assert re.ndim == 2 and re.shape == im.shape
        t, m = re.shape
        spec = np.zeros((t, self.num), np.float32)
        ph = np.zeros((t, self.num), np.float32) if is_phase_arr else None
        fn = self._lib.spectrogramObj_spectrogram1
        fn = _lib.checked(fn)
        fn.restype = None
        fn.argtypes = [c_void_p, _util.c_float_p, _util.c_float_p, c_int, c_int, _util.c_float_p, _util.c_float_p]
        fn(self._obj, _util.fptr(re), _util.fptr(im), t, m, _util.fptr(spec), _util.fptr(ph) if is_phase_arr else None)
        return (np.ascontiguousarray(spec.T), np.ascontiguousarray(ph.T)) if is_phase_arr else np.ascontiguousarray(spec.T)

    def spectrogram_device(self, x, out=None, stream=None):
        """Additive: x CUDA/HIP torch.float32 (clips, n) -> torch (clips, time, num), asynchronous
        on `stream` or torch's current stream; the streaming tail is not used."""
        import torch
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
        b, n = x.shape
        t = (n - self.fft_length) // self.slide_length + 1 if n >= self.fft_length else 0
        if out is None:
            out = torch.empty((b, t, self.num), dtype=torch.float32, device=x.device)
        s = stream if stream is not None else torch.cuda.current_stream(x.device)
        fn = self._lib.spectrogramObj_spectrogramBatchDevice
        fn.restype = c_int
        fn.argtypes = [c_void_p, c_void_p, c_int, c_int, c_longlong, c_void_p, c_void_p]
        _lib.check(fn(self._obj, x.data_ptr(), b, n, x.stride(0), out.data_ptr(), s.cuda_stream),
                   "spectrogramObj_spectrogramBatchDevice")
        return out

    def _cc(self, name, m_data_arr, cc_num, rectify=None):
        if cc_num > self.num:
            raise ValueError(f"cc_num={cc_num} must be less than num={self.num}")
        m = np.ascontiguousarray(_util.as_f32(m_data_arr).T)
        out = np.zeros((m.shape[0], cc_num), np.float32)
        fn = getattr(self._lib, name)
        fn = _lib.checked(fn)
        fn.restype = None
        if rectify is None:
            fn.argtypes = [c_void_p, _util.c_float_p, c_int, _util.c_float_p]
            fn(self._obj, _util.fptr(m), int(cc_num), _util.fptr(out))
        else:
            fn.argtypes = [c_void_p, _util.c_float_p, c_int, POINTER(c_int), _util.c_float_p]
            fn(self._obj, _util.fptr(m), int(cc_num), _util.opt_int(int(rectify)), _util.fptr(out))
        return np.ascontiguousarray(out.T)

    def mfcc(self, m_data_arr, cc_num=13):
        if not (self.filter_bank_type == SpectralFilterBankScaleType.MEL
                and self.style_type == SpectralFilterBankStyleType.SLANEY):
            raise ValueError("``filter_bank_type`` must be ``MEL`` and ``style_type`` must be ``SLANEY``")
        return self._cc("spectrogramObj_mfcc", m_data_arr, cc_num)

    def bfcc(self, m_data_arr, cc_num=13):
        if not (self.filter_bank_type == SpectralFilterBankScaleType.BARK
                and self.style_type == SpectralFilterBankStyleType.SLANEY):
            raise ValueError("``filter_bank_type`` must be ``BARK`` and ``style_type`` must be ``SLANEY``")
        return self._cc("spectrogramObj_bfcc", m_data_arr, cc_num)

    def gtcc(self, m_data_arr, cc_num=13):
        if not (self.filter_bank_type == SpectralFilterBankScaleType.ERB
                and self.style_type == SpectralFilterBankStyleType.GAMMATONE):
            raise ValueError("``filter_bank_type`` must be ``ERB`` and ``style_type`` must be ``GAMMATONE``")
        return self._cc("spectrogramObj_gtcc", m_data_arr, cc_num)

    def xxcc(self, m_data_arr, cc_num=13, rectify_type=CepstralRectifyType.LOG):
        return self._cc("spectrogramObj_xxcc", m_data_arr, cc_num, rectify_type)

    def deconv(self, m_data_arr):
        """(num, time) magnitudes -> (timbre, pitch), both (num, time)"""
        m = np.ascontiguousarray(_util.as_f32(m_data_arr).T)
        a, b = np.zeros_like(m), np.zeros_like(m)
        fn = self._lib.spectrogramObj_deconv
        fn.restype, fn.argtypes = None, [c_void_p, _util.c_float_p, _util.c_float_p, _util.c_float_p]
        fn(self._obj, _util.fptr(m), _util.fptr(a), _util.fptr(b))
        return np.ascontiguousarray(a.T), np.ascontiguousarray(b.T)

    def y_coords(self):
        return self.get_fre_band_arr()

    def __del__(self):
        if getattr(self, "_obj", None):
            fn = self._lib.spectrogramObj_free
            fn.argtypes, fn.restype = [c_void_p], None
            fn(self._obj)
            self._obj = c_void_p(None)


class Spectrogram(SpectrogramBase):
    """generic constructor (python/audioflux/spectrogram.py:1771-1946)"""

    def __init__(self, num=0, samplate=32000, low_fre=None, high_fre=None, bin_per_octave=12, radix2_exp=12,
                 window_type=None, slide_length=None, data_type=SpectralDataType.POWER,
                 filter_bank_type=SpectralFilterBankScaleType.LINEAR,
                 style_type=SpectralFilterBankStyleType.SLANEY,
                 normal_type=SpectralFilterBankNormalType.NONE, is_continue=False):
        super().__init__(num, samplate, low_fre, high_fre, bin_per_octave, radix2_exp, window_type, slide_length,
                         data_type, filter_bank_type, style_type, normal_type, is_continue)
        self._new()


def _scaled(scale):
    class _Cls(Spectrogram):
        def __init__(self, num=0, samplate=32000, low_fre=None, high_fre=None, radix2_exp=12,
                     window_type=WindowType.HANN, slide_length=None, data_type=SpectralDataType.POWER,
                     style_type=SpectralFilterBankStyleType.SLANEY,
                     normal_type=SpectralFilterBankNormalType.NONE):
            super().__init__(num=num, samplate=samplate, low_fre=low_fre, high_fre=high_fre, radix2_exp=radix2_exp,
                             window_type=window_type, slide_length=slide_length, data_type=data_type,
                             filter_bank_type=scale, style_type=style_type, normal_type=normal_type)
    return _Cls


MelSpectrogram = _scaled(SpectralFilterBankScaleType.MEL)     # spectrogram.py:1948-2054
MelSpectrogram.__name__ = "MelSpectrogram"
BarkSpectrogram = _scaled(SpectralFilterBankScaleType.BARK)   # :2056-2162
BarkSpectrogram.__name__ = "BarkSpectrogram"
ErbSpectrogram = _scaled(SpectralFilterBankScaleType.ERB)     # :2164-2270
ErbSpectrogram.__name__ = "ErbSpectrogram"


class Linear(SpectrogramBase):
    """spectrogramObj_newLinear (spectrogram.py:2272-2343)"""

    def __init__(self, samplate=32000, radix2_exp=12):
        super().__init__(num=0, samplate=samplate, low_fre=0.0, high_fre=samplate / 2, radix2_exp=radix2_exp,
                         window_type=WindowType.HANN, filter_bank_type=SpectralFilterBankScaleType.LINEAR)
        self._new_preset("spectrogramObj_newLinear", False)
        self.num = self.get_band_num()


def _preset(scale, fname):
    class _Cls(SpectrogramBase):
        def __init__(self, num=128, samplate=32000, radix2_exp=12):
            super().__init__(num=num, samplate=samplate, low_fre=0.0, high_fre=samplate / 2, radix2_exp=radix2_exp,
                             window_type=WindowType.HANN, filter_bank_type=scale)
            self._new_preset(fname, True)
    return _Cls


Mel = _preset(SpectralFilterBankScaleType.MEL, "spectrogramObj_newMel")      # :2345-2421
Mel.__name__ = "Mel"
Bark = _preset(SpectralFilterBankScaleType.BARK, "spectrogramObj_newBark")   # :2423-2503
Bark.__name__ = "Bark"
Erb = _preset(SpectralFilterBankScaleType.ERB, "spectrogramObj_newErb")      # :2505-2581
Erb.__name__ = "Erb"


class Chroma(SpectrogramBase):
    """spectrogramObj_newChroma (spectrogram.py:2583-2653)"""

    def __init__(self, samplate=32000, radix2_exp=12):
        super().__init__(num=12, samplate=samplate, low_fre=0.0, high_fre=None, radix2_exp=radix2_exp,
                         window_type=WindowType.HANN, filter_bank_type=SpectralFilterBankScaleType.CHROMA)
        self._new_preset("spectrogramObj_newChroma", False)
