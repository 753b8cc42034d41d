"""PWT -- ctypes mirror of python/audioflux/pwt.py:14-270 over libaudioflux_mi355x.so: the pseudo
wavelet transform (CWT pipeline with the auditory filter bank as the frequency-domain bank).
Result (..., num, time) complex64, bands in ascending order as the C entry writes them."""
import ctypes
from ctypes import POINTER, c_float, c_int, c_longlong, c_void_p

import numpy as np

from . import _lib, _util
from .types import SpectralFilterBankNormalType, SpectralFilterBankScaleType, SpectralFilterBankStyleType


class PWT:
    def __init__(self, num=84, radix2_exp=12, samplate=32000, low_fre=None, high_fre=None, bin_per_octave=12,
                 scale_type=SpectralFilterBankScaleType.OCTAVE, style_type=SpectralFilterBankStyleType.SLANEY,
                 normal_type=SpectralFilterBankNormalType.NONE, is_padding=True):
        self._lib = _lib.get_lib()
        self._obj = c_void_p(None)
        self.fft_length = 1 << radix2_exp
        if num > self.fft_length // 2 + 1:
            raise ValueError(f"num={num} is too large")
        if scale_type == SpectralFilterBankScaleType.OCTAVE and bin_per_octave < 1:
            raise ValueError(f"bin_per_octave={bin_per_octave} must be a positive integer")
        octave_like = scale_type in (SpectralFilterBankScaleType.OCTAVE, SpectralFilterBankScaleType.LOG)
        if low_fre is None:
            low_fre = 32.703195662574764 if octave_like else 0.0
        if high_fre is None:
            high_fre = samplate / 2
        if octave_like and low_fre < 32.703:
            raise ValueError(f"{scale_type.name} low_fre={low_fre} must be greater than or equal to 32.703")
        if low_fre < 0:
            raise ValueError(f"{scale_type.name} low_fre={low_fre} must be a non-negative number")
        self.num, self.radix2_exp, self.samplate = num, radix2_exp, samplate
        self.low_fre, self.high_fre, self.bin_per_octave = low_fre, high_fre, bin_per_octave
        self.scale_type, self.style_type, self.normal_type, self.is_padding = scale_type, style_type, normal_type, is_padding
        fn = self._lib.pwtObj_new
        fn.restype = c_int
        fn.argtypes = [POINTER(c_void_p), c_int, c_int, POINTER(c_int), POINTER(c_float), POINTER(c_float)] + \
                      [POINTER(c_int)] * 5
        st = fn(ctypes.byref(self._obj), num, radix2_exp, _util.opt_int(samplate), _util.opt_float(low_fre),
                _util.opt_float(high_fre), _util.opt_int(bin_per_octave), _util.opt_int(int(scale_type)),
                _util.opt_int(int(style_type)), _util.opt_int(int(normal_type)), _util.opt_int(int(is_padding)))
        if st != 0 or not self._obj:
            self._obj = c_void_p(None)
            raise RuntimeError(f"pwtObj_new failed with status {st}: {_lib.last_error()}")

    def get_fre_band_arr(self):
        fn = self._lib.pwtObj_getFreBandArr
        fn.argtypes, fn.restype = [c_void_p], POINTER(c_float)
        return np.ctypeslib.as_array(fn(self._obj), (self.num,)).copy()

    def get_bin_band_arr(self):
        fn = self._lib.pwtObj_getBinBandArr
        fn.argtypes, fn.restype = [c_void_p], POINTER(c_int)
        return np.ctypeslib.as_array(fn(self._obj), (self.num,)).copy()

    def _fit(self, x):
        n = self.fft_length  # truncate / zero-pad like utils/util.py:98-111
        if x.shape[-1] >= n:
            return np.ascontiguousarray(x[..., :n])
        out = np.zeros(x.shape[:-1] + (n,), np.float32)
        out[..., : x.shape[-1]] = x
        return out

    def _run(self, name, data_arr):
        x = self._fit(_util.as_f32(data_arr))
        clips, lead = _util.flatten_leading(x, 1)
        re = np.zeros((clips.shape[0], self.num, self.fft_length), np.float32)
        im = np.zeros_like(re)
        fn = getattr(self._lib, name)
        fn = _lib.checked(fn)
        fn.restype = None
        fn.argtypes = [c_void_p, _util.c_float_p, _util.c_float_p, _util.c_float_p]
        for i in range(clips.shape[0]):
            fn(self._obj, _util.fptr(clips[i]), _util.fptr(re[i]), _util.fptr(im[i]))
        return _util.restore_leading((re + 1j * im).astype(np.complex64), lead)

    def pwt(self, data_arr):
        """data_arr (..., 2**radix2_exp) -> complex64 (..., num, time)"""
        return self._run("pwtObj_pwt", data_arr)

    def enable_det(self, flag=True):
        fn = self._lib.pwtObj_enableDet
        fn.argtypes, fn.restype = [c_void_p, c_int], None
        fn(self._obj, int(flag))

    def pwt_det(self, data_arr):
        return self._run("pwtObj_pwtDet", data_arr)

    def pwt_device(self, x, stream=None):
        """Additive: x HIP torch.float32 (chunks, 2**radix2_exp) -> (real, imag) torch (chunks, num, n)"""
        import torch
        n = self.fft_length
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1 and x.shape[1] == n
        c = x.shape[0]
        re = torch.empty((c, self.num, n), dtype=torch.float32, device=x.device)
        im = torch.empty_like(re)
        s = stream if stream is not None else torch.cuda.current_stream(x.device)
        fn = self._lib.pwtObj_pwtBatchDevice
        fn.restype = c_int
        fn.argtypes = [c_void_p, c_void_p, c_int, c_longlong, c_void_p, c_void_p, c_void_p]
        _lib.check(fn(self._obj, x.data_ptr(), c, x.stride(0), re.data_ptr(), im.data_ptr(), s.cuda_stream),
                   "pwtObj_pwtBatchDevice")
        return re, im

    def __del__(self):
        if getattr(self, "_obj", None):
            fn = self._lib.pwtObj_free
            fn.argtypes, fn.restype = [c_void_p], None
            fn(self._obj)
            self._obj = c_void_p(None)
