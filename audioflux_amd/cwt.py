"""CWT -- ctypes mirror of python/audioflux/cwt.py:126-318 over libaudioflux_mi355x.so."""
import ctypes
from ctypes import POINTER, c_float, c_int, c_longlong, c_void_p

import numpy as np

from . import _lib, _util
from .types import SpectralFilterBankScaleType, WaveletContinueType

_DEFAULTS = {WaveletContinueType.MORSE: (3, 20), WaveletContinueType.MORLET: (6, 2),
             WaveletContinueType.BUMP: (5, 0.6), WaveletContinueType.PAUL: (4, 2),
             WaveletContinueType.DOG: (2, 2), WaveletContinueType.MEXICAN: (2, 2),
             WaveletContinueType.HERMIT: (5, 2), WaveletContinueType.RICKER: (4, 2)}


class CWT:
    def __init__(self, num=84, radix2_exp=12, samplate=32000, low_fre=None, high_fre=None,
                 bin_per_octave=12, wavelet_type=WaveletContinueType.MORSE,
                 scale_type=SpectralFilterBankScaleType.OCTAVE, gamma=None, beta=None,
                 is_padding=True):
        self._lib = _lib.get_lib()
        self._obj = c_void_p(None)
        self.fft_length = 1 << radix2_exp
        if num > self.fft_length // 2 + 1:
            raise ValueError(f"num={num} is too large")
        octave_like = scale_type in (SpectralFilterBankScaleType.OCTAVE, SpectralFilterBankScaleType.LOG)
        if low_fre is None:
            low_fre = 32.703 if octave_like else 0.0
        if high_fre is None:
            high_fre = samplate / 2
        if octave_like and low_fre < 32.703:
            raise ValueError(f"{scale_type.name} low_fre={low_fre} must be >= 32.703")
        dg, db = _DEFAULTS[WaveletContinueType(wavelet_type)]
        gamma = dg if gamma is None else gamma
        beta = db if beta is None else beta
        self.num, self.radix2_exp, self.samplate = num, radix2_exp, samplate
        self.low_fre, self.high_fre, self.bin_per_octave = low_fre, high_fre, bin_per_octave
        self.wavelet_type, self.scale_type = wavelet_type, scale_type
        self.gamma, self.beta, self.is_padding = gamma, beta, is_padding
        fn = self._lib.cwtObj_new
        fn.restype = c_int
        fn.argtypes = [POINTER(c_void_p), c_int, c_int, POINTER(c_int), POINTER(c_float),
                       POINTER(c_float), POINTER(c_int), POINTER(c_int), POINTER(c_int),
                       POINTER(c_float), POINTER(c_float), POINTER(c_int)]
        st = fn(ctypes.byref(self._obj), num, radix2_exp, _util.opt_int(samplate),
                _util.opt_float(low_fre), _util.opt_float(high_fre), _util.opt_int(bin_per_octave),
                _util.opt_int(int(wavelet_type)), _util.opt_int(int(scale_type)),
                _util.opt_float(gamma), _util.opt_float(beta), _util.opt_int(int(is_padding)))
        if st != 0 or not self._obj:
            self._obj = c_void_p(None)
            raise RuntimeError(f"cwtObj_new failed with status {st}: {_lib.last_error()}")

    def get_fre_band_arr(self):
        fn = self._lib.cwtObj_getFreBandArr
        fn.argtypes, fn.restype = [c_void_p], POINTER(c_float)
        return np.ctypeslib.as_array(fn(self._obj), (self.num,)).copy()

    def get_bin_band_arr(self):
        fn = self._lib.cwtObj_getBinBandArr
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
        out = _util.restore_leading((re + 1j * im).astype(np.complex64), lead)
        return np.ascontiguousarray(out[..., ::-1, :])  # row 0 of the C result is the HIGHEST frequency

    def cwt(self, data_arr):
        """data_arr (..., 2**radix2_exp) -> complex64 (..., num, time), ascending frequency"""
        return self._run("cwtObj_cwt", data_arr)

    def enable_det(self, flag=True):
        fn = self._lib.cwtObj_enableDet
        fn.argtypes = [c_void_p, c_int]
        fn = _lib.checked(fn)
        fn.restype = None
        fn(self._obj, int(flag))

    def cwt_det(self, data_arr):
        return self._run("cwtObj_cwtDet", data_arr)

    def cwt_device(self, x, out_real=None, out_imag=None, det=False, stream=None):
        """Additive (include/afx_batch.h: cwtObj_cwtBatchDevice): x is a HIP torch.float32
        tensor (chunks, 2**radix2_exp) with contiguous rows.  Returns (real, imag) torch
        tensors (chunks, num, 2**radix2_exp) in the LIBRARY's row order (row 0 = highest
        frequency, as cwtObj_cwt writes them); asynchronous on `stream`."""
        import torch
        n = 1 << self.radix2_exp
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
        assert x.shape[1] == n, f"chunks must be 2**radix2_exp = {n} samples"
        c = x.shape[0]
        if out_real is None:
            out_real = torch.empty((c, self.num, n), dtype=torch.float32, device=x.device)
        if out_imag is None:
            out_imag = torch.empty_like(out_real)
        s = stream if stream is not None else torch.cuda.current_stream(x.device)
        name = "cwtObj_cwtDetBatchDevice" if det else "cwtObj_cwtBatchDevice"
        fn = getattr(self._lib, name)
        fn.restype = c_int
        fn.argtypes = [c_void_p, c_void_p, c_int, c_longlong, c_void_p, c_void_p, c_void_p]
        _lib.check(fn(self._obj, x.data_ptr(), c, x.stride(0), out_real.data_ptr(),
                      out_imag.data_ptr(), s.cuda_stream), name)
        return out_real, out_imag

    def __del__(self):
        if getattr(self, "_obj", None):
            fn = self._lib.cwtObj_free
            fn.argtypes = [c_void_p]
            fn.restype = None
            fn(self._obj)
            self._obj = c_void_p(None)
