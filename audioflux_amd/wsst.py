"""WSST -- ctypes mirror of python/audioflux/wsst.py:14-348 over libaudioflux_mi355x.so: wavelet
synchrosqueezed transform.  `wsst` returns (squeezed, cwt) complex64, as the reference wrapper:
the squeezed matrix in the C row order, the CWT flipped to ascending frequency."""
import ctypes
from ctypes import POINTER, c_float, c_int, c_longlong, c_void_p

import numpy as np

from . import _lib, _util
from .types import SpectralFilterBankScaleType, WaveletContinueType


class WSST:
    def __init__(self, num=84, radix2_exp=12, samplate=32000, low_fre=None, high_fre=None, bin_per_octave=12,
                 wavelet_type=WaveletContinueType.MORLET, scale_type=SpectralFilterBankScaleType.OCTAVE,
                 gamma=None, beta=None, thresh=0.001, is_padding=True):
        self._lib = _lib.get_lib()
        self._obj = c_void_p(None)
        self.fft_length = 1 << radix2_exp
        if num > self.fft_length // 2 + 1:
            raise ValueError(f"num={num} is too large")
        octave_like = scale_type in (SpectralFilterBankScaleType.OCTAVE, SpectralFilterBankScaleType.LOG)
        if low_fre is None:
            low_fre = 32.703195662574764 if octave_like else 0.0
        if high_fre is None:
            high_fre = samplate / 2
        if octave_like and low_fre < 32.703:
            raise ValueError(f"{scale_type.name} low_fre={low_fre} must be greater than or equal to 32.703")
        if thresh < 0:
            raise ValueError("thresh must be >= 0")
        self.num, self.radix2_exp, self.samplate = num, radix2_exp, samplate
        self.low_fre, self.high_fre, self.bin_per_octave = low_fre, high_fre, bin_per_octave
        self.wavelet_type, self.scale_type, self.thresh, self.is_padding = wavelet_type, scale_type, thresh, is_padding
        fn = self._lib.wsstObj_new
        fn.restype = c_int
        fn.argtypes = [POINTER(c_void_p), c_int, c_int, POINTER(c_int), POINTER(c_float), POINTER(c_float),
                       POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_float), POINTER(c_float),
                       POINTER(c_float), POINTER(c_int)]
        st = fn(ctypes.byref(self._obj), num, radix2_exp, _util.opt_int(samplate), _util.opt_float(low_fre),
                _util.opt_float(high_fre), _util.opt_int(bin_per_octave), _util.opt_int(int(wavelet_type)),
                _util.opt_int(int(scale_type)), _util.opt_float(gamma), _util.opt_float(beta),
                _util.opt_float(thresh), _util.opt_int(int(is_padding)))
        if st != 0 or not self._obj:
            self._obj = c_void_p(None)
            raise RuntimeError(f"wsstObj_new failed with status {st}: {_lib.last_error()}")

    def get_fre_band_arr(self):
        fn = self._lib.wsstObj_getFreBandArr
        fn.argtypes, fn.restype = [c_void_p], POINTER(c_float)
        return np.ctypeslib.as_array(fn(self._obj), (self.num,)).copy()

    def get_bin_band_arr(self):
        fn = self._lib.wsstObj_getBinBandArr
        fn.argtypes, fn.restype = [c_void_p], POINTER(c_int)
        return np.ctypeslib.as_array(fn(self._obj), (self.num,)).copy()

    def set_order(self, order):
        fn = self._lib.wsstObj_setOrder
        fn.argtypes, fn.restype = [c_void_p, c_int], None
        fn(self._obj, int(order))

    def wsst_raw(self, data_arr):
        """one chunk (2**radix2_exp,) -> (squeezed, cwt) complex [num, n], both in the C row order"""
        x = _util.as_f32(data_arr)
        assert x.shape == (self.fft_length,)
        a = [np.zeros((self.num, self.fft_length), np.float32) for _ in range(4)]
        fn = self._lib.wsstObj_wsst
        fn = _lib.checked(fn)
        fn.restype = None
        fn.argtypes = [c_void_p] + [_util.c_float_p] * 5
        fn(self._obj, _util.fptr(x), *[_util.fptr(v) for v in a])
        return (a[0] + 1j * a[1]).astype(np.complex64), (a[2] + 1j * a[3]).astype(np.complex64)

    def wsst(self, data_arr):
        x = _util.as_f32(data_arr)
        n = self.fft_length
        if x.shape[-1] >= n:
            x = np.ascontiguousarray(x[..., :n])
        else:
            pad = np.zeros(x.shape[:-1] + (n,), np.float32)
            pad[..., : x.shape[-1]] = x
            x = pad
        clips, lead = _util.flatten_leading(x, 1)
        res = [self.wsst_raw(c) for c in clips]
        s = _util.restore_leading(np.stack([r[0] for r in res]), lead)
        w = _util.restore_leading(np.stack([r[1] for r in res]), lead)
        return s, np.ascontiguousarray(w[..., ::-1, :])

    def wsst_device(self, x, with_cwt=False, stream=None):
        """Additive: x HIP torch.float32 (chunks, n) -> squeezed (re, im) [+ cwt (re, im)] torch tensors
        (chunks, num, n) in the C row order"""
        import torch
        n = self.fft_length
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1 and x.shape[1] == n
        c = x.shape[0]
        sre = torch.zeros((c, self.num, n), dtype=torch.float32, device=x.device)
        sim = torch.zeros_like(sre)
        wre = torch.empty_like(sre) if with_cwt else None
        wim = torch.empty_like(sre) if with_cwt else None
        s = stream if stream is not None else torch.cuda.current_stream(x.device)
        fn = self._lib.wsstObj_wsstBatchDevice
        fn.restype = c_int
        fn.argtypes = [c_void_p, c_void_p, c_int, c_longlong] + [c_void_p] * 5
        _lib.check(fn(self._obj, x.data_ptr(), c, x.stride(0), sre.data_ptr(), sim.data_ptr(),
                      wre.data_ptr() if with_cwt else None, wim.data_ptr() if with_cwt else None, s.cuda_stream),
                   "wsstObj_wsstBatchDevice")
        return (sre, sim, wre, wim) if with_cwt else (sre, sim)

    def __del__(self):
        if getattr(self, "_obj", None):
            fn = self._lib.wsstObj_free
            fn.argtypes, fn.restype = [c_void_p], None
            fn(self._obj)
            self._obj = c_void_p(None)
