"""Synsq -- ctypes mirror of python/audioflux/synsq.py:9-192 over libaudioflux_mi355x.so:
synchrosqueezing of a given complex time-frequency matrix (num, time) on the band axis `fre_arr`."""
import ctypes
from ctypes import POINTER, c_float, c_int, c_void_p

import numpy as np

from . import _lib, _util
from .types import SpectralFilterBankScaleType


class Synsq:
    def __init__(self, num, radix2_exp=12, samplate=32000, order=1, thresh=0.001):
        self._lib = _lib.get_lib()
        self._obj = c_void_p(None)
        self.num, self.radix2_exp, self.samplate, self.order, self.thresh = num, radix2_exp, samplate, order, thresh
        self.fft_length = 1 << radix2_exp
        fn = self._lib.synsqObj_new
        fn.restype = c_int
        fn.argtypes = [POINTER(c_void_p), c_int, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_float)]
        st = fn(ctypes.byref(self._obj), num, radix2_exp, _util.opt_int(samplate), _util.opt_int(order),
                _util.opt_float(thresh))
        if st != 0 or not self._obj:
            self._obj = c_void_p(None)
            raise RuntimeError(f"synsqObj_new failed with status {st}: {_lib.last_error()}")

    def synsq(self, m_data_arr, filter_bank_type=SpectralFilterBankScaleType.OCTAVE, fre_arr=None):
        """m_data_arr complex (num, time) in the row order of `fre_arr` -> squeezed complex64 (num, time)"""
        m = np.asarray(m_data_arr)
        if not np.iscomplexobj(m) or m.shape != (self.num, self.fft_length):
            raise ValueError(f"m_data_arr must be complex with shape ({self.num}, {self.fft_length})")
        fre = _util.as_f32(fre_arr)
        if fre.shape != (self.num,):
            raise ValueError("fre_arr must hold num band centres")
        re, im = np.ascontiguousarray(m.real, np.float32), np.ascontiguousarray(m.imag, np.float32)
        a, b = np.zeros_like(re), np.zeros_like(re)
        fn = self._lib.synsqObj_synsq
        fn = _lib.checked(fn)
        fn.restype = None
        fn.argtypes = [c_void_p, _util.c_float_p, c_int] + [_util.c_float_p] * 4
        fn(self._obj, _util.fptr(fre), int(filter_bank_type), _util.fptr(re), _util.fptr(im), _util.fptr(a),
           _util.fptr(b))
        return (a + 1j * b).astype(np.complex64)

    def __del__(self):
        if getattr(self, "_obj", None):
            fn = self._lib.synsqObj_free
            fn.argtypes, fn.restype = [c_void_p], None
            fn(self._obj)
            self._obj = c_void_p(None)
