"""Cepstrogram -- ctypes mirror of python/audioflux/cepstrogram.py:81-189 over
libaudioflux_mi355x.so."""
import ctypes
from ctypes import POINTER, c_int, c_longlong, c_void_p

import numpy as np

from . import _lib, _util
from .types import WindowType


class Cepstrogram:
    def __init__(self, radix2_exp=12, samplate=32000, window_type=WindowType.RECT, slide_length=1024):
        self._lib = _lib.get_lib()
        self._obj = c_void_p(None)
        self.radix2_exp, self.samplate = radix2_exp, samplate
        self.window_type, self.slide_length = window_type, slide_length
        self.fft_length = 1 << radix2_exp
        fn = self._lib.cepstrogramObj_new
        fn.restype = c_int
        fn.argtypes = [POINTER(c_void_p), c_int, POINTER(c_int), POINTER(c_int)]
        st = fn(ctypes.byref(self._obj), radix2_exp, _util.opt_int(int(window_type)),
                _util.opt_int(slide_length))
        if st != 0 or not self._obj:
            self._obj = c_void_p(None)
            raise RuntimeError(f"cepstrogramObj_new failed with status {st}: {_lib.last_error()}")

    def cal_time_length(self, data_length):
        fn = self._lib.cepstrogramObj_calTimeLength
        fn.argtypes = [c_void_p, c_int]
        return int(fn(self._obj, int(data_length)))

    def cepstrogram(self, data_arr, cep_num=4):
        """data_arr (..., n) -> (cepstrums, envelope, details), each (..., fft_length/2+1, time)"""
        x = _util.as_f32(data_arr)
        n = x.shape[-1]
        if n < self.fft_length:
            raise ValueError(f"fft_length={self.fft_length} is too large for data length {n}")
        clips, lead = _util.flatten_leading(x, 1)
        t = self.cal_time_length(n)
        f = self.fft_length // 2 + 1
        outs = [np.zeros((clips.shape[0], t, f), np.float32) for _ in range(3)]
        fn = self._lib.cepstrogramObj_cepstrogram
        fn = _lib.checked(fn)
        fn.restype = None
        fn.argtypes = [c_void_p, c_int, _util.c_float_p, c_int] + [_util.c_float_p] * 3
        for i in range(clips.shape[0]):
            fn(self._obj, int(cep_num), _util.fptr(clips[i]), n, _util.fptr(outs[0][i]),
               _util.fptr(outs[1][i]), _util.fptr(outs[2][i]))
        return tuple(np.ascontiguousarray(np.swapaxes(_util.restore_leading(o, lead), -1, -2))
                     for o in outs)

    def cepstrogram_device(self, x, cep_num=4, stream=None):
        """Additive (cepstrogramObj_cepstrogramBatchDevice): x HIP torch.float32 (clips, n) ->
        three torch tensors (clips, time, fft_length/2+1), time-major."""
        import torch
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
        b, n = x.shape
        t = self.cal_time_length(n)
        f = self.fft_length // 2 + 1
        outs = [torch.empty((b, t, f), dtype=torch.float32, device=x.device) for _ in range(3)]
        s = stream if stream is not None else torch.cuda.current_stream(x.device)
        fn = self._lib.cepstrogramObj_cepstrogramBatchDevice
        fn.restype = c_int
        fn.argtypes = [c_void_p, c_int, c_void_p, c_int, c_int, c_longlong, c_void_p, c_void_p,
                       c_void_p, c_void_p]
        _lib.check(fn(self._obj, int(cep_num), x.data_ptr(), b, n, x.stride(0), outs[0].data_ptr(),
                      outs[1].data_ptr(), outs[2].data_ptr(), s.cuda_stream),
                   "cepstrogramObj_cepstrogramBatchDevice")
        return tuple(outs)

    def y_coords(self):
        return np.linspace(0, self.samplate / 2, self.fft_length // 2 + 1 + 1)

    def __del__(self):
        if getattr(self, "_obj", None):
            fn = self._lib.cepstrogramObj_free
            fn.argtypes = [c_void_p]
            fn.restype = None
            fn(self._obj)
            self._obj = c_void_p(None)
