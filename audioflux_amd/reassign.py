"""Reassign -- ctypes mirror of python/audioflux/reassign.py:8-282 over libaudioflux_mi355x.so:
time-frequency reassignment of the STFT.  `reassign` returns (reassigned, stft), both
(..., fft_length // 2 + 1, time); complex64 unless result_type == 1 (amplitudes)."""
import ctypes
from ctypes import POINTER, c_float, c_int, c_longlong, c_void_p

import numpy as np

from . import _lib, _util
from .types import ReassignType, WindowType


class Reassign:
    def __init__(self, radix2_exp=12, samplate=32000, window_type=WindowType.HANN, slide_length=None,
                 re_type=ReassignType.ALL, thresh=0.001, is_padding=False):
        self._lib = _lib.get_lib()
        self._obj = c_void_p(None)
        self.fft_length = 1 << radix2_exp
        if slide_length is None:
            slide_length = self.fft_length // 4
        self.radix2_exp, self.samplate, self.window_type = radix2_exp, samplate, window_type
        self.slide_length, self.re_type, self.thresh, self.is_padding = slide_length, re_type, thresh, is_padding
        self.result_type, self.order = 0, 1
        fn = self._lib.reassignObj_new
        fn.restype = c_int
        fn.argtypes = [POINTER(c_void_p), c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int),
                       POINTER(c_float), POINTER(c_int), POINTER(c_int)]
        st = fn(ctypes.byref(self._obj), radix2_exp, _util.opt_int(samplate), _util.opt_int(int(window_type)),
                _util.opt_int(slide_length), _util.opt_int(int(re_type)), _util.opt_float(thresh),
                _util.opt_int(int(is_padding)), _util.opt_int(0))
        if st != 0 or not self._obj:
            self._obj = c_void_p(None)
            raise RuntimeError(f"reassignObj_new failed with status {st}: {_lib.last_error()}")

    def cal_time_length(self, data_length):
        fn = self._lib.reassignObj_calTimeLength
        fn.restype, fn.argtypes = c_int, [c_void_p, c_int]
        return int(fn(self._obj, int(data_length)))

    def set_result_type(self, result_type):
        fn = self._lib.reassignObj_setResultType
        fn.restype, fn.argtypes = None, [c_void_p, c_int]
        fn(self._obj, int(result_type))
        self.result_type = result_type

    def set_order(self, order):
        if order < 1:
            raise ValueError("order must be >= 1")
        fn = self._lib.reassignObj_setOrder
        fn.restype, fn.argtypes = None, [c_void_p, c_int]
        fn(self._obj, int(order))
        self.order = order

    def reassign_raw(self, x):
        """one clip -> four float32 [T, F] planes exactly as the C entry fills them"""
        x = _util.as_f32(x)
        t, f = self.cal_time_length(x.shape[0]), self.fft_length // 2 + 1
        a = [np.zeros((t, f), np.float32) for _ in range(4)]
        fn = self._lib.reassignObj_reassign
        fn = _lib.checked(fn)
        fn.restype = None
        fn.argtypes = [c_void_p, _util.c_float_p, c_int] + [_util.c_float_p] * 4
        fn(self._obj, _util.fptr(x), x.shape[0], *[_util.fptr(v) for v in a])
        return a

    def reassign(self, data_arr, result_type=0):
        x = _util.as_f32(data_arr)
        if x.shape[-1] < self.fft_length and not self.is_padding:
            raise ValueError(f"fft_length={self.fft_length} is too large for data length {x.shape[-1]}")
        if result_type != self.result_type:
            self.set_result_type(result_type)
        clips, lead = _util.flatten_leading(x, 1)
        outs = [self.reassign_raw(c) for c in clips]
        m1 = np.stack([o[0] if result_type == 1 and self.re_type != ReassignType.NONE else o[0] + 1j * o[1] for o in outs])
        m2 = np.stack([o[2] + 1j * o[3] for o in outs])
        if np.iscomplexobj(m1):
            m1 = m1.astype(np.complex64)
        m1, m2 = _util.restore_leading(m1, lead), _util.restore_leading(m2.astype(np.complex64), lead)
        return np.ascontiguousarray(np.swapaxes(m1, -1, -2)), np.ascontiguousarray(np.swapaxes(m2, -1, -2))

    def reassign_device(self, x, with_stft=False, stream=None):
        """Additive: x HIP torch.float32 (clips, n) -> reassigned (re, im) [+ stft (re, im)] torch tensors
        (clips, time, F); im is None in amplitude result mode"""
        import torch
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
        b, n = x.shape
        t, f = self.cal_time_length(n), self.fft_length // 2 + 1
        re = torch.zeros((b, t, f), dtype=torch.float32, device=x.device)
        cplx = self.result_type == 0 or self.re_type == ReassignType.NONE
        im = torch.zeros_like(re) if cplx else None
        sre = torch.empty_like(re) if with_stft else None
        sim = torch.empty_like(re) if with_stft else None
        s = stream if stream is not None else torch.cuda.current_stream(x.device)
        fn = self._lib.reassignObj_reassignBatchDevice
        fn.restype = c_int
        fn.argtypes = [c_void_p, c_void_p, c_int, c_int, c_longlong] + [c_void_p] * 5
        _lib.check(fn(self._obj, x.data_ptr(), b, n, x.stride(0), re.data_ptr(), im.data_ptr() if cplx else None,
                      sre.data_ptr() if with_stft else None, sim.data_ptr() if with_stft else None, s.cuda_stream),
                   "reassignObj_reassignBatchDevice")
        return (re, im, sre, sim) if with_stft else (re, im)

    def __del__(self):
        if getattr(self, "_obj", None):
            fn = self._lib.reassignObj_free
            fn.argtypes, fn.restype = [c_void_p], None
            fn(self._obj)
            self._obj = c_void_p(None)
