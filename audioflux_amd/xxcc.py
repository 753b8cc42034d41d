"""XXCC -- ctypes mirror of python/audioflux/feature/xxcc.py:60-230 over
libaudioflux_mi355x.so (MFCC / BFCC / GTCC / CQCC from any spectrogram)."""
import ctypes
from ctypes import POINTER, c_int, c_longlong, c_void_p

import numpy as np

from . import _lib, _util
from .types import CepstralEnergyType, CepstralRectifyType


class XXCC:
    def __init__(self, num):
        self._lib = _lib.get_lib()
        self._obj = c_void_p(None)
        self.num = num
        self.time_length = 0
        fn = self._lib.xxccObj_new
        fn.restype = c_int
        fn.argtypes = [POINTER(c_void_p), c_int]
        st = fn(ctypes.byref(self._obj), int(num))
        if st != 0 or not self._obj:
            self._obj = c_void_p(None)
            raise RuntimeError(f"xxccObj_new failed with status {st}: {_lib.last_error()}")

    def set_time_length(self, time_length):
        fn = self._lib.xxccObj_setTimeLength
        fn.argtypes = [c_void_p, c_int]
        fn = _lib.checked(fn)
        fn.restype = None
        fn(self._obj, int(time_length))
        self.time_length = time_length

    def xxcc(self, m_data_arr, cc_num=13, rectify_type=CepstralRectifyType.LOG):
        """(..., fre, time) spectrogram -> (..., cc_num, time)"""
        if cc_num > self.num:
            raise ValueError(f"cc_num={cc_num} must be less than num={self.num}")
        m = np.asarray(m_data_arr)
        if np.iscomplexobj(m):
            m = np.abs(m)
        m = _util.as_f32(np.swapaxes(m, -1, -2))  # (..., time, fre)
        frames, lead = _util.flatten_leading(m, 2)
        t = m.shape[-2]
        out = np.zeros((frames.shape[0], t, cc_num), np.float32)
        fn = self._lib.xxccObj_xxcc
        fn = _lib.checked(fn)
        fn.restype = None
        fn.argtypes = [c_void_p, _util.c_float_p, c_int, POINTER(c_int), _util.c_float_p]
        self.set_time_length(t)
        for i in range(frames.shape[0]):
            fn(self._obj, _util.fptr(frames[i]), cc_num, _util.opt_int(int(rectify_type)),
               _util.fptr(out[i]))
        out = _util.restore_leading(out, lead)
        return np.ascontiguousarray(np.swapaxes(out, -1, -2))

    def xxcc_standard(self, m_data_arr, energy_arr, cc_num=13, delta_window_length=9,
                      energy_type=CepstralEnergyType.REPLACE,
                      rectify_type=CepstralRectifyType.LOG):
        """returns (coe, delta, delta_delta), each (..., cc_num(+1), time)"""
        if cc_num > self.num:
            raise ValueError(f"cc_num={cc_num} must be less than num={self.num}")
        m = np.asarray(m_data_arr)
        if np.iscomplexobj(m):
            m = np.abs(m)
        m = _util.as_f32(np.swapaxes(m, -1, -2))
        frames, lead = _util.flatten_leading(m, 2)
        t = m.shape[-2]
        e = _util.as_f32(energy_arr).reshape(frames.shape[0], t)
        n_out = cc_num + (1 if energy_type == CepstralEnergyType.APPEND else 0)
        outs = [np.zeros((frames.shape[0], t, n_out), np.float32) for _ in range(3)]
        fn = self._lib.xxccObj_xxccStandard
        fn = _lib.checked(fn)
        fn.restype = None
        fn.argtypes = [c_void_p, _util.c_float_p, c_int, _util.c_float_p, POINTER(c_int),
                       POINTER(c_int), POINTER(c_int), _util.c_float_p, _util.c_float_p,
                       _util.c_float_p]
        self.set_time_length(t)
        for i in range(frames.shape[0]):
            fn(self._obj, _util.fptr(frames[i]), cc_num, _util.fptr(e[i]),
               _util.opt_int(delta_window_length), _util.opt_int(int(energy_type)),
               _util.opt_int(int(rectify_type)), _util.fptr(outs[0][i]), _util.fptr(outs[1][i]),
               _util.fptr(outs[2][i]))
        return tuple(np.ascontiguousarray(np.swapaxes(_util.restore_leading(o, lead), -1, -2))
                     for o in outs)

    def xxcc_device(self, m, cc_num=13, rectify_type=CepstralRectifyType.LOG, out=None, stream=None):
        """Additive: m is a CUDA/HIP torch.float32 tensor (..., num) of frames; returns (..., cc_num)."""
        import torch
        assert m.is_cuda and m.dtype == torch.float32 and m.is_contiguous() and m.shape[-1] == self.num
        rows = m.numel() // self.num
        if out is None:
            out = torch.empty(m.shape[:-1] + (cc_num,), dtype=torch.float32, device=m.device)
        s = stream if stream is not None else torch.cuda.current_stream(m.device)
        fn = self._lib.xxccObj_xxccDevice
        fn.restype = c_int
        fn.argtypes = [c_void_p, c_void_p, c_longlong, c_int, POINTER(c_int), c_void_p, c_void_p]
        _lib.check(fn(self._obj, m.data_ptr(), rows, cc_num, _util.opt_int(int(rectify_type)),
                      out.data_ptr(), s.cuda_stream), "xxccObj_xxccDevice")
        return out

    def __del__(self):
        if getattr(self, "_obj", None):
            fn = self._lib.xxccObj_free
            fn.argtypes = [c_void_p]
            fn.restype = None
            fn(self._obj)
            self._obj = c_void_p(None)
