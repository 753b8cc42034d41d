"""BFT -- ctypes mirror of the reference wrapper class (python/audioflux/bft.py:
142-420) over libaudioflux_mi355x.so: same constructor arguments, same method
names, same (fre, time) result orientation.  Extra: batched / device-resident
calls that keep clips and features in HBM."""
import ctypes
from ctypes import POINTER, c_float, c_int, c_longlong, c_void_p

import numpy as np

from . import _lib, _util
from .types import (SpectralDataType, SpectralFilterBankNormalType, SpectralFilterBankScaleType,
                    SpectralFilterBankStyleType, WindowType)


class BFT:
    def __init__(self, num, radix2_exp=12, samplate=32000, low_fre=None, high_fre=None,
                 bin_per_octave=12, window_type=WindowType.HANN, slide_length=None,
                 scale_type=SpectralFilterBankScaleType.LINEAR,
                 style_type=SpectralFilterBankStyleType.SLANEY,
                 normal_type=SpectralFilterBankNormalType.NONE,
                 data_type=SpectralDataType.MAG, is_reassign=False, is_temporal=False):
        self._lib = _lib.get_lib()
        self._obj = c_void_p(None)
        self.fft_length = 1 << radix2_exp
        if num > self.fft_length // 2 + 1:
            raise ValueError(f"num={num} is too large")
        octave_like = scale_type in (SpectralFilterBankScaleType.OCTAVE,
                                     SpectralFilterBankScaleType.LOG)
        if low_fre is None:
            low_fre = 32.703 if octave_like else 0.0  # note C1, as the reference wrapper
        if high_fre is None:
            high_fre = samplate / 2
        if octave_like and low_fre < 32.703:
            raise ValueError(f"{scale_type.name} low_fre={low_fre} must be >= 32.703")
        if low_fre < 0:
            raise ValueError(f"low_fre={low_fre} must be non-negative")
        if slide_length is None:
            slide_length = self.fft_length // 4
        self.num, self.radix2_exp, self.samplate = num, radix2_exp, samplate
        self.low_fre, self.high_fre, self.bin_per_octave = low_fre, high_fre, bin_per_octave
        self.window_type, self.slide_length = window_type, slide_length
        self.scale_type, self.style_type, self.normal_type = scale_type, style_type, normal_type
        self.data_type, self.is_reassign, self.is_temporal = data_type, is_reassign, is_temporal
        self.result_type = 0
        self._temporal = None

        fn = self._lib.bftObj_new
        fn.restype = c_int
        fn.argtypes = [POINTER(c_void_p), c_int, c_int, POINTER(c_int), POINTER(c_float),
                       POINTER(c_float), POINTER(c_int), POINTER(c_int), POINTER(c_int),
                       POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int),
                       POINTER(c_int), POINTER(c_int)]
        st = fn(ctypes.byref(self._obj), num, radix2_exp, _util.opt_int(samplate),
                _util.opt_float(low_fre), _util.opt_float(high_fre), _util.opt_int(bin_per_octave),
                _util.opt_int(int(window_type)), _util.opt_int(slide_length),
                _util.opt_int(int(scale_type)), _util.opt_int(int(style_type)),
                _util.opt_int(int(normal_type)), _util.opt_int(int(data_type)),
                _util.opt_int(int(is_reassign)), _util.opt_int(int(is_temporal)))
        if st != 0 or not self._obj:
            self._obj = c_void_p(None)
            raise RuntimeError(f"bftObj_new failed with status {st}: {_lib.last_error()}")

    # -- plan queries -------------------------------------------------------
    def cal_time_length(self, data_length):
        fn = self._lib.bftObj_calTimeLength
        fn.argtypes = [c_void_p, c_int]
        return int(fn(self._obj, int(data_length)))

    def get_fre_band_arr(self):
        fn = self._lib.bftObj_getFreBandArr
        fn.argtypes, fn.restype = [c_void_p], POINTER(c_float)
        return np.ctypeslib.as_array(fn(self._obj), (self.num,)).copy()

    def get_bin_band_arr(self):
        fn = self._lib.bftObj_getBinBandArr
        fn.argtypes, fn.restype = [c_void_p], POINTER(c_int)
        return np.ctypeslib.as_array(fn(self._obj), (self.num,)).copy()

    def set_result_type(self, result_type):
        if result_type not in (0, 1):
            raise ValueError("`result_type` param error")
        fn = self._lib.bftObj_setResultType
        fn.argtypes = [c_void_p, c_int]
        fn(self._obj, int(result_type))
        self.result_type = result_type

    def set_data_norm_value(self, norm_value):
        fn = self._lib.bftObj_setDataNormValue
        fn.argtypes = [c_void_p, c_float]
        fn(self._obj, float(norm_value))

    def fused_plan_kind(self):
        """Additive, diagnostic (include/afx_batch.h: bftObj_fusedPlanKind): 0 size-generic kernels,
        1 fused n_fft-2048 kernel, 2 the same with rows cut into segments, 101 / 201 the fused
        n_fft 1024 / 4096 kernels."""
        fn = self._lib.bftObj_fusedPlanKind
        fn.argtypes = [c_void_p]
        fn.restype = c_int
        return int(fn(self._obj))

    # -- transforms ---------------------------------------------------------
    def bft(self, data_arr, result_type=0):
        """data_arr (..., n) -> (..., num, time); complex64 if result_type == 0."""
        x = _util.as_f32(data_arr)
        n = x.shape[-1]
        if n < self.fft_length:
            raise ValueError(f"fft_length={self.fft_length} is too large for data length {n}")
        if result_type != self.result_type:
            self.set_result_type(result_type)
        clips, lead = _util.flatten_leading(x, 1)
        t = self.cal_time_length(n)
        re = np.zeros((clips.shape[0], t, self.num), np.float32)
        im = np.zeros((clips.shape[0], t, self.num), np.float32)
        fn = self._lib.bftObj_bft  # the reference entry point, one clip per call
        fn = _lib.checked(fn)
        fn.restype = None
        fn.argtypes = [c_void_p, _util.c_float_p, c_int, _util.c_float_p, _util.c_float_p]
        temporal = []
        for i in range(clips.shape[0]):
            fn(self._obj, _util.fptr(clips[i]), n, _util.fptr(re[i]), _util.fptr(im[i]))
            if self.is_temporal:
                temporal.append(self._fetch_temporal(t))
        self._temporal = None
        if self.is_temporal:
            self._temporal = tuple(_util.restore_leading(np.stack([tt[k] for tt in temporal]), lead)
                                   for k in range(3))
        out = (re + 1j * im).astype(np.complex64) if self.result_type == 0 else re
        out = _util.restore_leading(out, lead)
        return np.ascontiguousarray(np.swapaxes(out, -1, -2))

    def _fetch_temporal(self, t):
        fn = self._lib.bftObj_getTemporalData
        fn = _lib.checked(fn)
        fn.restype = None
        fn.argtypes = [c_void_p, POINTER(_util.c_float_p)] * 1 + [POINTER(_util.c_float_p)] * 2
        e, r, z = _util.c_float_p(), _util.c_float_p(), _util.c_float_p()
        fn(self._obj, ctypes.byref(e), ctypes.byref(r), ctypes.byref(z))
        return tuple(np.ctypeslib.as_array(p, (t,)).copy() for p in (e, r, z))

    def get_temporal_data(self):
        if not self.is_temporal:
            raise ValueError("is_temporal=False")
        if self._temporal is None:
            raise ValueError("call bft first")
        return self._temporal

    def bft_batch(self, data_arr, result_type=1, out=None):
        """Additive: (clips, n) host array -> (clips, time, num) in ONE library call.  `out` (real results only): a
        float32 C-contiguous array of that shape to write into -- a caller that loops keeps its pages mapped."""
        x = _util.as_f32(data_arr)
        if x.ndim != 2:
            raise ValueError("bft_batch expects (clips, n)")
        if result_type != self.result_type:
            self.set_result_type(result_type)
        b, n = x.shape
        t = self.cal_time_length(n)
        if out is not None and result_type != 1:
            raise ValueError("out= is for real results (result_type=1); complex results are returned as a new complex64 array")
        if out is not None:
            if out.dtype != np.float32 or out.shape != (b, t, self.num) or not out.flags.c_contiguous:
                raise ValueError(f"out must be a C-contiguous float32 array of shape {(b, t, self.num)}")
            re = out
        else:
            re = np.zeros((b, t, self.num), np.float32)
        im = np.zeros((b, t, self.num), np.float32) if result_type == 0 else None
        fn = self._lib.bftObj_bftBatch
        fn.restype = c_int
        fn.argtypes = [c_void_p, _util.c_float_p, c_int, c_int, _util.c_float_p, _util.c_float_p]
        _lib.check(fn(self._obj, _util.fptr(x), b, n, _util.fptr(re),
                      _util.fptr(im) if im is not None else None), "bftObj_bftBatch")
        return (re + 1j * im).astype(np.complex64) if result_type == 0 else re

    def bft_device(self, x, out_real=None, out_imag=None, stream=None):
        """Additive: x is a CUDA/HIP torch.float32 tensor (clips, n), contiguous rows.
        Returns torch tensor(s) (clips, time, num) on the same device; asynchronous
        on `stream` (a torch stream) or on torch's current stream."""
        import torch
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
        b, n = x.shape
        t = self.cal_time_length(n)
        if out_real is None:
            out_real = torch.empty((b, t, self.num), dtype=torch.float32, device=x.device)
        if self.result_type == 0 and out_imag is None:
            out_imag = torch.empty_like(out_real)
        s = stream if stream is not None else torch.cuda.current_stream(x.device)
        fn = self._lib.bftObj_bftBatchDevice
        fn.restype = c_int
        fn.argtypes = [c_void_p, c_void_p, c_int, c_int, c_longlong, c_void_p, c_void_p, c_void_p]
        _lib.check(fn(self._obj, x.data_ptr(), b, n, x.stride(0), out_real.data_ptr(),
                      out_imag.data_ptr() if out_imag is not None else None, s.cuda_stream),
                   "bftObj_bftBatchDevice")
        return out_real if self.result_type == 1 else (out_real, out_imag)

    def __del__(self):
        if getattr(self, "_obj", None):
            fn = self._lib.bftObj_free
            fn.argtypes = [c_void_p]
            fn.restype = None
            fn(self._obj)
            self._obj = c_void_p(None)
