"""STFT -- ctypes mirror of the reference wrapper class (python/audioflux/stft.py:14-407) over
libaudioflux_mi355x.so: same constructor arguments, method names and result orientation
((..., fft_length // 2 + 1, time) complex64 from `stft`, (..., n) float32 from `istft`).
Extra: batched device-resident calls (`stft_device`, `istft_device`)."""
import ctypes
from ctypes import POINTER, c_float, c_int, c_longlong, c_void_p

import numpy as np

from . import _lib, _util
from .types import PaddingModeType, PaddingPositionType, WindowType


class STFT:
    def __init__(self, radix2_exp=12, window_type=WindowType.RECT, slide_length=1024, is_continue=False):
        self._lib = _lib.get_lib()
        self._obj = c_void_p(None)
        self.radix2_exp, self.window_type, self.slide_length = radix2_exp, window_type, slide_length
        self.is_continue = bool(is_continue)
        self.is_pad = False
        self.position_type, self.mode_type = PaddingPositionType.CENTER, PaddingModeType.CONSTANT
        self.pad_value1 = self.pad_value2 = 0.0
        self.fft_length = 1 << radix2_exp
        fn = self._lib.stftObj_new
        fn.restype = c_int
        fn.argtypes = [POINTER(c_void_p), c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int)]
        st = fn(ctypes.byref(self._obj), radix2_exp, _util.opt_int(int(window_type)),
                _util.opt_int(slide_length), _util.opt_int(int(self.is_continue)))
        if st != 0 or not self._obj:
            self._obj = c_void_p(None)
            raise RuntimeError(f"stftObj_new failed with status {st}: {_lib.last_error()}")

    # -- switches -----------------------------------------------------------
    def set_slide_length(self, slide_length):
        fn = self._lib.stftObj_setSlideLength
        fn.restype, fn.argtypes = None, [c_void_p, c_int]
        fn(self._obj, int(slide_length))
        if slide_length > 0:
            self.slide_length = slide_length

    def enable_padding(self, flag=False):
        fn = self._lib.stftObj_enablePadding
        fn.restype, fn.argtypes = None, [c_void_p, c_int]
        fn(self._obj, int(flag))
        self.is_pad = bool(flag)

    def enable_continue(self, flag=False):
        fn = self._lib.stftObj_enableContinue
        fn.restype, fn.argtypes = None, [c_void_p, c_int]
        fn(self._obj, int(flag))
        self.is_continue = bool(flag)

    def set_padding(self, position_type=PaddingPositionType.CENTER, mode_type=PaddingModeType.CONSTANT,
                    value1=0.0, value2=0.0):
        """only honoured after enable_padding(True), as in the reference"""
        fn = self._lib.stftObj_setPadding
        fn = _lib.checked(fn)
        fn.restype = None
        fn.argtypes = [c_void_p, POINTER(c_int), POINTER(c_int), POINTER(c_float), POINTER(c_float)]
        fn(self._obj, _util.opt_int(int(position_type)), _util.opt_int(int(mode_type)),
           _util.opt_float(value1), _util.opt_float(value2))
        if self.is_pad:
            self.position_type, self.mode_type = position_type, mode_type
            self.pad_value1, self.pad_value2 = value1, value2

    def use_window_data_arr(self, data_arr):
        w = _util.as_f32(data_arr)
        if w.ndim != 1:
            raise ValueError("data_arr.ndim must be 1")
        if w.shape[-1] != self.fft_length:
            raise ValueError(f"data_arr length[{w.shape[-1]}] must be {self.fft_length}")
        fn = self._lib.stftObj_useWindowDataArr
        fn.restype, fn.argtypes = None, [c_void_p, _util.c_float_p]
        fn(self._obj, _util.fptr(w))

    def get_window_data_arr(self):
        fn = self._lib.stftObj_getWindowDataArr
        fn.argtypes, fn.restype = [c_void_p], POINTER(c_float)
        return np.ctypeslib.as_array(fn(self._obj), (self.fft_length,)).copy()

    def cal_time_length(self, data_length):
        fn = self._lib.stftObj_calTimeLength
        fn.restype, fn.argtypes = c_int, [c_void_p, c_int]
        return int(fn(self._obj, int(data_length)))

    def cal_data_length(self, time_length):
        fn = self._lib.stftObj_calDataLength
        fn.restype, fn.argtypes = c_int, [c_void_p, c_int]
        return int(fn(self._obj, int(time_length)))

    # -- transforms ---------------------------------------------------------
    def stft_full(self, data_arr):
        """one clip (n,) -> (time, fft_length) re, im exactly as stftObj_stft stores them"""
        x = _util.as_f32(data_arr)
        t = self.cal_time_length(x.shape[-1])
        re = np.zeros((t, self.fft_length), np.float32)
        im = np.zeros((t, self.fft_length), np.float32)
        fn = self._lib.stftObj_stft
        fn = _lib.checked(fn)
        fn.restype = None
        fn.argtypes = [c_void_p, _util.c_float_p, c_int, _util.c_float_p, _util.c_float_p]
        fn(self._obj, _util.fptr(x), x.shape[-1], _util.fptr(re), _util.fptr(im))
        return re, im

    def stft(self, data_arr):
        """data_arr (..., n) -> (..., fft_length // 2 + 1, time) complex64"""
        x = _util.as_f32(data_arr)
        clips, lead = _util.flatten_leading(x, 1)
        outs = []
        for i in range(clips.shape[0]):
            re, im = self.stft_full(clips[i])
            outs.append((re + 1j * im).astype(np.complex64))
        out = _util.restore_leading(np.stack(outs), lead)
        out = np.ascontiguousarray(np.swapaxes(out, -1, -2))
        return out[..., : self.fft_length // 2 + 1, :]

    def istft(self, m_data_arr, method_type=0):
        """m_data_arr (..., fft_length // 2 + 1, time) complex -> (..., n) float32"""
        m = np.asarray(m_data_arr)
        if not np.iscomplexobj(m):
            raise ValueError("m_data_arr must be of type np.complex")
        if m.ndim < 2:
            raise ValueError("m_data_arr's dimensions must be greater than 1")
        mirror = np.copy(m)[..., ::-1, :][..., 1:-1, :]
        mirror.imag *= -1
        full = np.concatenate([m, mirror], axis=-2)  # (..., fft_length, time)
        t = full.shape[-1]
        full = np.ascontiguousarray(np.swapaxes(full, -1, -2))
        clips, lead = _util.flatten_leading(full, 2)
        n = self.cal_data_length(t)
        out = np.zeros((clips.shape[0], n), np.float32)
        fn = self._lib.stftObj_istft
        fn = _lib.checked(fn)
        fn.restype = None
        fn.argtypes = [c_void_p, _util.c_float_p, _util.c_float_p, c_int, c_int, _util.c_float_p]
        for i in range(clips.shape[0]):
            re = np.ascontiguousarray(clips[i].real, np.float32)
            im = np.ascontiguousarray(clips[i].imag, np.float32)
            fn(self._obj, _util.fptr(re), _util.fptr(im), t, int(method_type), _util.fptr(out[i]))
        return _util.restore_leading(out, lead)

    # -- additive: device-resident batches ----------------------------------
    def stft_device(self, x, stream=None):
        """x: CUDA/HIP torch.float32 (clips, n) -> (re, im) torch (clips, time, fft_length);
        asynchronous on `stream` or torch's current stream.  The streaming tail is not used."""
        import torch
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
        b, n = x.shape
        keep = self.is_continue
        if keep:
            self.enable_continue(False)
        t = self.cal_time_length(n)
        if keep:
            self.enable_continue(True)
        re = torch.empty((b, t, self.fft_length), dtype=torch.float32, device=x.device)
        im = torch.empty_like(re)
        s = stream if stream is not None else torch.cuda.current_stream(x.device)
        fn = self._lib.stftObj_stftBatchDevice
        fn.restype = c_int
        fn.argtypes = [c_void_p, c_void_p, c_int, c_int, c_longlong, c_void_p, c_void_p, c_void_p]
        _lib.check(fn(self._obj, x.data_ptr(), b, n, x.stride(0), re.data_ptr(), im.data_ptr(),
                      s.cuda_stream), "stftObj_stftBatchDevice")
        return re, im

    def istft_device(self, re, im, method_type=0, stream=None):
        """re, im: torch (clips, time, fft_length) -> torch (clips, n)"""
        import torch
        assert re.is_cuda and re.is_contiguous() and im.is_contiguous() and re.shape == im.shape and re.dim() == 3
        b, t, _ = re.shape
        n = self.cal_data_length(t)
        out = torch.zeros((b, n), dtype=torch.float32, device=re.device)
        s = stream if stream is not None else torch.cuda.current_stream(re.device)
        fn = self._lib.stftObj_istftBatchDevice
        fn.restype = c_int
        fn.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_longlong, c_void_p]
        _lib.check(fn(self._obj, re.data_ptr(), im.data_ptr(), b, t, int(method_type), out.data_ptr(), n,
                      s.cuda_stream), "stftObj_istftBatchDevice")
        return out

    def y_coords(self, samplate=32000):
        return np.linspace(0, samplate / 2, self.fft_length // 2 + 1)

    def __del__(self):
        if getattr(self, "_obj", None):
            fn = self._lib.stftObj_free
            fn.argtypes, fn.restype = [c_void_p], None
            fn(self._obj)
            self._obj = c_void_p(None)
