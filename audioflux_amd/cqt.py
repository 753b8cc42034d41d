"""CQT -- ctypes mirror of python/audioflux/cqt.py (CQTBase :20-222, CQT :600-660)
over libaudioflux_mi355x.so: constant-Q transform, CQT-chroma and CQCC."""
import ctypes
from ctypes import POINTER, c_float, c_int, c_longlong, c_void_p

import numpy as np

from . import _lib, _util
from .types import (CepstralRectifyType, ChromaDataNormalType, SpectralDataType,
                    SpectralFilterBankNormalType, WindowType)


class CQT:
    def __init__(self, num=84, samplate=32000, low_fre=32.703, bin_per_octave=12, factor=1.,
                 thresh=0.01, window_type=WindowType.HANN, slide_length=None, is_continue=False,
                 normal_type=SpectralFilterBankNormalType.AREA, is_scale=True, beta=0.):
        if bin_per_octave not in (12, 24, 36):
            raise ValueError(f"bin_per_octave={bin_per_octave} must be 12, 24 or 36")
        if num % bin_per_octave != 0:
            raise ValueError(f"num={num} must be an integer multiple of bin_per_octave={bin_per_octave}")
        self._lib = _lib.get_lib()
        self._obj = c_void_p(None)
        self.num, self.samplate, self.low_fre = num, samplate, low_fre
        self.bin_per_octave, self.factor, self.beta, self.thresh = bin_per_octave, factor, beta, thresh
        self.window_type, self.is_continue = window_type, is_continue
        self.normal_type, self.is_scale = normal_type, is_scale
        fn = self._lib.cqtObj_newWith
        fn.restype = c_int
        fn.argtypes = [POINTER(c_void_p), c_int, POINTER(c_int), POINTER(c_float), POINTER(c_int),
                       POINTER(c_float), POINTER(c_float), POINTER(c_float), POINTER(c_int),
                       POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int)]
        st = fn(ctypes.byref(self._obj), num, _util.opt_int(samplate), _util.opt_float(low_fre),
                _util.opt_int(bin_per_octave), _util.opt_float(factor), _util.opt_float(beta),
                _util.opt_float(thresh), _util.opt_int(int(window_type)),
                _util.opt_int(slide_length), _util.opt_int(int(is_continue)),
                _util.opt_int(int(normal_type)), _util.opt_int(int(is_scale)))
        if st != 0 or not self._obj:
            self._obj = c_void_p(None)
            raise RuntimeError(f"cqtObj_newWith failed with status {st}: {_lib.last_error()}")
        g = self._lib.cqtObj_getFFTLength
        g.argtypes = [c_void_p]
        self.fft_length = int(g(self._obj))
        self.slide_length = slide_length if slide_length else self.fft_length // 4

    def cal_time_length(self, data_length):
        fn = self._lib.cqtObj_calTimeLength
        fn.argtypes = [c_void_p, c_int]
        return int(fn(self._obj, int(data_length)))

    def get_fre_band_arr(self):
        fn = self._lib.cqtObj_getFreBandArr
        fn.argtypes, fn.restype = [c_void_p], POINTER(c_float)
        return np.ctypeslib.as_array(fn(self._obj), (self.num,)).copy()

    def set_scale(self, flag):
        fn = self._lib.cqtObj_setScale
        fn.argtypes = [c_void_p, c_int]
        fn = _lib.checked(fn)
        fn.restype = None
        fn(self._obj, int(flag))

    def cqt(self, data_arr):
        """data_arr (..., n) -> complex64 (..., num, time)"""
        x = _util.as_f32(data_arr)
        clips, lead = _util.flatten_leading(x, 1)
        n = x.shape[-1]
        t = self.cal_time_length(n)
        re = np.zeros((clips.shape[0], t, self.num), np.float32)
        im = np.zeros_like(re)
        fn = self._lib.cqtObj_cqt
        fn = _lib.checked(fn)
        fn.restype = None
        fn.argtypes = [c_void_p, _util.c_float_p, c_int, _util.c_float_p, _util.c_float_p]
        for i in range(clips.shape[0]):
            fn(self._obj, _util.fptr(clips[i]), n, _util.fptr(re[i]), _util.fptr(im[i]))
        out = _util.restore_leading((re + 1j * im).astype(np.complex64), lead)
        return np.ascontiguousarray(np.swapaxes(out, -1, -2))

    def chroma(self, m_cqt_data, chroma_num=12, data_type=SpectralDataType.POWER,
               norm_type=ChromaDataNormalType.MAX):
        """complex (..., num, time) from the LAST cqt call -> float32 (..., chroma_num, time)"""
        if not np.iscomplexobj(m_cqt_data):
            raise ValueError("m_cqt_data must be complex")
        m = np.swapaxes(np.asarray(m_cqt_data), -1, -2)  # (..., time, num)
        frames, lead = _util.flatten_leading(m, 2)
        t = m.shape[-2]
        out = np.zeros((frames.shape[0], t, chroma_num), np.float32)
        fn = self._lib.cqtObj_chroma
        fn = _lib.checked(fn)
        fn.restype = None
        fn.argtypes = [c_void_p, POINTER(c_int), POINTER(c_int), POINTER(c_int), _util.c_float_p,
                       _util.c_float_p, _util.c_float_p]
        for i in range(frames.shape[0]):
            re = _util.as_f32(frames[i].real)
            im = _util.as_f32(frames[i].imag)
            fn(self._obj, _util.opt_int(chroma_num), _util.opt_int(int(data_type)),
               _util.opt_int(int(norm_type)), _util.fptr(re), _util.fptr(im), _util.fptr(out[i]))
        return np.ascontiguousarray(np.swapaxes(_util.restore_leading(out, lead), -1, -2))

    def cqt_device(self, x, out_real=None, out_imag=None, stream=None):
        """Additive (include/afx_batch.h: cqtObj_cqtBatchDevice): x is a HIP torch.float32
        tensor (clips, n), contiguous rows -> (real, imag) torch tensors (clips, time, num),
        time-major as the library writes them; asynchronous on `stream`."""
        import torch
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
        b, n = x.shape
        t = self.cal_time_length(n)
        if out_real is None:
            out_real = torch.empty((b, t, self.num), dtype=torch.float32, device=x.device)
        if out_imag is None:
            out_imag = torch.empty_like(out_real)
        s = stream if stream is not None else torch.cuda.current_stream(x.device)
        fn = self._lib.cqtObj_cqtBatchDevice
        fn.restype = c_int
        fn.argtypes = [c_void_p, c_void_p, c_int, c_int, c_longlong, c_void_p, c_void_p, c_void_p]
        _lib.check(fn(self._obj, x.data_ptr(), b, n, x.stride(0), out_real.data_ptr(),
                      out_imag.data_ptr(), s.cuda_stream), "cqtObj_cqtBatchDevice")
        return out_real, out_imag

    def chroma_device(self, real, imag, chroma_num=12, data_type=SpectralDataType.POWER,
                      norm_type=ChromaDataNormalType.MAX, out=None, stream=None):
        """Additive (cqtObj_chromaBatchDevice): real/imag (..., time, num) HIP tensors from
        cqt_device -> (..., time, chroma_num)."""
        import torch
        assert real.is_cuda and real.is_contiguous() and imag.is_contiguous()
        assert real.shape == imag.shape and real.shape[-1] == self.num
        rows = real.numel() // self.num
        if out is None:
            out = torch.empty(real.shape[:-1] + (chroma_num,), dtype=torch.float32, device=real.device)
        s = stream if stream is not None else torch.cuda.current_stream(real.device)
        fn = self._lib.cqtObj_chromaBatchDevice
        fn.restype = c_int
        fn.argtypes = [c_void_p, POINTER(c_int), POINTER(c_int), POINTER(c_int), c_void_p, c_void_p,
                       c_longlong, c_void_p, c_void_p]
        _lib.check(fn(self._obj, _util.opt_int(chroma_num), _util.opt_int(int(data_type)),
                      _util.opt_int(int(norm_type)), real.data_ptr(), imag.data_ptr(), rows,
                      out.data_ptr(), s.cuda_stream), "cqtObj_chromaBatchDevice")
        return out

    def cqt_chroma_device(self, x, chroma_num=12, data_type=SpectralDataType.POWER, norm_type=ChromaDataNormalType.MAX,
                          out_real=None, out_imag=None, out=None, stream=None):
        """Additive (cqtObj_cqtChromaBatchDevice): cqt_device + chroma_device as one call, pass by pass over
        the clips -> (real, imag, chroma) HIP tensors; same results as the two calls."""
        import torch
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
        b, n = x.shape
        t = self.cal_time_length(n)
        if out_real is None:
            out_real = torch.empty((b, t, self.num), dtype=torch.float32, device=x.device)
        if out_imag is None:
            out_imag = torch.empty_like(out_real)
        if out is None:
            out = torch.empty((b, t, chroma_num), dtype=torch.float32, device=x.device)
        s = stream if stream is not None else torch.cuda.current_stream(x.device)
        fn = self._lib.cqtObj_cqtChromaBatchDevice
        fn.restype = c_int
        fn.argtypes = [c_void_p, c_void_p, c_int, c_int, c_longlong, c_void_p, c_void_p, POINTER(c_int),
                       POINTER(c_int), POINTER(c_int), c_void_p, c_void_p]
        _lib.check(fn(self._obj, x.data_ptr(), b, n, x.stride(0), out_real.data_ptr(), out_imag.data_ptr(),
                      _util.opt_int(chroma_num), _util.opt_int(int(data_type)), _util.opt_int(int(norm_type)),
                      out.data_ptr(), s.cuda_stream), "cqtObj_cqtChromaBatchDevice")
        return out_real, out_imag, out

    def cqcc(self, m_data_arr, cc_num=13, rectify_type=CepstralRectifyType.LOG):
        """real (..., num, time) magnitudes of the LAST cqt call -> (..., cc_num, time)"""
        m = _util.as_f32(np.swapaxes(np.abs(np.asarray(m_data_arr)), -1, -2))
        frames, lead = _util.flatten_leading(m, 2)
        t = m.shape[-2]
        out = np.zeros((frames.shape[0], t, cc_num), np.float32)
        fn = self._lib.cqtObj_cqcc
        fn = _lib.checked(fn)
        fn.restype = None
        fn.argtypes = [c_void_p, _util.c_float_p, c_int, POINTER(c_int), _util.c_float_p]
        for i in range(frames.shape[0]):
            fn(self._obj, _util.fptr(frames[i]), cc_num, _util.opt_int(int(rectify_type)),
               _util.fptr(out[i]))
        return np.ascontiguousarray(np.swapaxes(_util.restore_leading(out, lead), -1, -2))

    def cqhc(self, m_data_arr, hc_num=20):
        """(..., num, time) magnitudes/powers of the LAST cqt call (complex: power, as the
        reference wrapper, python/audioflux/cqt.py:300-301) -> (..., hc_num, time)"""
        m = np.asarray(m_data_arr)
        if np.iscomplexobj(m):
            m = np.abs(m) ** 2
        m = _util.as_f32(np.swapaxes(m, -1, -2))
        frames, lead = _util.flatten_leading(m, 2)
        t = m.shape[-2]
        out = np.zeros((frames.shape[0], t, hc_num), np.float32)
        fn = self._lib.cqtObj_cqhc
        fn = _lib.checked(fn)
        fn.restype = None
        fn.argtypes = [c_void_p, _util.c_float_p, c_int, _util.c_float_p]
        for i in range(frames.shape[0]):
            fn(self._obj, _util.fptr(frames[i]), hc_num, _util.fptr(out[i]))
        return np.ascontiguousarray(np.swapaxes(_util.restore_leading(out, lead), -1, -2))

    def deconv(self, m_data_arr):
        """(..., num, time) magnitudes of the LAST cqt call (complex: magnitude) ->
        (timbre, pitch), each (..., num, time)"""
        m = np.asarray(m_data_arr)
        if np.iscomplexobj(m):
            m = np.abs(m)
        m = _util.as_f32(np.swapaxes(m, -1, -2))
        frames, lead = _util.flatten_leading(m, 2)
        tone = np.zeros_like(frames)
        pitch = np.zeros_like(frames)
        fn = self._lib.cqtObj_deconv
        fn = _lib.checked(fn)
        fn.restype = None
        fn.argtypes = [c_void_p, _util.c_float_p, _util.c_float_p, _util.c_float_p]
        for i in range(frames.shape[0]):
            fn(self._obj, _util.fptr(frames[i]), _util.fptr(tone[i]), _util.fptr(pitch[i]))
        return tuple(np.ascontiguousarray(np.swapaxes(_util.restore_leading(o, lead), -1, -2))
                     for o in (tone, pitch))

    def __del__(self):
        if getattr(self, "_obj", None):
            fn = self._lib.cqtObj_free
            fn.argtypes = [c_void_p]
            fn.restype = None
            fn(self._obj)
            self._obj = c_void_p(None)
