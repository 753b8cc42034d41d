"""The north-star call: batched STFT -> filter bank -> cepstra on HBM-resident
clips (additive API afx_bftXxccBatchDevice, include/afx_batch.h)."""
import ctypes
from ctypes import POINTER, c_int, c_longlong, c_void_p

from . import _lib, _util
from .types import CepstralRectifyType


def mel_mfcc_device(bft, xxcc, x, cc_num=13, rectify_type=CepstralRectifyType.LOG,
                    want_mel=True, out_mel=None, out_cc=None, stream=None):
    """x: CUDA/HIP torch.float32 (clips, n).  Returns (mel, mfcc) tensors shaped
    (clips, time, num) / (clips, time, cc_num); mel is None when want_mel=False."""
    import torch
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
    if bft.result_type != 1:
        bft.set_result_type(1)
    b, n = x.shape
    t = bft.cal_time_length(n)
    if want_mel and out_mel is None:
        out_mel = torch.empty((b, t, bft.num), dtype=torch.float32, device=x.device)
    if out_cc is None:
        out_cc = torch.empty((b, t, cc_num), dtype=torch.float32, device=x.device)
    s = stream if stream is not None else torch.cuda.current_stream(x.device)
    fn = _lib.get_lib().afx_bftXxccBatchDevice
    fn.restype = c_int
    fn.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_longlong, c_int, POINTER(c_int),
                   c_void_p, c_void_p, c_void_p]
    _lib.check(fn(bft._obj, xxcc._obj, x.data_ptr(), b, n, x.stride(0), cc_num,
                  _util.opt_int(int(rectify_type)),
                  out_mel.data_ptr() if want_mel else None, out_cc.data_ptr(), s.cuda_stream),
               "afx_bftXxccBatchDevice")
    return (out_mel if want_mel else None), out_cc


class ClockProbe:
    """Shader clock held while other kernels run (include/afx_batch.h: afx_clock_probe_start): a sleeping wave on a side
    stream samples s_memtime against the constant reference clock; `start()` before the measured launches, `stop()` after
    them (ordered behind them on the current stream) returns MHz.  Measurement aid of bench.py, not part of the path."""

    def __init__(self, torch, dev, max_seconds=20.0):
        self.torch, self.dev, self.max_seconds = torch, dev, float(max_seconds)
        self.side = torch.cuda.Stream(device=dev)
        self.out = torch.zeros(2, dtype=torch.int64, device=dev)
        self.flag = torch.zeros(1, dtype=torch.int32, device=dev)
        self.khz = ctypes.c_int(0)

    def start(self):
        self.flag.zero_()
        self.out.zero_()
        self.side.wait_stream(self.torch.cuda.current_stream(self.dev))
        _lib.check(_lib.get_lib().afx_clock_probe_start(
            ctypes.c_void_p(self.side.cuda_stream), ctypes.c_void_p(self.out.data_ptr()), ctypes.c_void_p(self.flag.data_ptr()),
            ctypes.c_double(self.max_seconds), ctypes.byref(self.khz)), "afx_clock_probe_start")

    def stop(self):
        self.flag.fill_(1)  # behind the measured launches on the current stream
        self.side.synchronize()
        cyc, ticks = (int(v) for v in self.out.cpu())
        return {"clock_mhz": cyc / ticks * self.khz.value / 1e3 if ticks else None, "shader_cycles": cyc,
                "reference_ticks": ticks, "reference_khz": self.khz.value}
