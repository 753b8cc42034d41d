"""The north-star call: batched STFT -> filter bank -> cepstra on HBM-resident
clips (additive API afx_bftXxccBatchDevice, include/afx_batch.h)."""
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
