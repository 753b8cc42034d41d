"""Array plumbing shared by the wrapper classes (channel flattening mirrors
python/audioflux/utils/util.py: leading axes are treated as independent clips)."""
import ctypes

import numpy as np

c_float_p = ctypes.POINTER(ctypes.c_float)
c_int_p = ctypes.POINTER(ctypes.c_int)


def fptr(a):
    return a.ctypes.data_as(c_float_p)


def opt_int(v):
    return None if v is None else ctypes.pointer(ctypes.c_int(int(v)))


def opt_float(v):
    return None if v is None else ctypes.pointer(ctypes.c_float(float(v)))


def as_f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def flatten_leading(a, keep):
    """reshape (..., d1..dkeep) -> (clips, d1..dkeep); returns array and the leading shape"""
    lead = a.shape[: a.ndim - keep]
    n = int(np.prod(lead)) if lead else 1
    return a.reshape((n,) + a.shape[a.ndim - keep:]), lead


def restore_leading(a, lead):
    return a.reshape(tuple(lead) + a.shape[1:])


def is_torch(x):
    return type(x).__module__.startswith("torch")
