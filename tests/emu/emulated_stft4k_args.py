"""ctypes mirror of AfxStftArgs (audioflux_amd/csrc/hip/afx_device.h) for the emulated STFT scripts"""
import ctypes as C

fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int)


class AfxStftArgs(C.Structure):  # audioflux_amd/csrc/hip/afx_device.h
    _fields_ = [("x", fp), ("clipStride", C.c_longlong), ("batch", C.c_int), ("dataLength", C.c_int), ("timeLength", C.c_int),
                ("radix2Exp", C.c_int), ("hop", C.c_int), ("window", fp), ("twiddle", fp), ("mode", C.c_int),
                ("normValue", C.c_float), ("binLo", C.c_int), ("binCount", C.c_int), ("outPitch", C.c_longlong),
                ("outRe", fp), ("outIm", fp), ("energy", fp), ("rms", fp), ("zcr", fp), ("padLeft", C.c_int),
                ("bandStart", ip), ("bandLen", ip), ("bandOff", ip), ("bandW", fp), ("bandNum", C.c_int),
                ("bandPost", C.c_int), ("bandPostArg", C.c_float), ("fullSpectrum", C.c_int), ("padMode", C.c_int),
                ("padValueL", C.c_float), ("padValueR", C.c_float)]
