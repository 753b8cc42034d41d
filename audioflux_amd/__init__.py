"""audioflux_amd -- MI355X-native backend for audioFlux's batched
time-frequency hot path (BFT/STFT -> filter bank -> MFCC/xxcc, CWT, CQT,
cepstrogram).  The product is the C-ABI library lib/libaudioflux_mi355x.so
(hand-written gfx950 HIP kernels behind the reference's own C API); the
classes here mirror the reference's ctypes wrapper for that path."""
from . import _lib
from ._lib import LIB_PATH, build, get_lib, last_error, runtime_status
from .types import *  # noqa: F401,F403
from .bft import BFT
from .xxcc import XXCC
from .cepstrogram import Cepstrogram
from .cqt import CQT
from .cwt import CWT
from .pwt import PWT
from .reassign import Reassign
from .stft import STFT
from .synsq import Synsq
from .wsst import WSST
from .spectrogram import (Bark, BarkSpectrogram, Chroma, Erb, ErbSpectrogram, Linear, Mel, MelSpectrogram,
                          Spectrogram, SpectrogramBase, SpectralFilterBankType)
from .batch import mel_mfcc_device

__all__ = ["BFT", "XXCC", "Cepstrogram", "CQT", "CWT", "PWT", "Reassign", "STFT", "Synsq", "WSST", "Spectrogram", "SpectrogramBase", "MelSpectrogram", "BarkSpectrogram", "ErbSpectrogram",
           "Linear", "Mel", "Bark", "Erb", "Chroma", "SpectralFilterBankType", "mel_mfcc_device", "get_lib", "build", "runtime_status",
           "last_error", "LIB_PATH"]
