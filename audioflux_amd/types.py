"""Integer enums of the C ABI (values are ABI: include/flux_base.h, which follows
the reference's src/flux_base.h:14-187).  Member names follow the reference's
Python wrapper (python/audioflux/type/basic.py) so call sites read the same."""
from enum import IntEnum


class WindowType(IntEnum):
    RECT = 0
    HANN = 1
    HAMM = 2
    BLACKMAN = 3
    KAISER = 4
    BARTLETT = 5
    TRIANG = 6
    FLATTOP = 7
    GAUSS = 8
    BLACKMAN_HARRIS = 9
    BLACKMAN_NUTTALL = 10
    BARTLETT_HANN = 11
    BOHMAN = 12
    TUKEY = 13


class SpectralDataType(IntEnum):
    POWER = 0
    MAG = 1


class SpectralFilterBankScaleType(IntEnum):
    LINEAR = 0
    LINSPACE = 1
    MEL = 2
    BARK = 3
    ERB = 4
    OCTAVE = 5
    LOG = 6
    DEEP = 7
    CHROMA = 8
    OCTAVE_CHROMA = 9
    DEEP_CHROMA = 10


class SpectralFilterBankStyleType(IntEnum):
    SLANEY = 0
    ETSI = 1
    GAMMATONE = 2
    POINT = 3
    RECT = 4
    HANN = 5
    HAMM = 6
    BLACKMAN = 7
    BOHMAN = 8
    KAISER = 9
    GAUSS = 10


class SpectralFilterBankNormalType(IntEnum):
    NONE = 0
    AREA = 1
    BAND_WIDTH = 2


class ChromaDataNormalType(IntEnum):
    NONE = 0
    MAX = 1
    MIN = 2
    P2 = 3
    P1 = 4


class CepstralRectifyType(IntEnum):
    LOG = 0
    CUBIC_ROOT = 1


class CepstralEnergyType(IntEnum):
    REPLACE = 0
    APPEND = 1
    IGNORE = 2


class WaveletContinueType(IntEnum):
    MORSE = 0
    MORLET = 1
    BUMP = 2
    PAUL = 3
    DOG = 4
    MEXICAN = 5
    HERMIT = 6
    RICKER = 7


class PaddingPositionType(IntEnum):
    CENTER = 0
    RIGHT = 1
    LEFT = 2


class PaddingModeType(IntEnum):
    CONSTANT = 0
    REFLECT = 1
    WRAP = 2


class ReassignType(IntEnum):
    ALL = 0
    FRE = 1
    TIME = 2
    NONE = 3
