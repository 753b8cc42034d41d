/* flux_base.h -- integer ABI of every optional parameter on the hot path.
 *
 * The audioFlux ctypes wrapper passes enums as `int*`; the numeric VALUES
 * below are therefore part of the binary interface and equal the ones the
 * reference defines in src/flux_base.h:14-187.  Only the enums that the
 * BFT / XXCC / Cepstrogram / CWT / CQT entry points consume are declared.
 */
#ifndef FLUX_BASE_H
#define FLUX_BASE_H

#ifdef __cplusplus
extern "C" {
#endif

/* analysis window applied to each STFT frame (reference flux_base.h:14-36) */
typedef enum {
    Window_Rect = 0,
    Window_Hann = 1,
    Window_Hamm = 2,
    Window_Blackman = 3,
    Window_Kaiser = 4,
    Window_Bartlett = 5,
    Window_Triang = 6,
    Window_Flattop = 7,
    Window_Gauss = 8,
    Window_Blackman_Harris = 9,
    Window_Blackman_Nuttall = 10,
    Window_Bartlett_Hann = 11,
    Window_Bohman = 12,
    Window_Tukey = 13
} WindowType;

/* what the filter bank is applied to (reference flux_base.h:49-53) */
typedef enum {
    SpectralData_Power = 0,
    SpectralData_Mag = 1
} SpectralDataType;

/* frequency axis of the band centres (reference flux_base.h:55-74) */
typedef enum {
    SpectralFilterBankScale_Linear = 0,
    SpectralFilterBankScale_Linspace = 1,
    SpectralFilterBankScale_Mel = 2,
    SpectralFilterBankScale_Bark = 3,
    SpectralFilterBankScale_Erb = 4,
    SpectralFilterBankScale_Octave = 5,
    SpectralFilterBankScale_Log = 6,
    SpectralFilterBankScale_Deep = 7,
    SpectralFilterBankScale_Chroma = 8,
    SpectralFilterBankScale_LogChroma = 9,
    SpectralFilterBankScale_DeepChroma = 10
} SpectralFilterBankScaleType;

/* shape of each band (reference flux_base.h:76-93) */
typedef enum {
    SpectralFilterBankStyle_Slaney = 0,
    SpectralFilterBankStyle_ETSI = 1,
    SpectralFilterBankStyle_Gammatone = 2,
    SpectralFilterBankStyle_Point = 3,
    SpectralFilterBankStyle_Rect = 4,
    SpectralFilterBankStyle_Hann = 5,
    SpectralFilterBankStyle_Hamm = 6,
    SpectralFilterBankStyle_Blackman = 7,
    SpectralFilterBankStyle_Bohman = 8,
    SpectralFilterBankStyle_Kaiser = 9,
    SpectralFilterBankStyle_Gauss = 10
} SpectralFilterBankStyleType;

/* band normalisation (reference flux_base.h:95-101) */
typedef enum {
    SpectralFilterBankNormal_None = 0,
    SpectralFilterBankNormal_Area = 1,
    SpectralFilterBankNormal_BandWidth = 2
} SpectralFilterBankNormalType;

/* per-frame chroma normalisation (reference flux_base.h:117-127) */
typedef enum {
    ChromaDataNormal_None = 0,
    ChromaDataNormal_Max = 1,
    ChromaDataNormal_Min = 2,
    ChromaDataNormal_P2 = 3,
    ChromaDataNormal_P1 = 4
} ChromaDataNormalType;

/* cepstral rectification (reference flux_base.h:129-133) */
typedef enum {
    CepstralRectify_Log = 0,
    CepstralRectify_CubicRoot = 1
} CepstralRectifyType;

/* where the log-energy goes in the "standard" cepstra (reference flux_base.h:135-140) */
typedef enum {
    CepstralEnergy_Replace = 0,
    CepstralEnergy_Append = 1,
    CepstralEnergy_Ignore = 2
} CepstralEnergyType;

/* STFT frame padding (reference flux_base.h:142-154) */
typedef enum {
    PaddingPosition_Center = 0,
    PaddingPosition_Right = 1,
    PaddingPosition_Left = 2
} PaddingPositionType;

typedef enum {
    PaddingMode_Constant = 0,
    PaddingMode_Reflect = 1,
    PaddingMode_Wrap = 2
} PaddingModeType;

/* continuous wavelet family (reference flux_base.h:156-169) */
typedef enum {
    WaveletContinue_Morse = 0,
    WaveletContinue_Morlet = 1,
    WaveletContinue_Bump = 2,
    WaveletContinue_Paul = 3,
    WaveletContinue_DOG = 4,
    WaveletContinue_Mexican = 5,
    WaveletContinue_Hermit = 6,
    WaveletContinue_Ricker = 7
} WaveletContinueType;

#ifdef __cplusplus
}
#endif
#endif /* FLUX_BASE_H */
