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
    Window_Rect = 0,                          /* all ones */
    Window_Hann = 1,                          /* 0.5 - 0.5 cos, periodic (length+1 symmetric window minus its last sample) */
    Window_Hamm = 2,                          /* 0.54 - 0.46 cos, periodic */
    Window_Blackman = 3,                      /* 3-term cosine sum, end samples forced to 0 */
    Window_Kaiser = 4,                        /* beta 5, I0 by a 15-term series */
    Window_Bartlett = 5,                      /* symmetric */
    Window_Triang = 6,                        /* symmetric */
    Window_Flattop = 7,                       /* 5-term cosine sum */
    Window_Gauss = 8,                         /* alpha 2.5 */
    Window_Blackman_Harris = 9,               /* 4-term, -92 dB */
    Window_Blackman_Nuttall = 10,             /* 4-term */
    Window_Bartlett_Hann = 11,                /* symmetric */
    Window_Bohman = 12,                       /* symmetric */
    Window_Tukey = 13                         /* tapered cosine, alpha 0.5 */
} WindowType;

/* what the filter bank is applied to (reference flux_base.h:49-53) */
typedef enum {
    SpectralData_Power = 0,                   /* |S|^2 (optionally ^normValue) feeds the bank */
    SpectralData_Mag = 1                      /* |S| feeds the bank; normValue raises the bank's OUTPUT */
} SpectralDataType;

/* frequency axis of the band centres (reference flux_base.h:55-74) */
typedef enum {
    SpectralFilterBankScale_Linear = 0,       /* bin slice [lowIndex, highIndex], no bank matrix */
    SpectralFilterBankScale_Linspace = 1,     /* centres equally spaced in Hz */
    SpectralFilterBankScale_Mel = 2,          /* 2595 log10(1 + f/700) */
    SpectralFilterBankScale_Bark = 3,         /* Traunmueller with the low / high corrections */
    SpectralFilterBankScale_Erb = 4,          /* 21.3654 log10(1 + 0.004368 f) */
    SpectralFilterBankScale_Octave = 5,       /* binPerOctave steps per octave from the band of lowFre */
    SpectralFilterBankScale_Log = 6,          /* equally spaced in log2(f / 440) */
    SpectralFilterBankScale_Deep = 7,         /* spectrogram object only: salience model, refused (-4) */
    SpectralFilterBankScale_Chroma = 8,       /* spectrogram object only: Gaussian STFT-chroma bank */
    SpectralFilterBankScale_LogChroma = 9,    /* spectrogram object only: log bank folded onto chroma */
    SpectralFilterBankScale_DeepChroma = 10   /* spectrogram object only: refused (-4) */
} SpectralFilterBankScaleType;

/* shape of each band (reference flux_base.h:76-93) */
typedef enum {
    SpectralFilterBankStyle_Slaney = 0,       /* triangles in Hz between neighbouring centres */
    SpectralFilterBankStyle_ETSI = 1,         /* triangles in bins */
    SpectralFilterBankStyle_Gammatone = 2,    /* 4th-order gammatone magnitude response: dense rows */
    SpectralFilterBankStyle_Point = 3,        /* one unit weight at the centre bin */
    SpectralFilterBankStyle_Rect = 4,         /* ones between the neighbouring centres */
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
