/* spectrogram_algorithm.h -- C ABI of the spectrogram object (core): STFT -> power / magnitude
 * -> linear slice | mel / bark / erb / octave / linspace / log filter bank | STFT-chroma |
 * log-chroma, optional streaming continuation, cepstral coefficients of the result
 * (mfcc / bfcc / gtcc / xxcc) and the spectral deconvolution.  This is what the reference's
 * MelSpectrogram / BarkSpectrogram / ErbSpectrogram / Spectrogram / Linear / Mel / Bark / Erb /
 * Chroma wrapper classes and its published benchmark (benchmark/run_audioflux.py:15-22) call.
 *
 * Replaces the reference functions of the same names (src/spectrogram_algorithm.h:43-106,
 * src/spectrogram_algorithm.c:186-1525, :1540-1612, :3029-3200) as bound by
 * python/audioflux/spectrogram.py.  Execution: the kernels of the BFT path (fused STFT ->
 * banded filter bank, size-generic STFT, MFMA GEMM, MFMA cepstra) plus k_spec_map / k_row_post.
 *
 * Not provided by this backend: the "deep" salience scales (newDeep /
 * newDeepChroma return -4) and the spectral-descriptor family (flatness ... novelty), which the
 * wrapper resolves lazily and which are outside the batched time-frequency path.
 */
#ifndef SPECTROGRAM_ALGORITHM_H
#define SPECTROGRAM_ALGORITHM_H

#include "flux_base.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OpaqueSpectrogram *SpectrogramObj;

/* every pointer argument is optional (NULL -> reference default): samplate 32000, lowFre 0
 * (octave-like scales: C1 .. 38 semitones above A4), highFre samplate/2, binPerOctave 12,
 * radix2Exp 12, window Hann, slideLength fftLength/4, isContinue 0, dataType Power, scale
 * Linear, style Slaney, normal None.  Linear: num is derived from [lowFre, highFre]
 * (spectrogramObj_getBandNum).  returns 0, -100 bad radix2Exp, -1 bad num / range,
 * -4 deep scales, <= -2 backend failure.  replaces spectrogram_algorithm.c:326-582 */
int spectrogramObj_new(SpectrogramObj *spectrogramObj, int num, int *samplate, float *lowFre,
                       float *highFre, int *binPerOctave, int *radix2Exp, WindowType *windowType,
                       int *slideLength, int *isContinue, SpectralDataType *dataType,
                       SpectralFilterBankScaleType *filterScaleType,
                       SpectralFilterBankStyleType *filterStyleType,
                       SpectralFilterBankNormalType *filterNormalType);

/* replace spectrogram_algorithm.c:186-324 */
int spectrogramObj_newLinear(SpectrogramObj *spectrogramObj, int samplate, int radix2Exp, int *isContinue);
int spectrogramObj_newMel(SpectrogramObj *spectrogramObj, int num, int samplate, int radix2Exp, int *isContinue);
int spectrogramObj_newBark(SpectrogramObj *spectrogramObj, int num, int samplate, int radix2Exp, int *isContinue);
int spectrogramObj_newErb(SpectrogramObj *spectrogramObj, int num, int samplate, int radix2Exp, int *isContinue);
int spectrogramObj_newChroma(SpectrogramObj *spectrogramObj, int samplate, int radix2Exp, int *isContinue);
int spectrogramObj_newDeep(SpectrogramObj *spectrogramObj, int num, int samplate, int radix2Exp, int *isContinue);
int spectrogramObj_newDeepChroma(SpectrogramObj *spectrogramObj, int samplate, int radix2Exp, int *isContinue);

/* replace spectrogram_algorithm.c:822-846 */
void spectrogramObj_setDeepOrder(SpectrogramObj spectrogramObj, int deepOrder);
void spectrogramObj_setChromaDataNormalType(SpectrogramObj spectrogramObj, ChromaDataNormalType dataNormType);
void spectrogramObj_setDataNormValue(SpectrogramObj spectrogramObj, float normValue);

/* replace spectrogram_algorithm.c:848-853, :3171-3200 */
int spectrogramObj_calTimeLength(SpectrogramObj spectrogramObj, int dataLength);
void spectrogramObj_enableDebug(SpectrogramObj spectrogramObj, int flag);
float *spectrogramObj_getFreBandArr(SpectrogramObj spectrogramObj);
int *spectrogramObj_getBinBandArr(SpectrogramObj spectrogramObj);
int spectrogramObj_getBandNum(SpectrogramObj spectrogramObj);
int spectrogramObj_getBinBandLength(SpectrogramObj spectrogramObj);

/* dataArr[dataLength] -> mSpectArr [T, num]; mPhaseArr [T, num] (linear scale only, may be NULL).
 * replaces spectrogram_algorithm.c:864-1393 */
void spectrogramObj_spectrogram(SpectrogramObj spectrogramObj, float *dataArr, int dataLength,
                                float *mSpectArr, float *mPhaseArr);
/* the same from a caller-supplied STFT mRealArr/mImageArr [nLength, mLength == fftLength].
 * replaces spectrogram_algorithm.c:1397-1402 */
void spectrogramObj_spectrogram1(SpectrogramObj spectrogramObj, float *mRealArr, float *mImageArr,
                                 int nLength, int mLength, float *mSpectArr, float *mPhaseArr);

/* cepstral coefficients of the LAST spectrogram's frame count: mDataArr1 [T, num] ->
 * mDataArr2 [T, ccNum].  mfcc / bfcc / gtcc act only on a mel / bark scale / gammatone style
 * object, as in the reference.  replace spectrogram_algorithm.c:1409-1525 */
void spectrogramObj_mfcc(SpectrogramObj spectrogramObj, float *mDataArr1, int ccNum, float *mDataArr2);
void spectrogramObj_gtcc(SpectrogramObj spectrogramObj, float *mDataArr1, int ccNum, float *mDataArr2);
void spectrogramObj_bfcc(SpectrogramObj spectrogramObj, float *mDataArr1, int ccNum, float *mDataArr2);
void spectrogramObj_xxcc(SpectrogramObj spectrogramObj, float *mDataArr1, int ccNum,
                         CepstralRectifyType *rectifyType, float *mDataArr2);
/* empty in the reference (spectrogram_algorithm.c:1527-1538); kept for ABI compatibility */
void spectrogramObj_mfccStandard(SpectrogramObj spectrogramObj, float *mDataArr1, int *deltaWindowLength,
                                 CepstralEnergyType *energyType, CepstralRectifyType *rectifyType,
                                 float *mDataArr2);
void spectrogramObj_xxccStandard(SpectrogramObj spectrogramObj, float *mDataArr1, int *deltaWindowLength,
                                 CepstralEnergyType *energyType, CepstralRectifyType *rectifyType,
                                 float *mDataArr2);

/* mDataArr1 [T, num] -> timbre (formant) mDataArr2, pitch mDataArr3, both [T, num].
 * replaces spectrogram_algorithm.c:1546-1612 */
void spectrogramObj_deconv(SpectrogramObj spectrogramObj, float *mDataArr1, float *mDataArr2, float *mDataArr3);

/* NULL-safe.  replaces spectrogram_algorithm.c:3029-3169 */
void spectrogramObj_free(SpectrogramObj spectrogramObj);

#ifdef __cplusplus
}
#endif
#endif /* SPECTROGRAM_ALGORITHM_H */
