/* cqt_algorithm.h -- C ABI of the constant-Q transform object (octave
 * recursion: rectangular-window STFT x sparse spectral kernel, then a
 * half-band decimation per octave) plus CQT-chroma and CQCC, on an MI355X.
 *
 * Replaces the reference functions of the same names
 * (src/cqt_algorithm.h:14-61, src/cqt_algorithm.c:123-1417) as bound by
 * python/audioflux/cqt.py:60-660.
 */
#ifndef CQT_ALGORITHM_H
#define CQT_ALGORITHM_H

#include "flux_base.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OpaqueCQT *CQTObj;

/* shorthand for cqtObj_newWith with every optional left at its default */
int cqtObj_new(CQTObj *cqtObj, int num, int samplate, float minFre, int *isContinue);

/* num must be a multiple of binPerOctave.  defaults: samplate 32000,
 * minFre 32.703 (C1), binPerOctave 12, factor 1, beta 0, thresh 0.01,
 * window Hann, slideLength fftLength/4, isContinue 0, normal None, isScale 1.
 * isContinue 1: cqtObj_cqt keeps the samples that do not fill a frame for the next call and frames from
 * sample 0 (cqt_algorithm.c:345-456, :923-928); cqtObj_calTimeLength then counts the kept samples in.
 * returns 0 ok, -1 bad arguments, <= -2 backend.
 * replaces cqtObj_newWith, cqt_algorithm.c:123-247 */
int cqtObj_newWith(CQTObj *cqtObj, int num,
                   int *samplate, float *minFre, int *binPerOctave,
                   float *factor, float *beta, float *thresh,
                   WindowType *windowType, int *slideLength, int *isContinue,
                   SpectralFilterBankNormalType *filterNormalType, int *isScale);

int cqtObj_calTimeLength(CQTObj cqtObj, int dataLength);
int cqtObj_getFFTLength(CQTObj cqtObj);
/* library-owned, num entries */
float *cqtObj_getFreBandArr(CQTObj cqtObj);
void cqtObj_setScale(CQTObj cqtObj, int flag);

/* dataArr[dataLength] -> mRealArr/mImageArr [T, num].
 * replaces cqtObj_cqt, cqt_algorithm.c:463-482,845-1061 */
void cqtObj_cqt(CQTObj cqtObj, float *dataArr, int dataLength,
                float *mRealArr, float *mImageArr);

/* fold the last CQT's [T,num] result (passed back in) to chroma [T,chromaNum].
 * defaults: chromaNum 12, dataType Power, normType Max.
 * replaces cqtObj_chroma, cqt_algorithm.c:484-597 */
void cqtObj_chroma(CQTObj cqtObj, int *chromaNum, SpectralDataType *dataType,
                   ChromaDataNormalType *normType,
                   float *mRealArr, float *mImageArr,
                   float *mDataArr);

/* cepstral coefficients of a CQT magnitude/power matrix [T,num] -> [T,ccNum].
 * replaces cqtObj_cqcc, cqt_algorithm.c:599-660 */
void cqtObj_cqcc(CQTObj cqtObj, float *mDataArr1, int ccNum,
                 CepstralRectifyType *rectifyType, float *mDataArr2);

/* harmonic coefficients: mDataArr1 [T,num] (magnitude or power of the LAST cqt call) ->
 * mDataArr2 [T,hcNum]: the frame, zero padded to ceil_pow2(2 num), through FFT -> |.| ->
 * inverse FFT, sampled at round(binPerOctave * log2(j + 1)).
 * replaces cqtObj_cqhc, cqt_algorithm.c:662-711.
 * cqtObj_deconv: same transform -> mDataArr2 "timbre" [T,num] (Re IFFT |F|) and mDataArr3
 * "pitch" [T,num] (Re IFFT F / max(|F|, 1e-16)); replaces cqt_algorithm.c:718-781 */
void cqtObj_cqhc(CQTObj cqtObj, float *mDataArr1, int hcNum, float *mDataArr2);
void cqtObj_deconv(CQTObj cqtObj, float *mDataArr1, float *mDataArr2, float *mDataArr3);

/* NULL-safe */
void cqtObj_free(CQTObj cqtObj);

#ifdef __cplusplus
}
#endif
#endif /* CQT_ALGORITHM_H */
