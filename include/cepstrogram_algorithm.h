/* cepstrogram_algorithm.h -- C ABI of the cepstrogram object: STFT ->
 * ln|S|^2 -> inverse FFT (real cepstrum) -> low/high-quefrency lifter ->
 * FFT, giving cepstrum, spectral envelope and spectral detail per frame.
 *
 * Replaces the reference functions of the same names
 * (src/cepstrogram_algorithm.h:21-39, src/cepstrogram_algorithm.c:55-414) as
 * bound by python/audioflux/cepstrogram.py:60-189.
 */
#ifndef CEPSTROGRAM_ALGORITHM_H
#define CEPSTROGRAM_ALGORITHM_H

#include "flux_base.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OpaqueCepstrogram *CepstrogramObj;

/* radix2Exp 1..30 (fftLength = 2^radix2Exp); windowType NULL -> Rect;
 * slideLength NULL/<=0 -> fftLength/4.  returns 0, -100 bad radix2Exp,
 * <= -2 backend failure.  replaces cepstrogram_algorithm.c:55-109 */
int cepstrogramObj_new(CepstrogramObj *cepstrogramObj, int radix2Exp,
                       WindowType *windowType, int *slideLength);

/* replaces cepstrogram_algorithm.c (calTimeLength) */
int cepstrogramObj_calTimeLength(CepstrogramObj cepstrogramObj, int dataLength);

/* dataArr[dataLength] -> three [T, fftLength/2+1] matrices; mDataArr2 and
 * mDataArr3 may be NULL.  replaces cepstrogram_algorithm.c:111-125,127-298 */
void cepstrogramObj_cepstrogram(CepstrogramObj cepstrogramObj, int cepNum,
                                float *dataArr, int dataLength,
                                float *mDataArr1, float *mDataArr2, float *mDataArr3);

/* Spectrum-input variant -- WARNING, upstream quirk kept bit for bit: the reference's memcpy
 * runs the wrong way round (cepstrogram_algorithm.c:214-215).  The spectrum the CALLER passes
 * in mRealArr / mImageArr is NEVER USED: both arrays are OVERWRITTEN with the spectrum cached by
 * the previous cepstrogramObj_cepstrogram call (zeros if there was none) and the cepstra are
 * computed from that cache.  A drop-in must behave identically, so this backend does; callers
 * that want "cepstra of my spectrum" must not use this entry point -- run
 * cepstrogramObj_cepstrogram on the signal instead.  (The reference's Python wrapper does not
 * bind it.) */
void cepstrogramObj_cepstrogram2(CepstrogramObj cepstrogramObj, int cepNum,
                                 float *mRealArr, float *mImageArr, int nLength,
                                 float *mDataArr1, float *mDataArr2, float *mDataArr3);

/* kept for ABI compatibility; debug dumps are not produced by this backend */
void cepstrogramObj_enableDebug(CepstrogramObj cepstrogramObj, int flag);

/* NULL-safe */
void cepstrogramObj_free(CepstrogramObj cepstrogramObj);

#ifdef __cplusplus
}
#endif
#endif /* CEPSTROGRAM_ALGORITHM_H */
