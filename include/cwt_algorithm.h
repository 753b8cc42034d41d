/* cwt_algorithm.h -- C ABI of the continuous wavelet transform object:
 * (reflect pad) -> FFT(L) -> x num frequency-domain wavelets -> num inverse
 * FFT(L) -> crop, on an MI355X.
 *
 * Replaces the reference functions of the same names
 * (src/cwt_algorithm.h:28-43, src/cwt_algorithm.c:73-715) as bound by
 * python/audioflux/cwt.py:126-318.
 */
#ifndef CWT_ALGORITHM_H
#define CWT_ALGORITHM_H

#include "flux_base.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OpaqueCWT *CWTObj;

/* num scales; the transform length is 2^radix2Exp samples.
 * defaults: samplate 32000, binPerOctave 12, wavelet Morlet, scale Octave,
 * (gamma,beta) per wavelet family, isPadding 1.
 * returns 0 ok, negative on bad arguments or backend failure.
 * replaces cwtObj_new, cwt_algorithm.c:73-334 */
int cwtObj_new(CWTObj *cwtObj, int num, int radix2Exp,
               int *samplate, float *lowFre, float *highFre, int *binPerOctave,
               WaveletContinueType *waveletType,
               SpectralFilterBankScaleType *scaleType,
               float *gamma, float *beta,
               int *isPadding);

/* library-owned, num entries, ascending frequency */
float *cwtObj_getFreBandArr(CWTObj cwtObj);
int *cwtObj_getBinBandArr(CWTObj cwtObj);

/* dataArr[2^radix2Exp] -> mRealArr3/mImageArr3 [num, 2^radix2Exp];
 * row 0 is the HIGHEST frequency (the Python wrapper flips afterwards).
 * replaces cwtObj_cwt, cwt_algorithm.c:346-351,361-483 */
void cwtObj_cwt(CWTObj cwtObj, float *dataArr, float *mRealArr3, float *mImageArr3);

/* enable the d/dt variant (wavelet bank multiplied by j*omega) */
void cwtObj_enableDet(CWTObj cwtObj, int flag);
/* dataArr NULL -> re-use the spectrum of the preceding cwtObj_cwt call */
void cwtObj_cwtDet(CWTObj cwtObj, float *dataArr, float *mRealArr3, float *mImageArr3);

/* NULL-safe */
void cwtObj_free(CWTObj cwtObj);

#ifdef __cplusplus
}
#endif
#endif /* CWT_ALGORITHM_H */
