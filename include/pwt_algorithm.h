/* pwt_algorithm.h -- C ABI of the pseudo wavelet transform object: the CWT pipeline (forward FFT of
 * the (reflect-padded) signal, one frequency-domain product per band, one inverse FFT per band)
 * with the auditory filter bank (mel / bark / erb / octave / linspace / log / linear scale,
 * slaney / ETSI / window styles) as the bank instead of an analytic wavelet.
 *
 * Replaces the reference functions of the same names (src/pwt_algorithm.h:14-31,
 * src/pwt_algorithm.c:65-592) as bound by python/audioflux/pwt.py.  Execution: the CWT kernels
 * (csrc/hip/afx_cwt.hip) through the shared object plan (afx_cwt.c).
 */
#ifndef PWT_ALGORITHM_H
#define PWT_ALGORITHM_H

#include "flux_base.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OpaquePWT *PWTObj;

/* num 2..2^radix2Exp/2+1 bands over 2^radix2Exp samples per call; optional pointers NULL ->
 * samplate 32000, lowFre 0 (octave / log: C1), highFre samplate/2, binPerOctave 12 (4..48),
 * scale Octave, style Slaney, normal None, isPadding 0 (1: reflect-pad by half a block).
 * returns 0, -100 bad radix2Exp, 1 bad scale type, -1 bad num / range, -4 for what this
 * backend does not run: padding beyond 2^16 samples (non-power-of-two transform); <= -2 backend failure.
 * (The gammatone style is built as the reference builds it in this mode -- responses written at the half-spectrum
 * pitch, normalised and doubled at the full pitch, auditory_filterBank.c:509-591: the upper half of the rows is zero.)
 * replaces pwt_algorithm.c:65-293 */
int pwtObj_new(PWTObj *pwtObj, int num, int radix2Exp, int *samplate, float *lowFre, float *highFre,
               int *binPerOctave, SpectralFilterBankScaleType *scaleType,
               SpectralFilterBankStyleType *styleType, SpectralFilterBankNormalType *normalType,
               int *isPadding);

/* borrowed, num valid entries.  replace pwt_algorithm.c:295-303 */
float *pwtObj_getFreBandArr(PWTObj pwtObj);
int *pwtObj_getBinBandArr(PWTObj pwtObj);

/* dataArr[2^radix2Exp] -> mRealArr3 / mImageArr3 [num, 2^radix2Exp] (band order).
 * replaces pwt_algorithm.c:305-308, :398-515 */
void pwtObj_pwt(PWTObj pwtObj, float *dataArr, float *mRealArr3, float *mImageArr3);

/* time derivative variant (bank x j omega); dataArr NULL re-uses the last forward FFT.
 * replace pwt_algorithm.c:310-396 */
void pwtObj_enableDet(PWTObj pwtObj, int flag);
void pwtObj_pwtDet(PWTObj pwtObj, float *dataArr, float *mRealArr3, float *mImageArr3);

/* NULL-safe */
void pwtObj_free(PWTObj pwtObj);

#ifdef __cplusplus
}
#endif
#endif /* PWT_ALGORITHM_H */
