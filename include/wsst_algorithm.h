/* wsst_algorithm.h -- C ABI of the wavelet synchrosqueezed transform: CWT and its time derivative
 * (cwtObj_cwt + cwtObj_cwtDet), instantaneous frequency Im(W'/W)/2pi per coefficient, and the
 * reassignment ("squeezing") of every coefficient above the threshold to the band that frequency
 * maps to.
 *
 * Replaces the reference functions of the same names (src/wsst_algorithm.h:28-49,
 * src/wsst_algorithm.c:64-444) as bound by python/audioflux/wsst.py.  Execution: the CWT kernels
 * (afx_cwt.hip) + k_wsst_squeeze (afx_wsst.hip).
 *
 * Parity note: the target band is a ROUNDED function of a float32 ratio; a coefficient whose
 * frequency coordinate lies within float32 rounding of a .5 boundary may land in the neighbouring
 * band relative to the reference (any implementation whose CWT is not bit-identical does this).
 * tests/test_wsst_gpu.py bounds exactly that: every difference must be explained by such
 * boundary coefficients, everything else meets 1e-5.
 */
#ifndef WSST_ALGORITHM_H
#define WSST_ALGORITHM_H

#include "flux_base.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OpaqueWSST *WSSTObj;

/* parameters as cwtObj_new (cwt_algorithm.h) with defaults wavelet Morlet, scale Octave;
 * thresh NULL / < 0 -> 0.001.  returns cwtObj_new's status (0, -100, 1, -1, <= -2).
 * replaces wsst_algorithm.c:64-152 */
int wsstObj_new(WSSTObj *wsstObj, int num, int radix2Exp, int *samplate, float *lowFre, float *highFre,
                int *binPerOctave, WaveletContinueType *waveletType,
                SpectralFilterBankScaleType *scaleType, float *gamma, float *beta, float *thresh,
                int *isPadding);

/* borrowed, num valid entries.  replace wsst_algorithm.c:154-162 */
float *wsstObj_getFreBandArr(WSSTObj wsstObj);
int *wsstObj_getBinBandArr(WSSTObj wsstObj);

/* The reference's order > 1 branch dereferences a buffer it never allocates
 * (wsst_algorithm.c:296-315, mTempIndexArr); this backend stores the value and squeezes once. */
void wsstObj_setOrder(WSSTObj wsstObj, int order);

/* dataArr[2^radix2Exp] -> mRealArr1/mImageArr1 [num, 2^radix2Exp]: squeezed coefficients ADDED to
 * the caller's content (the wrapper passes zeros); mRealArr2/mImageArr2 (may be NULL): the CWT.
 * replaces wsst_algorithm.c:170-347 */
void wsstObj_wsst(WSSTObj wsstObj, float *dataArr, float *mRealArr1, float *mImageArr1, float *mRealArr2,
                  float *mImageArr2);

/* NULL-safe */
void wsstObj_free(WSSTObj wsstObj);

#ifdef __cplusplus
}
#endif
#endif /* WSST_ALGORITHM_H */
