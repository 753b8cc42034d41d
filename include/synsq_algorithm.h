/* synsq_algorithm.h -- C ABI of the synchrosqueezing object for ANY complex time-frequency matrix
 * (CWT, PWT, ST ...): instantaneous frequency from the time difference of the unwrapped phase,
 * mapped on the caller's band axis, and every coefficient above the threshold added to the band
 * that frequency maps to.
 *
 * Replaces the reference functions of the same names (src/synsq_algorithm.h:12-31,
 * src/synsq_algorithm.c:38-341) as bound by python/audioflux/synsq.py.  Execution:
 * k_synsq_phase + k_wsst_squeeze (csrc/hip/afx_wsst.hip).  Parity note: see wsst_algorithm.h.
 */
#ifndef SYNSQ_ALGORITHM_H
#define SYNSQ_ALGORITHM_H

#include "flux_base.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OpaqueSynsq *SynsqObj;

/* matrices are [num, 2^radix2Exp]; samplate NULL -> 32000 (values >= 196000 ignored); order is
 * taken only when > 1, thresh only when > 1 (default 0.001) -- the reference's conditions
 * (synsq_algorithm.c:66-82).  returns 0, -100 bad radix2Exp, -1 bad num, <= -2 backend failure.
 * replaces synsq_algorithm.c:38-124 */
int synsqObj_new(SynsqObj *synsqObj, int num, int radix2Exp, int *samplate, int *order, float *thresh);

/* freArr[num] band centres in Hz (ascending), scaleType of that axis (Linear .. Log);
 * mRealArr1/mImageArr1 [num, n] -> squeezed coefficients ADDED to mRealArr2/mImageArr2.
 * The reference's order > 1 branch indexes the [num, n] matrices with a [n, num] stride
 * (synsq_algorithm.c:238-252) and reads out of bounds for num != n; order 1 is used here.
 * replaces synsq_algorithm.c:134-281 */
void synsqObj_synsq(SynsqObj synsqObj, float *freArr, SpectralFilterBankScaleType scaleType,
                    float *mRealArr1, float *mImageArr1, float *mRealArr2, float *mImageArr2);

/* NULL-safe */
void synsqObj_free(SynsqObj synsqObj);

#ifdef __cplusplus
}
#endif
#endif /* SYNSQ_ALGORITHM_H */
