/* reassign_algorithm.h -- C ABI of the time-frequency reassignment object: three STFTs (window h,
 * its derivative, the time-weighted window), reassigned coordinates t' = t + Re(S_th/S_h)/sr,
 * f' = f - Im(S_dh/S_h) sr/2pi per coefficient, and accumulation of every coefficient at the
 * grid cell nearest to (t', f').  This is also what bftObj_new(isReassign = 1) runs in front of
 * its filter bank.
 *
 * Replaces the reference functions of the same names (src/reassign_algorithm.h:15-56,
 * src/reassign_algorithm.c:84-928) as bound by python/audioflux/reassign.py.  Execution:
 * k_stft_generic x 3 + k_reassign_index / k_reassign_order / k_reassign_scatter
 * (csrc/hip/afx_reassign.hip).
 *
 * Parity note (as for wsst_algorithm.h): the target cell is a ROUNDED function of float32 ratios;
 * coefficients whose coordinates sit on a rounding boundary may land in a neighbouring cell
 * relative to the reference.  tests/test_reassign_gpu.py bounds exactly that.
 */
#ifndef REASSIGN_ALGORITHM_H
#define REASSIGN_ALGORITHM_H

#include "flux_base.h"

#ifdef __cplusplus
extern "C" {
#endif

/* reference reassign_algorithm.h:15-24 */
typedef enum {
    Reassign_All = 0,
    Reassign_Fre = 1,
    Reassign_Time = 2,
    Reassign_None = 3
} ReassignType;

typedef struct OpaqueReassign *ReassignObj;

/* radix2Exp outside 2..30 -> 12 (as the reference); samplate NULL -> 32000, windowType NULL ->
 * Hann, slideLength NULL/<=0 -> fftLength/4, reType NULL -> All, thresh NULL/<0 -> 0.001,
 * isPadding NULL -> 0 (1: centre zero padding).  isContinue is accepted and ignored, as in the
 * reference (reassign_algorithm.c:101, :149).  returns 0 or <= -2 (backend failure).
 * replaces reassign_algorithm.c:84-173 */
int reassignObj_new(ReassignObj *reassignObj, int radix2Exp, int *samplate, WindowType *windowType,
                    int *slideLength, ReassignType *reType, float *thresh, int *isPadding,
                    int *isContinue);

/* replaces reassign_algorithm.c:175-178 */
int reassignObj_calTimeLength(ReassignObj reassignObj, int dataLength);
/* 0 complex coefficients, 1 amplitudes (into mRealArr1).  replaces :180-184 */
void reassignObj_setResultType(ReassignObj reassignObj, int type);
/* order >= 1: iterations of the frequency-index map.  replaces :186-190 */
void reassignObj_setOrder(ReassignObj reassignObj, int order);

/* dataArr[dataLength] -> mRealArr1/mImageArr1 [T, fftLength/2+1]: reassigned coefficients ADDED
 * to the caller's content (the wrapper passes zeros; Reassign_None overwrites with the plain
 * STFT); mRealArr2/mImageArr2 (may be NULL): the STFT itself.  replaces :199-250, :256-414 */
void reassignObj_reassign(ReassignObj reassignObj, float *dataArr, int dataLength, float *mRealArr1,
                          float *mImageArr1, float *mRealArr2, float *mImageArr2);

/* NULL-safe */
void reassignObj_free(ReassignObj reassignObj);

#ifdef __cplusplus
}
#endif
#endif /* REASSIGN_ALGORITHM_H */
