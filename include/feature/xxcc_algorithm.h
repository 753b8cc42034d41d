/* xxcc_algorithm.h -- C ABI of the cepstral-coefficient object (MFCC, BFCC,
 * GTCC, CQCC ...): rectify (log10 | cube root) -> orthonormal DCT-II ->
 * first ccNum coefficients, on an MI355X.
 *
 * Replaces the reference functions of the same names
 * (src/feature/xxcc_algorithm.h:14-38, src/feature/xxcc_algorithm.c:32-330)
 * as bound by python/audioflux/feature/xxcc.py:60-230.
 */
#ifndef XXCC_ALGORITHM_H
#define XXCC_ALGORITHM_H

#include "../flux_base.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OpaqueXXCC *XXCCObj;

/* num = bands of the input spectrogram (>= 2; this backend: <= 16384, -4 beyond -- the [num, num] DCT matrix is
 * built here).  returns 0, -1 on bad num, <= -2 on backend failure.  replaces xxcc_algorithm.c:32-62 */
int xxccObj_new(XXCCObj *xxccObj, int num);

/* number of frames the next xxcc call will process.  replaces xxcc_algorithm.c:64-89 */
void xxccObj_setTimeLength(XXCCObj xxccObj, int timeLength);

/* mDataArr1[T*num] -> mDataArr2[T*ccNum]; silently returns when ccNum > num
 * (reference behaviour).  rectifyType NULL -> Log: log10(max(x,1e-8)).
 * replaces xxccObj_xxcc, xxcc_algorithm.c:95-156 */
void xxccObj_xxcc(XXCCObj xxccObj, float *mDataArr1, int ccNum,
                  CepstralRectifyType *rectifyType, float *mDataArr2);

/* "standard" cepstra: coefficient 0 replaced by / prefixed with ln(energy),
 * then delta and delta-delta taken ALONG THE COEFFICIENT AXIS of each frame
 * (reference quirk, xxcc_algorithm.c:283-288).  Output row length is ccNum
 * (Replace/Ignore) or ccNum+1 (Append).
 * replaces xxccObj_xxccStandard, xxcc_algorithm.c:168-296 */
void xxccObj_xxccStandard(XXCCObj xxccObj, float *mDataArr1, int ccNum, float *energyArr,
                          int *deltaWindowLength, CepstralEnergyType *energyType,
                          CepstralRectifyType *rectifyType,
                          float *mCoeArr, float *mDeltaArr1, float *mDeltaArr2);

/* NULL-safe.  replaces xxcc_algorithm.c:298-320 */
void xxccObj_free(XXCCObj xxccObj);

#ifdef __cplusplus
}
#endif
#endif /* XXCC_ALGORITHM_H */
