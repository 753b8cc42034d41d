/* stft_algorithm.h -- C ABI of the short-time Fourier transform object: framing with
 * optional padding (constant / reflect / wrap at centre / left / right) or streaming
 * continuation, window multiply and one complex FFT per frame; and the inverse
 * ((weighted) overlap-add).
 *
 * Replaces the reference functions of the same names (src/stft_algorithm.h:16-39,
 * src/stft_algorithm.c:80-871) as bound by python/audioflux/stft.py.  The framed FFT runs in
 * k_stft_generic (csrc/hip/afx_stft.hip) with the padding applied as an index map while the
 * frame is gathered; the inverse in k_istft_frames / k_istft_ola (csrc/hip/afx_istft.hip).
 */
#ifndef STFT_ALGORITHM_H
#define STFT_ALGORITHM_H

#include "flux_base.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OpaqueSTFT *STFTObj;

/* radix2Exp 1..30 (this backend: <= 14); windowType NULL -> Rect; slideLength NULL/<=0 ->
 * fftLength/4 (1 at fftLength 2, where the reference's default of 0 divides by zero); isContinue NULL -> 0.  returns 0, -100 bad radix2Exp, <= -2 backend failure.
 * replaces stft_algorithm.c:80-161 */
int stftObj_new(STFTObj *stftObj, int radix2Exp, WindowType *windowType, int *slideLength,
                int *isContinue);

/* replaces stft_algorithm.c:164-171 (values <= 0 are ignored) */
void stftObj_setSlideLength(STFTObj stftObj, int slideLength);
/* only honoured while padding is enabled; value1 = constant of the left / only pad,
 * value2 = right pad of the centre position.  The left / right positions truncate the
 * constant to an integer, as the reference's helpers do (vector/flux_vectorOp.c:641-651).
 * replaces stft_algorithm.c:185-207 */
void stftObj_setPadding(STFTObj stftObj, PaddingPositionType *positionType,
                        PaddingModeType *modeType, float *value1, float *value2);

/* replaces stft_algorithm.c:209-217; the returned array is library-owned, fftLength long */
void stftObj_useWindowDataArr(STFTObj stftObj, float *winDataArr);
float *stftObj_getWindowDataArr(STFTObj stftObj);

/* replaces stft_algorithm.c:173-183 */
void stftObj_enablePadding(STFTObj stftObj, int flag);
void stftObj_enableContinue(STFTObj stftObj, int flag);

/* frames of a dataLength-sample call under the current switches (with isContinue the tail
 * kept from the previous call counts).  replaces stft_algorithm.c:225-262 */
int stftObj_calTimeLength(STFTObj stftObj, int dataLength);
/* (timeLength-1)*slideLength + fftLength.  replaces stft_algorithm.c:289-301 */
int stftObj_calDataLength(STFTObj stftObj, int timeLength);

/* dataArr[dataLength] -> mRealArr/mImageArr [T, fftLength] (all fftLength bins).
 * replaces stft_algorithm.c:264-287, :474-803 */
void stftObj_stft(STFTObj stftObj, float *dataArr, int dataLength, float *mRealArr,
                  float *mImageArr);
/* mRealArr/mImageArr [nLength, fftLength] -> dataArr[(nLength-1)*slideLength+fftLength];
 * type 0 weighted overlap-add, 1 overlap-add.  The result is ACCUMULATED onto the caller's
 * dataArr before the normalisation, as in the reference (the wrapper passes zeros).
 * replaces stft_algorithm.c:304-409 */
void stftObj_istft(STFTObj stftObj, float *mRealArr, float *mImageArr, int nLength, int type,
                   float *dataArr);

/* NULL-safe */
void stftObj_free(STFTObj stftObj);
/* prints the frame parameters, as the reference does (stft_algorithm.c:852-864) */
void stftObj_debug(STFTObj stftObj);

#ifdef __cplusplus
}
#endif
#endif /* STFT_ALGORITHM_H */
