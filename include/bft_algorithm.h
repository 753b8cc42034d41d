/* bft_algorithm.h -- C ABI of the BFT ("based Fourier transform") object:
 * framed STFT -> power|magnitude -> mel/bark/erb/... filter bank (or linear
 * bin slice), computed on an MI355X by hand-written gfx950 kernels.
 *
 * Every entry point replaces the reference function of the same name
 * (src/bft_algorithm.h:33-57 / src/bft_algorithm.c:87-626) so that the
 * audioFlux ctypes wrapper (python/audioflux/bft.py:142-389) binds to this
 * library unchanged.  Pointer-typed optional parameters follow the reference
 * convention: NULL selects the default.
 */
#ifndef BFT_ALGORITHM_H
#define BFT_ALGORITHM_H

#include "flux_base.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OpaqueBFT *BFTObj;

/* Build a BFT plan.  replaces bftObj_new, src/bft_algorithm.c:87-276.
 *   num           number of output bands, 2 .. fftLength/2+1
 *   radix2Exp     fftLength = 2^radix2Exp, 1..30 (0 -> 12)
 *   samplate      default 32000
 *   lowFre        default 0 (octave/log scales: note A0-ish, 440*2^(-45/12))
 *   highFre       default samplate/2
 *   binPerOctave  default 12, accepted range 4..48
 *   windowType    default Window_Hann
 *   slideLength   hop, default fftLength/4
 *   filterScaleType/StyleType/NormalType  defaults Linear / Slaney / None
 *   dataType      default SpectralData_Power
 *   isReassign    default 0; 1: the time-frequency reassigned spectrum replaces the STFT
 *                 (reassign_algorithm.c:203-414; ordered, deterministic accumulation); with radix2Exp 1
 *                 refused (-4): the reference's reassignment object then runs at 2^12 and overruns
 *   isTemporal    default 0; 1 also computes per-frame energy/rms/zcr (in the same kernel
 *                 launch at n_fft 2048 with real results)
 * returns 0 ok, -100 bad radix2Exp, 1 bad scale type, -1 bad num/frequency
 *         range, <= -2 backend/HIP failure (never leaves *bftObj dangling). */
int bftObj_new(BFTObj *bftObj, int num, int radix2Exp,
               int *samplate, float *lowFre, float *highFre, int *binPerOctave,
               WindowType *windowType, int *slideLength,
               SpectralFilterBankScaleType *filterScaleType,
               SpectralFilterBankStyleType *filterStyleType,
               SpectralFilterBankNormalType *filterNormalType,
               SpectralDataType *dataType,
               int *isReassign,
               int *isTemporal);

/* frames produced for dataLength samples: (dataLength-fftLength)/hop+1, or 0.
 * replaces bftObj_calTimeLength, src/bft_algorithm.c:550-555 */
int bftObj_calTimeLength(BFTObj bftObj, int dataLength);

/* library-owned arrays, >= num valid entries (centre frequency / FFT bin of
 * each band).  replaces src/bft_algorithm.c:557-565 */
float *bftObj_getFreBandArr(BFTObj bftObj);
int *bftObj_getBinBandArr(BFTObj bftObj);

/* type 0: complex result (S or S^2 through the bank) into both outputs;
 * type 1: real result (|S|^2 or |S| through the bank) into mRealArr3 only.
 * replaces src/bft_algorithm.c:567-571 */
void bftObj_setResultType(BFTObj bftObj, int type);
/* exponent applied to the power (before the bank) or to the magnitude result
 * (after the bank); ignored unless > 0.  replaces src/bft_algorithm.c:573-578 */
void bftObj_setDataNormValue(BFTObj bftObj, float normValue);

/* One clip, host pointers, synchronous.  dataArr[dataLength] ->
 * mRealArr3[T*num] (+ mImageArr3[T*num] when result type is 0).
 * replaces bftObj_bft, src/bft_algorithm.c:397-540 */
void bftObj_bft(BFTObj bftObj, float *dataArr, int dataLength,
                float *mRealArr3, float *mImageArr3);

/* borrowed per-frame energy / rms / zero-cross-rate arrays of the last
 * bftObj_bft call (isTemporal objects only).  replaces src/bft_algorithm.c:543-548 */
void bftObj_getTemporalData(BFTObj bftObj, float **eArr, float **rArr, float **zArr);

/* NULL-safe.  replaces src/bft_algorithm.c:580-626 */
void bftObj_free(BFTObj bftObj);

#ifdef __cplusplus
}
#endif
#endif /* BFT_ALGORITHM_H */
