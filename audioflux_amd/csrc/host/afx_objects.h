/* afx_objects.h -- layouts of the opaque handles (library-internal). */
#ifndef AFX_OBJECTS_H
#define AFX_OBJECTS_H

#include <stddef.h>

#include "flux_base.h"
#include "reassign_algorithm.h"

#ifdef __cplusplus
extern "C" {
#endif

struct AfxMelFusedPlan; /* afx_melfused.hip */

struct OpaqueBFT {
    /* plan */
    int fftLength, radix2Exp, F, num;
    int samplate;
    float lowFre, highFre;
    int lowIndex, highIndex; /* linear scale: first/last bin */
    int binPerOctave;
    WindowType windowType;
    int slideLength;
    SpectralDataType dataType;
    SpectralFilterBankScaleType scale;
    SpectralFilterBankStyleType style;
    SpectralFilterBankNormalType normal;
    int resultType;  /* 0 complex, 1 real */
    float normValue; /* default 1 */
    int isTemporal;
    float *freBandArr; /* host, num+2 */
    int *binBandArr;
    /* device constants */
    void *stream;
    float *dWindow, *dTwiddle, *dBank;
    int bankPitch; /* floats per row of dBank: F rounded up to 4 (zero padded) so that the MFMA GEMM
                    * loads 16-byte aligned rows */
    /* banded (row span) view of the bank for the in-kernel filter-bank epilogue of the
     * size-generic STFT kernel; NULL when the bank is too dense for it */
    int *dBandMeta;   /* [3][num]: start, len, offset */
    float *dBandW;
    struct AfxMelFusedPlan *fast; /* NULL when the fused kernel does not apply */
    struct OpaqueReassign *reassign; /* isReassign = 1: the reassigned spectrum replaces the STFT */
    /* grow-only device scratch of the legacy host-pointer calls */
    void *dBankImage; /* dense banks: dBank as three bf16 word planes in the GEMM's staging order (afxk_gemm_bank_prepare),
                       * built on the first dense call; bankImageTried: the stand-in / a failure left none -- float bank then */
    int bankImageTried;
    float *dX, *dSpec, *dOut, *dTemporal;
    size_t capX, capSpec, capOut, capTemporal;
    /* host copies for bftObj_getTemporalData */
    float *hTemporal;
    int hTemporalCap, hTemporalFrames;
    int lastTimeLength;
    int status; /* last failure of a void entry point */
    void *lastStream; /* stream of the previous launch (scratch is shared) */
    int lastStreamSet;
};

struct OpaqueXXCC {
    int num;
    int timeLength;
    void *stream;
    float *dDct; /* device [num, num] orthonormal DCT-II */
    float *dIn, *dOut;
    size_t capIn, capOut;
    int status;
};

struct OpaqueSTFT {
    int fftLength, radix2Exp, slideLength;
    WindowType windowType;
    int isContinue, isPad;
    PaddingPositionType positionType;
    PaddingModeType modeType;
    float padValue1, padValue2;
    float *windowDataArr; /* host [fftLength]; stftObj_useWindowDataArr overwrites it */
    int windowDirty;      /* dWindow is stale */
    float *tailDataArr;   /* host [fftLength]: samples carried to the next streaming call */
    int tailDataLength;   /* may be negative (hop > fftLength: samples still to skip) */
    int timeLength;       /* frames of the last stft call */
    int methodType;       /* inverse: 0 weighted overlap-add, 1 overlap-add, -1 not built */
    float *winArr1, *winArr2; /* host: window^e, window^(e+1) */
    void *stream;
    float *dWindow, *dTwiddle, *dWin12;
    float *dX, *dOut, *dFrames; /* grow-only device scratch */
    size_t capX, capOut, capFrames;
    void *lastStream;
    int lastStreamSet;
    int status;
};

struct OpaqueReassign {
    int radix2Exp, fftLength, F, samplate, slideLength, isPadding;
    WindowType windowType;
    ReassignType resType;
    float thresh;
    int resultType, order;
    void *stream;
    float *dWin;      /* device [3][N]: h, dh, t.h */
    float *dTwiddle, *dFre;
    float *dPlanes, *dX, *dOut; /* grow-only scratch */
    int *dIdx;
    size_t capPlanes, capIdx, capX, capOut;
    void *lastStream;
    int lastStreamSet, status;
};

/* framing state machine of a legacy stftObj_stft call, host fields of the object only
 * (afx_stft.c; also used by the spectrogram object for its isContinue mode) */
int afx_stft_deal_data(struct OpaqueSTFT *o, const float *dataArr, int dataLength, int *valid,
                       int *headTail, int *skip);
void afx_stft_keep_tail(struct OpaqueSTFT *o, const float *dataArr, int dataLength, int total);

/* validated parameters of a BFT execution plan (afx_bft.c) */
typedef struct {
    int num, radix2Exp, samplate;
    float lowFre, highFre;
    int lowIndex, highIndex, binPerOctave;
    WindowType windowType;
    int slideLength;
    SpectralDataType dataType;
    SpectralFilterBankScaleType scale;
    SpectralFilterBankStyleType style;
    SpectralFilterBankNormalType normal;
    int isTemporal, isReassign;
    const float *customBank; /* optional host [num, fftLength/2+1] matrix used instead of the
                              * auditory bank of `scale` (band arrays are then left zero) */
} AfxBftPlan;
int afx_bft_create(const AfxBftPlan *p, struct OpaqueBFT **bftObj);

/* fused-kernel hooks (afx_melfused.hip) */
int afx_bft_plan_fast(struct OpaqueBFT *o, const float *hWindow, const float *hBank);
int afx_bft_try_fast(struct OpaqueBFT *o, const float *dData, int batch, int dataLength,
                     long long clipStride, float *dRe, float *dIm, float *dTemporal, void *stream,
                     int *used);
int afx_bft_try_fast_cc(struct OpaqueBFT *o, struct OpaqueXXCC *x, const float *dData, int batch,
                        int dataLength, long long clipStride, int ccNum,
                        CepstralRectifyType *rectifyType, float *dMel, float *dCc, void *stream,
                        int *used);
void afx_bft_free_fast(struct OpaqueBFT *o);

int afx_bft_run_device(struct OpaqueBFT *o, const float *dData, int batch, int dataLength,
                       long long clipStride, float *dRe, float *dIm, float *dTemporal,
                       void *stream);

#ifdef __cplusplus
}
#endif
#endif /* AFX_OBJECTS_H */
