/* afx_xxcc.c -- the cepstral-coefficient object (C host side) behind
 * include/feature/xxcc_algorithm.h.
 *
 * Parameter semantics follow src/feature/xxcc_algorithm.c:32-330.  The
 * reference runs, per frame, log10/cube-root then a DCT-II (through its FFT
 * when num is a power of two, else a cosine matrix) and keeps the first ccNum
 * outputs; here the rectification is fused into the operand load of one MFMA
 * GEMM against the first ccNum rows of the orthonormal DCT-II matrix.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "afx_batch.h"
#include "afx_device.h"
#include "afx_host.h"
#include "afx_objects.h"
#include "feature/xxcc_algorithm.h"

int xxccObj_new(XXCCObj *xxccObj, int num) {
    if (!xxccObj) return -1;
    *xxccObj = NULL;
    if (num < 2) {
        printf("num is error!!!\n");
        return -1;
    }
    if (num > 16384) { /* the [num, num] DCT matrix is built here: no filter bank of the path has more rows than
                        * fftLength / 2 + 1 <= 8193; a larger num would sit in cos() for minutes and ask for > 1 GB */
        afxdev_set_error("xxccObj_new: num %d exceeds 16384", num);
        return AFX_ERR_UNSUPPORTED;
    }
    int st = afxdev_ensure();
    if (st != AFX_OK) return st;
    XXCCObj o = (XXCCObj)calloc(1, sizeof(struct OpaqueXXCC));
    if (!o) return AFX_ERR_NOMEM;
    o->num = num;
    float *d = afx_dct2_matrix(num, num);
    if (!d) st = AFX_ERR_NOMEM;
    if (st == AFX_OK) st = afxdev_stream_create(&o->stream);
    if (st == AFX_OK) st = afxdev_malloc((void **)&o->dDct, sizeof(float) * (size_t)num * num);
    if (st == AFX_OK) st = afxdev_h2d(o->dDct, d, sizeof(float) * (size_t)num * num, o->stream);
    if (st == AFX_OK) st = afxdev_stream_sync(o->stream);
    free(d);
    if (st != AFX_OK) {
        xxccObj_free(o);
        return st;
    }
    *xxccObj = o;
    return 0;
}

void xxccObj_setTimeLength(XXCCObj o, int timeLength) {
    if (o) o->timeLength = timeLength;
}

static int rectify_to_map(const CepstralRectifyType *rectifyType) {
    if (rectifyType && *rectifyType == CepstralRectify_CubicRoot) return AFX_MAP_CBRT;
    return AFX_MAP_LOG10;
}

int xxccObj_xxccDevice(XXCCObj o, const float *dIn, long long rows, int ccNum,
                       CepstralRectifyType *rectifyType, float *dOut, void *hipStream) {
    AFX_ENTER(o);
    if (!o || !dIn || !dOut) return AFX_ERR_ARG;
    if (ccNum > o->num || ccNum < 1) return AFX_ERR_ARG;
    if (afxk_cepstrum_supported(dIn, o->num, ccNum) && !afxdev_no_fused())
        return afxk_cepstrum(dIn, rows, o->num, o->dDct, ccNum, rectify_to_map(rectifyType), dOut,
                             hipStream);
    return afxk_gemm_nt(dIn, o->num, o->dDct, o->num, dOut, ccNum, rows, ccNum, o->num,
                        rectify_to_map(rectifyType), AFX_MAP_NONE, 1.f, hipStream);
}

/* upload [T,num], run the rectify+DCT GEMM into dOut[T,ccNum] */
static int run_cc(XXCCObj o, const float *hIn, int ccNum, CepstralRectifyType *rectifyType) {
    const size_t rows = (size_t)o->timeLength;
    int st = afxdev_reserve((void **)&o->dIn, &o->capIn, sizeof(float) * rows * o->num);
    if (st == AFX_OK)
        st = afxdev_reserve((void **)&o->dOut, &o->capOut, sizeof(float) * rows * (ccNum + 1) * 4);
    if (st == AFX_OK) st = afxdev_h2d(o->dIn, hIn, sizeof(float) * rows * o->num, o->stream);
    if (st == AFX_OK)
        st = xxccObj_xxccDevice(o, o->dIn, (long long)rows, ccNum, rectifyType, o->dOut, o->stream);
    return st;
}

void xxccObj_xxcc(XXCCObj o, float *mDataArr1, int ccNum, CepstralRectifyType *rectifyType,
                  float *mDataArr2) {
    AFX_ENTER(o);
    if (!o) {
        afxdev_set_error("xxccObj_xxcc: NULL object");
        return;
    }
    if (ccNum > o->num || ccNum < 1 || o->timeLength <= 0 || !mDataArr1 || !mDataArr2) return;
    int st = run_cc(o, mDataArr1, ccNum, rectifyType);
    if (st == AFX_OK)
        st = afxdev_d2h(mDataArr2, o->dOut, sizeof(float) * (size_t)o->timeLength * ccNum, o->stream);
    if (st == AFX_OK) st = afxdev_stream_sync(o->stream);
    if (st != AFX_OK) {
        o->status = st;
        afxdev_report_failure("xxccObj_xxcc", st);
    }
}

/* host pointers, explicit row count: mDataArr1[rows,num] -> mDataArr2[rows,ccNum]; the same
 * as xxccObj_setTimeLength(rows) + xxccObj_xxcc, with a status (include/afx_batch.h) */
int xxccObj_xxccBatch(XXCCObj o, const float *mDataArr1, long long rows, int ccNum,
                      CepstralRectifyType *rectifyType, float *mDataArr2) {
    AFX_ENTER(o);
    if (!o || !mDataArr1 || !mDataArr2 || rows <= 0 || rows > 2147483647LL || ccNum > o->num || ccNum < 1) {
        afxdev_set_error("xxccObj_xxccBatch: bad argument");
        return AFX_ERR_ARG;
    }
    const int keep = o->timeLength;
    o->timeLength = (int)rows;
    int st = run_cc(o, mDataArr1, ccNum, rectifyType);
    if (st == AFX_OK)
        st = afxdev_d2h(mDataArr2, o->dOut, sizeof(float) * (size_t)rows * ccNum, o->stream);
    if (st == AFX_OK) st = afxdev_stream_sync(o->stream);
    o->timeLength = keep;
    if (st != AFX_OK) {
        o->status = st;
        afxdev_report_failure("xxccObj_xxccBatch", st);
    }
    return st;
}

void xxccObj_xxccStandard(XXCCObj o, float *mDataArr1, int ccNum, float *energyArr,
                          int *deltaWindowLength, CepstralEnergyType *energyType,
                          CepstralRectifyType *rectifyType, float *mCoeArr, float *mDeltaArr1,
                          float *mDeltaArr2) {
    AFX_ENTER(o);
    if (!o) {
        afxdev_set_error("xxccObj_xxccStandard: NULL object");
        return;
    }
    if (ccNum > o->num || ccNum < 1 || o->timeLength <= 0 || !mDataArr1) return;
    int dLen = 9;
    CepstralEnergyType eType = CepstralEnergy_Replace;
    if (deltaWindowLength && *deltaWindowLength >= 3 && (*deltaWindowLength) % 2 == 1)
        dLen = *deltaWindowLength;
    if (energyType) eType = *energyType;
    if (eType != CepstralEnergy_Ignore && !energyArr) return;
    const size_t rows = (size_t)o->timeLength;
    const int outLen = ccNum + (eType == CepstralEnergy_Append ? 1 : 0);

    /* dOut layout: [cc rows*ccNum | energy rows | coe | d1 | d2 (rows*outLen each)] */
    int st = run_cc(o, mDataArr1, ccNum, rectifyType);
    float *dCc = o->dOut;
    float *dEnergy = dCc + rows * ccNum;
    float *dCoe = dEnergy + rows;
    float *dD1 = dCoe + rows * outLen;
    float *dD2 = dD1 + rows * outLen;
    if (st == AFX_OK && eType != CepstralEnergy_Ignore)
        st = afxdev_h2d(dEnergy, energyArr, sizeof(float) * rows, o->stream);
    if (st == AFX_OK)
        st = afxk_xxcc_standard(dCc, dEnergy, (long long)rows, ccNum, (int)eType, dLen, dCoe, dD1,
                                dD2, o->stream);
    const size_t outBytes = sizeof(float) * rows * outLen;
    if (st == AFX_OK && mCoeArr) st = afxdev_d2h(mCoeArr, dCoe, outBytes, o->stream);
    if (st == AFX_OK && mDeltaArr1) st = afxdev_d2h(mDeltaArr1, dD1, outBytes, o->stream);
    if (st == AFX_OK && mDeltaArr2) st = afxdev_d2h(mDeltaArr2, dD2, outBytes, o->stream);
    if (st == AFX_OK) st = afxdev_stream_sync(o->stream);
    if (st != AFX_OK) {
        o->status = st;
        afxdev_report_failure("xxccObj_xxccStandard", st);
    }
}

void xxccObj_free(XXCCObj o) {
    if (!o) return;
    if (o->stream) afxdev_stream_sync(o->stream);
    afxdev_free(o->dDct);
    afxdev_free(o->dIn);
    afxdev_free(o->dOut);
    afxdev_stream_destroy(o->stream);
    free(o);
}
