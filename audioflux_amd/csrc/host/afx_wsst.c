/* afx_wsst.c -- the wavelet synchrosqueezed transform object (C host side) behind
 * include/wsst_algorithm.h.  Parameter semantics follow wsstObj_new (src/wsst_algorithm.c:64-152);
 * the object owns a CWT object with the derivative bank enabled, keeps W and W' in HBM and runs
 * the squeezing pass there (afx_wsst.hip).  There is no CPU compute path.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "afx_batch.h"
#include "afx_device.h"
#include "afx_host.h"
#include "wsst_algorithm.h"

struct OpaqueWSST {
    CWTObj cwt;
    int num, fftLength, samplate, order;
    float thresh;
    WaveletContinueType waveletType;
    SpectralFilterBankScaleType scaleType;
    void *stream;
    float *dFreNorm;          /* device [num]: band centres / samplate */
    float *dX, *dW, *dOut;    /* grow-only device scratch: signal, W|W' (4 planes), result (2 planes) */
    size_t capX, capW, capOut;
    int status;
};

int wsstObj_new(WSSTObj *wsstObj, int num, int radix2Exp, int *samplate, float *lowFre, float *highFre,
                int *binPerOctave, WaveletContinueType *waveletType,
                SpectralFilterBankScaleType *scaleType, float *gamma, float *beta, float *thresh,
                int *isPadding) {
    float th = 0.001f;
    int sr = 32000;
    WaveletContinueType wt = WaveletContinue_Morlet;
    SpectralFilterBankScaleType sc = SpectralFilterBankScale_Octave;
    if (!wsstObj) return -1;
    *wsstObj = NULL;
    if (thresh && *thresh >= 0) th = *thresh;
    if (samplate && *samplate > 0 && *samplate <= 196000) sr = *samplate;
    if (waveletType) wt = *waveletType;
    if (scaleType) {
        sc = *scaleType;
        if ((int)sc > (int)SpectralFilterBankScale_Log) {
            printf("scaleType is error!\n");
            return 1;
        }
    }
    WSSTObj o = (WSSTObj)calloc(1, sizeof(struct OpaqueWSST));
    if (!o) return AFX_ERR_NOMEM;
    int st = cwtObj_new(&o->cwt, num, radix2Exp, samplate, lowFre, highFre, binPerOctave, &wt, &sc, gamma, beta,
                        isPadding);
    if (st == 0) {
        cwtObj_enableDet(o->cwt, 1);
        o->num = num;
        o->fftLength = 1 << radix2Exp;
        o->samplate = sr;
        o->thresh = th;
        o->waveletType = wt;
        o->scaleType = sc;
        st = afxdev_stream_create(&o->stream);
    }
    if (st == 0) {
        float *fn = (float *)malloc(sizeof(float) * (size_t)num);
        const float *fre = cwtObj_getFreBandArr(o->cwt);
        if (!fn) st = AFX_ERR_NOMEM;
        for (int i = 0; i < num && st == 0; i++) fn[i] = fre[i] / sr;
        if (st == 0) st = afxdev_malloc((void **)&o->dFreNorm, sizeof(float) * (size_t)num);
        if (st == 0) st = afxdev_h2d(o->dFreNorm, fn, sizeof(float) * (size_t)num, o->stream);
        if (st == 0) st = afxdev_stream_sync(o->stream);
        free(fn);
    }
    if (st != 0) {
        wsstObj_free(o);
        return st;
    }
    *wsstObj = o;
    return 0;
}

float *wsstObj_getFreBandArr(WSSTObj o) { return o ? cwtObj_getFreBandArr(o->cwt) : NULL; }
int *wsstObj_getBinBandArr(WSSTObj o) { return o ? cwtObj_getBinBandArr(o->cwt) : NULL; }

void wsstObj_setOrder(WSSTObj o, int order) {
    if (!o) return;
    o->order = order;
    if (order > 1)
        fprintf(stderr, "[audioflux_mi355x] wsstObj_setOrder(%d): higher-order squeezing is not run "
                        "(the reference crashes in that branch); order 1 is used\n", order);
}

int wsstObj_wsstBatchDevice(WSSTObj o, const float *dData, int chunks, long long chunkStride, float *dReal1,
                            float *dImag1, float *dReal2, float *dImag2, void *hipStream) {
    AFX_ENTER(o);
    if (!o || !dData || !dReal1 || !dImag1 || chunks <= 0) return AFX_ERR_ARG;
    const size_t plane = (size_t)o->num * o->fftLength * (size_t)chunks;
    /* W goes straight to the caller's second pair when it is wanted, else to scratch */
    const int own = !(dReal2 && dImag2);
    int st = afxdev_reserve((void **)&o->dW, &o->capW, sizeof(float) * plane * (own ? 4 : 2));
    if (st != AFX_OK) return st;
    float *wRe = own ? o->dW + 2 * plane : dReal2, *wIm = own ? o->dW + 3 * plane : dImag2;
    float *dRe = o->dW, *dIm = o->dW + plane;
    st = cwtObj_cwtBatchDevice(o->cwt, dData, chunks, chunkStride, wRe, wIm, hipStream);
    if (st == AFX_OK) st = cwtObj_cwtDetBatchDevice(o->cwt, dData, chunks, chunkStride, dRe, dIm, hipStream);
    if (st != AFX_OK) return st;
    const float *fre = cwtObj_getFreBandArr(o->cwt);
    AfxWsstArgs a;
    memset(&a, 0, sizeof(a));
    a.wRe = wRe;
    a.wIm = wIm;
    a.dRe = dRe;
    a.dIm = dIm;
    a.outRe = dReal1;
    a.outIm = dImag1;
    a.num = o->num;
    a.batch = chunks;
    a.length = o->fftLength;
    a.thresh = o->thresh;
    a.fmin = fre[0] / o->samplate;
    a.fmax = fre[o->num - 1] / o->samplate;
    a.logMin = log2f(a.fmin);
    a.logMax = log2f(a.fmax);
    a.freNorm = o->dFreNorm;
    if (o->scaleType == SpectralFilterBankScale_Octave || o->scaleType == SpectralFilterBankScale_Log) a.mode = 0;
    else if (o->scaleType == SpectralFilterBankScale_Linear || o->scaleType == SpectralFilterBankScale_Linspace) a.mode = 1;
    else a.mode = 2;
    return afxk_wsst_squeeze(&a, hipStream);
}

void wsstObj_wsst(WSSTObj o, float *dataArr, float *mRealArr1, float *mImageArr1, float *mRealArr2,
                  float *mImageArr2) {
    AFX_ENTER(o);
    if (!o) {
        afxdev_set_error("wsstObj_wsst: NULL object");
        return;
    }
    if (!dataArr || !mRealArr1 || !mImageArr1) return;
    const size_t plane = (size_t)o->num * o->fftLength, pB = sizeof(float) * plane;
    int st = afxdev_reserve((void **)&o->dX, &o->capX, sizeof(float) * (size_t)o->fftLength);
    if (st == AFX_OK) st = afxdev_reserve((void **)&o->dOut, &o->capOut, 4 * pB);
    float *dO = o->dOut, *dC = o->dOut + 2 * plane;
    if (st == AFX_OK) st = afxdev_h2d(o->dX, dataArr, sizeof(float) * (size_t)o->fftLength, o->stream);
    /* the reference ADDS onto the caller's arrays (:332-333): carry their content along */
    if (st == AFX_OK) st = afxdev_h2d(dO, mRealArr1, pB, o->stream);
    if (st == AFX_OK) st = afxdev_h2d(dO + plane, mImageArr1, pB, o->stream);
    if (st == AFX_OK)
        st = wsstObj_wsstBatchDevice(o, o->dX, 1, o->fftLength, dO, dO + plane, dC, dC + plane, o->stream);
    if (st == AFX_OK) st = afxdev_d2h(mRealArr1, dO, pB, o->stream);
    if (st == AFX_OK) st = afxdev_d2h(mImageArr1, dO + plane, pB, o->stream);
    if (st == AFX_OK && mRealArr2) st = afxdev_d2h(mRealArr2, dC, pB, o->stream);
    if (st == AFX_OK && mImageArr2) st = afxdev_d2h(mImageArr2, dC + plane, pB, o->stream);
    if (st == AFX_OK) st = afxdev_stream_sync(o->stream);
    if (st != AFX_OK) {
        o->status = st;
        afxdev_report_failure("wsstObj_wsst", st);
    }
}

void wsstObj_free(WSSTObj o) {
    if (!o) return;
    if (o->stream) afxdev_stream_sync(o->stream);
    afxdev_free(o->dFreNorm);
    afxdev_free(o->dX);
    afxdev_free(o->dW);
    afxdev_free(o->dOut);
    afxdev_stream_destroy(o->stream);
    cwtObj_free(o->cwt);
    free(o);
}
