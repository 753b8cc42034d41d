/* afx_synsq.c -- the synchrosqueezing object (C host side) behind include/synsq_algorithm.h;
 * parameter semantics of src/synsq_algorithm.c:38-124, execution by afx_wsst.hip. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "afx_device.h"
#include "afx_host.h"
#include "synsq_algorithm.h"

struct OpaqueSynsq {
    int num, fftLength, samplate, order;
    float thresh;
    void *stream;
    float *dBuf; /* device: W re | im | phase | out re | im (5 planes) + freNorm[num] */
    int status;
};

int synsqObj_new(SynsqObj *synsqObj, int num, int radix2Exp, int *samplate, int *order, float *thresh) {
    if (!synsqObj) return -1;
    *synsqObj = NULL;
    if (radix2Exp < 1 || radix2Exp > 30) return -100;
    if (num < 1) return -1;
    int st = afxdev_ensure();
    if (st != AFX_OK) return st;
    SynsqObj o = (SynsqObj)calloc(1, sizeof(struct OpaqueSynsq));
    if (!o) return AFX_ERR_NOMEM;
    o->num = num;
    o->fftLength = 1 << radix2Exp;
    o->samplate = 32000;
    if (samplate && *samplate > 0 && *samplate < 196000) o->samplate = *samplate;
    o->order = 1;
    if (order && *order > 1) o->order = *order;
    o->thresh = 0.001f;
    if (thresh && *thresh > 1) o->thresh = *thresh;
    const size_t plane = (size_t)num * o->fftLength;
    st = afxdev_stream_create(&o->stream);
    if (st == AFX_OK) st = afxdev_malloc((void **)&o->dBuf, sizeof(float) * (5 * plane + (size_t)num));
    if (st != AFX_OK) {
        synsqObj_free(o);
        return st;
    }
    *synsqObj = o;
    return 0;
}

void synsqObj_synsq(SynsqObj o, float *freArr, SpectralFilterBankScaleType scaleType, float *mRealArr1,
                    float *mImageArr1, float *mRealArr2, float *mImageArr2) {
    AFX_ENTER(o);
    if (!o) {
        afxdev_set_error("synsqObj_synsq: NULL object");
        return;
    }
    if (!freArr || !mRealArr1 || !mImageArr1 || !mRealArr2 || !mImageArr2) return;
    if ((int)scaleType > (int)SpectralFilterBankScale_Log) {
        printf("scaleType is error!\n");
        return;
    }
    const size_t plane = (size_t)o->num * o->fftLength, pB = sizeof(float) * plane;
    float *dWr = o->dBuf, *dWi = dWr + plane, *dPh = dWi + plane, *dOr = dPh + plane, *dOi = dOr + plane;
    float *dFn = dOi + plane;
    float *fn = (float *)malloc(sizeof(float) * (size_t)o->num);
    int st = fn ? AFX_OK : AFX_ERR_NOMEM;
    for (int i = 0; i < o->num && st == AFX_OK; i++) fn[i] = freArr[i] / o->samplate;
    if (st == AFX_OK) st = afxdev_h2d(dFn, fn, sizeof(float) * (size_t)o->num, o->stream);
    if (st == AFX_OK) st = afxdev_h2d(dWr, mRealArr1, pB, o->stream);
    if (st == AFX_OK) st = afxdev_h2d(dWi, mImageArr1, pB, o->stream);
    if (st == AFX_OK) st = afxdev_h2d(dOr, mRealArr2, pB, o->stream); /* accumulate semantics (:272-273) */
    if (st == AFX_OK) st = afxdev_h2d(dOi, mImageArr2, pB, o->stream);
    if (st == AFX_OK) st = afxk_synsq_phase(dWr, dWi, o->num, o->fftLength, dPh, o->stream);
    if (st == AFX_OK) {
        AfxWsstArgs a;
        memset(&a, 0, sizeof(a));
        a.wRe = dWr;
        a.wIm = dWi;
        a.dRe = dPh;
        a.phaseInput = 1;
        a.outRe = dOr;
        a.outIm = dOi;
        a.num = o->num;
        a.batch = 1;
        a.length = o->fftLength;
        a.thresh = o->thresh;
        a.fmin = freArr[0] / o->samplate;
        a.fmax = freArr[o->num - 1] / o->samplate;
        a.logMin = log2f(a.fmin);
        a.logMax = log2f(a.fmax);
        a.freNorm = dFn;
        if (scaleType == SpectralFilterBankScale_Octave || scaleType == SpectralFilterBankScale_Log) a.mode = 0;
        else if (scaleType == SpectralFilterBankScale_Linear || scaleType == SpectralFilterBankScale_Linspace) a.mode = 1;
        else a.mode = 2;
        st = afxk_wsst_squeeze(&a, o->stream);
    }
    if (st == AFX_OK) st = afxdev_d2h(mRealArr2, dOr, pB, o->stream);
    if (st == AFX_OK) st = afxdev_d2h(mImageArr2, dOi, pB, o->stream);
    if (st == AFX_OK) st = afxdev_stream_sync(o->stream);
    free(fn);
    if (st != AFX_OK) {
        o->status = st;
        afxdev_report_failure("synsqObj_synsq", st);
    }
}

void synsqObj_free(SynsqObj o) {
    if (!o) return;
    if (o->stream) afxdev_stream_sync(o->stream);
    afxdev_free(o->dBuf);
    afxdev_stream_destroy(o->stream);
    free(o);
}
