/* afx_reassign.c -- the reassignment object (C host side) behind include/reassign_algorithm.h.
 * Parameter semantics and the three analysis windows follow src/reassign_algorithm.c:84-173,
 * :417-452; the three STFTs, the coordinates and the accumulation run on the device
 * (afx_stft.hip, afx_reassign.hip).  There is no CPU compute path.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "afx_batch.h"
#include "afx_device.h"
#include "afx_host.h"
#include "afx_objects.h"
#include "reassign_algorithm.h"

/* the three windows [3][N]: h, dh (central difference of the periodically extended window,
 * :429-437 with __vgradient), t.h with t = -N/2 .. N/2-1 (:439-441).  Exported for host tests. */
int afx_reassign_windows(WindowType type, int radix2Exp, float *out) {
    const int N = 1 << radix2Exp;
    float *w = afx_window_fft(type, N);
    if (!w) return AFX_ERR_NOMEM;
    for (int n = 0; n < N; n++) {
        out[n] = w[n];
        out[N + n] = (w[(n + 1) % N] - w[(n - 1 + N) % N]) / 2;
        out[2 * N + n] = (float)(-N / 2 + n) * w[n];
    }
    free(w);
    return AFX_OK;
}

int reassignObj_new(ReassignObj *reassignObj, int radix2Exp, int *samplate, WindowType *windowType,
                    int *slideLength, ReassignType *reType, float *thresh, int *isPadding,
                    int *isContinue) {
    (void)isContinue; /* read nowhere in the reference either */
    if (!reassignObj) return -1;
    *reassignObj = NULL;
    int r = 12;
    if (radix2Exp > 1 && radix2Exp < 31) r = radix2Exp;
    if (r > 14) {
        afxdev_set_error("reassignObj_new: fftLength 2^%d exceeds the on-chip FFT limit 2^14", r);
        return AFX_ERR_UNSUPPORTED;
    }
    int st = afxdev_ensure();
    if (st != AFX_OK) return st;
    ReassignObj o = (ReassignObj)calloc(1, sizeof(struct OpaqueReassign));
    if (!o) return AFX_ERR_NOMEM;
    o->radix2Exp = r;
    o->fftLength = 1 << r;
    o->F = o->fftLength / 2 + 1;
    o->resType = reType ? *reType : Reassign_All;
    o->samplate = (samplate && *samplate > 0) ? *samplate : 32000;
    o->isPadding = isPadding ? *isPadding : 0;
    o->windowType = windowType ? *windowType : Window_Hann;
    o->slideLength = (slideLength && *slideLength > 0) ? *slideLength : (o->fftLength / 4 > 0 ? o->fftLength / 4 : 1);
    o->thresh = (thresh && *thresh >= 0) ? *thresh : 0.001f;
    const int N = o->fftLength;
    float *win = (float *)malloc(sizeof(float) * 3 * (size_t)N);
    float *tw = afx_twiddle_table(N);
    float *fre = afx_linspace(0, (float)(o->samplate / 2.0), o->F, 0);
    if (!win || !tw || !fre) st = AFX_ERR_NOMEM;
    if (st == AFX_OK) st = afx_reassign_windows(o->windowType, r, win);
    if (st == AFX_OK) st = afxdev_stream_create(&o->stream);
    if (st == AFX_OK) st = afxdev_malloc((void **)&o->dWin, sizeof(float) * 3 * (size_t)N);
    if (st == AFX_OK) st = afxdev_malloc((void **)&o->dTwiddle, sizeof(float) * (size_t)(N < 2 ? 2 : N));
    if (st == AFX_OK) st = afxdev_malloc((void **)&o->dFre, sizeof(float) * (size_t)o->F);
    if (st == AFX_OK) st = afxdev_h2d(o->dWin, win, sizeof(float) * 3 * (size_t)N, o->stream);
    if (st == AFX_OK) st = afxdev_h2d(o->dTwiddle, tw, sizeof(float) * (size_t)(N < 2 ? 2 : N), o->stream);
    if (st == AFX_OK) st = afxdev_h2d(o->dFre, fre, sizeof(float) * (size_t)o->F, o->stream);
    if (st == AFX_OK) st = afxdev_stream_sync(o->stream);
    free(win);
    free(tw);
    free(fre);
    if (st != AFX_OK) {
        reassignObj_free(o);
        return st;
    }
    *reassignObj = o;
    return 0;
}

int reassignObj_calTimeLength(ReassignObj o, int dataLength) {
    if (!o) return 0;
    if (o->isPadding) return dataLength <= 0 ? 0 : dataLength / o->slideLength + 1;
    if (dataLength < o->fftLength) return 0;
    return (dataLength - o->fftLength) / o->slideLength + 1;
}

void reassignObj_setResultType(ReassignObj o, int type) {
    if (o) o->resultType = type;
}

void reassignObj_setOrder(ReassignObj o, int order) {
    if (o) o->order = order;
}

/* one STFT [batch][T, F] of the clips with window `which` (0 h, 1 dh, 2 t.h) */
static int stft_planes(ReassignObj o, int which, const float *dData, int batch, int validLength,
                       long long clipStride, int T, float *dRe, float *dIm, void *stream) {
    AfxStftArgs a;
    memset(&a, 0, sizeof(a));
    a.x = dData;
    a.clipStride = clipStride;
    a.batch = batch;
    a.dataLength = validLength;
    a.timeLength = T;
    a.radix2Exp = o->radix2Exp;
    a.hop = o->slideLength;
    a.window = o->dWin + (size_t)which * o->fftLength;
    a.twiddle = o->dTwiddle;
    a.mode = AFX_SPEC_COMPLEX;
    a.binLo = 0;
    a.binCount = o->F;
    a.outRe = dRe;
    a.outIm = dIm;
    if (o->isPadding) a.padLeft = o->fftLength / 2; /* centre, zeros (stftObj_enablePadding default) */
    return afxk_stft(&a, stream);
}

int reassignObj_reassignBatchDevice(ReassignObj o, const float *dData, int batch, int dataLength,
                                    long long clipStride, float *dReal1, float *dImag1, float *dReal2,
                                    float *dImag2, void *hipStream) {
    AFX_ENTER(o);
    if (!o || !dData || !dReal1 || batch <= 0 || dataLength <= 0) return AFX_ERR_ARG;
    if (!o->resultType && !dImag1) return AFX_ERR_ARG;
    const int T = reassignObj_calTimeLength(o, dataLength);
    if (T <= 0) return AFX_OK;
    int valid = dataLength;
    if (o->isPadding && T > 1) valid = dataLength - dataLength % o->slideLength; /* tail dropped */
    const size_t plane = (size_t)batch * T * o->F;
    if (o->lastStreamSet && o->lastStream != hipStream) {
        int sst = afxdev_stream_sync(o->lastStream);
        if (sst != AFX_OK) return sst;
    }
    o->lastStream = hipStream;
    o->lastStreamSet = 1;
    if (o->resType == Reassign_None) {
        /* plain STFT, written (not accumulated) into the first pair (:226-249) */
        if (!dImag1) return AFX_ERR_ARG;
        return stft_planes(o, 0, dData, batch, valid, clipStride, T, dReal1, dImag1, hipStream);
    }
    const int doFre = (o->resType == Reassign_Fre || o->resType == Reassign_All);
    const int doTime = (o->resType == Reassign_Time || o->resType == Reassign_All);
    const int own = !(dReal2 && dImag2);
    /* scratch: [h re|im (when not handed out)] [dh re|im] [th re|im] + 2 (+1) index planes */
    int st = afxdev_reserve((void **)&o->dPlanes, &o->capPlanes, sizeof(float) * plane * 6);
    if (st == AFX_OK)
        st = afxdev_reserve((void **)&o->dIdx, &o->capIdx, sizeof(int) * plane * (o->order > 1 ? 3 : 2));
    if (st != AFX_OK) return st;
    float *hRe = own ? o->dPlanes : dReal2, *hIm = own ? o->dPlanes + plane : dImag2;
    float *dhRe = o->dPlanes + 2 * plane, *dhIm = o->dPlanes + 3 * plane;
    float *thRe = o->dPlanes + 4 * plane, *thIm = o->dPlanes + 5 * plane;
    st = stft_planes(o, 0, dData, batch, valid, clipStride, T, hRe, hIm, hipStream);
    if (st == AFX_OK && doFre) st = stft_planes(o, 1, dData, batch, valid, clipStride, T, dhRe, dhIm, hipStream);
    if (st == AFX_OK && doTime) st = stft_planes(o, 2, dData, batch, valid, clipStride, T, thRe, thIm, hipStream);
    if (st != AFX_OK) return st;
    AfxReassignArgs a;
    memset(&a, 0, sizeof(a));
    a.hRe = hRe;
    a.hIm = hIm;
    a.dhRe = dhRe;
    a.dhIm = dhIm;
    a.thRe = thRe;
    a.thIm = thIm;
    a.freArr = o->dFre;
    a.timeIdx = o->dIdx;
    a.freIdx = o->dIdx + plane;
    a.outRe = dReal1;
    a.outIm = dImag1;
    a.batch = batch;
    a.timeLength = T;
    a.F = o->F;
    a.hop = o->slideLength;
    a.samplate = o->samplate;
    a.doFre = doFre;
    a.doTime = doTime;
    a.resultType = o->resultType;
    a.thresh = o->thresh;
    a.freScale = (float)(-0.5 * o->samplate / M_PI);
    a.timeScale = (float)(1.0 / o->samplate);
    return afxk_reassign(&a, o->order, o->order > 1 ? o->dIdx + 2 * plane : NULL, hipStream);
}

void reassignObj_reassign(ReassignObj o, float *dataArr, int dataLength, float *mRealArr1,
                          float *mImageArr1, float *mRealArr2, float *mImageArr2) {
    AFX_ENTER(o);
    if (!o) {
        afxdev_set_error("reassignObj_reassign: NULL object");
        return;
    }
    if (!dataArr || dataLength <= 0 || !mRealArr1) return;
    const int T = reassignObj_calTimeLength(o, dataLength);
    if (T <= 0) return;
    const int complexOut = !o->resultType || o->resType == Reassign_None;
    if (complexOut && !mImageArr1) return;
    const size_t plane = (size_t)T * o->F, pB = sizeof(float) * plane;
    int st = afxdev_reserve((void **)&o->dX, &o->capX, sizeof(float) * (size_t)dataLength);
    if (st == AFX_OK) st = afxdev_reserve((void **)&o->dOut, &o->capOut, 4 * pB);
    float *dO = o->dOut, *dS = o->dOut + 2 * plane;
    if (st == AFX_OK) st = afxdev_h2d(o->dX, dataArr, sizeof(float) * (size_t)dataLength, o->stream);
    /* the reference ADDS onto the caller's arrays (:386-391): carry their content along */
    if (st == AFX_OK) st = afxdev_h2d(dO, mRealArr1, pB, o->stream);
    if (st == AFX_OK && complexOut) st = afxdev_h2d(dO + plane, mImageArr1, pB, o->stream);
    const int wantS = (mRealArr2 || mImageArr2) && o->resType != Reassign_None;
    if (st == AFX_OK)
        st = reassignObj_reassignBatchDevice(o, o->dX, 1, dataLength, dataLength, dO, dO + plane,
                                             wantS ? dS : NULL, wantS ? dS + plane : NULL, o->stream);
    if (st == AFX_OK) st = afxdev_d2h(mRealArr1, dO, pB, o->stream);
    if (st == AFX_OK && complexOut) st = afxdev_d2h(mImageArr1, dO + plane, pB, o->stream);
    if (st == AFX_OK && wantS && mRealArr2) st = afxdev_d2h(mRealArr2, dS, pB, o->stream);
    if (st == AFX_OK && wantS && mImageArr2) st = afxdev_d2h(mImageArr2, dS + plane, pB, o->stream);
    if (st == AFX_OK) st = afxdev_stream_sync(o->stream);
    if (st != AFX_OK) {
        o->status = st;
        afxdev_report_failure("reassignObj_reassign", st);
    }
}

void reassignObj_free(ReassignObj o) {
    if (!o) return;
    if (o->stream) afxdev_stream_sync(o->stream);
    afxdev_free(o->dWin);
    afxdev_free(o->dTwiddle);
    afxdev_free(o->dFre);
    afxdev_free(o->dPlanes);
    afxdev_free(o->dIdx);
    afxdev_free(o->dX);
    afxdev_free(o->dOut);
    afxdev_stream_destroy(o->stream);
    free(o);
}
