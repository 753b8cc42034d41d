/* afx_cepstrogram.c -- the cepstrogram object (C host side) behind
 * include/cepstrogram_algorithm.h; parameter semantics of
 * src/cepstrogram_algorithm.c:55-125, execution by afx_cepstrogram.hip. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "afx_device.h"
#include "afx_host.h"
#include "cepstrogram_algorithm.h"

struct OpaqueCepstrogram {
    int fftLength, radix2Exp, slideLength;
    WindowType windowType;
    void *stream;
    float *dWindow, *dTwiddle;
    float *dFastTab; /* twiddle tables of the N = 512 / 1024 / 2048 / 4096 wave kernels (NULL for other sizes) */
    float *dX, *dOut, *dSpec; /* grow-only scratch */
    size_t capX, capOut, capSpec;
    int cachedTime; /* frames of the spectrum kept in dSpec (for cepstrogram2) */
    int isDebug;
    int status;
};

int cepstrogramObj_new(CepstrogramObj *cepstrogramObj, int radix2Exp, WindowType *windowType,
                       int *slideLength) {
    if (!cepstrogramObj) return -1;
    *cepstrogramObj = NULL;
    if (radix2Exp < 1 || radix2Exp > 30) {
        printf("radix2Exp is error!\n");
        return -100;
    }
    if (radix2Exp > 13) {
        afxdev_set_error("cepstrogramObj_new: fftLength 2^%d exceeds the on-chip limit 2^13", radix2Exp);
        return AFX_ERR_UNSUPPORTED;
    }
    int st = afxdev_ensure();
    if (st != AFX_OK) return st;
    CepstrogramObj o = (CepstrogramObj)calloc(1, sizeof(struct OpaqueCepstrogram));
    if (!o) return AFX_ERR_NOMEM;
    o->radix2Exp = radix2Exp;
    o->fftLength = 1 << radix2Exp;
    o->windowType = windowType ? *windowType : Window_Rect;
    o->slideLength = o->fftLength / 4 > 0 ? o->fftLength / 4 : 1; /* fftLength 2: the reference divides by its default of 0 */
    if (slideLength && *slideLength > 0) o->slideLength = *slideLength;
    float *w = afx_window_fft(o->windowType, o->fftLength);
    float *tw = afx_twiddle_table(o->fftLength);
    if (!w || !tw) st = AFX_ERR_NOMEM;
    const size_t nb = sizeof(float) * (size_t)o->fftLength;
    if (st == AFX_OK) st = afxdev_stream_create(&o->stream);
    if (st == AFX_OK) st = afxdev_malloc((void **)&o->dWindow, nb);
    if (st == AFX_OK) st = afxdev_malloc((void **)&o->dTwiddle, nb < 8 ? 8 : nb);
    if (st == AFX_OK) st = afxdev_h2d(o->dWindow, w, nb, o->stream);
    if (st == AFX_OK) st = afxdev_h2d(o->dTwiddle, tw, nb < 8 ? 8 : nb, o->stream);
    if (st == AFX_OK && (o->fftLength == 512 || o->fftLength == 1024 || o->fftLength == 2048 || o->fftLength == 4096)) {
        float *ft = (float *)calloc(AFX_CEPSTROGRAM_FASTTAB_FLOATS, sizeof(float));
        if (!ft) st = AFX_ERR_NOMEM;
        if (st == AFX_OK) {
            afxk_cepstrogram_fast_tables(ft, o->fftLength);
            st = afxdev_malloc((void **)&o->dFastTab, sizeof(float) * AFX_CEPSTROGRAM_FASTTAB_FLOATS);
        }
        if (st == AFX_OK) st = afxdev_h2d(o->dFastTab, ft, sizeof(float) * AFX_CEPSTROGRAM_FASTTAB_FLOATS, o->stream);
        if (st == AFX_OK) st = afxdev_stream_sync(o->stream); /* ft is freed below */
        free(ft);
    }
    if (st == AFX_OK) st = afxdev_stream_sync(o->stream);
    free(w);
    free(tw);
    if (st != AFX_OK) {
        cepstrogramObj_free(o);
        return st;
    }
    *cepstrogramObj = o;
    return 0;
}

int cepstrogramObj_calTimeLength(CepstrogramObj o, int dataLength) {
    if (!o || dataLength < o->fftLength) return 0;
    return (dataLength - o->fftLength) / o->slideLength + 1;
}

static void run(CepstrogramObj o, int cepNum, const float *hData, int dataLength, int T,
                float *hSpecRe, float *hSpecIm, float *m1, float *m2, float *m3, const char *who) {
    const int N = o->fftLength, F = N / 2 + 1;
    const size_t outB = sizeof(float) * (size_t)T * F;
    int st = afxdev_reserve((void **)&o->dOut, &o->capOut, 3 * outB);
    if (st == AFX_OK && hData)
        st = afxdev_reserve((void **)&o->dX, &o->capX, sizeof(float) * (size_t)dataLength);
    const size_t specB = sizeof(float) * (size_t)T * N;
    if (st == AFX_OK && (hData || o->capSpec < 2 * specB)) {
        /* fresh or regrown cache starts zeroed, like the reference's calloc'ed scratch */
        const size_t before = o->capSpec;
        st = afxdev_reserve((void **)&o->dSpec, &o->capSpec, 2 * specB);
        if (st == AFX_OK && o->capSpec != before) {
            st = afxdev_memset(o->dSpec, 0, o->capSpec, o->stream);
            o->cachedTime = 0;
        }
    }
    if (st == AFX_OK && cepNum < 0) cepNum = 0;
    AfxCepstrogramArgs a;
    memset(&a, 0, sizeof(a));
    a.timeLength = T;
    a.radix2Exp = o->radix2Exp;
    a.hop = o->slideLength;
    a.cepNum = cepNum;
    a.window = o->dWindow;
    a.twiddle = o->dTwiddle;
    a.fastTab = o->dFastTab;
    a.specRe = o->dSpec;
    a.specIm = o->dSpec ? o->dSpec + (size_t)T * N : NULL;
    a.out1 = m1 ? o->dOut : NULL;
    a.out2 = m2 ? o->dOut + (size_t)T * F : NULL;
    a.out3 = m3 ? o->dOut + 2 * (size_t)T * F : NULL;
    if (st == AFX_OK && hData) {
        st = afxdev_h2d(o->dX, hData, sizeof(float) * (size_t)dataLength, o->stream);
        a.x = o->dX;
    }
    if (st == AFX_OK && !hData) {
        /* cepstrogram2: the reference copies its cached spectrum OUT to the caller's arrays
         * (cepstrogram_algorithm.c:214-215) and transforms the cache */
        st = afxdev_d2h(hSpecRe, a.specRe, specB, o->stream);
        if (st == AFX_OK) st = afxdev_d2h(hSpecIm, a.specIm, specB, o->stream);
    }
    if (st == AFX_OK) st = afxk_cepstrogram(&a, o->stream);
    if (st == AFX_OK && m1) st = afxdev_d2h(m1, a.out1, outB, o->stream);
    if (st == AFX_OK && m2) st = afxdev_d2h(m2, a.out2, outB, o->stream);
    if (st == AFX_OK && m3) st = afxdev_d2h(m3, a.out3, outB, o->stream);
    if (st == AFX_OK) st = afxdev_stream_sync(o->stream);
    if (st == AFX_OK && hData) o->cachedTime = T;
    if (st != AFX_OK) {
        o->status = st;
        afxdev_report_failure(who, st);
    }
}

/* clips already in HBM -> dOut1/2/3 [batch][T, N/2+1] left in HBM (include/afx_batch.h);
 * any output may be NULL.  The spectrum cache of cepstrogram2 is not touched. */
int cepstrogramObj_cepstrogramBatchDevice(CepstrogramObj o, int cepNum, const float *dData,
                                          int batch, int dataLength, long long clipStride,
                                          float *dOut1, float *dOut2, float *dOut3,
                                          void *hipStream) {
    AFX_ENTER(o);
    if (!o || !dData || batch <= 0 || dataLength <= 0 || clipStride < 0) {
        afxdev_set_error("cepstrogramObj_cepstrogramBatchDevice: bad argument");
        return AFX_ERR_ARG;
    }
    const int T = cepstrogramObj_calTimeLength(o, dataLength);
    if (T <= 0) return AFX_OK;
    if ((long long)T * batch > 2147483647LL) {
        afxdev_set_error("cepstrogramObj_cepstrogramBatchDevice: more than 2^31-1 frames in one call");
        return AFX_ERR_UNSUPPORTED;
    }
    AfxCepstrogramArgs a;
    memset(&a, 0, sizeof(a));
    a.x = dData;
    a.timeLength = T * batch;
    a.framesPerClip = T;
    a.clipStride = clipStride;
    a.radix2Exp = o->radix2Exp;
    a.hop = o->slideLength;
    a.cepNum = cepNum < 0 ? 0 : cepNum;
    a.window = o->dWindow;
    a.twiddle = o->dTwiddle;
    a.fastTab = o->dFastTab;
    a.out1 = dOut1;
    a.out2 = dOut2;
    a.out3 = dOut3;
    int st = afxk_cepstrogram(&a, hipStream);
    if (st != AFX_OK) {
        o->status = st;
        afxdev_report_failure("cepstrogramObj_cepstrogramBatchDevice", st);
    }
    return st;
}

void cepstrogramObj_cepstrogram(CepstrogramObj o, int cepNum, float *dataArr, int dataLength,
                                float *mDataArr1, float *mDataArr2, float *mDataArr3) {
    AFX_ENTER(o);
    if (!o) {
        afxdev_set_error("cepstrogramObj_cepstrogram: NULL object");
        return;
    }
    if (!dataArr || dataLength <= 0) return;
    const int T = cepstrogramObj_calTimeLength(o, dataLength);
    if (T <= 0) return;
    run(o, cepNum, dataArr, dataLength, T, NULL, NULL, mDataArr1, mDataArr2, mDataArr3,
        "cepstrogramObj_cepstrogram");
}

void cepstrogramObj_cepstrogram2(CepstrogramObj o, int cepNum, float *mRealArr, float *mImageArr,
                                 int nLength, float *mDataArr1, float *mDataArr2, float *mDataArr3) {
    AFX_ENTER(o);
    if (!o) {
        afxdev_set_error("cepstrogramObj_cepstrogram2: NULL object");
        return;
    }
    if (!mRealArr || !mImageArr || nLength <= 0) return;
    /* the cache is laid out [T_cached, N] re | im; a different nLength re-reads it with the
     * new row count exactly like the reference re-reads its flat scratch */
    if (nLength != o->cachedTime && o->dSpec) {
        /* re-pack so that the imaginary plane starts at nLength*N: only the common prefix
         * is meaningful (the reference would read stale memory beyond it) */
        const size_t N = (size_t)o->fftLength;
        const int common = nLength < o->cachedTime ? nLength : o->cachedTime;
        float *tmp = NULL;
        size_t cap = 0;
        const size_t need = sizeof(float) * 2 * (size_t)nLength * N;
        if (afxdev_reserve((void **)&tmp, &cap, need) == AFX_OK) {
            afxdev_memset(tmp, 0, need, o->stream);
            if (common > 0) {
                afxdev_d2d(tmp, o->dSpec, sizeof(float) * common * N, o->stream);
                afxdev_d2d(tmp + (size_t)nLength * N, o->dSpec + (size_t)o->cachedTime * N,
                           sizeof(float) * common * N, o->stream);
            }
            afxdev_stream_sync(o->stream);
            afxdev_free(o->dSpec);
            o->dSpec = tmp;
            o->capSpec = cap;
            o->cachedTime = nLength;
        }
    }
    run(o, cepNum, NULL, 0, nLength, mRealArr, mImageArr, mDataArr1, mDataArr2, mDataArr3,
        "cepstrogramObj_cepstrogram2");
}

void cepstrogramObj_enableDebug(CepstrogramObj o, int flag) {
    if (o) o->isDebug = flag; /* accepted for ABI compatibility; no dumps are produced */
}

void cepstrogramObj_free(CepstrogramObj o) {
    if (!o) return;
    if (o->stream) afxdev_stream_sync(o->stream);
    afxdev_free(o->dWindow);
    afxdev_free(o->dTwiddle);
    afxdev_free(o->dFastTab);
    afxdev_free(o->dX);
    afxdev_free(o->dOut);
    afxdev_free(o->dSpec);
    afxdev_stream_destroy(o->stream);
    free(o);
}
