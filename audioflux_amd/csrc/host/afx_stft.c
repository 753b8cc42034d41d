/* afx_stft.c -- the STFT object (C host side) behind include/stft_algorithm.h.
 *
 * Mirrors the parameter semantics and the framing state machine of the reference object
 * (src/stft_algorithm.c:80-871): defaults, padding switches, the streaming tail kept between
 * calls, frame counts, the inverse's method switch.  Execution differs: a call uploads the
 * samples, the framed FFT kernel gathers every frame straight from the (virtually padded)
 * clip -- the reference's padded copy `curDataArr` never exists, the padding is an index map
 * inside the kernel -- and all fftLength bins come back as split re / im planes.
 * There is no CPU compute path.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "afx_batch.h"
#include "afx_device.h"
#include "afx_host.h"
#include "afx_objects.h"
#include "stft_algorithm.h"

int stftObj_new(STFTObj *stftObj, int radix2Exp, WindowType *windowType, int *slideLength,
                int *isContinue) {
    if (!stftObj) return -1;
    *stftObj = NULL;
    if (radix2Exp < 1 || radix2Exp > 30) return -100; /* stft_algorithm.c:113-116 (silent) */
    if (radix2Exp > 14) {
        afxdev_set_error("stftObj_new: fftLength 2^%d exceeds the on-chip FFT limit 2^14", radix2Exp);
        return AFX_ERR_UNSUPPORTED;
    }
    int st = afxdev_ensure();
    if (st != AFX_OK) return st;
    STFTObj o = (STFTObj)calloc(1, sizeof(struct OpaqueSTFT));
    if (!o) return AFX_ERR_NOMEM;
    o->radix2Exp = radix2Exp;
    o->fftLength = 1 << radix2Exp;
    o->windowType = windowType ? *windowType : Window_Rect;
    o->slideLength = o->fftLength / 4 > 0 ? o->fftLength / 4 : 1; /* fftLength 2: the reference's default of 0 divides by
                                                                   * zero in its frame count (stft_algorithm.c:251) */
    if (slideLength && *slideLength > 0) o->slideLength = *slideLength;
    o->isContinue = isContinue ? *isContinue : 0;
    o->positionType = PaddingPosition_Center;
    o->modeType = PaddingMode_Constant;
    o->methodType = -1;

    const size_t nb = sizeof(float) * (size_t)o->fftLength;
    o->windowDataArr = afx_window_fft(o->windowType, o->fftLength);
    o->tailDataArr = (float *)calloc((size_t)o->fftLength, sizeof(float));
    o->winArr1 = (float *)calloc((size_t)o->fftLength, sizeof(float));
    o->winArr2 = (float *)calloc((size_t)o->fftLength, sizeof(float));
    float *tw = afx_twiddle_table(o->fftLength);
    if (!o->windowDataArr || !o->tailDataArr || !o->winArr1 || !o->winArr2 || !tw) st = AFX_ERR_NOMEM;
    if (st == AFX_OK) st = afxdev_stream_create(&o->stream);
    if (st == AFX_OK) st = afxdev_malloc((void **)&o->dWindow, nb);
    if (st == AFX_OK) st = afxdev_malloc((void **)&o->dTwiddle, nb < 8 ? 8 : nb);
    if (st == AFX_OK) st = afxdev_malloc((void **)&o->dWin12, 2 * nb);
    if (st == AFX_OK) st = afxdev_h2d(o->dTwiddle, tw, nb < 8 ? 8 : nb, o->stream);
    if (st == AFX_OK) st = afxdev_stream_sync(o->stream);
    o->windowDirty = 1;
    free(tw);
    if (st != AFX_OK) {
        stftObj_free(o);
        return st;
    }
    *stftObj = o;
    return 0;
}

void stftObj_setSlideLength(STFTObj o, int slideLength) {
    if (o && slideLength > 0) o->slideLength = slideLength;
}

void stftObj_enableContinue(STFTObj o, int flag) {
    if (o) o->isContinue = flag;
}

void stftObj_enablePadding(STFTObj o, int flag) {
    if (o) o->isPad = flag;
}

void stftObj_setPadding(STFTObj o, PaddingPositionType *positionType, PaddingModeType *modeType,
                        float *value1, float *value2) {
    if (!o || !o->isPad) return; /* stft_algorithm.c:189 */
    if (positionType) o->positionType = *positionType;
    if (modeType) o->modeType = *modeType;
    if (value1) o->padValue1 = *value1;
    if (value2) o->padValue2 = *value2;
}

void stftObj_useWindowDataArr(STFTObj o, float *winDataArr) {
    if (!o || !winDataArr) return;
    memcpy(o->windowDataArr, winDataArr, sizeof(float) * (size_t)o->fftLength);
    o->windowDirty = 1;
    o->methodType = -1; /* the synthesis windows are powers of this array */
}

float *stftObj_getWindowDataArr(STFTObj o) { return o ? o->windowDataArr : NULL; }

/* frames and dropped / kept tail of `dataLength` samples (stft_algorithm.c:826-850) */
static void time_and_tail(int dataLength, int fftLength, int slideLength, int isPad, int *timeLen,
                          int *tailLen) {
    if (!isPad) {
        *timeLen = (dataLength - fftLength) / slideLength + 1;
        *tailLen = (dataLength - fftLength) % slideLength + (fftLength - slideLength);
    } else {
        *timeLen = dataLength / slideLength + 1;
        *tailLen = (*timeLen > 1) ? dataLength % slideLength : 0;
    }
}

int stftObj_calTimeLength(STFTObj o, int dataLength) {
    if (!o) return 0;
    if (!o->isPad) {
        if (o->isContinue) dataLength += o->tailDataLength;
        if (dataLength < o->fftLength) return 0;
        return (dataLength - o->fftLength) / o->slideLength + 1;
    }
    if (dataLength <= 0) return 0;
    return dataLength / o->slideLength + 1;
}

int stftObj_calDataLength(STFTObj o, int timeLength) {
    if (!o) return 0;
    return (timeLength - 1) * o->slideLength + o->fftLength;
}

/* the window lives on the host (the caller may replace it); upload before a launch */
static int sync_window(STFTObj o, void *stream) {
    if (!o->windowDirty) return AFX_OK;
    int st = afxdev_h2d(o->dWindow, o->windowDataArr, sizeof(float) * (size_t)o->fftLength, stream);
    if (st == AFX_OK) st = afxdev_stream_sync(stream);
    if (st == AFX_OK) o->windowDirty = 0;
    return st;
}

/* kernel arguments of one batch of clips whose `validLength` samples are framed into
 * timeLength frames; padding per the object's switches */
static void fill_args(STFTObj o, AfxStftArgs *a, const float *dData, int batch, int validLength,
                      long long clipStride, int timeLength, float *dRe, float *dIm) {
    memset(a, 0, sizeof(*a));
    a->x = dData;
    a->clipStride = clipStride;
    a->batch = batch;
    a->dataLength = validLength;
    a->timeLength = timeLength;
    a->radix2Exp = o->radix2Exp;
    a->hop = o->slideLength;
    a->window = o->dWindow;
    a->twiddle = o->dTwiddle;
    a->mode = AFX_SPEC_COMPLEX;
    a->binLo = 0;
    a->binCount = o->fftLength;
    a->fullSpectrum = 1;
    a->outRe = dRe;
    a->outIm = dIm;
    if (!o->isPad) return;
    const int N = o->fftLength;
    /* where the data sits inside the padded clip (stft_algorithm.c:631-639) */
    if (o->positionType == PaddingPosition_Center) a->padLeft = N / 2;
    else if (o->positionType == PaddingPosition_Left) a->padLeft = N;
    else a->padLeft = 0;
    if (o->modeType == PaddingMode_Constant) {
        a->padMode = AFX_PAD_CONST;
        if (o->positionType == PaddingPosition_Center) {
            a->padValueL = o->padValue1;
            a->padValueR = o->padValue2;
        } else {
            /* __vpad_left1 / __vpad_right1 take the constant as an int (flux_vectorOp.c:641-651) */
            a->padValueL = a->padValueR = (float)(int)o->padValue1;
        }
    } else if (o->modeType == PaddingMode_Reflect && validLength > 1) {
        a->padMode = AFX_PAD_REFLECT;
    } else if (o->modeType == PaddingMode_Wrap && validLength > 1) {
        a->padMode = AFX_PAD_WRAP;
    } else {
        a->padMode = AFX_PAD_ZERO; /* nothing is written over the calloc'ed margins */
    }
}

int stftObj_stftBatchDevice(STFTObj o, const float *dData, int batch, int dataLength,
                            long long clipStride, float *dReal, float *dImag, void *hipStream) {
    AFX_ENTER(o);
    if (!o || !dData || !dReal || !dImag || batch <= 0 || dataLength <= 0) return AFX_ERR_ARG;
    int T, tail, valid = dataLength;
    if (o->isPad) {
        time_and_tail(dataLength, o->fftLength, o->slideLength, 1, &T, &tail);
        valid = dataLength - tail; /* the ragged tail is dropped before padding (:650-653) */
    } else {
        if (dataLength < o->fftLength) return AFX_OK;
        T = (dataLength - o->fftLength) / o->slideLength + 1;
    }
    if (T <= 0) return AFX_OK;
    int st = sync_window(o, hipStream);
    if (st != AFX_OK) return st;
    AfxStftArgs a;
    fill_args(o, &a, dData, batch, valid, clipStride, T, dReal, dImag);
    return afxk_stft(&a, hipStream);
}

/* streaming / padded framing state of one legacy call (stft_algorithm.c:474-599):
 * returns the frame count (0: nothing to transform), *total = samples to upload (tail + data),
 * *skip = leading samples of dataArr to drop (negative tail of a hop > fftLength stream) */
int afx_stft_deal_data(STFTObj o, const float *dataArr, int dataLength, int *valid, int *headTail,
                     int *skip) {
    const int N = o->fftLength, H = o->slideLength;
    int timeLen = 0, tailLen = 0;
    *headTail = 0;
    *skip = 0;
    if (o->isPad) {
        time_and_tail(dataLength, N, H, 1, &timeLen, &tailLen);
        *valid = dataLength - tailLen;
        o->tailDataLength = 0;
        o->timeLength = timeLen;
        return timeLen;
    }
    const int oldTail = o->isContinue ? o->tailDataLength : 0;
    const int total = oldTail + dataLength;
    if (total < N) {
        /* not enough for one frame: keep everything for the next call (:498-501, :563-580) */
        if (o->isContinue) {
            if (total > 0) {
                if (oldTail >= 0) memcpy(o->tailDataArr + oldTail, dataArr, sizeof(float) * (size_t)dataLength);
                else memcpy(o->tailDataArr, dataArr - oldTail, sizeof(float) * (size_t)(dataLength + oldTail));
            }
            o->tailDataLength = total;
        } else {
            o->tailDataLength = 0;
        }
        o->timeLength = 0;
        return 0;
    }
    time_and_tail(total, N, H, 0, &timeLen, &tailLen);
    if (oldTail < 0) *skip = -oldTail;
    else *headTail = oldTail;
    *valid = total;
    o->timeLength = timeLen;
    return timeLen;
}

/* after the upload: the last tailLen samples of [old tail | data] become the new tail (:548-557) */
void afx_stft_keep_tail(STFTObj o, const float *dataArr, int dataLength, int total) {
    if (!o->isContinue || o->isPad) {
        o->tailDataLength = 0;
        return;
    }
    int timeLen, tailLen;
    time_and_tail(total, o->fftLength, o->slideLength, 0, &timeLen, &tailLen);
    if (tailLen > 0) {
        if (tailLen <= dataLength) {
            memcpy(o->tailDataArr, dataArr + (dataLength - tailLen), sizeof(float) * (size_t)tailLen);
        } else {
            const int fromOld = tailLen - dataLength; /* <= old tail length */
            memmove(o->tailDataArr, o->tailDataArr + (o->tailDataLength - fromOld),
                    sizeof(float) * (size_t)fromOld);
            memcpy(o->tailDataArr + fromOld, dataArr, sizeof(float) * (size_t)dataLength);
        }
    }
    o->tailDataLength = tailLen;
}

static void fail(STFTObj o, int st, const char *who) {
    o->status = st;
    afxdev_report_failure(who, st);
}

void stftObj_stft(STFTObj o, float *dataArr, int dataLength, float *mRealArr, float *mImageArr) {
    AFX_ENTER(o);
    if (!o) {
        afxdev_set_error("stftObj_stft: NULL object");
        return;
    }
    if (!dataArr || dataLength <= 0) return; /* stft_algorithm.c:267-269 */
    int valid = dataLength, headTail = 0, skip = 0, T;
    if (o->isPad || o->isContinue) {
        T = afx_stft_deal_data(o, dataArr, dataLength, &valid, &headTail, &skip);
    } else {
        T = stftObj_calTimeLength(o, dataLength);
        o->timeLength = T;
    }
    if (T <= 0 || !mRealArr || !mImageArr) return;
    const int N = o->fftLength;
    const size_t outB = sizeof(float) * (size_t)T * N;
    /* device clip = [kept tail | data (minus a skipped head)]; in pad mode just the data */
    const int upData = o->isPad ? valid : dataLength - skip;
    const int total = headTail + upData;
    int st = afxdev_reserve((void **)&o->dX, &o->capX, sizeof(float) * (size_t)(total > 0 ? total : 1));
    if (st == AFX_OK) st = afxdev_reserve((void **)&o->dOut, &o->capOut, 2 * outB);
    if (st == AFX_OK && headTail > 0)
        st = afxdev_h2d(o->dX, o->tailDataArr, sizeof(float) * (size_t)headTail, o->stream);
    if (st == AFX_OK && upData > 0)
        st = afxdev_h2d(o->dX + headTail, dataArr + skip, sizeof(float) * (size_t)upData, o->stream);
    if (st == AFX_OK) st = sync_window(o, o->stream);
    if (st == AFX_OK) {
        AfxStftArgs a;
        fill_args(o, &a, o->dX, 1, o->isPad ? valid : total, total, T, o->dOut, o->dOut + (size_t)T * N);
        st = afxk_stft(&a, o->stream);
    }
    if (st == AFX_OK) st = afxdev_d2h(mRealArr, o->dOut, outB, o->stream);
    if (st == AFX_OK) st = afxdev_d2h(mImageArr, o->dOut + (size_t)T * N, outB, o->stream);
    if (st == AFX_OK) st = afxdev_stream_sync(o->stream);
    if (o->isContinue && !o->isPad) afx_stft_keep_tail(o, dataArr + skip, dataLength - skip, total);
    if (st != AFX_OK) fail(o, st, "stftObj_stft");
}

/* synthesis windows w^e and w^(e+1) (stft_algorithm.c:333-372), uploaded when they change */
static int sync_synthesis(STFTObj o, int methodType, void *stream) {
    if (o->methodType == methodType && o->methodType >= 0) return AFX_OK;
    const float e = (methodType == 0) ? 1.f : 0.f;
    for (int i = 0; i < o->fftLength; i++) {
        o->winArr1[i] = powf(o->windowDataArr[i], e);
        o->winArr2[i] = powf(o->windowDataArr[i], e + 1);
    }
    const size_t nb = sizeof(float) * (size_t)o->fftLength;
    int st = afxdev_h2d(o->dWin12, o->winArr1, nb, stream);
    if (st == AFX_OK) st = afxdev_h2d(o->dWin12 + o->fftLength, o->winArr2, nb, stream);
    if (st == AFX_OK) st = afxdev_stream_sync(stream);
    if (st == AFX_OK) o->methodType = methodType;
    return st;
}

int stftObj_istftBatchDevice(STFTObj o, const float *dReal, const float *dImag, int batch,
                             int nLength, int type, float *dData, long long dataStride,
                             void *hipStream) {
    AFX_ENTER(o);
    if (!o || !dReal || !dImag || !dData || batch <= 0 || nLength <= 0) return AFX_ERR_ARG;
    /* the frame scratch belongs to the object: drain the previous stream on a switch */
    if (o->lastStreamSet && o->lastStream != hipStream) {
        int sst = afxdev_stream_sync(o->lastStream);
        if (sst != AFX_OK) return sst;
    }
    o->lastStream = hipStream;
    o->lastStreamSet = 1;
    int st = sync_synthesis(o, type, hipStream);
    if (st != AFX_OK) return st;
    AfxIstftArgs a;
    memset(&a, 0, sizeof(a));
    a.re = dReal;
    a.im = dImag;
    a.batch = batch;
    a.timeLength = nLength;
    a.radix2Exp = o->radix2Exp;
    a.hop = o->slideLength;
    a.twiddle = o->dTwiddle;
    a.win1 = o->dWin12;
    a.win2 = o->dWin12 + o->fftLength;
    a.out = dData;
    a.outStride = dataStride;
    /* n_fft 2048: one launch, overlap-add on the chip, no [frames, N] scratch */
    if (!afxdev_no_fused()) {
        st = afxk_istft_fused(&a, hipStream);
        if (st != AFX_ERR_UNSUPPORTED) return st;
    }
    st = afxdev_reserve((void **)&o->dFrames, &o->capFrames, sizeof(float) * (size_t)batch * nLength * o->fftLength);
    if (st != AFX_OK) return st;
    a.frames = o->dFrames;
    return afxk_istft(&a, hipStream);
}

void stftObj_istft(STFTObj o, float *mRealArr, float *mImageArr, int nLength, int type,
                   float *dataArr) {
    AFX_ENTER(o);
    if (!o) {
        afxdev_set_error("stftObj_istft: NULL object");
        return;
    }
    if (!mRealArr || !mImageArr || !dataArr || nLength <= 0) return;
    const int N = o->fftLength;
    const size_t specB = sizeof(float) * (size_t)nLength * N;
    const int dataLength = (nLength - 1) * o->slideLength + N;
    const size_t dataB = sizeof(float) * (size_t)dataLength;
    int st = afxdev_reserve((void **)&o->dOut, &o->capOut, 2 * specB);
    if (st == AFX_OK) st = afxdev_reserve((void **)&o->dX, &o->capX, dataB);
    if (st == AFX_OK) st = afxdev_h2d(o->dOut, mRealArr, specB, o->stream);
    if (st == AFX_OK) st = afxdev_h2d(o->dOut + (size_t)nLength * N, mImageArr, specB, o->stream);
    /* the reference adds the frames ONTO dataArr (:382); carry the caller's content along */
    if (st == AFX_OK) st = afxdev_h2d(o->dX, dataArr, dataB, o->stream);
    if (st == AFX_OK)
        st = stftObj_istftBatchDevice(o, o->dOut, o->dOut + (size_t)nLength * N, 1, nLength, type,
                                      o->dX, dataLength, o->stream);
    if (st == AFX_OK) st = afxdev_d2h(dataArr, o->dX, dataB, o->stream);
    if (st == AFX_OK) st = afxdev_stream_sync(o->stream);
    if (st != AFX_OK) fail(o, st, "stftObj_istft");
}

void stftObj_debug(STFTObj o) {
    if (!o) return;
    printf("stft params is: fftLength=%d, slideLength=%d, timeLength=%d\n", o->fftLength,
           o->slideLength, o->timeLength);
}

void stftObj_free(STFTObj o) {
    if (!o) return;
    if (o->stream) afxdev_stream_sync(o->stream);
    afxdev_free(o->dWindow);
    afxdev_free(o->dTwiddle);
    afxdev_free(o->dWin12);
    afxdev_free(o->dX);
    afxdev_free(o->dOut);
    afxdev_free(o->dFrames);
    afxdev_stream_destroy(o->stream);
    free(o->windowDataArr);
    free(o->tailDataArr);
    free(o->winArr1);
    free(o->winArr2);
    free(o);
}

/* ---- test hooks (host logic only, no device): tests/test_stft_host.py ------------------- */
long long afx_test_pad_index(long long q, int n, int mode) { return afx_pad_index(q, n, mode); }

/* runs the framing state machine of stftObj_stft over `calls` consecutive chunks of one
 * stream WITHOUT a device: cur receives, call after call, exactly the samples stftObj_stft
 * would upload ([kept tail | chunk]), curLens[c] their count (0: no frame this call),
 * timeLens[c] the frame count, tails[c] the tail length kept afterwards */
int afx_test_stft_stream(int radix2Exp, int slideLength, int isPad, const float *data,
                         const int *chunkLens, int calls, float *cur, int *curLens, int *timeLens,
                         int *tails) {
    struct OpaqueSTFT s;
    memset(&s, 0, sizeof(s));
    s.radix2Exp = radix2Exp;
    s.fftLength = 1 << radix2Exp;
    s.slideLength = slideLength;
    s.isContinue = 1;
    s.isPad = isPad;
    s.tailDataArr = (float *)calloc((size_t)s.fftLength, sizeof(float));
    if (!s.tailDataArr) return AFX_ERR_NOMEM;
    long long off = 0, w = 0;
    for (int c = 0; c < calls; c++) {
        const float *chunk = data + off;
        const int n = chunkLens[c];
        int valid = n, headTail = 0, skip = 0;
        if (stftObj_calTimeLength(&s, n) < 0) {
            free(s.tailDataArr);
            return AFX_ERR_ARG;
        }
        const int T = afx_stft_deal_data(&s, chunk, n, &valid, &headTail, &skip);
        timeLens[c] = T;
        curLens[c] = 0;
        if (T > 0) {
            const int upData = s.isPad ? valid : n - skip;
            memcpy(cur + w, s.tailDataArr, sizeof(float) * (size_t)headTail);
            memcpy(cur + w + headTail, chunk + skip, sizeof(float) * (size_t)upData);
            curLens[c] = headTail + upData;
            w += curLens[c];
            if (!s.isPad) afx_stft_keep_tail(&s, chunk + skip, n - skip, headTail + upData);
        }
        tails[c] = s.tailDataLength;
        off += n;
    }
    free(s.tailDataArr);
    return AFX_OK;
}
