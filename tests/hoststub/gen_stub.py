#!/usr/bin/env python3
"""Generates stub.c: a CPU stand-in for the device layer (audioflux_amd/csrc/hip/afx_device.h) so that the C host
objects can run under AddressSanitizer / UBSan without a GPU.  "Device" memory is malloc'ed host memory, streams
are dummies, and the kernel launchers do no arithmetic -- but the ones of the CQT path READ every input range and
WRITE every output range they are handed, so that a level buffer that is too small or a pointer that is off shows
up as a sanitizer report.  --functional-cqt leaves the CQT launchers out: tests/hoststub/cqt_functional.c, which
computes, supplies them.  Test infrastructure (tests/test_hoststub.py), never linked into the product."""
import re
import sys

SPECIAL = r'''
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "afx_device.h"

static _Thread_local char g_err[512];
static _Thread_local int g_errs;
_Thread_local volatile float afx_stub_sink;
/* -DAFX_STUB_DRY: "device" allocations are address ranges without memory behind them and nothing is touched -- the
 * host code can then be driven at sizes that fill a 288 GB device (tests/hoststub/driver_scale.c: size arithmetic
 * under UBSan + clang's integer checks) */
#ifdef AFX_STUB_DRY
#define DRY 1
#else
#define DRY 0
#endif
static void touch_read(const float *p, long long n) { if (DRY) return; float s = 0; for (long long i = 0; i < n; i++) s += p[i]; afx_stub_sink = s; }
static void touch_write(float *p, long long n, float v) { if (DRY) return; for (long long i = 0; i < n; i++) p[i] = v; }
static void *dry_alloc(size_t bytes) {
    static unsigned long long next = 1ull << 44;
    void *p = (void *)(size_t)next;
    next += (bytes + 4095) & ~4095ull;
    return p;
}

int afxdev_ensure(void) { return AFX_OK; }
const char *afxdev_last_error(void) { return g_err; }
void afxdev_set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); g_errs++; }
int afxdev_error_count(void) { return g_errs; }
void afxdev_report_failure(const char *who, int st) { (void)who; (void)st; g_errs++; }
int afxdev_no_fused(void) { return getenv("AFX_NO_FUSED") != NULL; }
int afxdev_cqt_f32(void) { return getenv("AFX_CQT_F32") != NULL; }
/* (a request beyond 16 GiB fails like hipMalloc would on a full device: the constructor must hand the status on) */
int afxdev_malloc(void **dptr, size_t bytes) {
    if (DRY) { *dptr = bytes > ((size_t)288 << 30) ? NULL : dry_alloc(bytes); return *dptr ? AFX_OK : AFX_ERR_NOMEM; }
    *dptr = bytes > ((size_t)1 << 34) ? NULL : malloc(bytes ? bytes : 1);
    return *dptr ? AFX_OK : AFX_ERR_NOMEM;
}
void afxdev_free(void *dptr) { if (!DRY) free(dptr); }
int afxdev_memset(void *dptr, int value, size_t bytes, void *stream) { (void)stream; if (!DRY) memset(dptr, value, bytes); return AFX_OK; }
int afxdev_h2d(void *dst, const void *src, size_t bytes, void *stream) { (void)stream; if (!DRY) memcpy(dst, src, bytes); return AFX_OK; }
int afxdev_d2h(void *dst, const void *src, size_t bytes, void *stream) { (void)stream; if (!DRY) memcpy(dst, src, bytes); return AFX_OK; }
int afxdev_d2d(void *dst, const void *src, size_t bytes, void *stream) { (void)stream; if (!DRY) memmove(dst, src, bytes); return AFX_OK; }
int afxdev_stream_create(void **stream) { *stream = malloc(8); return *stream ? AFX_OK : AFX_ERR_NOMEM; }
void afxdev_stream_destroy(void *stream) { free(stream); }
int afxdev_reserve(void **dptr, size_t *capacity, size_t bytes) {
    if (*dptr && *capacity >= bytes) return AFX_OK;
    if (DRY) { *dptr = bytes > ((size_t)288 << 30) ? NULL : dry_alloc(bytes); *capacity = *dptr ? bytes : 0; return *dptr ? AFX_OK : AFX_ERR_NOMEM; }
    free(*dptr);
    *dptr = bytes > ((size_t)1 << 34) ? NULL : malloc(bytes ? bytes : 1);  /* exactly what was asked for: an overrun is an ASan report */
    *capacity = *dptr ? bytes : 0;
    return *dptr ? AFX_OK : AFX_ERR_NOMEM;
}

/* ---- CQT launchers: touch what the real kernels touch */
int afxk_cqt_decimate(const float *x, int srcLen, long long xStride, float *y, int dstLen, long long yStride, int batch,
                      const float *taps32, float sqrtRatio, void *stream) {
    (void)taps32; (void)sqrtRatio; (void)stream;
    for (int b = 0; b < batch; b++) { touch_read(x + b * xStride, srcLen); touch_write(y + b * yStride, dstLen, 0.25f); }
    return AFX_OK;
}
static void cqt_octave_touch(const AfxCqtOctaveArgs *a) {
    if (DRY) return;
    const int batch = a->batch > 0 ? a->batch : 1;
    for (int b = 0; b < batch; b++) {
        touch_read(a->x + b * a->xStride, a->validLength);
        for (long long t = 0; t < a->timeLength; t++) {
            touch_write(a->outRe + b * a->outStride + t * a->num + a->colBase, a->rows, 1.f);
            touch_write(a->outIm + b * a->outStride + t * a->num + a->colBase, a->rows, 2.f);
        }
    }
    touch_read(a->scale, a->num);
}
int afxk_cqt_octave(const AfxCqtOctaveArgs *a, void *stream) { (void)stream; cqt_octave_touch(a); return AFX_OK; }
int afxk_cqt_octave_f16(const AfxCqtOctaveArgs *a, void *stream) {
    (void)stream;
    if (!a->timeKernelH || !a->colMul || a->colTiles != 1 || a->radix2Exp != 9) return AFX_ERR_UNSUPPORTED;
    touch_read(a->colMul, 32);
    touch_read((const float *)a->timeKernelH, 2 * 32 * 64 * 8 / 2);
    cqt_octave_touch(a);
    return AFX_OK;
}
int afxk_cqt_chroma(const float *re, const float *im, long long rows, int num, const unsigned char *fold,
                    const AfxChromaLists *lists, int chromaNum, int isMag, int normType, float *out, void *stream) {
    (void)isMag; (void)normType; (void)stream;
    touch_read(re, rows * num);
    touch_read(im, rows * num);
    long long s = 0;
    for (int i = 0; !DRY && i < chromaNum * num; i++) s += fold[i];
    if (lists) s += lists->start[chromaNum];
    afx_stub_sink = (float)s;
    touch_write(out, rows * chromaNum, 3.f);
    return AFX_OK;
}

/* ---- the other launchers: footprints as documented in afx_device.h */
static void touch_rw(float *p, long long n) { if (DRY) return; for (long long i = 0; i < n; i++) p[i] = p[i] * 0.5f + 1.f; }
static void touch_rows_w(float *p, long long rows, long long width, long long pitch, float v) {
    if (DRY) return;
    for (long long r = 0; r < rows; r++) touch_write(p + r * pitch, width, v);
}
static void touch_rows_r(const float *p, long long rows, long long width, long long pitch) {
    if (DRY) return;
    for (long long r = 0; r < rows; r++) touch_read(p + r * pitch, width);
}
int afxk_temporal(const AfxStftArgs *a, void *stream) {
    (void)stream;
    const long long rows = (long long)a->batch * a->timeLength;
    if (!a->energy || !a->rms || !a->zcr) return AFX_ERR_ARG;
    touch_read(a->window, 1LL << a->radix2Exp);
    touch_write(a->energy, rows, 1.f);
    touch_write(a->rms, rows, 1.f);
    touch_write(a->zcr, rows, 0.f);
    return AFX_OK;
}
int afxk_stft(const AfxStftArgs *a, void *stream) {
    (void)stream;
    const long long N = 1LL << a->radix2Exp, rows = (long long)a->batch * a->timeLength;
    for (int b = 0; b < a->batch; b++) touch_read(a->x + b * a->clipStride, a->dataLength);
    if (a->window) touch_read(a->window, N);
    touch_read(a->twiddle, N);
    const long long width = a->bandW ? a->bandNum : a->binCount, pitch = a->outPitch ? a->outPitch : width;
    if (a->bandW) { touch_read((const float *)a->bandStart, a->bandNum); touch_read((const float *)a->bandLen, a->bandNum); }
    touch_rows_w(a->outRe, rows, width, pitch, 1.f);
    if (a->outIm && (a->mode == AFX_SPEC_COMPLEX || a->mode == AFX_SPEC_SQUARE)) touch_rows_w(a->outIm, rows, width, pitch, 2.f);
    if (a->energy) touch_write(a->energy, rows, 1.f);
    if (a->rms) touch_write(a->rms, rows, 1.f);
    if (a->zcr) touch_write(a->zcr, rows, 1.f);
    return AFX_OK;
}
/* the one-launch form: not on this stand-in, so that the two-launch path (scratch sizes, footprints) stays under the sanitizers */
int afxk_istft_fused(const AfxIstftArgs *a, void *stream) { (void)a; (void)stream; return AFX_ERR_UNSUPPORTED; }
int afxk_istft(const AfxIstftArgs *a, void *stream) {
    (void)stream;
    const long long N = 1LL << a->radix2Exp, rows = (long long)a->batch * a->timeLength;
    touch_read(a->re, rows * N);
    touch_read(a->im, rows * N);
    touch_read(a->twiddle, N);
    touch_read(a->win1, N);
    touch_read(a->win2, N);
    touch_write(a->frames, rows * N, 0.f);
    for (int b = 0; b < a->batch; b++) touch_rw(a->out + b * a->outStride, (long long)(a->timeLength - 1) * a->hop + N);
    return AFX_OK;
}
int afxk_spec_map(const float *re, const float *im, long long rows, int rowPitch, int binLo, int binCount, int mode,
                  float normValue, float *out, float *out2, void *stream) {
    (void)normValue; (void)stream;
    touch_rows_r(re + binLo, rows, binCount, rowPitch);
    touch_rows_r(im + binLo, rows, binCount, rowPitch);
    touch_write(out, rows * binCount, 1.f);
    if (out2 && (mode == AFX_SPEC_COMPLEX || mode == AFX_SPEC_SQUARE)) touch_write(out2, rows * binCount, 2.f);
    return AFX_OK;
}
int afxk_row_post(float *data, long long rows, int n, int doPow, float powArg, int normType, void *stream) {
    (void)doPow; (void)powArg; (void)normType; (void)stream;
    touch_rw(data, rows * n);
    return AFX_OK;
}
int afxk_gemm_nt(const float *A, long long lda, const float *B, int ldb, float *C, long long ldc, long long M, int N, int K,
                 int pre, int post, float postArg, void *stream) {
    (void)pre; (void)post; (void)postArg; (void)stream;
    touch_rows_r(A, M, K, lda);
    touch_rows_r(B, N, K, ldb);
    touch_rows_w(C, M, N, ldc, 1.f);
    return AFX_OK;
}
int afxk_xxcc_standard(const float *cc, const float *energy, long long rows, int ccNum, int energyType, int deltaLen,
                       float *coe, float *delta1, float *delta2, void *stream) {
    (void)deltaLen; (void)stream;
    const int outLen = ccNum + (energyType == 1);
    touch_read(cc, rows * ccNum);
    if (energyType != 2 && energy) touch_read(energy, rows);
    touch_write(coe, rows * outLen, 1.f);
    if (delta1) touch_write(delta1, rows * outLen, 1.f);
    if (delta2) touch_write(delta2, rows * outLen, 1.f);
    return AFX_OK;
}
int afxk_cwt_forward(const AfxCwtPlanDims *d, const float *tw, const float *x, long long xStride, int chunks,
                     float *scratchA, float *Xt, void *stream) {
    (void)tw; (void)stream;
    const long long L = 1LL << (d->r1 + d->r2);
    for (int c = 0; c < chunks; c++) touch_read(x + c * xStride, d->dataLength);
    touch_write(scratchA, 2 * L * chunks, 0.f);
    touch_write(Xt, 2 * L * chunks, 1.f);
    return AFX_OK;
}
int afxk_cwt_inverse(const AfxCwtPlanDims *d, const float *tw, const float *Xt, const float *bankT, int num, int isDet,
                     int chunks, float *scratchB, float *outRe, float *outIm, int parts, void *stream) {
    (void)tw; (void)isDet; (void)stream;
    const long long L = 1LL << (d->r1 + d->r2);
    touch_read(Xt, 2 * L * chunks);
    touch_read(bankT, (long long)num * L);
    if (parts & AFX_CWT_WIDE) touch_write(scratchB, 2 * L * chunks * (d->order ? d->nWide : num), 0.f);
    if (!d->order) { /* no execution order: every scale is in the wide part */
        if (parts & AFX_CWT_WIDE) {
            touch_write(outRe, (long long)chunks * num * d->dataLength, 1.f);
            touch_write(outIm, (long long)chunks * num * d->dataLength, 2.f);
        }
        return AFX_OK;
    }
#ifdef AFX_STUB_DRY
    (void)outRe; (void)outIm; /* dry mode: device addresses have no memory behind them (the order list included) */
    return AFX_OK;
#else
    /* the rows of the requested parts only (afx_device.h: [time-domain | two-pass | narrow-band classes]); the
     * time-domain rows of the plain transform belong to afxk_cwt_td */
    const int skip = (isDet ? d->tdDet : d->td) ? d->nTd : 0;
    int lo = num, hi = 0;
    if (parts & AFX_CWT_WIDE) lo = skip, hi = d->nTd + d->nWide;
    if (parts & AFX_CWT_NARROW) { if (lo > d->nTd + d->nWide) lo = d->nTd + d->nWide; hi = num; }
    for (int c = 0; c < chunks; c++)
        for (int i = lo; i < hi; i++) {
            const long long at = ((long long)c * num + d->order[i]) * d->dataLength;
            touch_write(outRe + at, d->dataLength, 1.f);
            touch_write(outIm + at, d->dataLength, 2.f);
        }
    return AFX_OK;
#endif
}
int afxk_cwt_small(const AfxCwtPlanDims *d, const float *tw, const float *x, long long xStride, int chunks,
                   const float *bankNatural, int num, int isDet, float *X, float *outRe, float *outIm, void *stream) {
    (void)tw; (void)isDet; (void)stream;
    const long long L = 1LL << (d->r1 + d->r2);
    if (x) { for (int c = 0; c < chunks; c++) touch_read(x + c * xStride, d->dataLength); touch_write(X, 2 * L * chunks, 1.f); }
    else touch_read(X, 2 * L * chunks);
    touch_read(bankNatural, (long long)num * L);
    touch_write(outRe, (long long)chunks * num * d->dataLength, 1.f);
    touch_write(outIm, (long long)chunks * num * d->dataLength, 2.f);
    return AFX_OK;
}
int afxk_wsst_squeeze(const AfxWsstArgs *a, void *stream) {
    (void)stream;
    const long long n = (long long)a->batch * a->num * a->length;
    touch_read(a->wRe, n);
    touch_read(a->wIm, n);
    touch_read(a->dRe, n);
    if (!a->phaseInput) touch_read(a->dIm, n);
    if (a->mode == 2) touch_read(a->freNorm, a->num);
    touch_rw(a->outRe, n);
    touch_rw(a->outIm, n);
    return AFX_OK;
}
int afxk_synsq_phase(const float *re, const float *im, int num, long long length, float *phase, void *stream) {
    (void)stream;
    touch_read(re, num * length);
    touch_read(im, num * length);
    touch_write(phase, num * length, 0.f);
    return AFX_OK;
}
int afxk_reassign(const AfxReassignArgs *a, int order, int *idxScratch, void *stream) {
    (void)order; (void)idxScratch; (void)stream;
    const long long n = (long long)a->batch * a->timeLength * a->F;
    touch_read(a->hRe, n);
    touch_read(a->hIm, n);
    if (a->doFre) { touch_read(a->dhRe, n); touch_read(a->dhIm, n); }
    if (a->doTime) { touch_read(a->thRe, n); touch_read(a->thIm, n); }
    touch_read(a->freArr, a->F);
    touch_write((float *)a->timeIdx, n, 0.f);
    touch_write((float *)a->freIdx, n, 0.f);
    touch_rw(a->outRe, n);
    if (a->outIm) touch_rw(a->outIm, n);
    return AFX_OK;
}
int afxk_cqt_deconv(const float *in, long long rows, int num, int radix2Exp, const float *twiddle, const int *hcIdx,
                    int hcNum, float *outTimbre, float *outPitch, float *outHc, void *stream) {
    (void)stream;
    touch_read(in, rows * num);
    touch_read(twiddle, 1LL << radix2Exp);
    if (outTimbre) touch_write(outTimbre, rows * num, 1.f);
    if (outPitch) touch_write(outPitch, rows * num, 1.f);
    if (outHc) { touch_read((const float *)hcIdx, hcNum); touch_write(outHc, rows * hcNum, 1.f); }
    return AFX_OK;
}
int afxk_cepstrogram(const AfxCepstrogramArgs *a, void *stream) {
    (void)stream;
    const long long N = 1LL << a->radix2Exp, T = a->timeLength;
    if (a->x) {
        if (a->framesPerClip > 0) {
            for (long long f = 0; f < T; f += a->framesPerClip)
                touch_read(a->x + (f / a->framesPerClip) * a->clipStride,
                           (long long)((T - f < a->framesPerClip ? T - f : a->framesPerClip) - 1) * a->hop + N);
        } else {
            touch_read(a->x, (T - 1) * a->hop + N);
        }
        if (a->specRe) touch_write(a->specRe, T * N, 1.f);
        if (a->specIm) touch_write(a->specIm, T * N, 1.f);
    } else {
        touch_read(a->specRe, T * N);
        touch_read(a->specIm, T * N);
    }
    touch_read(a->window, N);
    touch_read(a->twiddle, N);
    if (a->out1) touch_write(a->out1, T * (N / 2 + 1), 1.f);
    if (a->out2) touch_write(a->out2, T * (N / 2 + 1), 1.f);
    if (a->out3) touch_write(a->out3, T * (N / 2 + 1), 1.f);
    return AFX_OK;
}
int afxk_cepstrum_supported(const float *in, int num, int ccNum) { (void)in; return num <= 128 && ccNum <= 32; }
int afxk_cepstrum(const float *in, long long rows, int num, const float *dct, int ccNum, int pre, float *out, void *stream) {
    (void)pre; (void)stream;
    touch_read(in, rows * num);
    touch_read(dct, (long long)ccNum * num);
    touch_write(out, rows * ccNum, 1.f);
    return AFX_OK;
}
typedef struct { int num, radix2Exp; } StubPlan;
int afxk_melfused_variant(int radix2Exp, int tapsA, int tapsB) {
    return (radix2Exp >= 10 && radix2Exp <= 12 && tapsA <= 72 && tapsB <= 32) ? 0 : -1;
}
int afxk_melfused_create(void **plan, int radix2Exp, const float *hWindow, const AfxBandPlan *band, void *stream) {
    (void)stream;
    touch_read(hWindow, 1LL << radix2Exp);
    touch_read(band->wA, (long long)band->tapsA * 64);
    if (band->tapsB > 0) touch_read(band->wB, (long long)band->tapsB * 64);
    StubPlan *p = (StubPlan *)malloc(sizeof(StubPlan));
    if (!p) return AFX_ERR_NOMEM;
    p->num = band->num;
    p->radix2Exp = radix2Exp;
    *plan = p;
    return AFX_OK;
}
int afxk_melfused_run(void *plan, const AfxMelFusedArgs *a, void *stream) {
    (void)stream;
    const StubPlan *p = (const StubPlan *)plan;
    const long long rows = (long long)a->batch * a->timeLength;
    if ((a->cc || a->energy) && p->radix2Exp != 11) return AFX_ERR_UNSUPPORTED;
    for (int b = 0; b < a->batch; b++) touch_read(a->x + b * a->clipStride, a->dataLength);
    if (a->out) touch_write(a->out, rows * p->num, 1.f);
    if (a->outIm && a->specMap >= 3) touch_write(a->outIm, rows * p->num, 2.f);
    if (a->cc) { touch_read(a->dct, (long long)p->num * p->num); touch_write(a->cc, rows * a->ccNum, 1.f); }
    if (a->energy) touch_write(a->energy, rows, 1.f);
    if (a->rms) touch_write(a->rms, rows, 1.f);
    if (a->zcr) touch_write(a->zcr, rows, 1.f);
    return AFX_OK;
}
void afxk_melfused_destroy(void *plan) { free(plan); }
int afxk_melfused_kind(const void *plan) { return plan ? 1 : 0; }
'''

DONE = {"afxdev_ensure", "afxdev_last_error", "afxdev_set_error", "afxdev_error_count", "afxdev_report_failure", "afxdev_no_fused", "afxdev_cqt_f32", "afxdev_malloc", "afxdev_free",
        "afxdev_memset", "afxdev_h2d", "afxdev_d2h", "afxdev_d2d", "afxdev_stream_create", "afxdev_stream_destroy",
        "afxdev_reserve", "afxk_cqt_decimate", "afxk_cqt_octave", "afxk_cqt_octave_f16",
        "afxk_cqt_chroma", "afxk_stft", "afxk_temporal", "afxk_istft", "afxk_istft_fused", "afxk_spec_map", "afxk_row_post", "afxk_gemm_nt",
        "afxk_xxcc_standard", "afxk_cwt_forward", "afxk_cwt_inverse", "afxk_cwt_small", "afxk_wsst_squeeze",
        "afxk_synsq_phase", "afxk_reassign", "afxk_cqt_deconv", "afxk_cepstrogram", "afxk_cepstrum_supported",
        "afxk_cepstrum", "afxk_melfused_variant", "afxk_melfused_create", "afxk_melfused_run", "afxk_melfused_destroy",
        "afxk_melfused_kind"}


CQT_MARK = "/* ---- CQT launchers: touch what the real kernels touch */"
REST_MARK = "/* ---- the other launchers: footprints as documented in afx_device.h */"


def main(header, out, functional_cqt=False, omit=()):
    """functional_cqt: leave the CQT launchers out (tests/hoststub/cqt_functional.c, which computes, supplies them);
    omit: further launchers somebody else defines (tests/emu: kernels emulated on the host)"""
    special = SPECIAL
    if functional_cqt:
        head, rest = SPECIAL.split(CQT_MARK)
        _, tail = rest.split(REST_MARK)
        special = head + REST_MARK + tail
    src = open(header).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    protos = re.findall(r"^((?:const\s+)?[A-Za-z_][A-Za-z0-9_ ]*?[\s\*]+)(afx[dk][a-z]*_[A-Za-z0-9_]+)\s*\(([^;{}]*?)\)\s*;",
                        src, flags=re.M | re.S)
    body = [special, "\n/* ---- everything else: accepted, nothing done */\n"]
    for ret, name, args in protos:
        if name in DONE or name in omit:
            continue
        ret = " ".join(ret.split())
        args = " ".join(args.split())
        names = []
        if args != "void":
            for a in args.split(","):
                m = re.search(r"([A-Za-z_][A-Za-z0-9_]*)\s*(\[[^\]]*\])?$", a.strip())
                names.append(m.group(1))
        voids = " ".join(f"(void){n};" for n in names)
        if ret == "void":
            body.append(f"{ret} {name}({args}) {{ {voids} }}\n")
        elif "*" in ret:
            body.append(f"{ret} {name}({args}) {{ {voids} return 0; }}\n")
        else:
            body.append(f"{ret} {name}({args}) {{ {voids} return 0; }}\n")
    open(out, "w").write("".join(body))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], "--functional-cqt" in sys.argv[3:],
         [a.split("=", 1)[1] for a in sys.argv[3:] if a.startswith("--omit=")])
