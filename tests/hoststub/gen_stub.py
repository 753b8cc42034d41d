#!/usr/bin/env python3
"""Generates stub.c: a CPU stand-in for the device layer (audioflux_amd/csrc/hip/afx_device.h) so that the C host
objects can run under AddressSanitizer / UBSan without a GPU.  "Device" memory is malloc'ed host memory, streams
are dummies, and the kernel launchers do no arithmetic -- but the ones of the CQT path READ every input range and
WRITE every output range they are handed, so that a level buffer that is too small or a pointer that is off shows
up as a sanitizer report.  Test infrastructure (tests/test_hoststub.py), never linked into the product."""
import re
import sys

SPECIAL = r'''
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "afx_device.h"

static char g_err[512];
static int g_errs;
volatile float afx_stub_sink;
static void touch_read(const float *p, long long n) { float s = 0; for (long long i = 0; i < n; i++) s += p[i]; afx_stub_sink = s; }
static void touch_write(float *p, long long n, float v) { for (long long i = 0; i < n; i++) p[i] = v; }

int afxdev_ensure(void) { return AFX_OK; }
const char *afxdev_last_error(void) { return g_err; }
void afxdev_set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); g_errs++; }
int afxdev_error_count(void) { return g_errs; }
int afxdev_malloc(void **dptr, size_t bytes) { *dptr = malloc(bytes ? bytes : 1); return *dptr ? AFX_OK : AFX_ERR_NOMEM; }
void afxdev_free(void *dptr) { free(dptr); }
int afxdev_memset(void *dptr, int value, size_t bytes, void *stream) { (void)stream; memset(dptr, value, bytes); return AFX_OK; }
int afxdev_h2d(void *dst, const void *src, size_t bytes, void *stream) { (void)stream; memcpy(dst, src, bytes); return AFX_OK; }
int afxdev_d2h(void *dst, const void *src, size_t bytes, void *stream) { (void)stream; memcpy(dst, src, bytes); return AFX_OK; }
int afxdev_d2d(void *dst, const void *src, size_t bytes, void *stream) { (void)stream; memmove(dst, src, bytes); return AFX_OK; }
int afxdev_stream_create(void **stream) { *stream = malloc(8); return *stream ? AFX_OK : AFX_ERR_NOMEM; }
void afxdev_stream_destroy(void *stream) { free(stream); }
int afxdev_reserve(void **dptr, size_t *capacity, size_t bytes) {
    if (*dptr && *capacity >= bytes) return AFX_OK;
    free(*dptr);
    *dptr = malloc(bytes ? bytes : 1);  /* exactly what was asked for: an overrun is an ASan report */
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
int afxk_cqt_all_f16(const AfxCqtAllArgs *a, void *stream) {
    (void)stream;
    if (!a->imageH || !a->colMul || a->num != 84) return AFX_ERR_UNSUPPORTED;
    touch_read(a->colMul, 32);
    touch_read((const float *)a->imageH, 2 * 32 * 64 * 8 / 2);
    touch_read(a->scale, a->num);
    for (int b = 0; b < a->batch; b++) {
        for (int l = 0; l < 7; l++) touch_read(a->x[l] + b * a->xStride[l], a->validLength[l]);
        touch_write(a->outRe + b * a->outStride, (long long)a->timeLength * a->num, 1.f);
        touch_write(a->outIm + b * a->outStride, (long long)a->timeLength * a->num, 2.f);
        if (a->chroma) touch_write(a->chroma + b * a->chromaStride, (long long)a->timeLength * 12, 3.f);
    }
    if (a->chroma) for (int j = 0; j < 84; j++) if (a->cls[j] >= 12) return AFX_ERR_ARG;
    return AFX_OK;
}
int afxk_cqt_chroma(const float *re, const float *im, long long rows, int num, const unsigned char *fold,
                    const AfxChromaLists *lists, int chromaNum, int isMag, int normType, float *out, void *stream) {
    (void)isMag; (void)normType; (void)stream;
    touch_read(re, rows * num);
    touch_read(im, rows * num);
    long long s = 0;
    for (int i = 0; i < chromaNum * num; i++) s += fold[i];
    if (lists) s += lists->start[chromaNum];
    afx_stub_sink = (float)s;
    touch_write(out, rows * chromaNum, 3.f);
    return AFX_OK;
}
'''

DONE = {"afxdev_ensure", "afxdev_last_error", "afxdev_set_error", "afxdev_error_count", "afxdev_malloc", "afxdev_free",
        "afxdev_memset", "afxdev_h2d", "afxdev_d2h", "afxdev_d2d", "afxdev_stream_create", "afxdev_stream_destroy",
        "afxdev_reserve", "afxk_cqt_decimate", "afxk_cqt_octave", "afxk_cqt_octave_f16", "afxk_cqt_all_f16",
        "afxk_cqt_chroma"}


def main(header, out):
    src = open(header).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    protos = re.findall(r"^((?:const\s+)?[A-Za-z_][A-Za-z0-9_ ]*?[\s\*]+)(afx[dk][a-z]*_[A-Za-z0-9_]+)\s*\(([^;{}]*?)\)\s*;",
                        src, flags=re.M | re.S)
    body = [SPECIAL, "\n/* ---- everything else: accepted, nothing done */\n"]
    for ret, name, args in protos:
        if name in DONE:
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
    main(sys.argv[1], sys.argv[2])
