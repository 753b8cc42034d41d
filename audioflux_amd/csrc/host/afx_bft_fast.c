/* afx_bft_fast.c -- glue between the BFT object and the fused
 * STFT -> banded-filter-bank kernel (afx_melfused.hip).  The plan is built once
 * per object when the configuration qualifies (n_fft 2048, a banded bank whose
 * rows fit one of the compiled tap variants); everything else keeps using the
 * size-generic kernels. */
#include <stdlib.h>
#include <string.h>

#include "afx_device.h"
#include "afx_host.h"
#include "afx_objects.h"

int afx_bandplan_build(const float *bank, int num, int F, AfxBandPlan *p);
int afx_bandplan_build_split(const float *bank, int num, int F, int tapsA, int tapsB, int rowCap,
                             AfxBandPlan *p);
void afx_bandplan_free(AfxBandPlan *p);

int afx_bft_plan_fast(struct OpaqueBFT *o, const float *hWindow, const float *hBank) {
    o->fast = NULL;
    if (!hBank || o->scale == SpectralFilterBankScale_Linear) return AFX_OK;
    if (afxk_melfused_variant(o->radix2Exp, 1, 1) < 0) return AFX_OK;
    AfxBandPlan band;
    if (afx_bandplan_build(hBank, o->num, o->F, &band) != 0) return AFX_OK;
    int st = AFX_OK;
    int fits = afxk_melfused_variant(o->radix2Exp, band.tapsA, band.tapsB) >= 0;
    if (!fits && o->radix2Exp >= 9 && o->radix2Exp <= 12) {
        /* rows longer than the compiled tap variants (mel-40 / -64 / -80, bark, erb, higher sample
         * rates): cut them into segments, smallest variant first (afx_bandplan.c); the last number
         * is the length of the kernel's zero-padded power row (PROW_F of afx_melfused{512,1k,2,4k2}.hip) */
        static const int v512[4][3] = {{16, 4, 384}, {32, 4, 384}, {48, 4, 384}, {64, 8, 384}};
        static const int v1k[4][3] = {{24, 8, 640}, {32, 32, 640}, {48, 16, 640}, {72, 32, 640}};
        static const int v2k[2][3] = {{48, 16, 1104}, {72, 32, 1104}};
        static const int v4k[3][3] = {{96, 32, 2176}, {128, 64, 2176}, {176, 8, 2176}};
        const int (*v)[3] = o->radix2Exp == 9 ? v512 : o->radix2Exp == 10 ? v1k : o->radix2Exp == 11 ? v2k : v4k;
        const int nv = o->radix2Exp == 11 ? 2 : o->radix2Exp == 12 ? 3 : 4;
        afx_bandplan_free(&band);
        for (int i = 0; i < nv && !fits; i++)
            fits = afx_bandplan_build_split(hBank, o->num, o->F, v[i][0], v[i][1], v[i][2], &band) == 0;
        if (!fits) return AFX_OK;
    }
    if (fits) {
        void *plan = NULL;
        st = afxk_melfused_create(&plan, o->radix2Exp, hWindow, &band, o->stream);
        if (st == AFX_OK) o->fast = (struct AfxMelFusedPlan *)plan;
    }
    afx_bandplan_free(&band);
    return st;
}

/* argument block of a real- or complex-result run of the fused kernels */
static void fast_args(struct OpaqueBFT *o, const float *dData, int batch, int dataLength,
                      long long clipStride, float *dRe, float *dIm, AfxMelFusedArgs *a) {
    memset(a, 0, sizeof(*a));
    a->x = dData;
    a->clipStride = clipStride;
    a->batch = batch;
    a->dataLength = dataLength;
    a->timeLength = (dataLength - o->fftLength) / o->slideLength + 1;
    a->hop = o->slideLength;
    a->normValue = o->normValue;
    a->out = dRe;
    a->outIm = dIm;
    if (!o->resultType) {
        /* complex result (bft_algorithm.c:457-485): the spectrum itself for MAG, its complex
         * square for POWER; the norm exponent plays no part */
        a->specMap = (o->dataType == SpectralData_Power) ? 4 : 3;
    } else if (o->dataType == SpectralData_Mag) {
        a->specMap = 1;
        a->postPow = (o->normValue != 1);
    } else {
        a->specMap = (o->dataType == SpectralData_Power && o->normValue != 1) ? 2 : 0;
    }
}

/* dTemporal: 3 planes of batch*T floats (energy | rms | zcr) or NULL. */
int afx_bft_try_fast(struct OpaqueBFT *o, const float *dData, int batch, int dataLength,
                     long long clipStride, float *dRe, float *dIm, float *dTemporal, void *stream,
                     int *used) {
    *used = 0;
    if (!o->fast) return AFX_OK;
    if (!o->resultType && !dIm) return AFX_OK;
    AfxMelFusedArgs a;
    fast_args(o, dData, batch, dataLength, clipStride, dRe, dIm, &a);
    const long long frames = (long long)batch * a.timeLength;
    /* temporal features ride along in the n_fft 2048 real-result kernel; every other fused kernel runs without them and
     * k_temporal (afx_stft.hip) reads the frames a second time -- from L2 -- instead of the whole call falling back to
     * the size-generic kernels */
    int separate = 0;
    if (dTemporal) {
        if (o->resultType && o->radix2Exp == 11) {
            a.energy = dTemporal;
            a.rms = dTemporal + frames;
            a.zcr = dTemporal + 2 * frames;
        } else {
            separate = 1;
        }
    }
    int st = afxk_melfused_run(o->fast, &a, stream);
    if (st == AFX_ERR_UNSUPPORTED && a.energy) { /* a plan whose kernel has no temporal instantiation */
        a.energy = a.rms = a.zcr = NULL;
        separate = 1;
        st = afxk_melfused_run(o->fast, &a, stream);
    }
    if (st == AFX_ERR_UNSUPPORTED) return AFX_OK; /* size-generic kernels */
    if (st == AFX_OK && separate) {
        AfxStftArgs t;
        memset(&t, 0, sizeof(t));
        t.x = dData;
        t.clipStride = clipStride;
        t.batch = batch;
        t.dataLength = dataLength;
        t.timeLength = a.timeLength;
        t.radix2Exp = o->radix2Exp;
        t.hop = o->slideLength;
        t.window = o->dWindow;
        t.energy = dTemporal;
        t.rms = dTemporal + frames;
        t.zcr = dTemporal + 2 * frames;
        st = afxk_temporal(&t, stream);
    }
    if (st == AFX_OK) *used = 1;
    return st;
}

/* STFT -> bank -> rectify -> DCT-II in ONE launch (xxcc_algorithm.c:95-156 behind bftObj_bft): real-result objects on
 * any fused plan -- n_fft 512 / 1024 / 2048 / 4096, whole-row or split band plans -- with num <= 128 a multiple of 4,
 * ccNum <= 16, log or cube-root rectification.  The headline shape (n_fft 2048, num 128, log) keeps its own form with the
 * DCT operand in LDS; everything else runs afx_ccblock.h inside the size's kernel.  dMel may be NULL: the rows then go to
 * the object's scratch (the kernel re-reads every 16 of them from L2 to form the cepstra).  A plan / mode without a fused
 * form reports *used = 0 and the caller runs the two kernels. */
int afx_bft_try_fast_cc(struct OpaqueBFT *o, struct OpaqueXXCC *x, const float *dData, int batch,
                        int dataLength, long long clipStride, int ccNum,
                        CepstralRectifyType *rectifyType, float *dMel, float *dCc, void *stream,
                        int *used) {
    *used = 0;
    if (!o->fast || !o->resultType || o->isTemporal) return AFX_OK;
    const CepstralRectifyType rect = rectifyType ? *rectifyType : CepstralRectify_Log;
    if (rect != CepstralRectify_Log && rect != CepstralRectify_CubicRoot) return AFX_OK;
    if (x->num != o->num || ccNum < 1 || ccNum > 16 || o->num > 128 || (o->num & 3)) return AFX_OK;
    AfxMelFusedArgs a;
    fast_args(o, dData, batch, dataLength, clipStride, dMel, NULL, &a);
    if (!a.out) {
        int st = afxdev_reserve((void **)&o->dOut, &o->capOut,
                                sizeof(float) * (size_t)batch * a.timeLength * o->num);
        if (st != AFX_OK) return st;
        a.out = o->dOut;
    }
    a.dct = x->dDct;
    a.ccNum = ccNum;
    a.ccRectify = rect == CepstralRectify_CubicRoot ? 1 : 0;
    a.cc = dCc;
    int st = afxk_melfused_run(o->fast, &a, stream);
    if (st == AFX_ERR_UNSUPPORTED) return AFX_OK;
    if (st == AFX_OK) *used = 1;
    return st;
}

void afx_bft_free_fast(struct OpaqueBFT *o) {
    if (o && o->fast) {
        afxk_melfused_destroy(o->fast);
        o->fast = NULL;
    }
}
