/* afx_cqt.c -- the constant-Q transform object (C host side) behind
 * include/cqt_algorithm.h.
 *
 * Plan construction follows the reference in float32: bin frequencies and
 * lengths (src/filterbank/cqt_filterBank.c:159-246), the temporal kernels
 * w[n] e^{2 pi j n f / fs} centred in fftLength, their FFT and the magnitude
 * threshold (cqt_filterBank.c:57-148, :253-336), the 2:1 resampler table
 * (src/dsp/resample_algorithm.c:546-634) and the chroma folding matrix
 * (src/filterbank/chroma_filterBank.c:176-264).  Execution is the octave
 * recursion of _cqtObj_cqt (src/cqt_algorithm.c:845-1061) on the GPU: per
 * octave one fused frame-FFT + sparse-kernel launch, then a decimation launch.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "afx_device.h"
#include "afx_host.h"
#include "cqt_algorithm.h"

struct OpaqueCQT {
    int num, octaveNum, binPerOctave, samplate;
    int fftLength, radix2Exp, slideLength;
    int isScale, vFlag;
    float minFre;
    WindowType windowType;
    SpectralFilterBankNormalType normType;
    float *freBandArr;       /* host, num+2 */
    float *sLenArr;          /* host, num: sqrt(len) */
    float taps[32];          /* resampler FIR h_j = table[256 j] */
    int timeLength;          /* frames of the last cqt call */
    /* streaming (isContinue, cqt_algorithm.c:345-456): samples kept from the previous calls, and the assembled
     * tail + new data of a call; tailLength < 0: that many samples of the next call are skipped (hop > fftLength) */
    int isContinue;
    float *tailData;         /* host, fftLength + slideLength floats */
    int tailLength;
    float *validData;        /* host */
    size_t capValid;
    /* device */
    void *stream;
    void *stream2;           /* side stream of the device call when the caller's stream IS `stream` */
    float *dTwiddle, *dKTaps, *dScaleOn, *dScaleOff;
    int *dKStart, *dKLen, *dKOff;
    float *dTimeKernel;      /* [groups][N][colTiles*32]: time-domain image of the spectral kernels */
    int colTiles;            /* 0: the matrix-core path is not available for this plan */
    unsigned short *dTimeKernelH; /* [groups][2 words][N/16 steps][64 lanes][8]: the image as f16 (hi, lo) words of
                              * its power-of-two scaled columns, in MFMA fragment order (afx_cqt_f16.hip) */
    float *dColMul;          /* [groups][32]: 2^-s_j undoing the column scaling */
    unsigned char *dFold;
    AfxChromaLists foldLists; /* the folding matrix as per-class bin lists (k_cqt_chroma) */
    int haveLists;
    int foldChromaNum;
    float *dX;               /* staged input of the host-pointer calls */
    size_t capX;
    float *dSig[2];          /* ping-pong decimated signals, all clips of a batch */
    size_t capSig[2];
    float *dRing;            /* level rings of the one-launch ladder (k_cqt_pyramid), per workgroup */
    size_t capRing;
    int noPyramid;           /* AFX_CQT_PYRAMID=0 when the object was created: per-octave launches */
    int pyrTiles;            /* AFX_CQT_PYR_TILES when the object was created (tests): tiles per workgroup run, 0 = planner's */
    int pyrTiming;           /* -DAFX_EXPERIMENTS builds, AFX_CQT_PYR_TIMING set when the object was created */
    unsigned short *dDecTab; /* the resampler taps as the f16 table of k_cqt_pyramid (afx_cqt_dec_table) */
    unsigned long long *dTiming; /* AFX_CQT_PYR_TIMING=1: phase cycles of the instrumented kernel (afx_cqt_pyramid_timing) */
    void *lastStream;        /* stream of the previous device call (scratch ordering) */
    int lastUsed;
    float *dOut;             /* re | im [T,num] */
    size_t capOut;
    float *dIn;              /* chroma / cqcc staging */
    size_t capIn;
    float *dDct;             /* [num,num] for cqcc */
    float *dDevTw;           /* twiddles of the cqhc / deconv transform (length 2^devRadix) */
    int devRadix;
    int status;
};

static void fail(CQTObj o, int st, const char *who) {
    o->status = st;
    afxdev_report_failure(who, st);
}

/* float32 -> IEEE binary16, round to nearest even (gcc 11 on x86-64 has no _Float16) */
static unsigned short cqt_f32_to_f16(float f) {
    union { float f; unsigned u; } v;
    v.f = f;
    const unsigned sign = (v.u >> 16) & 0x8000u, ax = v.u & 0x7fffffffu;
    if (ax >= 0x7f800000u) return (unsigned short)(sign | (ax > 0x7f800000u ? 0x7e00u : 0x7c00u));
    if (ax >= 0x477ff000u) return (unsigned short)(sign | 0x7c00u); /* >= 65520 rounds to infinity */
    if (ax < 0x38800000u) {                                         /* below 2^-14: subnormal, step 2^-24 */
        const float t = fabsf(f) * 16777216.0f;
        return (unsigned short)(sign | (unsigned)lrintf(t));        /* default rounding mode: nearest even */
    }
    unsigned h = (((ax >> 23) - 112u) << 10) | ((ax & 0x7fffffu) >> 13);
    const unsigned rem = ax & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;
    return (unsigned short)(sign | h);
}

static float cqt_f16_to_f32(unsigned short h) {
    const int e = (h >> 10) & 0x1f, m = h & 0x3ff;
    float v;
    if (e == 0) v = ldexpf((float)m, -24);
    else if (e == 31) v = m ? NAN : INFINITY;
    else v = ldexpf((float)(1024 + m), e - 25);
    return (h & 0x8000) ? -v : v;
}

/* One group's image G [N][32] -> (hi, lo) f16 words of the columns scaled to a peak in [2^13, 2^14), in the
 * fragment order of v_mfma_f32_32x32x16_f16's B operand: word w, step ks, lane l (column j = l & 31, half
 * g = l >> 5), element e holds row k = 16 ks + 8 g + e.  colMul[j] = 2^-s_j. */
void afx_cqt_time_kernel_f16(const float *G, int N, unsigned short *out, float *colMul) {
    const int KS = N / 16;
    for (int j = 0; j < 32; j++) {
        float peak = 0.f;
        for (int k = 0; k < N; k++) peak = fmaxf(peak, fabsf(G[(size_t)k * 32 + j]));
        int s = 0;
        if (peak > 0.f && isfinite(peak)) {
            int ex;
            (void)frexpf(peak, &ex); /* peak = m 2^ex, m in [0.5, 1) */
            s = 14 - ex;             /* peak 2^s in [2^13, 2^14) */
            if (s > 100) s = 100;
            if (s < -100) s = -100;
        }
        colMul[j] = ldexpf(1.f, -s);
        for (int k = 0; k < N; k++) {
            const float v = ldexpf(G[(size_t)k * 32 + j], s); /* exact */
            const unsigned short hi = cqt_f32_to_f16(v);
            const unsigned short lo = cqt_f32_to_f16(v - cqt_f16_to_f32(hi));
            const int ks = k / 16, g = (k % 16) / 8, e = k % 8, l = 32 * g + j;
            out[(((size_t)0 * KS + ks) * 64 + l) * 8 + e] = hi;
            out[(((size_t)1 * KS + ks) * 64 + l) * 8 + e] = lo;
        }
    }
}

/* The 2:1 resampler's taps as k_cqt_pyramid reads them (afx_device.h: AfxCqtPyramidArgs.decTab): T[d] = h[|d|] 2^15 for
 * |d| <= 31, zero elsewhere, as binary16 (hi, lo) words, out[word][copy a][x] = T[x - 160 - 2a], x < 352: the copy
 * shifted by 2a entries makes the eight taps of column c' = 4m + a of a K step a 16-byte aligned fragment. */
void afx_cqt_dec_table(const float *taps32, unsigned short *out) {
    const int per = AFX_CQT_PYR_TAB_COPY / 2;
    memset(out, 0, sizeof(unsigned short) * AFX_CQT_PYR_TAB_HALFS);
    for (int a = 0; a < 4; a++)
        for (int x = 0; x < per; x++) {
            int d = x - 160 - 2 * a;
            if (d < 0) d = -d;
            if (d > 31) continue;
            const float v = ldexpf(taps32[d], 15); /* exact */
            const unsigned short hi = cqt_f32_to_f16(v);
            const unsigned short lo = cqt_f32_to_f16(v - cqt_f16_to_f32(hi));
            out[(0 * 4 + a) * per + x] = hi;
            out[(1 * 4 + a) * per + x] = lo;
        }
}

int cqtObj_new(CQTObj *cqtObj, int num, int samplate, float minFre, int *isContinue) {
    return cqtObj_newWith(cqtObj, num, &samplate, &minFre, NULL, NULL, NULL, NULL, NULL, NULL,
                          isContinue, NULL, NULL);
}

/* one octave of temporal kernels -> spectral kernels (cqt_filterBank.c:253-336 + FFT) */
static void octave_kernels(int bpo, const float *fre /* fre[-1], fre[bpo] readable */,
                           int samplate, const float *lenArr, int fftLength, int r,
                           SpectralFilterBankNormalType normType, WindowType winType,
                           float *outRe, float *outIm /* [bpo, fftLength] */) {
    float *tr = (float *)calloc((size_t)fftLength, sizeof(float));
    float *ti = (float *)calloc((size_t)fftLength, sizeof(float));
    if (winType == Window_Rect) winType = Window_Hann; /* cqt_filterBank.c:259-263 */
    for (int i = 0; i < bpo; i++) {
        const int len = (int)ceilf(lenArr[i]);
        const float f = fre[i];
        float *w = afx_window_fft(winType, len);
        const int start = (fftLength - len) / 2;
        memset(tr, 0, sizeof(float) * (size_t)fftLength);
        memset(ti, 0, sizeof(float) * (size_t)fftLength);
        float weight = 1;
        for (int j = 0; j < len; j++) {
            if (normType == SpectralFilterBankNormal_None) weight = lenArr[i];
            const float n = (float)j; /* arange(0, len, 1) */
            const float v = (float)(2 * M_PI * n * f / samplate);
            tr[j + start] = cosf(v) * w[j] / weight;
            ti[j + start] = sinf(v) * w[j] / weight;
        }
        weight = 0;
        if (normType == SpectralFilterBankNormal_Area) {
            for (int j = 0; j < len; j++) {
                const float a = tr[j + start], b = ti[j + start];
                weight += sqrtf(a * a + b * b);
            }
            for (int j = 0; j < len; j++) {
                tr[j + start] /= weight;
                ti[j + start] /= weight;
            }
        } else if (normType == SpectralFilterBankNormal_BandWidth) {
            weight = (fre[i + 1] - fre[i - 1]) / 2;
            for (int j = 0; j < len; j++) {
                tr[j + start] /= weight;
                ti[j + start] /= weight;
            }
        }
        for (int j = 0; j < len; j++) {
            tr[j + start] *= (lenArr[i] / fftLength);
            ti[j + start] *= (lenArr[i] / fftLength);
        }
        afx_fft_ref32(r, tr, ti, outRe + (size_t)i * fftLength, outIm + (size_t)i * fftLength);
        free(w);
    }
    free(tr);
    free(ti);
}

int cqtObj_newWith(CQTObj *cqtObj, int num, int *samplate, float *minFre, int *binPerOctave,
                   float *factor, float *beta, float *thresh, WindowType *windowType,
                   int *slideLength, int *isContinue, SpectralFilterBankNormalType *normalType,
                   int *isScale) {
    int sr = 32000, bpo = 12, slide = 0, cont = 0, vFlag = 0, scaleFlag = 1;
    float fmin = 32.703196f, fac = 1, bet = 0, thr = 0.01f;
    WindowType win = Window_Hann;
    SpectralFilterBankNormalType norm = SpectralFilterBankNormal_None;
    if (!cqtObj) return -1;
    *cqtObj = NULL;

    /* validation in the reference's order (cqt_algorithm.c:150-222) */
    if (binPerOctave && *binPerOctave > 0) bpo = *binPerOctave;
    if (bpo % 12 != 0) {
        printf("binPerOctave is error\n");
        return -1;
    }
    if (num < bpo || num % bpo != 0) {
        printf("num is error\n");
        return -1;
    }
    if (samplate && *samplate > 0) sr = *samplate;
    if (minFre && *minFre > 0) fmin = *minFre;
    if (factor && *factor > 0) fac = *factor;
    if (beta) {
        if (*beta > 0) bet = *beta;
        if (bet != 0) vFlag = 1;
    }
    if (thresh && *thresh > 0) thr = *thresh;
    if (windowType) win = *windowType;
    if (slideLength && *slideLength > 0) slide = *slideLength;
    if (isContinue) cont = *isContinue;
    if (normalType) norm = *normalType;
    if (isScale) scaleFlag = *isScale;
    const int octaveNum = num / bpo;
    if (norm == SpectralFilterBankNormal_BandWidth && (octaveNum < 2 || vFlag)) {
        afxdev_set_error("cqtObj_newWith: BandWidth normalisation of a single octave reads before the "
                         "frequency table in the reference (undefined); refused");
        return AFX_ERR_UNSUPPORTED;
    }

    int st = afxdev_ensure();
    if (st != AFX_OK) return st;
    CQTObj o = (CQTObj)calloc(1, sizeof(struct OpaqueCQT));
    if (!o) return AFX_ERR_NOMEM;
    o->num = num;
    o->octaveNum = octaveNum;
    o->binPerOctave = bpo;
    o->samplate = sr;
    o->isScale = scaleFlag;
    o->vFlag = vFlag;
    o->minFre = fmin;
    o->windowType = win;
    o->normType = norm;
    o->isContinue = cont ? 1 : 0;
    {
        const char *e = getenv("AFX_CQT_PYRAMID"); /* read per object, at creation (no process-wide latch) */
        o->noPyramid = e && e[0] == '0';
        e = getenv("AFX_CQT_PYR_TILES"); /* tests: at most this many tiles per workgroup run (read here, like the switch above) */
        o->pyrTiles = e ? atoi(e) : 0;
#ifdef AFX_EXPERIMENTS
        o->pyrTiming = getenv("AFX_CQT_PYR_TIMING") != NULL; /* measurement builds: the instrumented instantiation */
#endif
    }

    /* ---- frequencies, lengths (cqt_filterBank.c:159-246) */
    o->freBandArr = (float *)calloc((size_t)num + 2, sizeof(float));
    o->sLenArr = (float *)calloc((size_t)num, sizeof(float));
    float *lenTop = (float *)calloc((size_t)bpo, sizeof(float));
    const float ratio = powf(2, (float)(1.0 / bpo));
    for (int i = 0; i < octaveNum; i++) {
        float f = fmin * (1 << i);
        o->freBandArr[i * bpo] = f;
        for (int j = 1; j < bpo; j++) {
            f *= ratio;
            o->freBandArr[i * bpo + j] = f;
        }
    }
    const int topIndex = (octaveNum - 1) * bpo;
    const float value = powf(2, (float)(1.0 / bpo)) - 1;
    const float q = fac / value;
    {
        const int len = (int)ceilf(q * sr / (o->freBandArr[topIndex] + bet / value));
        o->fftLength = afx_ceil_pow2(len);
    }
    o->radix2Exp = afx_log2_exact(o->fftLength);
    for (int i = 0; i < bpo; i++) lenTop[i] = q * sr / (o->freBandArr[topIndex + i] + bet / value);
    for (int i = 0; i < num; i++) o->sLenArr[i] = sqrtf(q * sr / (o->freBandArr[i] + bet / value));
    if (slide <= 0) slide = o->fftLength / 4;
    o->slideLength = slide;
    if (o->isContinue) {
        o->tailData = (float *)calloc((size_t)o->fftLength, sizeof(float));
        if (!o->tailData) {
            cqtObj_free(o);
            free(lenTop);
            return AFX_ERR_NOMEM;
        }
    }
    if (o->radix2Exp < 1 || o->radix2Exp > 14 || (slide >> (octaveNum - 1)) < 1) {
        afxdev_set_error("cqtObj_newWith: fftLength %d / slideLength %d unsupported for %d octaves",
                         o->fftLength, slide, octaveNum);
        cqtObj_free(o);
        free(lenTop);
        return AFX_ERR_UNSUPPORTED;
    }

    /* ---- spectral kernels, thresholded, stored as bands */
    const int N = o->fftLength, F = N / 2 + 1;
    const int rowsTotal = vFlag ? num : bpo;
    float *kre = (float *)calloc((size_t)rowsTotal * N, sizeof(float));
    float *kim = (float *)calloc((size_t)rowsTotal * N, sizeof(float));
    if (!vFlag) {
        octave_kernels(bpo, o->freBandArr + topIndex, sr, lenTop, N, o->radix2Exp, norm, win, kre, kim);
    } else { /* variable-Q: every octave has its own kernels at its own (halved) rate */
        int srOct = sr;
        for (int i = octaveNum - 1; i >= 0; i--) {
            octave_kernels(bpo, o->freBandArr + i * bpo, srOct, lenTop, N, o->radix2Exp, norm, win,
                           kre + (size_t)i * bpo * N, kim + (size_t)i * bpo * N);
            srOct /= 2;
        }
    }
    int *kStart = (int *)calloc((size_t)rowsTotal, sizeof(int));
    int *kLen = (int *)calloc((size_t)rowsTotal, sizeof(int));
    int *kOff = (int *)calloc((size_t)rowsTotal, sizeof(int));
    float *kTaps = (float *)calloc((size_t)rowsTotal * F * 2 + 2, sizeof(float));
    int total = 0;
    const float thr2 = thr * thr;
    for (int i = 0; i < rowsTotal; i++) {
        int first = -1, last = -1;
        for (int j = 0; j < F; j++) {
            const float a = kre[(size_t)i * N + j], b = kim[(size_t)i * N + j];
            if (a * a + b * b > thr2) {
                if (first < 0) first = j;
                last = j;
            }
        }
        kStart[i] = first < 0 ? 0 : first;
        kLen[i] = first < 0 ? 0 : last - first + 1;
        kOff[i] = total;
        for (int j = 0; j < kLen[i]; j++) {
            const float a = kre[(size_t)i * N + first + j], b = kim[(size_t)i * N + first + j];
            const int keep = (a * a + b * b > thr2);
            kTaps[2 * (total + j)] = keep ? a : 0.f;
            kTaps[2 * (total + j) + 1] = keep ? b : 0.f;
        }
        total += kLen[i];
    }

    /* ---- time-domain image of the thresholded spectral kernels, for the matrix-core path:
     *      Q[t][j] = sum_k K_j[k] X_t[k] = sum_n x_t[n] G_j[n],  G_j[n] = sum_k K_j[k] e^{-2 pi i k n / N}
     *      (exactly the same linear map; evaluated in double, rounded once).  Columns of a
     *      group: [Re G_0 .. Re G_{bpo-1} | Im G_0 .. Im G_{bpo-1}], padded to 32*colTiles. */
    float *timeKernel = NULL;
    size_t timeKernelBytes = 0;
    o->colTiles = 0;
    if (N >= 256 && N <= 2048 && 2 * bpo <= 96 && !afxdev_no_fused()) {
        const int ct = (2 * bpo + 31) / 32, cols = ct * 32;
        const int groups = rowsTotal / bpo;
        timeKernelBytes = sizeof(float) * (size_t)groups * N * cols;
        timeKernel = (float *)calloc((size_t)groups * N * cols, sizeof(float));
        double *cs = (double *)malloc(sizeof(double) * 2 * (size_t)N);
        if (timeKernel && cs) {
            for (int m = 0; m < N; m++) {
                cs[2 * m] = cos(2.0 * M_PI * m / N);
                cs[2 * m + 1] = -sin(2.0 * M_PI * m / N);
            }
            for (int row = 0; row < rowsTotal; row++) {
                const int g = row / bpo, j = row % bpo;
                float *dst = timeKernel + (size_t)g * N * cols;
                for (int n = 0; n < N; n++) {
                    double re = 0, im = 0;
                    for (int q = 0; q < kLen[row]; q++) {
                        const double a = kTaps[2 * (kOff[row] + q)], b = kTaps[2 * (kOff[row] + q) + 1];
                        const int m = (int)(((long long)(kStart[row] + q) * n) % N);
                        re += a * cs[2 * m] - b * cs[2 * m + 1];
                        im += a * cs[2 * m + 1] + b * cs[2 * m];
                    }
                    dst[(size_t)n * cols + j] = (float)re;
                    dst[(size_t)n * cols + bpo + j] = (float)im;
                }
            }
            o->colTiles = ct;
        }
        free(cs);
    }

    /* ---- 2:1 resampler taps: h_j = 0.5 * rollOff*sinc(rollOff*j/2) * kaiser(j), j < 32
     *      (resample_algorithm.c:546-634 with zeroNum 16, nbit 9, beta 8.5555046, rollOff 0.85) */
    {
        const float rollOff = 0.85f;
        float *kw = afx_window_kaiser(2 * 8192 + 1, 8.5555046f);
        const float step = (16.f - 0.f) / 8192;
        for (int j = 0; j < 32; j++) {
            const int m = 256 * j;
            float x = 0.f + m * step;
            x *= rollOff;
            const float px = (float)(x * M_PI);
            float sc = (fabsf(px) < 1e-9) ? 1.f : sinf(px) / px;
            sc *= rollOff;
            sc = sc * kw[8192 + m];
            o->taps[j] = sc * 0.5f; /* interpArr *= ratio */
        }
        free(kw);
    }

    /* ---- device constants */
    float *tw = afx_twiddle_table(N);
    float *ones = (float *)calloc((size_t)num, sizeof(float));
    for (int i = 0; i < num; i++) ones[i] = 1.f;
    st = afxdev_stream_create(&o->stream);
#define UP(dst, src, bytes)                                                        \
    if (st == AFX_OK) st = afxdev_malloc((void **)&(dst), (bytes));                \
    if (st == AFX_OK) st = afxdev_h2d((dst), (src), (bytes), o->stream)
    UP(o->dTwiddle, tw, sizeof(float) * (size_t)(N < 2 ? 2 : N));
    UP(o->dKTaps, kTaps, sizeof(float) * 2 * (size_t)(total + 1));
    UP(o->dKStart, kStart, sizeof(int) * (size_t)rowsTotal);
    UP(o->dKLen, kLen, sizeof(int) * (size_t)rowsTotal);
    UP(o->dKOff, kOff, sizeof(int) * (size_t)rowsTotal);
    UP(o->dScaleOn, o->sLenArr, sizeof(float) * (size_t)num);
    UP(o->dScaleOff, ones, sizeof(float) * (size_t)num);
    if (o->colTiles) {
        UP(o->dTimeKernel, timeKernel, timeKernelBytes);
    }
    if (o->colTiles == 1 && N == 512 && !afxdev_cqt_f32()) {
        /* f16 (hi, lo) words of the image for the f16 matrix-core kernel (afx_cqt_f16.hip) */
        const int groups = rowsTotal / bpo;
        const size_t perGroup = (size_t)2 * (N / 16) * 64 * 8;
        unsigned short *kh = (unsigned short *)calloc((size_t)groups * perGroup, sizeof(unsigned short));
        float *cm = (float *)calloc((size_t)groups * 32, sizeof(float));
        if (kh && cm) {
            for (int g = 0; g < groups; g++)
                afx_cqt_time_kernel_f16(timeKernel + (size_t)g * N * 32, N, kh + (size_t)g * perGroup, cm + (size_t)g * 32);
            UP(o->dTimeKernelH, kh, sizeof(unsigned short) * (size_t)groups * perGroup);
            UP(o->dColMul, cm, sizeof(float) * (size_t)groups * 32);
            {   /* the resampler's tap table of the one-launch ladder (k_cqt_pyramid) */
                unsigned short dt[AFX_CQT_PYR_TAB_HALFS];
                afx_cqt_dec_table(o->taps, dt);
                UP(o->dDecTab, dt, sizeof(dt));
                if (st == AFX_OK) st = afxdev_stream_sync(o->stream); /* dt leaves scope */
            }
            /* the uploads above are asynchronous on o->stream: synced below before the host copies are freed */
            if (st == AFX_OK) st = afxdev_stream_sync(o->stream);
        }
        free(kh);
        free(cm);
    }
#undef UP
    if (st == AFX_OK) st = afxdev_stream_sync(o->stream);
    free(tw);
    free(ones);
    free(kre);
    free(kim);
    free(kStart);
    free(kLen);
    free(kOff);
    free(kTaps);
    free(lenTop);
    free(timeKernel);
    if (st != AFX_OK) {
        cqtObj_free(o);
        return st;
    }
    *cqtObj = o;
    return 0;
}

int cqtObj_calTimeLength(CQTObj o, int dataLength) {
    if (!o) return 0;
    if (o->isContinue) { /* cqt_algorithm.c:281-288: whole frames of tail + new samples */
        const long long total = (long long)dataLength + o->tailLength;
        return total < o->fftLength ? 0 : (int)((total - o->fftLength) / o->slideLength + 1);
    }
    if (dataLength <= 0) return 0;
    return dataLength / o->slideLength + 1; /* padded framing, cqt_algorithm.c:289-297 */
}

/* Streaming object: the samples left over by the previous calls followed by the new ones, the frames they hold, and
 * the new tail (_cqtObj_dealData, cqt_algorithm.c:345-456; __calTimeAndTailLen :309-327).  Returns the number of
 * frames (0: not a whole frame yet -- everything went to the tail), < 0 on allocation failure; *valid / *validLength
 * = the assembled signal.  (A negative tail -- hop > fftLength -- is that many samples of the next call to skip.) */
static int cqt_stream_take(CQTObj o, const float *data, int dataLength, const float **valid, int *validLength) {
    const int N = o->fftLength, hop = o->slideLength;
    const long long total = (long long)o->tailLength + dataLength;
    *valid = NULL;
    *validLength = 0;
    if (total < N) {
        if (total > 0) {
            if (o->tailLength >= 0) memcpy(o->tailData + o->tailLength, data, sizeof(float) * (size_t)dataLength);
            else memcpy(o->tailData, data - o->tailLength, sizeof(float) * (size_t)total);
        }
        o->tailLength = (int)total;
        return 0;
    }
    const int frames = (int)((total - N) / hop + 1);
    const int tailLen = (int)((total - N) % hop) + (N - hop);
    if ((size_t)total + (size_t)N > o->capValid) {
        float *p = (float *)realloc(o->validData, sizeof(float) * ((size_t)total + (size_t)N));
        if (!p) return -1;
        o->validData = p;
        o->capValid = (size_t)total + (size_t)N;
    }
    int vl = 0;
    if (o->tailLength < 0) {
        vl = dataLength + o->tailLength;
        memcpy(o->validData, data - o->tailLength, sizeof(float) * (size_t)vl);
    } else {
        if (o->tailLength > 0) memcpy(o->validData, o->tailData, sizeof(float) * (size_t)o->tailLength);
        vl = o->tailLength;
        memcpy(o->validData + vl, data, sizeof(float) * (size_t)dataLength);
        vl += dataLength;
    }
    if (tailLen > 0) memcpy(o->tailData, o->validData + (vl - tailLen), sizeof(float) * (size_t)tailLen);
    o->tailLength = tailLen;
    *valid = o->validData;
    *validLength = vl;
    return frames;
}

int cqtObj_getFFTLength(CQTObj o) { return o ? o->fftLength : 0; }
float *cqtObj_getFreBandArr(CQTObj o) { return o ? o->freBandArr : NULL; }

void cqtObj_setScale(CQTObj o, int flag) {
    if (o) o->isScale = flag;
}

/* ---- the default ladder in one launch (afx_cqt_f16.hip: k_cqt_pyramid) ----
 * N = 512, 12 bins per octave, seven octaves, hop 128, one image for all octaves, f16 matrix-core plan, centre
 * padding.  AFX_CQT_PYRAMID=0 (read when the object is created) keeps the per-octave launches. */
static int cqt_pyramid_ok(CQTObj o, int dataLength) {
    return !o->noPyramid && !afxdev_no_fused() && o->dTimeKernelH && o->dColMul && o->dDecTab && o->colTiles == 1 && o->radix2Exp == 9 &&
           o->binPerOctave == 12 && o->octaveNum == AFX_CQT_PYR_LEVELS && o->slideLength == 128 && !o->isContinue &&
           !o->vFlag && dataLength > 0 && dataLength <= (1 << 28) &&
           afxk_cqt_pyramid_plan(1, dataLength / 128 + 1, 0, NULL, NULL) > 0; /* (0: a device layer without the kernel) */
}

/* dX + b*xStride (b < batch) -> dRe/dIm [batch][T, num] (+ dChroma [batch][T, 12] when cn == 12); asynchronous on `stream` */
static int cqt_run_pyramid(CQTObj o, const float *dX, int batch, int dataLength, long long xStride, float *dRe,
                           float *dIm, float *dChroma, int isMag, int nrm, void *stream) {
    AfxCqtPyramidArgs a;
    memset(&a, 0, sizeof(a));
    const int T = dataLength / o->slideLength + 1;
    const int wgs = afxk_cqt_pyramid_plan(batch, T, o->pyrTiles, &a.chunksPerClip, &a.tilesPerChunk);
    int st = afxdev_reserve((void **)&o->dRing, &o->capRing, sizeof(float) * (size_t)wgs * AFX_CQT_PYR_RING_FLOATS);
    if (st != AFX_OK) return st;
    a.x = dX;
    a.xStride = xStride;
    a.batch = batch;
    a.timeLength = T;
    a.num = o->num;
    int len = dataLength, hop = o->slideLength;
    for (int k = 0; k < AFX_CQT_PYR_LEVELS; k++) {
        const int frames = len / hop + 1;
        a.len[k] = len;
        a.valid[k] = len - (frames > 1 ? len % hop : 0);          /* stft_algorithm.c:838-843 */
        a.octScale[k] = k == 0 ? 1.f : sqrtf((float)(1 << k));    /* dLenArr, cqt_algorithm.c:1218-1221 */
        len = (int)floorf(len * 0.5f);                            /* resampleObj_calDataLength */
        hop /= 2;
    }
    a.timeKernelH = o->dTimeKernelH;
    a.colMul = o->dColMul;
    a.scale = o->isScale ? o->dScaleOn : o->dScaleOff;
    a.outRe = dRe;
    a.outIm = dIm;
    a.outStride = (long long)T * o->num;
    a.ring = o->dRing;
    a.decTab = o->dDecTab;
    a.decMul = (float)(ldexp(1.0, -15) / (double)sqrtf(0.5f));
    if (o->pyrTiming) { /* -DAFX_EXPERIMENTS builds only: the instrumented instantiation */
        const size_t bytes = sizeof(unsigned long long) * AFX_CQT_PYR_MAX_WGS * 11 * 8;
        if (!o->dTiming) {
            st = afxdev_malloc((void **)&o->dTiming, bytes);
            if (st == AFX_OK) st = afxdev_memset(o->dTiming, 0, bytes, stream);
            if (st != AFX_OK) return st;
        }
        a.timing = o->dTiming;
    }
    a.chroma = dChroma;
    a.chromaMag = isMag;
    a.chromaNorm = nrm;
    if (dChroma) { /* class of bin j of an octave: the row of the folding matrix that holds it */
        for (int j = 0; j < 12; j++) {
            a.chromaClass[j] = -1;
            for (int c = 0; c < 12; c++)
                for (int q = o->foldLists.start[c]; q < o->foldLists.start[c + 1]; q++)
                    if (o->foldLists.bins[q] == j) a.chromaClass[j] = c;
            if (a.chromaClass[j] < 0) return AFX_ERR_UNSUPPORTED;
        }
    }
    return afxk_cqt_pyramid(&a, stream);
}

/* test hook: the level rings of the first `wgs` workgroups as the last k_cqt_pyramid launch left them
 * (AFX_CQT_PYR_RING_FLOATS floats each: levels 1 ... 6 at offsets 0, 8192, 12288, 14336, 15360, 16384; sample p of a
 * level at p mod the ring's size) -- tests/test_cqt_pyramid.py (device) and tests/emu/emulated_cqt_rings.py (emulated) check them against the resampler in float64 */
int afx_cqt_pyramid_rings(CQTObj o, float *host, int wgs) {
    if (!o || !o->dRing || !host || wgs <= 0) return 0;
    const size_t bytes = sizeof(float) * (size_t)wgs * AFX_CQT_PYR_RING_FLOATS;
    if (bytes > o->capRing) return 0;
    if (o->lastUsed) afxdev_stream_sync(o->lastStream);
    if (afxdev_d2h(host, o->dRing, bytes, o->stream) != AFX_OK || afxdev_stream_sync(o->stream) != AFX_OK) return 0;
    return wgs;
}

/* AFX_CQT_PYR_TIMING=1: copies out (and clears) the phase cycles the instrumented k_cqt_pyramid accumulated:
 * [256 workgroups][11 waves][8 phases]; returns the words copied, 0 when nothing was recorded (tools/pyr_phases.py) */
int afx_cqt_pyramid_timing(CQTObj o, unsigned long long *host) {
    if (!o || !o->dTiming || !host) return 0;
    const size_t bytes = sizeof(unsigned long long) * AFX_CQT_PYR_MAX_WGS * 11 * 8;
    if (o->lastUsed) afxdev_stream_sync(o->lastStream);
    if (afxdev_d2h(host, o->dTiming, bytes, o->stream) != AFX_OK || afxdev_stream_sync(o->stream) != AFX_OK) return 0;
    afxdev_memset(o->dTiming, 0, bytes, o->stream);
    afxdev_stream_sync(o->stream);
    return AFX_CQT_PYR_MAX_WGS * 11 * 8;
}

/* The octave recursion on HBM-resident clips: dX + b*xStride (b < batch, dataLength
 * samples each) -> dRe/dIm [batch][T, num].  dSig[0/1] hold the decimated signals of all
 * clips (pitch = dataLength/2 samples).  Asynchronous on `stream`. */
static int cqt_run_device(CQTObj o, const float *dX, int batch, int dataLength, long long xStride,
                          float *dRe, float *dIm, void *stream) {
    /* streaming objects frame from sample 0 (right padding) and keep whole frames only (cqt_algorithm.c:923-928) */
    const int T = o->isContinue ? (dataLength - o->fftLength) / o->slideLength + 1 : dataLength / o->slideLength + 1;
    const long long pitch = ((long long)dataLength / 2 + 3) & ~3LL;
    int st = AFX_OK;
    /* The decimation chain (signal of octave k from octave k+1: memory / latency bound) does not depend on
     * the octave products (matrix-core bound): it runs ahead on a side stream, every level in its own
     * slice of dSig[0] (pitch, pitch/2, ... samples per clip: < 2 pitch in all), and the octave kernel of a
     * level waits only for the decimation that produced its input (round 2: -0.25 ms per cfg-5 step). */
    const int overlap = o->octaveNum > 1;
    void *side = NULL;
    if (overlap) {
        side = o->stream != stream ? o->stream : o->stream2;
        if (!side) {
            st = afxdev_stream_create(&o->stream2);
            side = o->stream2;
        }
        if (st != AFX_OK) return st;
    }
    if (o->octaveNum > 1) {
        /* level offsets inside dSig[0]: level 1 at 0, level k at sum of the pitches before it */
        size_t total = 0;
        long long p = pitch;
        for (int k = 1; k < o->octaveNum; k++) {
            total += (size_t)p * batch;
            p = ((p / 2) + 3) & ~3LL;
        }
        st = afxdev_reserve((void **)&o->dSig[0], &o->capSig[0], sizeof(float) * total);
    }
    if (st != AFX_OK) return st;

    AfxCqtOctaveArgs a;
    memset(&a, 0, sizeof(a));
    a.timeLength = T;
    a.radix2Exp = o->radix2Exp;
    a.twiddle = o->dTwiddle;
    a.kStart = o->dKStart;
    a.kLen = o->dKLen;
    a.kOff = o->dKOff;
    a.kTaps = o->dKTaps;
    a.rows = o->binPerOctave;
    a.scale = o->isScale ? o->dScaleOn : o->dScaleOff;
    a.num = o->num;
    a.outRe = dRe;
    a.outIm = dIm;
    a.batch = batch;
    a.outStride = (long long)T * o->num;
    a.rightPad = o->isContinue;

    const float *cur = dX;
    long long curStride = xStride;
    int len = dataLength, hop = o->slideLength;
    long long levelPitch = pitch;
    size_t levelOff = 0;
    /* the side stream starts behind everything the caller's stream has enqueued: the input is ready and
     * the previous call's octave kernels no longer read the signal slices */
    if (side) st = afxdev_stream_wait_stream(side, stream);
    for (int oct = o->octaveNum - 1; oct >= 0 && st == AFX_OK; oct--) {
        const int k = o->octaveNum - 1 - oct; /* decimations so far */
        const int frames = len / hop + 1;
        /* the next level's signal first (side stream when overlapping), then this level's product */
        const int next = (int)floorf(len * 0.5f); /* resampleObj_calDataLength */
        float *dNext = o->dSig[0] ? o->dSig[0] + levelOff : NULL;
        if (oct > 0)
            st = afxk_cqt_decimate(cur, len, curStride, dNext, next, levelPitch, batch, o->taps, sqrtf(0.5f),
                                   side ? side : stream);
        if (st != AFX_OK) break;
        a.x = cur;
        a.xStride = curStride;
        a.hop = hop;
        a.validLength = len - (frames > 1 ? len % hop : 0); /* stft_algorithm.c:838-843 */
        a.rowBase = o->vFlag ? oct * o->binPerOctave : 0;
        a.colBase = oct * o->binPerOctave;
        a.octScale = k == 0 ? 1.f : sqrtf((float)(1 << k)); /* dLenArr, cqt_algorithm.c:1218-1221 */
        a.colTiles = o->colTiles;
        a.timeKernel = o->colTiles ? o->dTimeKernel + (size_t)(o->vFlag ? oct : 0) * o->fftLength * o->colTiles * 32
                                   : NULL;
        a.timeKernelH = o->dTimeKernelH ? o->dTimeKernelH + (size_t)(o->vFlag ? oct : 0) * 2 * (o->fftLength / 16) * 64 * 8
                                        : NULL;
        a.colMul = o->dColMul ? o->dColMul + (size_t)(o->vFlag ? oct : 0) * 32 : NULL;
        st = afxk_cqt_octave(&a, stream);
        if (st != AFX_OK || oct == 0) break;
        /* the next octave kernel reads dNext: wait for the decimation enqueued above (and only for it:
         * later ones are not enqueued yet) */
        if (side) st = afxdev_stream_wait_stream(stream, side);
        cur = dNext;
        curStride = levelPitch;
        levelOff += (size_t)levelPitch * batch;
        levelPitch = ((levelPitch / 2) + 3) & ~3LL;
        len = next;
        hop /= 2;
    }
    return st;
}

void cqtObj_cqt(CQTObj o, float *dataArr, int dataLength, float *mRealArr, float *mImageArr) {
    AFX_ENTER(o);
    if (!o) {
        afxdev_set_error("cqtObj_cqt: NULL object");
        return;
    }
    if (!dataArr || dataLength <= 0) return;
    if (!o->isContinue && (!mRealArr || !mImageArr)) return;
    const float *src = dataArr;
    int T = dataLength / o->slideLength + 1;
    if (o->isContinue) { /* tail of the previous calls + these samples; the rest waits for the next call */
        T = cqt_stream_take(o, dataArr, dataLength, &src, &dataLength);
        if (T < 0) {
            fail(o, AFX_ERR_NOMEM, "cqtObj_cqt");
            return;
        }
        o->timeLength = T;
        if (T == 0) return; /* (the samples are kept; nothing is written) */
        if (!mRealArr || !mImageArr) {
            afxdev_set_error("cqtObj_cqt: %d frames are due but an output pointer is NULL", T);
            fail(o, AFX_ERR_ARG, "cqtObj_cqt");
            return;
        }
    }
    const size_t outB = sizeof(float) * (size_t)T * o->num;
    int st = AFX_OK;
    if (o->lastUsed && o->lastStream != o->stream) st = afxdev_stream_sync(o->lastStream);
    o->lastUsed = 0;
    if (st == AFX_OK) st = afxdev_reserve((void **)&o->dOut, &o->capOut, 2 * outB);
    if (st == AFX_OK) st = afxdev_reserve((void **)&o->dX, &o->capX, sizeof(float) * (size_t)dataLength);
    if (st == AFX_OK) st = afxdev_h2d(o->dX, src, sizeof(float) * (size_t)dataLength, o->stream);
    float *dRe = o->dOut, *dIm = o->dOut + (size_t)T * o->num;
    if (st == AFX_OK)
        st = cqt_pyramid_ok(o, dataLength) ? cqt_run_pyramid(o, o->dX, 1, dataLength, dataLength, dRe, dIm, NULL, 0, 0, o->stream)
                                           : cqt_run_device(o, o->dX, 1, dataLength, dataLength, dRe, dIm, o->stream);
    if (st == AFX_OK) st = afxdev_d2h(mRealArr, dRe, outB, o->stream);
    if (st == AFX_OK) st = afxdev_d2h(mImageArr, dIm, outB, o->stream);
    if (st == AFX_OK) st = afxdev_stream_sync(o->stream);
    o->timeLength = T;
    if (st != AFX_OK) fail(o, st, "cqtObj_cqt");
}

/* Clips per pass of the octave ladder.  Every octave kernel writes its 12 of the num columns of each output
 * row (48-byte pieces of 336-byte rows), so a row is completed by seven launches; when the pass's output
 * (2 planes x clips x T x num floats) stays within reach of the memory-side cache those pieces merge before
 * they reach HBM, beyond it the partial-line writes throttle the store path (measured on cfg 5, 125 clips:
 * one pass of 868 MB 2.07 ms per step, passes of 64 / 50 / 42 / 32 / 16 clips 1.86 / 1.93 / 1.98 / 2.06 / 2.40 ms,
 * profiles/r02_cqt_f16_steps.txt).  Default: the fewest passes of <= 448 MB of output each, of equal size
 * (125 clips of cfg 5 -> 63 + 62); AFX_CQT_CHUNK=<clips> overrides. */
int afx_cqt_pass_clips(long long rowFloats, int batch) {
    const char *e = getenv("AFX_CQT_CHUNK");
    if (e && atoi(e) > 0) return atoi(e) > 32768 ? 32768 : atoi(e);
    const double perClip = 8.0 * (double)rowFloats;
    long long c = (long long)(448.0 * 1024 * 1024 / (perClip > 0 ? perClip : 1));
    if (c < 8) c = 8;
    if (c > 32768) c = 32768;
    if (batch > c) { /* equal passes instead of full ones and a small remainder */
        const long long passes = (batch + c - 1) / c;
        c = (batch + passes - 1) / passes;
    }
    return (int)c;
}

static int cqt_chunk_clips(CQTObj o, int T, int batch) { return afx_cqt_pass_clips((long long)T * o->num, batch); }

/* clips already in HBM, results left in HBM (include/afx_batch.h) */
int cqtObj_cqtBatchDevice(CQTObj o, const float *dData, int batch, int dataLength,
                          long long clipStride, float *dReal, float *dImag, void *hipStream) {
    AFX_ENTER(o);
    if (o && o->isContinue) { /* a streaming object carries one signal's tail from call to call */
        afxdev_set_error("cqtObj_cqtBatchDevice: the object was created with isContinue = 1; streaming objects take cqtObj_cqt");
        return AFX_ERR_UNSUPPORTED;
    }
    if (!o || !dData || !dReal || !dImag || batch <= 0 || dataLength <= 0 || clipStride < dataLength) {
        afxdev_set_error("cqtObj_cqtBatchDevice: bad argument");
        return AFX_ERR_ARG;
    }
    int st = AFX_OK;
    const int T = dataLength / o->slideLength + 1;
    /* scratch is shared between calls: order this call after the previous one's stream */
    if (o->lastStream != hipStream && o->lastUsed) st = afxdev_stream_sync(o->lastStream);
    if (st == AFX_OK && cqt_pyramid_ok(o, dataLength)) { /* one launch for all clips: nothing to merge between passes */
        st = cqt_run_pyramid(o, dData, batch, dataLength, clipStride, dReal, dImag, NULL, 0, 0, hipStream);
    } else {
        const int chunk = cqt_chunk_clips(o, T, batch);
        for (int b0 = 0; b0 < batch && st == AFX_OK; b0 += chunk) {
            const int nb = batch - b0 < chunk ? batch - b0 : chunk;
            st = cqt_run_device(o, dData + (long long)b0 * clipStride, nb, dataLength, clipStride,
                                dReal + (long long)b0 * T * o->num, dImag + (long long)b0 * T * o->num,
                                hipStream);
        }
    }
    o->lastStream = hipStream;
    o->lastUsed = 1;
    o->timeLength = T;
    if (st != AFX_OK) fail(o, st, "cqtObj_cqtBatchDevice");
    return st;
}

int cqtObj_cqtBatch(CQTObj o, const float *dataArr, int batch, int dataLength, float *mRealArr,
                    float *mImageArr) {
    AFX_ENTER(o);
    if (o && o->isContinue) { /* a streaming object carries one signal's tail from call to call */
        afxdev_set_error("cqtObj_cqtBatch: the object was created with isContinue = 1; streaming objects take cqtObj_cqt");
        return AFX_ERR_UNSUPPORTED;
    }
    if (!o || !dataArr || !mRealArr || !mImageArr || batch <= 0 || dataLength <= 0) {
        afxdev_set_error("cqtObj_cqtBatch: bad argument");
        return AFX_ERR_ARG;
    }
    const int T = dataLength / o->slideLength + 1;
    const size_t inB = sizeof(float) * (size_t)batch * dataLength;
    const size_t outB = sizeof(float) * (size_t)batch * T * o->num;
    int st = afxdev_reserve((void **)&o->dOut, &o->capOut, 2 * outB);
    if (st == AFX_OK) st = afxdev_reserve((void **)&o->dX, &o->capX, inB);
    if (st == AFX_OK) st = afxdev_h2d(o->dX, dataArr, inB, o->stream);
    float *dRe = o->dOut, *dIm = o->dOut + (size_t)batch * T * o->num;
    if (st == AFX_OK) st = cqtObj_cqtBatchDevice(o, o->dX, batch, dataLength, dataLength, dRe, dIm, o->stream);
    if (st == AFX_OK) st = afxdev_d2h(mRealArr, dRe, outB, o->stream);
    if (st == AFX_OK) st = afxdev_d2h(mImageArr, dIm, outB, o->stream);
    if (st == AFX_OK) st = afxdev_stream_sync(o->stream);
    if (st != AFX_OK) fail(o, st, "cqtObj_cqtBatch");
    return st;
}

/* 0/1 folding matrix bins -> chroma (chroma_filterBank.c:176-264), host side */
unsigned char *afx_chroma_fold(int chromaNum, int num, int bpo, float minFre) {
    unsigned char *tmp = (unsigned char *)calloc((size_t)chromaNum * num, 1);
    unsigned char *out = (unsigned char *)calloc((size_t)chromaNum * num, 1);
    if (!tmp || !out) {
        free(tmp);
        free(out);
        return NULL;
    }
    int n = bpo / chromaNum;
    const int offset = (int)ceilf((float)(n / 2.0));
    const int sub = n - offset;
    int midi = (int)roundf((float)(12 * log2(minFre / 440) + 69));
    midi = midi % 12;
    if (midi > 6) midi = 12 - midi;
    int start = 0;
    for (int i = 0; i < chromaNum; i++) {
        if (i) start = offset + (i - 1) * n;
        for (int j = 0; j < num; j++) {
            const int mod = j % bpo;
            if (i != 0) {
                if (mod >= start && mod < start + n) tmp[(size_t)i * num + j] = 1;
            } else {
                if (mod >= 0 && mod < offset) tmp[j] = 1;
                if (sub && mod >= bpo - sub && mod < bpo) tmp[j] = 1;
            }
        }
    }
    if (midi) { /* row rotation; n is chromaNum/bpo here, i.e. 1 only when they are equal */
        n = chromaNum / bpo;
        int k = 0;
        for (int i = midi * n; i < chromaNum; i++, k++) memcpy(out + (size_t)k * num, tmp + (size_t)i * num, (size_t)num);
        k = chromaNum - midi * n;
        for (int i = 0; i < midi * n; i++, k++) memcpy(out + (size_t)k * num, tmp + (size_t)i * num, (size_t)num);
        free(tmp);
        return out;
    }
    free(out);
    return tmp;
}

/* 0/1 folding matrix -> per-class bin lists (ascending bins: the order of the matrix product); -1 when the
 * matrix does not fit the list form (more than 64 classes, more than 256 entries) */
int afx_chroma_lists(const unsigned char *fold, int chromaNum, int num, AfxChromaLists *out) {
    if (!fold || !out || chromaNum < 1 || chromaNum > 64 || num < 1 || num > 255) return -1;
    int n = 0;
    memset(out, 0, sizeof(*out));
    for (int c = 0; c < chromaNum; c++) {
        out->start[c] = (unsigned short)n;
        for (int j = 0; j < num; j++) {
            if (!fold[(size_t)c * num + j]) continue;
            if (n >= 256) return -1;
            out->bins[n++] = (unsigned char)j;
        }
    }
    for (int c = chromaNum; c <= 64; c++) out->start[c] = (unsigned short)n;
    return 0;
}

/* resolve the optional chroma parameters; upload the folding matrix when chromaNum changed */
static int chroma_prepare(CQTObj o, int *chromaNum, SpectralDataType *dataType,
                          ChromaDataNormalType *normType, int *cnOut, int *isMag, int *nrmOut) {
    int cn = 12;
    SpectralDataType dt = SpectralData_Power;
    ChromaDataNormalType nt = ChromaDataNormal_Max;
    if (chromaNum) cn = *chromaNum;
    if (dataType) dt = *dataType;
    if (normType) nt = *normType;
    if (cn <= 0 || cn > o->binPerOctave || o->binPerOctave % cn != 0) {
        printf("chromaNum and binPerOctave not map!!!");
        return AFX_ERR_ARG;
    }
    int st = AFX_OK;
    if (cn != o->foldChromaNum) {
        unsigned char *fold = afx_chroma_fold(cn, o->num, o->binPerOctave, o->minFre);
        if (!fold) return AFX_ERR_NOMEM;
        if (o->lastUsed) afxdev_stream_sync(o->lastStream); /* a launch may still read the old one */
        afxdev_free(o->dFold);
        o->dFold = NULL;
        st = afxdev_malloc((void **)&o->dFold, (size_t)cn * o->num);
        if (st == AFX_OK) st = afxdev_h2d(o->dFold, fold, (size_t)cn * o->num, o->stream);
        if (st == AFX_OK) st = afxdev_stream_sync(o->stream);
        o->haveLists = fold && afx_chroma_lists(fold, cn, o->num, &o->foldLists) == 0;
        free(fold);
        if (st == AFX_OK) o->foldChromaNum = cn;
    }
    int nrm = 0;
    if (nt == ChromaDataNormal_Max) nrm = 1;
    else if (nt == ChromaDataNormal_Min) nrm = 2;
    else if (nt == ChromaDataNormal_P2) nrm = 3;
    else if (nt != ChromaDataNormal_None) nrm = 4;
    *cnOut = cn;
    *isMag = dt == SpectralData_Mag;
    *nrmOut = nrm;
    return st;
}

void cqtObj_chroma(CQTObj o, int *chromaNum, SpectralDataType *dataType,
                   ChromaDataNormalType *normType, float *mRealArr, float *mImageArr,
                   float *mDataArr) {
    AFX_ENTER(o);
    if (!o) {
        afxdev_set_error("cqtObj_chroma: NULL object");
        return;
    }
    int cn, isMag, nrm;
    if (chroma_prepare(o, chromaNum, dataType, normType, &cn, &isMag, &nrm) == AFX_ERR_ARG) return;
    const int T = o->timeLength;
    if (T <= 0 || !mRealArr || !mImageArr || !mDataArr) return;
    int st = o->foldChromaNum == cn ? AFX_OK : AFX_ERR_HIP;
    const size_t inB = sizeof(float) * (size_t)T * o->num;
    if (st == AFX_OK) st = afxdev_reserve((void **)&o->dIn, &o->capIn, 2 * inB + sizeof(float) * (size_t)T * cn);
    float *dRe = o->dIn, *dIm = o->dIn + (size_t)T * o->num, *dC = o->dIn + 2 * (size_t)T * o->num;
    if (st == AFX_OK) st = afxdev_h2d(dRe, mRealArr, inB, o->stream);
    if (st == AFX_OK) st = afxdev_h2d(dIm, mImageArr, inB, o->stream);
    if (st == AFX_OK)
        st = afxk_cqt_chroma(dRe, dIm, T, o->num, o->dFold, o->haveLists ? &o->foldLists : NULL, cn, isMag, nrm, dC, o->stream);
    if (st == AFX_OK) st = afxdev_d2h(mDataArr, dC, sizeof(float) * (size_t)T * cn, o->stream);
    if (st == AFX_OK) st = afxdev_stream_sync(o->stream);
    if (st != AFX_OK) fail(o, st, "cqtObj_chroma");
}

/* chroma of `rows` = batch*T HBM-resident CQT frames (include/afx_batch.h) */
int cqtObj_chromaBatchDevice(CQTObj o, int *chromaNum, SpectralDataType *dataType,
                             ChromaDataNormalType *normType, const float *dReal,
                             const float *dImag, long long rows, float *dData, void *hipStream) {
    AFX_ENTER(o);
    if (!o || !dReal || !dImag || !dData || rows <= 0) {
        afxdev_set_error("cqtObj_chromaBatchDevice: bad argument");
        return AFX_ERR_ARG;
    }
    int cn, isMag, nrm;
    int st = chroma_prepare(o, chromaNum, dataType, normType, &cn, &isMag, &nrm);
    if (st == AFX_OK)
        st = afxk_cqt_chroma(dReal, dImag, rows, o->num, o->dFold, o->haveLists ? &o->foldLists : NULL, cn, isMag, nrm, dData, hipStream);
    if (st == AFX_OK) {
        o->lastStream = hipStream;
        o->lastUsed = 1;
    } else if (st != AFX_ERR_ARG) {
        fail(o, st, "cqtObj_chromaBatchDevice");
    }
    return st;
}

/* CQT and its chroma in one call (include/afx_batch.h): pass by pass, so that the chroma kernel reads the
 * pass's CQT rows while they are still cached */
int cqtObj_cqtChromaBatchDevice(CQTObj o, const float *dData, int batch, int dataLength, long long clipStride,
                                float *dReal, float *dImag, int *chromaNum, SpectralDataType *dataType,
                                ChromaDataNormalType *normType, float *dChroma, void *hipStream) {
    AFX_ENTER(o);
    if (o && o->isContinue) { /* a streaming object carries one signal's tail from call to call */
        afxdev_set_error("cqtObj_cqtChromaBatchDevice: the object was created with isContinue = 1; streaming objects take cqtObj_cqt");
        return AFX_ERR_UNSUPPORTED;
    }
    if (!o || !dData || !dReal || !dImag || !dChroma || batch <= 0 || dataLength <= 0 || clipStride < dataLength) {
        afxdev_set_error("cqtObj_cqtChromaBatchDevice: bad argument");
        return AFX_ERR_ARG;
    }
    int cn, isMag, nrm;
    int st = chroma_prepare(o, chromaNum, dataType, normType, &cn, &isMag, &nrm);
    if (st == AFX_ERR_ARG) return st;
    const int T = dataLength / o->slideLength + 1;
    if (st == AFX_OK && o->lastStream != hipStream && o->lastUsed) st = afxdev_stream_sync(o->lastStream);
    if (st == AFX_OK && cqt_pyramid_ok(o, dataLength)) {
        /* 12 classes of 12 bins per octave: the sums travel with the rows through the one launch */
        const int fused = cn == 12 && o->haveLists;
        st = cqt_run_pyramid(o, dData, batch, dataLength, clipStride, dReal, dImag, fused ? dChroma : NULL, isMag, nrm, hipStream);
        if (st == AFX_OK && !fused)
            st = afxk_cqt_chroma(dReal, dImag, (long long)batch * T, o->num, o->dFold, o->haveLists ? &o->foldLists : NULL, cn, isMag,
                                 nrm, dChroma, hipStream);
    } else {
        const int chunk = cqt_chunk_clips(o, T, batch);
        for (int b0 = 0; b0 < batch && st == AFX_OK; b0 += chunk) {
            const int nb = batch - b0 < chunk ? batch - b0 : chunk;
            float *re = dReal + (long long)b0 * T * o->num, *im = dImag + (long long)b0 * T * o->num;
            st = cqt_run_device(o, dData + (long long)b0 * clipStride, nb, dataLength, clipStride, re, im, hipStream);
            if (st == AFX_OK)
                st = afxk_cqt_chroma(re, im, (long long)nb * T, o->num, o->dFold, o->haveLists ? &o->foldLists : NULL, cn, isMag, nrm,
                                     dChroma + (long long)b0 * T * cn, hipStream);
        }
    }
    o->lastStream = hipStream;
    o->lastUsed = 1;
    o->timeLength = T;
    if (st != AFX_OK) fail(o, st, "cqtObj_cqtChromaBatchDevice");
    return st;
}

void cqtObj_cqcc(CQTObj o, float *mDataArr1, int ccNum, CepstralRectifyType *rectifyType,
                 float *mDataArr2) {
    AFX_ENTER(o);
    if (!o) {
        afxdev_set_error("cqtObj_cqcc: NULL object");
        return;
    }
    const int T = o->timeLength;
    if (ccNum > o->num || ccNum < 1 || T <= 0 || !mDataArr1 || !mDataArr2) return;
    int st = AFX_OK;
    if (!o->dDct) {
        float *d = afx_dct2_matrix(o->num, o->num);
        st = afxdev_malloc((void **)&o->dDct, sizeof(float) * (size_t)o->num * o->num);
        if (st == AFX_OK) st = afxdev_h2d(o->dDct, d, sizeof(float) * (size_t)o->num * o->num, o->stream);
        if (st == AFX_OK) st = afxdev_stream_sync(o->stream);
        free(d);
    }
    const size_t inB = sizeof(float) * (size_t)T * o->num, outB = sizeof(float) * (size_t)T * ccNum;
    if (st == AFX_OK) st = afxdev_reserve((void **)&o->dIn, &o->capIn, inB + outB + 64);
    float *dA = o->dIn, *dC = o->dIn + (((size_t)T * o->num + 3) & ~(size_t)3);
    if (st == AFX_OK) st = afxdev_h2d(dA, mDataArr1, inB, o->stream);
    const int pre = (rectifyType && *rectifyType == CepstralRectify_CubicRoot) ? AFX_MAP_CBRT : AFX_MAP_LOG10;
    if (st == AFX_OK) {
        if (afxk_cepstrum_supported(dA, o->num, ccNum))
            st = afxk_cepstrum(dA, T, o->num, o->dDct, ccNum, pre, dC, o->stream);
        else
            st = afxk_gemm_nt(dA, o->num, o->dDct, o->num, dC, ccNum, T, ccNum, o->num, pre,
                              AFX_MAP_NONE, 1.f, o->stream);
    }
    if (st == AFX_OK) st = afxdev_d2h(mDataArr2, dC, outB, o->stream);
    if (st == AFX_OK) st = afxdev_stream_sync(o->stream);
    if (st != AFX_OK) fail(o, st, "cqtObj_cqcc");
}

/* transform length and twiddles of cqhc / deconv (_cqtObj_dealDeconv, cqt_algorithm.c:783-843) */
static int deconv_prepare(CQTObj o) {
    if (o->dDevTw) return AFX_OK;
    const int M = afx_ceil_pow2(2 * o->num);
    o->devRadix = afx_log2_exact(M);
    float *tw = afx_twiddle_table(M);
    if (!tw) return AFX_ERR_NOMEM;
    int st = afxdev_malloc((void **)&o->dDevTw, sizeof(float) * (size_t)(M < 2 ? 2 : M));
    if (st == AFX_OK) st = afxdev_h2d(o->dDevTw, tw, sizeof(float) * (size_t)(M < 2 ? 2 : M), o->stream);
    if (st == AFX_OK) st = afxdev_stream_sync(o->stream);
    free(tw);
    return st;
}

void cqtObj_cqhc(CQTObj o, float *mDataArr1, int hcNum, float *mDataArr2) {
    AFX_ENTER(o);
    if (!o) {
        afxdev_set_error("cqtObj_cqhc: NULL object");
        return;
    }
    const int T = o->timeLength;
    if (T <= 0 || hcNum < 1 || !mDataArr1 || !mDataArr2) return;
    int st = deconv_prepare(o);
    const int M = 1 << o->devRadix;
    int *idx = (int *)malloc(sizeof(int) * (size_t)hcNum);
    if (!idx) st = AFX_ERR_NOMEM;
    for (int j = 0; j < hcNum && st == AFX_OK; j++) {
        int v = (int)roundf(o->binPerOctave * log2f((float)(j + 1))); /* cqt_algorithm.c:706 */
        if (v > M - 1) v = M - 1; /* the reference reads past its buffer here */
        idx[j] = v;
    }
    const size_t inB = sizeof(float) * (size_t)T * o->num, outB = sizeof(float) * (size_t)T * hcNum;
    const size_t idxOff = ((inB + outB) + 15) & ~(size_t)15;
    if (st == AFX_OK) st = afxdev_reserve((void **)&o->dIn, &o->capIn, idxOff + sizeof(int) * (size_t)hcNum);
    float *dA = o->dIn, *dC = o->dIn + (size_t)T * o->num;
    int *dIdx = (int *)((char *)o->dIn + idxOff);
    if (st == AFX_OK) st = afxdev_h2d(dA, mDataArr1, inB, o->stream);
    if (st == AFX_OK) st = afxdev_h2d(dIdx, idx, sizeof(int) * (size_t)hcNum, o->stream);
    if (st == AFX_OK)
        st = afxk_cqt_deconv(dA, T, o->num, o->devRadix, o->dDevTw, dIdx, hcNum, NULL, NULL, dC, o->stream);
    if (st == AFX_OK) st = afxdev_d2h(mDataArr2, dC, outB, o->stream);
    if (st == AFX_OK) st = afxdev_stream_sync(o->stream);
    free(idx);
    if (st != AFX_OK) fail(o, st, "cqtObj_cqhc");
}

/* mDataArr1 [T,num] magnitudes -> mDataArr2 timbre (formant), mDataArr3 pitch, both [T,num] */
void cqtObj_deconv(CQTObj o, float *mDataArr1, float *mDataArr2, float *mDataArr3) {
    AFX_ENTER(o);
    if (!o) {
        afxdev_set_error("cqtObj_deconv: NULL object");
        return;
    }
    const int T = o->timeLength;
    if (T <= 0 || !mDataArr1 || !mDataArr2 || !mDataArr3) return;
    int st = deconv_prepare(o);
    const size_t inB = sizeof(float) * (size_t)T * o->num;
    if (st == AFX_OK) st = afxdev_reserve((void **)&o->dIn, &o->capIn, 3 * inB);
    float *dA = o->dIn, *dT = o->dIn + (size_t)T * o->num, *dP = o->dIn + 2 * (size_t)T * o->num;
    if (st == AFX_OK) st = afxdev_h2d(dA, mDataArr1, inB, o->stream);
    if (st == AFX_OK)
        st = afxk_cqt_deconv(dA, T, o->num, o->devRadix, o->dDevTw, NULL, 0, dT, dP, NULL, o->stream);
    if (st == AFX_OK) st = afxdev_d2h(mDataArr2, dT, inB, o->stream);
    if (st == AFX_OK) st = afxdev_d2h(mDataArr3, dP, inB, o->stream);
    if (st == AFX_OK) st = afxdev_stream_sync(o->stream);
    if (st != AFX_OK) fail(o, st, "cqtObj_deconv");
}

void cqtObj_free(CQTObj o) {
    if (!o) return;
    if (o->stream2) afxdev_stream_sync(o->stream2); /* side stream of the device call */
    if (o->stream) afxdev_stream_sync(o->stream);
    afxdev_free(o->dTwiddle);
    afxdev_free(o->dKTaps);
    afxdev_free(o->dKStart);
    afxdev_free(o->dKLen);
    afxdev_free(o->dKOff);
    afxdev_free(o->dScaleOn);
    afxdev_free(o->dScaleOff);
    afxdev_free(o->dFold);
    afxdev_free(o->dTimeKernel);
    afxdev_free(o->dTimeKernelH);
    afxdev_free(o->dColMul);
    afxdev_free(o->dX);
    afxdev_free(o->dSig[0]);
    afxdev_free(o->dSig[1]);
    afxdev_free(o->dRing);
    afxdev_free(o->dTiming);
    afxdev_free(o->dDecTab);
    afxdev_free(o->dOut);
    afxdev_free(o->dIn);
    afxdev_free(o->dDct);
    afxdev_free(o->dDevTw);
    afxdev_stream_destroy(o->stream2);
    afxdev_stream_destroy(o->stream);
    free(o->freBandArr);
    free(o->sLenArr);
    free(o->tailData);
    free(o->validData);
    free(o);
}
