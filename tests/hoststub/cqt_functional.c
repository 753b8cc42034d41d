/* A FUNCTIONAL stand-in for the CQT launchers of audioflux_amd/csrc/hip/afx_device.h: every launcher does, in plain
 * double-precision loops, what its contract in afx_device.h says the kernel does with the arguments it is handed
 * ("device" memory is host memory here, tests/hoststub/gen_stub.py --functional-cqt supplies the rest of the device
 * layer).  With it the C host code of the CQT object -- plan, time-domain image and its f16 (hi, lo) words, column
 * multipliers, decimation chain and level table, passes, chroma classes -- runs end to end on the CPU and its
 * results are compared with the reference's golden vectors (tests/test_hoststub.py): a wrong pointer offset,
 * scale, octave order or class table in the glue of a launch path that has not been on hardware yet
 * (AFX_CQT_FUSED) shows up here as a parity failure instead of costing GPU minutes.
 * It says nothing about the kernels themselves.  Test infrastructure, never linked into the product. */
#define _USE_MATH_DEFINES
#define _GNU_SOURCE
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "afx_device.h"

/* launches seen, by kind (read by tests/hoststub/functional_cqt.py: which path a switch really took) */
int afx_functional_launches[4]; /* octave f16 | octave f32 | all-octave | chroma */

static double f16_value(unsigned short h) {
    const int e = (h >> 10) & 0x1f, m = h & 0x3ff;
    double v;
    if (e == 0) v = ldexp((double)m, -24);
    else if (e == 31) v = m ? NAN : INFINITY;
    else v = ldexp((double)(1024 + m), e - 25);
    return (h & 0x8000) ? -v : v;
}

/* y[i] = (sum_{j=0..31} h_j x[2i - j] + sum_{j=1..31} h_j x[2i + j]) / sqrtRatio, x = 0 outside [0, srcLen) */
int afxk_cqt_decimate(const float *x, int srcLen, long long xStride, float *y, int dstLen, long long yStride, int batch,
                      const float *taps32, float sqrtRatio, void *stream) {
    (void)stream;
    for (int b = 0; b < batch; b++) {
        const float *xb = x + b * xStride;
        float *yb = y + b * yStride;
        for (int i = 0; i < dstLen; i++) {
            double acc = 0;
            for (int j = 0; j < 32; j++) {
                const long long s = 2LL * i - j;
                if (s >= 0 && s < srcLen) acc += (double)taps32[j] * xb[s];
            }
            for (int j = 1; j < 32; j++) {
                const long long s = 2LL * i + j;
                if (s >= 0 && s < srcLen) acc += (double)taps32[j] * xb[s];
            }
            yb[i] = (float)(acc / sqrtRatio);
        }
    }
    return AFX_OK;
}

/* out[b][t][colBase + j] = octScale / scale[colBase + j] * sum_n x_b[t hop - N/2 + n] G[n][j] (Re), G[n][rows + j] (Im);
 * samples outside [0, validLength) are zero */
static void octave_product(const float *x, long long xStride, int validLength, int hop, int rightPad, int N, const double *G, int rows,
                           const float *scale, float octScale, int num, int colBase, float *outRe, float *outIm,
                           long long outStride, int batch, int timeLength) {
    double *acc = (double *)malloc(sizeof(double) * 2 * (size_t)rows);
    if (!acc) abort();
    for (int b = 0; b < batch; b++)
        for (int t = 0; t < timeLength; t++) {
            memset(acc, 0, sizeof(double) * 2 * (size_t)rows);
            for (int n = 0; n < N; n++) {
                const long long s = (long long)t * hop - (rightPad ? 0 : N / 2) + n;
                if (s < 0 || s >= validLength) continue;
                const double v = x[b * xStride + s];
                const double *g = G + (size_t)n * 2 * rows;
                for (int j = 0; j < 2 * rows; j++) acc[j] += v * g[j];
            }
            for (int j = 0; j < rows; j++) {
                const double m = (double)octScale / scale[colBase + j];
                outRe[b * outStride + (long long)t * num + colBase + j] = (float)(acc[j] * m);
                outIm[b * outStride + (long long)t * num + colBase + j] = (float)(acc[rows + j] * m);
            }
        }
    free(acc);
}

/* the (hi, lo) f16 words in fragment order -> G [N][2 rows] */
static double *image_from_words(const unsigned short *words, const float *colMul, int N, int rows) {
    double *G = (double *)calloc((size_t)N * 2 * rows, sizeof(double));
    if (!G) abort();
    const size_t plane = (size_t)(N / 16) * 64 * 8;
    for (int ks = 0; ks < N / 16; ks++)
        for (int lane = 0; lane < 64; lane++)
            for (int e = 0; e < 8; e++) {
                const int col = lane & 31, n = 16 * ks + 8 * (lane >> 5) + e;
                const size_t w = ((size_t)ks * 64 + lane) * 8 + e;
                const double v = (f16_value(words[w]) + f16_value(words[plane + w])) * colMul[col];
                if (col < 2 * rows) G[(size_t)n * 2 * rows + col] = v;
                else if (v != 0) abort(); /* padding columns of the image must be empty */
            }
    return G;
}

static int hop_is_f16(int hop) { return hop == 128 || hop == 64 || hop == 32 || hop == 16 || hop == 8 || hop == 4 || hop == 2; }

int afxk_cqt_octave_f16(const AfxCqtOctaveArgs *a, void *stream) {
    (void)stream;
    if (!a->timeKernelH || !a->colMul || a->colTiles != 1 || a->radix2Exp != 9 || !hop_is_f16(a->hop)) return AFX_ERR_UNSUPPORTED;
    if (2 * a->rows > 32) return AFX_ERR_UNSUPPORTED;
    afx_functional_launches[0]++;
    double *G = image_from_words(a->timeKernelH, a->colMul, 512, a->rows);
    octave_product(a->x, a->xStride, a->validLength, a->hop, a->rightPad, 512, G, a->rows, a->scale, a->octScale, a->num, a->colBase,
                   a->outRe, a->outIm, a->outStride, a->batch > 0 ? a->batch : 1, a->timeLength);
    free(G);
    return AFX_OK;
}

int afxk_cqt_octave(const AfxCqtOctaveArgs *a, void *stream) {
    const int st = afxk_cqt_octave_f16(a, stream);
    if (st != AFX_ERR_UNSUPPORTED) return st;
    const int N = 1 << a->radix2Exp, rows = a->rows;
    afx_functional_launches[1]++;
    double *G = (double *)calloc((size_t)N * 2 * rows, sizeof(double));
    if (!G) abort();
    if (a->timeKernel) { /* [N][32 colTiles], columns [Re 0..rows-1 | Im 0..rows-1] */
        const int cols = 32 * a->colTiles;
        for (int n = 0; n < N; n++)
            for (int j = 0; j < 2 * rows; j++) G[(size_t)n * 2 * rows + j] = a->timeKernel[(size_t)n * cols + j];
    } else {             /* spectral kernels: Q_j = sum_k K_j[k] X[k], X[k] = sum_n x[n] e^{-2 pi i k n / N} */
        for (int j = 0; j < rows; j++) {
            const int row = a->rowBase + j;
            for (int n = 0; n < N; n++) {
                double re = 0, im = 0;
                for (int q = 0; q < a->kLen[row]; q++) {
                    const double kr = a->kTaps[2 * (a->kOff[row] + q)], ki = a->kTaps[2 * (a->kOff[row] + q) + 1];
                    const double ph = -2.0 * M_PI * (double)(((long long)(a->kStart[row] + q) * n) % N) / N;
                    re += kr * cos(ph) - ki * sin(ph);
                    im += kr * sin(ph) + ki * cos(ph);
                }
                G[(size_t)n * 2 * rows + j] = re;
                G[(size_t)n * 2 * rows + rows + j] = im;
            }
        }
    }
    octave_product(a->x, a->xStride, a->validLength, a->hop, a->rightPad, N, G, rows, a->scale, a->octScale, a->num, a->colBase,
                   a->outRe, a->outIm, a->outStride, a->batch > 0 ? a->batch : 1, a->timeLength);
    free(G);
    return AFX_OK;
}

/* out[r][c] = sum over the bins j of class c (ascending) of |Q_rj|^2 (or |Q_rj|), then the row's normalisation:
 * 0 none, 1 max, 2 min, 3 P2, 4 P1 (a zero norm leaves the row as it is) */
static void chroma_rows(const float *re, const float *im, long long rows, int num, const unsigned char *cls, int chromaNum,
                        int isMag, int normType, float *out) {
    for (long long r = 0; r < rows; r++) {
        double v[64];
        for (int c = 0; c < chromaNum; c++) v[c] = 0;
        for (int j = 0; j < num; j++) {
            if (cls[j] == 255) continue;
            const double a = re[r * num + j], b = im[r * num + j];
            v[cls[j]] += isMag ? sqrt(a * a + b * b) : a * a + b * b;
        }
        if (normType != 0) {
            double red = normType == 2 ? 3.4e38 : 0;
            for (int c = 0; c < chromaNum; c++) {
                const double av = fabs(v[c]);
                if (normType == 1) red = av > red ? av : red;
                else if (normType == 2) red = av < red ? av : red;
                else if (normType == 3) red += av * av;
                else red += av;
            }
            if (normType == 3) red = sqrt(red);
            if (red != 0)
                for (int c = 0; c < chromaNum; c++) v[c] /= red;
        }
        for (int c = 0; c < chromaNum; c++) out[r * chromaNum + c] = (float)v[c];
    }
}

int afxk_cqt_chroma(const float *re, const float *im, long long rows, int num, const unsigned char *fold,
                    const AfxChromaLists *lists, int chromaNum, int isMag, int normType, float *out, void *stream) {
    (void)stream;
    if (chromaNum > 64 || num > 255) return AFX_ERR_UNSUPPORTED;
    /* the 0/1 matrix has to be a partition for the class table below (every plan of the tests is); a bin in two
     * classes is reported, not computed */
    unsigned char cls[256];
    memset(cls, 255, sizeof cls);
    for (int c = 0; c < chromaNum; c++)
        for (int j = 0; j < num; j++)
            if (fold[c * num + j]) {
                if (cls[j] != 255) return AFX_ERR_UNSUPPORTED;
                cls[j] = (unsigned char)c;
            }
    if (lists) /* the same matrix as bin lists: must agree */
        for (int c = 0; c < chromaNum; c++)
            for (int q = lists->start[c]; q < lists->start[c + 1]; q++)
                if (cls[lists->bins[q]] != c) return AFX_ERR_ARG;
    afx_functional_launches[3]++;
    chroma_rows(re, im, rows, num, cls, chromaNum, isMag, normType, out);
    return AFX_OK;
}

