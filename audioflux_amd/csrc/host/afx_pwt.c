/* afx_pwt.c -- the pseudo wavelet transform object (C host side) behind include/pwt_algorithm.h.
 *
 * Parameter semantics follow pwtObj_new (src/pwt_algorithm.c:65-293); the bank is the auditory
 * filter bank laid out over all L transform bins ("pseudo" layout: row pitch L, bins above
 * Nyquist zero; auditory_filterBank.c:56-207 with isPseudo = 1).  Execution is the CWT object's
 * (afx_cwt.c: four-step / in-LDS FFT kernels, one product + inverse per band).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "afx_batch.h"
#include "afx_device.h"
#include "afx_host.h"
#include "cwt_algorithm.h"
#include "pwt_algorithm.h"

struct OpaquePWT {
    CWTObj core;
    int num;
};

/* the "pseudo" bank [num][L] + band arrays; 0 or a negative status.  Exported for the host tests. */
int afx_pwt_bank_host(int num, long long L, int samplate, SpectralFilterBankScaleType scale,
                      SpectralFilterBankStyleType style, SpectralFilterBankNormalType normal, float low,
                      float high, int bpo, float *bank, float *fre, int *bin) {
    if (L > 0x7fffffffLL) return AFX_ERR_UNSUPPORTED;
    const int N = (int)L, F = N / 2 + 1;
    if (scale == SpectralFilterBankScale_Linear) {
        /* auditory_filterBank.c:92-96, :121-124, :339-365: band edges on the bin grid, one unit
         * weight per band at (edge bin - 1), and the returned bin array carries that decrement */
        const float det = (float)(samplate * 1.0 / N);
        afx_auditory_revise_linear(num, low, high, det, 0, &low, &high);
        float *edge = afx_linspace(roundf(low / det), roundf(high / det), num + 2, 0);
        int *b = (int *)calloc((size_t)num + 2, sizeof(int));
        float *grid = NULL;
        if (!edge || !b) {
            free(edge); free(b);
            return AFX_ERR_NOMEM;
        }
        for (int i = 0; i < num + 2; i++) edge[i] = edge[i] * det;
        if (style != SpectralFilterBankStyle_Slaney) {
            for (int i = 0; i < num + 2; i++) b[i] = (int)roundf(N * edge[i] / samplate);
        } else {
            grid = afx_linspace(0, samplate - samplate / (float)N, N, 0);
            for (int i = 0; grid && i < num + 2; i++)
                for (int j = 0; j < N; j++)
                    if (grid[j] > edge[i]) {
                        b[i] = j;
                        break;
                    }
            free(grid);
        }
        for (int i = 1; i < num + 1; i++) {
            b[i] -= 1;
            if (b[i] >= 0 && b[i] < N) bank[(size_t)(i - 1) * N + b[i]] = 1.f;
        }
        memcpy(fre, edge + 1, sizeof(float) * (size_t)num);
        memcpy(bin, b + 1, sizeof(int) * (size_t)num);
        free(edge);
        free(b);
        return AFX_OK;
    }
    if (style == SpectralFilterBankStyle_Gammatone) {
        /* auditory_filterBank.c:509-582 in its pseudo layout: the magnitude responses are written back to back at
         * the HALF pitch F = N/2 + 1 (:545-551, `i*len`), the normalisation and the interior doubling then walk the
         * same memory at the FULL pitch N (:554-581, `i*mLength`).  Row i of the bank the transform applies is
         * therefore floats [i N, (i + 1) N) of the concatenated half rows -- the tail of response 2i - 1, response 2i,
         * most of response 2i + 1 (its upper part lands on the negative-frequency bins) -- and rows from about num / 2 on
         * are zero (0 / 0 = NaN under Area normalisation, as in the reference).  Reproduced as it is. */
        float *resp = (float *)calloc((size_t)num * F, sizeof(float));
        if (!resp) return AFX_ERR_NOMEM;
        int st = afx_auditory_bank(num, N, samplate, scale, style, SpectralFilterBankNormal_None, low, high, bpo, resp, fre, bin);
        if (st == AFX_OK) {
            /* (the half-layout builder has doubled the interior bins: exact to undo) */
            for (int i = 0; i < num; i++)
                for (int j = 1; j < F - 1; j++) resp[(size_t)i * F + j] *= 0.5f;
            memcpy(bank, resp, sizeof(float) * (size_t)num * F);
            if (normal == SpectralFilterBankNormal_Area || normal == SpectralFilterBankNormal_BandWidth) {
                for (int i = 0; i < num; i++) {
                    float *row = bank + (size_t)i * N;
                    float w;
                    if (normal == SpectralFilterBankNormal_Area) {
                        double sum = 0.0; /* __vsum accumulates in double (flux_vector.c:1493-1501) */
                        for (int j = 1; j < N - 1; j++) sum += row[j];
                        w = row[0] + row[N - 1];
                        w += (float)sum * 2;
                    } else {
                        w = (float)(1.019 * 24.7 * (0.00437 * fre[i] + 1));
                        w /= 2;
                    }
                    for (int j = 0; j < N; j++) row[j] = row[j] / w;
                }
            }
            for (int i = 0; i < num; i++)
                for (int j = 1; j < N - 1; j++) bank[(size_t)i * N + j] *= 2;
        }
        free(resp);
        return st;
    }
    float *half = (float *)calloc((size_t)num * F, sizeof(float));
    if (!half) return AFX_ERR_NOMEM;
    const int st = afx_auditory_bank(num, N, samplate, scale, style, normal, low, high, bpo, half, fre, bin);
    for (int i = 0; st == AFX_OK && i < num; i++)
        memcpy(bank + (size_t)i * N, half + (size_t)i * F, sizeof(float) * (size_t)F);
    free(half);
    return st;
}

int pwtObj_new(PWTObj *pwtObj, int num, int radix2Exp, int *samplate, float *lowFre, float *highFre,
               int *binPerOctave, SpectralFilterBankScaleType *scaleType,
               SpectralFilterBankStyleType *styleType, SpectralFilterBankNormalType *normalType,
               int *isPadding) {
    int sr = 32000, bpo = 12, isPad = 0;
    float low = 0, high = 0;
    SpectralFilterBankScaleType sc = SpectralFilterBankScale_Octave;
    SpectralFilterBankStyleType style = SpectralFilterBankStyle_Slaney;
    SpectralFilterBankNormalType normal = SpectralFilterBankNormal_None;
    if (!pwtObj) return -1;
    *pwtObj = NULL;
    if (radix2Exp) {
        if (radix2Exp < 1 || radix2Exp > 30) {
            printf("radix2Exp is error!\n");
            return -100;
        }
    }
    const long long D = 1LL << radix2Exp;
    if (samplate && *samplate > 0 && *samplate <= 196000) sr = *samplate;
    if (scaleType) {
        sc = *scaleType;
        if ((int)sc > (int)SpectralFilterBankScale_Log) {
            printf("scaleType is error!\n");
            return 1;
        }
    }
    if (styleType) style = *styleType;
    if (normalType) normal = *normalType;
    high = (float)(sr / 2.0);
    if (lowFre && *lowFre >= 0 && *lowFre < sr / 2.0) low = *lowFre;
    const int logLike = (sc == SpectralFilterBankScale_Octave || sc == SpectralFilterBankScale_Log);
    if (low == 0 && logLike) {
        low = (float)(powf(2, (float)(-45 / 12.0)) * 440);
        high = (float)(powf(2, (float)(38 / 12.0)) * 440);
    }
    if (highFre && *highFre > 0 && *highFre <= sr / 2.0) high = *highFre;
    if (high < low) {
        low = 0;
        high = (float)(sr / 2.0);
        if (logLike) {
            low = (float)(powf(2, (float)(-45 / 12.0)) * 440);
            high = (float)(powf(2, (float)(38 / 12.0)) * 440);
        }
    }
    if (binPerOctave && *binPerOctave >= 4 && *binPerOctave <= 48) bpo = *binPerOctave;
    if (sc == SpectralFilterBankScale_Linear) {
        const float det = sr / (float)D;
        afx_auditory_revise_linear(num, low, high, det, 1, &low, &high);
        if (high > sr / 2.0) {
            printf("scale linear: lowFre and num is large, overflow error\n");
            return -1;
        }
    } else if (sc == SpectralFilterBankScale_Octave) {
        afx_auditory_revise_log(num, low, high, bpo, 1, &low, &high);
        if (high > sr / 2.0) {
            printf("scale log: lowFre and num is large, overflow error!\n");
            return -1;
        }
    }
    if (isPadding) isPad = *isPadding;
    if (num < 2 || num > D / 2 + 1) {
        printf("num is error!\n");
        return -1;
    }
    const long long L = afx_cwt_fft_length(radix2Exp, isPad);
    if (L < 0) return AFX_ERR_UNSUPPORTED;

    PWTObj o = (PWTObj)calloc(1, sizeof(struct OpaquePWT));
    float *bank = (float *)calloc((size_t)num * (size_t)L, sizeof(float));
    float *fre = (float *)calloc((size_t)num + 2, sizeof(float));
    int *bin = (int *)calloc((size_t)num + 2, sizeof(int));
    int st = (o && bank && fre && bin) ? AFX_OK : AFX_ERR_NOMEM;
    if (st == AFX_OK) st = afx_pwt_bank_host(num, L, sr, sc, style, normal, low, high, bpo, bank, fre, bin);
    if (st == AFX_OK) st = afx_cwt_create_custom(&o->core, num, radix2Exp, sr, isPad, bank, fre, bin, "pwtObj_new");
    free(bank);
    free(fre);
    free(bin);
    if (st != AFX_OK) {
        free(o);
        return st;
    }
    o->num = num;
    *pwtObj = o;
    return 0;
}

float *pwtObj_getFreBandArr(PWTObj o) { return o ? cwtObj_getFreBandArr(o->core) : NULL; }
int *pwtObj_getBinBandArr(PWTObj o) { return o ? cwtObj_getBinBandArr(o->core) : NULL; }

void pwtObj_pwt(PWTObj o, float *dataArr, float *mRealArr3, float *mImageArr3) {
    if (!o) {
        afxdev_set_error("pwtObj_pwt: NULL object");
        return;
    }
    cwtObj_cwt(o->core, dataArr, mRealArr3, mImageArr3);
}

void pwtObj_enableDet(PWTObj o, int flag) {
    if (o) cwtObj_enableDet(o->core, flag);
}

void pwtObj_pwtDet(PWTObj o, float *dataArr, float *mRealArr3, float *mImageArr3) {
    if (o) cwtObj_cwtDet(o->core, dataArr, mRealArr3, mImageArr3);
}

int pwtObj_pwtBatchDevice(PWTObj o, const float *dData, int chunks, long long chunkStride, float *dReal,
                          float *dImag, void *hipStream) {
    if (!o) return AFX_ERR_ARG;
    return cwtObj_cwtBatchDevice(o->core, dData, chunks, chunkStride, dReal, dImag, hipStream);
}

void pwtObj_free(PWTObj o) {
    if (!o) return;
    cwtObj_free(o->core);
    free(o);
}
