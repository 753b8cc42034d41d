/* afx_auditory.c -- auditory filter-bank construction, host side, float32.
 *
 * Builds the num x (fftLength/2+1) matrix the filter-bank GEMM multiplies the
 * spectrogram with, plus the band-centre arrays the getters return.  It
 * restates the reference's auditory_filterBank (src/filterbank/
 * auditory_filterBank.c:56-207) and its helpers; because band edges are
 * decided by float32 comparisons and roundf, every expression keeps the
 * reference's float/double promotion and operation order.  The work is
 * O(num * fftLength), done once per object.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "afx_device.h"
#include "afx_host.h"

/* ---- frequency <-> scale maps (auditory_filterBank.c:1024-1190) --------- */
typedef float (*ScaleFn)(float v, float ref);

static float id_map(float v, float ref) { (void)ref; return v; }
static float fre_to_mel(float f, float ref) { (void)ref; return 2595 * log10f(1 + f / 700); }
static float mel_to_fre(float m, float ref) { (void)ref; return 700 * (powf(10, m / 2595) - 1); }

static float fre_to_bark(float f, float ref) {
    (void)ref;
    float bark = (float)(26.81 * f / (1960 + f) - 0.53);
    if (bark < 2) {
        bark = (float)(bark + 0.15 * (2 - bark));
    } else if (bark > 20.1) {
        bark = (float)(bark + 0.22 * (bark - 20.1));
    }
    return bark;
}

static float bark_to_fre(float bark, float ref) {
    (void)ref;
    if (bark < 2) {
        bark = (float)((bark - 0.3) / 0.85);
    } else if (bark > 20.1) {
        bark = (float)((bark + 4.422) / 1.22);
    }
    return (float)(1960 * (bark + 0.53) / (26.28 - bark));
}

static float fre_to_erb(float f, float ref) {
    (void)ref;
    float a = 21.3654f;
    return a * log10f((float)(1 + f * 0.004368));
}

static float erb_to_fre(float erb, float ref) {
    (void)ref;
    float a = 21.3654f;
    return (float)((powf(10, erb / a) - 1) / 0.004368);
}

float afx_fre_to_log(float fre, float binPerOctave) {
    return roundf((float)(binPerOctave * log2(fre / 440)));
}

float afx_log_to_fre(float value, float binPerOctave) {
    return (float)(pow(2, value / binPerOctave) * 440);
}

static float fre_to_logspace(float f, float ref) { (void)ref; return (float)log2(f / 440); }
static float logspace_to_fre(float v, float ref) { (void)ref; return (float)(pow(2, v) * 440); }

/* ---- low/high revision so that `num` centres fit (auditory_filterBank.c:927-1021) */
void afx_auditory_revise_log(int num, float lowFre, float highFre, int binPerOctave, int isEdge,
                             float *lowOut, float *highOut) {
    int det = isEdge ? 0 : 2, offset = isEdge ? 0 : 1;
    (void)highFre;
    float low = afx_fre_to_log(lowFre, (float)binPerOctave) - offset;
    float high = low + num - 1 + det;
    *lowOut = afx_log_to_fre(low, (float)binPerOctave);
    *highOut = afx_log_to_fre(high, (float)binPerOctave);
}

void afx_auditory_revise_linear(int num, float lowFre, float highFre, float detFre, int isEdge,
                                float *lowOut, float *highOut) {
    int det = isEdge ? 0 : 2, offset = isEdge ? 0 : 1;
    (void)highFre;
    float low = roundf(lowFre / detFre) - offset;
    float high = low + num - 1 + det;
    *lowOut = low * detFre;
    *highOut = high * detFre;
}

static void revise_linspace(int num, float lowFre, float highFre, int isEdge, float *lowOut,
                            float *highOut) {
    if (!isEdge) {
        float d = (highFre - lowFre) / (num - 1);
        *lowOut = lowFre - d;
        *highOut = highFre + d;
    } else {
        *lowOut = lowFre;
        *highOut = highFre;
    }
}

static void revise_logspace(int num, float lowFre, float highFre, int isEdge, float *lowOut,
                            float *highOut) {
    if (!isEdge) {
        float low = fre_to_logspace(lowFre, 0), high = fre_to_logspace(highFre, 0);
        float d = (high - low) / (num - 1);
        low = low - d;
        high = high + d;
        *lowOut = logspace_to_fre(low, 0);
        *highOut = logspace_to_fre(high, 0);
    } else {
        *lowOut = lowFre;
        *highOut = highFre;
    }
}

/* ---- row normalisation shared by slaney / etsi / window styles ---------- */
static void normalise_rows(float *bank, int num, int F, SpectralFilterBankNormalType normal,
                           const float *edgeFre /* num+2 */) {
    if (normal != SpectralFilterBankNormal_Area && normal != SpectralFilterBankNormal_BandWidth)
        return;
    for (int i = 0; i < num; i++) {
        float w;
        if (normal == SpectralFilterBankNormal_Area) {
            double acc = 0; /* __msum accumulates in double (flux_vector.c:399-431) */
            for (int j = 0; j < F; j++) acc += bank[(size_t)i * F + j];
            w = (float)acc;
        } else {
            w = edgeFre[i + 2] - edgeFre[i];
            w = w / 2;
        }
        /* __mdiv_vector keeps exact zeros (flux_vector.c:267-308) */
        for (int j = 0; j < F; j++) {
            float v = bank[(size_t)i * F + j];
            bank[(size_t)i * F + j] = v ? v / w : 0.f;
        }
    }
}

/* ---- styles -------------------------------------------------------------- */

/* triangles in Hz on the FFT grid (auditory_filterBank.c:435-500) */
static int style_slaney(int num, int fftLength, int samplate, SpectralFilterBankNormalType normal,
                         const float *fre, const int *bin, float *bank) {
    const int F = fftLength / 2 + 1;
    float *grid = afx_linspace(0, samplate - samplate / (float)fftLength, fftLength, 0);
    float *width = (float *)calloc((size_t)num + 1, sizeof(float));
    if (!grid || !width) {
        free(grid);
        free(width);
        return AFX_ERR_NOMEM;
    }
    for (int i = 0; i < num + 1; i++) width[i] = fre[i + 1] - fre[i];
    for (int i = 0; i < num; i++) {
        for (int j = bin[i]; j <= bin[i + 1] - 1 && j < F; j++) {
            bank[(size_t)i * F + j] = (grid[j] - fre[i]) / width[i];
        }
        for (int j = bin[i + 1]; j <= bin[i + 2] - 1 && j < F; j++) {
            bank[(size_t)i * F + j] = (fre[i + 2] - grid[j]) / width[i + 1];
        }
    }
    normalise_rows(bank, num, F, normal, fre);
    free(grid);
    free(width);
    return AFX_OK;
}

/* triangles in bins (auditory_filterBank.c:373-426) */
static void style_etsi(int num, int fftLength, SpectralFilterBankNormalType normal,
                       const float *fre, const int *bin, float *bank) {
    const int F = fftLength / 2 + 1;
    for (int i = 1; i < num + 1; i++) {
        int left = bin[i - 1], cur = bin[i], right = bin[i + 1];
        if (cur > left) {
            for (int j = left; j <= cur && j < F; j++) {
                if (j >= 0) bank[(size_t)(i - 1) * F + j] = (float)(1.0 * (j - left) / (cur - left));
            }
        }
        for (int j = cur + 1; j <= right && j < F; j++) {
            if (j >= 0) bank[(size_t)(i - 1) * F + j] = (float)(1.0 * (right - j) / (right - cur));
        }
    }
    normalise_rows(bank, num, F, normal, fre);
}

/* point / rect / window-shaped bands (auditory_filterBank.c:210-337) */
static int style_window(int num, int fftLength, SpectralFilterBankStyleType style,
                         SpectralFilterBankNormalType normal, const float *fre, const int *bin,
                         float *bank) {
    const int F = fftLength / 2 + 1;
    WindowType wt;
    switch (style) {
        case SpectralFilterBankStyle_Hann: wt = Window_Hann; break;
        case SpectralFilterBankStyle_Hamm: wt = Window_Hamm; break;
        case SpectralFilterBankStyle_Blackman: wt = Window_Blackman; break;
        case SpectralFilterBankStyle_Bohman: wt = Window_Bohman; break;
        case SpectralFilterBankStyle_Kaiser: wt = Window_Kaiser; break;
        default: wt = Window_Gauss; break;
    }
    for (int i = 1; i < num + 1; i++) {
        int left = bin[i - 1], cur = bin[i], right = bin[i + 1];
        float *row = bank + (size_t)(i - 1) * F;
        if (style == SpectralFilterBankStyle_Point) {
            if (cur >= 0 && cur < F) row[cur] = 1.0f;
        } else if (style == SpectralFilterBankStyle_Rect) {
            for (int j = left; j <= right && j < F; j++)
                if (j >= 0) row[j] = 1.0f;
        } else {
            if (cur > left) { /* rising half of a symmetric window of 2(cur-left)+1 */
                float *w = afx_window_create(wt, 2 * (cur - left) + 1, 0);
                if (!w) return AFX_ERR_NOMEM;
                for (int j = left, k = 0; j <= cur && j < F; j++, k++)
                    if (j >= 0) row[j] = w[k];
                free(w);
            }
            if (right > cur) { /* falling half, starting one past the peak */
                int n = 2 * (right - cur) + 1;
                float *w = afx_window_create(wt, n, 0);
                if (!w) return AFX_ERR_NOMEM;
                for (int j = cur + 1, k = n / 2 + 1; j <= right && j < F; j++, k++)
                    if (j >= 0) row[j] = w[k];
                free(w);
            }
        }
    }
    normalise_rows(bank, num, F, normal, fre);
    return AFX_OK;
}

/* 4th-order gammatone magnitude responses (auditory_filterBank.c:509-591,
 * coefficients :696-925, cascade response dsp/filterDesign_freqz.c:12-138) */
static int style_gammatone(int num, int fftLength, int samplate,
                            SpectralFilterBankNormalType normal, const float *fre, float *bank) {
    const int F = fftLength / 2 + 1;
    const float t = (float)(1.0 / samplate);
    const float pv = sqrtf(3 + powf(2, 1.5)), nv = sqrtf(3 - powf(2, 1.5));
    const float end = (float)(2 * M_PI);
    float *omega = afx_linspace(0, end - end / fftLength, fftLength, 0);
    if (!omega) return AFX_ERR_NOMEM;

    for (int b = 0; b < num; b++) {
        const float f = fre[b];
        const float erb = (float)((f / 9.26449 + 24.7) * 2 * M_PI * 1.019);
        const float arg = (float)(f * 2 * M_PI * t);
        const float v = -t * expf(-t * erb);
        const float cs = cosf(arg), sn = sinf(arg);
        const float cRe = cosf((float)(4 * M_PI * t * f)), cIm = sinf((float)(4 * M_PI * t * f));
        const float gRe = (float)(2 * t * expf(-erb * t) * cos(2 * M_PI * t * f));
        const float gIm = (float)(2 * t * expf(-erb * t) * sin(2 * M_PI * t * f));
        const float b1 = -2 * cs / expf(erb * t);
        const float b2 = expf(-2 * t * erb);
        const float k1 = cs + pv * sn, k2 = cs - pv * sn, k3 = cs + nv * sn, k4 = cs - nv * sn;
        const float a1[4] = {v * k1, v * k2, v * k3, v * k4};
        const float kk[4] = {k1, k2, k3, k4};
        float mag[4];
        for (int q = 0; q < 4; q++) {
            float re = -2 * t * cRe + gRe * kk[q];
            float im = -2 * t * cIm + gIm * kk[q];
            mag[q] = sqrtf(re * re + im * im);
        }
        const float r5 = -2 / expf(2 * t * erb) - 2 * cRe + 2 * (1 + cRe) / expf(t * erb);
        const float i5 = -2 * cIm + 2 * cIm / expf(t * erb);
        const float gain = mag[0] * mag[1] * mag[2] * mag[3] /
                           ((r5 * r5 + i5 * i5) * (r5 * r5 + i5 * i5));

        /* four biquads: numerator (a0, a1q, 0), denominator (1, b1, b2); the
         * first section's numerator is divided by the gain */
        float *row = bank + (size_t)b * F;
        for (int j = 0; j < F; j++) {
            float accRe = 0, accIm = 0;
            for (int q = 0; q < 4; q++) {
                float num3[3] = {t, a1[q], 0.f};
                const float den3[3] = {1.f, b1, b2};
                if (q == 0) {
                    num3[0] = t / gain;
                    num3[1] = a1[0] / gain;
                    num3[2] = 0.f / gain;
                }
                float nRe = 0, nIm = 0, dRe = 0, dIm = 0;
                for (int m = 0; m < 3; m++) {
                    nRe += cosf(-omega[j] * m) * num3[m];
                    nIm += sinf(-omega[j] * m) * num3[m];
                }
                for (int m = 0; m < 3; m++) {
                    dRe += cosf(-omega[j] * m) * den3[m];
                    dIm += sinf(-omega[j] * m) * den3[m];
                }
                float dd = dRe * dRe + dIm * dIm;
                float hRe = (nRe * dRe + nIm * dIm) / dd;
                float hIm = (nIm * dRe - nRe * dIm) / dd;
                if (q == 0) {
                    accRe = hRe;
                    accIm = hIm;
                } else {
                    float r = accRe * hRe - accIm * hIm;
                    float i = accIm * hRe + accRe * hIm;
                    accRe = r;
                    accIm = i;
                }
            }
            row[j] = sqrtf(accRe * accRe + accIm * accIm);
        }
    }

    if (normal == SpectralFilterBankNormal_Area || normal == SpectralFilterBankNormal_BandWidth) {
        for (int b = 0; b < num; b++) {
            float *row = bank + (size_t)b * F;
            float w;
            if (normal == SpectralFilterBankNormal_Area) {
                double acc = 0;
                w = row[0] + row[F - 1];
                for (int j = 1; j < F - 1; j++) acc += row[j];
                w += (float)acc * 2;
            } else {
                w = (float)(1.019 * 24.7 * (0.00437 * fre[b] + 1));
                w = w / 2;
            }
            for (int j = 0; j < F; j++) row[j] = row[j] ? row[j] / w : 0.f;
        }
    }
    /* one-sided spectrum: interior bins count twice */
    for (int b = 0; b < num; b++) {
        float *row = bank + (size_t)b * F;
        for (int j = 1; j < F - 1; j++) row[j] *= 2;
    }
    free(omega);
    return AFX_OK;
}

int afx_auditory_bank(int num, int fftLength, int samplate, SpectralFilterBankScaleType scale,
                       SpectralFilterBankStyleType style, SpectralFilterBankNormalType normal,
                       float lowFre, float highFre, int binPerOctave, float *bank, float *freOut,
                       int *binOut) {
    const int isEdge = (style == SpectralFilterBankStyle_Gammatone);
    const int offset = isEdge ? 0 : 1;
    const int count = num + (isEdge ? 0 : 2);
    float ref = 0;
    ScaleFn fwd = id_map, inv = id_map;

    /* 0. widen [low, high] by one band on each side (auditory_filterBank.c:82-119) */
    switch (scale) {
        case SpectralFilterBankScale_Octave:
            ref = (binPerOctave >= 4 && binPerOctave <= 48) ? (float)binPerOctave : 12.f;
            afx_auditory_revise_log(num, lowFre, highFre, (int)ref, isEdge, &lowFre, &highFre);
            fwd = afx_fre_to_log;
            inv = afx_log_to_fre;
            break;
        case SpectralFilterBankScale_LogChroma: /* the spectrogram object's log-chroma base bank (:106-117, :147-150) */
            ref = (binPerOctave >= 12 && binPerOctave % 12 == 0) ? (float)binPerOctave : 12.f;
            afx_auditory_revise_log(num, lowFre, highFre, (int)ref, isEdge, &lowFre, &highFre);
            fwd = afx_fre_to_log;
            inv = afx_log_to_fre;
            break;
        case SpectralFilterBankScale_Linspace:
            revise_linspace(num, lowFre, highFre, isEdge, &lowFre, &highFre);
            break;
        case SpectralFilterBankScale_Log:
            revise_logspace(num, lowFre, highFre, isEdge, &lowFre, &highFre);
            fwd = fre_to_logspace;
            inv = logspace_to_fre;
            break;
        case SpectralFilterBankScale_Mel:
            fwd = fre_to_mel;
            inv = mel_to_fre;
            break;
        case SpectralFilterBankScale_Bark:
            fwd = fre_to_bark;
            inv = bark_to_fre;
            break;
        case SpectralFilterBankScale_Erb:
            fwd = fre_to_erb;
            inv = erb_to_fre;
            break;
        default:
            break;
    }

    /* 1. band edges: equally spaced on the scale axis, mapped back to Hz, then
     *    to FFT bins (auditory_filterBank.c:594-677) */
    float *fre = afx_linspace(fwd(lowFre, ref), fwd(highFre, ref), count, 0);
    int *bin = (int *)calloc((size_t)num + 2, sizeof(int));
    if (!fre || !bin) {
        free(fre);
        free(bin);
        return AFX_ERR_NOMEM;
    }
    for (int i = 0; i < count; i++) fre[i] = inv(fre[i], ref);
    if (style != SpectralFilterBankStyle_Slaney) {
        for (int i = 0; i < count; i++) bin[i] = (int)roundf(fftLength * fre[i] / samplate);
    } else { /* first grid point strictly above the edge frequency */
        float *grid = afx_linspace(0, samplate - samplate / (float)fftLength, fftLength, 0);
        if (!grid) {
            free(fre);
            free(bin);
            return AFX_ERR_NOMEM;
        }
        for (int i = 0; i < num + 2; i++) {
            for (int j = 0; j < fftLength; j++) {
                if (grid[j] > fre[i]) {
                    bin[i] = j;
                    break;
                }
            }
        }
        free(grid);
    }

    /* 2. the bank itself */
    int st = AFX_OK;
    switch (style) {
        case SpectralFilterBankStyle_Slaney:
            st = style_slaney(num, fftLength, samplate, normal, fre, bin, bank);
            break;
        case SpectralFilterBankStyle_ETSI:
            style_etsi(num, fftLength, normal, fre, bin, bank);
            break;
        case SpectralFilterBankStyle_Gammatone:
            st = style_gammatone(num, fftLength, samplate, normal, fre, bank);
            break;
        default:
            st = style_window(num, fftLength, style, normal, fre, bin, bank);
            break;
    }

    if (freOut) memcpy(freOut, fre + offset, sizeof(float) * (size_t)num);
    if (binOut) memcpy(binOut, bin + offset, sizeof(int) * (size_t)num);
    free(fre);
    free(bin);
    return st;
}
