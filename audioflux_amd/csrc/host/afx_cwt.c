/* afx_cwt.c -- the continuous wavelet transform object (C host side) behind
 * include/cwt_algorithm.h.
 *
 * Parameter handling follows cwtObj_new (src/cwt_algorithm.c:73-334); the
 * frequency-domain wavelet bank follows cwt_filterBank
 * (src/filterbank/cwt_filterBank.c:85-290 and the per-family functions :368-600)
 * in the reference's own float / long double arithmetic, then is stored in the
 * transposed frequency layout the four-step FFT kernels use (afx_cwt.hip).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "afx_device.h"
#include "afx_host.h"
#include "cwt_algorithm.h"

#ifndef M_E
#define M_E 2.7182818284590452354
#endif

struct OpaqueCWT {
    int num, radix2Exp, dataLength, padLength;
    long long fftLength;
    int samplate, binPerOctave;
    float lowFre, highFre, gamma, beta;
    WaveletContinueType waveletType;
    SpectralFilterBankScaleType scaleType;
    float *freBandArr; /* host, num+2 */
    int *binBandArr;
    float *hBank;      /* host copy, natural layout [num][L] (kept for the det bank) */
    AfxCwtPlanDims dims;
    void *stream;
    void *stream2;           /* side stream of the batched device call: narrow-band scales */
    void *chain[3];          /* side streams: [0], [1] extra two-pass chains, [2] the time-domain scales */
    float *dTw, *dBankT, *dBankDetT;
    float *dX, *dA, *dXt, *dB, *dOut; /* scratch of the one-chunk calls */
    float *dFastTw;          /* twiddle tables of the register-FFT kernels (L = 2^17) */
    float *dBankN, *dBankDetN; /* natural-layout banks of the in-LDS path (L <= 16384), else NULL */
    int *dSupport;           /* [num][2]: k2 range holding every non-zero of the scale's wavelet */
    int *dOrder;             /* [num]: wide scales first, then the narrow-band classes (afx_device.h) */
    AfxCwtTdPlan td;         /* time-domain plan of the short-kernel scales (afx_cwt_td.hip); nPairs 0: none */
    AfxCwtTdPair *dTdPairs;
    int *tdKs;               /* host: K steps of the pairs */
    unsigned char *dTdImage;
    AfxCwtTdPlan tdDet;      /* the same scales with the derivative bank (cwtObj_enableDet); nPairs 0: two-pass */
    AfxCwtTdPair *dTdPairsDet;
    int *tdKsDet;
    unsigned char *dTdImageDet;
    int *hOrder;             /* host copy of the first nTd entries of the order list */
    float *dGA, *dGXt, *dGB; /* scratch of the batched calls: `group` chunks at a time */
    size_t capGA, capGXt, capGB;
    int haveSpectrum;
    int status;
    void *lastStream;        /* stream of the previous batched device call (scratch ordering) */
    int lastUsed;
};

/* ---- scale maps shared with the auditory bank (same formulas, see afx_auditory.c) ---- */
static float c_fre_to_mel(float f) { return 2595 * log10f(1 + f / 700); }
static float c_mel_to_fre(float m) { return 700 * (powf(10, m / 2595) - 1); }
static float c_fre_to_bark(float f) {
    float bark = (float)(26.81 * f / (1960 + f) - 0.53);
    if (bark < 2) bark = (float)(bark + 0.15 * (2 - bark));
    else if (bark > 20.1) bark = (float)(bark + 0.22 * (bark - 20.1));
    return bark;
}
static float c_bark_to_fre(float bark) {
    if (bark < 2) bark = (float)((bark - 0.3) / 0.85);
    else if (bark > 20.1) bark = (float)((bark + 4.422) / 1.22);
    return (float)(1960 * (bark + 0.53) / (26.28 - bark));
}
static float c_fre_to_erb(float f) { return 21.3654f * log10f((float)(1 + f * 0.004368)); }
static float c_erb_to_fre(float e) { return (float)((powf(10, e / 21.3654f) - 1) / 0.004368); }
static float c_fre_to_logspace(float f) { return (float)log2(f / 440); }
static float c_logspace_to_fre(float v) { return (float)(pow(2, v) * 440); }

/* util_gammal (src/util/flux_util.c:731-768) */
static long double gammal_ref(long double x) {
    const long double err = 1e-5;
    if (fabsl(x - 1.0) < err) return 1.0;
    if (fabsl(x - 0.5) < err) return sqrt(M_PI);
    if (x > 1.0) return (x - 1) * gammal_ref(x - 1);
    if (x < 0) return gammal_ref(x + 1) / x;
    int i = 1;
    long double cur = 1.0, pre = 0;
    while (fabsl(cur - pre) / cur > err) {
        pre = cur;
        cur *= i / (x - 1 + i);
        i++;
    }
    return cur * powl(i, x - 1);
}

/* natural-layout bank [num][L] (cwt_filterBank.c:85-290) */
static int build_bank(struct OpaqueCWT *o, float *bank) {
    const int num = o->num, D = o->dataLength, sr = o->samplate;
    const long long L = o->fftLength;
    float low = o->lowFre, high = o->highFre, ref = 0;
    const WaveletContinueType type = o->waveletType;
    const float gamma = o->gamma, beta = o->beta;

    /* band centres: num+2 points equally spaced on the scale axis, edges excluded */
    switch (o->scaleType) {
        case SpectralFilterBankScale_Octave:
            ref = (o->binPerOctave >= 4 && o->binPerOctave <= 48) ? (float)o->binPerOctave : 12.f;
            afx_auditory_revise_log(num, low, high, (int)ref, 0, &low, &high);
            break;
        case SpectralFilterBankScale_Linear:
            ref = (float)(sr * 1.0 / D);
            afx_auditory_revise_linear(num, low, high, ref, 0, &low, &high);
            break;
        case SpectralFilterBankScale_Linspace: {
            float d = (high - low) / (num - 1);
            low = low - d;
            high = high + d;
            break;
        }
        case SpectralFilterBankScale_Log: {
            float a = c_fre_to_logspace(low), b = c_fre_to_logspace(high);
            float d = (b - a) / (num - 1);
            low = c_logspace_to_fre(a - d);
            high = c_logspace_to_fre(b + d);
            break;
        }
        default:
            break;
    }
    float a, b;
    switch (o->scaleType) {
        case SpectralFilterBankScale_Linear: a = roundf(low / ref); b = roundf(high / ref); break;
        case SpectralFilterBankScale_Mel: a = c_fre_to_mel(low); b = c_fre_to_mel(high); break;
        case SpectralFilterBankScale_Bark: a = c_fre_to_bark(low); b = c_fre_to_bark(high); break;
        case SpectralFilterBankScale_Erb: a = c_fre_to_erb(low); b = c_fre_to_erb(high); break;
        case SpectralFilterBankScale_Octave: a = afx_fre_to_log(low, ref); b = afx_fre_to_log(high, ref); break;
        case SpectralFilterBankScale_Log: a = c_fre_to_logspace(low); b = c_fre_to_logspace(high); break;
        default: a = low; b = high; break;
    }
    float *f = afx_linspace(a, b, num + 2, 0);
    if (!f) return AFX_ERR_NOMEM;
    for (int i = 0; i < num + 2; i++) {
        switch (o->scaleType) {
            case SpectralFilterBankScale_Linear: f[i] = f[i] * ref; break;
            case SpectralFilterBankScale_Mel: f[i] = c_mel_to_fre(f[i]); break;
            case SpectralFilterBankScale_Bark: f[i] = c_bark_to_fre(f[i]); break;
            case SpectralFilterBankScale_Erb: f[i] = c_erb_to_fre(f[i]); break;
            case SpectralFilterBankScale_Octave: f[i] = afx_log_to_fre(f[i], ref); break;
            case SpectralFilterBankScale_Log: f[i] = c_logspace_to_fre(f[i]); break;
            default: break;
        }
    }
    for (int i = 0; i < num; i++) {
        o->freBandArr[i] = f[i + 1];
        o->binBandArr[i] = (int)roundf(D * f[i + 1] / sr);
    }

    /* centre frequency of the mother wavelet (cwt_filterBank.c:136-163) */
    float cf = gamma;
    if (type == WaveletContinue_Morse) cf = expf((float)(1.0 / gamma * (logf(beta) - logf(gamma))));
    else if (type == WaveletContinue_Paul) cf = (float)(gamma + 0.5);
    else if (type == WaveletContinue_DOG) cf = sqrtf((float)(gamma + 0.5));
    else if (type == WaveletContinue_Mexican) cf = sqrtf((float)(2 + 0.5));
    else if (type == WaveletContinue_Hermit) cf = gamma + 1;

    /* angular frequency grid and scales, highest frequency first (:218-236) */
    float *w = (float *)calloc((size_t)L, sizeof(float));
    float *s = (float *)calloc((size_t)num, sizeof(float));
    if (!w || !s) {
        free(f);
        free(w);
        free(s);
        return AFX_ERR_NOMEM;
    }
    for (long long i = 0; i <= L / 2; i++) w[i] = (float)(i * 2 * M_PI / L);
    for (long long i = L / 2 + 1, j = L / 2 - 1; i < L && j >= 0; i++, j--) w[i] = -w[j];
    for (int i = num, j = 0; i >= 1; i--, j++) {
        float v = f[i];
        if (v < 1e-6) v = 1e-6f;
        s[j] = (float)(cf / (v / sr * 2 * M_PI));
    }

    long double factor = 0;
    float factorF = 0;
    if (type == WaveletContinue_Morse) factorF = expf(-beta * logf(cf) + powf(cf, gamma));
    else if (type == WaveletContinue_Paul) {
        long double v1 = 1;
        const int p = (int)roundf(gamma);
        for (int i = 2 * p - 1; i >= 2; i--) v1 *= i;
        factor = powl(2, p) / sqrtl(p * v1);
    } else if (type == WaveletContinue_DOG || type == WaveletContinue_Mexican) {
        const int p = (int)roundf(type == WaveletContinue_Mexican ? 2.f : gamma);
        factor = -1.0 / sqrtl(gammal_ref(p + 0.5));
        if ((p / 2) % 2 == 1) factor = -factor;
    } else if (type == WaveletContinue_Hermit) factor = 2.0 / sqrtf(gamma) * powl(M_PI, -0.25);
    else if (type == WaveletContinue_Ricker) factor = 2.0 / sqrtf((float)M_PI);

    const float dogGamma = (type == WaveletContinue_Mexican) ? 2.f : gamma;
    for (int i = 0; i < num; i++) {
        float *row = bank + (size_t)i * L;
        for (long long j = 0; j < L; j++) {
            const float x = s[i] * w[j]; /* __mdot of a column by a row: one float product */
            float out = 0.f;
            switch (type) {
                case WaveletContinue_Morse:
                    if (x > 0) {
                        const float v1 = fabsf(x);
                        const float v2 = (gamma == 3) ? v1 * v1 * v1 : powf(v1, gamma);
                        out = 2 * factorF * expf(beta * logf(v1) - v2);
                    }
                    break;
                case WaveletContinue_Morlet:
                    if (x > 0) {
                        float v1 = -(x - gamma) * (x - gamma) / beta;
                        out = 2 * expf(v1);
                    }
                    break;
                case WaveletContinue_Bump: {
                    const float eps = 1e-6f;
                    const float v1 = (x - gamma) / beta;
                    float v2 = v1 * v1;
                    v2 = -1 / (1 - v2);
                    if (fabsf(v1) < 1 - eps) {
                        const float v3 = (float)(2 * M_E * expf(v2));
                        out = isnan(v3) ? 0.f : v3;
                    }
                    break;
                }
                case WaveletContinue_Paul:
                    if (x > 0) {
                        long double v = x;
                        v = powl(v, gamma) * expl(-v);
                        out = (float)(factor * v);
                    }
                    break;
                case WaveletContinue_DOG:
                case WaveletContinue_Mexican:
                    if (x > 0) {
                        long double v = x;
                        v = powl(v, dogGamma) * expl(-v * v / beta);
                        out = (float)(factor * v);
                    }
                    break;
                case WaveletContinue_Hermit:
                    if (x > 0) {
                        long double v = x;
                        v = (v - gamma) * (1 + v - gamma) * expl(-(v - gamma) * (v - gamma) / beta);
                        out = (float)(factor * v);
                    }
                    break;
                case WaveletContinue_Ricker:
                    if (x > 0) {
                        long double v = x;
                        v = v * v / (gamma * gamma * gamma) * expl(-v * v / (gamma * gamma));
                        out = (float)(factor * v);
                    }
                    break;
                default:
                    break;
            }
            row[j] = out;
        }
    }
    free(f);
    free(w);
    free(s);
    return AFX_OK;
}

/* device-free entry used by the CPU tests: the same bank cwtObj_new uploads, natural
 * layout, for already-resolved parameters (the signature of the reference's
 * cwt_filterBank, src/filterbank/cwt_filterBank.c:85) */
int afx_cwt_bank_host(int num, int dataLength, int samplate, int padLength, int waveletType,
                      float gamma, float beta, int scaleType, float lowFre, float highFre,
                      int binPerOctave, float *bank, float *freBandArr, int *binBandArr) {
    struct OpaqueCWT o;
    memset(&o, 0, sizeof(o));
    o.num = num;
    o.dataLength = dataLength;
    o.padLength = padLength;
    o.fftLength = (long long)dataLength + 2LL * padLength;
    o.samplate = samplate;
    o.binPerOctave = binPerOctave;
    o.lowFre = lowFre;
    o.highFre = highFre;
    o.gamma = gamma;
    o.beta = beta;
    o.waveletType = (WaveletContinueType)waveletType;
    o.scaleType = (SpectralFilterBankScaleType)scaleType;
    o.freBandArr = (float *)calloc((size_t)num + 2, sizeof(float));
    o.binBandArr = (int *)calloc((size_t)num + 2, sizeof(int));
    if (!o.freBandArr || !o.binBandArr) return AFX_ERR_NOMEM;
    const int bst = build_bank(&o, bank);
    if (bst != AFX_OK) {
        free(o.freBandArr);
        free(o.binBandArr);
        return bst;
    }
    if (freBandArr) memcpy(freBandArr, o.freBandArr, sizeof(float) * (size_t)num);
    if (binBandArr) memcpy(binBandArr, o.binBandArr, sizeof(int) * (size_t)num);
    free(o.freBandArr);
    free(o.binBandArr);
    return 0;
}

/* natural [num][L] -> transposed frequency layout [num][k1][k2], k = k1 + L1*k2 */
static float *to_transposed(const float *bank, int num, int r1, int r2, const float *mul) {
    const long long L1 = 1LL << r1, L2 = 1LL << r2, L = L1 * L2;
    float *t = (float *)malloc(sizeof(float) * (size_t)num * L);
    if (!t) return NULL;
    for (int i = 0; i < num; i++)
        for (long long k1 = 0; k1 < L1; k1++)
            for (long long k2 = 0; k2 < L2; k2++) {
                const long long k = k1 + L1 * k2;
                float v = bank[(size_t)i * L + k];
                if (mul) v = v * mul[k];
                t[(size_t)i * L + k1 * L2 + k2] = v;
            }
    return t;
}

static int cwt_create(CWTObj *cwtObj, const struct OpaqueCWT *proto, int rL, const float *customBank,
                      const float *customFre, const int *customBin);

int cwtObj_new(CWTObj *cwtObj, int num, int radix2Exp, int *samplate, float *lowFre, float *highFre,
               int *binPerOctave, WaveletContinueType *waveletType,
               SpectralFilterBankScaleType *scaleType, float *gamma, float *beta, int *isPadding) {
    int sr = 32000, bpo = 12, isPad = 0;
    float low = 0, high = 0, g = 3, b = 20;
    WaveletContinueType wt = WaveletContinue_Morse;
    SpectralFilterBankScaleType sc = SpectralFilterBankScale_Octave;
    if (!cwtObj) return -1;
    *cwtObj = NULL;
    if (radix2Exp) {
        if (radix2Exp < 1 || radix2Exp > 30) {
            printf("radix2Exp is error!\n");
            return -100;
        }
    }
    long long fftLength = 1LL << radix2Exp;
    if (samplate && *samplate > 0 && *samplate <= 196000) sr = *samplate;
    if (waveletType) wt = *waveletType;
    if (scaleType) {
        sc = *scaleType;
        if ((int)sc > (int)SpectralFilterBankScale_Log) {
            printf("scaleType is error!\n");
            return 1;
        }
    }
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
        float det = sr / (float)fftLength;
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
    switch (wt) { /* family defaults (cwt_algorithm.c:202-232) */
        case WaveletContinue_Morlet: g = 6; b = 2; break;
        case WaveletContinue_Bump: g = 5; b = 0.6f; break;
        case WaveletContinue_Paul: g = 4; break;
        case WaveletContinue_DOG: g = 2; b = 2; break;
        case WaveletContinue_Mexican: b = 2; break;
        case WaveletContinue_Hermit: g = 5; b = 2; break;
        case WaveletContinue_Ricker: g = 4; break;
        default: break;
    }
    if (gamma && *gamma > 0) {
        g = *gamma;
        if (wt == WaveletContinue_DOG) {
            const int p = (int)roundf(g);
            g = (p % 2 == 0) ? (float)p : 2.f;
        }
    }
    if (beta && *beta > 0) b = *beta;
    if (isPadding) isPad = *isPadding;
    if (num < 2 || num > fftLength / 2 + 1) {
        printf("num is error!\n");
        return -1;
    }
    const int D = 1 << radix2Exp;
    int pad = 0;
    if (isPad) {
        if (D <= 1e5) {
            pad = D / 2;
        } else {
            /* the reference pads by ceil(log2 N) here, which makes the length a
             * non-power-of-two and sends it down an O(N^2) DFT with N x N double tables
             * (cwt_algorithm.c:264-289) -- unusable at these sizes */
            afxdev_set_error("cwtObj_new: padding with 2^%d samples needs a non-power-of-two transform; "
                             "use isPadding=0 or radix2Exp<=16", radix2Exp);
            return AFX_ERR_UNSUPPORTED;
        }
    }
    fftLength = (long long)D + 2LL * pad;
    int rL = 0;
    while ((1LL << rL) < fftLength) rL++;
    if (rL > 26) {
        afxdev_set_error("cwtObj_new: transform length 2^%d exceeds the supported 2^26", rL);
        return AFX_ERR_UNSUPPORTED;
    }

    struct OpaqueCWT proto;
    memset(&proto, 0, sizeof(proto));
    proto.num = num;
    proto.radix2Exp = radix2Exp;
    proto.dataLength = D;
    proto.padLength = pad;
    proto.fftLength = fftLength;
    proto.samplate = sr;
    proto.binPerOctave = bpo;
    proto.lowFre = low;
    proto.highFre = high;
    proto.gamma = g;
    proto.beta = b;
    proto.waveletType = wt;
    proto.scaleType = sc;
    return cwt_create(cwtObj, &proto, rL, NULL, NULL, NULL);
}

/* transform-length rule shared by cwtObj_new and pwtObj_new (cwt_algorithm.c:264-289,
 * pwt_algorithm.c:205-219); returns the padding or a negative status */
static int pad_rule(int radix2Exp, int isPad, const char *who, long long *fftLength, int *rL) {
    const int D = 1 << radix2Exp;
    int pad = 0;
    if (isPad) {
        if (D <= 1e5) {
            pad = D / 2;
        } else {
            afxdev_set_error("%s: padding with 2^%d samples needs a non-power-of-two transform; "
                             "use isPadding=0 or radix2Exp<=16", who, radix2Exp);
            return AFX_ERR_UNSUPPORTED;
        }
    }
    *fftLength = (long long)D + 2LL * pad;
    *rL = 0;
    while ((1LL << *rL) < *fftLength) (*rL)++;
    if (*rL > 26) {
        afxdev_set_error("%s: transform length 2^%d exceeds the supported 2^26", who, *rL);
        return AFX_ERR_UNSUPPORTED;
    }
    return pad;
}

/* the pseudo-wavelet object (afx_pwt.c) is this object with a caller-built frequency-domain
 * bank [num][L] and band arrays instead of the analytic wavelet bank */
int afx_cwt_create_custom(CWTObj *cwtObj, int num, int radix2Exp, int samplate, int isPadding,
                          const float *bank, const float *fre, const int *bin, const char *who) {
    long long fftLength;
    int rL;
    const int pad = pad_rule(radix2Exp, isPadding, who, &fftLength, &rL);
    if (pad < 0) return pad;
    struct OpaqueCWT proto;
    memset(&proto, 0, sizeof(proto));
    proto.num = num;
    proto.radix2Exp = radix2Exp;
    proto.dataLength = 1 << radix2Exp;
    proto.padLength = pad;
    proto.fftLength = fftLength;
    proto.samplate = samplate;
    return cwt_create(cwtObj, &proto, rL, bank, fre, bin);
}

long long afx_cwt_fft_length(int radix2Exp, int isPadding) {
    long long fftLength;
    int rL;
    return pad_rule(radix2Exp, isPadding, "transform length", &fftLength, &rL) < 0 ? -1 : fftLength;
}

/* ---- time-domain plan (afx_cwt_td.hip) -------------------------------------------------------------------
 * A wavelet that is wide in frequency is short in time: g_j = IFFT(psi_j), evaluated here in double from the bank's
 * own float32 row.  Scale j qualifies when the taps beyond |t| = Kh_j (rule below) hold less than 5e-7 of its L2 norm and the kernel (2 Kh_j + 1 taps + the 8 phase shifts) fits the
 * LDS-resident image (AFX_CWT_TD_MAXK).  Qualifying scales are paired longest first; `order` is rewritten to
 * [time-domain scales | remaining two-pass scales | narrow-band classes].  Nothing qualifies -> nPairs = 0 and the
 * plan is the round-2 one.  (morlet at BASELINE cfg 4: 36 of the 40 two-pass scales, 121 .. 1009 taps.) */
typedef struct {
    int scale, kh;
    double *re, *im; /* taps t = -kh .. kh at [t + kh] */
    double dcRe, dcIm; /* sum of ALL L taps = the spectrum's bin 0: the whole kernel's response to a constant */
} TdCand;

static int td_cand_cmp(const void *a, const void *b) {
    const TdCand *x = (const TdCand *)a, *y = (const TdCand *)b;
    if (x->kh != y->kh) return y->kh - x->kh; /* longest first */
    return x->scale - y->scale;
}

static void td_cand_free(TdCand *cand, int nc) {
    for (int c = 0; c < nc && cand; c++) {
        free(cand[c].re);
        free(cand[c].im);
    }
    free(cand);
}

/* the scales of `list` whose kernel is short: g = IFFT(psi) for the plain bank, IFFT(j w psi) for the derivative bank
 * (weights = the angular frequencies of cwtObj_enableDet; the float32 product the two-pass path multiplies with) */
static int td_candidates(CWTObj o, int rL, const int *list, int n, const float *weights, TdCand **out, int *nOut) {
    const long long L = o->fftLength;
    const int D = o->dataLength, pad = o->padLength;
    double *re = (double *)malloc(sizeof(double) * (size_t)L), *im = (double *)malloc(sizeof(double) * (size_t)L);
    TdCand *cand = (TdCand *)calloc((size_t)n, sizeof(TdCand));
    int st = (re && im && cand) ? AFX_OK : AFX_ERR_NOMEM, nc = 0;
    for (int w = 0; w < n && st == AFX_OK; w++) {
        const int j = list[w];
        const float *row = o->hBank + (size_t)j * L;
        if (weights)
            for (long long k = 0; k < L; k++) re[k] = 0.0, im[k] = (float)(row[k] * weights[k]);
        else
            for (long long k = 0; k < L; k++) re[k] = row[k], im[k] = 0.0;
        const double dcRe = re[0], dcIm = im[0];
        st = afx_fft_f64(rL, re, im, 1); /* g = IFFT(psi): the 1 / L goes into the taps below */
        if (st != AFX_OK) break;
        /* Kh: the last |t| above 1e-6 of the peak, stretched by 8 % (a Gaussian envelope falls from 1e-6 to 1e-7
         * over that much; the float32 arithmetic of the bank leaves a floor of isolated values up to 5e-8 of the
         * peak all over the period -- a threshold at 1e-7 or below would chase it).  Accepted when the dropped
         * taps hold less than 5e-7 of the kernel's L2 norm = the relative error against the reference for a white
         * input (morlet, cfg 4: 1.5e-7, all of it the floor). */
        double peak = 0.0, energy = 0.0;
        for (long long k = 0; k < L; k++) {
            const double a2 = re[k] * re[k] + im[k] * im[k];
            if (a2 > peak) peak = a2;
            energy += a2;
        }
        if (!(peak > 0.0)) continue;
        long long k6 = 0;
        for (long long k = 0; k < L; k++)
            if (re[k] * re[k] + im[k] * im[k] > 1e-12 * peak) {
                const long long t = k <= L / 2 ? k : L - k;
                if (t > k6) k6 = t;
            }
        const long long kh = (long long)ceil(1.08 * (double)k6) + 2;
        if (2 * kh + 8 > AFX_CWT_TD_MAXK - 56 || (pad > 0 && kh > pad) || kh >= D) continue;
        double tail = 0.0;
        for (long long k = kh + 1; k < L - kh; k++) tail += re[k] * re[k] + im[k] * im[k];
        if (tail > 2.5e-13 * energy) continue; /* (5e-7)^2 */
        TdCand *c = &cand[nc];
        c->scale = j;
        c->kh = (int)kh;
        c->dcRe = dcRe;
        c->dcIm = dcIm;
        c->re = (double *)malloc(sizeof(double) * (size_t)(2 * kh + 1));
        c->im = (double *)malloc(sizeof(double) * (size_t)(2 * kh + 1));
        nc++; /* (counted before the check: td_cand_free releases a half-allocated entry too) */
        if (!c->re || !c->im) {
            st = AFX_ERR_NOMEM;
            break;
        }
        for (long long t = -kh; t <= kh; t++) {
            const long long k = t < 0 ? L + t : t;
            c->re[t + kh] = re[k] / (double)L;
            c->im[t + kh] = im[k] / (double)L;
        }
    }
    free(re);
    free(im);
    if (st != AFX_OK) {
        td_cand_free(cand, nc);
        cand = NULL;
        nc = 0;
    }
    *out = cand;
    *nOut = nc;
    return st;
}

typedef struct {
    AfxCwtTdPlan *plan;
    AfxCwtTdPair **dPairs;
    unsigned char **dImage;
    int **hostKs;
} TdSlot;

static void td_slot_clear(TdSlot t) {
    afxdev_free(*t.dPairs);
    afxdev_free(*t.dImage);
    *t.dPairs = NULL;
    *t.dImage = NULL;
    free(*t.hostKs);
    *t.hostKs = NULL;
    memset(t.plan, 0, sizeof(*t.plan));
}

/* pairs (longest first), matrix-core images and the device copies of `nc` sorted candidates; the slot stays empty
 * (AFX_OK, plan->nPairs 0) when the launch would refuse the plan */
static int td_upload(CWTObj o, const TdCand *cand, int nc, TdSlot t) {
    const int nPairs = (nc + 1) / 2;
    AfxCwtTdPair *pairs = (AfxCwtTdPair *)calloc((size_t)nPairs, sizeof(AfxCwtTdPair));
    unsigned char *blob = NULL;
    float *G = NULL;
    int st = pairs ? AFX_OK : AFX_ERR_NOMEM;
    size_t blobBytes = 0;
    int maxKs = 0;
    for (int p = 0; p < nPairs && st == AFX_OK; p++) {
        const TdCand *a = &cand[2 * p], *b = 2 * p + 1 < nc ? &cand[2 * p + 1] : NULL;
        const int kh = (a->kh + 7) & ~7; /* a is the longer one */
        int kt = (2 * kh + 8 + 31) & ~31; /* K steps come in pairs, at least four (afx_cwt_td.hip) */
        if (kt < 64) kt = 64;
        pairs[p].scale[0] = a->scale;
        pairs[p].scale[1] = b ? b->scale : -1;
        pairs[p].kh = kh;
        pairs[p].ks = kt / 16;
        pairs[p].img = (long long)blobBytes;
        blobBytes += (size_t)2 * pairs[p].ks * 1024;
        if (pairs[p].ks > maxKs) maxKs = pairs[p].ks;
    }
    if (st == AFX_OK) {
        blob = (unsigned char *)malloc(blobBytes);
        G = (float *)malloc(sizeof(float) * 32 * (size_t)(16 * maxKs));
        if (!blob || !G) st = AFX_ERR_NOMEM;
    }
    for (int p = 0; p < nPairs && st == AFX_OK; p++) {
        const int kt = 16 * pairs[p].ks, kh = pairs[p].kh;
        memset(G, 0, sizeof(float) * 32 * (size_t)kt);
        for (int c = 0; c < 32; c++) {
            const TdCand *sc = (c >> 4) == 0 ? &cand[2 * p] : (2 * p + 1 < nc ? &cand[2 * p + 1] : NULL);
            if (!sc) continue;
            const int part = (c >> 3) & 1, ph = c & 7;
            for (int m = 0; m < kt; m++) {
                const int tt = ph + kh - m; /* y[n0 + 8 i + ph] = sum_m win[8 i + m] g[ph + kh - m] */
                if (tt < -sc->kh || tt > sc->kh) continue;
                G[(size_t)m * 32 + c] = (float)(part ? sc->im[tt + sc->kh] : sc->re[tt + sc->kh]);
            }
            /* the response to a constant (afx_cwt_td.hip takes one out of a window that sits on an offset): that of the
             * WHOLE kernel, the spectrum's bin 0 -- the dropped tails add up coherently on a constant (4e-8 of it, 7e-6 of
             * a quiet row's peak under an offset of 50 x the signal) */
            pairs[p].colSum[c] = (float)(part ? sc->dcIm : sc->dcRe);
        }
        afx_cqt_time_kernel_f16(G, kt, (unsigned short *)(blob + pairs[p].img), pairs[p].colMul);
    }
    if (st == AFX_OK) st = afxdev_malloc((void **)t.dPairs, sizeof(AfxCwtTdPair) * (size_t)nPairs);
    if (st == AFX_OK) st = afxdev_h2d(*t.dPairs, pairs, sizeof(AfxCwtTdPair) * (size_t)nPairs, o->stream);
    if (st == AFX_OK) st = afxdev_malloc((void **)t.dImage, blobBytes);
    if (st == AFX_OK) st = afxdev_h2d(*t.dImage, blob, blobBytes, o->stream);
    if (st == AFX_OK) st = afxdev_stream_sync(o->stream);
    if (st == AFX_OK) {
        *t.hostKs = (int *)malloc(sizeof(int) * (size_t)nPairs);
        if (!*t.hostKs) st = AFX_ERR_NOMEM;
    }
    if (st == AFX_OK) {
        for (int p = 0; p < nPairs; p++) (*t.hostKs)[p] = pairs[p].ks;
        t.plan->pairs = *t.dPairs;
        t.plan->image = *t.dImage;
        t.plan->nPairs = nPairs;
        t.plan->maxKs = maxKs;
        t.plan->hostKs = *t.hostKs;
        t.plan->wrap = o->padLength > 0 ? 0 : 1;
        if (afxk_cwt_td_fits(t.plan, o->dataLength, o->num) != AFX_OK) td_slot_clear(t);
    } else {
        td_slot_clear(t);
    }
    free(pairs);
    free(blob);
    free(G);
    return st;
}

static int cwt_td_plan(CWTObj o, int rL, int *order, int *nWide) {
    /* (AFX_NO_FUSED: the size-generic inverse computes EVERY scale -- no second writer of those rows) */
    if (o->dataLength < 8192 || *nWide < 1 || afxdev_no_fused()) return AFX_OK;
    TdCand *cand = NULL;
    int nc = 0;
    int st = td_candidates(o, rL, order, *nWide, NULL, &cand, &nc);
    if (st == AFX_OK && nc > 0) {
        qsort(cand, (size_t)nc, sizeof(TdCand), td_cand_cmp);
        /* a launch takes AFX_CWT_TD_MAXPAIRS pairs: the candidates beyond that -- the LONGEST kernels, the dearest ones in
         * the time domain -- stay on the FFT path (a bank of 36 bins per octave, or a linear one of 256 scales, has more
         * than 96 short-kernel scales) */
        const int cap = 2 * AFX_CWT_TD_MAXPAIRS;
        if (nc > cap) {
            for (int c = 0; c < nc - cap; c++) {
                free(cand[c].re);
                free(cand[c].im);
            }
            memmove(cand, cand + (nc - cap), sizeof(TdCand) * (size_t)cap);
            memset(cand + cap, 0, sizeof(TdCand) * (size_t)(nc - cap));
            nc = cap;
        }
        const TdSlot slot = {&o->td, &o->dTdPairs, &o->dTdImage, &o->tdKs};
        st = td_upload(o, cand, nc, slot);
        if (st == AFX_OK && o->td.nPairs > 0) {
            /* order := [time-domain scales, pair by pair | the other two-pass scales | narrow-band classes] */
            int *rest = (int *)malloc(sizeof(int) * (size_t)*nWide);
            if (!rest) st = AFX_ERR_NOMEM;
            int nr = 0;
            for (int w = 0; w < *nWide && rest; w++) {
                int taken = 0;
                for (int c = 0; c < nc; c++) taken |= cand[c].scale == order[w];
                if (!taken) rest[nr++] = order[w];
            }
            if (rest) {
                for (int c = 0; c < nc; c++) order[c] = cand[c].scale;
                memcpy(order + nc, rest, sizeof(int) * (size_t)nr);
                o->dims.nTd = nc;
                *nWide = nr;
            }
            free(rest);
        }
    }
    td_cand_free(cand, nc);
    return st;
}

/* cwtObj_enableDet: the same nTd scales with the derivative bank's kernels -- all of them or none (the order list is
 * shared with the plain transform; a scale whose weighted kernel no longer qualifies keeps every derivative scale on
 * the two-pass path) */
static int cwt_td_plan_det(CWTObj o, const float *weights) {
    if (o->dims.nTd < 1 || !o->dims.td || !o->hOrder) return AFX_OK;
    {
        const char *e = getenv("AFX_CWT_TD_DET"); /* "0": the derivative scales keep round 3's two-pass path (A/B, tools/wsst_traffic.py) */
        if (e && e[0] == '0') return AFX_OK;
    }
    int rL = 0;
    while ((1LL << rL) < o->fftLength) rL++;
    TdCand *cand = NULL;
    int nc = 0;
    int st = td_candidates(o, rL, o->hOrder, o->dims.nTd, weights, &cand, &nc);
    if (st == AFX_OK && nc == o->dims.nTd) {
        qsort(cand, (size_t)nc, sizeof(TdCand), td_cand_cmp);
        const TdSlot slot = {&o->tdDet, &o->dTdPairsDet, &o->dTdImageDet, &o->tdKsDet};
        st = td_upload(o, cand, nc, slot);
        if (st == AFX_OK && o->tdDet.nPairs > 0) o->dims.tdDet = &o->tdDet;
    }
    td_cand_free(cand, nc);
    return st;
}

static int cwt_create(CWTObj *cwtObj, const struct OpaqueCWT *proto, int rL, const float *customBank,
                      const float *customFre, const int *customBin) {
    const int num = proto->num, D = proto->dataLength, pad = proto->padLength;
    const long long fftLength = proto->fftLength;
    *cwtObj = NULL;
    int st = afxdev_ensure();
    if (st != AFX_OK) return st;
    CWTObj o = (CWTObj)calloc(1, sizeof(struct OpaqueCWT));
    if (!o) return AFX_ERR_NOMEM;
    *o = *proto;
    o->dims.r1 = rL / 2;
    o->dims.r2 = rL - rL / 2;
    o->dims.dataLength = D;
    o->dims.pad = pad;
    {
        int c = 8192 >> o->dims.r1; /* <= 64 KB of LDS per column tile */
        if (c > 16) c = 16; /* 16 float2 = one 128-byte line per row of a tile */
        if (c > (1 << o->dims.r2)) c = 1 << o->dims.r2;
        if (c < 1) c = 1;
        o->dims.tileCols = c;
    }
    o->freBandArr = (float *)calloc((size_t)num + 2, sizeof(float));
    o->binBandArr = (int *)calloc((size_t)num + 2, sizeof(int));
    o->hBank = (float *)calloc((size_t)num * fftLength, sizeof(float));
    float *tw = NULL, *bankT = NULL;
    if (!o->freBandArr || !o->binBandArr || !o->hBank) st = AFX_ERR_NOMEM;
    if (st == AFX_OK) {
        if (customBank) {
            memcpy(o->hBank, customBank, sizeof(float) * (size_t)num * fftLength);
            memcpy(o->freBandArr, customFre, sizeof(float) * (size_t)num);
            memcpy(o->binBandArr, customBin, sizeof(int) * (size_t)num);
        } else {
            st = build_bank(o, o->hBank);
        }
        if (st == AFX_OK) bankT = to_transposed(o->hBank, num, o->dims.r1, o->dims.r2, NULL);
        tw = afx_twiddle_table((int)fftLength);
        if (st == AFX_OK && (!bankT || !tw)) st = AFX_ERR_NOMEM;
    }
    const size_t L = (size_t)fftLength;
    if (st == AFX_OK) st = afxdev_stream_create(&o->stream);
    if (st == AFX_OK) st = afxdev_malloc((void **)&o->dTw, sizeof(float) * (L < 2 ? 2 : L));
    if (st == AFX_OK) st = afxdev_h2d(o->dTw, tw, sizeof(float) * (L < 2 ? 2 : L), o->stream);
    if (st == AFX_OK && o->dims.r1 == 8 && o->dims.r2 == 9) {
        /* [8][64] W_512^(lane d) | [8][8] W_64^(c d) | [16][16] W_256^(g p), in double, rounded once */
        float *ft = (float *)malloc(sizeof(float) * AFX_CWT_FASTTW_FLOATS);
        if (!ft) st = AFX_ERR_NOMEM;
        if (st == AFX_OK) {
            int q = 0;
            for (int d = 0; d < 8; d++)
                for (int l = 0; l < 64; l++, q++) {
                    ft[2 * q] = (float)cos(2.0 * M_PI * (l * d) / 512.0);
                    ft[2 * q + 1] = (float)-sin(2.0 * M_PI * (l * d) / 512.0);
                }
            for (int d = 0; d < 8; d++)
                for (int c = 0; c < 8; c++, q++) {
                    ft[2 * q] = (float)cos(2.0 * M_PI * (c * d) / 64.0);
                    ft[2 * q + 1] = (float)-sin(2.0 * M_PI * (c * d) / 64.0);
                }
            for (int pp = 0; pp < 16; pp++)
                for (int g2 = 0; g2 < 16; g2++, q++) {
                    ft[2 * q] = (float)cos(2.0 * M_PI * (g2 * pp) / 256.0);
                    ft[2 * q + 1] = (float)-sin(2.0 * M_PI * (g2 * pp) / 256.0);
                }
            /* support of each wavelet in the transposed layout: k = k1 + 2^r1 k2, so the rows
             * k2 in [kmin >> r1, (kmax >> r1) + 1) hold every non-zero; the row pass skips the
             * rest (exact: those products are zeros in the reference too) */
            int *sup = (int *)malloc(sizeof(int) * 2 * (size_t)num);
            if (sup) {
                afx_cwt_support_host(o->hBank, num, fftLength, o->dims.r1, sup);
                st = afxdev_malloc((void **)&o->dSupport, sizeof(int) * 2 * (size_t)num);
                if (st == AFX_OK) st = afxdev_h2d(o->dSupport, sup, sizeof(int) * 2 * (size_t)num, o->stream);
                if (st == AFX_OK) st = afxdev_stream_sync(o->stream);
                o->dims.support = o->dSupport;
                /* narrow-band scales (<= maxR rows of the transposed spectrum hold every
                 * non-zero) skip the row pass: list the wide scales first, then the classes
                 * R = 2, 4, 8, 16, 20, 24, 32 (afx_device.h).  AFX_CWT_NARROW_MAX bounds the widest class used
                 * (0: every scale takes both passes). */
                int *order = (int *)malloc(sizeof(int) * (size_t)num);
                const char *em = getenv("AFX_CWT_NARROW_MAX");
                const int maxR = em ? atoi(em) : AFX_CWT_NARROW_MAX_DEFAULT;
                if (order && st == AFX_OK && maxR >= 2) {
                    afx_cwt_classify_host(sup, num, maxR, order, &o->dims.nWide, o->dims.nNarrow);
                    if (st == AFX_OK) st = cwt_td_plan(o, rL, order, &o->dims.nWide);
                    if (o->dims.nTd > 0) {
                        o->dims.td = &o->td;
                        o->hOrder = (int *)malloc(sizeof(int) * (size_t)o->dims.nTd);
                        if (!o->hOrder) st = AFX_ERR_NOMEM;
                        else memcpy(o->hOrder, order, sizeof(int) * (size_t)o->dims.nTd);
                    }
                    /* device image: order[num] followed by the (scale, first support row) pairs */
                    int *img = (int *)malloc(sizeof(int) * 3 * (size_t)num);
                    if (!img) st = AFX_ERR_NOMEM;
                    for (int i = 0; i < num && img; i++) {
                        img[i] = order[i];
                        img[num + 2 * i] = order[i];
                        img[num + 2 * i + 1] = sup[2 * order[i]];
                    }
                    /* the pairs are read as int2: keep them 8-byte aligned */
                    const size_t pairOff = ((size_t)num + 1) & ~(size_t)1;
                    if (st == AFX_OK) st = afxdev_malloc((void **)&o->dOrder, sizeof(int) * (pairOff + 2 * (size_t)num));
                    if (st == AFX_OK) st = afxdev_h2d(o->dOrder, img, sizeof(int) * (size_t)num, o->stream);
                    if (st == AFX_OK)
                        st = afxdev_h2d(o->dOrder + pairOff, img + num, sizeof(int) * 2 * (size_t)num, o->stream);
                    if (st == AFX_OK) st = afxdev_stream_sync(o->stream);
                    if (st == AFX_OK) {
                        o->dims.order = o->dOrder;
                        o->dims.orderLo = o->dOrder + pairOff;
                    }
                    free(img);
                }
                free(order);
                free(sup);
            }
            if (st == AFX_OK) st = afxdev_malloc((void **)&o->dFastTw, sizeof(float) * AFX_CWT_FASTTW_FLOATS);
            if (st == AFX_OK) st = afxdev_h2d(o->dFastTw, ft, sizeof(float) * AFX_CWT_FASTTW_FLOATS, o->stream);
            if (st == AFX_OK) st = afxdev_stream_sync(o->stream);
            o->dims.fastTw = o->dFastTw;
        }
        free(ft);
    }
    if (st == AFX_OK) st = afxdev_malloc((void **)&o->dBankT, sizeof(float) * num * L);
    if (st == AFX_OK) st = afxdev_h2d(o->dBankT, bankT, sizeof(float) * num * L, o->stream);
    if (st == AFX_OK && fftLength <= 16384 && !afxdev_no_fused()) {
        st = afxdev_malloc((void **)&o->dBankN, sizeof(float) * num * L);
        if (st == AFX_OK) st = afxdev_h2d(o->dBankN, o->hBank, sizeof(float) * num * L, o->stream);
    }
    if (st == AFX_OK) st = afxdev_malloc((void **)&o->dX, sizeof(float) * (size_t)D);
    if (st == AFX_OK) st = afxdev_malloc((void **)&o->dA, sizeof(float) * 2 * L);
    if (st == AFX_OK) st = afxdev_malloc((void **)&o->dXt, sizeof(float) * 2 * L);
    if (st == AFX_OK) st = afxdev_malloc((void **)&o->dB, sizeof(float) * 2 * L * num);
    if (st == AFX_OK) st = afxdev_malloc((void **)&o->dOut, sizeof(float) * 2 * (size_t)D * num);
    if (st == AFX_OK) st = afxdev_stream_sync(o->stream);
    free(tw);
    free(bankT);
    if (st != AFX_OK) {
        cwtObj_free(o);
        return st;
    }
    *cwtObj = o;
    return 0;
}

/* Support of every wavelet in the transposed spectrum layout (frequency k = k1 + 2^r1 k2):
 * sup[2i], sup[2i+1] = the range of rows k2 that holds all non-zeros of scale i ((0,0) for an
 * all-zero row).  Host-only, exported for tests. */
void afx_cwt_support_host(const float *bank, int num, long long fftLength, int r1, int *sup) {
    for (int i = 0; i < num; i++) {
        long long kmin = -1, kmax = -1;
        const float *row = bank + (size_t)i * fftLength;
        for (long long k = 0; k < fftLength; k++)
            if (row[k] != 0.f) {
                if (kmin < 0) kmin = k;
                kmax = k;
            }
        sup[2 * i] = kmin < 0 ? 0 : (int)(kmin >> r1);
        sup[2 * i + 1] = kmin < 0 ? 0 : (int)(kmax >> r1) + 1;
    }
}

/* Execution order of the scales in the register-FFT inverse (afx_device.h): scales whose support
 * spans more than maxR rows ("wide": row pass + column pass, or the time-domain kernel) first, then the
 * narrow-band classes R = 2, 4, 8, 16 (support width <= R, > R/2) and the two-block classes R = 20, 24, 32, each
 * in ascending scale order.  maxR is clamped to 32; maxR < 2 makes every scale wide. */
void afx_cwt_classify_host(const int *sup, int num, int maxR, int *order, int *nWide, int nNarrow[7]) {
    if (maxR > 32) maxR = 32;
    int q = 0;
    for (int cls = -1; cls < 7; cls++) {
        int n = 0;
        for (int i = 0; i < num; i++) {
            const int w = sup[2 * i + 1] - sup[2 * i];
            int c = -1;
            if (w <= maxR && maxR >= 2) c = w <= 2 ? 0 : w <= 4 ? 1 : w <= 8 ? 2 : w <= 16 ? 3 : w <= 20 ? 4 : w <= 24 ? 5 : 6;
            if (c == cls) order[q++] = i, n++;
        }
        if (cls < 0) *nWide = n;
        else nNarrow[cls] = n;
    }
}

float *cwtObj_getFreBandArr(CWTObj o) { return o ? o->freBandArr : NULL; }
int *cwtObj_getBinBandArr(CWTObj o) { return o ? o->binBandArr : NULL; }

static void run(CWTObj o, float *dataArr, const float *dBank, int isDet, float *re, float *im,
                const char *who) {
    int st = AFX_OK;
    if (o->lastUsed && o->lastStream != o->stream) {
        st = afxdev_stream_sync(o->lastStream);
        o->lastUsed = 0;
    }
    const int small = o->dBankN != NULL; /* whole transform in LDS: natural-order spectrum in dXt */
    if (st == AFX_OK && dataArr) {
        st = afxdev_h2d(o->dX, dataArr, sizeof(float) * (size_t)o->dataLength, o->stream);
        if (st == AFX_OK && !small)
            st = afxk_cwt_forward(&o->dims, o->dTw, o->dX, 0, 1, o->dA, o->dXt, o->stream);
        if (st == AFX_OK) o->haveSpectrum = 1;
    } else if (!dataArr && !o->haveSpectrum) {
        return; /* nothing to re-use yet */
    }
    const size_t outB = sizeof(float) * (size_t)o->num * o->dataLength;
    float *dRe = o->dOut, *dIm = o->dOut + (size_t)o->num * o->dataLength;
    if (st == AFX_OK && small)
        st = afxk_cwt_small(&o->dims, o->dTw, dataArr ? o->dX : NULL, 0, 1, isDet ? o->dBankDetN : o->dBankN,
                            o->num, isDet, o->dXt, dRe, dIm, o->stream);
    else if (st == AFX_OK) {
        st = afxk_cwt_inverse(&o->dims, o->dTw, o->dXt, dBank, o->num, isDet, 1, o->dB, dRe, dIm,
                              AFX_CWT_WIDE | AFX_CWT_NARROW, o->stream);
        /* the short-kernel scales straight from the signal (dX keeps the last uploaded chunk) */
        const AfxCwtTdPlan *td = isDet ? o->dims.tdDet : o->dims.td;
        if (st == AFX_OK && td && o->dims.nTd > 0)
            st = afxk_cwt_td(td, o->dX, 0, 1, o->dataLength, o->num, dRe, dIm, o->stream, NULL);
    }
    if (st == AFX_OK && re) st = afxdev_d2h(re, dRe, outB, o->stream);
    if (st == AFX_OK && im) st = afxdev_d2h(im, dIm, outB, o->stream);
    if (st == AFX_OK) st = afxdev_stream_sync(o->stream);
    if (st != AFX_OK) {
        o->status = st;
        afxdev_report_failure(who, st);
    }
}

void cwtObj_cwt(CWTObj o, float *dataArr, float *mRealArr3, float *mImageArr3) {
    AFX_ENTER(o);
    if (!o) {
        afxdev_set_error("cwtObj_cwt: NULL object");
        return;
    }
    run(o, dataArr, o->dBankT, 0, mRealArr3, mImageArr3, "cwtObj_cwt");
}

/* chunks of 2^radix2Exp samples already in HBM (chunk c at dData + c*chunkStride) ->
 * dReal/dImag [chunks][num][2^radix2Exp], left in HBM (include/afx_batch.h).  The
 * spectrum / intermediate scratch belongs to the object, so chunks run back to back on
 * `hipStream`. */
static int cwt_batch_device(CWTObj o, const float *dData, int chunks, long long chunkStride,
                            const float *dBank, int isDet, float *dReal, float *dImag,
                            void *hipStream, const char *who) {
    if (!o || !dData || !dReal || !dImag || chunks <= 0 || chunkStride < 0) {
        afxdev_set_error("%s: bad argument", who);
        return AFX_ERR_ARG;
    }
    if (!dBank) {
        afxdev_set_error("%s: cwtObj_enableDet was not called", who);
        return AFX_ERR_ARG;
    }
    int st = AFX_OK;
    if (o->lastUsed && o->lastStream != hipStream) st = afxdev_stream_sync(o->lastStream);
    const size_t plane = (size_t)o->num * o->dataLength;
    const size_t L = (size_t)o->fftLength;
    if (st == AFX_OK && o->dBankN) {
        /* L <= 16384: forward and inverse entirely in LDS, up to 4096 chunks per launch pair */
        const int step = 4096;
        st = afxdev_reserve((void **)&o->dGXt, &o->capGXt, sizeof(float) * 2 * L * (chunks < step ? chunks : step));
        for (int c = 0; c < chunks && st == AFX_OK; c += step) {
            const int n = chunks - c < step ? chunks - c : step;
            st = afxk_cwt_small(&o->dims, o->dTw, dData + (long long)c * chunkStride, chunkStride, n,
                                isDet ? o->dBankDetN : o->dBankN, o->num, isDet, o->dGXt, dReal + c * plane,
                                dImag + c * plane, hipStream);
        }
        o->lastStream = hipStream;
        o->lastUsed = 1;
        if (st != AFX_OK) {
            o->status = st;
            afxdev_report_failure(who, st);
        }
        return st;
    }
    /* Forward transforms of up to 32 chunks share one launch (a single chunk is only
     * 2^r2/tileCols workgroups).  The two-pass part of the inverse runs `group` chunks per launch:
     * its intermediate is group * nTwoPass * L complex (1 MB per scale and chunk at L 2^17; the
     * buffer is indexed by scale, so it is sized for all of them), and it is re-read while still
     * resident in the 256 MB memory-side cache when about 96 MB are in flight per launch pair
     * whatever the transform length: cfg 4 with all 84 scales on two passes: 1 chunk (column
     * pass 22.6 us per chunk at group 1, 32 us at group 4); with the 40 wide scales the
     * narrow-band plan leaves: 2 chunks (+9 %, profiles/r01_cwt_narrowband.txt); 16 chunks at
     * the wrapper's default L = 2^13 (otherwise launch-bound). */
    /* The short-kernel scales of the plain transform run in the time domain (afx_cwt_td.hip) -- from the signal, not
     * from the spectrum: all chunks of the call at once on a side stream of their own, beside the forward transforms
     * and the FFT-path scales (they write disjoint rows), joined before the call returns.  (This overlap is what exposed
     * the operand-select rule of afx_asm.h: packed-f32 instructions with op_sel[0] = 0 and op_sel[1] = 1 are not exact
     * beside a wave that streams v_mfma + ds_read_b128 -- DESIGN.md section 4.3, profiles/r03_pk_add_opsel.txt; with
     * the rule kept every schedule is bit-reproducible.) */
    const AfxCwtTdPlan *tdPlan = isDet ? o->dims.tdDet : o->dims.td; /* (derivative bank: cwt_td_plan_det) */
    const int useTd = tdPlan && o->dims.nTd > 0;
    void *tds = NULL;
    if (st == AFX_OK && useTd) {
        if (!o->chain[2]) st = afxdev_stream_create(&o->chain[2]);
        tds = o->chain[2];
        if (st == AFX_OK) st = afxdev_stream_wait_stream(tds, hipStream);
        if (st == AFX_OK)
            st = afxk_cwt_td(tdPlan, dData, chunkStride, chunks, o->dataLength, o->num, dReal, dImag, tds, NULL);
    }
    const int nTwoPass = o->dims.order ? o->dims.nWide + (useTd ? 0 : o->dims.nTd) : o->num; /* scales that write the intermediate */
    /* (no two-pass scale at all: the group only paces the loop below -- one forward batch) */
    /* (two chains of 48 MB each: round 2's three-stream schedule, +10 % over one stream) */
    int group = nTwoPass > 0 ? (int)(48.0e6 / ((double)nTwoPass * L * 8.0)) : 32;
    if (group < 1) group = 1;
    {
        const char *e = getenv("AFX_CWT_GROUP");
        if (e && atoi(e) > 0) group = atoi(e);
        while (group > 1 && nTwoPass > 0 && (double)group * o->num * L * 8.0 > 4.0e9) group /= 2;
        if (group > chunks) group = chunks;
    }
    int fwdBatch = group > 32 ? group : 32;
    if (fwdBatch > chunks) fwdBatch = chunks;
    if (st == AFX_OK) st = afxdev_reserve((void **)&o->dGA, &o->capGA, sizeof(float) * 2 * L * fwdBatch);
    if (st == AFX_OK) st = afxdev_reserve((void **)&o->dGXt, &o->capGXt, sizeof(float) * 2 * L * fwdBatch);
    const size_t gbFloats = (size_t)2 * L * group * o->num;
    /* independent two-pass chains in flight (each with its own intermediate): 2 measured best on cfg 4
     * with one chunk per launch pair (26.8 k vs 25.5 k chunks/s on one chain of two chunks) */
    const int chains = nTwoPass > 0 ? 2 : 1;
    if (st == AFX_OK && nTwoPass > 0)
        st = afxdev_reserve((void **)&o->dGB, &o->capGB, sizeof(float) * gbFloats * chains);
    /* The narrow-band scales (no intermediate; bound by their instruction stream) run on a side stream
     * beside the two-pass launches of the wide scales (bound by the intermediate's round trip, and
     * short launches with < 2 waves per SIMD): both depend only on the forward transform of the
     * batch.  Joined before the next forward batch overwrites the spectra. */
    void *side = NULL, *pipe = NULL; /* narrow-band side stream */
    if (nTwoPass > 0 && nTwoPass < o->num) {
        side = o->stream != hipStream ? o->stream : o->stream2;
        if (!side) {
            st = afxdev_stream_create(&o->stream2);
            side = o->stream2;
        }
    }
    for (int i = 0; i + 1 < chains && st == AFX_OK; i++)
        if (!o->chain[i]) st = afxdev_stream_create(&o->chain[i]);
    (void)pipe;
    for (int c0 = 0; c0 < chunks && st == AFX_OK; c0 += fwdBatch) {
        const int nf = chunks - c0 < fwdBatch ? chunks - c0 : fwdBatch;
        st = afxk_cwt_forward(&o->dims, o->dTw, dData + (long long)c0 * chunkStride, chunkStride, nf,
                              o->dGA, o->dGXt, hipStream);
        for (int i = 0; i + 1 < chains && st == AFX_OK; i++) st = afxdev_stream_wait_stream(o->chain[i], hipStream);
        if (st == AFX_OK && side) {
            st = afxdev_stream_wait_stream(side, hipStream);
            if (st == AFX_OK)
                st = afxk_cwt_inverse(&o->dims, o->dTw, o->dGXt, dBank, o->num, isDet, nf, NULL,
                                      dReal + (size_t)c0 * plane, dImag + (size_t)c0 * plane, AFX_CWT_NARROW, side);
        }
        /* two-pass launches of consecutive chunk groups alternate between the caller's stream and a
         * second side stream, each with its own intermediate: the row pass of one group overlaps the
         * column pass of the other (each launch alone leaves most CUs under two waves per SIMD) */
        int k = 0;
        for (int c = 0; c < nf && st == AFX_OK; c += group, ++k) {
            const int n = nf - c < group ? nf - c : group;
            const int ch = k % chains;
            void *s = ch ? o->chain[ch - 1] : hipStream;
            float *gb = o->dGB + (size_t)ch * gbFloats;
            st = afxk_cwt_inverse(&o->dims, o->dTw, o->dGXt + 2 * L * (size_t)c, dBank, o->num, isDet, n,
                                  gb, dReal + (size_t)(c0 + c) * plane, dImag + (size_t)(c0 + c) * plane,
                                  AFX_CWT_WIDE, s);
        }
        for (int i = 0; i + 1 < chains && st == AFX_OK; i++) st = afxdev_stream_wait_stream(hipStream, o->chain[i]);
        /* the narrow-band scales have no intermediate: all chunks of the forward batch at once */
        if (st == AFX_OK && !side)
            st = afxk_cwt_inverse(&o->dims, o->dTw, o->dGXt, dBank, o->num, isDet, nf, NULL,
                                  dReal + (size_t)c0 * plane, dImag + (size_t)c0 * plane, AFX_CWT_NARROW,
                                  hipStream);
        if (st == AFX_OK && side) st = afxdev_stream_wait_stream(hipStream, side);
    }
    if (tds) { /* also on the way out of a failed call: the time-domain kernel may still be writing dReal / dImag */
        const int js = st == AFX_OK ? afxdev_stream_wait_stream(hipStream, tds) : afxdev_stream_sync(tds);
        if (st == AFX_OK) st = js;
    }
    o->lastStream = hipStream;
    o->lastUsed = 1;

    if (st != AFX_OK) {
        o->status = st;
        afxdev_report_failure(who, st);
    }
    return st;
}

int cwtObj_cwtBatchDevice(CWTObj o, const float *dData, int chunks, long long chunkStride,
                          float *dReal, float *dImag, void *hipStream) {
    AFX_ENTER(o);
    return cwt_batch_device(o, dData, chunks, chunkStride, o ? o->dBankT : NULL, 0, dReal, dImag,
                            hipStream, "cwtObj_cwtBatchDevice");
}

int cwtObj_cwtDetBatchDevice(CWTObj o, const float *dData, int chunks, long long chunkStride,
                             float *dReal, float *dImag, void *hipStream) {
    AFX_ENTER(o);
    return cwt_batch_device(o, dData, chunks, chunkStride, o ? o->dBankDetT : NULL, 1, dReal, dImag,
                            hipStream, "cwtObj_cwtDetBatchDevice");
}

/* host pointers: dataArr[chunks][2^r] -> mRealArr3/mImageArr3 [chunks][num][2^r]; chunks are
 * streamed through the object's one-chunk staging buffers */
int cwtObj_cwtBatch(CWTObj o, const float *dataArr, int chunks, float *mRealArr3, float *mImageArr3) {
    AFX_ENTER(o);
    if (!o || !dataArr || !mRealArr3 || !mImageArr3 || chunks <= 0) {
        afxdev_set_error("cwtObj_cwtBatch: bad argument");
        return AFX_ERR_ARG;
    }
    const size_t plane = (size_t)o->num * o->dataLength;
    for (int c = 0; c < chunks; c++) {
        o->status = AFX_OK;
        run(o, (float *)dataArr + (size_t)c * o->dataLength, o->dBankT, 0, mRealArr3 + c * plane,
            mImageArr3 + c * plane, "cwtObj_cwtBatch");
        if (o->status != AFX_OK) return o->status;
    }
    return AFX_OK;
}

void cwtObj_enableDet(CWTObj o, int flag) {
    AFX_ENTER(o);
    if (!o || !flag || o->dBankDetT) return;
    /* bank times the angular frequency (cwt_algorithm.c:485-528) */
    const long long L = o->fftLength;
    float *w = (float *)calloc((size_t)L, sizeof(float));
    if (!w) return;
    for (long long i = 0; i <= L / 2; i++) w[i] = (float)(i * 2 * M_PI / L);
    for (long long i = L / 2 + 1, j = L / 2 - 1; i < L && j >= 0; i++, j--) w[i] = -w[j];
    if (o->dBankN && !o->dBankDetN) { /* in-LDS path: natural layout, same products bank * w */
        float *dn = (float *)malloc(sizeof(float) * (size_t)o->num * L);
        if (dn) {
            for (int i = 0; i < o->num; i++)
                for (long long k = 0; k < L; k++) dn[(size_t)i * L + k] = o->hBank[(size_t)i * L + k] * w[k];
            int s2 = afxdev_malloc((void **)&o->dBankDetN, sizeof(float) * (size_t)o->num * L);
            if (s2 == AFX_OK) s2 = afxdev_h2d(o->dBankDetN, dn, sizeof(float) * (size_t)o->num * L, o->stream);
            if (s2 == AFX_OK) s2 = afxdev_stream_sync(o->stream);
            if (s2 != AFX_OK) o->status = s2;
            free(dn);
        }
    }
    float *t = to_transposed(o->hBank, o->num, o->dims.r1, o->dims.r2, w);
    int st = t ? AFX_OK : AFX_ERR_NOMEM;
    if (st == AFX_OK) st = afxdev_malloc((void **)&o->dBankDetT, sizeof(float) * (size_t)o->num * L);
    if (st == AFX_OK) st = afxdev_h2d(o->dBankDetT, t, sizeof(float) * (size_t)o->num * L, o->stream);
    if (st == AFX_OK) st = afxdev_stream_sync(o->stream);
    free(t);
    if (st == AFX_OK) st = cwt_td_plan_det(o, w);
    free(w);
    if (st != AFX_OK) {
        afxdev_free(o->dBankDetT);
        o->dBankDetT = NULL;
        o->status = st;
    }
}

void cwtObj_cwtDet(CWTObj o, float *dataArr, float *mRealArr3, float *mImageArr3) {
    AFX_ENTER(o);
    if (!o) {
        afxdev_set_error("cwtObj_cwtDet: NULL object");
        return;
    }
    if (!o->dBankDetT) return; /* enableDet was not called: the reference does nothing either */
    run(o, dataArr, o->dBankDetT, 1, mRealArr3, mImageArr3, "cwtObj_cwtDet");
}

void cwtObj_free(CWTObj o) {
    if (!o) return;
    if (o->stream) afxdev_stream_sync(o->stream);
    if (o->stream2) afxdev_stream_sync(o->stream2);
    for (int i = 0; i < 3; i++)
        if (o->chain[i]) afxdev_stream_sync(o->chain[i]);
    if (o->lastUsed && o->lastStream) afxdev_stream_sync(o->lastStream); /* the caller's stream may still run our kernels */
    afxdev_free(o->dTw);
    afxdev_free(o->dBankT);
    afxdev_free(o->dBankDetT);
    afxdev_free(o->dX);
    afxdev_free(o->dA);
    afxdev_free(o->dXt);
    afxdev_free(o->dB);
    afxdev_free(o->dFastTw);
    afxdev_free(o->dBankN);
    afxdev_free(o->dBankDetN);
    afxdev_free(o->dSupport);
    afxdev_free(o->dOrder);
    afxdev_free(o->dTdPairs);
    afxdev_free(o->dTdImage);
    afxdev_free(o->dGA);
    afxdev_free(o->dGXt);
    afxdev_free(o->dGB);
    afxdev_free(o->dOut);
    afxdev_stream_destroy(o->stream2);
    for (int i = 0; i < 3; i++) afxdev_stream_destroy(o->chain[i]);
    afxdev_stream_destroy(o->stream);
    free(o->freBandArr);
    free(o->binBandArr);
    free(o->hBank);
    free(o->tdKs);
    afxdev_free(o->dTdPairsDet);
    afxdev_free(o->dTdImageDet);
    free(o->tdKsDet);
    free(o->hOrder);
    free(o);
}
