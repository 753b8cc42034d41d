#ifdef AFX_EXPERIMENTS
#define _POSIX_C_SOURCE 199309L /* clock_gettime of the measurement build */
#endif
/* afx_spectrogram.c -- the spectrogram object (C host side) behind
 * include/spectrogram_algorithm.h.
 *
 * Parameter semantics follow src/spectrogram_algorithm.c:326-853 (defaults, range checks, status
 * codes, the band arrays of every scale); execution re-uses the BFT execution plan (afx_bft.c):
 * the object owns a real-result BFT core whose bank is the auditory bank of the scale, the
 * Gaussian STFT-chroma bank, or the log-chroma base bank -- so the mel / bark / erb spectrograms
 * run on the same fused STFT -> filter-bank kernels as bftObj_bft.  Chroma scales add a fold GEMM
 * and the per-frame normalisation kernel.  There is no CPU compute path.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "afx_batch.h"
#include "afx_device.h"
#include "afx_host.h"
#include "afx_objects.h"
#include "spectrogram_algorithm.h"

struct OpaqueSpectrogram {
    int fftLength, radix2Exp, num;
    int samplate;
    float lowFre, highFre;
    int lowIndex, highIndex, binPerOctave;
    int baseNum;   /* chroma: bins in [lowIndex, highIndex]; log-chroma: rows of the base bank */
    float baseFre;
    WindowType windowType;
    int slideLength, isContinue;
    SpectralDataType dataType;
    SpectralFilterBankScaleType scale;
    SpectralFilterBankStyleType style;
    SpectralFilterBankNormalType normal;
    ChromaDataNormalType dataNormType;
    float normValue;
    int deepOrder, isDebug;
    float *freBandArr;
    int *binBandArr;
    int timeLength; /* frames of the last spectrogram call (the cepstra / deconv row count) */

    BFTObj core;             /* STFT -> [coreRows] bank rows (or the linear bin slice) */
    int coreRows;
    float *dFold;            /* log-chroma: device [num, baseNum] 0/1 fold matrix */
    struct OpaqueSTFT tail;  /* host-only: the streaming tail state machine (afx_stft.c) */
    XXCCObj xxcc;            /* lazily built cepstra plan over num bands */
    float *dDevTw;           /* deconv: twiddles of the 2^devRadix transform */
    int devRadix;
    /* grow-only device scratch of the host-pointer calls */
    float *dX, *dOut, *dTmp, *dSpec;
    size_t capX, capOut, capTmp, capSpec;
    int status;
};

static int is_octave_like(SpectralFilterBankScaleType s) {
    return s == SpectralFilterBankScale_Octave || s == SpectralFilterBankScale_Log ||
           s == SpectralFilterBankScale_LogChroma || s == SpectralFilterBankScale_Deep ||
           s == SpectralFilterBankScale_DeepChroma;
}

static int is_chroma_like(SpectralFilterBankScaleType s) {
    return s == SpectralFilterBankScale_Chroma || s == SpectralFilterBankScale_LogChroma;
}

/* spectrogram_algorithm.c:3210-3223, :3276-3285 */
static int cal_base_num(float lowFre, float highFre, float bpo) {
    const float midi1 = afx_fre_to_log(lowFre, bpo), midi2 = afx_fre_to_log(highFre, bpo);
    return (int)(midi2 - midi1 + 1);
}

static float cal_base_fre(float lowFre, float bpo) { return afx_log_to_fre(afx_fre_to_log(lowFre, bpo), bpo); }

/* Gaussian STFT-chroma bank, operation for operation (chroma_filterBank.c:13-174) */
float *afx_chroma_stft_bank(int num, int fftLength, int samplate) {
    const float center = 5, width = 2, baseFre = 440.0f;
    const int F = fftLength / 2 + 1, n = num / 12;
    float *oct = (float *)calloc((size_t)fftLength, sizeof(float));
    float *wid = (float *)calloc((size_t)fftLength, sizeof(float));
    float *m1 = (float *)calloc((size_t)num * fftLength, sizeof(float));
    float *colSq = (float *)calloc((size_t)fftLength, sizeof(float));
    float *bank = (float *)calloc((size_t)num * F, sizeof(float));
    if (!oct || !wid || !m1 || !colSq || !bank) {
        free(oct); free(wid); free(m1); free(colSq); free(bank);
        return NULL;
    }
    for (int i = 1; i < fftLength; i++) {
        const float fre = (float)(1.0 * i / fftLength * samplate);
        oct[i] = num * logf(fre / (baseFre / 16)) / logf(2.0f);
    }
    oct[0] = (float)(oct[1] - 1.5 * num);
    for (int i = 1; i < fftLength; i++) {
        const float v = oct[i] - oct[i - 1];
        wid[i - 1] = v > 1 ? v : 1;
    }
    wid[fftLength - 1] = 1;
    const int half = (int)roundf((float)(num / 2.0));
    for (int i = 0; i < num; i++) {
        for (int j = 0; j < fftLength; j++) {
            float v1 = oct[j] - i;
            v1 = v1 + half + 10 * num;
            const int k = (int)floorf(v1 / num);
            float v2 = v1 - k * num;
            v2 = v2 - half;
            /* gauss */
            float g = 2 * v2 / wid[j];
            g = (float)(-0.5 * g * g);
            m1[(size_t)i * fftLength + j] = expf(g);
        }
    }
    for (int j = 0; j < fftLength; j++) { /* __msum over rows of the squares, double accumulator */
        double acc = 0;
        for (int i = 0; i < num; i++) {
            const float e = m1[(size_t)i * fftLength + j];
            acc += (float)(e * e);
        }
        colSq[j] = (float)acc;
    }
    /* normalise, scale by the octave weighting, keep F columns; rows rotated by 3n (:139-171) */
    for (int i = 0; i < num; i++) {
        const int dst = (i >= 3 * n) ? i - 3 * n : i + (num - 3 * n);
        for (int j = 0; j < F; j++) {
            float v = m1[(size_t)i * fftLength + j] / sqrtf(colSq[j]);
            float s = (oct[j] / num - center) / width;
            s = (float)(-0.5 * s * s);
            s = expf(s);
            bank[(size_t)dst * F + j] = v * s;
        }
    }
    free(oct); free(wid); free(m1); free(colSq);
    return bank;
}

static int new_preset(SpectrogramObj *obj, int num, int samplate, int radix2Exp, int *isContinue,
                      SpectralFilterBankScaleType scale) {
    int sr = samplate, r = radix2Exp;
    return spectrogramObj_new(obj, num, &sr, NULL, NULL, NULL, &r, NULL, NULL, isContinue, NULL, &scale,
                              NULL, NULL);
}

int spectrogramObj_newLinear(SpectrogramObj *o, int samplate, int radix2Exp, int *isContinue) {
    return new_preset(o, 2, samplate, radix2Exp, isContinue, SpectralFilterBankScale_Linear);
}
int spectrogramObj_newMel(SpectrogramObj *o, int num, int samplate, int radix2Exp, int *isContinue) {
    return new_preset(o, num, samplate, radix2Exp, isContinue, SpectralFilterBankScale_Mel);
}
int spectrogramObj_newBark(SpectrogramObj *o, int num, int samplate, int radix2Exp, int *isContinue) {
    return new_preset(o, num, samplate, radix2Exp, isContinue, SpectralFilterBankScale_Bark);
}
int spectrogramObj_newErb(SpectrogramObj *o, int num, int samplate, int radix2Exp, int *isContinue) {
    return new_preset(o, num, samplate, radix2Exp, isContinue, SpectralFilterBankScale_Erb);
}
int spectrogramObj_newChroma(SpectrogramObj *o, int samplate, int radix2Exp, int *isContinue) {
    return new_preset(o, 12, samplate, radix2Exp, isContinue, SpectralFilterBankScale_Chroma);
}
int spectrogramObj_newDeep(SpectrogramObj *o, int num, int samplate, int radix2Exp, int *isContinue) {
    return new_preset(o, num, samplate, radix2Exp, isContinue, SpectralFilterBankScale_Deep);
}
int spectrogramObj_newDeepChroma(SpectrogramObj *o, int samplate, int radix2Exp, int *isContinue) {
    return new_preset(o, 12, samplate, radix2Exp, isContinue, SpectralFilterBankScale_DeepChroma);
}

static int new_impl(SpectrogramObj *spectrogramObj, int num, int *samplate, float *lowFre,
                    float *highFre, int *binPerOctave, int *radix2Exp, WindowType *windowType,
                    int *slideLength, int *isContinue, SpectralDataType *dataType,
                    SpectralFilterBankScaleType *filterScaleType,
                    SpectralFilterBankStyleType *filterStyleType,
                    SpectralFilterBankNormalType *filterNormalType, float *planOnly);

int spectrogramObj_new(SpectrogramObj *spectrogramObj, int num, int *samplate, float *lowFre,
                       float *highFre, int *binPerOctave, int *radix2Exp, WindowType *windowType,
                       int *slideLength, int *isContinue, SpectralDataType *dataType,
                       SpectralFilterBankScaleType *filterScaleType,
                       SpectralFilterBankStyleType *filterStyleType,
                       SpectralFilterBankNormalType *filterNormalType) {
    return new_impl(spectrogramObj, num, samplate, lowFre, highFre, binPerOctave, radix2Exp, windowType,
                    slideLength, isContinue, dataType, filterScaleType, filterStyleType, filterNormalType,
                    NULL);
}

/* test hook (tests/test_spectrogram_host.py): the parameter resolution of spectrogramObj_new
 * without a device -> plan[0..9] = num, lowFre, highFre, lowIndex, highIndex, binPerOctave,
 * baseNum, baseFre, windowType, slideLength */
int afx_test_spectrogram_plan(int num, int *samplate, float *lowFre, float *highFre, int *binPerOctave,
                              int *radix2Exp, WindowType *windowType, int *slideLength,
                              SpectralFilterBankScaleType *filterScaleType, float *plan) {
    SpectrogramObj dummy = NULL;
    return new_impl(&dummy, num, samplate, lowFre, highFre, binPerOctave, radix2Exp, windowType, slideLength,
                    NULL, NULL, filterScaleType, NULL, NULL, plan);
}

static int new_impl(SpectrogramObj *spectrogramObj, int num, int *samplate, float *lowFre,
                    float *highFre, int *binPerOctave, int *radix2Exp, WindowType *windowType,
                    int *slideLength, int *isContinue, SpectralDataType *dataType,
                    SpectralFilterBankScaleType *filterScaleType,
                    SpectralFilterBankStyleType *filterStyleType,
                    SpectralFilterBankNormalType *filterNormalType, float *planOnly) {
    int r = 12, sr = 32000, bpo = 12, hop, fftLength, cont = 0;
    float low = 0, high = 0, baseFre = 0;
    int lowIndex = 0, highIndex = 0, baseNum = 0;
    WindowType win = Window_Hann;
    SpectralDataType dtype = SpectralData_Power;
    SpectralFilterBankScaleType scale = SpectralFilterBankScale_Linear;
    SpectralFilterBankStyleType style = SpectralFilterBankStyle_Slaney;
    SpectralFilterBankNormalType normal = SpectralFilterBankNormal_None;

    if (!spectrogramObj) return -1;
    *spectrogramObj = NULL;

    /* --- validation & defaults in the reference's order (spectrogram_algorithm.c:365-526) */
    if (radix2Exp) {
        r = *radix2Exp;
        if (r < 1 || r > 30) {
            printf("radix2Exp is error!\n");
            return -100;
        }
    }
    fftLength = 1 << r;
    if (samplate && *samplate > 0 && *samplate <= 196000) sr = *samplate;
    if (dataType) dtype = *dataType;
    if (filterScaleType) scale = *filterScaleType;
    if (filterStyleType) style = *filterStyleType;
    if (filterNormalType) normal = *filterNormalType;

    high = (float)(sr / 2.0);
    if (lowFre && *lowFre >= 0 && *lowFre < sr / 2.0) low = *lowFre;
    if (low == 0 && is_octave_like(scale)) {
        low = (float)(powf(2, (float)(-45 / 12.0)) * 440);
        high = (float)(powf(2, (float)(38 / 12.0)) * 440);
    }
    if (highFre && *highFre > 0 && *highFre <= sr / 2.0) high = *highFre;
    if (high < low) {
        low = 0;
        high = (float)(sr / 2.0);
        if (is_octave_like(scale)) {
            low = (float)(powf(2, (float)(-45 / 12.0)) * 440);
            high = (float)(powf(2, (float)(38 / 12.0)) * 440);
        }
    }
    if (binPerOctave && *binPerOctave > 0) bpo = *binPerOctave;
    if (bpo % 12 != 0) bpo = 12;
    if (scale == SpectralFilterBankScale_Linear || scale == SpectralFilterBankScale_Chroma) {
        const float det = sr / (float)fftLength;
        lowIndex = (int)roundf(low / det);
        highIndex = (int)roundf(high / det);
    }
    const int deep = (scale == SpectralFilterBankScale_Deep || scale == SpectralFilterBankScale_DeepChroma);
    if (deep) win = Window_Hamm;
    if (windowType) win = *windowType;
    if (deep && win > Window_Hamm) win = Window_Hamm;
    hop = fftLength / 4;
    if (hop < 1) hop = 1; /* fftLength 2 */
    if (slideLength && *slideLength > 0) hop = *slideLength;
    if (isContinue) cont = *isContinue;

    if (scale == SpectralFilterBankScale_Linear) {
        num = highIndex - lowIndex + 1;
    } else if (scale == SpectralFilterBankScale_Octave) {
        afx_auditory_revise_log(num, low, high, bpo, 1, &low, &high);
        if (high > sr / 2.0) {
            printf("scale log: lowFre and num is large, overflow error!\n");
            return -1;
        }
        baseNum = num;
        baseFre = low;
    } else if (scale == SpectralFilterBankScale_Deep) {
        baseNum = num;
        baseFre = cal_base_fre(low, 12);
    } else if (scale == SpectralFilterBankScale_Chroma) {
        if (num < 12 || num % 12 != 0) num = 12;
        baseNum = highIndex - lowIndex + 1;
    } else if (scale == SpectralFilterBankScale_LogChroma) {
        if (num <= 0) num = 12;
        else if (num > bpo || bpo % num != 0) num = 12;
        baseNum = cal_base_num(low, high, (float)bpo);
        baseFre = cal_base_fre(low, (float)bpo);
    } else if (scale == SpectralFilterBankScale_DeepChroma) {
        if (num < 12 || num % 12 != 0) num = 12;
        baseNum = cal_base_num(low, high, 12);
        baseFre = cal_base_fre(low, 12);
    }
    if (num < 2 || num > fftLength / 2 + 1) {
        printf("num is error!\n");
        return -1;
    }
    if (planOnly) {
        planOnly[0] = (float)num; planOnly[1] = low; planOnly[2] = high; planOnly[3] = (float)lowIndex;
        planOnly[4] = (float)highIndex; planOnly[5] = (float)bpo; planOnly[6] = (float)baseNum;
        planOnly[7] = baseFre; planOnly[8] = (float)win; planOnly[9] = (float)hop;
        return 0;
    }
    if (deep) {
        /* the salience ("deep") scales are a peak-picking model outside the batched
         * time-frequency path; refuse instead of computing something else */
        afxdev_set_error("spectrogramObj_new: the deep / deep-chroma scales are not implemented by the MI355X backend");
        return AFX_ERR_UNSUPPORTED;
    }
    if ((int)scale < 0 || (int)scale > (int)SpectralFilterBankScale_DeepChroma) {
        afxdev_set_error("spectrogramObj_new: unknown scale type %d", (int)scale);
        return AFX_ERR_UNSUPPORTED;
    }
    if (scale == SpectralFilterBankScale_LogChroma && (baseNum < 2 || baseNum > fftLength / 2 + 1)) {
        afxdev_set_error("spectrogramObj_new: log-chroma base bank of %d rows", baseNum);
        return AFX_ERR_ARG;
    }

    SpectrogramObj o = (SpectrogramObj)calloc(1, sizeof(struct OpaqueSpectrogram));
    if (!o) return AFX_ERR_NOMEM;
    o->fftLength = fftLength;
    o->radix2Exp = r;
    o->num = num;
    o->samplate = sr;
    o->lowFre = low;
    o->highFre = high;
    o->lowIndex = lowIndex;
    o->highIndex = highIndex;
    o->binPerOctave = bpo;
    o->baseNum = baseNum;
    o->baseFre = baseFre;
    o->windowType = win;
    o->slideLength = hop;
    o->isContinue = cont;
    o->dataType = dtype;
    o->scale = scale;
    o->style = style;
    o->normal = normal;
    o->dataNormType = ChromaDataNormal_Max;
    o->normValue = 1;
    o->deepOrder = 1;
    o->tail.radix2Exp = r;
    o->tail.fftLength = fftLength;
    o->tail.slideLength = hop;
    o->tail.isContinue = cont;
    o->tail.tailDataArr = (float *)calloc((size_t)fftLength, sizeof(float));

    /* --- the execution plan (spectrogram_algorithm.c:587-790) */
    const int F = fftLength / 2 + 1;
    AfxBftPlan p;
    memset(&p, 0, sizeof(p));
    p.num = num;
    p.radix2Exp = r;
    p.samplate = sr;
    p.lowFre = low;
    p.highFre = high;
    p.lowIndex = lowIndex;
    p.highIndex = highIndex;
    p.binPerOctave = bpo;
    p.windowType = win;
    p.slideLength = hop;
    p.dataType = dtype;
    p.scale = scale;
    p.style = style;
    p.normal = normal;
    float *chromaBank = NULL;
    int st = o->tail.tailDataArr ? AFX_OK : AFX_ERR_NOMEM;
    const int bandLen = (is_chroma_like(scale) ? baseNum : num) + 2;
    o->freBandArr = (float *)calloc((size_t)bandLen, sizeof(float));
    o->binBandArr = (int *)calloc((size_t)bandLen, sizeof(int));
    if (!o->freBandArr || !o->binBandArr) st = AFX_ERR_NOMEM;
    if (st == AFX_OK && scale == SpectralFilterBankScale_Chroma) {
        chromaBank = afx_chroma_stft_bank(num, fftLength, sr);
        if (!chromaBank) st = AFX_ERR_NOMEM;
        /* bins outside [lowIndex, highIndex] are zeroed before the product (:1129-1139) */
        for (int i = 0; i < num && st == AFX_OK; i++)
            for (int j = 0; j < F; j++)
                if (j < lowIndex || j > highIndex) chromaBank[(size_t)i * F + j] = 0.f;
        p.customBank = chromaBank;
    }
    if (scale == SpectralFilterBankScale_LogChroma) p.num = baseNum;
    o->coreRows = p.num;
    if (st == AFX_OK) st = afx_bft_create(&p, &o->core);
    free(chromaBank);
    if (st == AFX_OK) {
        o->core->resultType = 1;
        if (scale == SpectralFilterBankScale_Linear || scale == SpectralFilterBankScale_Chroma) {
            /* __spectrogramObj_calLinearBandArr (:1909-1942): slice of linspace(0, sr/2, F) */
            float *grid = afx_linspace(0, (float)(sr / 2.0), F, 0);
            const int len = (scale == SpectralFilterBankScale_Linear) ? num : baseNum;
            if (!grid) st = AFX_ERR_NOMEM;
            for (int i = 0; i < len && st == AFX_OK && lowIndex + i < F; i++) {
                o->freBandArr[i] = grid[lowIndex + i];
                o->binBandArr[i] = lowIndex + i;
            }
            free(grid);
        } else {
            memcpy(o->freBandArr, o->core->freBandArr, sizeof(float) * (size_t)o->coreRows);
            memcpy(o->binBandArr, o->core->binBandArr, sizeof(int) * (size_t)o->coreRows);
        }
    }
    if (st == AFX_OK && scale == SpectralFilterBankScale_LogChroma) {
        unsigned char *fold = afx_chroma_fold(num, baseNum, bpo, baseFre);
        float *ff = (float *)calloc((size_t)num * baseNum, sizeof(float));
        if (!fold || !ff) st = AFX_ERR_NOMEM;
        for (size_t i = 0; st == AFX_OK && i < (size_t)num * baseNum; i++) ff[i] = fold[i];
        if (st == AFX_OK) st = afxdev_malloc((void **)&o->dFold, sizeof(float) * (size_t)num * baseNum);
        if (st == AFX_OK) st = afxdev_h2d(o->dFold, ff, sizeof(float) * (size_t)num * baseNum, o->core->stream);
        if (st == AFX_OK) st = afxdev_stream_sync(o->core->stream);
        free(fold);
        free(ff);
    }
    if (st != AFX_OK) {
        spectrogramObj_free(o);
        return st;
    }
    *spectrogramObj = o;
    return 0;
}

void spectrogramObj_setDeepOrder(SpectrogramObj o, int deepOrder) {
    if (o && deepOrder >= 1 && deepOrder <= 4) o->deepOrder = deepOrder;
}

void spectrogramObj_setChromaDataNormalType(SpectrogramObj o, ChromaDataNormalType dataNormType) {
    if (o) o->dataNormType = dataNormType;
}

void spectrogramObj_setDataNormValue(SpectrogramObj o, float normValue) {
    if (o && normValue > 0) o->normValue = normValue;
}

int spectrogramObj_calTimeLength(SpectrogramObj o, int dataLength) {
    return o ? stftObj_calTimeLength(&o->tail, dataLength) : 0;
}

void spectrogramObj_enableDebug(SpectrogramObj o, int flag) {
    (void)flag;
    if (o) o->isDebug = 1; /* the reference ignores the flag too (:3171-3174); no dumps here */
}

float *spectrogramObj_getFreBandArr(SpectrogramObj o) { return o ? o->freBandArr : NULL; }
int *spectrogramObj_getBinBandArr(SpectrogramObj o) { return o ? o->binBandArr : NULL; }
int spectrogramObj_getBandNum(SpectrogramObj o) { return o ? o->num : 0; }
int spectrogramObj_getBinBandLength(SpectrogramObj o) { return o ? o->num : 0; }

static void fail(SpectrogramObj o, int st, const char *who) {
    o->status = st;
    afxdev_report_failure(who, st);
}

/* the norm exponent the core applies: for a magnitude chroma the reference raises the FOLDED
 * result (:1146-1152, :1196-1202), which k_row_post does; everything else is the BFT rule */
static void sync_core_switches(SpectrogramObj o) {
    o->core->normValue = (is_chroma_like(o->scale) && o->dataType == SpectralData_Mag) ? 1.f : o->normValue;
}

/* chroma tail on `rows` device rows: [fold GEMM] -> [pow] -> per-frame normalisation */
static int chroma_tail(SpectrogramObj o, const float *dBase, long long rows, float *dOut, void *stream) {
    int st = AFX_OK;
    if (o->scale == SpectralFilterBankScale_LogChroma)
        st = afxk_gemm_nt(dBase, o->baseNum, o->dFold, o->baseNum, dOut, o->num, rows, o->num, o->baseNum,
                          AFX_MAP_NONE, AFX_MAP_NONE, 1.f, stream);
    if (st != AFX_OK) return st;
    const int doPow = (o->dataType == SpectralData_Mag && o->normValue != 1);
    return afxk_row_post(dOut, rows, o->num, doPow, o->normValue, (int)o->dataNormType, stream);
}

int spectrogramObj_spectrogramBatchDevice(SpectrogramObj o, const float *dData, int batch, int dataLength,
                                          long long clipStride, float *dSpect, void *hipStream) {
    AFX_ENTER(o ? o->core : NULL);
    if (!o || !dData || !dSpect || batch <= 0 || dataLength <= 0) return AFX_ERR_ARG;
    const int T = bftObj_calTimeLength(o->core, dataLength);
    if (T <= 0) return AFX_OK;
    const long long rows = (long long)batch * T;
    sync_core_switches(o);
    float *dBase = dSpect;
    int st = AFX_OK;
    if (o->scale == SpectralFilterBankScale_LogChroma) {
        st = afxdev_reserve((void **)&o->dTmp, &o->capTmp, sizeof(float) * (size_t)rows * o->baseNum);
        dBase = o->dTmp;
    }
    if (st == AFX_OK)
        st = afx_bft_run_device(o->core, dData, batch, dataLength, clipStride, dBase, NULL, NULL, hipStream);
    if (st == AFX_OK && is_chroma_like(o->scale)) st = chroma_tail(o, dBase, rows, dSpect, hipStream);
    return st;
}

#ifdef AFX_EXPERIMENTS /* measurement builds only: where a one-clip call spends its time (printed every 200 calls) */
#include <stdio.h>
#include <time.h>
static double g_phase[5], g_phaseLast;
static int g_phaseCalls;
static double phase_now(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return 1e6 * (double)ts.tv_sec + 1e-3 * (double)ts.tv_nsec;
}
#define AFX_PHASE_T(i)                                                                                          \
    do {                                                                                                        \
        const double now_ = phase_now();                                                                        \
        if ((i) > 0) g_phase[i] += now_ - g_phaseLast;                                                          \
        g_phaseLast = now_;                                                                                     \
        if ((i) == 4 && ++g_phaseCalls % 200 == 0) {                                                            \
            fprintf(stderr, "[afx phases, us per call] h2d %.1f launch %.1f d2h %.1f sync %.1f\n", g_phase[1] / 200, \
                    g_phase[2] / 200, g_phase[3] / 200, g_phase[4] / 200);                                      \
            g_phase[1] = g_phase[2] = g_phase[3] = g_phase[4] = 0;                                              \
        }                                                                                                       \
    } while (0)
#else
#define AFX_PHASE_T(i) ((void)0)
#endif

void spectrogramObj_spectrogram(SpectrogramObj o, float *dataArr, int dataLength, float *mSpectArr,
                                float *mPhaseArr) {
    AFX_ENTER(o ? o->core : NULL);
    if (!o) {
        afxdev_set_error("spectrogramObj_spectrogram: NULL object");
        return;
    }
    if (!dataArr || dataLength <= 0) return;
    int valid = dataLength, headTail = 0, skip = 0, T;
    o->tail.isContinue = o->isContinue;
    if (o->isContinue) {
        T = afx_stft_deal_data(&o->tail, dataArr, dataLength, &valid, &headTail, &skip);
    } else {
        T = stftObj_calTimeLength(&o->tail, dataLength);
    }
    if (T <= 0) return; /* the reference keeps its previous frame count here as well */
    o->timeLength = T;
    if (!mSpectArr) return;
    const int upData = dataLength - skip, total = headTail + upData;
    void *stream = o->core->stream;
    const size_t outB = sizeof(float) * (size_t)T * o->num;
    AFX_PHASE_T(0);
    int st = afxdev_reserve((void **)&o->dX, &o->capX, sizeof(float) * (size_t)total);
    if (st == AFX_OK) st = afxdev_reserve((void **)&o->dOut, &o->capOut, 2 * outB);
    if (st == AFX_OK && headTail > 0)
        st = afxdev_h2d(o->dX, o->tail.tailDataArr, sizeof(float) * (size_t)headTail, stream);
    if (st == AFX_OK) st = afxdev_h2d(o->dX + headTail, dataArr + skip, sizeof(float) * (size_t)upData, stream);
    AFX_PHASE_T(1);
    if (st == AFX_OK) st = spectrogramObj_spectrogramBatchDevice(o, o->dX, 1, total, total, o->dOut, stream);
    AFX_PHASE_T(2);
    if (st == AFX_OK) st = afxdev_d2h(mSpectArr, o->dOut, outB, stream);
    AFX_PHASE_T(3);
    if (st == AFX_OK && mPhaseArr && o->scale == SpectralFilterBankScale_Linear) {
        /* phase of the sliced bins, real part clamped at 1e-16 (:1037-1053) */
        AfxStftArgs a;
        memset(&a, 0, sizeof(a));
        a.x = o->dX;
        a.clipStride = total;
        a.batch = 1;
        a.dataLength = total;
        a.timeLength = T;
        a.radix2Exp = o->radix2Exp;
        a.hop = o->slideLength;
        a.window = o->core->dWindow;
        a.twiddle = o->core->dTwiddle;
        a.mode = AFX_SPEC_PHASE;
        a.binLo = o->lowIndex;
        a.binCount = o->num;
        a.outRe = o->dOut + (size_t)T * o->num;
        st = afxk_stft(&a, stream);
        if (st == AFX_OK) st = afxdev_d2h(mPhaseArr, a.outRe, outB, stream);
    }
    if (st == AFX_OK) st = afxdev_stream_sync(stream);
    AFX_PHASE_T(4);
    if (o->isContinue) afx_stft_keep_tail(&o->tail, dataArr + skip, upData, total);
    if (st != AFX_OK) fail(o, st, "spectrogramObj_spectrogram");
}

void spectrogramObj_spectrogram1(SpectrogramObj o, float *mRealArr, float *mImageArr, int nLength,
                                 int mLength, float *mSpectArr, float *mPhaseArr) {
    AFX_ENTER(o ? o->core : NULL);
    if (!o) {
        afxdev_set_error("spectrogramObj_spectrogram1: NULL object");
        return;
    }
    if (!mRealArr || !mImageArr || nLength <= 0 || mLength != o->fftLength) return; /* :951-960 */
    const int T = nLength, N = o->fftLength, F = N / 2 + 1;
    o->timeLength = T;
    if (!mSpectArr) return;
    void *stream = o->core->stream;
    const size_t specB = sizeof(float) * (size_t)T * N, outB = sizeof(float) * (size_t)T * o->num;
    const int linear = (o->scale == SpectralFilterBankScale_Linear);
    sync_core_switches(o);
    /* per-bin value and what follows the bank: the rules of bftObj_bft's real result mode */
    int mode = AFX_SPEC_POWER, post = AFX_MAP_NONE;
    if (o->dataType == SpectralData_Mag) {
        mode = AFX_SPEC_MAG;
        if (o->core->normValue != 1) {
            if (linear) mode = AFX_SPEC_MAG_NORM;
            else post = AFX_MAP_POW;
        }
    } else if (o->dataType == SpectralData_Power && o->core->normValue != 1) {
        mode = AFX_SPEC_POWER_NORM;
    }
    int st = afxdev_reserve((void **)&o->dSpec, &o->capSpec, 2 * specB + sizeof(float) * (size_t)T * F);
    if (st == AFX_OK) st = afxdev_reserve((void **)&o->dOut, &o->capOut, 2 * outB);
    if (st == AFX_OK && o->scale == SpectralFilterBankScale_LogChroma)
        st = afxdev_reserve((void **)&o->dTmp, &o->capTmp, sizeof(float) * (size_t)T * o->baseNum);
    float *dRe = o->dSpec, *dIm = o->dSpec + (size_t)T * N, *dS = o->dSpec + 2 * (size_t)T * N;
    if (st == AFX_OK) st = afxdev_h2d(dRe, mRealArr, specB, stream);
    if (st == AFX_OK) st = afxdev_h2d(dIm, mImageArr, specB, stream);
    if (st == AFX_OK && linear) {
        st = afxk_spec_map(dRe, dIm, T, N, o->lowIndex, o->num, mode, o->core->normValue, o->dOut, NULL, stream);
        if (st == AFX_OK && mPhaseArr) {
            float *dPh = o->dOut + (size_t)T * o->num;
            st = afxk_spec_map(dRe, dIm, T, N, o->lowIndex, o->num, AFX_SPEC_PHASE, 1.f, dPh, NULL, stream);
            if (st == AFX_OK) st = afxdev_d2h(mPhaseArr, dPh, outB, stream);
        }
    } else if (st == AFX_OK) {
        float *dBase = (o->scale == SpectralFilterBankScale_LogChroma) ? o->dTmp : o->dOut;
        st = afxk_spec_map(dRe, dIm, T, N, 0, F, mode, o->core->normValue, dS, NULL, stream);
        if (st == AFX_OK)
            st = afxk_gemm_nt(dS, F, o->core->dBank, o->core->bankPitch, dBase, o->coreRows, T, o->coreRows, F, AFX_MAP_NONE,
                              post, o->core->normValue, stream);
        if (st == AFX_OK && is_chroma_like(o->scale)) st = chroma_tail(o, dBase, T, o->dOut, stream);
    }
    if (st == AFX_OK) st = afxdev_d2h(mSpectArr, o->dOut, outB, stream);
    if (st == AFX_OK) st = afxdev_stream_sync(stream);
    if (st != AFX_OK) fail(o, st, "spectrogramObj_spectrogram1");
}

/* log10 / cube-root rectification + DCT-II, first ccNum coefficients (:1409-1475) */
static void run_xxcc(SpectrogramObj o, float *mDataArr1, int ccNum, CepstralRectifyType *rectifyType,
                     float *mDataArr2, const char *who) {
    if (!o || !mDataArr1 || !mDataArr2) return;
    if (ccNum > o->num || ccNum < 1 || o->timeLength <= 0) return;
    if (!o->xxcc) {
        int st = xxccObj_new(&o->xxcc, o->num);
        if (st != 0) {
            fail(o, st, who);
            return;
        }
    }
    xxccObj_setTimeLength(o->xxcc, o->timeLength);
    xxccObj_xxcc(o->xxcc, mDataArr1, ccNum, rectifyType, mDataArr2);
}

void spectrogramObj_mfcc(SpectrogramObj o, float *mDataArr1, int ccNum, float *mDataArr2) {
    AFX_ENTER(o ? o->core : NULL);
    if (o && o->scale == SpectralFilterBankScale_Mel) run_xxcc(o, mDataArr1, ccNum, NULL, mDataArr2, "spectrogramObj_mfcc");
}

void spectrogramObj_gtcc(SpectrogramObj o, float *mDataArr1, int ccNum, float *mDataArr2) {
    AFX_ENTER(o ? o->core : NULL);
    if (o && o->style == SpectralFilterBankStyle_Gammatone) run_xxcc(o, mDataArr1, ccNum, NULL, mDataArr2, "spectrogramObj_gtcc");
}

void spectrogramObj_bfcc(SpectrogramObj o, float *mDataArr1, int ccNum, float *mDataArr2) {
    AFX_ENTER(o ? o->core : NULL);
    if (o && o->scale == SpectralFilterBankScale_Bark) run_xxcc(o, mDataArr1, ccNum, NULL, mDataArr2, "spectrogramObj_bfcc");
}

void spectrogramObj_xxcc(SpectrogramObj o, float *mDataArr1, int ccNum, CepstralRectifyType *rectifyType,
                         float *mDataArr2) {
    AFX_ENTER(o ? o->core : NULL);
    run_xxcc(o, mDataArr1, ccNum, rectifyType, mDataArr2, "spectrogramObj_xxcc");
}

void spectrogramObj_mfccStandard(SpectrogramObj o, float *mDataArr1, int *deltaWindowLength,
                                 CepstralEnergyType *energyType, CepstralRectifyType *rectifyType,
                                 float *mDataArr2) {
    AFX_ENTER(o ? o->core : NULL);
    (void)o; (void)mDataArr1; (void)deltaWindowLength; (void)energyType; (void)rectifyType; (void)mDataArr2;
}

void spectrogramObj_xxccStandard(SpectrogramObj o, float *mDataArr1, int *deltaWindowLength,
                                 CepstralEnergyType *energyType, CepstralRectifyType *rectifyType,
                                 float *mDataArr2) {
    AFX_ENTER(o ? o->core : NULL);
    (void)o; (void)mDataArr1; (void)deltaWindowLength; (void)energyType; (void)rectifyType; (void)mDataArr2;
}

void spectrogramObj_deconv(SpectrogramObj o, float *mDataArr1, float *mDataArr2, float *mDataArr3) {
    AFX_ENTER(o ? o->core : NULL);
    if (!o) {
        afxdev_set_error("spectrogramObj_deconv: NULL object");
        return;
    }
    const int T = o->timeLength;
    if (T <= 0 || !mDataArr1 || !mDataArr2 || !mDataArr3) return;
    void *stream = o->core->stream;
    int st = AFX_OK;
    if (!o->dDevTw) { /* __spectrogramObj_dealDeconv (:1614-1672): transform length ceilPow2(2 num) */
        const int M = afx_ceil_pow2(2 * o->num);
        o->devRadix = afx_log2_exact(M);
        float *tw = afx_twiddle_table(M);
        if (!tw) st = AFX_ERR_NOMEM;
        if (st == AFX_OK) st = afxdev_malloc((void **)&o->dDevTw, sizeof(float) * (size_t)(M < 2 ? 2 : M));
        if (st == AFX_OK) st = afxdev_h2d(o->dDevTw, tw, sizeof(float) * (size_t)(M < 2 ? 2 : M), stream);
        if (st == AFX_OK) st = afxdev_stream_sync(stream);
        free(tw);
    }
    const size_t inB = sizeof(float) * (size_t)T * o->num;
    if (st == AFX_OK) st = afxdev_reserve((void **)&o->dTmp, &o->capTmp, 3 * inB);
    float *dA = o->dTmp, *dT = o->dTmp + (size_t)T * o->num, *dP = o->dTmp + 2 * (size_t)T * o->num;
    if (st == AFX_OK) st = afxdev_h2d(dA, mDataArr1, inB, stream);
    if (st == AFX_OK) st = afxk_cqt_deconv(dA, T, o->num, o->devRadix, o->dDevTw, NULL, 0, dT, dP, NULL, stream);
    if (st == AFX_OK) st = afxdev_d2h(mDataArr2, dT, inB, stream);
    if (st == AFX_OK) st = afxdev_d2h(mDataArr3, dP, inB, stream);
    if (st == AFX_OK) st = afxdev_stream_sync(stream);
    if (st != AFX_OK) fail(o, st, "spectrogramObj_deconv");
}

void spectrogramObj_free(SpectrogramObj o) {
    if (!o) return;
    if (o->core && o->core->stream) afxdev_stream_sync(o->core->stream);
    afxdev_free(o->dFold);
    afxdev_free(o->dDevTw);
    afxdev_free(o->dX);
    afxdev_free(o->dOut);
    afxdev_free(o->dTmp);
    afxdev_free(o->dSpec);
    xxccObj_free(o->xxcc);
    bftObj_free(o->core);
    free(o->tail.tailDataArr);
    free(o->freBandArr);
    free(o->binBandArr);
    free(o);
}
