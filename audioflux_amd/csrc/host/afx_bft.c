/* afx_bft.c -- the BFT object (C host side) behind include/bft_algorithm.h.
 *
 * Mirrors the parameter semantics of the reference object
 * (src/bft_algorithm.c:87-626): defaults, range checks, status codes, the
 * low/high frequency revision, getters and result-type switches.  What it does
 * NOT mirror is the execution: instead of STFT -> [T,N] complex scratch ->
 * crop -> square -> serial matmul on the CPU, a call uploads the clip(s),
 * launches the framed-FFT kernel (spectrum values computed in its epilogue)
 * and the MFMA filter-bank GEMM on this object's HIP stream, and downloads the
 * [T,num] result.  There is no CPU compute path.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "afx_batch.h"
#include "afx_device.h"
#include "afx_host.h"
#include "afx_objects.h"
#include "bft_algorithm.h"

/* largest [frames, F] spectrum scratch one launch may use; larger batches are
 * processed in clip chunks (the fused kernel needs no such scratch) */
/* spectrum scratch of the dense-bank route per chunk.  Measured on cfg 2 (tools/bench_dense.py,
 * profiles/r02_dense_gemm.txt): 96 MB chunks (spectrum resident in the Infinity Cache between the
 * two kernels) give grids of < 1 workgroup per CU and 115 M frames/s; 1 GB chunks 171 M frames/s
 * with the GEMM at 107 TFLOP/s -- launch width beats cache residency here */
static size_t dense_chunk_bytes(void) {
    const char *s = getenv("AFX_SCRATCH_MB");
    size_t mb = 1024;
    if (s && atoi(s) > 0) mb = (size_t)atoi(s);
    return mb << 20;
}

/* (narrow banks too: STFT-chroma-12 through the 128-column tile, nine tenths of it padding, 0.581 ms per 187 k frames against 0.767 on
 * the small-tile float32 product, profiles/r06_dense.txt)
 * rows[frames, F] (pitched) . bank^T -> out[frames, num]: the prepared-bank kernel (the bank split into its bf16 word planes
 * once per object, afx_gemm_bf16.hip), else the generic product on the float bank (__mdot1, flux_vector.c:55-86) */
static int dense_product(BFTObj o, const float *rows, int pitch, float *out, long long frames, int post, float postArg,
                         void *stream) {
    if (!o->bankImageTried) {
        o->bankImageTried = 1;
        o->dBankImage = NULL;
        int st = afxk_gemm_bank_prepare(o->dBank, o->bankPitch, o->num, o->F, &o->dBankImage, stream);
        if (st == AFX_OK) st = afxdev_stream_sync(stream); /* once per object: later calls may come on another stream */
        if (st != AFX_OK) {
            afxdev_free(o->dBankImage);
            o->dBankImage = NULL;
        }
    }
    if (o->dBankImage) {
        const int st = afxk_gemm_nt_bank(rows, pitch, o->dBankImage, o->num, o->F, out, o->num, frames, post, postArg, stream);
        if (st != AFX_ERR_UNSUPPORTED) return st;
    }
    return afxk_gemm_nt(rows, pitch, o->dBank, o->bankPitch, out, o->num, frames, o->num, o->F, AFX_MAP_NONE, post, postArg,
                        stream);
}

int bftObj_new(BFTObj *bftObj, int num, int radix2Exp, int *samplate, float *lowFre,
               float *highFre, int *binPerOctave, WindowType *windowType, int *slideLength,
               SpectralFilterBankScaleType *filterScaleType,
               SpectralFilterBankStyleType *filterStyleType,
               SpectralFilterBankNormalType *filterNormalType, SpectralDataType *dataType,
               int *isReassign, int *isTemporal) {
    int r = 12, sr = 32000, bpo = 12, hop, fftLength;
    float low = 0, high = 0;
    int lowIndex = 0, highIndex = 0;
    WindowType win = Window_Hann;
    SpectralDataType dtype = SpectralData_Power;
    SpectralFilterBankScaleType scale = SpectralFilterBankScale_Linear;
    SpectralFilterBankStyleType style = SpectralFilterBankStyle_Slaney;
    SpectralFilterBankNormalType normal = SpectralFilterBankNormal_None;

    if (!bftObj) return -1;
    *bftObj = NULL;

    /* --- validation & defaults, in the reference's order (bft_algorithm.c:122-243) */
    if (radix2Exp) {
        r = radix2Exp;
        if (r < 1 || r > 30) {
            printf("radix2Exp is error!\n");
            return -100;
        }
    }
    fftLength = 1 << r;
    if (samplate && *samplate > 0 && *samplate <= 196000) sr = *samplate;
    if (dataType) dtype = *dataType;
    if (filterScaleType) {
        scale = *filterScaleType;
        if ((int)scale > (int)SpectralFilterBankScale_Log) {
            printf("scaleType is error!\n");
            return 1;
        }
    }
    if (filterStyleType) style = *filterStyleType;
    if (filterNormalType) normal = *filterNormalType;

    high = (float)(sr / 2.0);
    if (lowFre && *lowFre >= 0 && *lowFre < sr / 2.0) low = *lowFre;
    const int logLike =
        (scale == SpectralFilterBankScale_Octave || scale == SpectralFilterBankScale_Log);
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
    if (windowType) win = *windowType;
    hop = fftLength / 4;
    if (hop < 1) hop = 1; /* fftLength 2: the reference's default of 0 divides by zero in its frame count */
    if (slideLength && *slideLength > 0) hop = *slideLength;

    if (scale == SpectralFilterBankScale_Linear) {
        float det = sr / (float)fftLength;
        afx_auditory_revise_linear(num, low, high, det, 1, &low, &high);
        lowIndex = (int)roundf(low / det);
        highIndex = (int)roundf(high / det);
        if (high > sr / 2.0) {
            printf("scale linear: lowFre and num is large, overflow error\n");
            return -1;
        }
    } else if (scale == SpectralFilterBankScale_Octave) {
        afx_auditory_revise_log(num, low, high, bpo, 1, &low, &high);
        if (high > sr / 2.0) {
            printf("scale log: lowFre and num is large, overflow error!\n");
            return -1;
        }
    }
    if (num < 2 || num > fftLength / 2 + 1) {
        printf("num is error!\n");
        return -1;
    }
    if (r > 14) {
        afxdev_set_error("bftObj_new: fftLength 2^%d exceeds the on-chip FFT limit 2^14", r);
        return AFX_ERR_UNSUPPORTED;
    }

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
    p.isTemporal = isTemporal ? *isTemporal : 0;
    p.isReassign = (isReassign && *isReassign) ? 1 : 0;
    return afx_bft_create(&p, bftObj);
}

/* builds the object for already-validated parameters: host plan (window, band arrays, bank),
 * device constants, banded view of the bank, fused-kernel plan.  Shared by bftObj_new and
 * spectrogramObj_new (their defaults and checks differ, the execution plan does not). */
int afx_bft_create(const AfxBftPlan *p, BFTObj *bftObj) {
    const int num = p->num, r = p->radix2Exp, fftLength = 1 << p->radix2Exp, sr = p->samplate;
    const float low = p->lowFre, high = p->highFre;
    const int lowIndex = p->lowIndex, highIndex = p->highIndex, bpo = p->binPerOctave;
    const WindowType win = p->windowType;
    const SpectralFilterBankScaleType scale = p->scale;
    *bftObj = NULL;
    if (r > 14) {
        afxdev_set_error("fftLength 2^%d exceeds the on-chip FFT limit 2^14", r);
        return AFX_ERR_UNSUPPORTED;
    }
    int st = afxdev_ensure();
    if (st != AFX_OK) return st;

    BFTObj o = (BFTObj)calloc(1, sizeof(struct OpaqueBFT));
    if (!o) return AFX_ERR_NOMEM;
    o->fftLength = fftLength;
    o->radix2Exp = r;
    o->F = fftLength / 2 + 1;
    o->num = num;
    o->samplate = sr;
    o->lowFre = low;
    o->highFre = high;
    o->lowIndex = lowIndex;
    o->highIndex = highIndex;
    o->binPerOctave = bpo;
    o->windowType = win;
    o->slideLength = p->slideLength;
    o->dataType = p->dataType;
    o->scale = scale;
    o->style = p->style;
    o->normal = p->normal;
    o->normValue = 1;
    o->isTemporal = p->isTemporal;

    /* --- host-side plan: window, band arrays, bank (bft_algorithm.c:278-389) */
    float *hWindow = afx_window_fft(win, fftLength);
    float *hTw = afx_twiddle_table(fftLength);
    float *hBank = NULL;
    o->freBandArr = (float *)calloc((size_t)num + 2, sizeof(float));
    o->binBandArr = (int *)calloc((size_t)num + 2, sizeof(int));
    if (!hWindow || !hTw || !o->freBandArr || !o->binBandArr) st = AFX_ERR_NOMEM;

    if (st == AFX_OK) {
        if (scale == SpectralFilterBankScale_Linear) {
            float det = sr / (float)fftLength;
            for (int i = lowIndex, j = 0; i <= highIndex && j < num + 2; i++, j++) {
                o->freBandArr[j] = i * det;
                o->binBandArr[j] = i;
            }
        } else {
            hBank = (float *)calloc((size_t)num * o->F, sizeof(float));
            if (!hBank) {
                st = AFX_ERR_NOMEM;
            } else if (p->customBank) {
                /* caller-built [num, F] matrix (the STFT-chroma bank of the spectrogram object) */
                memcpy(hBank, p->customBank, sizeof(float) * (size_t)num * o->F);
            } else {
                st = afx_auditory_bank(num, fftLength, sr, scale, p->style, p->normal, low, high, bpo, hBank,
                                       o->freBandArr, o->binBandArr);
            }
        }
    }

    /* --- device constants */
    if (st == AFX_OK) st = afxdev_stream_create(&o->stream);
    if (st == AFX_OK) st = afxdev_malloc((void **)&o->dWindow, sizeof(float) * fftLength);
    if (st == AFX_OK) st = afxdev_malloc((void **)&o->dTwiddle, sizeof(float) * fftLength);
    if (st == AFX_OK) st = afxdev_h2d(o->dWindow, hWindow, sizeof(float) * fftLength, o->stream);
    if (st == AFX_OK)
        st = afxdev_h2d(o->dTwiddle, hTw, sizeof(float) * (fftLength / 2 > 0 ? fftLength : 2),
                        o->stream);
    o->bankPitch = (o->F + 3) & ~3;
    if (st == AFX_OK && hBank) {
        float *padded = (float *)calloc((size_t)num * o->bankPitch, sizeof(float));
        if (!padded) st = AFX_ERR_NOMEM;
        if (st == AFX_OK) {
            for (int r = 0; r < num; r++)
                memcpy(padded + (size_t)r * o->bankPitch, hBank + (size_t)r * o->F, sizeof(float) * o->F);
            st = afxdev_malloc((void **)&o->dBank, sizeof(float) * (size_t)num * o->bankPitch);
        }
        if (st == AFX_OK)
            st = afxdev_h2d(o->dBank, padded, sizeof(float) * (size_t)num * o->bankPitch, o->stream);
        if (st == AFX_OK) st = afxdev_stream_sync(o->stream); /* the staging copy is freed below */
        free(padded);
    }
    if (st == AFX_OK && hBank) {
        /* row spans of the bank: [first non-zero, last non-zero] of every row.  Used when the
         * spans hold at most 1/4 of the dense matrix (triangular / window banks: ~1-3 %) */
        int *meta = (int *)calloc(3 * (size_t)num, sizeof(int));
        size_t total = 0;
        if (!meta) st = AFX_ERR_NOMEM;
        for (int i = 0; i < num && st == AFX_OK; i++) {
            const float *row = hBank + (size_t)i * o->F;
            int first = -1, last = -1;
            for (int k = 0; k < o->F; k++)
                if (row[k] != 0.f) {
                    if (first < 0) first = k;
                    last = k;
                }
            meta[i] = first < 0 ? 0 : first;
            meta[num + i] = first < 0 ? 0 : last - first + 1;
            meta[2 * num + i] = (int)total;
            total += (size_t)meta[num + i];
        }
        if (st == AFX_OK && total * 4 <= (size_t)num * o->F) {
            /* tap-major weights [maxLen][num]: the kernel's lanes are bank rows */
            int maxLen = 1;
            for (int i = 0; i < num; i++)
                if (meta[num + i] > maxLen) maxLen = meta[num + i];
            total = (size_t)maxLen * num;
            float *w = (float *)calloc(total, sizeof(float));
            if (!w) st = AFX_ERR_NOMEM;
            for (int i = 0; i < num && st == AFX_OK; i++)
                for (int q = 0; q < meta[num + i]; q++) w[(size_t)q * num + i] = hBank[(size_t)i * o->F + meta[i] + q];
            if (st == AFX_OK) st = afxdev_malloc((void **)&o->dBandMeta, sizeof(int) * 3 * (size_t)num);
            if (st == AFX_OK) st = afxdev_h2d(o->dBandMeta, meta, sizeof(int) * 3 * (size_t)num, o->stream);
            if (st == AFX_OK) st = afxdev_malloc((void **)&o->dBandW, sizeof(float) * (total ? total : 1));
            if (st == AFX_OK) st = afxdev_h2d(o->dBandW, w, sizeof(float) * (total ? total : 1), o->stream);
            if (st == AFX_OK) st = afxdev_stream_sync(o->stream);
            free(w);
        }
        free(meta);
    }
    if (st == AFX_OK && p->isReassign && r < 2) {
        /* reassignObj_new takes radix2Exp 1 for "unset" and builds a 2^12 transform (reassign_algorithm.c:125-127),
         * while this object's arrays are sized for fftLength 2: the reference writes past them.  Refused. */
        afxdev_set_error("bftObj_new: isReassign with radix2Exp 1 is undefined in the reference (its reassignment "
                         "object falls back to 2^12); refused");
        st = AFX_ERR_UNSUPPORTED;
    }
    if (st == AFX_OK && p->isReassign) {
        /* __bftObj_init (bft_algorithm.c:332-340): reassignment of both axes, default threshold */
        ReassignType reType = Reassign_All;
        int srv = sr, hopv = p->slideLength;
        WindowType wv = win;
        st = reassignObj_new(&o->reassign, r, &srv, &wv, &hopv, &reType, NULL, NULL, NULL);
    }
    if (st == AFX_OK && !p->isReassign) st = afx_bft_plan_fast(o, hWindow, hBank);
    if (st == AFX_OK) st = afxdev_stream_sync(o->stream);
    free(hWindow);
    free(hTw);
    free(hBank);
    if (st != AFX_OK) {
        bftObj_free(o);
        return st;
    }
    *bftObj = o;
    return 0;
}

int bftObj_calTimeLength(BFTObj o, int dataLength) {
    /* stftObj_calTimeLength without padding (src/stft_algorithm.c:225-262) */
    if (!o || dataLength < o->fftLength) return 0;
    return (dataLength - o->fftLength) / o->slideLength + 1;
}

float *bftObj_getFreBandArr(BFTObj o) { return o ? o->freBandArr : NULL; }
int *bftObj_getBinBandArr(BFTObj o) { return o ? o->binBandArr : NULL; }

void bftObj_setResultType(BFTObj o, int type) {
    if (o) o->resultType = type;
}

void bftObj_setDataNormValue(BFTObj o, float normValue) {
    if (o && normValue > 0) o->normValue = normValue;
}

/* spectrum mode + post-op for the current switches (bft_algorithm.c:457-529) */
static void pick_modes(const struct OpaqueBFT *o, int *specMode, int *post) {
    *post = AFX_MAP_NONE;
    const int linear = (o->scale == SpectralFilterBankScale_Linear);
    if (!o->resultType) {
        *specMode = (o->dataType == SpectralData_Power) ? AFX_SPEC_SQUARE : AFX_SPEC_COMPLEX;
    } else if (o->dataType == SpectralData_Mag) {
        *specMode = AFX_SPEC_MAG;
        if (o->normValue != 1) {
            if (linear) {
                *specMode = AFX_SPEC_MAG_NORM;
            } else {
                *post = AFX_MAP_POW;
            }
        }
    } else {
        /* any other dataType value falls through the reference's if/else-if as
         * plain |S|^2 unless it equals Power with a norm exponent */
        *specMode = AFX_SPEC_POWER;
        if (o->dataType == SpectralData_Power && o->normValue != 1) *specMode = AFX_SPEC_POWER_NORM;
    }
}

/* isReassign = 1 (bft_algorithm.c:451-529): the reassigned complex spectrum [frames, F] takes the
 * place of the STFT; per-bin value, bank and norm exponent follow the same rules as below.
 * The reference accumulates into a scratch it zeroes only when (re)allocated, so a second call on
 * the same object adds onto the previous call's post-processed content; here every call starts
 * from zero. */
static int run_reassigned(BFTObj o, const float *dData, int batch, int dataLength, long long clipStride,
                          float *dRe, float *dIm, float *dTemporal, void *stream) {
    const int T = bftObj_calTimeLength(o, dataLength), F = o->F;
    const long long frames = (long long)batch * T;
    int specMode, post;
    pick_modes(o, &specMode, &post);
    const int complexOut = !o->resultType;
    const int linear = (o->scale == SpectralFilterBankScale_Linear);
    const size_t plane = (size_t)frames * F;
    int st = afxdev_reserve((void **)&o->dSpec, &o->capSpec, sizeof(float) * plane * 4);
    if (st != AFX_OK) return st;
    float *rRe = o->dSpec, *rIm = o->dSpec + plane, *mRe = o->dSpec + 2 * plane, *mIm = o->dSpec + 3 * plane;
    st = afxdev_memset(rRe, 0, sizeof(float) * plane * 2, stream);
    if (st == AFX_OK)
        st = reassignObj_reassignBatchDevice(o->reassign, dData, batch, dataLength, clipStride, rRe, rIm, NULL,
                                             NULL, stream);
    if (st != AFX_OK) return st;
    if (dTemporal) {
        /* energy / rms / zcr come from the windowed frames, not from the spectrum: one STFT pass
         * whose bins are discarded (binCount 1) */
        AfxStftArgs a;
        memset(&a, 0, sizeof(a));
        a.x = dData;
        a.clipStride = clipStride;
        a.batch = batch;
        a.dataLength = dataLength;
        a.timeLength = T;
        a.radix2Exp = o->radix2Exp;
        a.hop = o->slideLength;
        a.window = o->dWindow;
        a.twiddle = o->dTwiddle;
        a.mode = AFX_SPEC_POWER;
        a.binCount = 1;
        a.outRe = mRe;
        a.energy = dTemporal;
        a.rms = dTemporal + frames;
        a.zcr = dTemporal + 2 * frames;
        st = afxk_stft(&a, stream);
        if (st != AFX_OK) return st;
    }
    if (linear) {
        if (o->highIndex - o->lowIndex + 1 != o->num) {
            afxdev_set_error("bft linear: %d bins for num=%d", o->highIndex - o->lowIndex + 1, o->num);
            return AFX_ERR_ARG;
        }
        return afxk_spec_map(rRe, rIm, frames, F, o->lowIndex, o->num, specMode, o->normValue, dRe,
                             complexOut ? dIm : NULL, stream);
    }
    st = afxk_spec_map(rRe, rIm, frames, F, 0, F, specMode, o->normValue, mRe, complexOut ? mIm : NULL, stream);
    if (st == AFX_OK)
        st = afxk_gemm_nt(mRe, F, o->dBank, o->bankPitch, dRe, o->num, frames, o->num, F, AFX_MAP_NONE, post,
                          o->normValue, stream);
    if (st == AFX_OK && complexOut)
        st = afxk_gemm_nt(mIm, F, o->dBank, o->bankPitch, dIm, o->num, frames, o->num, F, AFX_MAP_NONE, AFX_MAP_NONE,
                          1.f, stream);
    return st;
}

/* device-resident core: dData -> dRe (, dIm); dTemporal = 3 planes of frames or NULL */
int afx_bft_run_device(BFTObj o, const float *dData, int batch, int dataLength,
                       long long clipStride, float *dRe, float *dIm, float *dTemporal,
                       void *stream) {
    const int T = bftObj_calTimeLength(o, dataLength);
    if (T <= 0 || batch <= 0) return AFX_OK;
    /* scratch buffers belong to the object: drain the previous stream when the
     * caller switches streams between calls */
    if (o->lastStreamSet && o->lastStream != stream) {
        int sst = afxdev_stream_sync(o->lastStream);
        if (sst != AFX_OK) return sst;
    }
    o->lastStream = stream;
    o->lastStreamSet = 1;
    int specMode, post;
    pick_modes(o, &specMode, &post);
    const int complexOut = !o->resultType;
    const int linear = (o->scale == SpectralFilterBankScale_Linear);
    const long long framesAll = (long long)batch * T;

    AfxStftArgs a;
    memset(&a, 0, sizeof(a));
    a.clipStride = clipStride;
    a.dataLength = dataLength;
    a.timeLength = T;
    a.radix2Exp = o->radix2Exp;
    a.hop = o->slideLength;
    a.window = o->dWindow;
    a.twiddle = o->dTwiddle;
    a.mode = specMode;
    a.normValue = o->normValue;

    if (o->reassign) return run_reassigned(o, dData, batch, dataLength, clipStride, dRe, dIm, dTemporal, stream);

    if (linear) {
        /* the "bank" is a bin slice: store straight into the result */
        int count = o->highIndex - o->lowIndex + 1;
        if (count > o->num) count = o->num;
        if (count != o->num) {
            afxdev_set_error("bft linear: %d bins for num=%d", count, o->num);
            return AFX_ERR_ARG;
        }
        a.x = dData;
        a.batch = batch;
        a.binLo = o->lowIndex;
        a.binCount = o->num;
        a.outRe = dRe;
        a.outIm = dIm;
        if (dTemporal) {
            a.energy = dTemporal;
            a.rms = dTemporal + framesAll;
            a.zcr = dTemporal + 2 * framesAll;
        }
        return afxk_stft(&a, stream);
    }

    /* fused register-resident kernel for the hot configurations (temporal features ride along
     * in the n_fft 2048 real-result kernel) */
    {
        int used = 0;
        int st = afx_bft_try_fast(o, dData, batch, dataLength, clipStride, dRe, dIm, dTemporal, stream, &used);
        if (st != AFX_OK || used) return st;
    }

    /* banded bank: STFT and filter bank in one launch of the size-generic kernel */
    if (o->dBandMeta) {
        a.x = dData;
        a.batch = batch;
        a.binLo = 0;
        a.binCount = o->F;
        a.outRe = dRe;
        a.outIm = complexOut ? dIm : NULL;
        a.bandStart = o->dBandMeta;
        a.bandLen = o->dBandMeta + o->num;
        a.bandOff = o->dBandMeta + 2 * o->num;
        a.bandW = o->dBandW;
        a.bandNum = o->num;
        a.bandPost = post;
        a.bandPostArg = o->normValue;
        if (dTemporal) {
            a.energy = dTemporal;
            a.rms = dTemporal + framesAll;
            a.zcr = dTemporal + 2 * framesAll;
        }
        return afxk_stft(&a, stream);
    }

    /* dense bank (gammatone): spectrum scratch + MFMA GEMM, chunked over clips.  The scratch rows
     * are pitched to 4 floats (16-byte aligned rows for the GEMM's dwordx4 loads); chunk size:
     * AFX_SCRATCH_MB, default 1024 (dense_chunk_bytes) */
    const int planes = complexOut ? 2 : 1;
    const int pitch = o->bankPitch;
    const size_t perClip = (size_t)T * pitch * sizeof(float) * planes;
    long long chunk = (long long)(dense_chunk_bytes() / (perClip ? perClip : 1));
    if (chunk < 1) chunk = 1;
    if (chunk > batch) chunk = batch;
    int st = afxdev_reserve((void **)&o->dSpec, &o->capSpec, perClip * (size_t)chunk);
    if (st != AFX_OK) return st;

    for (long long b0 = 0; b0 < batch; b0 += chunk) {
        const int nb = (int)((batch - b0 < chunk) ? batch - b0 : chunk);
        const long long frames = (long long)nb * T;
        a.x = dData + b0 * clipStride;
        a.batch = nb;
        a.binLo = 0;
        a.binCount = o->F;
        a.outPitch = pitch;
        a.outRe = o->dSpec;
        a.outIm = complexOut ? o->dSpec + frames * pitch : NULL;
        if (dTemporal) {
            a.energy = dTemporal + b0 * T;
            a.rms = dTemporal + framesAll + b0 * T;
            a.zcr = dTemporal + 2 * framesAll + b0 * T;
        }
        st = afxk_stft(&a, stream);
        if (st != AFX_OK) return st;
        st = dense_product(o, a.outRe, pitch, dRe + b0 * T * o->num, frames, post, o->normValue, stream);
        if (st != AFX_OK) return st;
        if (complexOut) {
            st = dense_product(o, a.outIm, pitch, dIm + b0 * T * o->num, frames, AFX_MAP_NONE, 1.f, stream);
            if (st != AFX_OK) return st;
        }
    }
    return AFX_OK;
}

int bftObj_bftBatchDevice(BFTObj o, const float *dData, int batch, int dataLength,
                          long long clipStride, float *dReal, float *dImag, void *hipStream) {
    AFX_ENTER(o);
    if (!o || !dData || !dReal) return AFX_ERR_ARG;
    if (!o->resultType && !dImag) return AFX_ERR_ARG;
    return afx_bft_run_device(o, dData, batch, dataLength, clipStride, dReal, dImag, NULL,
                              hipStream);
}

int bftObj_bftBatch(BFTObj o, const float *dataArr, int batch, int dataLength, float *mRealArr3,
                    float *mImageArr3) {
    AFX_ENTER(o);
    if (!o || !dataArr || dataLength <= 0 || batch <= 0 || !mRealArr3) return AFX_ERR_ARG;
    const int T = bftObj_calTimeLength(o, dataLength);
    if (T <= 0) return AFX_OK;
    const int complexOut = !o->resultType;
    if (complexOut && !mImageArr3) return AFX_ERR_ARG;
    const long long frames = (long long)batch * T;
    const size_t inBytes = sizeof(float) * (size_t)batch * dataLength;
    const size_t outBytes = sizeof(float) * (size_t)frames * o->num;

    int st = afxdev_reserve((void **)&o->dX, &o->capX, inBytes);
    if (st == AFX_OK)
        st = afxdev_reserve((void **)&o->dOut, &o->capOut, outBytes * (complexOut ? 2 : 1));
    float *dTemporal = NULL;
    if (st == AFX_OK && o->isTemporal) {
        st = afxdev_reserve((void **)&o->dTemporal, &o->capTemporal,
                            sizeof(float) * 3 * (size_t)frames);
        dTemporal = o->dTemporal;
    }
    if (st == AFX_OK) st = afxdev_h2d(o->dX, dataArr, inBytes, o->stream);
    float *dRe = o->dOut, *dIm = complexOut ? o->dOut + frames * o->num : NULL;
    if (st == AFX_OK)
        st = afx_bft_run_device(o, o->dX, batch, dataLength, dataLength, dRe, dIm, dTemporal,
                                o->stream);
    if (st == AFX_OK) st = afxdev_d2h(mRealArr3, dRe, outBytes, o->stream);
    if (st == AFX_OK && complexOut) st = afxdev_d2h(mImageArr3, dIm, outBytes, o->stream);
    if (st == AFX_OK && o->isTemporal) {
        if (o->hTemporalCap < frames) {
            free(o->hTemporal);
            o->hTemporal = (float *)calloc(3 * (size_t)frames, sizeof(float));
            o->hTemporalCap = o->hTemporal ? (int)frames : 0;
            if (!o->hTemporal) st = AFX_ERR_NOMEM;
        }
        if (st == AFX_OK) {
            /* keep the three planes contiguous at stride `frames` */
            st = afxdev_d2h(o->hTemporal, dTemporal, sizeof(float) * 3 * (size_t)frames, o->stream);
            o->hTemporalFrames = (int)frames;
        }
    }
    if (st == AFX_OK) st = afxdev_stream_sync(o->stream);
    o->lastTimeLength = T;
    return st;
}

void bftObj_bft(BFTObj o, float *dataArr, int dataLength, float *mRealArr3, float *mImageArr3) {
    AFX_ENTER(o);
    if (!o) {
        afxdev_set_error("bftObj_bft: NULL object");
        return;
    }
    if (!dataArr || dataLength <= 0) return; /* stftObj_stft returns silently (stft_algorithm.c:267-269) */
    int st = bftObj_bftBatch(o, dataArr, 1, dataLength, mRealArr3, mImageArr3);
    if (st != AFX_OK) {
        o->status = st;
        afxdev_report_failure("bftObj_bft", st);
    }
}

void bftObj_getTemporalData(BFTObj o, float **eArr, float **rArr, float **zArr) {
    if (!o || !o->isTemporal || !o->hTemporal) return;
    const int n = o->hTemporalFrames;
    if (eArr) *eArr = o->hTemporal;
    if (rArr) *rArr = o->hTemporal + n;
    if (zArr) *zArr = o->hTemporal + 2 * n;
}

void bftObj_free(BFTObj o) {
    if (!o) return;
    if (o->stream) afxdev_stream_sync(o->stream);
    afx_bft_free_fast(o);
    reassignObj_free(o->reassign);
    afxdev_free(o->dWindow);
    afxdev_free(o->dTwiddle);
    afxdev_free(o->dBank);
    afxdev_free(o->dBankImage);
    afxdev_free(o->dBandMeta);
    afxdev_free(o->dBandW);
    afxdev_free(o->dX);
    afxdev_free(o->dSpec);
    afxdev_free(o->dOut);
    afxdev_free(o->dTemporal);
    afxdev_stream_destroy(o->stream);
    free(o->freBandArr);
    free(o->binBandArr);
    free(o->hTemporal);
    free(o);
}
