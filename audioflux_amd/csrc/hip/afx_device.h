/* afx_device.h -- the thin C interface between the C host objects
 * (csrc/host) and the HIP layer (csrc/hip).
 *
 * The host side is plain C99 and never sees a HIP type: device buffers are
 * `void*`/`float*` device addresses, streams are opaque `void*`
 * (a hipStream_t underneath).  Every function returns 0 on success and a
 * negative status on failure; the text of the last failure is available from
 * afxdev_last_error().  There is NO CPU fallback anywhere behind this
 * interface: when no gfx950 device/runtime is usable the calls fail.
 */
#ifndef AFX_DEVICE_H
#define AFX_DEVICE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes returned through the public *_new functions as well */
#define AFX_OK 0
#define AFX_ERR_NODEVICE (-2)  /* HIP runtime / device missing            */
#define AFX_ERR_HIP (-3)       /* a HIP call or kernel launch failed      */
#define AFX_ERR_UNSUPPORTED (-4) /* parameter combination not implemented */
#define AFX_ERR_NOMEM (-5)
#define AFX_ERR_ARG (-6)

/* ---- runtime ---------------------------------------------------------- */
int afxdev_ensure(void);              /* initialise HIP, select device      */
const char *afxdev_last_error(void);
void afxdev_set_error(const char *fmt, ...);
int afxdev_device_count(void);
int afxdev_set_device(int ordinal);   /* process-wide default for new objects */
int afxdev_current_device(void);      /* this thread's HIP device, -1 if none */
int afxdev_bind_stream(void *stream); /* thread's device := the stream's device */
#define AFX_MAX_DEVICES 16           /* power of two; per-device one-time flags */
int afxdev_error_count(void);         /* failures reported on this thread so far */
/* a void entry point ends in failure `st`: one stderr line (who, status, last message) and the thread's
 * failure count advances even when no message was recorded on the way (the wrappers raise on it) */
void afxdev_report_failure(const char *who, int st);
/* process-wide switches, read from the environment once: AFX_NO_FUSED (size-generic kernels only: no wave-level /
 * register-resident / matrix-core specialisation), AFX_CQT_F32 (CQT octave products on the float32 matrix cores) */
int afxdev_no_fused(void);
int afxdev_cqt_f32(void);
/* first statement of every compute entry point: the object's device becomes current */
#define AFX_ENTER(o) do { if ((o) != NULL) (void)afxdev_bind_stream((o)->stream); } while (0)

int afxdev_malloc(void **dptr, size_t bytes);
void afxdev_free(void *dptr);
int afxdev_memset(void *dptr, int value, size_t bytes, void *stream);
int afxdev_h2d(void *dst, const void *src, size_t bytes, void *stream);
int afxdev_d2h(void *dst, const void *src, size_t bytes, void *stream);
int afxdev_d2d(void *dst, const void *src, size_t bytes, void *stream);
int afxdev_stream_create(void **stream);
void afxdev_stream_destroy(void *stream);
int afxdev_stream_sync(void *stream);
int afxdev_stream_wait_stream(void *waiter, void *signaler); /* device-side join: waiter waits for signaler's work so far */

/* grow-only scratch helper: (re)allocates *dptr when *capacity < bytes */
int afxdev_reserve(void **dptr, size_t *capacity, size_t bytes);

/* ---- kernels ---------------------------------------------------------- */

/* what the STFT kernel stores per bin */
enum {
    AFX_SPEC_COMPLEX = 0, /* re, im                       */
    AFX_SPEC_POWER = 1,   /* re^2+im^2                    */
    AFX_SPEC_MAG = 2,     /* sqrt(re^2+im^2)              */
    AFX_SPEC_SQUARE = 3,  /* (re+j im)^2 = re^2-im^2, 2 re im */
    AFX_SPEC_POWER_NORM = 4, /* powf(re^2+im^2, normValue) */
    AFX_SPEC_MAG_NORM = 5,   /* powf(sqrt(re^2+im^2), normValue) */
    AFX_SPEC_PHASE = 6       /* atan2f(im, max(re, 1e-16)) (spectrogram_algorithm.c:1037-1053) */
};

typedef struct {
    const float *x;        /* device, clip b starts at x + b*clipStride      */
    long long clipStride;  /* in samples                                      */
    int batch;             /* clips                                           */
    int dataLength;        /* samples per clip                                */
    int timeLength;        /* frames per clip                                 */
    int radix2Exp;         /* fftLength = 1 << radix2Exp                      */
    int hop;
    const float *window;   /* device [fftLength]                              */
    const float *twiddle;  /* device [fftLength/2] float2: (cos, -sin)(2*pi*m/fftLength)  */
    int mode;              /* AFX_SPEC_*                                      */
    float normValue;
    int binLo;             /* first bin stored                                */
    int binCount;          /* bins stored per frame                           */
    long long outPitch;    /* floats between output rows; 0: binCount         */
    float *outRe;          /* device [batch*timeLength, binCount]             */
    float *outIm;          /* device, modes COMPLEX/SQUARE only               */
    float *energy;         /* optional device [batch*timeLength] (or NULL)    */
    float *rms;
    float *zcr;
    /* centre zero padding used by the CQT octaves: frame i covers samples
     * [i*hop - padLeft, i*hop - padLeft + fftLength) of the clip, reads
     * outside [0,dataLength) give 0 */
    int padLeft;
    /* optional banded filter bank applied in the same launch (all NULL/0: bins are stored):
     * row j = sum_q bandW[q * bandNum + j] * value[bandStart[j] + q], q < bandLen[j]
     * (weights tap-major so that the rows of a wave read neighbouring words; bandOff unused);
     * outRe/outIm then are [batch*timeLength, bandNum]; bandPost = AFX_MAP_POW applies
     * powf(., bandPostArg) to the real plane.  The spans must lie inside [binLo, binLo+binCount) */
    const int *bandStart, *bandLen, *bandOff;
    const float *bandW;
    int bandNum, bandPost;
    float bandPostArg;
    /* stftObj_stft: bins binLo+j above fftLength/2 are stored as the conjugate mirror
     * X[N-k]* (the reference keeps all fftLength bins of its complex transform) */
    int fullSpectrum;
    /* what a frame reads outside [0, dataLength): AFX_PAD_ZERO, or the stftObj padding modes
     * (stft_algorithm.c:601-694) applied as an index map -- no padded copy of the clip exists */
    int padMode;
    float padValueL, padValueR; /* AFX_PAD_CONST: left of sample 0 / right of the last sample */
} AfxStftArgs;
enum { AFX_PAD_ZERO = 0, AFX_PAD_CONST = 1, AFX_PAD_REFLECT = 2, AFX_PAD_WRAP = 3 };
/* clip index a padded position maps to (-1: constant), shared by the kernel and the host tests */
#ifdef __HIPCC__
#define AFX_HD __host__ __device__
#else
#define AFX_HD
#endif
AFX_HD static inline long long afx_pad_index(long long q, int n, int mode) {
    if (q >= 0 && q < n) return q;
    if (n < 2) return -1;
    if (mode == AFX_PAD_REFLECT) {
        const long long P = 2LL * (n - 1);
        long long m = q % P;
        if (m < 0) m += P;
        return m < n ? m : P - m;
    }
    if (mode == AFX_PAD_WRAP) {
        long long m = q % n;
        if (m < 0) m += n;
        return m;
    }
    return -1;
}

/* generic framed FFT, any radix2Exp in 1..14 */
int afxk_stft(const AfxStftArgs *a, void *stream);
/* energy / rms / zcr of the windowed frames alone (x, clipStride, batch, dataLength, timeLength, radix2Exp, hop, window,
 * padLeft / padMode, energy, rms, zcr are read): for bank rows that come from a fused kernel without them */
int afxk_temporal(const AfxStftArgs *a, void *stream);

/* inverse STFT (afx_istft.hip): re/im [batch*timeLength, N] -> out[b*outStride + j],
 * j < (timeLength-1)*hop + N: out = (out + sum_frames ifft*win1) / clamp(sum_frames win2) */
typedef struct {
    const float *re, *im;  /* device, split planes, row pitch N = 1 << radix2Exp            */
    int batch, timeLength, radix2Exp, hop;
    const float *twiddle;  /* device [N/2] float2                                            */
    const float *win1;     /* device [N]: window^e   (e = 1 weighted overlap-add, 0 plain)  */
    const float *win2;     /* device [N]: window^(e+1)                                       */
    float *frames;         /* device scratch [batch*timeLength, N]                           */
    float *out;            /* device; read-modify-write                                      */
    long long outStride;
} AfxIstftArgs;
int afxk_istft(const AfxIstftArgs *a, void *stream);
/* n_fft 2048: one wave per frame (the inverse as ONE forward real transform of re + im of the Hermitian part), overlap-add in an
 * LDS ring, no frame scratch (a->frames / a->twiddle are not read); AFX_ERR_UNSUPPORTED: not its case -- run afxk_istft */
int afxk_istft_fused(const AfxIstftArgs *a, void *stream);

/* afx_spectral.hip: per-bin value (AFX_SPEC_*) of the bins [binLo, binLo+binCount) of a complex
 * spectrum re/im [rows, rowPitch] -> out [rows, binCount] (+ out2 for COMPLEX / SQUARE) */
int afxk_spec_map(const float *re, const float *im, long long rows, int rowPitch, int binLo,
                  int binCount, int mode, float normValue, float *out, float *out2, void *stream);
/* in place on data [rows, n]: optional powf(., powArg), then per-row normalisation
 * (normType = ChromaDataNormalType: 0 none, 1 max, 2 min, 3 P2, 4 P1) */
int afxk_row_post(float *data, long long rows, int n, int doPow, float powArg, int normType,
                  void *stream);

enum { AFX_MAP_NONE = 0, AFX_MAP_LOG10 = 1, AFX_MAP_CBRT = 2, AFX_MAP_POW = 3 };

/* C[M,N] = post( pre(A)[M,K] * B[N,K]^T ), row-major, f32 MFMA.
 * pre: AFX_MAP_NONE | LOG10 (log10f(max(a,1e-8))) | CBRT (powf(a,1/3))
 * post: AFX_MAP_NONE | POW (powf(c, postArg))
 * lda/ldc are row pitches in floats. */
int afxk_gemm_nt(const float *A, long long lda, const float *B, int ldb,
                 float *C, long long ldc, long long M, int N, int K,
                 int pre, int post, float postArg, void *stream);

/* the same product with every operand as three bf16 words on the bf16 matrix cores (afx_gemm_bf16.hip;
 * AFX_GEMM_BF16=1, off by default); pre is AFX_MAP_NONE; AFX_ERR_UNSUPPORTED for unaligned operands */
int afxk_gemm_nt128_bf16(const float *A, long long lda, const float *B, int ldb, float *C, long long ldc,
                         long long M, int N, int K, int post, float postArg, void *stream);

/* the dense FILTER-BANK product with the bank prepared once per object (afx_gemm_bf16.hip, k_gemm_bank_bf16x3):
 * afxk_gemm_bank_prepare splits bank[N, K] (device, row pitch ldb floats) into its three bf16 word planes in the order the
 * kernel stages them ("bank image", device memory owned by the caller: afxdev_free); afxk_gemm_nt_bank computes
 * C[M, N] = post(A[M, K] . bank^T) with the float32 rows of A split while they are staged.  A: 16-byte aligned, lda a
 * multiple of 4 floats; AFX_ERR_UNSUPPORTED otherwise (callers then run afxk_gemm_nt on the float bank).
 * The reference's product: __mdot1, src/vector/flux_vector.c:55-86 */
int afxk_gemm_bank_prepare(const float *B, int ldb, int N, int K, void **bankImage, void *stream);
int afxk_gemm_nt_bank(const float *A, long long lda, const void *bankImage, int N, int K, float *C, long long ldc,
                      long long M, int post, float postArg, void *stream);

/* "standard" cepstra post-pass (xxcc_algorithm.c:244-292): per frame, put
 * ln(max(energy,1e-8)) in front of / in place of coefficient 0 and take the
 * causal smoothing-derivative FIR (util_delta, util/flux_util.c:803-815) along
 * the coefficient axis twice.
 *   cc[rows, ccNum] -> coe/delta1/delta2 [rows, outLen], outLen = ccNum (+1 if append)
 *   energyType: 0 replace, 1 append, 2 ignore */
int afxk_xxcc_standard(const float *cc, const float *energy, long long rows, int ccNum,
                       int energyType, int deltaLen, float *coe, float *delta1, float *delta2,
                       void *stream);

/* ---- continuous wavelet transform (afx_cwt.hip) ---------------------------- */
typedef struct {
    int r1, r2;      /* L = 2^(r1+r2): columns FFT 2^r1, rows FFT 2^r2               */
    int dataLength;  /* 2^radix2Exp samples in / per scale out                       */
    int pad;         /* reflect padding on each side                                 */
    int tileCols;    /* columns per workgroup in the column passes (divides 2^r2)    */
    const float *fastTw; /* device float2 tables of the register-FFT kernels (r1 = 8, r2 = 9):
                          * [8][64] W_512^(lane d) | [8][8] W_64^(c d) | [16][16] W_256^(g p);
                          * NULL: size-generic kernels only                            */
    const int *support;  /* device [num][2]: k2 range of each wavelet's non-zeros (or NULL)  */
    /* narrow-band scales (register-FFT plan only): a wavelet whose non-zeros lie in <= 16 rows
     * k2 of the transposed spectrum needs no row pass and no intermediate -- the column kernel
     * forms its operands from the spectrum directly (k_cwt_inv_cols256_nb).
     * order: device [num] scale indices, the nWide wide scales first, then the scales of the
     * classes R = 2, 4, 8, 16 (nNarrow[0..3] of them) and the two-block classes R = 20, 24, 32 (nNarrow[4..6]);
     * NULL: every scale takes both passes */
    const int *order;
    const int *orderLo;  /* device [num][2]: (order[i], support[2 order[i]]) -- one read per workgroup */
    int nWide;
    int nNarrow[7];
    int nTd;                 /* > 0: the FIRST nTd entries of `order` run the time-domain kernel (afx_cwt_td.hip),
                              * the nWide two-pass scales and the narrow-band classes follow */
    const struct AfxCwtTdPlan_ *td;
    const struct AfxCwtTdPlan_ *tdDet; /* the derivative bank's kernels for the same nTd scales; NULL: its scales take both passes */
} AfxCwtPlanDims;
/* ---- short-kernel ("wide") scales in the time domain on the f16 matrix cores (afx_cwt_td.hip) ----
 * pair: two scales share one MFMA column tile -- 32 columns = 2 scales x (re, im) x 8 output phases */
#define AFX_CWT_TD_MAXPAIRS 48 /* pairs of scales one time-domain launch takes (two launches: long and short kernels) */
#define AFX_CWT_TD_MAXK 1024   /* taps (incl. the 8 phase shifts) of the longest kernel the LDS-resident image takes */
typedef struct {
    int scale[2];            /* result rows (C order: row 0 = highest frequency); scale[1] < 0: a single scale */
    int kh;                  /* the window of output n0 starts at position n0 - kh (multiple of 8)           */
    int ks;                  /* K steps of 16 taps (even, >= 4)                                               */
    long long img;           /* byte offset of the pair's image in the blob: [word 2][ks][64 lanes][8] f16,
                              * the B-fragment order of v_mfma_f32_32x32x16_f16 (afx_cqt_time_kernel_f16)     */
    float colMul[32];        /* 2^-s_c of the image columns                                                   */
    float colSum[32];        /* the WHOLE kernel's response to a constant 1 (its spectrum's bin 0), per image column      */
} AfxCwtTdPair;
typedef struct AfxCwtTdPlan_ {
    const AfxCwtTdPair *pairs;   /* device [nPairs], longest kernels first */
    const unsigned char *image;  /* device blob */
    const int *hostKs;           /* host [nPairs]: K steps of every pair (the launcher sizes the workgroup shares) */
    int nPairs, maxKs;
    int wrap;                    /* 0: reflect padding (isPadding, cwt_algorithm.c:404-414), 1: circular */
} AfxCwtTdPlan;
/* chunk c at x + c xStride (dataLength = 2^r samples) -> outRe/outIm [chunks][num][dataLength], rows p->pairs[].scale */
/* AFX_OK when afxk_cwt_td takes this plan for chunks of dataLength samples and `num` output scales (every precondition
 * of the launch: checked once, when the object is planned -- a plan that fails goes back to the FFT path) */
int afxk_cwt_td_fits(const AfxCwtTdPlan *p, int dataLength, int num);
int afxk_cwt_td(const AfxCwtTdPlan *p, const float *x, long long xStride, int chunks, int dataLength, int num,
                float *outRe, float *outIm, void *stream, void *streamShort /* short-kernel class; NULL: `stream` */);
#define AFX_CWT_FASTTW_FLOATS (2 * (8 * 64 + 8 * 8 + 16 * 16))
/* `chunks` signals, chunk c at x + c*xStride -> Xt[c][L] complex (transposed layout:
 * frequency k1 + 2^r1 k2 at [k1][k2]); scratchA: chunks*L complex */
int afxk_cwt_forward(const AfxCwtPlanDims *d, const float *tw, const float *x, long long xStride,
                     int chunks, float *scratchA, float *Xt, void *stream);
/* Xt[chunks][L], bankT[num][L] (same layout) -> outRe/outIm [chunks][num][dataLength];
 * scratchB: chunks*num*L complex (wide part only).  parts: AFX_CWT_WIDE = the scales that take
 * the row pass + column pass (all of them when the plan has no narrow-band order),
 * AFX_CWT_NARROW = the narrow-band scales (no scratch, any number of chunks per launch) */
#define AFX_CWT_WIDE 1
#define AFX_CWT_NARROW 2
/* (the time-domain scales of the plan, d->nTd, are not part of either: afxk_cwt_td runs them from the signal itself) */
int afxk_cwt_inverse(const AfxCwtPlanDims *d, const float *tw, const float *Xt, const float *bankT,
                     int num, int isDet, int chunks, float *scratchB, float *outRe, float *outIm,
                     int parts, void *stream);

/* transforms with L = 2^(r1+r2) <= 8192, entirely in LDS: x (NULL: re-use the spectra in X) ->
 * X[chunks][L] natural order -> outRe/outIm [chunks][num][dataLength]; bankNatural [num][L] */
int afxk_cwt_small(const AfxCwtPlanDims *d, const float *tw, const float *x, long long xStride,
                   int chunks, const float *bankNatural, int num, int isDet, float *X, float *outRe,
                   float *outIm, void *stream);

/* synchrosqueezing pass (afx_wsst.hip): W, W' [batch][num][length] -> out += W at the row the
 * instantaneous frequency maps to.  mode 0: log axis (logMin/logMax = log2f(fmin), log2f(fmax)),
 * 1: linear axis (fmin, fmax), 2: nearest entry of freNorm[num] (band centres / samplate) */
typedef struct {
    const float *wRe, *wIm, *dRe, *dIm;
    float *outRe, *outIm; /* read-modify-write */
    int num, batch;
    long long length;
    int mode;
    float fmin, fmax, logMin, logMax, thresh;
    const float *freNorm;
    int phaseInput; /* 1: dRe holds the instantaneous frequency itself (synsqObj_synsq), dIm unused */
} AfxWsstArgs;
int afxk_wsst_squeeze(const AfxWsstArgs *a, void *stream);
/* phase-difference frequency estimate of a complex matrix re/im [num][length] -> phase [num][length] */
int afxk_synsq_phase(const float *re, const float *im, int num, long long length, float *phase, void *stream);

/* time-frequency reassignment (afx_reassign.hip): planes are [batch][timeLength][F] */
typedef struct {
    const float *hRe, *hIm;   /* STFT with the analysis window                         */
    const float *dhRe, *dhIm; /* ... with its derivative (doFre)                       */
    const float *thRe, *thIm; /* ... with the time-weighted window (doTime)            */
    const float *freArr;      /* device [F]: bin centre frequencies                    */
    int *timeIdx, *freIdx;    /* device scratch, same shape as the planes              */
    float *outRe, *outIm;     /* accumulated into (float atomics); outIm unused for amplitudes */
    int batch, timeLength, F, hop, samplate;
    int doFre, doTime, resultType;
    float thresh, freScale /* -0.5 sr / pi */, timeScale /* 1 / sr */;
} AfxReassignArgs;
int afxk_reassign(const AfxReassignArgs *a, int order, int *idxScratch, void *stream);

/* ---- constant-Q transform (afx_cqt.hip) ----------------------------------- */
typedef struct {
    const float *x;        /* device: this octave's signal                            */
    int validLength;       /* samples framed (length minus the dropped tail)          */
    int timeLength, radix2Exp, hop;
    const float *twiddle;  /* device [N/2] float2                                     */
    const int *kStart, *kLen, *kOff; /* device, per kernel row: first bin, taps, offset */
    const float *kTaps;    /* device float2 taps, rows packed back to back            */
    int rowBase;           /* first kernel row of this octave (0 unless variable-Q)   */
    int rows;              /* bins per octave                                         */
    const float *scale;    /* device [num]: sqrt(len_j) (or 1 when scaling is off)    */
    float octScale;        /* sqrt(2^k) of the octave                                 */
    int num, colBase;      /* output row pitch / first output column of this octave   */
    float *outRe, *outIm;  /* device [batch][T, num]                                  */
    int batch;             /* clips per launch (grid.y)                               */
    long long xStride;     /* samples between consecutive clips in x                  */
    long long outStride;   /* floats between consecutive clips in outRe/outIm         */
    const float *timeKernel; /* device [N][colTiles*32] time-domain kernels of this octave,
                              * columns [Re 0..rows-1 | Im 0..rows-1]; NULL: FFT path      */
    int colTiles;
    const unsigned short *timeKernelH; /* device [2][N/16][64][8] f16 (hi, lo) words of the scaled image in
                              * MFMA fragment order (afx_cqt_f16.hip); NULL: float32 kernels     */
    const float *colMul;     /* device [32]: 2^-s_j of the image columns                          */
    int rightPad;            /* 0: frame t covers samples [t hop - N/2, t hop + N/2) (centre padding, the default);
                              * 1: [t hop, t hop + N) (cqt_algorithm.c:1303-1318: the streaming object pads on the right) */
} AfxCqtOctaveArgs;
int afxk_cqt_octave(const AfxCqtOctaveArgs *a, void *stream);
/* f16 matrix-core variant; AFX_ERR_UNSUPPORTED when the plan / alignment is outside its scope */
int afxk_cqt_octave_f16(const AfxCqtOctaveArgs *a, void *stream);
/* The whole default ladder in ONE persistent launch (afx_cqt_f16.hip: k_cqt_pyramid): N = 512, 12 bins per octave,
 * seven octaves, hop 128 halving to 2.  Every workgroup walks a run of 32-frame tiles of one clip; its seven waves own
 * one octave each (window -> f16 (hi, lo) planes -> matrix-core product -> rows) and make the next level's samples
 * from the same planes (the 63-tap 2:1 resampler as one more matrix-core product); the level signals live in
 * per-workgroup rings of `ring` that stay in the L2, so a clip is read from HBM once and no level signal goes back to it.
 * cqt_algorithm.c:951-1048 (octave recursion), dsp/resample_algorithm.c:430-521 (the resampler). */
#define AFX_CQT_PYR_LEVELS 7
#define AFX_CQT_PYR_RING_FLOATS 17408 /* per workgroup: rings of 8192, 4096, 2048, 1024, 1024, 1024 samples */
#define AFX_CQT_PYR_MAX_WGS 256
#define AFX_CQT_PYR_TAB_COPY 704       /* bytes of one shifted copy of the tap table (328 f16 entries, padded) */
#define AFX_CQT_PYR_TAB_HALFS (2 * 4 * AFX_CQT_PYR_TAB_COPY / 2) /* [2 words][4 copies]                */
typedef struct {
    const float *x;          /* device clips (level 0)                                  */
    long long xStride;       /* samples between clips                                   */
    int batch, timeLength, num; /* clips, frames per clip, output row pitch (84)       */
    int len[AFX_CQT_PYR_LEVELS];   /* samples of every level (floorf(len/2) ladder)    */
    int valid[AFX_CQT_PYR_LEVELS]; /* framed samples of every level (stft_algorithm.c:838-843) */
    const unsigned short *timeKernelH; /* as AfxCqtOctaveArgs                           */
    const float *colMul, *scale;
    float octScale[AFX_CQT_PYR_LEVELS]; /* sqrt(2^k) of level k (cqt_algorithm.c:1218-1221)   */
    float *outRe, *outIm;    /* device [batch][T, num]                                  */
    long long outStride;
    float *ring;             /* device scratch, AFX_CQT_PYR_RING_FLOATS floats per workgroup */
    const unsigned short *decTab; /* device [AFX_CQT_PYR_TAB_HALFS]: the resampler taps h[|d|] 2^15 as f16 (hi, lo)
                              * words, d = -160 ... 167, four copies shifted by 0, 2, 4, 6 entries (afx_cqt_dec_table) */
    float decMul;            /* 2^-15 / sqrt(ratio): undoes the table scaling, applies the resampler's isScale   */
    int chunksPerClip, tilesPerChunk; /* work items: clip-major runs of tiles            */
    /* chroma in the same launch (chromaNum == 12 == bins per octave): NULL = off         */
    float *chroma;           /* device [batch][T, 12]                                   */
    int chromaClass[12];     /* class of bin j of an octave (the 0/1 folding matrix)    */
    int chromaMag, chromaNorm; /* |Q| instead of |Q|^2; 0 none 1 max 2 min 3 P2 4 P1     */
    unsigned long long *timing; /* NULL, or device [workgroups][11 waves][8]: shader cycles per phase, summed
                              * (the instrumented instantiation; tools/pyr_phases.py)      */
} AfxCqtPyramidArgs;
/* workgroups the launch will use (the caller sizes `ring` with it) */
int afxk_cqt_pyramid_plan(int batch, int timeLength, int maxTiles, int *chunksPerClip, int *tilesPerChunk);
int afxk_cqt_pyramid(const AfxCqtPyramidArgs *a, void *stream);
/* batch clips: x + b*xStride -> y + b*yStride */
int afxk_cqt_decimate(const float *x, int srcLen, long long xStride, float *y, int dstLen,
                      long long yStride, int batch, const float *taps32, float sqrtRatio,
                      void *stream);
/* cqhc / deconv: in[rows, num] -> timbre / pitch [rows, num] and/or hc [rows, hcNum]
 * (hcIdx[hcNum]: positions in the timbre sequence); transform length 2^radix2Exp >= 2 num;
 * twiddle: device [M/2] float2 */
int afxk_cqt_deconv(const float *in, long long rows, int num, int radix2Exp, const float *twiddle,
                    const int *hcIdx, int hcNum, float *outTimbre, float *outPitch, float *outHc,
                    void *stream);
/* the 0/1 folding matrix as per-class bin lists: class c owns bins[start[c] .. start[c+1]) in ascending order
 * (the order of the matrix product); chromaNum <= 64, num <= 255 (afx_cqt.c: afx_chroma_lists) */
typedef struct AfxChromaLists_ {
    unsigned short start[65];
    unsigned char bins[256];
} AfxChromaLists;
/* fold: device 0/1 matrix [chromaNum][num]; lists: the same as bin lists (host struct, a kernel argument of
 * k_cqt_chroma); NULL or a plan beyond the list form (num > 255, chromaNum > 64): the size-generic flag scan */
int afxk_cqt_chroma(const float *re, const float *im, long long rows, int num,
                    const unsigned char *fold, const AfxChromaLists *lists, int chromaNum, int isMag,
                    int normType, float *out, void *stream);

/* cepstrogram (afx_cepstrogram.hip): one clip, timeLength frames */
typedef struct {
    const float *x;      /* device samples, or NULL to start from specRe/specIm     */
    int timeLength, radix2Exp, hop, cepNum;
    const float *window; /* device [N]                                              */
    const float *twiddle;/* device [N/2] float2                                     */
    float *specRe;       /* device [T,N] spectrum cache: written when x != NULL     */
    float *specIm;       /*   (may be NULL), read when x == NULL                    */
    float *out1, *out2, *out3; /* device [T, N/2+1]; any may be NULL                */
    int framesPerClip;   /* > 0: timeLength counts the frames of several clips,      */
    long long clipStride;/*      frame f starts at x + (f / fpc) * clipStride + (f % fpc) * hop */
    const float *fastTab;/* device tables of the wave kernels for N = 2048 / 4096
                          * (afxk_cepstrogram_fast_tables), or NULL: size-generic kernel only */
} AfxCepstrogramArgs;
int afxk_cepstrogram(const AfxCepstrogramArgs *a, void *stream);
/* host: fills tab[AFX_CEPSTROGRAM_FASTTAB_FLOATS] for AfxCepstrogramArgs.fastTab (fftLength 2048 or 4096) */
#define AFX_CEPSTROGRAM_FASTTAB_FLOATS (2 * (16 * 64 + 72 + 1024 + 1025))
void afxk_cepstrogram_fast_tables(float *tab, int fftLength);

/* specialised rectify + DCT for cepstra (afx_cepstrum.hip): out[rows, ccNum] =
 * pre(in)[rows, num] . dct[ccNum, num]^T */
int afxk_cepstrum_supported(const float *in, int num, int ccNum);
int afxk_cepstrum(const float *in, long long rows, int num, const float *dct, int ccNum, int pre,
                  float *out, void *stream);

/* ---- fused STFT -> banded filter bank kernel (afx_melfused.hip) ---------- */

/* Banded view of a filter bank, one entry per lane of a 64-lane wave: every
 * lane owns up to two bank rows, a long one (A) and a short one (B); each row
 * is the contiguous bin range [start, start+len) that holds all its non-zero
 * weights.  Weight arrays are tap-major, [taps][64], zero padded. */
typedef struct {
    int num;        /* bank rows                                  */
    int tapsA;      /* longest A row                              */
    int tapsB;      /* longest B row                              */
    int startA[64];
    int startB[64];
    int rowA[64];   /* bank row stored by lane (-1: none)         */
    int rowB[64];
    float *wA;      /* host, [tapsA][64]                          */
    float *wB;      /* host, [tapsB][64]                          */
    /* split plans (afx_bandplan_build_split): a slot holds a SEGMENT of a row, rowA/rowB are -1
     * and bank row r is the sum of up to four slot results, in ascending bin order:
     * segIdx[r] packs four slot indices, 8 bits each, lowest byte first -- lane l's A slot is
     * l, its B slot 64 + l, 128 = none (contributes zero) */
    int split;
    unsigned segIdx[128];
} AfxBandPlan;

typedef struct {
    const float *x;
    long long clipStride;
    int batch, dataLength, timeLength, hop;
    int specMap;    /* 0 |S|^2, 1 |S|, 2 |S|^(2*normValue); complex results:
                     * 3 S, 4 S^2 (real plane -> out, imaginary plane -> outIm) */
    int postPow;    /* 1: powf(result, normValue)                           */
    float normValue;
    float *out;     /* device [batch*timeLength, num]                       */
    float *outIm;   /* device, same shape; complex result modes only        */
    /* optional fusions: cepstra (every fused kernel, real results) and temporal features (n_fft 2048); a run that cannot
     * honour them returns AFX_ERR_UNSUPPORTED and the caller takes the separate kernels */
    const float *dct; /* device [num, num] orthonormal DCT-II: cepstra of the rows in the same launch */
    int ccNum;        /*   first ccNum (<= 16) coefficients of the rectified rows; needs `out`           */
    int ccRectify;    /*   0: log10f(max(row, 1e-8)), 1: powf(row, 1/3) (xxcc_algorithm.c:124-137)        */
    float *cc;        /*   device [batch*timeLength, ccNum]                                          */
    float *energy, *rms, *zcr; /* device [batch*timeLength]: temporal features of the windowed frame */
} AfxMelFusedArgs;

/* variant index able to run (radix2Exp, tapsA, tapsB), or -1 */
int afxk_melfused_variant(int radix2Exp, int tapsA, int tapsB);
int afxk_melfused_create(void **plan, int radix2Exp, const float *hWindow,
                         const AfxBandPlan *band, void *stream);
int afxk_melfused_run(void *plan, const AfxMelFusedArgs *a, void *stream);
void afxk_melfused_destroy(void *plan);
/* 0: no plan, 1: rows in whole slots, 2: split plan (row segments); 101: the n_fft 1024 kernel; 201 / 202: the n_fft 4096 kernel, whole rows / split plan */
int afxk_melfused_kind(const void *plan);

#ifdef __cplusplus
}
#endif
#endif /* AFX_DEVICE_H */
