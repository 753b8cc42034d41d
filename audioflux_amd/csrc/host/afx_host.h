/* afx_host.h -- declarations shared by the C host objects.  Setup-time work
 * (windows, filter banks, DCT matrices, wavelet scales, CQT kernels) is done
 * here on the host in float32 with the same operation order as the reference
 * so that band edges / rounding decisions come out identical, then uploaded
 * once per object.  All per-sample arithmetic runs in the HIP kernels.
 */
#ifndef AFX_HOST_H
#define AFX_HOST_H

#include <stddef.h>

#include "flux_base.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ---- afx_window.c ------------------------------------------------------ */
/* length-`length` window; periodic != 0 builds the length+1 symmetric window
 * and drops its last sample.  Kaiser beta / Gauss alpha / Tukey alpha take the
 * reference defaults (5, 2.5, 0.5).  Returns a malloc'ed array or NULL. */
float *afx_window_create(WindowType type, int length, int periodic);
/* the window an STFT of frame length `length` uses (periodic for the cosine
 * family, symmetric for bartlett/triang/bartlett-hann/bohman) */
float *afx_window_fft(WindowType type, int length);
float *afx_window_kaiser(int length, float beta);

/* ---- afx_auditory.c ---------------------------------------------------- */
/* fills bank[num*(fftLength/2+1)] (must be zeroed), fre[num], bin[num]; 0 or AFX_ERR_NOMEM */
int afx_auditory_bank(int num, int fftLength, int samplate,
                       SpectralFilterBankScaleType scale,
                       SpectralFilterBankStyleType style,
                       SpectralFilterBankNormalType normal,
                       float lowFre, float highFre, int binPerOctave,
                       float *bank, float *fre, int *bin);
void afx_auditory_revise_linear(int num, float lowFre, float highFre, float detFre, int isEdge,
                                float *lowOut, float *highOut);
void afx_auditory_revise_log(int num, float lowFre, float highFre, int binPerOctave, int isEdge,
                             float *lowOut, float *highOut);
float afx_fre_to_log(float fre, float binPerOctave);
float afx_log_to_fre(float value, float binPerOctave);

/* ---- afx_util.c -------------------------------------------------------- */
float *afx_linspace(float start, float stop, int length, int noStop);
/* float2 table (cos, -sin)(2*pi*m/n), m < n/2, evaluated in double */
float *afx_twiddle_table(int n);
/* orthonormal DCT-II matrix rows 0..rows-1, D[c*num+n] */
float *afx_dct2_matrix(int num, int rows);
int afx_is_pow2(int v);
int afx_ceil_pow2(int v);
int afx_log2_exact(int v);
/* the reference's float32 radix-2 DIT FFT, operation for operation (kernel banks are
 * thresholded on its output, so the host copy must round identically) */
void afx_fft_ref32(int radix2Exp, const float *re1, const float *im1, float *re2, float *im2);
/* in-place radix-2 complex FFT in double (inverse != 0: e^{+...}, unnormalised); 0 or AFX_ERR_NOMEM */
int afx_fft_f64(int log2n, double *re, double *im, int inverse);

/* ---- chroma banks ------------------------------------------------------ */
/* afx_cqt.c: 0/1 folding matrix [chromaNum, num] of log-spaced bins onto chroma classes
 * (src/filterbank/chroma_filterBank.c:176-264); calloc'ed, caller frees */
unsigned char *afx_chroma_fold(int chromaNum, int num, int bpo, float minFre);
/* afx_cqt.c: one octave group's time-domain image G [N][32] -> IEEE binary16 (hi, lo) words of its columns
 * scaled by 2^s_j to a peak in [2^13, 2^14) (round to nearest even; lo = f16(v - hi)), in the B-fragment order
 * of v_mfma_f32_32x32x16_f16: out[word][N/16 steps][64 lanes][8], lane l = 32 g + j holds rows
 * 16 ks + 8 g + e of column j; colMul[32] = 2^-s_j (afx_cqt_f16.hip) */
void afx_cqt_time_kernel_f16(const float *G, int N, unsigned short *out, float *colMul);
/* afx_cqt.c: the 2:1 resampler's 32 taps -> out[AFX_CQT_PYR_TAB_HALFS] (afx_device.h): T[d] = h[|d|] 2^15 as binary16
 * (hi, lo) words, [word][copy a][x] = T[x - 160 - 2a] -- the operand table of k_cqt_pyramid's resampler product */
void afx_cqt_dec_table(const float *taps32, unsigned short *out);
/* afx_cqt.c: 0/1 folding matrix -> per-class bin lists (afx_device.h: AfxChromaLists); 0 on success */
struct AfxChromaLists_;
int afx_chroma_lists(const unsigned char *fold, int chromaNum, int num, struct AfxChromaLists_ *out);
/* afx_cqt.c: clips per pass of the batched CQT calls for clips of rowFloats = T * num output floats per plane:
 * the fewest equal passes of <= 448 MB of output (AFX_CQT_CHUNK=<clips> overrides) */
int afx_cqt_pass_clips(long long rowFloats, int batch);
/* afx_spectrogram.c: Gaussian STFT-chroma bank [num, fftLength/2+1]
 * (src/filterbank/chroma_filterBank.c:13-174, default octave centre 5 / width 2) */
float *afx_chroma_stft_bank(int num, int fftLength, int samplate);

/* ---- afx_cwt.c: shared with the pseudo wavelet object (afx_pwt.c) ------------------------ */
struct OpaqueCWT;
/* the CWT execution plan over a caller-built frequency-domain bank [num][L] (natural bin order)
 * and band arrays; L = afx_cwt_fft_length(radix2Exp, isPadding) (-1: unsupported) */
int afx_cwt_create_custom(struct OpaqueCWT **cwtObj, int num, int radix2Exp, int samplate, int isPadding,
                          const float *bank, const float *fre, const int *bin, const char *who);
long long afx_cwt_fft_length(int radix2Exp, int isPadding);
/* narrow-band scale planning of the register-FFT inverse (host-only, exported for tests):
 * rows k2 of the transposed spectrum (k = k1 + 2^r1 k2) that hold each wavelet's non-zeros, and
 * the execution order wide scales | classes R = 2, 4, 8, 16 | two-block classes R = 20, 24, 32 (see afx_device.h) */
void afx_cwt_support_host(const float *bank, int num, long long fftLength, int r1, int *sup);
void afx_cwt_classify_host(const int *sup, int num, int maxR, int *order, int *nWide, int nNarrow[7]);
/* widest narrow-band class used unless AFX_CWT_NARROW_MAX says otherwise (0: every scale takes
 * both passes).  BASELINE cfg 4, round 3 (profiles/r03_cwt_nb2.txt): 20 -- the four scales of 17 ... 20 rows that
 * are too long for the time-domain kernel -- 34.5 k chunks/s; 16: 33.2 k; 24 / 32 take scales away from the
 * time-domain kernel, which is the cheaper one for them: 32.9 k / 30.2 k */
#define AFX_CWT_NARROW_MAX_DEFAULT 20

/* ---- afx_bandplan.c ----------------------------------------------------- */
struct AfxBandPlanTag; /* AfxBandPlan is declared in afx_device.h */

#endif /* AFX_HOST_H */
