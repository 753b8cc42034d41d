/* afx_util.c -- small host helpers (linspace, twiddle and DCT tables). */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "afx_host.h"

/* float32 linspace with the reference's accumulation arr[i] = start + i*step
 * (src/vector/flux_vector.c:2145-2162); band edges depend on these bits */
float *afx_linspace(float start, float stop, int length, int noStop) {
    float *arr = (float *)calloc((size_t)(length > 0 ? length : 1), sizeof(float));
    float step;
    if (!arr) return NULL;
    if (!noStop) {
        step = (stop - start) / (length - 1 > 0 ? length - 1 : 1);
    } else {
        step = (stop - start) / length;
    }
    for (int i = 0; i < length; i++) arr[i] = start + i * step;
    return arr;
}

/* (cos, -sin)(2 pi m / n) for m < n/2, interleaved.  The reference builds the
 * same table with cosf/sinf (src/dsp/fft_algorithm.c:882-891); evaluating in
 * double and rounding once is at least as accurate. */
float *afx_twiddle_table(int n) {
    int half = n / 2 > 0 ? n / 2 : 1;
    float *t = (float *)malloc(sizeof(float) * 2 * (size_t)half);
    if (!t) return NULL;
    for (int m = 0; m < half; m++) {
        double a = 2.0 * M_PI * (double)m / (double)n;
        t[2 * m] = (float)cos(a);
        t[2 * m + 1] = (float)(-sin(a));
    }
    return t;
}

/* orthonormal DCT-II: D[c][n] = s_c cos(pi (2n+1) c / (2 num)),
 * s_0 = sqrt(1/num), s_c = sqrt(2/num).  Equals what the reference obtains
 * through its FFT-based DCT (src/dsp/fft_algorithm.c:625-674) or its cosine
 * matrix (src/dsp/dct_algorithm.c:170-181) with norm enabled. */
float *afx_dct2_matrix(int num, int rows) {
    float *d = (float *)malloc(sizeof(float) * (size_t)num * (size_t)rows);
    if (!d) return NULL;
    double s0 = sqrt(1.0 / num), s1 = sqrt(2.0 / num);
    for (int c = 0; c < rows; c++) {
        for (int n = 0; n < num; n++) {
            double v = cos(M_PI * (2.0 * n + 1.0) * c / (2.0 * num));
            d[(size_t)c * num + n] = (float)(v * (c == 0 ? s0 : s1));
        }
    }
    return d;
}

int afx_is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

/* util_ceilPowerTwo (src/util/flux_util.c:33-51) */
int afx_ceil_pow2(int value) {
    int n = 1;
    if (value < 1) return n;
    if (afx_is_pow2(value)) return value;
    while (value) {
        value >>= 1;
        n <<= 1;
    }
    return n;
}

int afx_log2_exact(int v) {
    int r = 0;
    if (!afx_is_pow2(v)) return 0;
    while ((1 << r) < v) r++;
    return r;
}

/* Radix-2 decimation-in-time FFT in float32 with cosf/sinf twiddles, bit-reversed
 * load then log2(N) butterfly passes -- the arithmetic of the reference's built-in
 * _fftObj_fft (src/dsp/fft_algorithm.c:450-519, twiddles :882-891).  Only used at plan
 * time for the CQT spectral kernels, whose entries are kept or dropped by a float32
 * threshold test (src/filterbank/cqt_filterBank.c:124-138). */
void afx_fft_ref32(int r, const float *re1, const float *im1, float *re2, float *im2) {
    const int n = 1 << r, half = n / 2;
    float *wc = (float *)malloc(sizeof(float) * (size_t)(half > 0 ? half : 1));
    float *ws = (float *)malloc(sizeof(float) * (size_t)(half > 0 ? half : 1));
    if (!wc || !ws) { /* out of memory: a zero spectrum (every kernel entry falls under the threshold) */
        free(wc);
        free(ws);
        memset(re2, 0, sizeof(float) * (size_t)n);
        memset(im2, 0, sizeof(float) * (size_t)n);
        return;
    }
    for (int i = 0; i < half; i++) {
        wc[i] = cosf((float)(2 * M_PI * i / n));
        ws[i] = -sinf((float)(2 * M_PI * i / n));
    }
    for (int i = 0; i < n; i++) {
        unsigned rev = 0;
        for (int b = 0; b < r; b++) rev |= ((unsigned)(i >> b) & 1u) << (r - 1 - b);
        re2[i] = re1[rev];
        im2[i] = im1 ? im1[rev] : 0.f;
    }
    for (int p = 1; p <= r; p++) {
        const int s = 1 << (p - 1);
        for (int g = 0; g < s; g++) {
            const int w = g * (1 << (r - p));
            const float c = wc[w], sn = ws[w];
            for (int b = g; b <= n - 1; b += (1 << p)) {
                const float tr = re2[b + s] * c - im2[b + s] * sn;
                const float ti = re2[b + s] * sn + im2[b + s] * c;
                re2[b + s] = re2[b] - tr;
                im2[b + s] = im2[b] - ti;
                re2[b] = re2[b] + tr;
                im2[b] = im2[b] + ti;
            }
        }
    }
    free(wc);
    free(ws);
}

/* in-place radix-2 complex FFT in double: X[k] = sum_n x[n] e^{-+ 2 pi i k n / N} (inverse != 0: the + sign, no 1/N).
 * Plan-time helper (time-domain images of the wide CWT scales, afx_cwt.c); 0 or AFX_ERR_NOMEM */
int afx_fft_f64(int log2n, double *re, double *im, int inverse) {
    const size_t n = (size_t)1 << log2n, half = n / 2;
    if (log2n < 1) return 0;
    double *wc = (double *)malloc(sizeof(double) * half), *ws = (double *)malloc(sizeof(double) * half);
    if (!wc || !ws) {
        free(wc);
        free(ws);
        return -5;
    }
    for (size_t i = 0; i < half; i++) {
        const double ang = 2.0 * M_PI * (double)i / (double)n;
        wc[i] = cos(ang);
        ws[i] = inverse ? sin(ang) : -sin(ang);
    }
    for (size_t i = 0, j = 0; i < n; i++) { /* bit reversal */
        if (i < j) {
            double t = re[i]; re[i] = re[j]; re[j] = t;
            t = im[i]; im[i] = im[j]; im[j] = t;
        }
        size_t m = half;
        while (m >= 1 && (j & m)) {
            j ^= m;
            m >>= 1;
        }
        j |= m;
    }
    for (size_t len = 2, step = half; len <= n; len <<= 1, step >>= 1) {
        const size_t h = len / 2;
        for (size_t b = 0; b < n; b += len)
            for (size_t k = 0; k < h; k++) {
                const double c = wc[k * step], sn = ws[k * step];
                const double ur = re[b + k], ui = im[b + k];
                const double vr = re[b + k + h] * c - im[b + k + h] * sn, vi = re[b + k + h] * sn + im[b + k + h] * c;
                re[b + k] = ur + vr;
                im[b + k] = ui + vi;
                re[b + k + h] = ur - vr;
                im[b + k + h] = ui - vi;
            }
    }
    free(wc);
    free(ws);
    return 0;
}

/* ---- on-wire format of gathered features ------------------------------------
 * NumPy .npy, format version 1.0: magic, header dict {'descr': '<f4', 'fortran_order': False,
 * 'shape': (...)}, padded with spaces to a multiple of 64 bytes, then the C-ordered float32 data.
 * Any consumer reads it with numpy.load / memory-maps it; no reference counterpart (the reference
 * leaves persistence to its Python callers). */
#include <stdio.h>

int afx_write_npy_f32(const char *path, const float *data, int ndim, const long long *shape) {
    if (!path || !data || ndim < 1 || ndim > 8 || !shape) return -6;
    char dict[256];
    int n = snprintf(dict, sizeof(dict), "{'descr': '<f4', 'fortran_order': False, 'shape': (");
    size_t count = 1;
    for (int i = 0; i < ndim; i++) {
        if (shape[i] < 0) return -6;
        count *= (size_t)shape[i];
        n += snprintf(dict + n, sizeof(dict) - (size_t)n, i + 1 < ndim ? "%lld, " : (ndim == 1 ? "%lld," : "%lld"), shape[i]);
    }
    n += snprintf(dict + n, sizeof(dict) - (size_t)n, "), }");
    /* 10 bytes of preamble + header, newline-terminated, total a multiple of 64 */
    int headerLen = n + 1;
    const int pad = (64 - (10 + headerLen) % 64) % 64;
    headerLen += pad;
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    const unsigned char pre[10] = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0, (unsigned char)(headerLen & 255),
                                   (unsigned char)(headerLen >> 8)};
    int ok = fwrite(pre, 1, 10, f) == 10 && fwrite(dict, 1, (size_t)n, f) == (size_t)n;
    for (int i = 0; ok && i < pad; i++) ok = fputc(' ', f) != EOF;
    ok = ok && fputc('\n', f) != EOF;
    ok = ok && fwrite(data, sizeof(float), count, f) == count;
    if (fclose(f) != 0) ok = 0;
    return ok ? 0 : -1;
}
