/* afx_window.c -- analysis windows, host side, float32.
 *
 * Restates what the reference's window_calFFTWindow (src/dsp/flux_window.c:
 * 890-940) and window_create* (flux_window.c:64-302) produce, as closed
 * forms: every reference window is "evaluate a half-window formula h(i) for
 * i < halfLen, mirror it for the rest" (flux_window.c:281-600), and a
 * periodic window is the symmetric window of length+1 with the last sample
 * dropped (flux_window.c:64-78).  The float/double promotion of each formula
 * follows the reference expression so the stored float32 values agree.
 */
#include <math.h>
#include <stdlib.h>

#include "afx_host.h"

/* I0 by its power series, 15 terms, float accumulators (flux_window.c:668-689) */
static float bessel_i0(float a) {
    float sum = 1, b = a / 2, num = 1, den = 1, mid;
    for (int k = 1; k < 16; k++) {
        num = num * b;
        den = den * k;
        mid = num / den;
        sum = sum + mid * mid;
    }
    return sum;
}

#define AFX_KAISER_BETA 5.0f /* window_calFFTWindow's Kaiser family; afx_window_kaiser passes its own */

typedef struct {
    int n;          /* symmetric length being generated                 */
    int half;       /* samples computed from the formula                */
    float den_i0;   /* kaiser                                           */
    float beta;     /* kaiser (no global state: constructors run concurrently on different threads) */
    float step;     /* bohman linspace step                             */
    int gauss_half; /* gauss: halfLen of its own mirrored construction  */
} WinCtx;

/* h(i) for i in [0, half) -- one case per family */
static float half_value(WindowType type, const WinCtx *c, int i) {
    const int n = c->n;
    const int d = n - 1; /* denominator most families use (flux_window.c:292,312,...) */
    switch (type) {
        case Window_Hann: /* flux_window.c:738-747 */
            return (float)(0.5 - 0.5 * cosf((float)(2 * M_PI * i / d)));
        case Window_Hamm: /* :749-758 */
            return (float)(0.54 - 0.46 * cosf((float)(2 * M_PI * i / d)));
        case Window_Blackman: /* :761-770, sample 0 stays 0 */
            if (i == 0) return 0.f;
            return (float)(0.42 - 0.5 * cosf((float)(2 * M_PI * i / d)) +
                           0.08 * cosf((float)(4 * M_PI * i / d)));
        case Window_Blackman_Harris: { /* :789-812, float coefficients */
            const float a0 = 0.35875f, a1 = 0.48829f, a2 = 0.14128f, a3 = 0.01168f;
            float v = a0 - a1 * cosf((float)(2 * M_PI * i / d));
            v = v + a2 * cosf((float)(4 * M_PI * i / d));
            v = v - a3 * cosf((float)(6 * M_PI * i / d));
            return v;
        }
        case Window_Blackman_Nuttall: { /* :814-837 */
            const float a0 = 0.3635819f, a1 = 0.4891775f, a2 = 0.1365995f, a3 = 0.0106411f;
            float v = a0 - a1 * cosf((float)(2 * M_PI * i / d));
            v = v + a2 * cosf((float)(4 * M_PI * i / d));
            v = v - a3 * cosf((float)(6 * M_PI * i / d));
            return v;
        }
        case Window_Flattop: { /* :839-865 */
            const float a0 = 0.21557895f, a1 = 0.41663158f, a2 = 0.277263158f,
                        a3 = 0.083578947f, a4 = 0.006947368f;
            float v = a0 - a1 * cosf((float)(2 * M_PI * i / d));
            v = v + a2 * cosf((float)(4 * M_PI * i / d));
            v = v - a3 * cosf((float)(6 * M_PI * i / d));
            v = v + a4 * cosf((float)(8 * M_PI * i / d));
            return v;
        }
        case Window_Kaiser: { /* :696-716, beta 5 */
            const float a = c->beta;
            float u = (float)(2.0 * i / d - 1);
            float b = a * sqrtf(1 - u * u);
            return bessel_i0(b) / c->den_i0;
        }
        case Window_Bartlett: /* :616-625 */
            return (float)(2.0 * i / d);
        case Window_Bartlett_Hann: /* :628-637, sample 0 stays 0 */
            if (i == 0) return 0.f;
            return (float)(0.62 - 0.48 * fabs(1.0 * i / d - 0.5) +
                           0.38 * cosf((float)(2 * M_PI * (1.0 * i / d - 0.5))));
        case Window_Triang: { /* :639-660, denominator n (+1 when n is odd) */
            float det = (n & 1) ? 1.f : 0.5f;
            int extra = (n & 1) ? 1 : 0;
            return (float)(2.0 * (i + det) / (n + extra));
        }
        case Window_Bohman: { /* :773-787, grid = linspace(-1,1,n) */
            if (i == 0) return 0.f;
            float l = fabsf(-1.f + i * c->step);
            return (float)((1 - l) * cosf((float)(M_PI * l)) + 1 / M_PI * sinf((float)(M_PI * l)));
        }
        case Window_Gauss: { /* :718-736: stored reversed, alpha 2.5 */
            const float a = 2.5f;
            float det = (n & 1) ? 0.f : 0.5f;
            int k = c->gauss_half - 1 - i;
            float v = 2 * a * (k - det) / (n - 1);
            v = (float)(-0.5 * v * v);
            return expf(v);
        }
        default:
            return 1.f;
    }
}

/* symmetric window of length n into out[n] */
static void fill_symmetric(WindowType type, int n, float *out, float kaiserBeta) {
    WinCtx c;
    c.n = n;
    c.beta = kaiserBeta;
    c.den_i0 = 0;
    c.step = 0;
    c.gauss_half = 0;
    int half = (n & 1) ? (n + 1) / 2 : n / 2;

    if (type == Window_Rect) {
        for (int i = 0; i < n; i++) out[i] = 1.f;
        return;
    }
    if (type == Window_Tukey) { /* flux_window.c:573-614, alpha 0.5 */
        const float a = 0.5f;
        float step = (1.f - 0.f) / (n - 1 > 0 ? n - 1 : 1);
        for (int i = 0; i < n; i++) {
            float x = 0.f + i * step;
            if (x >= 0 && x < a / 2) {
                out[i] = (float)(0.5 * (1 + cosf((float)(2 * M_PI / a * (x - a / 2)))));
            } else if (x >= a / 2 && x < (1 - a / 2)) {
                out[i] = 1.f;
            } else {
                out[i] = (float)(0.5 * (1 + cosf((float)(2 * M_PI / a * (x - 1 + a / 2)))));
            }
        }
        return;
    }
    if (type == Window_Kaiser) c.den_i0 = bessel_i0(c.beta);
    if (type == Window_Bohman) c.step = (1.f - (-1.f)) / (n - 1 > 0 ? n - 1 : 1);
    if (type == Window_Gauss) {
        /* the gauss family computes one extra sample for even n (flux_window.c:533-538) */
        half = (n & 1) ? (n + 1) / 2 : n / 2 + 1;
        c.gauss_half = half;
    }
    /* triang evaluates i <= halfLen (flux_window.c:655); the extra sample is
     * overwritten by the mirror, so computing i < half is equivalent */
    c.half = half;
    for (int i = 0; i < half && i < n; i++) out[i] = half_value(type, &c, i);
    for (int i = n - 1; i >= half; i--) out[i] = out[n - 1 - i];
}

static float *window_create_beta(WindowType type, int length, int periodic, float kaiserBeta) {
    if (length <= 0) return NULL;
    float *w = (float *)calloc((size_t)length + 2, sizeof(float));
    if (!w) return NULL;
    if (length == 1) { /* flux_window.c:68-71 */
        w[0] = 1.f;
        return w;
    }
    int n = periodic ? length + 1 : length;
    float *tmp = (float *)calloc((size_t)n + 2, sizeof(float));
    if (!tmp) {
        free(w);
        return NULL;
    }
    fill_symmetric(type, n, tmp, kaiserBeta);
    for (int i = 0; i < length; i++) w[i] = tmp[i];
    free(tmp);
    return w;
}

float *afx_window_create(WindowType type, int length, int periodic) {
    return window_create_beta(type, length, periodic, AFX_KAISER_BETA);
}

float *afx_window_fft(WindowType type, int length) {
    /* flux_window.c:890-940: symmetric-only families keep flag 0 */
    int periodic = 1;
    if (type == Window_Bartlett || type == Window_Triang || type == Window_Bartlett_Hann ||
        type == Window_Bohman || type == Window_Rect) {
        periodic = 0;
    }
    if ((int)type < 0 || (int)type > (int)Window_Tukey) type = Window_Rect;
    return afx_window_create(type, length, periodic);
}

/* symmetric Kaiser window with an explicit beta (window_createKaiser(length, 0, &beta),
 * flux_window.c:112-127); used by the resampler table */
float *afx_window_kaiser(int length, float beta) {
    return window_create_beta(Window_Kaiser, length, 0, beta > 0 ? beta : AFX_KAISER_BETA);
}
