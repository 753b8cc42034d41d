/* The device-pointer ("...BatchDevice") entry points of include/afx_batch.h under AddressSanitizer / UBSan with the
 * stand-in device layer of tests/hoststub/gen_stub.py: every buffer handed in has exactly the documented size, the
 * stand-in launchers read / write every range the real kernels would -- a scratch buffer that is too small, or a
 * pointer that is off by a pitch, is a sanitizer report.  (The Python GPU tests reach these entry points through
 * torch tensors, which the sanitizer run cannot use.)  Covers: mel + MFCC in one call (fused and separate cepstra),
 * the dense-bank route cut into several chunks (AFX_SCRATCH_MB), temporal features, complex results, STFT / inverse
 * STFT, spectrogram object, cepstrogram, CWT at 2^16 with several chunk groups and chains, PWT, WSST, reassignment.
 * Exit status 0 and no sanitizer report = pass. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "afx_batch.h"

#define CHECK(x)                                                        \
    do {                                                                \
        int _s = (x);                                                   \
        if (_s != 0) {                                                  \
            fprintf(stderr, "%s -> %d (line %d)\n", #x, _s, __LINE__);  \
            return 1;                                                   \
        }                                                               \
    } while (0)

static float *buf(size_t n) {
    float *p = (float *)malloc(sizeof(float) * (n ? n : 1));
    if (!p) {
        fprintf(stderr, "out of memory\n");
        exit(2);
    }
    for (size_t i = 0; i < n; i++) p[i] = (float)((i * 2654435761u) % 997) * 1e-3f - 0.5f;
    return p;
}

static int bft_paths_at(void *stream, int r2) {
    const int clips = 5, n = 16000, hop = (1 << r2) / 4;
    const long long stride = n + 24;
    int sr = 16000, slide = hop;
    float lo = 0.f, hi = 8000.f;
    float *x = buf((size_t)clips * stride);
    for (int variant = 0; variant < 4; variant++) {
        /* 0 mel-128 (fused kernel), 1 gammatone-128 (dense route), 2 mel-40 (split plan), 3 mel-128 + temporal */
        BFTObj o = NULL;
        SpectralFilterBankScaleType scale = variant == 1 ? SpectralFilterBankScale_Erb : SpectralFilterBankScale_Mel;
        SpectralFilterBankStyleType style = variant == 1 ? SpectralFilterBankStyle_Gammatone : SpectralFilterBankStyle_Slaney;
        SpectralDataType dt = SpectralData_Power;
        int temporal = variant == 3;
        const int num = variant == 2 ? (r2 <= 10 ? 13 : 40) : 128;  /* rows longer than the size's tap variants */
        CHECK(bftObj_new(&o, num, r2, &sr, &lo, &hi, NULL, NULL, &slide, &scale, &style, NULL, &dt, NULL, &temporal));
        const int T = bftObj_calTimeLength(o, n);
        float *re = buf((size_t)clips * T * num), *im = buf((size_t)clips * T * num);
        bftObj_setResultType(o, 1);
        CHECK(bftObj_bftBatchDevice(o, x, clips, n, stride, re, NULL, stream));
        bftObj_setResultType(o, 0);
        CHECK(bftObj_bftBatchDevice(o, x, clips, n, stride, re, im, stream));
        bftObj_setResultType(o, 1);
        XXCCObj c = NULL;
        CHECK(xxccObj_new(&c, num));
        float *cc = buf((size_t)clips * T * 13);
        CHECK(afx_bftXxccBatchDevice(o, c, x, clips, n, stride, 13, NULL, re, cc, stream));
        CHECK(afx_bftXxccBatchDevice(o, c, x, clips, n, stride, 13, NULL, NULL, cc, stream));
        CHECK(xxccObj_xxccDevice(c, re, (long long)clips * T, 13, NULL, cc, stream));
        /* host-pointer batch and the legacy one-clip call */
        CHECK(bftObj_bftBatch(o, x, 1, n, re, NULL));
        bftObj_bft(o, x, n, re, NULL);
        xxccObj_free(c);
        bftObj_free(o);
        free(re);
        free(im);
        free(cc);
    }
    free(x);
    return 0;
}

/* the fused kernels of every transform size: n_fft 512 (k_stft_band_512), 1024, 2048, 4096 */
static int bft_paths(void *stream) {
    for (int r2 = 9; r2 <= 12; r2++)
        if (bft_paths_at(stream, r2)) return 1;
    return 0;
}

static int stft_paths_at(void *stream, int r2) {
    const int clips = 4, n = 3 * (1 << r2) + 808, hop = (1 << r2) / 4, N = 1 << r2;
    int slide = hop;
    float *x = buf((size_t)clips * n);
    STFTObj s = NULL;
    CHECK(stftObj_new(&s, r2, NULL, &slide, NULL));
    const int T = stftObj_calTimeLength(s, n);
    float *re = buf((size_t)clips * T * N), *im = buf((size_t)clips * T * N);
    CHECK(stftObj_stftBatchDevice(s, x, clips, n, n, re, im, stream));
    const long long outLen = (long long)(T - 1) * hop + N;
    float *y = buf((size_t)clips * outLen);
    CHECK(stftObj_istftBatchDevice(s, re, im, clips, T, 0, y, outLen, stream));
    CHECK(stftObj_istftBatchDevice(s, re, im, clips, T, 1, y, outLen, stream));
    stftObj_free(s);
    free(y);
    /* spectrogram object (mel) and cepstrogram */
    SpectrogramObj sp = NULL;
    CHECK(spectrogramObj_newMel(&sp, 64, 16000, r2, NULL));
    const int Ts = spectrogramObj_calTimeLength(sp, n);
    float *m = buf((size_t)clips * Ts * 64);
    CHECK(spectrogramObj_spectrogramBatchDevice(sp, x, clips, n, n, m, stream));
    spectrogramObj_free(sp);
    free(m);
    for (int cr2 = 10; cr2 <= 11 && r2 == 10; cr2++) {
        CepstrogramObj ce = NULL;
        int cslide = (1 << cr2) / 4;
        CHECK(cepstrogramObj_new(&ce, cr2, NULL, &cslide));
        const int Tc = cepstrogramObj_calTimeLength(ce, n), F = (1 << cr2) / 2 + 1;
        float *o1 = buf((size_t)clips * Tc * F), *o2 = buf((size_t)clips * Tc * F), *o3 = buf((size_t)clips * Tc * F);
        CHECK(cepstrogramObj_cepstrogramBatchDevice(ce, 4, x, clips, n, n, o1, o2, o3, stream));
        CHECK(cepstrogramObj_cepstrogramBatchDevice(ce, 4, x, clips, n, n, o1, NULL, NULL, stream));
        cepstrogramObj_free(ce);
        free(o1);
        free(o2);
        free(o3);
    }
    /* reassignment */
    ReassignObj r = NULL;
    int sr = 16000;
    CHECK(reassignObj_new(&r, r2, &sr, NULL, &slide, NULL, NULL, NULL, NULL));
    const int Tr = reassignObj_calTimeLength(r, n), F = N / 2 + 1;
    float *a1 = buf((size_t)clips * Tr * F), *a2 = buf((size_t)clips * Tr * F), *b1 = buf((size_t)clips * Tr * F),
          *b2 = buf((size_t)clips * Tr * F);
    CHECK(reassignObj_reassignBatchDevice(r, x, clips, n, n, a1, a2, b1, b2, stream));
    CHECK(reassignObj_reassignBatchDevice(r, x, clips, n, n, a1, a2, NULL, NULL, stream));
    reassignObj_free(r);
    free(a1);
    free(a2);
    free(b1);
    free(b2);
    free(re);
    free(im);
    free(x);
    return 0;
}

/* the STFT object / spectrogram / reassignment at the sizes whose spectrum comes from the bank kernels' transforms (512, 1024, 4096) and at 2048 */
static int stft_paths(void *stream) {
    for (int r2 = 9; r2 <= 12; r2++)
        if (stft_paths_at(stream, r2)) return 1;
    return 0;
}

static int wavelet_paths(void *stream, int r2, int chunks, int num) {
    const long long L = 1LL << r2;
    int sr = 44100, pad = 1;
    float lo = 32.703f;
    float *x = buf((size_t)chunks * L);
    float *re = buf((size_t)chunks * num * L), *im = buf((size_t)chunks * num * L);
    for (int padding = 0; padding <= (r2 <= 16 ? 1 : 0); padding++) {
        CWTObj w = NULL;
        pad = padding;
        CHECK(cwtObj_new(&w, num, r2, &sr, &lo, NULL, NULL, NULL, NULL, NULL, NULL, &pad));
        CHECK(cwtObj_cwtBatchDevice(w, x, chunks, L, re, im, stream));
        cwtObj_enableDet(w, 1);
        CHECK(cwtObj_cwtDetBatchDevice(w, x, chunks, L, re, im, stream));
        cwtObj_cwt(w, x, re, im);
        cwtObj_free(w);
    }
    if (r2 <= 13) {
        PWTObj p = NULL;
        CHECK(pwtObj_new(&p, num, r2, &sr, &lo, NULL, NULL, NULL, NULL, NULL, NULL));
        CHECK(pwtObj_pwtBatchDevice(p, x, chunks, L, re, im, stream));
        pwtObj_free(p);
        WSSTObj s = NULL;
        CHECK(wsstObj_new(&s, num, r2, &sr, &lo, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL));
        float *re2 = buf((size_t)chunks * num * L), *im2 = buf((size_t)chunks * num * L);
        CHECK(wsstObj_wsstBatchDevice(s, x, chunks, L, re, im, re2, im2, stream));
        CHECK(wsstObj_wsstBatchDevice(s, x, chunks, L, re, im, NULL, NULL, stream));
        wsstObj_free(s);
        free(re2);
        free(im2);
    }
    free(x);
    free(re);
    free(im);
    return 0;
}

int main(void) {
    void *stream = malloc(8); /* an opaque stream handle: the stand-in device layer never dereferences it */
    if (!stream) return 2;
    if (bft_paths(stream)) return 1;
    if (stft_paths(stream)) return 1;
    if (wavelet_paths(stream, 12, 5, 84)) return 1;   /* the in-LDS transform (L <= 8192), padded and not */
    if (wavelet_paths(stream, 16, 7, 84)) return 1;   /* the BASELINE cfg 4 plan: four-step + narrow-band, chunk groups */
    free(stream);
    printf("OK\n");
    return 0;
}
