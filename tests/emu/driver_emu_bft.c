/* The fused STFT -> filter-bank kernels on the emulated library for ThreadSanitizer: n_fft 512 (k_stft_band_512), n_fft 1024 (k_stft_band_1k), n_fft 4096
 * (k_stft_band_4k2), n_fft 2048 complex and real + MFCC (k_stft_mel_v2), a few frames each.  All three issue their DS
 * instructions by hand and rely on their issue order; the emulation makes each a rendezvous, so those cannot race here by
 * construction -- what ThreadSanitizer watches are the plain loads and stores around them (table fills, zero pads, segment
 * sums, staging) between explicit ordering points.  Exit 0 and no report = pass. */
#include <stdio.h>
#include <stdlib.h>

#include "afx_batch.h"

static int run(int r2, int num, int hop, int resultType, int withCc) {
    BFTObj o = NULL;
    XXCCObj c = NULL;
    int sr = 16000, slide = hop;
    float lo = 0.f, hi = 8000.f;
    SpectralFilterBankScaleType scale = SpectralFilterBankScale_Mel;
    SpectralDataType dt = SpectralData_Power;
    if (bftObj_new(&o, num, r2, &sr, &lo, &hi, NULL, NULL, &slide, &scale, NULL, NULL, &dt, NULL, NULL)) return 1;
    if (xxccObj_new(&c, num)) return 1;
    const int n = (1 << r2) + 37 * hop + 11, T = bftObj_calTimeLength(o, n), clips = 2;
    float *x = (float *)malloc(sizeof(float) * clips * n), *re = (float *)calloc((size_t)clips * T * num, 4),
          *im = (float *)calloc((size_t)clips * T * num, 4), *cc = (float *)calloc((size_t)clips * T * 13, 4);
    void *stream = malloc(8);
    if (!x || !re || !im || !cc || !stream) return 2;
    for (int i = 0; i < clips * n; i++) x[i] = (float)((i * 2654435761u) % 1000) * 1e-3f - 0.5f;
    bftObj_setResultType(o, resultType);
    if (bftObj_bftBatchDevice(o, x, clips, n, n, re, resultType ? NULL : im, stream)) return 3;
    if (withCc && afx_bftXxccBatchDevice(o, c, x, clips, n, n, 13, NULL, re, cc, stream)) return 4;
    double s = 0;
    for (int i = 0; i < clips * T * num; i++) s += re[i];
    printf("n_fft %d, %d bands, result type %d: %d frames, sum %.6g\n", 1 << r2, num, resultType, T, s);
    xxccObj_free(c);
    bftObj_free(o);
    free(x), free(re), free(im), free(cc), free(stream);
    return 0;
}

int main(void) {
    if (run(9, 40, 128, 1, 0)) return 1;
    if (run(9, 128, 100, 0, 0)) return 1;
    if (run(10, 80, 160, 1, 0)) return 1;
    if (run(12, 128, 1024, 1, 0)) return 1;
    if (run(11, 128, 512, 0, 0)) return 1;
    if (run(11, 128, 512, 1, 1)) return 1;
    printf("OK\n");
    return 0;
}
