/* Host logic of the CQT object under AddressSanitizer / UBSan, device layer replaced by tests/hoststub/gen_stub.py's
 * stand-in: every batched / one-clip entry point, pass splitting (AFX_CQT_CHUNK), the fused-launch glue
 * (AFX_CQT_FUSED=1|2), chroma with a changing class count (the folding matrix is rebuilt), free.  Buffers handed
 * to the library have exactly the documented sizes; the stand-in kernels read / write every range they are given.
 * Exit status 0 and no sanitizer report = pass. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "afx_batch.h"
#include "cqt_algorithm.h"

#define CHECK(x)                                                        \
    do {                                                                \
        int _s = (x);                                                   \
        if (_s != 0) {                                                  \
            fprintf(stderr, "%s -> %d (line %d)\n", #x, _s, __LINE__);  \
            return 1;                                                   \
        }                                                               \
    } while (0)

static int run(int samplate, int num, int n, int clips, long long stride) {
    CQTObj o = NULL;
    float minFre = 32.703f;
    int bpo = 12;
    CHECK(cqtObj_newWith(&o, num, &samplate, &minFre, &bpo, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL));
    const int T = cqtObj_calTimeLength(o, n);
    float *x = (float *)calloc((size_t)clips * stride, sizeof(float));
    float *re = (float *)malloc(sizeof(float) * (size_t)clips * T * num);
    float *im = (float *)malloc(sizeof(float) * (size_t)clips * T * num);
    float *ch12 = (float *)malloc(sizeof(float) * (size_t)clips * T * 12);
    float *ch6 = (float *)malloc(sizeof(float) * (size_t)clips * T * 6);
    if (!x || !re || !im || !ch12 || !ch6) return 2;
    for (size_t i = 0; i < (size_t)clips * stride; i++) x[i] = (float)((i * 2654435761u) % 1000) * 1e-3f - 0.5f;
    void *stream = malloc(8); /* an opaque stream handle: the stand-in device layer never dereferences it */
    if (!stream) return 3;
    CHECK(cqtObj_cqtBatchDevice(o, x, clips, n, stride, re, im, stream));
    CHECK(cqtObj_chromaBatchDevice(o, NULL, NULL, NULL, re, im, (long long)clips * T, ch12, stream));
    int six = 6, twelve = 12;
    CHECK(cqtObj_chromaBatchDevice(o, &six, NULL, NULL, re, im, (long long)clips * T, ch6, stream));
    CHECK(cqtObj_cqtChromaBatchDevice(o, x, clips, n, stride, re, im, &twelve, NULL, NULL, ch12, stream));
    CHECK(cqtObj_cqtChromaBatchDevice(o, x, clips, n, stride, re, im, &six, NULL, NULL, ch6, stream));
    CHECK(cqtObj_cqtChromaBatchDevice(o, x, clips, n, stride, re, im, NULL, NULL, NULL, ch12, stream));
    /* host-pointer entry points */
    CHECK(cqtObj_cqtBatch(o, x, 1, n, re, im));
    cqtObj_cqt(o, x, n, re, im);
    cqtObj_chroma(o, NULL, NULL, NULL, re, im, ch12);
    cqtObj_chroma(o, &six, NULL, NULL, re, im, ch6);
    free(stream);
    cqtObj_free(o);
    free(x);
    free(re);
    free(im);
    free(ch12);
    free(ch6);
    return 0;
}

/* present only in the launch-audit build (tests/hoststub/fake_hip.cpp) */
extern int fakehip_races(void) __attribute__((weak));
extern int fakehip_violations(void) __attribute__((weak));

int main(void) {
    /* default plan (84 bins: the f16 / fused paths), an unaligned row stride, a short clip, a 48-bin plan */
    if (run(44100, 84, 30000, 5, 30000)) return 1;
    if (run(44100, 84, 29987, 3, 30077)) return 1;
    if (run(44100, 84, 700, 2, 700)) return 1;
    if (run(16000, 48, 9000, 4, 9000)) return 1;
    if (fakehip_races && (fakehip_races() || fakehip_violations())) {
        fprintf(stderr, "%d unordered conflicting launches, %d launch configurations outside the HIP limits\n", fakehip_races(),
                fakehip_violations());
        return 1;
    }
    printf("OK\n");
    return 0;
}
