/* Size arithmetic of the host objects at batch sizes that fill (and overfill) a 288 GB device.  Built with the
 * stand-in device layer in its DRY mode (tests/hoststub/gen_stub.py, -DAFX_STUB_DRY: "device" pointers are address
 * ranges without memory, launchers do nothing) under clang's UBSan + integer checks (signed / unsigned overflow,
 * implicit truncation and sign change): every device-pointer entry point is called at the BASELINE sizes, at 20x and
 * at sizes whose element counts pass 2^31 and 2^32.  A call returns 0 or a negative status (a refusal is fine);
 * what must not happen is a wrapped product on the way to an allocation size, a grid size or a pointer offset --
 * the GPU tests cannot see those, they run at one device's share.
 * Exit status 0 and no "runtime error" line = pass. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "afx_batch.h"
#include "cqt_algorithm.h"

/* present only in the launch-audit build (tests/hoststub/fake_hip.cpp): launch configurations that break a HIP limit */
extern int fakehip_violations(void) __attribute__((weak));
extern void fakehip_report(void) __attribute__((weak));
extern int fakehip_launched(const char *substr) __attribute__((weak));

static float *fake(unsigned long long bytes) {
    static unsigned long long next = 1ull << 46;
    float *p = (float *)(uintptr_t)next;
    next += (bytes + 4095) & ~4095ull;
    return p;
}

static int g_calls, g_refused, g_first;
/* g_first: the call is at the BASELINE size (first entry of its size table) and must succeed */
static void note(const char *what, long long size, int st) {
    g_calls++;
    if (st > 0 || (st != 0 && g_first)) {
        fprintf(stderr, "%s (%lld): status %d\n", what, size, st);
        exit(1);
    }
    if (st < 0) {
        g_refused++;
        printf("  %s at %lld: refused (%d)\n", what, size, st);
    }
}

#define NEW(x)                                                          \
    do {                                                                \
        int _s = (x);                                                   \
        if (_s != 0) {                                                  \
            fprintf(stderr, "%s -> %d (line %d)\n", #x, _s, __LINE__);  \
            return 1;                                                   \
        }                                                               \
    } while (0)

static const int BATCHES[] = {1000, 20000, 250000, 2000000};

static int framed(void *stream) {
    const int n = 480000, r2 = 11, hop = 512;
    int sr = 16000, slide = hop, one = 1;
    float lo = 0.f, hi = 8000.f;
    for (int variant = 0; variant < 3; variant++) {
        /* 0 mel-128 (fused kernel), 1 gammatone-128 (dense route, chunked scratch), 2 mel-128 + temporal + reassigned */
        BFTObj o = NULL;
        XXCCObj c = NULL;
        SpectralFilterBankScaleType scale = variant == 1 ? SpectralFilterBankScale_Erb : SpectralFilterBankScale_Mel;
        SpectralFilterBankStyleType style = variant == 1 ? SpectralFilterBankStyle_Gammatone : SpectralFilterBankStyle_Slaney;
        SpectralDataType dt = SpectralData_Power;
        NEW(bftObj_new(&o, 128, r2, &sr, &lo, &hi, NULL, NULL, &slide, &scale, &style, NULL, &dt, variant == 2 ? &one : NULL,
                       variant == 2 ? &one : NULL));
        NEW(xxccObj_new(&c, 128));
        const long long T = bftObj_calTimeLength(o, n);
        for (size_t k = 0; k < sizeof BATCHES / sizeof *BATCHES; k++) {
            const int b = BATCHES[k];
            g_first = (k == 0 && variant != 2); /* (the reassigned route keeps 6 planes of [frames, F] scratch) */
            const unsigned long long out = (unsigned long long)b * T * 128 * 4;
            float *x = fake((unsigned long long)b * n * 4), *re = fake(out), *im = fake(out), *cc = fake((unsigned long long)b * T * 13 * 4);
            bftObj_setResultType(o, 1);
            note("bftObj_bftBatchDevice", b, bftObj_bftBatchDevice(o, x, b, n, n, re, NULL, stream));
            bftObj_setResultType(o, 0);
            note("bftObj_bftBatchDevice complex", b, bftObj_bftBatchDevice(o, x, b, n, n, re, im, stream));
            bftObj_setResultType(o, 1);
            note("afx_bftXxccBatchDevice", b, afx_bftXxccBatchDevice(o, c, x, b, n, n, 13, NULL, re, cc, stream));
            note("afx_bftXxccBatchDevice, cepstra only", b, afx_bftXxccBatchDevice(o, c, x, b, n, n, 13, NULL, NULL, cc, stream));
            note("xxccObj_xxccDevice", b, xxccObj_xxccDevice(c, re, (long long)b * T, 13, NULL, cc, stream));
        }
        xxccObj_free(c);
        bftObj_free(o);
    }
    /* STFT / inverse STFT, spectrogram object, cepstrogram, reassignment */
    STFTObj s = NULL;
    NEW(stftObj_new(&s, r2, NULL, &slide, NULL));
    SpectrogramObj sp = NULL;
    NEW(spectrogramObj_newMel(&sp, 128, sr, r2, NULL));
    CepstrogramObj ce = NULL;
    NEW(cepstrogramObj_new(&ce, r2, NULL, &slide));
    ReassignObj ra = NULL;
    NEW(reassignObj_new(&ra, r2, &sr, NULL, &slide, NULL, NULL, NULL, NULL));
    for (size_t k = 0; k < sizeof BATCHES / sizeof *BATCHES; k++) {
        const int b = BATCHES[k], N = 1 << r2, F = N / 2 + 1;
        g_first = (k == 0);
        const long long T = stftObj_calTimeLength(s, n);
        float *x = fake((unsigned long long)b * n * 4);
        float *re = fake((unsigned long long)b * T * N * 4), *im = fake((unsigned long long)b * T * N * 4);
        note("stftObj_stftBatchDevice", b, stftObj_stftBatchDevice(s, x, b, n, n, re, im, stream));
        const long long outLen = (T - 1) * hop + N;
        float *y = fake((unsigned long long)b * outLen * 4);
        note("stftObj_istftBatchDevice", b, stftObj_istftBatchDevice(s, re, im, b, (int)T, 0, y, outLen, stream));
        const long long Ts = spectrogramObj_calTimeLength(sp, n);
        note("spectrogramObj_spectrogramBatchDevice", b,
             spectrogramObj_spectrogramBatchDevice(sp, x, b, n, n, fake((unsigned long long)b * Ts * 128 * 4), stream));
        const long long Tc = cepstrogramObj_calTimeLength(ce, n);
        float *o1 = fake((unsigned long long)b * Tc * F * 4), *o2 = fake((unsigned long long)b * Tc * F * 4),
              *o3 = fake((unsigned long long)b * Tc * F * 4);
        note("cepstrogramObj_cepstrogramBatchDevice", b, cepstrogramObj_cepstrogramBatchDevice(ce, 4, x, b, n, n, o1, o2, o3, stream));
        const long long Tr = reassignObj_calTimeLength(ra, n);
        float *a1 = fake((unsigned long long)b * Tr * F * 4), *a2 = fake((unsigned long long)b * Tr * F * 4);
        note("reassignObj_reassignBatchDevice", b, reassignObj_reassignBatchDevice(ra, x, b, n, n, a1, a2, NULL, NULL, stream));
    }
    reassignObj_free(ra);
    cepstrogramObj_free(ce);
    spectrogramObj_free(sp);
    stftObj_free(s);
    return 0;
}

static int wavelets(void *stream) {
    static const int CHUNKS[] = {7000, 140000, 3000000};
    int sr = 44100, pad = 1;
    float lo = 32.703f;
    CWTObj w = NULL;
    NEW(cwtObj_new(&w, 84, 16, &sr, &lo, NULL, NULL, NULL, NULL, NULL, NULL, &pad)); /* BASELINE cfg 4 */
    cwtObj_enableDet(w, 1);
    PWTObj p = NULL;
    NEW(pwtObj_new(&p, 84, 13, &sr, &lo, NULL, NULL, NULL, NULL, NULL, NULL));
    WSSTObj ws = NULL;
    NEW(wsstObj_new(&ws, 84, 13, &sr, &lo, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL));
    for (size_t k = 0; k < sizeof CHUNKS / sizeof *CHUNKS; k++) {
        const int c = CHUNKS[k];
        g_first = (k == 0);
        long long L = 65536;
        float *x = fake((unsigned long long)c * L * 4);
        float *re = fake((unsigned long long)c * 84 * L * 4), *im = fake((unsigned long long)c * 84 * L * 4);
        note("cwtObj_cwtBatchDevice", c, cwtObj_cwtBatchDevice(w, x, c, L, re, im, stream));
        note("cwtObj_cwtDetBatchDevice", c, cwtObj_cwtDetBatchDevice(w, x, c, L, re, im, stream));
        L = 8192;
        note("pwtObj_pwtBatchDevice", c, pwtObj_pwtBatchDevice(p, x, c, L, re, im, stream));
        float *re2 = fake((unsigned long long)c * 84 * L * 4), *im2 = fake((unsigned long long)c * 84 * L * 4);
        note("wsstObj_wsstBatchDevice", c, wsstObj_wsstBatchDevice(ws, x, c, L, re, im, re2, im2, stream));
    }
    wsstObj_free(ws);
    pwtObj_free(p);
    cwtObj_free(w);
    return 0;
}

static int cqt(void *stream) {
    static const int CLIPS[] = {125, 2500, 40000, 400000};
    const int n = 1323000, num = 84; /* BASELINE cfg 5: 30 s at 44.1 kHz */
    int sr = 44100, bpo = 12;
    float minFre = 32.703f;
    CQTObj o = NULL;
    NEW(cqtObj_newWith(&o, num, &sr, &minFre, &bpo, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL));
    const long long T = cqtObj_calTimeLength(o, n);
    for (size_t k = 0; k < sizeof CLIPS / sizeof *CLIPS; k++) {
        const int b = CLIPS[k];
        g_first = (k == 0);
        float *x = fake((unsigned long long)b * n * 4);
        float *re = fake((unsigned long long)b * T * num * 4), *im = fake((unsigned long long)b * T * num * 4);
        float *ch = fake((unsigned long long)b * T * 12 * 4);
        note("cqtObj_cqtBatchDevice", b, cqtObj_cqtBatchDevice(o, x, b, n, n, re, im, stream));
        note("cqtObj_chromaBatchDevice", b, cqtObj_chromaBatchDevice(o, NULL, NULL, NULL, re, im, (long long)b * T, ch, stream));
        note("cqtObj_cqtChromaBatchDevice", b, cqtObj_cqtChromaBatchDevice(o, x, b, n, n, re, im, NULL, NULL, NULL, ch, stream));
    }
    cqtObj_free(o);
    return 0;
}

int main(void) {
    void *stream = malloc(8); /* an opaque stream handle: the stand-in device layer never dereferences it */
    if (!stream) return 2;
    if (framed(stream)) return 1;
    if (wavelets(stream)) return 1;
    if (cqt(stream)) return 1;
    free(stream);
    if (fakehip_report) fakehip_report();
    if (fakehip_violations && fakehip_violations()) {
        fprintf(stderr, "%d launch configurations outside the HIP limits\n", fakehip_violations());
        return 1;
    }
    /* AFX_AUDIT_EXPECT=<kernel name part>: the switch under test must have reached that kernel */
    const char *expect = getenv("AFX_AUDIT_EXPECT");
    if (expect && fakehip_launched && !fakehip_launched(expect)) {
        fprintf(stderr, "no launch of a kernel named *%s*\n", expect);
        return 1;
    }
    printf("%d calls, %d refused\nOK\n", g_calls, g_refused);
    return 0;
}
