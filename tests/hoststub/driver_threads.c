/* Objects are built and used concurrently on different threads (the reference's rule: one object by one thread at
 * a time, distinct objects independent).  Under ThreadSanitizer with the stand-in device layer: constructors of
 * every family at once -- Kaiser-window BFT / STFT objects beside CQT objects, whose resampler table is a Kaiser
 * window with another beta (afx_window.c once kept that beta in a temporarily overwritten global) -- then one
 * compute call each, then free.  Each thread also checks its own window against a single-threaded reference. */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "afx_batch.h"

float *afx_window_fft(WindowType type, int length);  /* afx_host.h */

static float *g_refKaiser; /* computed before the threads start */
static int g_fail;

static void *worker(void *arg) {
    const int id = (int)(size_t)arg;
    void *stream = malloc(8);
    float *x = (float *)calloc(16000, sizeof(float));
    for (int rep = 0; rep < 6; rep++) {
        if (id % 2 == 0) {
            WindowType wt = Window_Kaiser;
            int sr = 16000, slide = 512;
            BFTObj o = NULL;
            if (bftObj_new(&o, 64, 11, &sr, NULL, NULL, NULL, &wt, &slide, NULL, NULL, NULL, NULL, NULL, NULL)) g_fail = 1;
            float *w = afx_window_fft(Window_Kaiser, 2048);
            if (!w || memcmp(w, g_refKaiser, sizeof(float) * 2048)) g_fail = 1;
            free(w);
            const int T = bftObj_calTimeLength(o, 16000);
            float *re = (float *)malloc(sizeof(float) * (size_t)T * 64);
            bftObj_setResultType(o, 1);
            if (bftObj_bftBatchDevice(o, x, 1, 16000, 16000, re, NULL, stream)) g_fail = 1;
            free(re);
            bftObj_free(o);
        } else {
            CQTObj q = NULL;
            if (cqtObj_new(&q, 84, 44100, 32.703f, NULL)) g_fail = 1;
            const int T = cqtObj_calTimeLength(q, 16000);
            float *re = (float *)malloc(sizeof(float) * (size_t)T * 84), *im = (float *)malloc(sizeof(float) * (size_t)T * 84);
            if (cqtObj_cqtBatchDevice(q, x, 1, 16000, 16000, re, im, stream)) g_fail = 1;
            free(re);
            free(im);
            cqtObj_free(q);
        }
    }
    free(x);
    free(stream);
    return NULL;
}

int main(void) {
    g_refKaiser = afx_window_fft(Window_Kaiser, 2048);
    pthread_t th[6];
    for (size_t i = 0; i < 6; i++) pthread_create(&th[i], NULL, worker, (void *)i);
    for (int i = 0; i < 6; i++) pthread_join(th[i], NULL);
    free(g_refKaiser);
    if (g_fail) {
        fprintf(stderr, "a thread saw a wrong window or a failing call\n");
        return 1;
    }
    printf("OK\n");
    return 0;
}
