/* One short clip through the CQT + chroma call on the fully emulated library (tests/emu), for ThreadSanitizer: the lanes of
 * the emulated kernels are host threads that meet only at the kernels' own cross-lane operations and LDS-ordering points,
 * so an LDS word written by one lane and read by another without such a point in between is a reported data race --
 * i.e. a missing wave_lds_order() / __syncthreads() in the kernel.  AFX_CQT_F32 selects
 * the kernels.  Exit status 0 and no ThreadSanitizer report = pass. */
#include <stdio.h>
#include <stdlib.h>

#include "afx_batch.h"
#include "cqt_algorithm.h"

int main(void) {
    CQTObj o = NULL;
    int sr = 32000, bpo = 12;
    float minFre = 32.703f;
    if (cqtObj_newWith(&o, 84, &sr, &minFre, &bpo, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL)) return 1;
    const int n = 6000, T = cqtObj_calTimeLength(o, n);
    float *x = (float *)malloc(sizeof(float) * n), *re = (float *)calloc((size_t)T * 84, 4), *im = (float *)calloc((size_t)T * 84, 4),
          *ch = (float *)calloc((size_t)T * 12, 4);
    void *stream = malloc(8);
    if (!x || !re || !im || !ch || !stream) return 2;
    for (int i = 0; i < n; i++) x[i] = (float)((i * 2654435761u) % 1000) * 1e-3f - 0.5f;
    if (cqtObj_cqtChromaBatchDevice(o, x, 1, n, n, re, im, NULL, NULL, NULL, ch, stream)) return 3;
    double s = 0;
    for (int i = 0; i < T * 12; i++) s += ch[i];
    printf("chroma sum %.6f\nOK\n", s);
    cqtObj_free(o);
    free(x), free(re), free(im), free(ch), free(stream);
    return 0;
}
