/* afx_bandplan.c -- turns a filter-bank matrix into the per-lane banded view
 * the fused kernel consumes (AfxBandPlan, afx_device.h).
 *
 * Triangular / window-shaped auditory banks are banded: each row's non-zeros
 * sit in one short contiguous bin range (mel-128 @ n_fft 2048: 2025 non-zeros
 * of 131 200, <= 2 per column).  The reference multiplies the full dense
 * matrix (src/vector/flux_vector.c:55-86); zero weights contribute exact
 * zeros, so summing only the band in ascending bin order gives the same value.
 * Rows are sorted by band length and dealt to the 64 lanes: lane i gets the
 * i-th longest row as "A" and, when num > 64, one of the short rows as "B"
 * (longest A with shortest B) to even the work out.
 */
#include <stdlib.h>
#include <string.h>

#include "afx_device.h"
#include "afx_host.h"

typedef struct {
    int row, start, len;
} RowBand;

static int cmp_len_desc(const void *a, const void *b) {
    const RowBand *x = (const RowBand *)a, *y = (const RowBand *)b;
    if (x->len != y->len) return y->len - x->len;
    return x->row - y->row;
}

void afx_bandplan_free(AfxBandPlan *p) {
    if (!p) return;
    free(p->wA);
    free(p->wB);
    p->wA = p->wB = NULL;
}

/* returns 0 and fills *p when the bank fits the banded scheme, 1 otherwise */
int afx_bandplan_build(const float *bank, int num, int F, AfxBandPlan *p) {
    memset(p, 0, sizeof(*p));
    if (!bank || num < 1 || num > 128) return 1;
    RowBand *rb = (RowBand *)calloc((size_t)num, sizeof(RowBand));
    if (!rb) return 1;
    for (int m = 0; m < num; m++) {
        const float *row = bank + (size_t)m * F;
        int first = -1, last = -1;
        for (int k = 0; k < F; k++) {
            if (row[k] != 0.f) {
                if (first < 0) first = k;
                last = k;
            }
        }
        rb[m].row = m;
        rb[m].start = first < 0 ? 0 : first;
        rb[m].len = first < 0 ? 0 : last - first + 1;
    }
    qsort(rb, (size_t)num, sizeof(RowBand), cmp_len_desc);

    const int nA = num < 64 ? num : 64;
    const int nB = num - nA;
    p->num = num;
    for (int i = 0; i < 64; i++) {
        p->rowA[i] = p->rowB[i] = -1;
    }
    for (int i = 0; i < nA; i++) {
        p->rowA[i] = i; /* index into rb for now */
        if (rb[i].len > p->tapsA) p->tapsA = rb[i].len;
    }
    for (int i = 0; i < nB; i++) {
        const int j = num - 1 - i; /* shortest rows first */
        p->rowB[i] = j;
        if (rb[j].len > p->tapsB) p->tapsB = rb[j].len;
    }
    if (p->tapsA < 1) p->tapsA = 1;
    if (p->tapsB < 1) p->tapsB = 1;
    p->wA = (float *)calloc((size_t)p->tapsA * 64, sizeof(float));
    p->wB = (float *)calloc((size_t)p->tapsB * 64, sizeof(float));
    if (!p->wA || !p->wB) {
        free(rb);
        afx_bandplan_free(p);
        return 1;
    }
    for (int i = 0; i < 64; i++) {
        if (p->rowA[i] >= 0) {
            const RowBand *r = &rb[p->rowA[i]];
            for (int t = 0; t < r->len; t++)
                p->wA[(size_t)t * 64 + i] = bank[(size_t)r->row * F + r->start + t];
            p->startA[i] = r->start;
            p->rowA[i] = r->row;
        }
        if (p->rowB[i] >= 0) {
            const RowBand *r = &rb[p->rowB[i]];
            for (int t = 0; t < r->len; t++)
                p->wB[(size_t)t * 64 + i] = bank[(size_t)r->row * F + r->start + t];
            p->startB[i] = r->start;
            p->rowB[i] = r->row;
        }
    }
    free(rb);
    return 0;
}
