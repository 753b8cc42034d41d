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

/* ---- LDS-bank-aware lane assignment ------------------------------------------
 * In the kernel every lane reads the pairs P[start_lane + 2t .. +1], t = 0,1,2,...
 * with one ds_read_b64 per pair (start_lane even); ds_read_b64 sees 64 dword
 * banks, so a 32-lane half-wave is conflict-free exactly when its 32 values of
 * start/2 are distinct mod 32.  A row may start up to `d` bins early (leading
 * zero weights) to land on a free residue.  Rows are matched to the 64 (half, residue) slots so that the
 * longest padded row, max(len + d), is as short as possible: bottleneck
 * assignment by binary search over the bound + Kuhn's augmenting-path matching
 * (64 x 64, plan time only). */
static int shift_for(const RowBand *r, int residue) {
    int d = (r->start - 2 * residue) % 64;
    if (d < 0) d += 64;
    return d; /* new start = start - d is even and (start - d)/2 is congruent to residue mod 32 */
}

static int try_augment(int row, int n, const RowBand *rows, int bound, int *slotOwner, char *seen) {
    (void)n;
    for (int slot = 0; slot < 64; slot++) {
        if (seen[slot]) continue;
        const int d = shift_for(&rows[row], slot & 31);
        if (d > rows[row].start || rows[row].len + d > bound) continue;
        seen[slot] = 1;
        if (slotOwner[slot] < 0 ||
            try_augment(slotOwner[slot], n, rows, bound, slotOwner, seen)) {
            slotOwner[slot] = row;
            return 1;
        }
    }
    return 0;
}

static int match_with_bound(const RowBand *rows, int n, int bound, int *slotOwner) {
    for (int s = 0; s < 64; s++) slotOwner[s] = -1;
    for (int i = 0; i < n; i++) {
        char seen[64];
        memset(seen, 0, sizeof(seen));
        if (!try_augment(i, n, rows, bound, slotOwner, seen)) return 0;
    }
    return 1;
}

/* assigns n (<= 64) rows to lanes; returns the padded tap count, fills
 * laneRow[64] (index into rows, -1 = idle lane) and laneShift[64] */
static int assign_lanes(const RowBand *rows, int n, int *laneRow, int *laneShift) {
    int slotOwner[64];
    int lo = 1, hi = 1;
    for (int i = 0; i < n; i++) {
        if (rows[i].len > lo) lo = rows[i].len;
    }
    hi = lo + 63;
    if (!match_with_bound(rows, n, hi, slotOwner)) {
        /* cannot happen (a shift of < 32 always exists unless start is tiny and
         * residues clash); fall back to identity placement, conflicts allowed */
        for (int l = 0; l < 64; l++) {
            laneRow[l] = l < n ? l : -1;
            laneShift[l] = 0;
        }
        return lo;
    }
    while (lo < hi) {
        const int mid = (lo + hi) / 2;
        if (match_with_bound(rows, n, mid, slotOwner)) hi = mid; else lo = mid + 1;
    }
    match_with_bound(rows, n, lo, slotOwner);
    for (int slot = 0; slot < 64; slot++) { /* slot = 32*half + residue = lane */
        laneRow[slot] = slotOwner[slot];
        laneShift[slot] = slotOwner[slot] >= 0 ? shift_for(&rows[slotOwner[slot]], slot & 31) : 0;
    }
    return lo;
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

    /* the 64 longest rows form set A, the rest set B: every lane gets one of each */
    const int nA = num < 64 ? num : 64;
    const int nB = num - nA;
    int laneRowA[64], laneShiftA[64], laneRowB[64], laneShiftB[64];
    p->num = num;
    p->tapsA = assign_lanes(rb, nA, laneRowA, laneShiftA);
    p->tapsB = 1;
    for (int l = 0; l < 64; l++) {
        laneRowB[l] = -1;
        laneShiftB[l] = 0;
    }
    if (nB > 0) p->tapsB = assign_lanes(rb + nA, nB, laneRowB, laneShiftB);
    p->wA = (float *)calloc((size_t)p->tapsA * 64, sizeof(float));
    p->wB = (float *)calloc((size_t)p->tapsB * 64, sizeof(float));
    if (!p->wA || !p->wB) {
        free(rb);
        afx_bandplan_free(p);
        return 1;
    }
    for (int l = 0; l < 64; l++) {
        p->rowA[l] = p->rowB[l] = -1;
        /* idle lanes still execute the reads: give them their own residue so they
         * do not collide with a working lane of the same half */
        p->startA[l] = 2 * (l & 31);
        p->startB[l] = 2 * (l & 31);
        if (laneRowA[l] >= 0) {
            const RowBand *r = &rb[laneRowA[l]];
            const int d = laneShiftA[l];
            for (int t = 0; t < r->len; t++)
                p->wA[(size_t)(t + d) * 64 + l] = bank[(size_t)r->row * F + r->start + t];
            p->startA[l] = r->start - d;
            p->rowA[l] = r->row;
        }
        if (laneRowB[l] >= 0) {
            const RowBand *r = &rb[nA + laneRowB[l]];
            const int d = laneShiftB[l];
            for (int t = 0; t < r->len; t++)
                p->wB[(size_t)(t + d) * 64 + l] = bank[(size_t)r->row * F + r->start + t];
            p->startB[l] = r->start - d;
            p->rowB[l] = r->row;
        }
    }
    free(rb);
    return 0;
}

/* ---- split plans ---------------------------------------------------------------------------
 * Banks whose rows are longer than the compiled tap variants (mel-40 / mel-64, bark and erb banks,
 * higher sample rates) still have ~2 non-zeros per bin in total: their rows are cut into
 * segments of at most tapsA (A slots) or tapsB (B slots) bins, every lane gets one A and one B
 * segment, and the kernel adds a row's segment results in ascending bin order.  Segment starts are
 * matched to LDS-bank residues like whole rows; because a segment may have to start a few bins
 * early, the payload per slot is reduced until the padded segments fit the variant.
 * rowCap = floats of the kernel's zero-padded power row: no slot may read beyond it.
 * Returns 0 and fills *p (split = 1) when the bank fits, 1 otherwise. */
int afx_bandplan_build_split(const float *bank, int num, int F, int tapsA, int tapsB, int rowCap,
                             AfxBandPlan *p) {
    memset(p, 0, sizeof(*p));
    if (!bank || num < 1 || num > 128 || tapsA < 4 || tapsB < 4) return 1;
    RowBand *rb = (RowBand *)calloc((size_t)num, sizeof(RowBand));
    RowBand *segA = (RowBand *)calloc(64, sizeof(RowBand)), *segB = (RowBand *)calloc(64, sizeof(RowBand));
    if (!rb || !segA || !segB) {
        free(rb);
        free(segA);
        free(segB);
        return 1;
    }
    for (int m = 0; m < num; m++) {
        const float *row = bank + (size_t)m * F;
        int first = -1, last = -1;
        for (int k = 0; k < F; k++)
            if (row[k] != 0.f) {
                if (first < 0) first = k;
                last = k;
            }
        rb[m].row = m;
        rb[m].start = first < 0 ? 0 : first;
        rb[m].len = first < 0 ? 0 : last - first + 1;
    }
    int laneA[64], shiftA[64], laneB[64], shiftB[64];
    /* segment lists per row: type (0 A, 1 B) and index into segA / segB, ascending bins */
    unsigned char segType[128][4], segOf[128][4], segCnt[128];
    int payA = tapsA, payB = tapsB, ok = 0, nA = 0, nB = 0;
    for (int iter = 0; iter < 40 && !ok && payA >= 8 && payB >= 0; iter++) {
        nA = nB = 0;
        int feasible = 1;
        for (int m = 0; m < num && feasible; m++) {
            int rem = rb[m].len, pos = rb[m].start;
            segCnt[m] = 0;
            while (rem > 0) {
                if (segCnt[m] == 4) {
                    feasible = 0;
                    break;
                }
                const int c = segCnt[m]++;
                if (rem <= payB && nB < 64) {
                    segB[nB].row = m, segB[nB].start = pos, segB[nB].len = rem;
                    segType[m][c] = 1, segOf[m][c] = (unsigned char)nB++;
                    rem = 0;
                } else {
                    if (nA == 64) {
                        feasible = 0;
                        break;
                    }
                    const int take = rem < payA ? rem : payA;
                    segA[nA].row = m, segA[nA].start = pos, segA[nA].len = take;
                    segType[m][c] = 0, segOf[m][c] = (unsigned char)nA++;
                    pos += take, rem -= take;
                }
            }
        }
        if (!feasible) break; /* smaller payloads only make more segments */
        int needA = nA ? assign_lanes(segA, nA, laneA, shiftA) : 1;
        int needB = nB ? assign_lanes(segB, nB, laneB, shiftB) : 1;
        /* the kernel reads the power row in pairs: a placement that left a start odd (the
         * matcher's identity fallback) does not qualify */
        for (int l = 0; l < 64; l++) {
            if (nA && laneA[l] >= 0 && ((segA[laneA[l]].start - shiftA[l]) & 1)) needA = tapsA + 1;
            if (nB && laneB[l] >= 0 && ((segB[laneB[l]].start - shiftB[l]) & 1)) needB = tapsB + 1;
            /* a short segment near the top of the spectrum must not read past the row's zero pad:
             * start it 64 bins (one full residue cycle) earlier while its taps allow */
            if (nA && laneA[l] >= 0) {
                const RowBand *g = &segA[laneA[l]];
                while (g->start - shiftA[l] + tapsA > rowCap && shiftA[l] + 64 + g->len <= tapsA &&
                       g->start - shiftA[l] - 64 >= 0)
                    shiftA[l] += 64;
                if (g->start - shiftA[l] + tapsA > rowCap) needA = tapsA + 1;
            }
            if (nB && laneB[l] >= 0) {
                const RowBand *g = &segB[laneB[l]];
                while (g->start - shiftB[l] + tapsB > rowCap && shiftB[l] + 64 + g->len <= tapsB &&
                       g->start - shiftB[l] - 64 >= 0)
                    shiftB[l] += 64;
                if (g->start - shiftB[l] + tapsB > rowCap) needB = tapsB + 1;
            }
        }
        if (needA <= tapsA && needB <= tapsB) ok = 1;
        else {
            if (needA > tapsA) payA -= 2;
            if (needB > tapsB) payB -= 2;
        }
    }
    if (ok) {
        p->num = num;
        p->split = 1;
        p->tapsA = tapsA;
        p->tapsB = tapsB;
        p->wA = (float *)calloc((size_t)tapsA * 64, sizeof(float));
        p->wB = (float *)calloc((size_t)tapsB * 64, sizeof(float));
        if (!p->wA || !p->wB) {
            afx_bandplan_free(p);
            ok = 0;
        }
    }
    if (ok) {
        int slotOfA[64], slotOfB[64]; /* segment index -> lane */
        for (int l = 0; l < 64; l++) {
            p->rowA[l] = p->rowB[l] = -1;
            p->startA[l] = 2 * (l & 31);
            p->startB[l] = 2 * (l & 31);
            if (nA && laneA[l] >= 0) {
                const RowBand *g = &segA[laneA[l]];
                for (int t = 0; t < g->len; t++)
                    p->wA[(size_t)(t + shiftA[l]) * 64 + l] = bank[(size_t)g->row * F + g->start + t];
                p->startA[l] = g->start - shiftA[l];
                slotOfA[laneA[l]] = l;
            }
            if (nB && laneB[l] >= 0) {
                const RowBand *g = &segB[laneB[l]];
                for (int t = 0; t < g->len; t++)
                    p->wB[(size_t)(t + shiftB[l]) * 64 + l] = bank[(size_t)g->row * F + g->start + t];
                p->startB[l] = g->start - shiftB[l];
                slotOfB[laneB[l]] = l;
            }
        }
        for (int m = 0; m < 128; m++) {
            unsigned packed = 0;
            for (int c = 0; c < 4; c++) {
                unsigned slot = 128;
                if (m < num && c < segCnt[m])
                    slot = segType[m][c] ? 64u + (unsigned)slotOfB[segOf[m][c]] : (unsigned)slotOfA[segOf[m][c]];
                packed |= slot << (8 * c);
            }
            p->segIdx[m] = packed;
        }
    }
    free(rb);
    free(segA);
    free(segB);
    return ok ? 0 : 1;
}
