// afx_reassign.hip -- time-frequency reassignment of an STFT (reference:
// src/reassign_algorithm.c:203-414, :611-832).
//
// Inputs are the three STFTs of the signal taken with the window h, its derivative dh and the
// time-weighted window t.h (computed by k_stft_generic, afx_stft.hip).  Per coefficient:
//   f' = f_j - Im(S_dh / S_h) sr / 2pi,   t' = t_i + Re(S_th / S_h) / sr     (:694-716)
//   below the power threshold the coordinate stays (t_i, f_j); clip to the grid   (:772-822)
//   target cell = rounded position of (t', f') on the (frame, bin) grid           (:312-320)
// k_reassign_index stores the two target indices (-1: dropped), k_reassign_order applies the
// reference's "order" iteration of the frequency index (:340-358); every coefficient, sign-flipped
// on odd bins (:378-381), is then added to its target cell.  A target cell collects sources from
// several frames and bins.  The reference adds them in its loop order (frames outer, bins inner,
// :360-414); so does this file, WITHOUT atomics: (target cell, source index) pairs are sorted by
// target with a stable LSD radix sort written here (4-bit digits, the bits in use only; ties keep
// the ascending source order), and one thread per target cell walks its run and sums sequentially
// -- the float32 additions happen in the reference's order, the result is deterministic run to run
// and bit-identical to the reference's accumulation whenever the transforms and target indices agree.
#include <hip/hip_runtime.h>
#include <cstring>

#include "afx_device.h"
#include "afx_hipcheck.h"

namespace {

__device__ __forceinline__ void cdiv(float r1, float i1, float r2, float i2, float &r3, float &i3) {
    const float value = r2 * r2 + i2 * i2;  // __complexDiv (vector/flux_complex.c:771-780)
    r3 = (r1 * r2 + i1 * i2) / value;
    i3 = (i1 * r2 - r1 * i2) / value;
}

__global__ void k_reassign_index(AfxReassignArgs a) {
    const int F = a.F, T = a.timeLength;
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)T * F) return;
    const int i = (int)(e / F), j = (int)(e - (long long)i * F);
    const long long g = (long long)blockIdx.y * T * F + e;
    const float hr = a.hRe[g], hi = a.hIm[g];
    const float power = hr * hr + hi * hi;
    const bool strong = power >= a.thresh * a.thresh;
    const float tI = ((float)i * a.hop) / a.samplate;          // timeArr[i]  (:560-567)
    const float tmax = ((float)(T - 1) * a.hop) / a.samplate;
    const float fJ = a.freArr[j], fmax = a.freArr[F - 1];
    float reF = fJ, reT = tI;
    if (a.doFre) {
        float qr, qi;
        cdiv(a.dhRe[g], a.dhIm[g], hr, hi, qr, qi);
        float v = strong ? qi * a.freScale + fJ : fJ;
        if (v < 0) v = 0;
        if (v > fmax) v = fmax;
        reF = v;
    }
    if (a.doTime) {
        float qr, qi;
        cdiv(a.thRe[g], a.thIm[g], hr, hi, qr, qi);
        float v = strong ? qr * a.timeScale + tI : tI;
        if (v < 0) v = 0;
        if (v > tmax) v = tmax;
        reT = v;
    }
    float ti = 0.f;
    if (T > 1) ti = roundf((reT - 0.f) * (T - 1) / (tmax - 0.f));  // tmin = timeArr[0] = 0
    const float fi = roundf((reF - a.freArr[0]) * (F - 1) / (fmax - a.freArr[0]));
    a.timeIdx[g] = (ti >= 0.f && ti < (float)T) ? (int)ti : -1;   // also rejects NaN
    a.freIdx[g] = (fi >= 0.f && fi < (float)F) ? (int)fi : -1;
}

// one pass of the "order" iteration: next[i][j] = cur[i][cur[i][j]] where that is in range,
// else what next held before (zero on the first pass) -- reassign_algorithm.c:343-357
__global__ void k_reassign_order(const int *cur, int *next, int T, int F) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)T * F) return;
    const long long base = (long long)blockIdx.y * T * F;
    const int i = (int)(e / F);
    const int v = cur[base + e];
    if (v >= 0 && v < F) next[base + e] = cur[base + (long long)i * F + v];
}

// sort key of source e of clip c (local to a sort chunk): cLocal * cells + target cell, or `dropKey` (= the
// number of cells of a full chunk, behind every cell) when the coefficient is dropped
__global__ void k_reassign_keys(AfxReassignArgs a, int clip0, unsigned dropKey, unsigned *keys) {
    const int F = a.F, T = a.timeLength;
    const long long cells = (long long)T * F;
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= cells) return;
    const long long base = (long long)(clip0 + blockIdx.y) * cells;
    const int i1 = a.timeIdx[base + e], j1 = a.freIdx[base + e];
    const bool ok = !(i1 < 0 || i1 >= T || j1 < 0 || j1 >= F);
    keys[(long long)blockIdx.y * cells + e] = ok ? (unsigned)((long long)blockIdx.y * cells + (long long)i1 * F + j1) : dropKey;
}

// ---- stable LSD radix sort of (key, source index) pairs, 4 bits per pass -------------------------------------------
// A workgroup of 256 threads owns a tile of 2048 consecutive elements, a thread 8 consecutive ones: the order inside
// the tile is (thread, item).  Pass d:
//   k_sort_count   per tile the number of elements of every digit value        -> hist[digit][tile]
//   k_sort_scan    exclusive prefix sum over hist in (digit, tile) order        -> the tile's first output slot per digit
//   k_sort_scatter every element goes to (its digit's slot) + (elements of the same digit before it in the tile):
//                  per-thread counts -> [digit][thread] matrix in LDS -> row prefix sums -> + rank among the thread's items
// Equal digits keep their order, so after the last pass equal KEYS are in ascending source order.  Pass 0 takes the
// source index of an element from its position.  No atomics, no cross-lane instructions: LDS and barriers only.
constexpr int SORT_ITEMS = 8, SORT_TILE = 256 * SORT_ITEMS;

__device__ __forceinline__ void sort_load(const unsigned *keys, long long n, long long tile0, int tid, int shift,
                                          unsigned (&k)[SORT_ITEMS], int (&d)[SORT_ITEMS]) {
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; ++i) {
        const long long e = tile0 + (long long)tid * SORT_ITEMS + i;
        k[i] = e < n ? keys[e] : 0xffffffffu;
        d[i] = e < n ? (int)((k[i] >> shift) & 15u) : -1;
    }
}

__global__ __launch_bounds__(256) void k_sort_count(const unsigned *keys, long long n, int shift, unsigned *hist, int tiles) {
    __shared__ unsigned cnt[16][256];
    const int tid = threadIdx.x;
    unsigned k[SORT_ITEMS];
    int d[SORT_ITEMS];
    sort_load(keys, n, (long long)blockIdx.x * SORT_TILE, tid, shift, k, d);
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        unsigned c = 0;
#pragma unroll
        for (int i = 0; i < SORT_ITEMS; ++i) c += d[i] == v;
        cnt[v][tid] = c;
    }
    __syncthreads();
    if (tid < 16) {
        unsigned sum = 0;
        for (int t = 0; t < 256; ++t) sum += cnt[tid][t];
        hist[(long long)tid * tiles + blockIdx.x] = sum;
    }
}

// exclusive prefix sum over m = 16 * tiles counters, one workgroup of 1024 threads
__global__ __launch_bounds__(1024) void k_sort_scan(unsigned *hist, long long m) {
    __shared__ unsigned part[1024];
    const int tid = threadIdx.x;
    const long long per = (m + 1023) / 1024, lo = (long long)tid * per, hi = lo + per < m ? lo + per : m;
    unsigned sum = 0;
    for (long long i = lo; i < hi; ++i) sum += hist[i];
    part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        unsigned run = 0;
        for (int t = 0; t < 1024; ++t) {
            const unsigned v = part[t];
            part[t] = run;
            run += v;
        }
    }
    __syncthreads();
    unsigned run = part[tid];
    for (long long i = lo; i < hi; ++i) {
        const unsigned v = hist[i];
        hist[i] = run;
        run += v;
    }
}

__global__ __launch_bounds__(256) void k_sort_scatter(const unsigned *keysIn, const unsigned *srcIn, long long n, int shift,
                                                      const unsigned *hist, int tiles, unsigned *keysOut, unsigned *srcOut) {
    __shared__ unsigned cnt[16][256];
    __shared__ unsigned seg[16][16];
    const int tid = threadIdx.x;
    const long long tile0 = (long long)blockIdx.x * SORT_TILE;
    unsigned k[SORT_ITEMS];
    int d[SORT_ITEMS];
    sort_load(keysIn, n, tile0, tid, shift, k, d);
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        unsigned c = 0;
#pragma unroll
        for (int i = 0; i < SORT_ITEMS; ++i) c += d[i] == v;
        cnt[v][tid] = c;
    }
    __syncthreads();
    // exclusive prefix sums along every row of cnt: thread (row = tid >> 4, part = tid & 15) owns 16 entries
    const int row = tid >> 4, part16 = tid & 15;
    unsigned own = 0;
    for (int t = 0; t < 16; ++t) own += cnt[row][16 * part16 + t];
    seg[row][part16] = own;
    __syncthreads();
    unsigned run = 0;
    for (int q = 0; q < part16; ++q) run += seg[row][q];
    for (int t = 0; t < 16; ++t) {
        const unsigned v = cnt[row][16 * part16 + t];
        cnt[row][16 * part16 + t] = run;
        run += v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; ++i) {
        const long long e = tile0 + (long long)tid * SORT_ITEMS + i;
        if (d[i] < 0) continue;
        unsigned before = 0;  // elements of this digit among the thread's earlier items
#pragma unroll
        for (int q = 0; q < i; ++q) before += d[q] == d[i];
        const long long pos = (long long)hist[(long long)d[i] * tiles + blockIdx.x] + cnt[d[i]][tid] + before;
        keysOut[pos] = k[i];
        srcOut[pos] = srcIn ? srcIn[e] : (unsigned)e;
    }
}

// one thread per sorted position; the thread at the first position of a run of equal keys owns the
// target cell and adds the run's sources in order (ascending source index = the reference's loop order)
__global__ void k_reassign_sum(AfxReassignArgs a, int clip0, long long n, unsigned dropKey, const unsigned *keys,
                               const unsigned *src) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const unsigned key = keys[p];
    if (key == dropKey || (p > 0 && keys[p - 1] == key)) return;
    const int F = a.F;
    const long long cells = (long long)a.timeLength * F;
    const long long clipBase = (long long)clip0 * cells;  // key / src are local to the chunk starting at clip0
    float sr = 0.f, si = 0.f;
    for (long long q = p; q < n && keys[q] == key; ++q) {
        const long long e = src[q];  // chunk-local source index: cLocal * cells + i * F + j
        float v1 = a.hRe[clipBase + e], v2 = a.hIm[clipBase + e];
        if ((e % cells % F) & 1) {
            v1 = -v1;
            v2 = -v2;
        }
        if (!a.resultType) {
            sr += v1;
            si += v2;
        } else {
            sr += sqrtf(v1 * v1 + v2 * v2);
        }
    }
    a.outRe[clipBase + (long long)key] += sr;  // the caller's (zeroed) accumulator, one writer per cell
    if (!a.resultType) a.outIm[clipBase + (long long)key] += si;
}

}  // namespace

extern "C" int afxk_reassign(const AfxReassignArgs *a, int order, int *idxScratch, void *stream) {
    const long long cells = (long long)a->timeLength * a->F;
    if (cells <= 0 || a->batch <= 0) return AFX_OK;
    const long long blocks = (cells + 255) / 256;
    if (blocks > 0x7fffffffLL || a->batch > 65535) {
        afxdev_set_error("reassign: %lld cells x %d clips in one launch", cells, a->batch);
        return AFX_ERR_UNSUPPORTED;
    }
    const dim3 grid((unsigned)blocks, (unsigned)a->batch);
    hipLaunchKernelGGL(k_reassign_index, grid, dim3(256), 0, (hipStream_t)stream, *a);
    AFX_LAUNCH_CHECK("k_reassign_index");
    AfxReassignArgs b = *a;
    if (order > 1) {
        if (!idxScratch) {
            afxdev_set_error("reassign: order %d needs index scratch", order);
            return AFX_ERR_ARG;
        }
        AFX_HIP(hipMemsetAsync(idxScratch, 0, sizeof(int) * (size_t)cells * a->batch, (hipStream_t)stream));
        int *cur = a->freIdx, *next = idxScratch;
        for (int k = 0; k < order - 1; ++k) {
            hipLaunchKernelGGL(k_reassign_order, grid, dim3(256), 0, (hipStream_t)stream, cur, next, a->timeLength, a->F);
            AFX_LAUNCH_CHECK("k_reassign_order");
            // the reference copies the result back and keeps iterating on it; the scratch keeps
            // its content between passes, exactly like mTempIndexArr
            AFX_HIP(hipMemcpyAsync(cur, next, sizeof(int) * (size_t)cells * a->batch, hipMemcpyDeviceToDevice,
                                   (hipStream_t)stream));
        }
    }
    // ordered accumulation, chunks of clips of at most ~32 M coefficients per sort
    const long long maxN = 32LL << 20;
    int per = (int)(maxN / cells);
    if (per < 1) per = 1;
    if (per > a->batch) per = a->batch;
    if ((long long)per * cells > 0xffffffffLL) {
        afxdev_set_error("reassign: %lld cells per clip exceed the 32-bit source index", cells);
        return AFX_ERR_UNSUPPORTED;
    }
    const long long nMax = (long long)per * cells;
    if (nMax >= 0xffffffffLL) {  // keys are 32-bit, nMax itself is the key of a dropped coefficient
        afxdev_set_error("reassign: %lld cells per sort chunk exceed the 32-bit key", nMax);
        return AFX_ERR_UNSUPPORTED;
    }
    hipStream_t hs = (hipStream_t)stream;
    const unsigned dropKey = (unsigned)nMax;
    int bits = 1;  // significant bits of the largest key (dropKey)
    while (bits < 32 && (1ull << bits) <= (unsigned long long)nMax) ++bits;
    const int passes = (bits + 3) / 4;
    const int tilesMax = (int)((nMax + SORT_TILE - 1) / SORT_TILE);
    // stream-ordered scratch: two (key, source) buffers that the passes alternate between + the tile histograms;
    // any failure frees what was obtained
    unsigned *kbuf[2] = {nullptr, nullptr}, *sbuf[2] = {nullptr, nullptr}, *hist = nullptr;
    int st = AFX_OK;
    {
        hipError_t e = hipSuccess;
        for (int q = 0; q < 2 && e == hipSuccess; ++q) {
            e = hipMallocAsync(reinterpret_cast<void **>(&kbuf[q]), sizeof(unsigned) * (size_t)nMax, hs);
            if (e == hipSuccess) e = hipMallocAsync(reinterpret_cast<void **>(&sbuf[q]), sizeof(unsigned) * (size_t)nMax, hs);
        }
        if (e == hipSuccess) e = hipMallocAsync(reinterpret_cast<void **>(&hist), sizeof(unsigned) * 16 * (size_t)tilesMax, hs);
        if (e != hipSuccess) {
            afxdev_set_error("reassign: %lld coefficients of sort scratch: %s", nMax, hipGetErrorString(e));
            st = AFX_ERR_NOMEM;
        }
    }
    for (int c0 = 0; c0 < a->batch && st == AFX_OK; c0 += per) {
        const int nc = (a->batch - c0 < per) ? a->batch - c0 : per;
        const long long n = (long long)nc * cells;
        const int tiles = (int)((n + SORT_TILE - 1) / SORT_TILE);
        hipLaunchKernelGGL(k_reassign_keys, dim3((unsigned)blocks, (unsigned)nc), dim3(256), 0, hs, b, c0, dropKey, kbuf[0]);
        int cur = 0;
        for (int pass = 0; pass < passes; ++pass, cur ^= 1) {
            hipLaunchKernelGGL(k_sort_count, dim3((unsigned)tiles), dim3(256), 0, hs, kbuf[cur], n, 4 * pass, hist, tiles);
            hipLaunchKernelGGL(k_sort_scan, dim3(1), dim3(1024), 0, hs, hist, (long long)16 * tiles);
            hipLaunchKernelGGL(k_sort_scatter, dim3((unsigned)tiles), dim3(256), 0, hs, kbuf[cur], pass ? sbuf[cur] : nullptr, n,
                               4 * pass, hist, tiles, kbuf[cur ^ 1], sbuf[cur ^ 1]);
        }
        hipLaunchKernelGGL(k_reassign_sum, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, hs, b, c0, n, dropKey, kbuf[cur],
                           sbuf[cur]);
    }
    for (int q = 0; q < 2; ++q) {
        if (kbuf[q]) (void)hipFreeAsync(kbuf[q], hs);
        if (sbuf[q]) (void)hipFreeAsync(sbuf[q], hs);
    }
    if (hist) (void)hipFreeAsync(hist, hs);
    if (st != AFX_OK) return st;
    AFX_LAUNCH_CHECK("k_reassign_sum");
    return AFX_OK;
}
