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
// target with a stable device radix sort (rocPRIM; ties keep the ascending source order), and one
// thread per target cell walks its run and sums sequentially -- the float32 additions happen in
// the reference's order, the result is deterministic run to run and bit-identical to the
// reference's accumulation whenever the transforms and target indices agree.
#include <hip/hip_runtime.h>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

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

// sort key of source e of clip c (local to a sort chunk): cLocal * cells + target cell, or all ones
// when the coefficient is dropped (sorted behind every cell)
__global__ void k_reassign_keys(AfxReassignArgs a, int clip0, unsigned long long *keys) {
    const int F = a.F, T = a.timeLength;
    const long long cells = (long long)T * F;
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= cells) return;
    const long long base = (long long)(clip0 + blockIdx.y) * cells;
    const int i1 = a.timeIdx[base + e], j1 = a.freIdx[base + e];
    const bool ok = !(i1 < 0 || i1 >= T || j1 < 0 || j1 >= F);
    keys[(long long)blockIdx.y * cells + e] =
        ok ? (unsigned long long)((long long)blockIdx.y * cells + (long long)i1 * F + j1) : ~0ull;
}

// one thread per sorted position; the thread at the first position of a run of equal keys owns the
// target cell and adds the run's sources in order (ascending source index = the reference's loop order)
__global__ void k_reassign_sum(AfxReassignArgs a, int clip0, long long n, const unsigned long long *keys,
                               const unsigned *src) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const unsigned long long key = keys[p];
    if (key == ~0ull || (p > 0 && keys[p - 1] == key)) return;
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
    hipStream_t hs = (hipStream_t)stream;
    unsigned long long *keysIn = nullptr, *keysOut = nullptr;
    unsigned *srcOut = nullptr;
    void *tmp = nullptr;
    size_t tmpBytes = 0;
    rocprim::counting_iterator<unsigned> srcIn(0);
    // 64-bit keys span [0, per * cells): only the bits in use are sorted
    unsigned endBit = 1;
    while (endBit < 64 && (1ull << endBit) < (unsigned long long)nMax) ++endBit;
    AFX_HIP(rocprim::radix_sort_pairs(nullptr, tmpBytes, keysIn, keysOut, srcIn, srcOut, (size_t)nMax, 0u, 64u, hs));
    int st = AFX_OK;
    {   // stream-ordered scratch; any failure frees what was obtained (the frees below take nullptr)
        hipError_t e = hipMallocAsync(reinterpret_cast<void **>(&keysIn), sizeof(unsigned long long) * nMax, hs);
        if (e == hipSuccess) e = hipMallocAsync(reinterpret_cast<void **>(&keysOut), sizeof(unsigned long long) * nMax, hs);
        if (e == hipSuccess) e = hipMallocAsync(reinterpret_cast<void **>(&srcOut), sizeof(unsigned) * nMax, hs);
        if (e == hipSuccess) e = hipMallocAsync(&tmp, tmpBytes ? tmpBytes : 4, hs);
        if (e != hipSuccess) {
            afxdev_set_error("reassign: %lld coefficients of sort scratch: %s", nMax, hipGetErrorString(e));
            st = AFX_ERR_NOMEM;
        }
    }
    for (int c0 = 0; c0 < a->batch && st == AFX_OK; c0 += per) {
        const int nc = (a->batch - c0 < per) ? a->batch - c0 : per;
        const long long n = (long long)nc * cells;
        hipLaunchKernelGGL(k_reassign_keys, dim3((unsigned)blocks, (unsigned)nc), dim3(256), 0, hs, b, c0, keysIn);
        // dropped coefficients carry the all-ones key: all 64 bits take part so that they sort last
        hipError_t e = rocprim::radix_sort_pairs(tmp, tmpBytes, keysIn, keysOut, srcIn, srcOut, (size_t)n, 0u, 64u, hs);
        if (e != hipSuccess) {
            afxdev_set_error("reassign: rocprim::radix_sort_pairs: %s", hipGetErrorString(e));
            st = AFX_ERR_HIP;
            break;
        }
        hipLaunchKernelGGL(k_reassign_sum, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, hs, b, c0, n, keysOut, srcOut);
    }
    (void)endBit;
    if (keysIn) (void)hipFreeAsync(keysIn, hs);
    if (keysOut) (void)hipFreeAsync(keysOut, hs);
    if (srcOut) (void)hipFreeAsync(srcOut, hs);
    if (tmp) (void)hipFreeAsync(tmp, hs);
    if (st != AFX_OK) return st;
    AFX_LAUNCH_CHECK("k_reassign_sum");
    return AFX_OK;
}
