// afx_reassign.hip -- time-frequency reassignment of an STFT (reference:
// src/reassign_algorithm.c:203-414, :611-832).
//
// Inputs are the three STFTs of the signal taken with the window h, its derivative dh and the
// time-weighted window t.h (computed by k_stft_generic, afx_stft.hip).  Per coefficient:
//   f' = f_j - Im(S_dh / S_h) sr / 2pi,   t' = t_i + Re(S_th / S_h) / sr     (:694-716)
//   below the power threshold the coordinate stays (t_i, f_j); clip to the grid   (:772-822)
//   target cell = rounded position of (t', f') on the (frame, bin) grid           (:312-320)
// k_reassign_index stores the two target indices (-1: dropped), k_reassign_order applies the
// reference's "order" iteration of the frequency index (:340-358), and k_reassign_scatter adds
// every coefficient, sign-flipped on odd bins (:378-381), to its target cell.  A target cell
// collects sources from several frames and bins, so the scatter uses float atomics in HBM; the
// order of the float32 additions differs from the reference's loop order by ~1e-7 of the sum.
#include <hip/hip_runtime.h>

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

__global__ void k_reassign_scatter(AfxReassignArgs a) {
    const int F = a.F, T = a.timeLength;
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)T * F) return;
    const int j = (int)(e % F);
    const long long base = (long long)blockIdx.y * T * F;
    const int i1 = a.timeIdx[base + e], j1 = a.freIdx[base + e];
    if (i1 < 0 || i1 >= T || j1 < 0 || j1 >= F) return;
    float v1 = a.hRe[base + e], v2 = a.hIm[base + e];
    if (j & 1) {
        v1 = -v1;
        v2 = -v2;
    }
    const long long o = base + (long long)i1 * F + j1;
    if (!a.resultType) {
        atomicAdd(a.outRe + o, v1);
        atomicAdd(a.outIm + o, v2);
    } else {
        atomicAdd(a.outRe + o, sqrtf(v1 * v1 + v2 * v2));
    }
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
    hipLaunchKernelGGL(k_reassign_scatter, grid, dim3(256), 0, (hipStream_t)stream, b);
    AFX_LAUNCH_CHECK("k_reassign_scatter");
    return AFX_OK;
}
