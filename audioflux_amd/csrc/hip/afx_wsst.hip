// afx_wsst.hip -- the squeezing pass of the wavelet synchrosqueezed transform
// (reference: src/wsst_algorithm.c:242-347).
//
// For every coefficient W[i][j] the instantaneous frequency Im(W'[i][j] / W[i][j]) / 2 pi picks
// the target row i1 (log / linear / nearest-band mapping) and the coefficient is ADDED to
// out[i1][j] when |W|^2 > thresh^2.  The reference scatters row by row; all contributions to a
// cell (i1, j) come from column j, so ONE THREAD PER TIME SAMPLE walks the rows of its column
// in ascending order and accumulates in place: the same float32 summation order as the
// reference, no atomics, and lanes of a wave read 64 neighbouring samples of a row (coalesced).
// HBM traffic: 16 bytes read per coefficient (W, W'), up to 16 read-modify-written.
#include <hip/hip_runtime.h>

#include "afx_device.h"
#include "afx_hipcheck.h"

namespace {

// __arr_roundIndex (wsst_algorithm.c:351-378): nearer neighbour of an ascending table, -1 outside
__device__ __forceinline__ int round_index(const float *arr, int length, float value) {
    const float a = fabsf(value);
    for (int i = 0; i < length - 1; ++i) {
        if (a >= arr[i] && a < arr[i + 1]) return (a - arr[i] < arr[i + 1] - a) ? i : i + 1;
    }
    return -1;
}

__global__ void k_wsst_squeeze(AfxWsstArgs a) {
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= a.length) return;
    const long long plane = (long long)a.num * a.length * blockIdx.y;
    const float *wr = a.wRe + plane, *wi = a.wIm + plane, *dr = a.dRe + plane;
    const float *di = a.phaseInput ? dr : a.dIm + plane;
    float *outR = a.outRe + plane, *outI = a.outIm + plane;
    const float t2 = a.thresh * a.thresh;
    const float twoPi = (float)(2 * 3.14159265358979323846);
    for (int i = 0; i < a.num; ++i) {
        const long long e = (long long)i * a.length + j;
        const float v1 = wr[e], v2 = wi[e];
        const float d1 = dr[e], d2 = a.phaseInput ? 0.f : di[e];
        const float value = v1 * v1 + v2 * v2;
        // instantaneous frequency: Im(W'/W) / 2 pi (__complexDiv), or the phase-difference estimate
        // of synsqObj_synsq handed in through the dRe plane
        const float ph = a.phaseInput ? d1 : ((d2 * v1 - d1 * v2) / value) / twoPi;
        float idx;
        if (a.mode == 0) {
            idx = roundf((log2f(fabsf(ph)) - a.logMin) * a.num / (a.logMax - a.logMin));
        } else if (a.mode == 1) {
            idx = roundf(fabsf(ph - a.fmin) * a.num / (a.fmax - a.fmin));
        } else {
            idx = (float)round_index(a.freNorm, a.num, ph);
        }
        if (!(idx >= 0.f && idx < (float)a.num)) continue;  // also rejects NaN / inf
        if (!(value > t2)) continue;
        const long long o = (long long)(int)idx * a.length + j;
        outR[o] += v1;
        outI[o] += v2;
    }
}

// synsqObj_synsq steps 1-3 (src/synsq_algorithm.c:181-193): phase angle atan2f(re, im) -- the
// reference passes the arguments in this order --, in-place unwrap along time (__vunwrap,
// vector/flux_vector.c:1792-1830, which compares the raw sample with the already unwrapped
// predecessor), first difference (column 0 is 0, the last column repeats its neighbour), / 2 pi.
// The unwrap is a running recurrence: one thread per row walks its row.
__global__ void k_synsq_phase(const float *__restrict__ re, const float *__restrict__ im, int num, long long length,
                              float *__restrict__ phase) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= num) return;
    const float *r = re + (long long)i * length, *q = im + (long long)i * length;
    float *p = phase + (long long)i * length;
    const float PI = 3.14159265358979323846f;
    const double PI_D = 3.14159265358979323846;
    const float twoPi = (float)(2 * PI_D);
    float prev = atan2f(r[0], q[0]);  // unwrapped predecessor
    p[0] = 0.f;
    float lastDiff = 0.f;
    for (long long j = 1; j < length; ++j) {
        const float raw = atan2f(r[j], q[j]);
        float cur = raw;
        const float sub = fabsf(raw - prev);
        if (!((double)sub < PI_D)) {
            int t = (int)floorf((float)((double)sub / (2 * PI_D)));
            const float mod = (float)((double)sub - t * 2 * PI_D);
            if ((double)mod > PI_D) t++;
            cur = (raw > prev) ? (float)((double)raw - t * 2 * PI_D) : (float)((double)raw + t * 2 * PI_D);
        }
        lastDiff = cur - prev;
        p[j] = lastDiff / twoPi;
        prev = cur;
    }
    (void)PI;
    if (length >= 2) p[length - 1] = p[length - 2];
}

}  // namespace

extern "C" int afxk_synsq_phase(const float *re, const float *im, int num, long long length, float *phase,
                                void *stream) {
    if (num <= 0 || length <= 0) return AFX_OK;
    hipLaunchKernelGGL(k_synsq_phase, dim3((unsigned)((num + 63) / 64)), dim3(64), 0, (hipStream_t)stream, re, im,
                       num, length, phase);
    AFX_LAUNCH_CHECK("k_synsq_phase");
    return AFX_OK;
}

extern "C" int afxk_wsst_squeeze(const AfxWsstArgs *a, void *stream) {
    if (a->length <= 0 || a->num <= 0 || a->batch <= 0) return AFX_OK;
    const long long blocks = (a->length + 255) / 256;
    if (blocks > 0x7fffffffLL || a->batch > 65535) {
        afxdev_set_error("wsst: %lld samples x %d chunks in one launch", a->length, a->batch);
        return AFX_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(k_wsst_squeeze, dim3((unsigned)blocks, (unsigned)a->batch), dim3(256), 0,
                       (hipStream_t)stream, *a);
    AFX_LAUNCH_CHECK("k_wsst_squeeze");
    return AFX_OK;
}
