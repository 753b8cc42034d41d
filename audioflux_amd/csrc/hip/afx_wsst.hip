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
    const float *wr = a.wRe + plane, *wi = a.wIm + plane, *dr = a.dRe + plane, *di = a.dIm + plane;
    float *outR = a.outRe + plane, *outI = a.outIm + plane;
    const float t2 = a.thresh * a.thresh;
    const float twoPi = (float)(2 * 3.14159265358979323846);
    for (int i = 0; i < a.num; ++i) {
        const long long e = (long long)i * a.length + j;
        const float v1 = wr[e], v2 = wi[e];
        const float d1 = dr[e], d2 = di[e];
        const float value = v1 * v1 + v2 * v2;
        const float ph = ((d2 * v1 - d1 * v2) / value) / twoPi;  // __complexDiv, imaginary part
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

}  // namespace

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
