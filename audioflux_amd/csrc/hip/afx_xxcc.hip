// afx_xxcc.hip -- post-pass of the "standard" cepstral coefficients ("K7" of
// SURVEY.md 2b): energy insertion + delta / delta-delta.  One lane per frame;
// rows are short (ccNum+1 <= a few dozen) so the whole row lives in registers
// / scratch-free loops over global memory are fine -- this pass is a few
// hundred bytes per frame next to the 2 KB the STFT reads.
//
// Follows xxccObj_xxccStandard, src/feature/xxcc_algorithm.c:244-292, with
// util_delta (src/util/flux_util.c:803-815) = filterDesign_filter
// (src/dsp/filterDesign_fir.c:229-248) over filterDesign_smooth1 taps
// (filterDesign_fir.c:194-217).  NOTE the reference differentiates along the
// COEFFICIENT axis of each frame, not along time; reproduced as is.
#include <hip/hip_runtime.h>

#include "afx_device.h"
#include "afx_hipcheck.h"

namespace {

constexpr int MAX_TAPS = 63;

struct Taps {
    float b[MAX_TAPS];
};

// y[i] = sum_{j<=i, j<order} b[j] x[i-j]
__device__ __forceinline__ float fir_at(const Taps &tp, int order, const float *x, int i) {
    float y = 0.f;
    for (int j = 0; j < order; ++j) {
        if (i >= j) y = y + tp.b[j] * x[i - j];
    }
    return y;
}

__global__ void k_xxcc_standard(const float *__restrict__ cc, const float *__restrict__ energy,
                                long long rows, int ccNum, int energyType, int order, Taps tp,
                                float *__restrict__ coe, float *__restrict__ d1,
                                float *__restrict__ d2) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const int outLen = ccNum + (energyType == 1 ? 1 : 0);
    const float *in = cc + r * ccNum;
    float *c = coe + r * outLen;
    float *a = d1 + r * outLen;
    float *b = d2 + r * outLen;

    float e = 0.f;
    if (energyType != 2) {
        const float v = energy[r];
        e = (v < 1e-8f) ? logf(1e-8f) : logf(v);
    }
    if (energyType == 0) {
        for (int j = 0; j < ccNum; ++j) c[j] = (j == 0) ? e : in[j];
    } else if (energyType == 1) {
        c[0] = e;
        for (int j = 0; j < ccNum; ++j) c[j + 1] = in[j];
    } else {
        for (int j = 0; j < ccNum; ++j) c[j] = in[j];
    }
    for (int i = 0; i < outLen; ++i) a[i] = fir_at(tp, order, c, i);
    for (int i = 0; i < outLen; ++i) b[i] = fir_at(tp, order, a, i);
}

}  // namespace

extern "C" int afxk_xxcc_standard(const float *cc, const float *energy, long long rows, int ccNum,
                                  int energyType, int deltaLen, float *coe, float *delta1,
                                  float *delta2, void *stream) {
    if (rows <= 0) return AFX_OK;
    if (deltaLen > MAX_TAPS || deltaLen < 3 || !(deltaLen & 1)) {
        afxdev_set_error("xxccStandard: deltaWindowLength %d unsupported (odd, 3..%d)", deltaLen,
                         MAX_TAPS);
        return AFX_ERR_UNSUPPORTED;
    }
    Taps tp;
    const int m = deltaLen / 2;
    float v1 = 0.f;
    for (int i = 1; i <= m; ++i) v1 += (float)(i * i);
    for (int i = m, j = 0; i >= -m; --i, ++j) tp.b[j] = (float)i / v1;
    const int threads = 256;
    const long long blocks = (rows + threads - 1) / threads;
    hipLaunchKernelGGL(k_xxcc_standard, dim3((unsigned)blocks), dim3(threads), 0,
                       (hipStream_t)stream, cc, energy, rows, ccNum, energyType, deltaLen, tp, coe,
                       delta1, delta2);
    AFX_LAUNCH_CHECK("k_xxcc_standard");
    return AFX_OK;
}
