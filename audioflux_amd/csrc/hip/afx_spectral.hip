// afx_spectral.hip -- small streaming kernels of the spectrogram object (afx_spectrogram.c):
//
//   k_spec_map   spectrogramObj_spectrogram1's entry: a caller-supplied complex STFT
//                [rows, fftLength] (split planes) -> per-bin power / magnitude / normed value /
//                phase of the bins [binLo, binLo+binCount)  (spectrogram_algorithm.c:1037-1087).
//   k_row_post   tail of the chroma scales: optional powf(., normValue) and the per-frame
//                normalisation of the [rows, n] chroma matrix (__mnormalize,
//                src/vector/flux_vector.c:1058-1160; spectrogram_algorithm.c:1146-1174).
//
// Both touch every element once: HBM-bound, one element (k_spec_map) / one row (k_row_post) per thread.
#include <hip/hip_runtime.h>

#include "afx_device.h"
#include "afx_hipcheck.h"

namespace {

__global__ void k_spec_map(const float *__restrict__ re, const float *__restrict__ im, long long rows,
                           int rowPitch, int binLo, int binCount, int mode, float normValue,
                           float *__restrict__ out, float *__restrict__ out2) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * binCount) return;
    const long long row = e / binCount;
    const int j = (int)(e - row * binCount);
    const float a = re[row * rowPitch + binLo + j], b = im[row * rowPitch + binLo + j];
    float v;
    switch (mode) {
        case AFX_SPEC_COMPLEX: v = a; out2[e] = b; break;
        case AFX_SPEC_SQUARE: v = a * a - b * b; out2[e] = 2 * a * b; break;  // bft_algorithm.c:459-468
        case AFX_SPEC_POWER: v = a * a + b * b; break;
        case AFX_SPEC_MAG: v = sqrtf(a * a + b * b); break;
        case AFX_SPEC_MAG_NORM: v = powf(sqrtf(a * a + b * b), normValue); break;
        case AFX_SPEC_PHASE: v = atan2f(b, a < 1e-16f ? 1e-16f : a); break;
        default: v = powf(a * a + b * b, normValue); break;  // AFX_SPEC_POWER_NORM
    }
    out[e] = v;
}

// normType: 0 none, 1 max, 2 min, 3 P2, 4 P1 (ChromaDataNormalType)
__global__ void k_row_post(float *__restrict__ data, long long rows, int n, int doPow, float powArg,
                           int normType) {
    const long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    float *d = data + row * n;
    float red = normType == 2 ? 3.4e38f : 0.f;
    for (int k = 0; k < n; ++k) {
        float v = d[k];
        if (doPow) {
            v = powf(v, powArg);
            d[k] = v;
        }
        const float av = fabsf(v);
        if (normType == 1) red = fmaxf(red, av);
        else if (normType == 2) red = fminf(red, av);
        else if (normType == 3) red += av * av;
        else red += av;
    }
    if (normType == 0) return;
    if (normType == 3) red = sqrtf(red);
    if (red == 0.f) return;  // __mnormalize leaves an all-zero row alone
    for (int k = 0; k < n; ++k) d[k] = d[k] / red;
}

}  // namespace

extern "C" int afxk_spec_map(const float *re, const float *im, long long rows, int rowPitch, int binLo,
                             int binCount, int mode, float normValue, float *out, float *out2, void *stream) {
    const long long total = rows * binCount;
    if (total <= 0) return AFX_OK;
    const long long blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffLL) {
        afxdev_set_error("spec_map: %lld elements in one launch", total);
        return AFX_ERR_UNSUPPORTED;
    }
    if ((mode == AFX_SPEC_COMPLEX || mode == AFX_SPEC_SQUARE) && !out2) {
        afxdev_set_error("spec_map: complex modes need a second output plane");
        return AFX_ERR_ARG;
    }
    hipLaunchKernelGGL(k_spec_map, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, re, im, rows,
                       rowPitch, binLo, binCount, mode, normValue, out, out2);
    AFX_LAUNCH_CHECK("k_spec_map");
    return AFX_OK;
}

extern "C" int afxk_row_post(float *data, long long rows, int n, int doPow, float powArg, int normType,
                             void *stream) {
    if (rows <= 0 || n <= 0 || (!doPow && normType == 0)) return AFX_OK;
    const long long blocks = (rows + 127) / 128;
    if (blocks > 0x7fffffffLL) {
        afxdev_set_error("row_post: %lld rows in one launch", rows);
        return AFX_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(k_row_post, dim3((unsigned)blocks), dim3(128), 0, (hipStream_t)stream, data, rows, n,
                       doPow, powArg, normType);
    AFX_LAUNCH_CHECK("k_row_post");
    return AFX_OK;
}
