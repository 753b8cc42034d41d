// afx_stft.hip -- generic framed FFT kernel ("K1/K2/K4/K5" of SURVEY.md 2b for
// ANY power-of-two frame length): one workgroup per frame, the frame is
// gathered from HBM with the analysis window applied, transformed by an
// in-place radix-2 decimation-in-frequency FFT held entirely in LDS, and the
// requested bins are stored as complex / power / magnitude / squared-complex.
//
// Computes what the reference computes in stft_algorithm.c:696-715 (window
// multiply + per-frame FFT), flux_complex.c:254-286 (crop to N/2+1 bins),
// flux_complex.c:469-503 / bft_algorithm.c:459-504 (power, magnitude, S^2,
// power^p) and temporal_algorithm.c:138-144 (energy / rms / zcr of the
// windowed frame) -- in one pass, without materialising the [T,N] complex
// scratch the reference allocates.
//
// This is the size-generic path.  The headline configuration (N=2048) is
// served by the register-resident fused kernel in afx_melfused.hip.
#include <hip/hip_runtime.h>

#include "afx_device.h"
#include "afx_hipcheck.h"

namespace {

__device__ __forceinline__ float block_sum(float v, float *red /* >= 16 floats of LDS */) {
    // wave reduction (64 lanes) then one LDS hop between waves
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nwave = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < nwave; ++w) t += red[w];
    return t;
}

__global__ void k_stft_generic(AfxStftArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *s = reinterpret_cast<float2 *>(smem_raw);
    const int r = a.radix2Exp;
    const int N = 1 << r;
    float *red = reinterpret_cast<float *>(s + N);

    const long long frame = blockIdx.x;
    const int b = (int)(frame / a.timeLength);
    const int t = (int)(frame - (long long)b * a.timeLength);
    const float *x = a.x + (long long)b * a.clipStride;
    const long long start = (long long)t * a.hop - a.padLeft;
    const int tid = threadIdx.x, nth = blockDim.x;

    // 1. gather + window (frames overlap by N-hop samples: neighbouring
    //    workgroups re-read them through L2, HBM sees each sample once)
    for (int i = tid; i < N; i += nth) {
        const long long p = start + i;
        float v = (p >= 0 && p < a.dataLength) ? x[p] : 0.f;
        v *= a.window[i];
        s[i] = make_float2(v, 0.f);
    }
    __syncthreads();

    // 1b. temporal features of the windowed frame (temporal_algorithm.c:138-144)
    if (a.energy) {
        float e = 0.f, z = 0.f;
        for (int i = tid; i < N; i += nth) {
            const float v = s[i].x;
            e += v * v;
            if (i > 0 && v * s[i - 1].x < 0.f) z += 1.f;
        }
        e = block_sum(e, red);
        z = block_sum(z, red);
        if (tid == 0) {
            a.energy[frame] = e;
            a.rms[frame] = sqrtf(e / (float)N);
            a.zcr[frame] = (float)((double)z / (double)N);
        }
        __syncthreads();
    }

    // 2. in-place radix-2 DIF: after r stages X[k] sits at index bitrev_r(k)
    const float2 *tw = reinterpret_cast<const float2 *>(a.twiddle);
    for (int st = 0; st < r; ++st) {
        const int half = N >> (st + 1);
        for (int j = tid; j < (N >> 1); j += nth) {
            const int pos = j & (half - 1);
            const int i0 = ((j - pos) << 1) + pos;
            const int i1 = i0 + half;
            const float2 u = s[i0], v = s[i1];
            const float2 w = tw[pos << st];
            const float dx = u.x - v.x, dy = u.y - v.y;
            s[i0] = make_float2(u.x + v.x, u.y + v.y);
            s[i1] = make_float2(dx * w.x - dy * w.y, dx * w.y + dy * w.x);
        }
        __syncthreads();
    }

    // 3. store the requested bins
    const long long row = frame * (long long)a.binCount;
    for (int j = tid; j < a.binCount; j += nth) {
        const unsigned k = (unsigned)(a.binLo + j);
        const unsigned idx = __brev(k) >> (32 - r);
        const float2 c = s[idx];
        switch (a.mode) {
            case AFX_SPEC_COMPLEX:
                a.outRe[row + j] = c.x;
                a.outIm[row + j] = c.y;
                break;
            case AFX_SPEC_POWER:
                a.outRe[row + j] = c.x * c.x + c.y * c.y;
                break;
            case AFX_SPEC_MAG:
                a.outRe[row + j] = sqrtf(c.x * c.x + c.y * c.y);
                break;
            case AFX_SPEC_SQUARE:
                a.outRe[row + j] = c.x * c.x - c.y * c.y;
                a.outIm[row + j] = 2.f * c.x * c.y;
                break;
            case AFX_SPEC_MAG_NORM:
                a.outRe[row + j] = powf(sqrtf(c.x * c.x + c.y * c.y), a.normValue);
                break;
            default:  // AFX_SPEC_POWER_NORM
                a.outRe[row + j] = powf(c.x * c.x + c.y * c.y, a.normValue);
                break;
        }
    }
}

}  // namespace

extern "C" int afxk_stft(const AfxStftArgs *a, void *stream) {
    if (a->radix2Exp < 1 || a->radix2Exp > 14) {
        afxdev_set_error("stft: fftLength 2^%d is outside the supported 2..16384", a->radix2Exp);
        return AFX_ERR_UNSUPPORTED;
    }
    const long long frames = (long long)a->batch * a->timeLength;
    if (frames <= 0) return AFX_OK;
    if (frames > 0x7fffffffLL) {
        afxdev_set_error("stft: %lld frames in one launch", frames);
        return AFX_ERR_UNSUPPORTED;
    }
    const int N = 1 << a->radix2Exp;
    int threads = N / 2;
    if (threads < 64) threads = 64;
    if (threads > 256) threads = 256;
    const size_t lds = (size_t)N * sizeof(float2) + 16 * sizeof(float);
    if (lds > 48 * 1024) {
        AFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_stft_generic),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    hipLaunchKernelGGL(k_stft_generic, dim3((unsigned)frames), dim3(threads), lds,
                       (hipStream_t)stream, *a);
    AFX_LAUNCH_CHECK("k_stft_generic");
    return AFX_OK;
}
