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

#include <cstdlib>

#include "afx_device.h"
#include "afx_hipcheck.h"
#include "afx_ldsfft.h"

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

// W_N^k for 0 <= k <= N/2 from the half table tw[0..N/2)
__device__ __forceinline__ float2 twn(const float2 *tw, int k, int halfN) {
    return k < halfN ? tw[k] : make_float2(-1.f, 0.f);
}

// spectrum value of one bin -> up to two planes (re / im); returns the first
__device__ __forceinline__ void map_bin(float2 c, int mode, float normValue, float &v0, float &v1) {
    v1 = 0.f;
    switch (mode) {
        case AFX_SPEC_COMPLEX: v0 = c.x; v1 = c.y; break;
        case AFX_SPEC_POWER: v0 = c.x * c.x + c.y * c.y; break;
        case AFX_SPEC_MAG: v0 = sqrtf(c.x * c.x + c.y * c.y); break;
        case AFX_SPEC_SQUARE: v0 = c.x * c.x - c.y * c.y; v1 = 2.f * c.x * c.y; break;
        case AFX_SPEC_MAG_NORM: v0 = powf(sqrtf(c.x * c.x + c.y * c.y), normValue); break;
        case AFX_SPEC_PHASE: v0 = atan2f(c.y, c.x < 1e-16f ? 1e-16f : c.x); break;
        default: v0 = powf(c.x * c.x + c.y * c.y, normValue); break;  // AFX_SPEC_POWER_NORM
    }
}

// sample q of the (virtually padded) clip
__device__ __forceinline__ float fetch(const float *x, long long q, const AfxStftArgs &a) {
    if (q >= 0 && q < a.dataLength) return x[q];
    if (a.padMode == AFX_PAD_ZERO) return 0.f;
    if (a.padMode == AFX_PAD_CONST) return q < 0 ? a.padValueL : a.padValueR;
    const long long m = afx_pad_index(q, a.dataLength, a.padMode);
    return m < 0 ? 0.f : x[m];
}

// The real frame is packed as M = N/2 complex samples z[n] = (x[2n], x[2n+1]) w (half the
// transform of the reference's zero-imaginary complex FFT), transformed in LDS by in-place
// DIF passes -- two radix-2 stages per pass held in registers (a radix-4 butterfly with the
// radix-2 layout: Z[k] ends at bitrev_m(k), one barrier per two stages) -- and un-packed per
// requested bin: X[k] = E + W_N^k O, E/O from Z[k], conj Z[M-k].  With a banded bank
// (AfxStftArgs::band*) the spectrum values stay in LDS and the filter-bank rows are formed in
// the same launch: no [T,F] round trip through HBM and no dense GEMM.
__global__ void k_stft_generic(AfxStftArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *s = reinterpret_cast<float2 *>(smem_raw);
    const int r = a.radix2Exp;
    const int N = 1 << r, M = N >> 1, m = r - 1;
    float *red = reinterpret_cast<float *>(s + afx_lds_padded_size(M));
    float *prow = red + 16;  // [planes][binCount], band mode only

    const long long frame = blockIdx.x;
    const int b = (int)(frame / a.timeLength);
    const int t = (int)(frame - (long long)b * a.timeLength);
    const float *x = a.x + (long long)b * a.clipStride;
    const long long start = (long long)t * a.hop - a.padLeft;
    const int tid = threadIdx.x, nth = blockDim.x;

    // 1. gather + window (frames overlap by N-hop samples: neighbouring workgroups re-read
    //    them through L2, HBM sees each sample once)
    for (int i = tid; i < M; i += nth) {
        const long long p = start + 2 * i;
        const float v0 = fetch(x, p, a);
        const float v1 = fetch(x, p + 1, a);
        s[afx_lds_pad(i)] = make_float2(v0 * a.window[2 * i], v1 * a.window[2 * i + 1]);
    }
    __syncthreads();

    // 1b. temporal features of the windowed frame (temporal_algorithm.c:138-144)
    if (a.energy) {
        float e = 0.f, z = 0.f;
        for (int i = tid; i < M; i += nth) {
            const float2 v = s[afx_lds_pad(i)];
            e += v.x * v.x;
            e += v.y * v.y;
            if (i > 0 && v.x * s[afx_lds_pad(i - 1)].y < 0.f) z += 1.f;
            if (v.y * v.x < 0.f) z += 1.f;
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

    // 2. M-point complex FFT; W_M^j = W_N^(2j) = tw[2j]
    const float2 *tw = reinterpret_cast<const float2 *>(a.twiddle);
    afx_lds_fft_dif_t<true>(s, m, tw, 2, tid, nth);

    // 3. un-pack the requested bins, map them, store (or keep for the filter bank)
    const bool band = a.bandStart != nullptr;
    const bool two = (a.mode == AFX_SPEC_COMPLEX || a.mode == AFX_SPEC_SQUARE);
    const long long row = frame * (long long)a.binCount;
    for (int j = tid; j < a.binCount; j += nth) {
        int k = a.binLo + j;  // 0 <= k <= M, or up to N-1 with fullSpectrum
        const bool mirror = k > M;
        if (mirror) k = N - k;
        const int ka = k & (M - 1), kb = (M - k) & (M - 1);
        const float2 zk = s[afx_lds_pad(m ? (int)(__brev((unsigned)ka) >> (32 - m)) : 0)];
        const float2 zm = s[afx_lds_pad(m ? (int)(__brev((unsigned)kb) >> (32 - m)) : 0)];
        const float2 E = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
        const float2 O = make_float2(0.5f * (zk.y + zm.y), -0.5f * (zk.x - zm.x));
        const float2 w = twn(tw, k, M);
        float2 c = make_float2(E.x + (w.x * O.x - w.y * O.y), E.y + (w.x * O.y + w.y * O.x));
        if (mirror) c.y = -c.y;
        float v0, v1;
        map_bin(c, a.mode, a.normValue, v0, v1);
        if (band) {
            prow[j] = v0;
            if (two) prow[a.binCount + j] = v1;
        } else {
            a.outRe[row + j] = v0;
            if (two) a.outIm[row + j] = v1;
        }
    }
    if (!band) return;
    __syncthreads();
    // 4. filter bank rows: ascending taps of the row's non-zero span (src/vector/flux_vector.c:55-86)
    for (int j = tid; j < a.bandNum; j += nth) {
        const int k0 = a.bandStart[j] - a.binLo, n = a.bandLen[j];
        const float *wj = a.bandW + j;  // tap-major [maxLen][bandNum]: lanes (rows) read neighbours
        float acc0 = 0.f, acc1 = 0.f;
        for (int q = 0; q < n; ++q) {
            const float wv = wj[(long long)q * a.bandNum];
            acc0 += wv * prow[k0 + q];
            if (two) acc1 += wv * prow[a.binCount + k0 + q];
        }
        if (a.bandPost == AFX_MAP_POW) acc0 = powf(acc0, a.bandPostArg);
        a.outRe[frame * a.bandNum + j] = acc0;
        if (two) a.outIm[frame * a.bandNum + j] = acc1;
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
    int threads = N / 4;
    if (threads < 64) threads = 64;
    if (threads > 256) threads = 256;
    const bool two = (a->mode == AFX_SPEC_COMPLEX || a->mode == AFX_SPEC_SQUARE);
    const size_t lds = (size_t)afx_lds_padded_size(N / 2 > 0 ? N / 2 : 1) * sizeof(float2) + 16 * sizeof(float) +
                       (a->bandStart ? sizeof(float) * (size_t)a->binCount * (two ? 2 : 1) : 0);
    if (lds > 150 * 1024) {
        afxdev_set_error("stft: %zu bytes of LDS per frame", lds);
        return AFX_ERR_UNSUPPORTED;
    }
    if (lds > 48 * 1024) {
        AFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_stft_generic),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    hipLaunchKernelGGL(k_stft_generic, dim3((unsigned)frames), dim3(threads), lds,
                       (hipStream_t)stream, *a);
    AFX_LAUNCH_CHECK("k_stft_generic");
    return AFX_OK;
}
