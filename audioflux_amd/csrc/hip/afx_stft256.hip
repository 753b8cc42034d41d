// afx_stft256.hip -- STFT at n_fft 256: TWO frames per wave transform (round 6).
//
// A frame of 256 real samples is 128 complex points -- half a wave's worth of the register transforms of afx_wavefft_small.h.  Two
// real frames a, b (consecutive frames of one clip) therefore ride through ONE 256-point complex transform as
// z = a w + i b w:   A[k] = (Z[k] + conj Z[256 - k]) / 2,   B[k] = (Z[k] - conj Z[256 - k]) / (2 i),   0 <= k < 256
// -- every bin of both spectra, mirrors included, straight from the natural-order image of Z (stft_algorithm.c:717-803 runs a
// complex transform of each real frame and keeps all N bins).  Lanes hold consecutive bins: every store covers 256 contiguous bytes.
// The size-generic kernel (one workgroup per frame, LDS transform of 128 points) stays for frames that leave their clip.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <mutex>

#include "afx_device.h"
#include "afx_hipcheck.h"
#include "afx_wavefft_small.h"

namespace {

typedef afxws::Fft512 F;  // its 256-point complex core
constexpr int NW = 12;    // waves per workgroup (12 x 2.5 KB of exchange image, 1 KB of window, 3.7 KB of twiddles)

__device__ __forceinline__ void map_value(float re, float im, int mode, float normValue, float &v0, float &v1) {
    v1 = 0.f;
    switch (mode) {
        case AFX_SPEC_COMPLEX: v0 = re; v1 = im; break;
        case AFX_SPEC_POWER: v0 = re * re + im * im; break;
        case AFX_SPEC_MAG: v0 = sqrtf(re * re + im * im); break;
        case AFX_SPEC_SQUARE: v0 = re * re - im * im; v1 = 2.f * re * im; break;
        case AFX_SPEC_MAG_NORM: v0 = powf(sqrtf(re * re + im * im), normValue); break;
        case AFX_SPEC_PHASE: v0 = atan2f(im, re < 1e-16f ? 1e-16f : re); break;
        default: v0 = powf(re * re + im * im, normValue); break;  // AFX_SPEC_POWER_NORM
    }
}

// FULL: complex results, all 256 bins (stftObj_stft's layout): no range checks, no map
template <bool FULL>
__global__ __launch_bounds__(NW * 64) void k_stft_256(AfxStftArgs a, const float2 *__restrict__ tab, int pairsPerWave) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int N = 256;
    float *win = reinterpret_cast<float *>(smem_raw);
    v2 *tabTw = reinterpret_cast<v2 *>(win + N);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v2 *ex = tabTw + F::TAB_F2 + wave * F::EX_F2;
    for (int i = threadIdx.x; i < N; i += NW * 64) win[i] = a.window[i];
    for (int i = threadIdx.x; i < F::TAB_F2; i += NW * 64) tabTw[i] = v2{tab[i].x, tab[i].y};
    __syncthreads();
    // pairs never straddle two clips (a quiet clip's frame beside a loud one's would take that frame's rounding errors)
    const long long ppc = (a.timeLength + 1) / 2, pairs = ppc * a.batch;
    long long p = ((long long)blockIdx.x * NW + wave) * pairsPerWave, pEnd = p + pairsPerWave;
    if (pEnd > pairs) pEnd = pairs;
    const bool two = (a.mode == AFX_SPEC_COMPLEX || a.mode == AFX_SPEC_SQUARE);
    const long long pitch = a.outPitch ? a.outPitch : (long long)a.binCount;
    float w[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) w[r] = win[64 * r + lane];
    for (; p < pEnd; ++p) {
        const int b = (int)(p / ppc), ta = 2 * (int)(p - (long long)b * ppc);
        const int tb = ta + 1 < a.timeLength ? ta + 1 : ta;  // (an odd frame count: the clip's last frame rides twice, stored once)
        const long long fa = (long long)b * a.timeLength + ta, fb = (long long)b * a.timeLength + tb;
        const float *xa = a.x + (long long)b * a.clipStride + (long long)ta * a.hop;
        const float *xb = a.x + (long long)b * a.clipStride + (long long)tb * a.hop;
        v2 v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = v2{xa[64 * r + lane] * w[r], xb[64 * r + lane] * w[r]};
        F::cfft(v, ex, tabTw, lane);
        __builtin_amdgcn_s_setprio(0);  // (the transform runs above the other waves' loads and stores, like the other wave kernels)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = lane + 64 * q;
            const v2 Zk = v[q], Zm = ex[(N - k) & (N - 1)];
            const float are = 0.5f * (Zk.x + Zm.x), aim = 0.5f * (Zk.y - Zm.y);
            const float bre = 0.5f * (Zk.y + Zm.y), bim = -0.5f * (Zk.x - Zm.x);
            if constexpr (FULL) {
                a.outRe[fa * N + k] = are;
                a.outIm[fa * N + k] = aim;
                if (fb != fa) {
                    a.outRe[fb * N + k] = bre;
                    a.outIm[fb * N + k] = bim;
                }
            } else {
                const int j = k - a.binLo;
                if (j >= 0 && j < a.binCount) {
                    float v0, v1;
                    map_value(are, aim, a.mode, a.normValue, v0, v1);
                    a.outRe[fa * pitch + j] = v0;
                    if (two) a.outIm[fa * pitch + j] = v1;
                    if (fb != fa) {
                        map_value(bre, bim, a.mode, a.normValue, v0, v1);
                        a.outRe[fb * pitch + j] = v0;
                        if (two) a.outIm[fb * pitch + j] = v1;
                    }
                }
            }
        }
        wave_lds_order();  // the image is read: the next pair's first exchange may overwrite it
    }
}

// twiddle tables of the 256-point core, one device copy per device (never freed)
const float2 *tables256() {
    static std::mutex mu;
    static float2 *dTab[AFX_MAX_DEVICES] = {};
    const int dev = afxdev_current_device();
    if (dev < 0 || dev >= AFX_MAX_DEVICES) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    if (!dTab[dev]) {
        float *h = static_cast<float *>(calloc(2 * F::TAB_F2, sizeof(float)));
        if (!h) return nullptr;
        F::fill_tables(h);
        float2 *d = nullptr;
        int st = afxdev_malloc(reinterpret_cast<void **>(&d), sizeof(float) * 2 * F::TAB_F2);
        if (st == AFX_OK && hipMemcpy(d, h, sizeof(float) * 2 * F::TAB_F2, hipMemcpyHostToDevice) != hipSuccess) st = AFX_ERR_HIP;
        free(h);
        if (st != AFX_OK) {
            afxdev_free(d);
            return nullptr;
        }
        dTab[dev] = d;
    }
    return dTab[dev];
}

template <bool FULL>
int launch256(const AfxStftArgs *a, const float2 *tab, void *stream) {
    const long long pairs = (long long)a->batch * ((a->timeLength + 1) / 2);
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    long long ppw = (pairs + 2LL * cus * NW - 1) / (2LL * cus * NW);  // two rounds of workgroups
    if (ppw < 1) ppw = 1;
    const long long waves = (pairs + ppw - 1) / ppw, blocks = (waves + NW - 1) / NW;
    if (blocks > 0x7fffffffLL) return AFX_ERR_UNSUPPORTED;
    const size_t lds = sizeof(float) * 256 + sizeof(float2) * (size_t)(F::TAB_F2 + NW * F::EX_F2);
    hipLaunchKernelGGL(k_stft_256<FULL>, dim3((unsigned)blocks), dim3(NW * 64), lds, (hipStream_t)stream, *a, tab, (int)ppw);
    AFX_LAUNCH_CHECK("k_stft_256");
    return AFX_OK;
}

}  // namespace

// AFX_ERR_UNSUPPORTED: the caller runs the size-generic kernel (frames that leave their clip, a bank in the same launch, temporal features)
extern "C" int afxk_stft256(const AfxStftArgs *a, void *stream) {
    if (a->radix2Exp != 8 || a->bandStart || a->energy || a->binLo < 0 || a->binCount < 1 || a->binLo + a->binCount > 256 ||
        a->padLeft != 0 || a->hop < 1 || (long long)(a->timeLength - 1) * a->hop + 256 > a->dataLength)
        return AFX_ERR_UNSUPPORTED;
    const bool two = (a->mode == AFX_SPEC_COMPLEX || a->mode == AFX_SPEC_SQUARE);
    if (!a->outRe || (two && !a->outIm)) return AFX_ERR_ARG;
    const long long total = (long long)a->batch * a->timeLength;
    if (total <= 0) return AFX_OK;
    if (total > 0x7fffffffLL) return AFX_ERR_UNSUPPORTED;
    const float2 *tab = tables256();
    if (!tab) return AFX_ERR_UNSUPPORTED;
    if (a->mode == AFX_SPEC_COMPLEX && a->binLo == 0 && a->binCount == 256 && (a->outPitch == 0 || a->outPitch == 256))
        return launch256<true>(a, tab, stream);
    return launch256<false>(a, tab, stream);
}
