// afx_cepstrum.hip -- cepstral coefficients of a [rows, num] spectrogram:
// rectify (log10 | cube root) -> first ccNum rows of the orthonormal DCT-II
// ("K6" of SURVEY.md 2b), specialised for the shapes cepstra have (num <= 256
// bands, ccNum <= 32): memory-bound, 4*num bytes in + 4*ccNum bytes out per frame.
//
// Replaces the per-frame loop of xxccObj_xxcc (src/feature/xxcc_algorithm.c:
// 124-155: log10f(max(x,1e-8)) or powf(x,1/3), fftObj_dct / dctObj_dct, copy of
// the first ccNum outputs).  Shapes outside the specialisation use the generic
// MFMA GEMM (afx_gemm.hip) with the same rectification fused into its A load.
//
// A workgroup takes 64 frames: coalesced float4 loads, rectified on the fly,
// staged in LDS with a 4-float row pad (conflict-free ds_read_b128 when 64 lanes
// read 64 different rows at one column); then wave g computes coefficients
// g, g+4, g+8, ... for all 64 frames -- the DCT row is wave-uniform, so its
// elements arrive through scalar loads and each LDS read feeds NQ FMAs.
#include <hip/hip_runtime.h>

#include "afx_device.h"
#include "afx_hipcheck.h"

namespace {

__device__ __forceinline__ float rect(float v, int pre) {
    if (pre == AFX_MAP_LOG10) {
        if (v < 1e-8f) v = 1e-8f;
        return log10f(v);
    }
    if (pre == AFX_MAP_CBRT) return powf(v, (float)(1.0 / 3));
    return v;
}

template <int NQ>
__global__ __launch_bounds__(256) void k_cepstrum(const float *__restrict__ in, long long rows,
                                                  int num, const float *__restrict__ dct,
                                                  int ccNum, int pre, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float s[];
    const int pitch = num + 4;
    const int tid = threadIdx.x;
    const int f = tid & 63;
    const int g = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n4 = num >> 2;
    const long long tiles = (rows + 63) / 64;
    for (long long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const long long r0 = tile * 64;
        const int nr = (int)((rows - r0 < 64) ? rows - r0 : 64);
        const float4 *in4 = reinterpret_cast<const float4 *>(in + r0 * num);
        for (int i = tid; i < nr * n4; i += 256) {
            float4 v = in4[i];
            const int row = i / n4, col = (i - row * n4) << 2;
            v.x = rect(v.x, pre);
            v.y = rect(v.y, pre);
            v.z = rect(v.z, pre);
            v.w = rect(v.w, pre);
            *reinterpret_cast<float4 *>(&s[row * pitch + col]) = v;
        }
        __syncthreads();
        if (f < nr) {
            float acc[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) acc[q] = 0.f;
            const float4 *srow = reinterpret_cast<const float4 *>(&s[f * pitch]);
            for (int m4 = 0; m4 < n4; ++m4) {
                const float4 p = srow[m4];
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int j = g + 4 * q;
                    if (j < ccNum) {
                        const float *d = dct + (long long)j * num + 4 * m4;  // wave-uniform
                        acc[q] = fmaf(d[0], p.x, acc[q]);
                        acc[q] = fmaf(d[1], p.y, acc[q]);
                        acc[q] = fmaf(d[2], p.z, acc[q]);
                        acc[q] = fmaf(d[3], p.w, acc[q]);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int j = g + 4 * q;
                if (j < ccNum) out[(r0 + f) * ccNum + j] = acc[q];
            }
        }
        __syncthreads();
    }
}

// ---- MFMA variant for the common shape (num a multiple of 16, <= 16 coefficients) -----
// One wave owns 16 consecutive frames: C[16 frames, 16 coeffs] = rect(A)[16, num] . D^T,
// accumulated with v_mfma_f32_16x16x4_f32.  Lane l loads float4 A[frame l&15][16u + 4(l>>4) ..+3]
// straight from HBM (no LDS, no barrier), rectifies it, and feeds MFMA (u,q) whose k-slot
// (l>>4) stands for band 16u + 4(l>>4) + q; the matching D elements are loop-invariant and
// live in VGPRs.  Memory-bound: 4*num bytes in, 4*ccNum out per frame.
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NUM>
__global__ __launch_bounds__(256) void k_cepstrum_mfma(const float *__restrict__ in, long long rows,
                                                       const float *__restrict__ dct, int ccNum,
                                                       int pre, float *__restrict__ out) {
    constexpr int U = NUM / 16;
    const int lane = threadIdx.x & 63;
    const int fi = lane & 15, g = lane >> 4;
    float4 d[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        d[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (fi < ccNum) d[u] = *reinterpret_cast<const float4 *>(dct + (long long)fi * NUM + 16 * u + 4 * g);
    }
    const long long groups = (rows + 15) / 16;
    const long long gw = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long nw = (long long)gridDim.x * 4;
    for (long long grp = gw; grp < groups; grp += nw) {
        const long long r0 = grp * 16;
        long long r = r0 + fi;
        if (r >= rows) r = rows - 1;  // tail: duplicate the last row, its results are not stored
        const float4 *src = reinterpret_cast<const float4 *>(in + r * NUM) + g;
        float4 a[U];
#pragma unroll
        for (int u = 0; u < U; ++u) a[u] = src[4 * u];
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < U; ++u) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(rect(a[u].x, pre), d[u].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(rect(a[u].y, pre), d[u].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(rect(a[u].z, pre), d[u].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(rect(a[u].w, pre), d[u].w, acc, 0, 0, 0);
        }
        // C layout: column (coefficient) = lane & 15, row (frame) = 4*(lane>>4) + reg
        if (fi < ccNum) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const long long rr = r0 + 4 * g + reg;
                if (rr < rows) out[rr * ccNum + fi] = acc[reg];
            }
        }
    }
}

}  // namespace

// 1 when the specialised kernel can run this shape
extern "C" int afxk_cepstrum_supported(const float *in, int num, int ccNum) {
    return num >= 4 && num <= 256 && (num & 3) == 0 && ccNum >= 1 && ccNum <= 32 &&
           (reinterpret_cast<uintptr_t>(in) & 15) == 0;
}

extern "C" int afxk_cepstrum(const float *in, long long rows, int num, const float *dct, int ccNum,
                             int pre, float *out, void *stream) {
    if (rows <= 0) return AFX_OK;
    if (!afxk_cepstrum_supported(in, num, ccNum)) return AFX_ERR_UNSUPPORTED;
    if (ccNum <= 16 && (num == 64 || num == 128 || num == 256)) {
        const long long groups = (rows + 15) / 16;
        long long nb = (groups + 3) / 4;
        if (nb > 256 * 16) nb = 256 * 16;
        if (num == 64)
            hipLaunchKernelGGL((k_cepstrum_mfma<64>), dim3((unsigned)nb), dim3(256), 0,
                               (hipStream_t)stream, in, rows, dct, ccNum, pre, out);
        else if (num == 128)
            hipLaunchKernelGGL((k_cepstrum_mfma<128>), dim3((unsigned)nb), dim3(256), 0,
                               (hipStream_t)stream, in, rows, dct, ccNum, pre, out);
        else
            hipLaunchKernelGGL((k_cepstrum_mfma<256>), dim3((unsigned)nb), dim3(256), 0,
                               (hipStream_t)stream, in, rows, dct, ccNum, pre, out);
        AFX_LAUNCH_CHECK("k_cepstrum_mfma");
        return AFX_OK;
    }
    const long long tiles = (rows + 63) / 64;
    const long long blocks = tiles < 256 * 8 ? tiles : 256 * 8;
    const size_t lds = (size_t)64 * (num + 4) * sizeof(float);
    const int nq = (ccNum + 3) / 4;
#define AFX_CEP_LAUNCH(NQ)                                                                        \
    do {                                                                                          \
        if (lds > 48 * 1024)                                                                      \
            AFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_cepstrum<NQ>),           \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));   \
        hipLaunchKernelGGL((k_cepstrum<NQ>), dim3((unsigned)blocks), dim3(256), lds,              \
                           (hipStream_t)stream, in, rows, num, dct, ccNum, pre, out);             \
    } while (0)
    if (nq <= 1) AFX_CEP_LAUNCH(1);
    else if (nq <= 2) AFX_CEP_LAUNCH(2);
    else if (nq <= 4) AFX_CEP_LAUNCH(4);
    else AFX_CEP_LAUNCH(8);
#undef AFX_CEP_LAUNCH
    AFX_LAUNCH_CHECK("k_cepstrum");
    return AFX_OK;
}
