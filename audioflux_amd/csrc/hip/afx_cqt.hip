// afx_cqt.hip -- constant-Q transform kernels ("K11-K13" of SURVEY.md 2b).
//
//   k_cqt_octave   one octave of the recursion of _cqtObj_cqt (src/cqt_algorithm.c:
//                  951-989 top octave, :999-1041 lower octaves): rectangular-window,
//                  centre-zero-padded STFT frame -> N-point FFT in LDS -> complex dot with
//                  the octave's sparse spectral kernel (plain product, no conjugate:
//                  __mcdot1, src/vector/flux_complex.c:53-87) -> sqrt(2^k) / sqrt(len)
//                  scaling -> out[t, octave*bpo + j].  One workgroup per frame; the
//                  [T,N] spectra the reference materialises never leave the CU.
//   k_cqt_decimate the "Fast" 2:1 windowed-sinc resampler with isScale
//                  (src/dsp/resample_algorithm.c:430-521): for ratio 1/2 every output is
//                  a fixed 63-tap symmetric FIR at even input positions, h_j = table[256 j].
//   k_cqt_chroma   |Q|^2 or |Q| -> fold bins to chroma (0/1 matrix of
//                  src/filterbank/chroma_filterBank.c:176-264) -> per-frame normalisation
//                  (__mnormalize, src/vector/flux_vector.c:1058-1160), cqt_algorithm.c:542-592.
#include <hip/hip_runtime.h>

#include "afx_device.h"
#include "afx_hipcheck.h"

namespace {

__device__ __forceinline__ int brev(int k, int r) { return (int)(__brev((unsigned)k) >> (32 - r)); }

__global__ void k_cqt_octave(AfxCqtOctaveArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *s = reinterpret_cast<float2 *>(smem_raw);
    const int r = a.radix2Exp, N = 1 << r;
    const float2 *tw = reinterpret_cast<const float2 *>(a.twiddle);
    const int tid = threadIdx.x, nth = blockDim.x;
    const long long frame = blockIdx.x;
    const long long start = frame * (long long)a.hop - (N >> 1);

    // frame of the zero-padded signal (rect window; samples past validLength are dropped by
    // the reference's padded framing, src/stft_algorithm.c:650-653)
    for (int i = tid; i < N; i += nth) {
        const long long p = start + i;
        const float v = (p >= 0 && p < a.validLength) ? a.x[p] : 0.f;
        s[i] = make_float2(v, 0.f);
    }
    __syncthreads();
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
    // banded complex kernel: row j covers bins [kStart[j], kStart[j]+kLen[j])
    for (int j = tid; j < a.rows; j += nth) {
        const int k0 = a.kStart[a.rowBase + j], n = a.kLen[a.rowBase + j];
        const float2 *taps = reinterpret_cast<const float2 *>(a.kTaps) + a.kOff[a.rowBase + j];
        float re = 0.f, im = 0.f;
        for (int q = 0; q < n; ++q) {
            const float2 sv = s[brev(k0 + q, r)];
            const float2 kv = taps[q];
            re += sv.x * kv.x - sv.y * kv.y;
            im += sv.y * kv.x + sv.x * kv.y;
        }
        const float sl = a.scale[a.colBase + j];  // sqrt(len_j), or 1 when scaling is off
        a.outRe[frame * a.num + a.colBase + j] = (re * a.octScale) / sl;
        a.outIm[frame * a.num + a.colBase + j] = (im * a.octScale) / sl;
    }
}

struct Taps32 {
    float h[32];
};

__global__ void k_cqt_decimate(const float *__restrict__ x, int srcLen, float *__restrict__ y,
                               int dstLen, Taps32 tp, float sqrtRatio) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= dstLen) return;
    const int n = 2 * i;
    float acc = 0.f;
    int left = n + 1 < 32 ? n + 1 : 32;
#pragma unroll 8
    for (int j = 0; j < left; ++j) acc += tp.h[j] * x[n - j];
    int right = srcLen - n - 1;
    if (right > 31) right = 31;
#pragma unroll 8
    for (int j = 0; j < right; ++j) acc += tp.h[j + 1] * x[n + j + 1];
    y[i] = acc / sqrtRatio;
}

__global__ void k_cqt_chroma(const float *__restrict__ re, const float *__restrict__ im,
                             long long rows, int num, const unsigned char *__restrict__ fold,
                             int chromaNum, int isMag, int normType, float *__restrict__ out) {
    // one thread per (frame): chromaNum <= 48 accumulators in registers would need static
    // indexing; use one thread per (frame, chroma bin) with a wave-local normalisation instead
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long frame = gid / 64;
    const int c = (int)(gid & 63);
    if (frame >= rows) return;
    float v = 0.f;
    if (c < chromaNum) {
        const float *pr = re + frame * num, *pi = im + frame * num;
        const unsigned char *f = fold + (long long)c * num;
        for (int j = 0; j < num; ++j) {
            if (f[j]) {
                float p = pr[j] * pr[j] + pi[j] * pi[j];
                if (isMag) p = sqrtf(p);
                v += p;
            }
        }
    }
    if (normType != 0) {  // 1 max, 2 min, 3 P2, 4 P1 over the frame's chroma vector
        const float av = fabsf(v);
        float red;
        if (normType == 1) {
            red = (c < chromaNum) ? av : 0.f;
            for (int off = 32; off > 0; off >>= 1) red = fmaxf(red, __shfl_xor(red, off, 64));
        } else if (normType == 2) {
            red = (c < chromaNum) ? av : 3.4e38f;
            for (int off = 32; off > 0; off >>= 1) red = fminf(red, __shfl_xor(red, off, 64));
        } else {
            red = (c < chromaNum) ? (normType == 3 ? av * av : av) : 0.f;
            for (int off = 32; off > 0; off >>= 1) red += __shfl_xor(red, off, 64);
            if (normType == 3) red = sqrtf(red);
        }
        if (red != 0.f) v = v / red;
    }
    if (c < chromaNum) out[frame * chromaNum + c] = v;
}

}  // namespace

extern "C" int afxk_cqt_octave(const AfxCqtOctaveArgs *a, void *stream) {
    if (a->radix2Exp < 1 || a->radix2Exp > 14) return AFX_ERR_UNSUPPORTED;
    if (a->timeLength <= 0) return AFX_OK;
    const int N = 1 << a->radix2Exp;
    int threads = N / 2;
    if (threads < 64) threads = 64;
    if (threads > 256) threads = 256;
    const size_t lds = (size_t)N * sizeof(float2);
    if (lds > 48 * 1024) {
        AFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_cqt_octave),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    hipLaunchKernelGGL(k_cqt_octave, dim3((unsigned)a->timeLength), dim3(threads), lds,
                       (hipStream_t)stream, *a);
    AFX_LAUNCH_CHECK("k_cqt_octave");
    return AFX_OK;
}

extern "C" int afxk_cqt_decimate(const float *x, int srcLen, float *y, int dstLen,
                                 const float *taps32, float sqrtRatio, void *stream) {
    if (dstLen <= 0) return AFX_OK;
    Taps32 tp;
    for (int i = 0; i < 32; ++i) tp.h[i] = taps32[i];
    hipLaunchKernelGGL(k_cqt_decimate, dim3((unsigned)((dstLen + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, x, srcLen, y, dstLen, tp, sqrtRatio);
    AFX_LAUNCH_CHECK("k_cqt_decimate");
    return AFX_OK;
}

extern "C" int afxk_cqt_chroma(const float *re, const float *im, long long rows, int num,
                               const unsigned char *fold, int chromaNum, int isMag, int normType,
                               float *out, void *stream) {
    if (rows <= 0) return AFX_OK;
    if (chromaNum > 64) return AFX_ERR_UNSUPPORTED;
    const long long threads = rows * 64;
    hipLaunchKernelGGL(k_cqt_chroma, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, re, im, rows, num, fold, chromaNum, isMag, normType, out);
    AFX_LAUNCH_CHECK("k_cqt_chroma");
    return AFX_OK;
}
