// afx_cepstrogram.hip -- cepstrogram kernel ("K8" of SURVEY.md 2b): per frame
//   S = FFT_N(x w) ;  L = ln(max(|S|^2, 1e-16)) over ALL N bins
//   c = Re(IFFT_N(L))                               -> out1 = c[0..N/2]
//   envelope = Re(FFT_N(low-quefrency lifter of c)) -> out2
//   detail   = Re(FFT_N(high-quefrency part of c))  -> out3
// following __cepstrogramObj_spectrogram, src/cepstrogram_algorithm.c:127-298
// (log :219-229, iFFT :232-234, envelope :249-266, details :275-288).
//
// One workgroup per frame; the whole chain lives in LDS (two N-point complex
// buffers), so HBM sees 4*hop bytes in and 3*4*(N/2+1) bytes out per frame instead
// of the reference's ten [T,N] scratch matrices.  Three FFTs instead of four: the
// two real lifter inputs ride one complex transform as l + i d and are separated
// with the conjugate-symmetry identities (exact, no evenness assumption).
#include <hip/hip_runtime.h>

#include "afx_device.h"
#include "afx_hipcheck.h"
#include "afx_ldsfft.h"

namespace {

// the in-place DIF transform in LDS (afx_ldsfft.h) leaves X[k] at bitrev_r(k)
__device__ __forceinline__ int brev(int k, int r) { return (int)(__brev((unsigned)k) >> (32 - r)); }

__global__ void k_cepstrogram(AfxCepstrogramArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int r = a.radix2Exp, N = 1 << r, F = N / 2 + 1;
    float2 *s = reinterpret_cast<float2 *>(smem_raw);
    float2 *t = s + afx_lds_padded_size(N);  // both buffers use skewed addressing (afx_ldsfft.h)
    const float2 *tw = reinterpret_cast<const float2 *>(a.twiddle);
    const int tid = threadIdx.x, nth = blockDim.x;
    const long long frame = blockIdx.x;

    // 1. spectrum of the windowed frame (or the cached spectrum: cepstrogram2)
    if (a.x) {
        const float *x = a.framesPerClip > 0
                             ? a.x + (frame / a.framesPerClip) * a.clipStride +
                                   (frame % a.framesPerClip) * (long long)a.hop
                             : a.x + frame * (long long)a.hop;
        for (int i = tid; i < N; i += nth) s[afx_lds_pad(i)] = make_float2(x[i] * a.window[i], 0.f);
        __syncthreads();
        afx_lds_fft_dif_t<true>(s, r, tw, 1, tid, nth);
        for (int k = tid; k < N; k += nth) {
            const float2 c = s[afx_lds_pad(brev(k, r))];
            if (a.specRe) {
                a.specRe[frame * N + k] = c.x;
                a.specIm[frame * N + k] = c.y;
            }
            float p = c.x * c.x + c.y * c.y;
            if (p < 1e-16f) p = 1e-16f;
            t[afx_lds_pad(k)] = make_float2(logf(p), 0.f);
        }
    } else {
        for (int k = tid; k < N; k += nth) {
            const float re = a.specRe[frame * N + k], im = a.specIm[frame * N + k];
            float p = re * re + im * im;
            if (p < 1e-16f) p = 1e-16f;
            t[afx_lds_pad(k)] = make_float2(logf(p), 0.f);
        }
    }
    __syncthreads();

    // 2. real cepstrum: IFFT(L) = conj(FFT(conj L))/N; L is real, only the real part is kept
    afx_lds_fft_dif_t<true>(t, r, tw, 1, tid, nth);
    const float invN = 1.f / (float)N;
    const int q = a.cepNum;
    for (int n = tid; n < N; n += nth) {
        const float y = t[afx_lds_pad(brev(n, r))].x * invN;
        if (a.out1 && n < F) a.out1[frame * F + n] = y;
        // lifters (cepstrogram_algorithm.c:258-263, :282-283)
        float l = 0.f, d = 0.f;
        if (n <= q) l = y;
        if (n >= q + 1 && n <= N - q) d = y;
        s[afx_lds_pad(n)] = make_float2(l, d);
    }
    __syncthreads();
    // mirrored low-quefrency part: l[N-1-j] = l[j+1], j < cepNum
    for (int j = tid; j < q && j + 1 < N; j += nth) {
        const int dst = N - 1 - j;
        if (dst > q) s[afx_lds_pad(dst)].x = s[afx_lds_pad(j + 1)].x;
    }
    __syncthreads();
    if (!a.out2 && !a.out3) return;

    // 3. one complex FFT carries both real sequences: F = FFT(l) + i FFT(d)
    afx_lds_fft_dif_t<true>(s, r, tw, 1, tid, nth);
    for (int k = tid; k < F; k += nth) {
        const float2 A = s[afx_lds_pad(brev(k, r))];
        const float2 B = s[afx_lds_pad(brev((N - k) & (N - 1), r))];
        if (a.out2) a.out2[frame * F + k] = 0.5f * (A.x + B.x);  // Re FFT(l)[k]
        if (a.out3) a.out3[frame * F + k] = 0.5f * (A.y + B.y);  // Re FFT(d)[k]
    }
}

}  // namespace

extern "C" int afxk_cepstrogram(const AfxCepstrogramArgs *a, void *stream) {
    if (a->radix2Exp < 1 || a->radix2Exp > 13) {
        afxdev_set_error("cepstrogram: fftLength 2^%d is outside the supported 2..8192", a->radix2Exp);
        return AFX_ERR_UNSUPPORTED;
    }
    if (a->timeLength <= 0) return AFX_OK;
    const int N = 1 << a->radix2Exp;
    int threads = N / 2;
    if (threads < 64) threads = 64;
    if (threads > 512) threads = 512;
    const size_t lds = (size_t)2 * afx_lds_padded_size(N) * sizeof(float2);
    if (lds > 48 * 1024) {
        AFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_cepstrogram),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    hipLaunchKernelGGL(k_cepstrogram, dim3((unsigned)a->timeLength), dim3(threads), lds,
                       (hipStream_t)stream, *a);
    AFX_LAUNCH_CHECK("k_cepstrogram");
    return AFX_OK;
}
