// afx_istft.hip -- inverse short-time Fourier transform, the device side of stftObj_istft
// (reference: src/stft_algorithm.c:304-409).
//
//   k_istft_frames  one workgroup per frame: the fftLength complex bins of the frame (split
//                   re / im planes, as stftObj_stft stores them) are read once, transformed by the
//                   shared in-LDS FFT (inverse = conj . forward . conj, /N), and the real part
//                   times the synthesis window w^e goes to a [frames, N] scratch.
//   k_istft_ola     one thread per output sample: the <= ceil(N/hop) frames that cover the sample
//                   are GATHERED in ascending frame order -- the order the reference's
//                   scatter loop adds them in (:378-386), so the float32 sums round the same way
//                   and no atomics are needed -- together with the window-power normaliser
//                   sum w^(e+1), clamped (< 1e-6 -> 1) and divided out (:389-396).
//
// Both are HBM streaming kernels: per frame 8 N bytes in, 4 N out, then 4 N in and 4 hop out.
#include <hip/hip_runtime.h>

#include "afx_device.h"
#include "afx_hipcheck.h"
#include "afx_ldsfft.h"

namespace {

__global__ void k_istft_frames(AfxIstftArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *s = reinterpret_cast<float2 *>(smem_raw);
    const int r = a.radix2Exp, N = 1 << r;
    const long long frame = blockIdx.x;
    const int tid = threadIdx.x, nth = blockDim.x;
    const float *re = a.re + frame * N, *im = a.im + frame * N;
    for (int i = tid; i < N; i += nth) s[afx_lds_pad(i)] = make_float2(re[i], -im[i]);
    __syncthreads();
    const float2 *tw = reinterpret_cast<const float2 *>(a.twiddle);
    afx_lds_fft_dif_t<true>(s, r, tw, 1, tid, nth);
    const float invN = 1.f / (float)N;
    float *dst = a.frames + frame * N;
    for (int n = tid; n < N; n += nth) {
        const int src = (int)(__brev((unsigned)n) >> (32 - r));
        dst[n] = (s[afx_lds_pad(src)].x * invN) * a.win1[n];
    }
}

__global__ void k_istft_ola(AfxIstftArgs a) {
    const int N = 1 << a.radix2Exp, H = a.hop, T = a.timeLength;
    const long long outLen = (long long)(T - 1) * H + N;
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= outLen) return;
    const int b = blockIdx.y;
    const float *frames = a.frames + (long long)b * T * N;
    float *out = a.out + (long long)b * a.outStride;
    long long iLo = j >= N ? (j - N) / H + 1 : 0;
    long long iHi = j / H;
    if (iHi > T - 1) iHi = T - 1;
    float acc = out[j], nrm = 0.f;
    for (long long i = iLo; i <= iHi; ++i) {
        const int k = (int)(j - i * H);
        acc += frames[i * N + k];
        nrm += a.win2[k];
    }
    if (nrm < 1e-6f) nrm = 1.f;
    out[j] = acc / nrm;
}

}  // namespace

extern "C" int afxk_istft(const AfxIstftArgs *a, void *stream) {
    if (a->radix2Exp < 1 || a->radix2Exp > 14) {
        afxdev_set_error("istft: fftLength 2^%d is outside the supported 2..16384", a->radix2Exp);
        return AFX_ERR_UNSUPPORTED;
    }
    const long long frames = (long long)a->batch * a->timeLength;
    if (frames <= 0) return AFX_OK;
    // k_istft_frames: one workgroup of <= 256 threads per frame (HIP rejects 2^32 or more threads in one dimension);
    // k_istft_ola: one grid row per clip (65 535).  Larger batches go out as several launches of whole clips.
    const long long maxFrames = ((1LL << 32) - 1) / 256;
    if ((frames > maxFrames || a->batch > 65535) && a->batch > 1 && a->timeLength <= maxFrames) {
        long long clipsPer = maxFrames / a->timeLength;
        if (clipsPer > 65535) clipsPer = 65535;
        const long long rowFloats = 1LL << a->radix2Exp;
        for (long long b0 = 0; b0 < a->batch; b0 += clipsPer) {
            AfxIstftArgs s = *a;
            const long long row0 = b0 * a->timeLength;
            s.batch = (int)(a->batch - b0 < clipsPer ? a->batch - b0 : clipsPer);
            s.re = a->re + row0 * rowFloats;
            s.im = a->im + row0 * rowFloats;
            s.frames = a->frames + row0 * rowFloats;
            s.out = a->out + b0 * a->outStride;
            const int st = afxk_istft(&s, stream);
            if (st != AFX_OK) return st;
        }
        return AFX_OK;
    }
    if (frames > 0x7fffffffLL || a->batch > 65535) {
        afxdev_set_error("istft: %lld frames / %d clips in one launch", frames, a->batch);
        return AFX_ERR_UNSUPPORTED;
    }
    const int N = 1 << a->radix2Exp;
    int threads = N / 4;
    if (threads < 64) threads = 64;
    if (threads > 256) threads = 256;
    const size_t lds = (size_t)afx_lds_padded_size(N) * sizeof(float2);
    if (lds > 48 * 1024) {
        AFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_istft_frames),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    hipLaunchKernelGGL(k_istft_frames, dim3((unsigned)frames), dim3(threads), lds, (hipStream_t)stream, *a);
    AFX_LAUNCH_CHECK("k_istft_frames");
    const long long outLen = (long long)(a->timeLength - 1) * a->hop + N;
    const long long blocks = (outLen + 255) / 256;
    if (blocks > 0x7fffffffLL) {
        afxdev_set_error("istft: %lld output samples per clip", outLen);
        return AFX_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(k_istft_ola, dim3((unsigned)blocks, (unsigned)a->batch), dim3(256), 0,
                       (hipStream_t)stream, *a);
    AFX_LAUNCH_CHECK("k_istft_ola");
    return AFX_OK;
}
