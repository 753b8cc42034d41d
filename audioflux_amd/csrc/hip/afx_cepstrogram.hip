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
#include "afx_wavefft2048.h"

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

// ---- N = 2048: one wave per frame, four wave-level real transforms --------------------------
// Every sequence of the chain is real, so each transform is the 1024-point complex transform of
// afx_wavefft2048.h (registers + two LDS exchanges, no workgroup barrier) instead of a 2048-point
// complex radix-2 transform in LDS with a barrier per stage pair:
//   S = rfft(x w)                    -> L[k] = ln max(|S[k]|^2, 1e-16), k <= 1024
//   L is real and even (L[2048 - k] = L[k]), so IFFT(L) = FFT(L) / N is real and even:
//   c = Re rfft(L_even) / N          -> out1 = c[0..1024]
//   l, d = the two lifter sequences of c (cepstrogram_algorithm.c:258-263, :282-283; c[m] for
//          m > 1024 is c[2048 - m]: the reference's own value there differs by rounding only)
//   out2 = Re rfft(l), out3 = Re rfft(d)
// Between the transforms the spectrum / cepstrum goes through a 1025-float natural-order row in
// the wave's exchange buffer.  HBM per frame: 4 hop in (frames overlap in L2), 12 (N/2 + 1) out.
constexpr int CW = 8;  // waves per workgroup (2 per SIMD: the 20 KB of tables are shared)

struct CepWArgs {
    const float *x;
    long long clipStride, totalFrames;
    int framesPerClip, hop, framesPerWave, aligned, cepNum;
    const float2 *win2;  // [1024] (w[2n], w[2n+1])
    const float2 *tab;   // afxw tables: tw1 | tw2 | tw3
    float *out1, *out2, *out3;
};

__global__ __launch_bounds__(CW * 64) void k_cepstrogram_w2048(CepWArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int N = 2048, F = 1025;
    v2 *tabWin = reinterpret_cast<v2 *>(smem_raw);
    v2 *tabTw = tabWin + 1024;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v2 *ex = tabTw + afxw::TAB_F2 + wave * afxw::EX_F2;
    float *row = reinterpret_cast<float *>(ex);  // natural-order row between transforms (1025 floats)
    for (int i = threadIdx.x; i < 1024; i += CW * 64) tabWin[i] = v2{a.win2[i].x, a.win2[i].y};
    for (int i = threadIdx.x; i < afxw::TAB_F2; i += CW * 64) tabTw[i] = v2{a.tab[i].x, a.tab[i].y};
    __syncthreads();
    const afxw::Tables tb = {tabTw, tabTw + afxw::TAB_TW1_F2, tabTw + afxw::TAB_TW1_F2 + afxw::TAB_TW2_F2};

    const long long gw = (long long)blockIdx.x * CW + wave;
    long long f = gw * a.framesPerWave, fEnd = f + a.framesPerWave;
    if (fEnd > a.totalFrames) fEnd = a.totalFrames;
    if (f >= fEnd) return;
    auto frame_ptr = [&](long long fr) {
        return a.framesPerClip > 0
                   ? a.x + (fr / a.framesPerClip) * a.clipStride + (fr % a.framesPerClip) * (long long)a.hop
                   : a.x + fr * (long long)a.hop;
    };
    v2 raw[16];
    auto fetch = [&](const float *px) {
        if (a.aligned) {
            const v2 *p2 = reinterpret_cast<const v2 *>(px);
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) raw[n1] = p2[64 * n1 + lane];
        } else {
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) raw[n1] = v2{px[2 * (64 * n1 + lane)], px[2 * (64 * n1 + lane) + 1]};
        }
    };
    // spectrum values of one transform -> natural-order row; val(S[k]) for the bin itself and its partner
    auto to_row = [&](const afxw::Bins &b, auto val) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = lane + 64 * s + 256 * j;
                row[k] = val(b.x[s][j]);
                row[1024 - k] = val(b.y[s][j]);
            }
        if (lane == 0) {
            row[128] = val(b.xc[0]);
            row[896] = val(b.yc[0]);
            row[384] = val(b.xc[1]);
            row[640] = val(b.yc[1]);
        }
    };
    // real parts of one transform -> out[0..1024] of this frame
    auto to_out = [&](const afxw::Bins &b, float *out, float scale) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = lane + 64 * s + 256 * j;
                out[k] = b.x[s][j].x * scale;
                out[1024 - k] = b.y[s][j].x * scale;
            }
        if (lane == 0) {
            out[128] = b.xc[0].x * scale;
            out[896] = b.yc[0].x * scale;
            out[384] = b.xc[1].x * scale;
            out[640] = b.yc[1].x * scale;
        }
    };
    const int q = a.cepNum;
    fetch(frame_ptr(f));
    for (; f < fEnd; ++f) {
        v2 v[16];
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) v[n1] = raw[n1] * tabWin[64 * n1 + lane];
        if (f + 1 < fEnd) fetch(frame_ptr(f + 1));  // in flight under the four transforms
        afxw::Bins b;
        // 1. spectrum -> log power row (cepstrogram_algorithm.c:219-229)
        afxw::rfft2048(v, ex, tb, lane, b);
        to_row(b, [](v2 z) {
            float p = z.x * z.x + z.y * z.y;
            if (p < 1e-16f) p = 1e-16f;
            return logf(p);
        });
        wave_lds_order();
        // 2. real cepstrum: rfft of the even extension L[m] = L[2048 - m]
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            const int m = 2 * (64 * n1 + lane);
            v[n1] = v2{row[m <= 1024 ? m : N - m], row[m + 1 <= 1024 ? m + 1 : N - m - 1]};
        }
        wave_lds_order();
        afxw::rfft2048(v, ex, tb, lane, b);
        const float invN = 1.f / (float)N;
        if (a.out1) to_out(b, a.out1 + f * F, invN);
        to_row(b, [invN](v2 z) { return z.x * invN; });
        wave_lds_order();
        if (!a.out2 && !a.out3) continue;
        // 3. lifters: l keeps c[0..q] and its mirror l[N-1-j] = c[j+1], j < q (:258-263);
        //    d keeps c[q+1 .. N-q] (:282-283)
        v2 vd[16];
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            const int m = 2 * (64 * n1 + lane);
            const float c0 = row[m <= 1024 ? m : N - m], c1 = row[m + 1 <= 1024 ? m + 1 : N - m - 1];
            const bool l0 = m <= q || m >= N - q, l1 = m + 1 <= q || m + 1 >= N - q;
            const bool d0 = m >= q + 1 && m <= N - q, d1 = m + 1 >= q + 1 && m + 1 <= N - q;
            v[n1] = v2{l0 ? c0 : 0.f, l1 ? c1 : 0.f};
            vd[n1] = v2{d0 ? c0 : 0.f, d1 ? c1 : 0.f};
        }
        wave_lds_order();
        if (a.out2) {
            afxw::rfft2048(v, ex, tb, lane, b);
            to_out(b, a.out2 + f * F, 1.f);
        }
        if (a.out3) {
            afxw::rfft2048(vd, ex, tb, lane, b);
            to_out(b, a.out3 + f * F, 1.f);
        }
    }
}

}  // namespace

// host: twiddle tables of the N = 2048 wave kernel, tab[AFX_CEPSTROGRAM_FASTTAB_FLOATS]
extern "C" void afxk_cepstrogram_fast_tables(float *tab) { afxw::fill_tables(tab); }

extern "C" int afxk_cepstrogram(const AfxCepstrogramArgs *a, void *stream) {
    if (a->radix2Exp < 1 || a->radix2Exp > 13) {
        afxdev_set_error("cepstrogram: fftLength 2^%d is outside the supported 2..8192", a->radix2Exp);
        return AFX_ERR_UNSUPPORTED;
    }
    if (a->timeLength <= 0) return AFX_OK;
    const int N = 1 << a->radix2Exp;
    if (N == 2048 && a->x && !a->specRe && a->fastTab && 2 * a->cepNum + 2 < N && !getenv("AFX_NO_FUSED")) {
        CepWArgs w;
        w.x = a->x;
        w.clipStride = a->clipStride;
        w.totalFrames = a->timeLength;
        w.framesPerClip = a->framesPerClip;
        w.hop = a->hop;
        // float2 loads need every frame start 8-byte aligned
        w.aligned = ((reinterpret_cast<size_t>(a->x) & 7) == 0 && (a->hop & 1) == 0 &&
                     (a->framesPerClip <= 0 || (a->clipStride & 1) == 0))
                        ? 1
                        : 0;
        w.cepNum = a->cepNum;
        w.win2 = reinterpret_cast<const float2 *>(a->window);
        w.tab = reinterpret_cast<const float2 *>(a->fastTab);
        w.out1 = a->out1;
        w.out2 = a->out2;
        w.out3 = a->out3;
        // enough waves for ~4 workgroups per CU, at most 16 frames per wave
        long long fpw = w.totalFrames / (256LL * CW * 4);
        w.framesPerWave = fpw < 1 ? 1 : (fpw > 16 ? 16 : (int)fpw);
        const long long waves = (w.totalFrames + w.framesPerWave - 1) / w.framesPerWave;
        const long long blocks = (waves + CW - 1) / CW;
        const size_t lds = sizeof(float2) * (size_t)(1024 + afxw::TAB_F2 + CW * afxw::EX_F2);
        AFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_cepstrogram_w2048),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_cepstrogram_w2048, dim3((unsigned)blocks), dim3(CW * 64), lds, (hipStream_t)stream, w);
        AFX_LAUNCH_CHECK("k_cepstrogram_w2048");
        return AFX_OK;
    }
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
