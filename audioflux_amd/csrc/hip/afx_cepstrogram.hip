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
#include "afx_wavefft_small.h"

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

// ---- N = 2048 and N = 4096: one wave per frame, wave-level real transforms -----------------
// Every sequence of the chain is real, so each transform is built from the 1024-point complex
// transform of afx_wavefft2048.h (registers + two LDS exchanges, no workgroup barrier) instead of
// an N-point complex radix-2 transform in LDS with a barrier per stage pair:
//   S = rfft(x w)                    -> L[k] = ln max(|S[k]|^2, 1e-16), k <= N/2
//   L is real and even (L[N - k] = L[k]), so IFFT(L) = FFT(L) / N is real and even:
//   c = Re rfft(L_even) / N          -> out1 = c[0..N/2]
//   l, d = the two lifter sequences of c (cepstrogram_algorithm.c:258-263, :282-283; c[m] for
//          m > N/2 is c[N - m]: the reference's own value there differs by rounding only)
//   out2 = Re rfft(l), out3 = Re rfft(d)
// For cepNum <= DIRECT_Q (the wrapper's default is 4) the lifter transforms are evaluated in
// closed form instead: l = c on {0..q} and mirrored onto {N-q..N-1}, d = c on {q+1..N-q}, so
//   out2[k] = c[0] + 2 sum_{m=1..q} c[m] cos(2 pi k m / N)
//   out3[k] = L[k] - out2[k] + c[q] cos(2 pi k q / N)      (index N-q is in both sequences;
//                                                            q = 0: out3 = L - c[0])
// with cos(m theta_k) by rotating W_N^k (an error of ~m ulp; q <= 16) -- two transforms per frame.
// Between the transforms the spectrum / cepstrum goes through an (N/2 + 1)-float natural-order
// row in the wave's exchange buffer.  HBM per frame: 4 hop in (frames overlap in L2), 12 (N/2 + 1) out.
// N = 4096 splits every transform into the 2048-point real transforms E, O of the even / odd
// samples: X[k] = E[k] + W_4096^k O[k], X[2048 - k] = conj(E[k] - W_4096^k O[k]); it has the
// closed-form lifters only (larger cepNum takes the size-generic kernel).
// (waves per workgroup 4 / 8 / 12, lifter batches of 10 / 20 bins and the fast logarithm were measured as
// compile-time variants in round 1: profiles/r01_cepstrogram_wave.txt)
constexpr int CW = 8;     // waves per workgroup, N = 2048 (the tables are shared)
constexpr int CW4 = 8;    // N = 4096
constexpr int LNB = 20;   // bins per batch of the closed-form lifters (divides 20)
constexpr int DIRECT_Q = 16;       // largest cepNum of the closed-form lifters

struct CepWArgs {
    const float *x;
    long long clipStride, totalFrames;
    int framesPerClip, hop, framesPerWave, aligned, cepNum;
    const float *win;    // [N]
    const float2 *tab;   // afxw tables: tw1 | tw2 | tw3 ( | W_4096^k, k <= 1024, for N = 4096)
    float *out1, *out2, *out3;
};

__device__ __forceinline__ float log_power(v2 z) {
    float p = z.x * z.x + z.y * z.y;
    if (p < 1e-16f) p = 1e-16f;  // cepstrogram_algorithm.c:219-229
    return logf(p);
}

// closed-form lifter outputs of NB bins: w[i] = W_N^k of the bin, Lk[i] its log power;
// c = the cepstrum row in LDS (c[0..q] are read, the same address in every lane)
template <int NB>
__device__ __forceinline__ void lifters_direct(const float *c, int q, const v2 (&w)[NB], const float (&Lk)[NB],
                                               float (&env)[NB], float (&det)[NB]) {
    v2 z[NB];
    float acc[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        z[i] = v2{1.f, 0.f};
        acc[i] = 0.f;
    }
    for (int m = 1; m <= q; ++m) {
        const float cm = c[m];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            z[i] = cmul(z[i], w[i]);  // W_N^(k m): real part cos(2 pi k m / N)
            acc[i] = fmaf(cm, z[i].x, acc[i]);
        }
    }
    const float c0 = c[0], cq = q >= 1 ? c[q] : 0.f;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        env[i] = c0 + 2.f * acc[i];
        det[i] = Lk[i] - env[i] + cq * z[i].x;
    }
}

__global__ __launch_bounds__(CW * 64) void k_cepstrogram_w2048(CepWArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int N = 2048, F = 1025;
    v2 *tabWin = reinterpret_cast<v2 *>(smem_raw);
    v2 *tabTw = tabWin + 1024;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // (uniform: frame counters and row pointers stay scalar)
    v2 *ex = tabTw + afxw::TAB_F2 + wave * afxw::EX_F2;
    float *row = reinterpret_cast<float *>(ex);  // natural-order row between transforms (1025 floats)
    {
        const float2 *win2 = reinterpret_cast<const float2 *>(a.win);
        for (int i = threadIdx.x; i < 1024; i += CW * 64) tabWin[i] = v2{win2[i].x, win2[i].y};
        for (int i = threadIdx.x; i < afxw::TAB_F2; i += CW * 64) tabTw[i] = v2{a.tab[i].x, a.tab[i].y};
    }
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
    // one value per bin of a transform, in the lane layout of afxw::Bins: slot 8 s + j is bin
    // k = lane + 64 s + 256 j, slot 16 + 8 s + j... see bin_of()
    // -> natural-order row
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
    const bool direct = q <= DIRECT_Q;
    fetch(frame_ptr(f));
    for (; f < fEnd; ++f) {
        v2 v[16];
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) v[n1] = raw[n1] * tabWin[64 * n1 + lane];
        if (f + 1 < fEnd) fetch(frame_ptr(f + 1));  // in flight under the transforms
        afxw::Bins b;
        // 1. spectrum -> log power row
        afxw::rfft2048(v, ex, tb, lane, b);
        float Lk[20];  // slots: 8 s + j -> bin k, 8 s + 4 + j -> bin 1024 - k, 16..19 -> 128, 896, 384, 640
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                Lk[8 * s + j] = log_power(b.x[s][j]);
                Lk[8 * s + 4 + j] = log_power(b.y[s][j]);
            }
        Lk[16] = log_power(b.xc[0]);
        Lk[17] = log_power(b.yc[0]);
        Lk[18] = log_power(b.xc[1]);
        Lk[19] = log_power(b.yc[1]);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = lane + 64 * s + 256 * j;
                row[k] = Lk[8 * s + j];
                row[1024 - k] = Lk[8 * s + 4 + j];
            }
        if (lane == 0) {
            row[128] = Lk[16];
            row[896] = Lk[17];
            row[384] = Lk[18];
            row[640] = Lk[19];
        }
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
        if (!a.out2 && !a.out3) continue;
        to_row(b, [invN](v2 z) { return z.x * invN; });
        wave_lds_order();
        if (direct) {
            // 3'. closed-form lifters; W_2048^k = 2 tab3[k], W_2048^(1024 - k) = -conj(W_2048^k);
            //     slots as in Lk[], LNB at a time (register pressure)
            float *o2 = a.out2 ? a.out2 + f * F : nullptr, *o3 = a.out3 ? a.out3 + f * F : nullptr;
#pragma unroll
            for (int b0 = 0; b0 < 20; b0 += LNB) {
                v2 w[LNB];
                float lk[LNB], env[LNB], det[LNB];
#pragma unroll
                for (int i = 0; i < LNB; ++i) {
                    const int slot = b0 + i;
                    const int kb = slot < 16 ? lane + 64 * (slot >> 3) + 256 * (slot & 3) : 128 + 256 * ((slot - 16) >> 1);
                    const bool partner = slot < 16 ? ((slot >> 2) & 1) : ((slot - 16) & 1);
                    const v2 t = tb.tw3[kb] * 2.f;
                    w[i] = partner ? v2{-t.x, t.y} : t;
                    lk[i] = Lk[slot];
                }
                lifters_direct<LNB>(row, q, w, lk, env, det);
#pragma unroll
                for (int i = 0; i < LNB; ++i) {
                    const int slot = b0 + i;
                    const int kb = slot < 16 ? lane + 64 * (slot >> 3) + 256 * (slot & 3) : 128 + 256 * ((slot - 16) >> 1);
                    const bool partner = slot < 16 ? ((slot >> 2) & 1) : ((slot - 16) & 1);
                    const int k = partner ? 1024 - kb : kb;
                    if (slot < 16 || lane == 0) {
                        if (o2) o2[k] = env[i];
                        if (o3) o3[k] = det[i];
                    }
                }
            }
            wave_lds_order();  // the row is read; the next frame's transform may overwrite it
            continue;
        }
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

// ---- N = 1024 and N = 512: the same chain on the wave transforms of the fused STFT kernels at those sizes (afx_wavefft_small.h;
// round 6 -- the size-generic kernel ran these at 0.08-0.09 of the HBM roofline).  Closed-form lifters only (cepNum <= DIRECT_Q;
// larger cepNum takes the size-generic kernel).  Bin layout: k = lane + 64 j and its partner N/2 - k, j < NJ, + bin N/4 in every lane.
constexpr int CWS = 8;  // waves per workgroup

template <class X>
__global__ __launch_bounds__(CWS * 64) void k_cepstrogram_wsmall(CepWArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int N = X::N, M = X::M, F = M + 1, NR = X::NR, NJ = X::NJ, NB = 2 * NJ + 1;
    v2 *tabWin = reinterpret_cast<v2 *>(smem_raw);
    v2 *tabTw = tabWin + M;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    v2 *ex = tabTw + X::TAB_F2 + wave * X::EX_F2;
    float *row = reinterpret_cast<float *>(ex);  // natural-order row between transforms (F floats)
    {
        const float2 *win2 = reinterpret_cast<const float2 *>(a.win);
        for (int i = threadIdx.x; i < M; i += CWS * 64) tabWin[i] = v2{win2[i].x, win2[i].y};
        for (int i = threadIdx.x; i < X::TAB_F2; i += CWS * 64) tabTw[i] = v2{a.tab[i].x, a.tab[i].y};
    }
    __syncthreads();
    const v2 *tw3 = X::tw3_of(tabTw);  // 0.5 W_N^k, k <= N/4

    const long long gw = (long long)blockIdx.x * CWS + wave;
    long long f = gw * a.framesPerWave, fEnd = f + a.framesPerWave;
    if (fEnd > a.totalFrames) fEnd = a.totalFrames;
    if (f >= fEnd) return;
    auto frame_ptr = [&](long long fr) {
        return a.framesPerClip > 0
                   ? a.x + (fr / a.framesPerClip) * a.clipStride + (fr % a.framesPerClip) * (long long)a.hop
                   : a.x + fr * (long long)a.hop;
    };
    v2 raw[NR];
    auto fetch = [&](const float *px) {
        if (a.aligned) {
            const v2 *p2 = reinterpret_cast<const v2 *>(px);
#pragma unroll
            for (int r = 0; r < NR; ++r) raw[r] = p2[64 * r + lane];
        } else {
#pragma unroll
            for (int r = 0; r < NR; ++r) raw[r] = v2{px[2 * (64 * r + lane)], px[2 * (64 * r + lane) + 1]};
        }
    };
    const int q = a.cepNum;
    const float invN = 1.f / (float)N;
    fetch(frame_ptr(f));
    for (; f < fEnd; ++f) {
        v2 v[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) v[r] = raw[r] * tabWin[64 * r + lane];
        if (f + 1 < fEnd) fetch(frame_ptr(f + 1));  // in flight under the transforms
        typename X::B b;
        // 1. spectrum -> log power row.  Slots: j -> bin k = lane + 64 j, NJ + j -> bin M - k, 2 NJ -> bin N/4
        X::rfft(v, ex, tabTw, lane, b);
        float Lk[NB];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            Lk[j] = log_power(b.x[j]);
            Lk[NJ + j] = log_power(b.y[j]);
        }
        Lk[2 * NJ] = log_power(b.xm);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            row[lane + 64 * j] = Lk[j];
            row[M - lane - 64 * j] = Lk[NJ + j];
        }
        if (lane == 0) row[N / 4] = Lk[2 * NJ];
        wave_lds_order();
        // 2. real cepstrum: rfft of the even extension L[m] = L[N - m]
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int m = 2 * (64 * r + lane);
            v[r] = v2{row[m <= M ? m : N - m], row[m + 1 <= M ? m + 1 : N - m - 1]};
        }
        wave_lds_order();
        X::rfft(v, ex, tabTw, lane, b);
        if (a.out1) {
            float *o1 = a.out1 + f * F;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                o1[lane + 64 * j] = b.x[j].x * invN;
                o1[M - lane - 64 * j] = b.y[j].x * invN;
            }
            if (lane == 0) o1[N / 4] = b.xm.x * invN;
        }
        if (!a.out2 && !a.out3) continue;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            row[lane + 64 * j] = b.x[j].x * invN;
            row[M - lane - 64 * j] = b.y[j].x * invN;
        }
        if (lane == 0) row[N / 4] = b.xm.x * invN;
        wave_lds_order();
        // 3'. closed-form lifters; W_N^k = 2 tw3[k], W_N^(M - k) = -conj(W_N^k)
        {
            v2 w[NB];
            float env[NB], det[NB];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const v2 t = tw3[lane + 64 * j] * 2.f;
                w[j] = t;
                w[NJ + j] = v2{-t.x, t.y};
            }
            w[2 * NJ] = tw3[N / 4] * 2.f;
            lifters_direct<NB>(row, q, w, Lk, env, det);
            float *o2 = a.out2 ? a.out2 + f * F : nullptr, *o3 = a.out3 ? a.out3 + f * F : nullptr;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (o2) {
                    o2[lane + 64 * j] = env[j];
                    o2[M - lane - 64 * j] = env[NJ + j];
                }
                if (o3) {
                    o3[lane + 64 * j] = det[j];
                    o3[M - lane - 64 * j] = det[NJ + j];
                }
            }
            if (lane == 0) {
                if (o2) o2[N / 4] = env[2 * NJ];
                if (o3) o3[N / 4] = det[2 * NJ];
            }
        }
        wave_lds_order();  // the row is read; the next frame's transform may overwrite it
    }
}

// ---- N = 4096 (combine4096 / bin4096: afx_wavefft2048.h) ---------------------------------------
using afxw::bin4096;
using afxw::combine4096;

__global__ __launch_bounds__(CW4 * 64) void k_cepstrogram_w4096(CepWArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int N = 4096, F = 2049;
    typedef float v4 __attribute__((ext_vector_type(4)));
    v4 *tabWin = reinterpret_cast<v4 *>(smem_raw);                // [1024] window quads
    v2 *tabTw = reinterpret_cast<v2 *>(tabWin + 1024);
    v2 *tabW4 = tabTw + afxw::TAB_F2;                             // W_4096^k, k <= 1024 (1032 slots)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // (uniform: frame counters and row pointers stay scalar)
    v2 *ex = tabW4 + 1032 + wave * afxw::EX_F2;
    float *row = reinterpret_cast<float *>(ex);  // natural-order row between transforms (2049 floats)
    {
        const v4 *win4 = reinterpret_cast<const v4 *>(a.win);
        for (int i = threadIdx.x; i < 1024; i += CW4 * 64) tabWin[i] = win4[i];
        for (int i = threadIdx.x; i < afxw::TAB_F2 + 1025; i += CW4 * 64) tabTw[i] = v2{a.tab[i].x, a.tab[i].y};
    }
    __syncthreads();
    const afxw::Tables tb = {tabTw, tabTw + afxw::TAB_TW1_F2, tabTw + afxw::TAB_TW1_F2 + afxw::TAB_TW2_F2};

    const long long gw = (long long)blockIdx.x * CW4 + wave;
    long long f = gw * a.framesPerWave, fEnd = f + a.framesPerWave;
    if (fEnd > a.totalFrames) fEnd = a.totalFrames;
    if (f >= fEnd) return;
    auto frame_ptr = [&](long long fr) {
        return a.framesPerClip > 0
                   ? a.x + (fr / a.framesPerClip) * a.clipStride + (fr % a.framesPerClip) * (long long)a.hop
                   : a.x + fr * (long long)a.hop;
    };
    const int q = a.cepNum;
    for (; f < fEnd; ++f) {
        const float *px = frame_ptr(f);
        v2 ve[16], vo[16];
        // even / odd samples of the windowed frame: lane holds x[4n .. 4n+3], n = 64 n1 + lane
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            const int n = 64 * n1 + lane;
            v4 xv;
            if (a.aligned) xv = reinterpret_cast<const v4 *>(px)[n];
            else xv = v4{px[4 * n], px[4 * n + 1], px[4 * n + 2], px[4 * n + 3]};
            const v4 wv = tabWin[n];
            ve[n1] = v2{xv.x * wv.x, xv.z * wv.z};
            vo[n1] = v2{xv.y * wv.y, xv.w * wv.w};
        }
        afxw::Bins be, bo;
        // 1. spectrum -> log power (kept in registers for the closed-form details) -> row
        afxw::rfft2048(ve, ex, tb, lane, be);
        afxw::rfft2048(vo, ex, tb, lane, bo);
        float Lk[40];
        combine4096(be, bo, tabW4, lane, [&](int slot, v2 X) { Lk[slot] = log_power(X); });
#pragma unroll
        for (int slot = 0; slot < 32; ++slot) row[bin4096(slot, lane)] = Lk[slot];
        if (lane == 0) {
#pragma unroll
            for (int slot = 32; slot < 40; ++slot) row[bin4096(slot, lane)] = Lk[slot];
        }
        wave_lds_order();
        // 2. real cepstrum: transform of the even extension L[m] = L[4096 - m]
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            const int m = 4 * (64 * n1 + lane);
            auto ext = [&](int i) { return row[i <= 2048 ? i : N - i]; };
            ve[n1] = v2{ext(m), ext(m + 2)};
            vo[n1] = v2{ext(m + 1), ext(m + 3)};
        }
        wave_lds_order();
        afxw::rfft2048(ve, ex, tb, lane, be);
        afxw::rfft2048(vo, ex, tb, lane, bo);
        const float invN = 1.f / (float)N;
        float ck[40];
        combine4096(be, bo, tabW4, lane, [&](int slot, v2 X) { ck[slot] = X.x * invN; });
        float *o1 = a.out1 ? a.out1 + f * F : nullptr;
        // only c[0 .. q] is read back (by every lane): bins 0 .. 16 live in lanes 0 .. 16, slot 0
        if (lane <= DIRECT_Q) row[lane] = ck[0];
        if (o1) {
#pragma unroll
            for (int slot = 0; slot < 32; ++slot) o1[bin4096(slot, lane)] = ck[slot];
            if (lane == 0) {
#pragma unroll
                for (int slot = 32; slot < 40; ++slot) o1[bin4096(slot, lane)] = ck[slot];
            }
        }
        if (!a.out2 && !a.out3) {
            wave_lds_order();
            continue;
        }
        wave_lds_order();
        // 3'. closed-form lifters, W_4096^k per slot: k', 2048 - k' -> -conj, 1024 -+ k' from the table
        float *o2 = a.out2 ? a.out2 + f * F : nullptr, *o3 = a.out3 ? a.out3 + f * F : nullptr;
#pragma unroll
        for (int b0 = 0; b0 < 40; b0 += LNB) {  // LNB slots at a time: register pressure
            v2 w[LNB];
            float lk[LNB], env[LNB], det[LNB];
#pragma unroll
            for (int i = 0; i < LNB; ++i) {
                const int slot = b0 + i;
                const int p = slot >> 2, r = slot & 3;
                const int kp = p < 8 ? lane + 64 * (p >> 2) + 256 * (p & 3) : 128 + 256 * (p - 8);
                const v2 wk = tabW4[kp], wp = tabW4[1024 - kp];
                // W^(2048 - k') = -conj(W^k'),  W^(1024 + k') = -conj(W^(1024 - k'))
                w[i] = r == 0 ? wk : r == 1 ? v2{-wk.x, wk.y} : r == 2 ? wp : v2{-wp.x, wp.y};
                lk[i] = Lk[slot];
            }
            lifters_direct<LNB>(row, q, w, lk, env, det);
#pragma unroll
            for (int i = 0; i < LNB; ++i) {
                const int slot = b0 + i;
                if (slot < 32 || lane == 0) {
                    const int k = bin4096(slot, lane);
                    if (o2) o2[k] = env[i];
                    if (o3) o3[k] = det[i];
                }
            }
        }
        wave_lds_order();  // the row is read; the next frame's transform may overwrite it
    }
}

}  // namespace

// host: twiddle tables of the wave kernels, tab[AFX_CEPSTROGRAM_FASTTAB_FLOATS]; N = 4096 appends
// W_4096^k, k <= 1024 (in double, rounded once)
extern "C" void afxk_cepstrogram_fast_tables(float *tab, int fftLength) {
    if (fftLength == 1024) {
        afxws::Fft1k::fill_tables(tab);
        return;
    }
    if (fftLength == 512) {
        afxws::Fft512::fill_tables(tab);
        return;
    }
    afxw::fill_tables(tab);
    if (fftLength == 4096) {
        const double PI = 3.14159265358979323846;
        float *w4 = tab + 2 * afxw::TAB_F2;
        for (int k = 0; k <= 1024; ++k) {
            w4[2 * k] = (float)cos(-2.0 * PI * (double)k / 4096.0);
            w4[2 * k + 1] = (float)sin(-2.0 * PI * (double)k / 4096.0);
        }
    }
}

extern "C" int afxk_cepstrogram(const AfxCepstrogramArgs *a, void *stream) {
    if (a->radix2Exp < 1 || a->radix2Exp > 13) {
        afxdev_set_error("cepstrogram: fftLength 2^%d is outside the supported 2..8192", a->radix2Exp);
        return AFX_ERR_UNSUPPORTED;
    }
    if (a->timeLength <= 0) return AFX_OK;
    const int N = 1 << a->radix2Exp;
    const bool wave2k = N == 2048 && 2 * a->cepNum + 2 < N, wave4k = N == 4096 && a->cepNum <= DIRECT_Q;
    const bool waveS = (N == 1024 || N == 512) && a->cepNum <= DIRECT_Q;
    if ((wave2k || wave4k || waveS) && a->x && !a->specRe && a->fastTab && !afxdev_no_fused()) {
        CepWArgs w;
        w.x = a->x;
        w.clipStride = a->clipStride;
        w.totalFrames = a->timeLength;
        w.framesPerClip = a->framesPerClip;
        w.hop = a->hop;
        // vector loads (float2 / float4 per lane) need every frame start aligned to them
        const int am = (wave2k || waveS) ? 1 : 3;
        w.aligned = ((reinterpret_cast<size_t>(a->x) & (size_t)(4 * am + 3)) == 0 && (a->hop & am) == 0 &&
                     (a->framesPerClip <= 0 || (a->clipStride & am) == 0))
                        ? 1
                        : 0;
        w.cepNum = a->cepNum;
        w.win = a->window;
        w.tab = reinterpret_cast<const float2 *>(a->fastTab);
        w.out1 = a->out1;
        w.out2 = a->out2;
        w.out3 = a->out3;
        const int cw = waveS ? CWS : wave2k ? CW : CW4;
        // enough waves for ~4 workgroups per CU, at most 16 frames per wave
        long long fpw = w.totalFrames / (256LL * cw * 4);
        w.framesPerWave = fpw < 1 ? 1 : (fpw > 16 ? 16 : (int)fpw);
        const long long waves = (w.totalFrames + w.framesPerWave - 1) / w.framesPerWave;
        const long long blocks = (waves + cw - 1) / cw;
        if (waveS) {
            if (N == 1024) {
                using X = afxws::Fft1k;
                const size_t lds = sizeof(float2) * (size_t)(X::M + X::TAB_F2 + CWS * X::EX_F2);
                AFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_cepstrogram_wsmall<X>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(k_cepstrogram_wsmall<X>, dim3((unsigned)blocks), dim3(CWS * 64), lds, (hipStream_t)stream, w);
            } else {
                using X = afxws::Fft512;
                const size_t lds = sizeof(float2) * (size_t)(X::M + X::TAB_F2 + CWS * X::EX_F2);
                AFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_cepstrogram_wsmall<X>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(k_cepstrogram_wsmall<X>, dim3((unsigned)blocks), dim3(CWS * 64), lds, (hipStream_t)stream, w);
            }
            AFX_LAUNCH_CHECK("k_cepstrogram_wsmall");
        } else if (wave2k) {
            const size_t lds = sizeof(float2) * (size_t)(1024 + afxw::TAB_F2 + CW * afxw::EX_F2);
            AFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_cepstrogram_w2048),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(k_cepstrogram_w2048, dim3((unsigned)blocks), dim3(CW * 64), lds, (hipStream_t)stream,
                               w);
            AFX_LAUNCH_CHECK("k_cepstrogram_w2048");
        } else {
            const size_t lds = 16384 + sizeof(float2) * (size_t)(afxw::TAB_F2 + 1032 + CW4 * afxw::EX_F2);
            AFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_cepstrogram_w4096),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(k_cepstrogram_w4096, dim3((unsigned)blocks), dim3(CW4 * 64), lds, (hipStream_t)stream,
                               w);
            AFX_LAUNCH_CHECK("k_cepstrogram_w4096");
        }
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
    // one workgroup per frame; HIP rejects 2^32 or more threads in one dimension: beyond that, several launches of
    // whole clips (or, for one long clip, of whole frames)
    const long long maxFrames = ((1LL << 32) - 1) / threads;
    if (a->timeLength > maxFrames) {
        const long long F = N / 2 + 1;
        long long per = maxFrames;
        if (a->framesPerClip > 0) {
            if (a->framesPerClip > maxFrames) {
                afxdev_set_error("cepstrogram: %d frames per clip in one launch", a->framesPerClip);
                return AFX_ERR_UNSUPPORTED;
            }
            per = maxFrames / a->framesPerClip * a->framesPerClip;
        }
        for (long long f0 = 0; f0 < a->timeLength; f0 += per) {
            AfxCepstrogramArgs s = *a;
            s.timeLength = (int)(a->timeLength - f0 < per ? a->timeLength - f0 : per);
            if (a->x) s.x = a->x + (a->framesPerClip > 0 ? f0 / a->framesPerClip * a->clipStride : f0 * a->hop);
            if (a->specRe) s.specRe = a->specRe + f0 * N;
            if (a->specIm) s.specIm = a->specIm + f0 * N;
            if (a->out1) s.out1 = a->out1 + f0 * F;
            if (a->out2) s.out2 = a->out2 + f0 * F;
            if (a->out3) s.out3 = a->out3 + f0 * F;
            hipLaunchKernelGGL(k_cepstrogram, dim3((unsigned)s.timeLength), dim3(threads), lds, (hipStream_t)stream, s);
            AFX_LAUNCH_CHECK("k_cepstrogram");
        }
        return AFX_OK;
    }
    hipLaunchKernelGGL(k_cepstrogram, dim3((unsigned)a->timeLength), dim3(threads), lds,
                       (hipStream_t)stream, *a);
    AFX_LAUNCH_CHECK("k_cepstrogram");
    return AFX_OK;
}
