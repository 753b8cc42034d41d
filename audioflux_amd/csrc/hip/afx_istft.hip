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

#include <cstdint>
#include <cstdlib>
#include <mutex>

#include "afx_device.h"
#include "afx_hipcheck.h"
#include "afx_ldsfft.h"
#include "afx_wavefft2048.h"
#include "afx_wavefft_small.h"

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

// ---- n_fft 2048: one wave per frame, overlap-add in the LDS, no frame scratch (round 6) -----------------------------------
//
// The inverse transform of a frame is ONE forward real transform (afx_wavefft2048.h, the function the forward kernels run): with
// Xs = the Hermitian part of the given bins (what the real part of the reference's complex inverse keeps, stft_algorithm.c:304-409),
//     u[k] = Re Xs[k] + Im Xs[k] = (re[k] + re[N-k] + im[k] - im[N-k]) / 2      (a real sequence),   U = FFT(u):
//     x[n] = (Re U[n] + Im U[n]) / N,   x[N-n] = (Re U[n] - Im U[n]) / N,   0 <= n <= N/2
// (the cosine sums of the even part and the sine sums of the odd part, each once).  A lane holds u[2n], u[2n+1], n = 64 n1 + lane;
// the mirrored bins N - 2n - 1 are element .y of register 15 - n1 in lane 63 - lane, N - 2n element .x of register 15 - n1 in lane
// 64 - lane (lane 0: its own register 16 - n1): two cross-lane reads per plane instead of reversed loads.
// Overlap-add: a wave walks a run of consecutive frames of one clip (the ceil(N / hop) - 1 frames before its run first: their tails
// reach into it) with a ring of N floats in the LDS: a sample enters the ring with the caller's value of out[] (the reference adds
// to what is there), takes the frames in ascending order -- the order of the reference's scatter loop (:378-386) and of k_istft_ola,
// so the float32 sums round the same way -- and leaves it, divided by the window-power sum (:389-396), when frame i has been
// added to [i hop, (i + 1) hop).  Per frame 8 N bytes in and 4 hop out (+ 4 hop of out[] read): the [frames, N] scratch round trip
// (8 N bytes) and the second launch are gone.
namespace {

#ifndef AFX_ISTFT_WAVES
#define AFX_ISTFT_WAVES 7
#endif
constexpr int IW = AFX_ISTFT_WAVES;  // waves per workgroup: 25 KB of tables (window w^e, twiddles) + <= 8 KB of window-power sums + 7 x 16.5 KB (exchange image + ring)

__global__ __launch_bounds__(IW * 64) void k_istft_w2048(AfxIstftArgs a, const float2 *__restrict__ tab, int framesPerRun, int runsPerClip) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int N = 2048;
    float *win1 = reinterpret_cast<float *>(smem_raw);  // [N] synthesis window w^e
    const float *__restrict__ win2 = a.win2;             // w^(e+1): read at the clip's ends only
    v2 *tabTw = reinterpret_cast<v2 *>(win1 + N);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v2 *ex = tabTw + afxw::TAB_F2 + wave * afxw::EX_F2;
    float *ring = reinterpret_cast<float *>(tabTw + afxw::TAB_F2 + IW * afxw::EX_F2) + wave * N;
    // window-power sum of a sample every covering frame of which exists (N <= j, j / hop <= T - 1): a function of j mod hop, added
    // in the order of k_istft_ola's loop (ascending frames = descending window positions)
    float *nrmTab = ring + (IW - wave) * N;              // [hop], behind the last wave's ring
    for (int i = threadIdx.x; i < N; i += IW * 64) win1[i] = a.win1[i];
    for (int i = threadIdx.x; i < afxw::TAB_F2; i += IW * 64) tabTw[i] = v2{tab[i].x, tab[i].y};
    for (int t = threadIdx.x; t < a.hop; t += IW * 64) {
        float sum = 0.f;
        for (int k = t + ((N - 1 - t) / a.hop) * a.hop; k >= 0; k -= a.hop) sum += win2[k];
        nrmTab[t] = sum;
    }
    __syncthreads();
    const afxw::Tables tb = {tabTw, tabTw + afxw::TAB_TW1_F2, tabTw + afxw::TAB_TW1_F2 + afxw::TAB_TW2_F2};

    const long long run = (long long)blockIdx.x * IW + wave;
    if (run >= (long long)a.batch * runsPerClip) return;
    const int b = (int)(run / runsPerClip), T = a.timeLength, H = a.hop;
    const int f0 = (int)(run - (long long)b * runsPerClip) * framesPerRun;
    const int f1 = f0 + framesPerRun < T ? f0 + framesPerRun : T;
    const int halo = (N - 1) / H;
    const int fs = f0 > halo ? f0 - halo : 0;
    const long long outLen = (long long)(T - 1) * H + N;
    const long long ownLo = (long long)f0 * H, ownHi = f1 == T ? outLen : (long long)f1 * H;
    float *out = a.out + (long long)b * a.outStride;
    const float scale = 0.5f / (float)N;
    const bool lane0 = lane == 0;

    // samples that enter the ring with a frame: the caller's values where this wave will store, zeros elsewhere
    auto entering = [&](long long j) { return (j >= ownLo && j < ownHi) ? out[j] : 0.f; };
    for (int t = lane; t < N; t += 64) ring[((long long)fs * H + t) & (N - 1)] = entering((long long)fs * H + t);
    // the bins of the frame about to be transformed; the next frame's are requested behind the transform, under the overlap-add
    v2 r[16], m[16];
    auto fetch = [&](int i) {
        const v2 *re2 = reinterpret_cast<const v2 *>(a.re + ((long long)b * T + i) * N);
        const v2 *im2 = reinterpret_cast<const v2 *>(a.im + ((long long)b * T + i) * N);
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            r[n1] = re2[64 * n1 + lane];
            m[n1] = im2[64 * n1 + lane];
        }
    };
    fetch(fs);
    const bool shortHop = H <= 512;  // the samples entering with the next frame fit eight registers per lane: requested a frame ahead

    for (int i = fs; i < f1; ++i) {
        const long long j0 = (long long)i * H;
        float nxt[8];
        if (shortHop && i + 1 < f1) {
#pragma unroll
            for (int t = 0; t < 8; ++t) nxt[t] = lane + 64 * t < H ? entering(j0 + N + lane + 64 * t) : 0.f;
        }
        // 2. the frame's bins -> u
        v2 v[16];
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            // mirrors: .y from lane 63 - lane, .x from lane 64 - lane (lane 0: own register 16 - n1, bin N = bin 0 for n1 = 0)
            const float ry = __shfl(r[15 - n1].y, 63 - lane, 64), my = __shfl(m[15 - n1].y, 63 - lane, 64);
            float rx = __shfl(r[15 - n1].x, (64 - lane) & 63, 64), mx = __shfl(m[15 - n1].x, (64 - lane) & 63, 64);
            if (lane0) {
                rx = r[n1 == 0 ? 0 : 16 - n1].x;
                mx = m[n1 == 0 ? 0 : 16 - n1].x;
            }
            v[n1] = v2{(r[n1].x + rx) + (m[n1].x - mx), (r[n1].y + ry) + (m[n1].y - my)};
        }
        afxw::Bins bn;
        afxw::rfft2048(v, ex, tb, lane, bn);
        if (i + 1 < f1) fetch(i + 1);  // (behind the transform: in flight across it the 64 registers spill)
        // 3. x[n] = (Re U[n] + Im U[n]) / N and its mirror, windowed, into the ring.  (The 0.5 of the Hermitian part rides in scale.)
        // (every output index occurs once per frame, so the slots of a batch of contributions are distinct: its reads, then its
        //  writes -- written one by one the compiler must order each read behind the previous write)
        int at[16];
        float val[16];
        bool ok[16];
        auto put = [&](int e, bool valid, int n, float x) {
            ok[e] = valid;
            at[e] = (int)((j0 + n) & (N - 1));
            val[e] = valid ? (x * scale) * win1[n] : 0.f;
        };
        auto flush = [&](int cnt) {
            float cur[16];
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (e < cnt) cur[e] = ok[e] ? ring[at[e]] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (e < cnt && ok[e]) ring[at[e]] = cur[e] + val[e];
        };
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = lane + 64 * s + 256 * j, e = 4 * j;
                const v2 X = bn.x[s][j], Y = bn.y[s][j];  // U[k], conj(U[1024 - k])
                // lane 0, s = 0 holds k = 0, 256, 512, 768: the partners 768, 512, 256 are its own x slots -- only 1024 is new
                const bool partner = !(lane0 && s == 0 && j > 0);
                put(e, true, k, X.x + X.y);
                put(e + 1, k > 0, (N - k) & (N - 1), X.x - X.y);
                put(e + 2, partner, 1024 - k, Y.x - Y.y);
                put(e + 3, partner && k > 0, (1024 + k) & (N - 1), Y.x + Y.y);
            }
            flush(16);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {  // bins 128 + 256 i and their partners 896 - 256 i (every lane computed them: lane 0 adds)
            const int k = 128 + 256 * q, e = 4 * q;
            put(e, lane0, k, bn.xc[q].x + bn.xc[q].y);
            put(e + 1, lane0, N - k, bn.xc[q].x - bn.xc[q].y);
            put(e + 2, lane0, 1024 - k, bn.yc[q].x - bn.yc[q].y);
            put(e + 3, lane0, 1024 + k, bn.yc[q].x + bn.yc[q].y);
        }
        flush(8);
        wave_lds_order();
        // 4. samples no later frame reaches: [i hop, (i + 1) hop), everything to the clip's end behind its last frame
        if (i >= f0) {
            const int cnt = i == T - 1 ? N : H;
            auto power = [&](long long j) {  // the window-power sum of sample j (k_istft_ola's loop; interior samples: the table)
                if (j >= N && j / H <= T - 1) return nrmTab[(int)(j % H)];
                long long iLo = j >= N ? (j - N) / H + 1 : 0, iHi = j / H;
                if (iHi > T - 1) iHi = T - 1;
                float nrm = 0.f;
                for (long long q = iLo; q <= iHi; ++q) nrm += win2[(int)(j - q * H)];
                return nrm;
            };
            for (int t0 = 0; t0 < cnt; t0 += 512) {  // eight samples per lane at a time: their reads first, then the stores
                float acc[8], nrm[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int t = t0 + lane + 64 * u;
                    acc[u] = t < cnt ? ring[(j0 + t) & (N - 1)] : 0.f;
                    nrm[u] = t < cnt ? power(j0 + t) : 1.f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int t = t0 + lane + 64 * u;
                    if (t < cnt) out[j0 + t] = acc[u] / (nrm[u] < 1e-6f ? 1.f : nrm[u]);
                }
            }
        }
        wave_lds_order();
        // 1'. the next frame's new samples take the slots just stored
        if (i + 1 < f1) {
            if (shortHop) {
#pragma unroll
                for (int t = 0; t < 8; ++t)
                    if (lane + 64 * t < H) ring[(j0 + N + lane + 64 * t) & (N - 1)] = nxt[t];
            } else {
                for (int t = lane; t < H; t += 64) ring[(j0 + N + t) & (N - 1)] = entering(j0 + N + t);
            }
        }
    }
}

// ---- n_fft 4096 (the reference wrapper's default): the same scheme; U = the real transform of the 4096 values u[k] from the wave
// transforms of its even and odd samples (afxw::combine4096, as the forward kernels do).  A lane holds u[4n .. 4n + 3], n = 64 n1 + lane:
// the mirrors 4096 - 4n - c are element 0 of quad 1024 - n (lane 64 - lane; lane 0: its own register 16 - n1) for c = 0 and
// elements 3, 2, 1 of quad 1023 - n (lane 63 - lane, register 15 - n1) for c = 1, 2, 3.
constexpr int IW4 = 4;  // waves per workgroup: 16 KB window + 25 KB twiddles + <= 16 KB window-power sums + 4 x 24.5 KB (exchange image + ring)

__global__ __launch_bounds__(IW4 * 64) void k_istft_w4096(AfxIstftArgs a, const float2 *__restrict__ tab, int framesPerRun, int runsPerClip) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    typedef float v4 __attribute__((ext_vector_type(4)));
    constexpr int N = 4096, NT = afxw::TAB_F2 + 1032;  // the wave tables + W_4096^k, k <= 1024 (padded to a 16-byte multiple)
    float *win1 = reinterpret_cast<float *>(smem_raw);
    const float *__restrict__ win2 = a.win2;
    v2 *tabTw = reinterpret_cast<v2 *>(win1 + N);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v2 *ex = tabTw + NT + wave * afxw::EX_F2;
    float *ring = reinterpret_cast<float *>(tabTw + NT + IW4 * afxw::EX_F2) + wave * N;
    float *nrmTab = ring + (IW4 - wave) * N;
    for (int i = threadIdx.x; i < N; i += IW4 * 64) win1[i] = a.win1[i];
    for (int i = threadIdx.x; i < afxw::TAB_F2 + 1025; i += IW4 * 64) tabTw[i] = v2{tab[i].x, tab[i].y};
    for (int t = threadIdx.x; t < a.hop; t += IW4 * 64) {
        float sum = 0.f;
        for (int k = t + ((N - 1 - t) / a.hop) * a.hop; k >= 0; k -= a.hop) sum += win2[k];
        nrmTab[t] = sum;
    }
    __syncthreads();
    const afxw::Tables tb = {tabTw, tabTw + afxw::TAB_TW1_F2, tabTw + afxw::TAB_TW1_F2 + afxw::TAB_TW2_F2};
    const v2 *tabW4 = tabTw + afxw::TAB_F2;

    const long long run = (long long)blockIdx.x * IW4 + wave;
    if (run >= (long long)a.batch * runsPerClip) return;
    const int b = (int)(run / runsPerClip), T = a.timeLength, H = a.hop;
    const int f0 = (int)(run - (long long)b * runsPerClip) * framesPerRun;
    const int f1 = f0 + framesPerRun < T ? f0 + framesPerRun : T;
    const int halo = (N - 1) / H;
    const int fs = f0 > halo ? f0 - halo : 0;
    const long long outLen = (long long)(T - 1) * H + N;
    const long long ownLo = (long long)f0 * H, ownHi = f1 == T ? outLen : (long long)f1 * H;
    float *out = a.out + (long long)b * a.outStride;
    const float scale = 0.5f / (float)N;
    const bool lane0 = lane == 0;
    auto entering = [&](long long j) { return (j >= ownLo && j < ownHi) ? out[j] : 0.f; };
    for (int t = lane; t < N; t += 64) ring[((long long)fs * H + t) & (N - 1)] = entering((long long)fs * H + t);

    for (int i = fs; i < f1; ++i) {
        const long long j0 = (long long)i * H;
        afxw::Bins be, bo;
        {
            // the frame's bins -> u: even samples (u[4n], u[4n + 2]) and odd samples (u[4n + 1], u[4n + 3]) of every quad
            const v4 *re4 = reinterpret_cast<const v4 *>(a.re + ((long long)b * T + i) * N);
            const v4 *im4 = reinterpret_cast<const v4 *>(a.im + ((long long)b * T + i) * N);
            v4 r[16], m[16];
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {
                r[n1] = re4[64 * n1 + lane];
                m[n1] = im4[64 * n1 + lane];
            }
            v2 ve[16], vo[16];
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {
                const v4 rq = r[15 - n1], mq = m[15 - n1];
                float r0 = __shfl(rq.x, (64 - lane) & 63, 64), m0 = __shfl(mq.x, (64 - lane) & 63, 64);
                if (lane0) {
                    r0 = r[n1 == 0 ? 0 : 16 - n1].x;
                    m0 = m[n1 == 0 ? 0 : 16 - n1].x;
                }
                const float r1 = __shfl(rq.w, 63 - lane, 64), m1 = __shfl(mq.w, 63 - lane, 64);
                const float r2 = __shfl(rq.z, 63 - lane, 64), m2 = __shfl(mq.z, 63 - lane, 64);
                const float r3 = __shfl(rq.y, 63 - lane, 64), m3 = __shfl(mq.y, 63 - lane, 64);
                ve[n1] = v2{(r[n1].x + r0) + (m[n1].x - m0), (r[n1].z + r2) + (m[n1].z - m2)};
                vo[n1] = v2{(r[n1].y + r1) + (m[n1].y - m1), (r[n1].w + r3) + (m[n1].w - m3)};
            }
            afxw::rfft2048(ve, ex, tb, lane, be);
            afxw::rfft2048(vo, ex, tb, lane, bo);
        }
        // x[n] = (Re U[n] + Im U[n]) / N and its mirror N - n, windowed, into the ring: 16 contributions at a time (distinct slots:
        // their reads, then their writes)
        int at[16];
        float val[16];
        bool ok[16];
        int fill = 0;
        auto flush = [&]() {
            float cur[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) cur[e] = ok[e] ? ring[at[e]] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (ok[e]) ring[at[e]] = cur[e] + val[e];
        };
        afxw::combine4096(be, bo, tabW4, lane, [&](int slot, v2 X) {
            const int p = slot >> 2, rr = slot & 3;
            const int kp = p < 8 ? lane + 64 * (p >> 2) + 256 * (p & 3) : 128 + 256 * (p - 8);
            const int bin = afxw::bin4096(slot, lane);
            // lane 0 at p < 4 holds kp = 0, 256, 512, 768: their partners 1024 -+ kp are its own slots of 4 - j (kp = 0: 1024 once);
            // the positions 128 + 256 i are lane 0's alone
            bool valid = p < 8 || lane0;
            if (lane0 && p < 4 && p > 0 && rr >= 2) valid = false;
            if (p < 8 && kp == 0 && rr == 3) valid = false;
            const int e = 2 * (slot & 7);
            ok[e] = valid;
            at[e] = (int)((j0 + bin) & (N - 1));
            val[e] = valid ? ((X.x + X.y) * scale) * win1[bin & (N - 1)] : 0.f;
            const bool mirror = valid && bin > 0 && bin < 2048;
            ok[e + 1] = mirror;
            at[e + 1] = (int)((j0 + N - bin) & (N - 1));
            val[e + 1] = mirror ? ((X.x - X.y) * scale) * win1[(N - bin) & (N - 1)] : 0.f;
            if ((slot & 7) == 7) flush();
            (void)fill;
        });
        wave_lds_order();
        if (i >= f0) {
            const int cnt = i == T - 1 ? N : H;
            auto power = [&](long long j) {
                if (j >= N && j / H <= T - 1) return nrmTab[(int)(j % H)];
                long long iLo = j >= N ? (j - N) / H + 1 : 0, iHi = j / H;
                if (iHi > T - 1) iHi = T - 1;
                float nrm = 0.f;
                for (long long q = iLo; q <= iHi; ++q) nrm += win2[(int)(j - q * H)];
                return nrm;
            };
            for (int t0 = 0; t0 < cnt; t0 += 512) {
                float acc[8], nrm[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int t = t0 + lane + 64 * u;
                    acc[u] = t < cnt ? ring[(j0 + t) & (N - 1)] : 0.f;
                    nrm[u] = t < cnt ? power(j0 + t) : 1.f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int t = t0 + lane + 64 * u;
                    if (t < cnt) out[j0 + t] = acc[u] / (nrm[u] < 1e-6f ? 1.f : nrm[u]);
                }
            }
        }
        wave_lds_order();
        if (i + 1 < f1)
            for (int t = lane; t < H; t += 64) ring[(j0 + N + t) & (N - 1)] = entering(j0 + N + t);
    }
}

// ---- n_fft 1024 / 512: the same scheme on the wave transforms of afx_wavefft_small.h (8 x 8 x 8 in eight registers, 4 x 4 x 4 x 4
// in four).  Their bins come out once each: k = lane + 64 j < N / 4 with its partner N / 2 - k, N / 4 in every lane.
constexpr int ISW = 12;  // waves per workgroup (1024: 12 x 9 KB of exchange image + ring, 11 KB of tables)

template <class F>
__global__ __launch_bounds__(ISW * 64) void k_istft_wsmall(AfxIstftArgs a, const float2 *__restrict__ tab, int framesPerRun, int runsPerClip) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int N = F::N, NR = F::NR, NJ = F::NJ;
    float *win1 = reinterpret_cast<float *>(smem_raw);
    const float *__restrict__ win2 = a.win2;
    v2 *tabTw = reinterpret_cast<v2 *>(win1 + N);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v2 *ex = tabTw + F::TAB_F2 + wave * F::EX_F2;
    float *ring = reinterpret_cast<float *>(tabTw + F::TAB_F2 + ISW * F::EX_F2) + wave * N;
    float *nrmTab = ring + (ISW - wave) * N;
    for (int i = threadIdx.x; i < N; i += ISW * 64) win1[i] = a.win1[i];
    for (int i = threadIdx.x; i < F::TAB_F2; i += ISW * 64) tabTw[i] = v2{tab[i].x, tab[i].y};
    for (int t = threadIdx.x; t < a.hop; t += ISW * 64) {
        float sum = 0.f;
        for (int k = t + ((N - 1 - t) / a.hop) * a.hop; k >= 0; k -= a.hop) sum += win2[k];
        nrmTab[t] = sum;
    }
    __syncthreads();

    const long long run = (long long)blockIdx.x * ISW + wave;
    if (run >= (long long)a.batch * runsPerClip) return;
    const int b = (int)(run / runsPerClip), T = a.timeLength, H = a.hop;
    const int f0 = (int)(run - (long long)b * runsPerClip) * framesPerRun;
    const int f1 = f0 + framesPerRun < T ? f0 + framesPerRun : T;
    const int halo = (N - 1) / H;
    const int fs = f0 > halo ? f0 - halo : 0;
    const long long outLen = (long long)(T - 1) * H + N;
    const long long ownLo = (long long)f0 * H, ownHi = f1 == T ? outLen : (long long)f1 * H;
    float *out = a.out + (long long)b * a.outStride;
    const float scale = 0.5f / (float)N;
    const bool lane0 = lane == 0;
    auto entering = [&](long long j) { return (j >= ownLo && j < ownHi) ? out[j] : 0.f; };
    for (int t = lane; t < N; t += 64) ring[((long long)fs * H + t) & (N - 1)] = entering((long long)fs * H + t);
    v2 r[NR], m[NR];
    auto fetch = [&](int i) {
        const v2 *re2 = reinterpret_cast<const v2 *>(a.re + ((long long)b * T + i) * N);
        const v2 *im2 = reinterpret_cast<const v2 *>(a.im + ((long long)b * T + i) * N);
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            r[q] = re2[64 * q + lane];
            m[q] = im2[64 * q + lane];
        }
    };
    fetch(fs);
    for (int i = fs; i < f1; ++i) {
        const long long j0 = (long long)i * H;
        v2 v[NR];
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            const float ry = __shfl(r[NR - 1 - q].y, 63 - lane, 64), my = __shfl(m[NR - 1 - q].y, 63 - lane, 64);
            float rx = __shfl(r[NR - 1 - q].x, (64 - lane) & 63, 64), mx = __shfl(m[NR - 1 - q].x, (64 - lane) & 63, 64);
            if (lane0) {
                rx = r[q == 0 ? 0 : NR - q].x;
                mx = m[q == 0 ? 0 : NR - q].x;
            }
            v[q] = v2{(r[q].x + rx) + (m[q].x - mx), (r[q].y + ry) + (m[q].y - my)};
        }
        if (i + 1 < f1) fetch(i + 1);
        typename F::B bn;
        F::rfft(v, ex, tabTw, lane, bn);
        int at[4 * NJ + 2];
        float val[4 * NJ + 2];
        bool ok[4 * NJ + 2];
        auto put = [&](int e, bool valid, int n, float x) {
            ok[e] = valid;
            at[e] = (int)((j0 + n) & (N - 1));
            val[e] = valid ? (x * scale) * win1[n] : 0.f;
        };
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int k = lane + 64 * j;
            const v2 X = bn.x[j], Y = bn.y[j];  // U[k], conj(U[N / 2 - k])
            put(4 * j, true, k, X.x + X.y);
            put(4 * j + 1, k > 0, (N - k) & (N - 1), X.x - X.y);
            put(4 * j + 2, true, N / 2 - k, Y.x - Y.y);
            put(4 * j + 3, k > 0, (N / 2 + k) & (N - 1), Y.x + Y.y);
        }
        put(4 * NJ, lane0, N / 4, bn.xm.x + bn.xm.y);
        put(4 * NJ + 1, lane0, 3 * N / 4, bn.xm.x - bn.xm.y);
        float cur[4 * NJ + 2];
#pragma unroll
        for (int e = 0; e < 4 * NJ + 2; ++e) cur[e] = ok[e] ? ring[at[e]] : 0.f;
#pragma unroll
        for (int e = 0; e < 4 * NJ + 2; ++e)
            if (ok[e]) ring[at[e]] = cur[e] + val[e];
        wave_lds_order();
        if (i >= f0) {
            const int cnt = i == T - 1 ? N : H;
            auto power = [&](long long j) {
                if (j >= N && j / H <= T - 1) return nrmTab[(int)(j % H)];
                long long iLo = j >= N ? (j - N) / H + 1 : 0, iHi = j / H;
                if (iHi > T - 1) iHi = T - 1;
                float nrm = 0.f;
                for (long long q = iLo; q <= iHi; ++q) nrm += win2[(int)(j - q * H)];
                return nrm;
            };
            for (int t0 = 0; t0 < cnt; t0 += 256) {
                float acc[4], nrm[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int t = t0 + lane + 64 * u;
                    acc[u] = t < cnt ? ring[(j0 + t) & (N - 1)] : 0.f;
                    nrm[u] = t < cnt ? power(j0 + t) : 1.f;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int t = t0 + lane + 64 * u;
                    if (t < cnt) out[j0 + t] = acc[u] / (nrm[u] < 1e-6f ? 1.f : nrm[u]);
                }
            }
        }
        wave_lds_order();
        if (i + 1 < f1)
            for (int t = lane; t < H; t += 64) ring[(j0 + N + t) & (N - 1)] = entering(j0 + N + t);
    }
}

// ---- n_fft 256: TWO frames per 256-point complex wave transform (the inverse of afx_stft256.hip's trick).  With As, Bs the Hermitian
// parts of the bins of two consecutive frames a, b:  x_a + i x_b = IFFT(As + i Bs) = conj(FFT(conj(As + i Bs))) / N, and
// conj(As + i Bs)[k] = (Re As - Im Bs, -Im As - Re Bs).  The transform's natural-order output gives sample n = lane + 64 q of both frames
// in one lane: frame a is added to the ring, then frame b one hop further -- ascending frames, as everywhere.
__global__ __launch_bounds__(ISW * 64) void k_istft_w256(AfxIstftArgs a, const float2 *__restrict__ tab, int framesPerRun, int runsPerClip) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    typedef afxws::Fft512 F;
    constexpr int N = 256;
    float *win1 = reinterpret_cast<float *>(smem_raw);
    const float *__restrict__ win2 = a.win2;
    v2 *tabTw = reinterpret_cast<v2 *>(win1 + N);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v2 *ex = tabTw + F::TAB_F2 + wave * F::EX_F2;
    float *ring = reinterpret_cast<float *>(tabTw + F::TAB_F2 + ISW * F::EX_F2) + wave * N;
    float *nrmTab = ring + (ISW - wave) * N;
    for (int i = threadIdx.x; i < N; i += ISW * 64) win1[i] = a.win1[i];
    for (int i = threadIdx.x; i < F::TAB_F2; i += ISW * 64) tabTw[i] = v2{tab[i].x, tab[i].y};
    for (int t = threadIdx.x; t < a.hop; t += ISW * 64) {
        float sum = 0.f;
        for (int k = t + ((N - 1 - t) / a.hop) * a.hop; k >= 0; k -= a.hop) sum += win2[k];
        nrmTab[t] = sum;
    }
    __syncthreads();

    const long long run = (long long)blockIdx.x * ISW + wave;
    if (run >= (long long)a.batch * runsPerClip) return;
    const int b = (int)(run / runsPerClip), T = a.timeLength, H = a.hop;
    const int f0 = (int)(run - (long long)b * runsPerClip) * framesPerRun;
    const int f1 = f0 + framesPerRun < T ? f0 + framesPerRun : T;
    const int halo = (N - 1) / H;
    const int fs = f0 > halo ? f0 - halo : 0;
    const long long outLen = (long long)(T - 1) * H + N;
    const long long ownLo = (long long)f0 * H, ownHi = f1 == T ? outLen : (long long)f1 * H;
    float *out = a.out + (long long)b * a.outStride;
    const float scale = 0.5f / (float)N;
    const bool lane0 = lane == 0;
    auto entering = [&](long long j) { return (j >= ownLo && j < ownHi) ? out[j] : 0.f; };
    for (int t = lane; t < N; t += 64) ring[((long long)fs * H + t) & (N - 1)] = entering((long long)fs * H + t);
    auto power = [&](long long j) {
        if (j >= N && j / H <= T - 1) return nrmTab[(int)(j % H)];
        long long iLo = j >= N ? (j - N) / H + 1 : 0, iHi = j / H;
        if (iHi > T - 1) iHi = T - 1;
        float nrm = 0.f;
        for (long long q = iLo; q <= iHi; ++q) nrm += win2[(int)(j - q * H)];
        return nrm;
    };
    // one frame's samples (x[q] = sample lane + 64 q) into the ring, its finished samples out, the next frame's new samples in
    auto overlap_add = [&](int i, const float (&x)[4]) {
        const long long j0 = (long long)i * H;
        float cur[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) cur[q] = ring[(j0 + lane + 64 * q) & (N - 1)];
#pragma unroll
        for (int q = 0; q < 4; ++q) ring[(j0 + lane + 64 * q) & (N - 1)] = cur[q] + (x[q] * scale) * win1[lane + 64 * q];
        wave_lds_order();
        if (i >= f0) {
            const int cnt = i == T - 1 ? N : H;
            float acc[4], nrm[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = lane + 64 * u;
                acc[u] = t < cnt ? ring[(j0 + t) & (N - 1)] : 0.f;
                nrm[u] = t < cnt ? power(j0 + t) : 1.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = lane + 64 * u;
                if (t < cnt) out[j0 + t] = acc[u] / (nrm[u] < 1e-6f ? 1.f : nrm[u]);
            }
        }
        wave_lds_order();
        if (i + 1 < f1)
            for (int t = lane; t < H; t += 64) ring[(j0 + N + t) & (N - 1)] = entering(j0 + N + t);
        wave_lds_order();
    };

    for (int i = fs; i < f1; i += 2) {
        const int ib = i + 1 < f1 ? i + 1 : i;  // an odd run: the last frame rides twice, added once
        const float *ra = a.re + ((long long)b * T + i) * N, *ia = a.im + ((long long)b * T + i) * N;
        const float *rb = a.re + ((long long)b * T + ib) * N, *ib_ = a.im + ((long long)b * T + ib) * N;
        float are[4], aim[4], bre[4], bim[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            are[r] = ra[64 * r + lane];
            aim[r] = ia[64 * r + lane];
            bre[r] = rb[64 * r + lane];
            bim[r] = ib_[64 * r + lane];
        }
        v2 v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            // bin N - k, k = 64 r + lane: register 3 - r of lane 64 - lane (lane 0: its own register 4 - r; r = 0: bin 0 itself)
            const int src = (64 - lane) & 63;
            float mar = __shfl(are[3 - r], src, 64), mai = __shfl(aim[3 - r], src, 64);
            float mbr = __shfl(bre[3 - r], src, 64), mbi = __shfl(bim[3 - r], src, 64);
            if (lane0) {
                mar = are[r == 0 ? 0 : 4 - r];
                mai = aim[r == 0 ? 0 : 4 - r];
                mbr = bre[r == 0 ? 0 : 4 - r];
                mbi = bim[r == 0 ? 0 : 4 - r];
            }
            const float asr = are[r] + mar, asi = aim[r] - mai, bsr = bre[r] + mbr, bsi = bim[r] - mbi;  // 2 x the Hermitian parts
            v[r] = v2{asr - bsi, -asi - bsr};
        }
        F::cfft(v, ex, tabTw, lane);
        __builtin_amdgcn_s_setprio(0);
        wave_lds_order();  // (the natural-order image in `ex` is not needed: the lanes' registers hold their samples)
        float xa[4], xb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            xa[q] = v[q].x;
            xb[q] = -v[q].y;
        }
        overlap_add(i, xa);
        if (ib != i) overlap_add(ib, xb);
    }
}

// Frames per run.  A wave walks one run (+ `halo` frames before it) and every wave slot of the device takes one run per round, so a launch
// costs rounds x (frames per run + halo) frame times: the cheapest (rounds, run length) pair with all runs placed -- 64 clips of 938 frames
// on 1792 slots as 1920 runs of 32 is two rounds of 35 frame times, as 1792 runs of 34 one round of 37.  Runs never shorter than `least`.
int frames_per_run(int batch, int T, long long slots, int halo, int least) {
    long long best = T, bestCost = -1;
    for (int rounds = 1; rounds <= 8; ++rounds) {
        const long long perClip = slots * rounds / batch;  // runs a clip may take
        if (perClip < 1) continue;
        long long fpr = (T + perClip - 1) / perClip;
        if (fpr < least) fpr = least;
        if (fpr > T) fpr = T;
        const long long runs = (long long)batch * ((T + fpr - 1) / fpr);
        const long long cost = ((runs + slots - 1) / slots) * (fpr + halo);
        if (bestCost < 0 || cost < bestCost) {
            bestCost = cost;
            best = fpr;
        }
    }
    return (int)best;
}

// workgroups of `waves` waves and `lds` bytes a CU holds at a time (160 KB of LDS, 32 waves)
int resident_groups(size_t lds, int waves) {
    long long g = lds ? (long long)(163840 / lds) : 1;
    if (g > 32 / waves) g = 32 / waves;
    return g < 1 ? 1 : (int)g;
}

// twiddle tables of the small wave transforms, one device copy per device and size (never freed)
template <class F>
const float2 *small_tables() {
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

template <class F>
int launch_istft_small(const AfxIstftArgs *a, void *stream) {
    const float2 *tab = small_tables<F>();
    if (!tab) return AFX_ERR_UNSUPPORTED;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const size_t lds = sizeof(float) * F::N + sizeof(float2) * (size_t)(F::TAB_F2 + ISW * F::EX_F2) + sizeof(float) * F::N * ISW +
                       sizeof(float) * (size_t)a->hop;
    const long long fpr = frames_per_run(a->batch, a->timeLength, (long long)cus * ISW * resident_groups(lds, ISW), (F::N - 1) / a->hop, 16);
    const long long runsPerClip = (a->timeLength + fpr - 1) / fpr, runs = runsPerClip * a->batch;
    const long long blocks = (runs + ISW - 1) / ISW;
    if (blocks > 0x7fffffffLL) return AFX_ERR_UNSUPPORTED;
    AFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_istft_wsmall<F>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_istft_wsmall<F>, dim3((unsigned)blocks), dim3(ISW * 64), lds, (hipStream_t)stream, *a, tab, (int)fpr, (int)runsPerClip);
    AFX_LAUNCH_CHECK("k_istft_wsmall");
    return AFX_OK;
}

}  // namespace

extern "C" const void *afxk_wave_tables(void);  // afx_stft.hip

// AFX_ERR_UNSUPPORTED: not this kernel's case (afxk_istft then runs the two size-generic launches, which need a->frames)
extern "C" int afxk_istft_fused(const AfxIstftArgs *a, void *stream) {
    if (a->hop < 1 || a->hop > (1 << a->radix2Exp) || a->timeLength < 1 || (reinterpret_cast<uintptr_t>(a->re) & 7) ||
        (reinterpret_cast<uintptr_t>(a->im) & 7))
        return AFX_ERR_UNSUPPORTED;
    if (a->radix2Exp == 12) {
        const float2 *tab4 = static_cast<const float2 *>(afxk_wave_tables());
        if (!tab4 || (reinterpret_cast<uintptr_t>(a->re) & 15) || (reinterpret_cast<uintptr_t>(a->im) & 15)) return AFX_ERR_UNSUPPORTED;
        int dev4 = 0, cus4 = 256;
        if (hipGetDevice(&dev4) == hipSuccess) (void)hipDeviceGetAttribute(&cus4, hipDeviceAttributeMultiprocessorCount, dev4);
        const long long fpr4 = frames_per_run(a->batch, a->timeLength, (long long)cus4 * IW4, 4095 / a->hop, 16);
        const long long rpc4 = (a->timeLength + fpr4 - 1) / fpr4, blocks4 = (rpc4 * a->batch + IW4 - 1) / IW4;
        if (blocks4 > 0x7fffffffLL) return AFX_ERR_UNSUPPORTED;
        const size_t lds4 = sizeof(float) * 4096 + sizeof(float2) * (size_t)(afxw::TAB_F2 + 1032 + IW4 * afxw::EX_F2) + sizeof(float) * 4096 * IW4 +
                            sizeof(float) * (size_t)a->hop;
        if (lds4 > 160 * 1024) return AFX_ERR_UNSUPPORTED;  // (hops beyond ~3000: the size-generic launches)
        AFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_istft_w4096), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4));
        hipLaunchKernelGGL(k_istft_w4096, dim3((unsigned)blocks4), dim3(IW4 * 64), lds4, (hipStream_t)stream, *a, tab4, (int)fpr4, (int)rpc4);
        AFX_LAUNCH_CHECK("k_istft_w4096");
        return AFX_OK;
    }
    if (a->radix2Exp == 10) return launch_istft_small<afxws::Fft1k>(a, stream);
    if (a->radix2Exp == 9) return launch_istft_small<afxws::Fft512>(a, stream);
    if (a->radix2Exp == 8) {
        typedef afxws::Fft512 F8;
        const float2 *tab8 = small_tables<F8>();
        if (!tab8) return AFX_ERR_UNSUPPORTED;
        int dev8 = 0, cus8 = 256;
        if (hipGetDevice(&dev8) == hipSuccess) (void)hipDeviceGetAttribute(&cus8, hipDeviceAttributeMultiprocessorCount, dev8);
        const size_t lds8 = sizeof(float) * 256 + sizeof(float2) * (size_t)(F8::TAB_F2 + ISW * F8::EX_F2) + sizeof(float) * 256 * ISW +
                            sizeof(float) * (size_t)a->hop;
        const long long fpr8 = frames_per_run(a->batch, a->timeLength, (long long)cus8 * ISW * resident_groups(lds8, ISW), 255 / a->hop, 32);
        const long long rpc8 = (a->timeLength + fpr8 - 1) / fpr8, blocks8 = (rpc8 * a->batch + ISW - 1) / ISW;
        if (blocks8 > 0x7fffffffLL) return AFX_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(k_istft_w256, dim3((unsigned)blocks8), dim3(ISW * 64), lds8, (hipStream_t)stream, *a, tab8, (int)fpr8, (int)rpc8);
        AFX_LAUNCH_CHECK("k_istft_w256");
        return AFX_OK;
    }
    if (a->radix2Exp != 11) return AFX_ERR_UNSUPPORTED;
    const float2 *tab = static_cast<const float2 *>(afxk_wave_tables());
    if (!tab) return AFX_ERR_UNSUPPORTED;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    // runs of >= 32 frames (the frames before a run are transformed again for their tails: 3 at hop N / 4), two rounds of waves
    const long long fpr = frames_per_run(a->batch, a->timeLength, (long long)cus * IW, 2047 / a->hop, 16);
    const long long runsPerClip = (a->timeLength + fpr - 1) / fpr, runs = runsPerClip * a->batch;
    const long long blocks = (runs + IW - 1) / IW;
    if (blocks > 0x7fffffffLL) return AFX_ERR_UNSUPPORTED;
    const size_t lds = sizeof(float) * 2048 + sizeof(float2) * (size_t)(afxw::TAB_F2 + IW * afxw::EX_F2) + sizeof(float) * 2048 * IW +
                       sizeof(float) * (size_t)a->hop;
    AFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_istft_w2048), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_istft_w2048, dim3((unsigned)blocks), dim3(IW * 64), lds, (hipStream_t)stream, *a, tab, (int)fpr, (int)runsPerClip);
    AFX_LAUNCH_CHECK("k_istft_w2048");
    return AFX_OK;
}

extern "C" int afxk_istft(const AfxIstftArgs *a, void *stream) {
    if (!a->frames) {
        afxdev_set_error("istft: the size-generic kernels need the frame scratch");
        return AFX_ERR_ARG;
    }
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
