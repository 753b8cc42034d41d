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
#include "afx_wavefft2048.h"

#include <mutex>

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
    const long long row = frame * (a.outPitch ? a.outPitch : (long long)a.binCount);
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

// ---- n_fft 2048 / 4096 without a filter bank: one wave per frame ----------------------------
// The STFT object (complex spectrum, all N bins as conjugate mirrors), the linear-scale spectrogram
// (bin slices, every AFX_SPEC_* map) and the three transforms of the reassignment object store
// 8-33 KB per frame and no workgroup needs to see a whole spectrum: each wave runs the register /
// LDS transform of afx_wavefft2048.h on its own frames (n_fft 4096: even / odd samples + combine)
// and stores its bins straight from registers -- lanes hold consecutive bins, so every store
// instruction covers 256 contiguous bytes.  Frames that touch the clip's ends (padding modes) or
// start unaligned are gathered sample by sample through the same index map as the generic kernel.
constexpr int SW = 12;  // waves per workgroup: 148 - 162 registers keep three waves per SIMD (8: 198 -> 12: 208 M frames/s, profiles/r05_ab_other.txt)

// CPLX: mode AFX_SPEC_COMPLEX only (the stores are the spectrum itself); the general maps are
// compiled for n_fft 2048 only (at 4096 their 40 unrolled bin slots exceed the unroller's budget
// and the bins would be indexed dynamically, i.e. live in scratch)
template <int R2, bool CPLX>
__global__ __launch_bounds__(SW * 64) void k_stft_wave(AfxStftArgs a, const float2 *__restrict__ tab,
                                                      int framesPerWave, int vecOk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int N = 1 << R2, M = N / 2;
    constexpr int NTAB = afxw::TAB_F2 + (R2 == 12 ? 1032 : 0);
    float *tabWin = reinterpret_cast<float *>(smem_raw);       // [N]
    v2 *tabTw = reinterpret_cast<v2 *>(tabWin + N);            // afxw tables (| W_4096^k)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v2 *ex = tabTw + NTAB + wave * afxw::EX_F2;
    for (int i = threadIdx.x; i < N; i += SW * 64) tabWin[i] = a.window[i];
    for (int i = threadIdx.x; i < afxw::TAB_F2 + (R2 == 12 ? 1025 : 0); i += SW * 64) tabTw[i] = v2{tab[i].x, tab[i].y};
    __syncthreads();
    const afxw::Tables tb = {tabTw, tabTw + afxw::TAB_TW1_F2, tabTw + afxw::TAB_TW1_F2 + afxw::TAB_TW2_F2};
    const v2 *tabW4 = tabTw + afxw::TAB_F2;

    const long long total = (long long)a.batch * a.timeLength;
    const long long gw = (long long)blockIdx.x * SW + wave;
    long long f = gw * framesPerWave, fEnd = f + framesPerWave;
    if (fEnd > total) fEnd = total;
    const bool two = (a.mode == AFX_SPEC_COMPLEX || a.mode == AFX_SPEC_SQUARE);
    for (; f < fEnd; ++f) {
        const int b = (int)(f / a.timeLength);
        const int t = (int)(f - (long long)b * a.timeLength);
        const float *x = a.x + (long long)b * a.clipStride;
        const long long start = (long long)t * a.hop - a.padLeft;
        const bool inside = start >= 0 && start + N <= a.dataLength;
        const long long row = f * (a.outPitch ? a.outPitch : (long long)a.binCount);
        // bin k of this frame (0 <= k <= N/2) and, for 0 < k < N/2, its mirror N - k = conj
        auto emit = [&](int k, v2 X) {
            float v0, v1;
            const int j = k - a.binLo;
            if (j >= 0 && j < a.binCount) {
                if constexpr (CPLX) {
                    a.outRe[row + j] = X.x;
                    a.outIm[row + j] = X.y;
                } else {
                    map_bin(make_float2(X.x, X.y), a.mode, a.normValue, v0, v1);
                    a.outRe[row + j] = v0;
                    if (two) a.outIm[row + j] = v1;
                }
            }
            const int j2 = N - k - a.binLo;
            if (k > 0 && k < M && j2 >= 0 && j2 < a.binCount) {
                if constexpr (CPLX) {
                    a.outRe[row + j2] = X.x;
                    a.outIm[row + j2] = -X.y;
                } else {
                    map_bin(make_float2(X.x, -X.y), a.mode, a.normValue, v0, v1);
                    a.outRe[row + j2] = v0;
                    if (two) a.outIm[row + j2] = v1;
                }
            }
        };
        if constexpr (R2 == 11) {
            v2 v[16];
            if (inside && vecOk) {
                const v2 *p2 = reinterpret_cast<const v2 *>(x + start);
#pragma unroll
                for (int n1 = 0; n1 < 16; ++n1) v[n1] = p2[64 * n1 + lane];
            } else {
                // sample by sample through the padding index map, staged in the exchange buffer
                float *stage = reinterpret_cast<float *>(ex);
#pragma unroll 1
                for (int i = 0; i < 32; ++i) stage[lane + 64 * i] = fetch(x, start + lane + 64 * i, a);
                wave_lds_order();
#pragma unroll
                for (int n1 = 0; n1 < 16; ++n1) v[n1] = reinterpret_cast<const v2 *>(stage)[64 * n1 + lane];
                wave_lds_order();
            }
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) v[n1] *= reinterpret_cast<const v2 *>(tabWin)[64 * n1 + lane];
            afxw::Bins bn;
            afxw::rfft2048(v, ex, tb, lane, bn);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = lane + 64 * s + 256 * j;
                    emit(k, bn.x[s][j]);
                    emit(1024 - k, v2{bn.y[s][j].x, -bn.y[s][j].y});
                }
            if (lane == 0) {
                emit(128, bn.xc[0]);
                emit(896, v2{bn.yc[0].x, -bn.yc[0].y});
                emit(384, bn.xc[1]);
                emit(640, v2{bn.yc[1].x, -bn.yc[1].y});
            }
        } else {
            typedef float v4 __attribute__((ext_vector_type(4)));
            // even samples, transform, then odd samples (re-read through L1/L2: holding both halves
            // across the first transform costs 32 VGPRs and spills)
            auto load_half = [&](int odd, v2 (&v)[16]) {
                if (inside && vecOk) {
#pragma unroll
                    for (int n1 = 0; n1 < 16; ++n1) {
                        const int n = 64 * n1 + lane;
                        const v4 xv = reinterpret_cast<const v4 *>(x + start)[n];
                        const v4 wv = reinterpret_cast<const v4 *>(tabWin)[n];
                        v[n1] = odd ? v2{xv.y * wv.y, xv.w * wv.w} : v2{xv.x * wv.x, xv.z * wv.z};
                    }
                } else {
                    // (windowed) samples 2 i + odd staged in the exchange buffer: 2048 floats
                    float *stage = reinterpret_cast<float *>(ex);
#pragma unroll 1
                    for (int i = 0; i < 32; ++i) {
                        const int m = 2 * (lane + 64 * i) + odd;
                        stage[lane + 64 * i] = fetch(x, start + m, a) * tabWin[m];
                    }
                    wave_lds_order();
#pragma unroll
                    for (int n1 = 0; n1 < 16; ++n1) v[n1] = reinterpret_cast<const v2 *>(stage)[64 * n1 + lane];
                    wave_lds_order();
                }
            };
            afxw::Bins be, bo;
            {
                v2 v[16];
                load_half(0, v);
                afxw::rfft2048(v, ex, tb, lane, be);
                load_half(1, v);
                afxw::rfft2048(v, ex, tb, lane, bo);
            }
            afxw::combine4096(be, bo, tabW4, lane, [&](int slot, v2 X) {
                if (slot < 32 || lane == 0) emit(afxw::bin4096(slot, lane), X);
            });
        }
    }
}

// ---- temporal features alone (temporal_algorithm.c:138-144): energy, rms and zero-crossing rate of the windowed frames for the
// objects whose bank rows come from a fused kernel without them (every transform size but 2048, and the complex results): one
// wave per frame reads the frame again (its samples are in L2: the bank kernel has just read them) in the sample order of the
// fused kernels' own temporal code -- lane l, step i holds samples 2 (64 i + l), + 1.
__global__ __launch_bounds__(256) void k_temporal(AfxStftArgs a) {
    const int lane = threadIdx.x & 63;
    const long long frame = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (frame >= (long long)a.batch * a.timeLength) return;
    const int N = 1 << a.radix2Exp, M = N >> 1;
    const int b = (int)(frame / a.timeLength);
    const int t = (int)(frame - (long long)b * a.timeLength);
    const float *x = a.x + (long long)b * a.clipStride;
    const long long start = (long long)t * a.hop - a.padLeft;
    float e = 0.f, z = 0.f, prevTop = 0.f;  // prevTop: the second sample of lane 63 one step earlier, wave-uniform
    for (int i0 = 0; i0 < M; i0 += 64) {
        const int i = i0 + lane;
        float vx = 0.f, vy = 0.f;
        if (i < M) {
            vx = fetch(x, start + 2 * i, a) * a.window[2 * i];
            vy = fetch(x, start + 2 * i + 1, a) * a.window[2 * i + 1];
        }
        e = fmaf(vx, vx, e);
        e = fmaf(vy, vy, e);
        float py = __shfl_up(vy, 1, 64);
        if (lane == 0) py = prevTop;
        if (i > 0 && i < M && vx * py < 0.f) z += 1.f;  // sample 0 has no predecessor
        if (i < M && vy * vx < 0.f) z += 1.f;
        prevTop = __shfl(vy, 63, 64);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        e += __shfl_xor(e, o, 64);
        z += __shfl_xor(z, o, 64);
    }
    if (lane == 0) {
        a.energy[frame] = e;
        a.rms[frame] = sqrtf(e / (float)N);
        a.zcr[frame] = (float)((double)z / (double)N);
    }
}

// twiddle tables of the wave kernels, one device copy per device (never freed)
const float2 *wave_tables() {
    static std::mutex mu;
    static float2 *dTab[64] = {nullptr};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    if (!dTab[dev]) {
        const size_t n = (size_t)afxw::TAB_F2 + 1025;
        float *h = static_cast<float *>(calloc(2 * n, sizeof(float)));
        if (!h) return nullptr;
        afxw::fill_tables(h);
        const double PI = 3.14159265358979323846;
        for (int k = 0; k <= 1024; ++k) {
            h[2 * (afxw::TAB_F2 + k)] = (float)cos(-2.0 * PI * (double)k / 4096.0);
            h[2 * (afxw::TAB_F2 + k) + 1] = (float)sin(-2.0 * PI * (double)k / 4096.0);
        }
        float2 *d = nullptr;
        if (hipMalloc(reinterpret_cast<void **>(&d), 2 * n * sizeof(float)) == hipSuccess &&
            hipMemcpy(d, h, 2 * n * sizeof(float), hipMemcpyHostToDevice) == hipSuccess)
            dTab[dev] = d;
        else if (d) (void)hipFree(d);
        free(h);
    }
    return dTab[dev];
}

}  // namespace

// the wave kernels' twiddle tables (afxw::TAB_F2 float2 + 1025 of W_4096) for the other translation units (afx_istft.hip)
extern "C" const void *afxk_wave_tables(void) { return wave_tables(); }

namespace {

template <int R2, bool CPLX>
int launch_stft_wave(const AfxStftArgs *a, const float2 *tab, long long frames, void *stream) {
    constexpr int N = 1 << R2;
    const int vm = R2 == 11 ? 1 : 3;  // float2 / float4 loads: every interior frame start aligned to them
    const int vecOk = ((reinterpret_cast<size_t>(a->x) & (size_t)(4 * vm + 3)) == 0 && (a->hop & vm) == 0 &&
                       (a->padLeft & vm) == 0 && (a->clipStride & vm) == 0)
                          ? 1
                          : 0;
    long long fpw = frames / (256LL * SW * 4);
    fpw = fpw < 1 ? 1 : (fpw > 16 ? 16 : fpw);
    const long long waves = (frames + fpw - 1) / fpw, blocks = (waves + SW - 1) / SW;
    const size_t lds = sizeof(float) * N + sizeof(float2) * (size_t)(afxw::TAB_F2 + (R2 == 12 ? 1032 : 0) + SW * afxw::EX_F2);
    AFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_stft_wave<R2, CPLX>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_stft_wave<R2, CPLX>), dim3((unsigned)blocks), dim3(SW * 64), lds, (hipStream_t)stream, *a, tab,
                       (int)fpw, vecOk);
    AFX_LAUNCH_CHECK("k_stft_wave");
    return AFX_OK;
}

}  // namespace

extern "C" int afxk_stft2k(const AfxStftArgs *a, void *stream);   // afx_melfused2.hip
extern "C" int afxk_stft4k(const AfxStftArgs *a, void *stream);   // afx_melfused4k2.hip
extern "C" int afxk_stft1k(const AfxStftArgs *a, void *stream);   // afx_melfused1k.hip
extern "C" int afxk_stft512(const AfxStftArgs *a, void *stream);  // afx_melfused512.hip
extern "C" int afxk_stft256(const AfxStftArgs *a, void *stream);  // afx_stft256.hip

extern "C" int afxk_temporal(const AfxStftArgs *a, void *stream);

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
    // n_fft 2048: one wave per frame (208 vs 111 M frames/s for the full complex spectrum).  The
    // n_fft 4096 instantiation (<12, true>: two transforms + combine, 40 bin slots per lane) needs
    // 252 VGPRs + 620 B/lane of scratch in 8-wave workgroups (38 M frames/s) or 412 registers in 4-wave
    // workgroups (47 M) and measured SLOWER than the size-generic kernel (58 M) both ways: not dispatched.
    const bool cplx = a->mode == AFX_SPEC_COMPLEX;
    // (temporal features asked for beside the bins -- the linear-scale objects with isTemporal: the wave kernels run without
    //  them and k_temporal follows, instead of the whole call running the size-generic kernel)
    AfxStftArgs bins = *a;
    bins.energy = bins.rms = bins.zcr = nullptr;
    if (a->radix2Exp == 11 && !a->bandStart && a->binLo >= 0 && !afxdev_no_fused()) {
        // real results of frames that lie inside their clips: the headline kernel's transform storing its row (afxk_stft2k:
        // 2.4 x k_stft_wave<11, false> on the dense route's rows); AFX_ERR_UNSUPPORTED = not its case
        if (!cplx) {
            const int st = afxk_stft2k(&bins, stream);
            if (st != AFX_ERR_UNSUPPORTED) return (st == AFX_OK && a->energy) ? afxk_temporal(a, stream) : st;
        }
        if (const float2 *tab = wave_tables()) {
            const int st = cplx ? launch_stft_wave<11, true>(&bins, tab, frames, stream) : launch_stft_wave<11, false>(&bins, tab, frames, stream);
            return (st == AFX_OK && a->energy) ? afxk_temporal(a, stream) : st;
        }
    }
    // n_fft 4096 (the wrapper's default), 1024, 512: the transform of that size's bank kernel storing its spectrum
    // (afx_melfused4k2 / 1k / 512.hip), 256: two frames per 256-point complex wave transform (afx_stft256.hip) -- when every frame
    // lies inside its clip; AFX_ERR_UNSUPPORTED = not its case
    if ((a->radix2Exp == 12 || a->radix2Exp == 10 || a->radix2Exp == 9 || a->radix2Exp == 8) && !a->bandStart && !afxdev_no_fused()) {
        const int st = a->radix2Exp == 12 ? afxk_stft4k(&bins, stream) : a->radix2Exp == 10 ? afxk_stft1k(&bins, stream)
                       : a->radix2Exp == 9 ? afxk_stft512(&bins, stream) : afxk_stft256(&bins, stream);
        if (st != AFX_ERR_UNSUPPORTED) return (st == AFX_OK && a->energy) ? afxk_temporal(a, stream) : st;
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
    // One workgroup per frame, and HIP rejects a launch with 2^32 or more threads in one dimension (16.7 M frames
    // of 256 threads -- 4 477 clips of 30 s at n_fft 512 / hop 128, 8.6 GB of mel output): such a batch goes out as
    // several launches of whole clips.  (A single clip beyond the limit still fails in the launch check below.)
    const long long maxFrames = ((1LL << 32) - 1) / threads;
    if (frames > maxFrames && a->batch > 1 && a->timeLength <= maxFrames) {
        const int clipsPer = (int)(maxFrames / a->timeLength);
        // floats between output rows, as the kernel indexes them: banded rows are packed, bin rows may be pitched
        const long long pitch = a->bandW ? a->bandNum : (a->outPitch ? a->outPitch : (long long)a->binCount);
        for (int b0 = 0; b0 < a->batch; b0 += clipsPer) {
            AfxStftArgs s = *a;
            const long long row0 = (long long)b0 * a->timeLength;
            s.batch = a->batch - b0 < clipsPer ? a->batch - b0 : clipsPer;
            s.x = a->x + (long long)b0 * a->clipStride;
            s.outRe = a->outRe + row0 * pitch;
            if (a->outIm) s.outIm = a->outIm + row0 * pitch;
            if (a->energy) s.energy = a->energy + row0;
            if (a->rms) s.rms = a->rms + row0;
            if (a->zcr) s.zcr = a->zcr + row0;
            hipLaunchKernelGGL(k_stft_generic, dim3((unsigned)((long long)s.batch * s.timeLength)), dim3(threads), lds,
                               (hipStream_t)stream, s);
            AFX_LAUNCH_CHECK("k_stft_generic");
        }
        return AFX_OK;
    }
    hipLaunchKernelGGL(k_stft_generic, dim3((unsigned)frames), dim3(threads), lds,
                       (hipStream_t)stream, *a);
    AFX_LAUNCH_CHECK("k_stft_generic");
    return AFX_OK;
}

extern "C" int afxk_temporal(const AfxStftArgs *a, void *stream) {
    if (!a->x || !a->window || !a->energy || !a->rms || !a->zcr || a->radix2Exp < 1 || a->radix2Exp > 14) return AFX_ERR_ARG;
    const long long frames = (long long)a->batch * a->timeLength;
    if (frames <= 0) return AFX_OK;
    if (frames > 0x7fffffffLL) {
        afxdev_set_error("temporal: %lld frames in one launch", frames);
        return AFX_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(k_temporal, dim3((unsigned)((frames + 3) / 4)), dim3(256), 0, (hipStream_t)stream, *a);
    AFX_LAUNCH_CHECK("k_temporal");
    return AFX_OK;
}
