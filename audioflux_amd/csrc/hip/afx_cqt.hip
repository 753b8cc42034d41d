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

#include <cstdint>
#include <cstdlib>

#include "afx_device.h"
#include "afx_hipcheck.h"
#include "afx_ldsfft.h"
#include "afx_pkmath.h"

namespace {

__device__ __forceinline__ int brev(int k, int r) { return (int)(__brev((unsigned)k) >> (32 - r)); }

__global__ void k_cqt_octave(AfxCqtOctaveArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *s = reinterpret_cast<float2 *>(smem_raw);
    const int r = a.radix2Exp, N = 1 << r;
    const float2 *tw = reinterpret_cast<const float2 *>(a.twiddle);
    const int tid = threadIdx.x, nth = blockDim.x;
    const long long frame = blockIdx.x;
    const long long start = frame * (long long)a.hop - (a.rightPad ? 0 : (N >> 1));
    const float *x = a.x + (long long)blockIdx.y * a.xStride;
    float *outRe = a.outRe + (long long)blockIdx.y * a.outStride;
    float *outIm = a.outIm + (long long)blockIdx.y * a.outStride;

    // frame of the zero-padded signal (rect window; samples past validLength are dropped by
    // the reference's padded framing, src/stft_algorithm.c:650-653)
    for (int i = tid; i < N; i += nth) {
        const long long p = start + i;
        const float v = (p >= 0 && p < a.validLength) ? x[p] : 0.f;
        s[i] = make_float2(v, 0.f);
    }
    __syncthreads();
    afx_lds_fft_dif(s, r, tw, 1, tid, nth);
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
        outRe[frame * a.num + a.colBase + j] = (re * a.octScale) / sl;
        outIm[frame * a.num + a.colBase + j] = (im * a.octScale) / sl;
    }
}


// ---- matrix-core path -----------------------------------------------------------------
// Q[t][j] = sum_n x_t[n] G_j[n]: the octave's frames (a Toeplitz view of the signal) times the
// time-domain image G of the thresholded spectral kernels (AfxCqtOctaveArgs::timeKernel) --
// the same linear map as FFT + sparse spectral product, as one [32 frames] x [N] x [2*rows]
// product per workgroup on v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate).
//
// Workgroup = N/128 waves, all on the same 32 consecutive frames; wave w owns the k range
// [128 w, 128 w + 128): its slice of G (64 MFMA steps x CT column tiles) stays in VGPRs for
// the life of the workgroup, the frames come from a zero-filled LDS copy of the signal
// window (A operand: lane (i = lane&31, kk = lane>>5) reads sample (t0+i) hop + k + kk; the
// copy is skewed by one word per 2^SH samples, hop = odd * 2^SH, so the 32 frames of a read
// fall on 32 distinct banks).  The waves' partial tiles are summed through LDS.
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int CQ_KS = 64;  // MFMA steps per wave (2 samples each)

template <int CT, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_cqt_octave_mfma(AfxCqtOctaveArgs a, int tilesPerClip,
                                                                int SH, int sigWords, int PRE) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *sig = reinterpret_cast<float *>(smem_raw);
    float *red = sig + sigWords;  // [waves][16][64]
    const int N = 1 << a.radix2Exp;
    const int tid = threadIdx.x, lane = tid & 63;
    constexpr int nth = 64 * WAVES;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: k offsets stay on the SALU
    const int i = lane & 31, kk = lane >> 5;
    const int cols = CT * 32;

    // this wave's slice of G: B operand of step ks is G[k = 128 wave + 2 ks + kk][j = lane & 31]
    float breg[CT][CQ_KS];
    {
        const float *g = a.timeKernel + (long long)(128 * wave + kk) * cols + i;
#pragma unroll
        for (int ks = 0; ks < CQ_KS; ++ks)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) breg[ct][ks] = g[(long long)(2 * ks) * cols + ct * 32];
    }
    // output column owned by this lane in each column tile (re block | im block)
    bool colOk[CT], colIm[CT];
    int colOff[CT];
    float colScale[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int col = ct * 32 + i;
        colOk[ct] = col < 2 * a.rows;
        colIm[ct] = col >= a.rows;
        const int j = colIm[ct] ? col - a.rows : col;
        colOff[ct] = a.colBase + (colOk[ct] ? j : 0);
        colScale[ct] = a.scale[colOff[ct]];  // sqrt(len_j), or 1 when scaling is off
    }
    const int S = 31 * a.hop + N;  // samples of the signal window of one tile
    const int ih = i * a.hop;
    const int laneBase = SH ? ih + (ih >> SH) + kk : 2 * (ih + kk);

    // persistent workgroups: tile g = (clip, 32-frame block), g = blockIdx.x, += gridDim.x
    const int totalTiles = tilesPerClip * a.batch;
    // PRE: the window of the NEXT tile is fetched into registers (float4, 16-byte aligned by
    // construction: t0 hop and N/2 are multiples of 4) while this tile's MFMAs run
    constexpr int NV = 5;
    float4 wnd[NV];
    auto fetch = [&](int g) {
        const int clip = g / tilesPerClip, t0 = (g - clip * tilesPerClip) * 32;
        const float *x = a.x + (long long)clip * a.xStride;
        const long long p0 = (long long)t0 * a.hop - (a.rightPad ? 0 : (N >> 1));
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int s = 4 * (tid + u * nth);
            const long long p = p0 + s;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (s < S) {
                if (p >= 0 && p + 3 < a.validLength) {
                    v = *reinterpret_cast<const float4 *>(x + p);
                } else {
                    if (p >= 0 && p < a.validLength) v.x = x[p];
                    if (p + 1 >= 0 && p + 1 < a.validLength) v.y = x[p + 1];
                    if (p + 2 >= 0 && p + 2 < a.validLength) v.z = x[p + 2];
                    if (p + 3 >= 0 && p + 3 < a.validLength) v.w = x[p + 3];
                }
            }
            wnd[u] = v;
        }
    };
    auto skew = [&](int s) { return SH ? s + (s >> SH) : 2 * s; };
    if (PRE && (int)blockIdx.x < totalTiles) fetch(blockIdx.x);
    for (int g = blockIdx.x; g < totalTiles; g += gridDim.x) {
        const int clip = g / tilesPerClip, t0 = (g - clip * tilesPerClip) * 32;
        const float *x = a.x + (long long)clip * a.xStride;
        float *outRe = a.outRe + (long long)clip * a.outStride;
        float *outIm = a.outIm + (long long)clip * a.outStride;
        // zero-filled, skewed copy of the window (src/stft_algorithm.c:650-653: samples past
        // validLength are dropped by the reference's padded framing)
        if (PRE) {
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const int s = 4 * (tid + u * nth);
                if (s < S) {  // up to 3 samples past S land in the buffer's slack
                    const float4 v = wnd[u];
                    sig[skew(s)] = v.x;
                    sig[skew(s + 1)] = v.y;
                    sig[skew(s + 2)] = v.z;
                    sig[skew(s + 3)] = v.w;
                }
            }
            __syncthreads();
            if (g + (int)gridDim.x < totalTiles) fetch(g + gridDim.x);
        } else {
            const long long p0 = (long long)t0 * a.hop - (a.rightPad ? 0 : (N >> 1));
            // (eight independent loads in flight per thread: one load per trip would serialise
            // the whole window on memory latency)
            for (int sb = tid; sb < S; sb += 8 * nth) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int s = sb + u * nth;
                    const long long p = p0 + s;
                    v[u] = (s < S && p >= 0 && p < a.validLength) ? x[p] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int s = sb + u * nth;
                    if (s < S) sig[skew(s)] = v[u];
                }
            }
            __syncthreads();
        }

        f32x16 acc[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
        const int kBase = 128 * wave;
        // opaque every 8 steps: otherwise the 64 loop-invariant read addresses are formed up
        // front into 64 resident VGPRs (occupancy); one v_add per MFMA is free
        int lb = laneBase;
#pragma unroll
        for (int ks = 0; ks < CQ_KS; ++ks) {
            if ((ks & 7) == 0) asm volatile("" : "+v"(lb));
            const int k0 = kBase + 2 * ks;                       // wave-uniform
            const int off = SH ? k0 + (k0 >> SH) : 2 * k0;
            const float av = sig[lb + off];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
                acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, breg[ct][ks], acc[ct], 0, 0, 0);
        }

        // sum the waves' partial tiles; D layout: col = lane & 31, row = (r&3) + 8 (r>>2) + 4 (lane>>5)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[ct][r];
            __syncthreads();
            // thread (wave, lane) sums the 16/WAVES accumulator registers r = wave + q WAVES
            float part[16 / WAVES];
#pragma unroll
            for (int q = 0; q < 16 / WAVES; ++q) {
                const int r = wave + q * WAVES;
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < WAVES; ++w) v += red[(w * 16 + r) * 64 + lane];
                part[q] = v;
            }
            if (colOk[ct]) {
#pragma unroll
                for (int q = 0; q < 16 / WAVES; ++q) {
                    const int r = wave + q * WAVES;
                    const long long frame = t0 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                    if (frame < a.timeLength)
                        (colIm[ct] ? outIm : outRe)[frame * a.num + colOff[ct]] = (part[q] * a.octScale) / colScale[ct];
                }
            }
            __syncthreads();
        }
    }
}

// ---- matrix-core path, independent waves (N <= 512, one column tile) ------------------------
// The whole kernel image G [N][32] sits in LDS (64 KB at N = 512) and every WAVE owns its own
// 32-frame tiles: private signal window, full k range (N/2 MFMA steps, A and B operands by
// ds_read_b32, double-buffered in registers eight steps ahead), results stored straight from
// the accumulator -- no cross-wave reduction and no workgroup barrier after the one-time load
// of G, so the waves of a CU drift apart and the MFMA pipe is fed while others stage or store.
// Two accumulators (even / odd steps) break the dependent-issue chain; they are added at the end.
// SHC > 0: the skew shift is a compile-time constant, so every operand read of the k loop is
// `ds_read vbase offset:imm` -- no address arithmetic between the MFMAs (regular VALU work does
// not overlap the matrix pipe of its own SIMD: measured 112 cycles per MFMA with ~5 VALU per
// step, 72 with none, tools/micro/mfma_peak.hip).  SHC = 0: shift taken from the argument.
template <int NBLK /* N / 16 */, int NV /* float4 window loads per lane */, int SHC>
__global__ __launch_bounds__(NV > 10 ? 256 : 512) void k_cqt_octave_mfma_w(AfxCqtOctaveArgs a, int tilesPerClip,
                                                           int SHrt, int sigWords) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int N = NBLK * 16;
    const int SH = SHC > 0 ? SHC : SHrt;
    float *Bl = reinterpret_cast<float *>(smem_raw);  // [N][32]
    const int tid = threadIdx.x, lane = tid & 63, nth = blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), waves = nth >> 6;
    float *sig = Bl + N * 32 + wave * sigWords;
    const int i = lane & 31, kk = lane >> 5;
    {   // G -> LDS, eight float4 in flight per thread (a load-store loop pays one memory
        // latency per trip: 30 us for the 64 KB)
        const float4 *src = reinterpret_cast<const float4 *>(a.timeKernel);
        float4 *dstl = reinterpret_cast<float4 *>(Bl);
        for (int e0 = tid; e0 < N * 8; e0 += 8 * nth) {
            float4 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = src[min(e0 + u * nth, N * 8 - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (e0 + u * nth < N * 8) dstl[e0 + u * nth] = t[u];
        }
    }
    __syncthreads();

    const bool colOk = i < 2 * a.rows, colIm = i >= a.rows;
    const int colOff = a.colBase + (colOk ? (colIm ? i - a.rows : i) : 0);
    const float colScale = a.scale[colOff];
    const int S = 31 * a.hop + N;
    const int ih = i * a.hop;
    const int laneBase = SH ? ih + (ih >> SH) + kk : 2 * (ih + kk);
    const float *bl = Bl + kk * 32 + i;  // + 64 * step

    const int totalTiles = tilesPerClip * a.batch;
    const int stride = gridDim.x * waves;
    float4 wnd[NV];
    auto fetch = [&](int g) {
        const int clip = g / tilesPerClip, t0 = (g - clip * tilesPerClip) * 32;
        const float *x = a.x + (long long)clip * a.xStride;
        const long long p0 = (long long)t0 * a.hop - (a.rightPad ? 0 : (N >> 1));
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int s = 4 * (lane + 64 * u);
            const long long p = p0 + s;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (s < S) {
                if (p >= 0 && p + 3 < a.validLength) {
                    v = *reinterpret_cast<const float4 *>(x + p);
                } else {
                    if (p >= 0 && p < a.validLength) v.x = x[p];
                    if (p + 1 >= 0 && p + 1 < a.validLength) v.y = x[p + 1];
                    if (p + 2 >= 0 && p + 2 < a.validLength) v.z = x[p + 2];
                    if (p + 3 >= 0 && p + 3 < a.validLength) v.w = x[p + 3];
                }
            }
            wnd[u] = v;
        }
    };
    auto skew = [&](int s) { return SH ? s + (s >> SH) : 2 * s; };
    int g = blockIdx.x * waves + wave;
    if (g < totalTiles) fetch(g);
    for (; g < totalTiles; g += stride) {
        const int clip = g / tilesPerClip, t0 = (g - clip * tilesPerClip) * 32;
        wave_lds_order();  // the MFMA reads of the previous tile are done
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int s = 4 * (lane + 64 * u);
            if (s < S) {  // up to 3 samples past S land in the buffer's slack
                sig[skew(s)] = wnd[u].x;
                sig[skew(s + 1)] = wnd[u].y;
                sig[skew(s + 2)] = wnd[u].z;
                sig[skew(s + 3)] = wnd[u].w;
            }
        }
        wave_lds_order();
        if (g + stride < totalTiles) fetch(g + stride);

        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
        float av[2][8], bv[2][8];
        auto load_block = [&](int blk, float (&ao)[8], float (&bo)[8]) {
            // opaque per block: the 256 loop-invariant A addresses would otherwise be formed
            // once, outside the tile loop, and live in (spilled) VGPRs
            int lb = laneBase;
            if (SHC == 0) asm volatile("" : "+v"(lb));
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8) {
                const int k0 = 16 * blk + 2 * s8;  // compile-time after unrolling
                ao[s8] = sig[lb + (SHC > 0 ? k0 + (k0 >> SHC) : SH ? k0 + (k0 >> SH) : 2 * k0)];
                bo[s8] = bl[32 * k0];
            }
        };
        load_block(0, av[0], bv[0]);
#pragma unroll
        for (int blk = 0; blk < NBLK; ++blk) {
            // the scheduler must not pull later blocks' reads up here (512 live operands)
            __builtin_amdgcn_sched_barrier(0);
            if (blk + 1 < NBLK) load_block(blk + 1, av[(blk + 1) & 1], bv[(blk + 1) & 1]);
#pragma unroll
            for (int s8 = 0; s8 < 8; s8 += 2) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[blk & 1][s8], bv[blk & 1][s8], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[blk & 1][s8 + 1], bv[blk & 1][s8 + 1], acc1, 0, 0, 0);
            }
            // issue order inside the block: one MFMA, then the address adds and the two operand
            // reads of one step of the NEXT block -- the reads land ~8 MFMAs (512 cycles) ahead
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
                if (SHC == 0) __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);  // VALU (addresses)
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // DS read
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // D layout: col = lane & 31, row = (r&3) + 8 (r>>2) + 4 (lane>>5)
        if (colOk) {
            float *dst = (colIm ? a.outIm : a.outRe) + (long long)clip * a.outStride + colOff;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long frame = t0 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                if (frame < a.timeLength) dst[frame * a.num] = ((acc0[r] + acc1[r]) * a.octScale) / colScale;
            }
        }
    }
}

// ---- cqhc / deconv (src/cqt_algorithm.c:662-781) ---------------------------------------------
// Per frame: the num magnitudes, zero padded to M = ceil_pow2(2 num), go through FFT_M; the
// magnitude of that spectrum back through an inverse FFT gives the "timbre" cepstrum-like
// sequence (real part), the spectrum divided by its magnitude gives the "pitch" sequence.
//   cqhc  : out[j] = timbre[round(bpo log2(j + 1))], j < hcNum     (idx precomputed on the host)
//   deconv: timbre[0..num), pitch[0..num)
// One workgroup per frame, radix-2 DIF in LDS (M is 256 for 84 bins: a few kFLOP per frame).
__device__ __forceinline__ void lds_fft_dif(float2 *s, int r, const float2 *tw, int tid, int nth) {
    afx_lds_fft_dif(s, r, tw, 1, tid, nth);
}

__global__ void k_cqt_deconv(const float *__restrict__ in, long long rows, int num, int r,
                             const float2 *__restrict__ tw, const int *__restrict__ hcIdx, int hcNum,
                             float *__restrict__ outTimbre, float *__restrict__ outPitch,
                             float *__restrict__ outHc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int M = 1 << r;
    float2 *s = reinterpret_cast<float2 *>(smem_raw);  // spectrum, then the pitch transform
    float2 *t = s + M;                                  // magnitude transform
    const int tid = threadIdx.x, nth = blockDim.x;
    const long long frame = blockIdx.x;
    const float *x = in + frame * num;
    for (int i = tid; i < M; i += nth) s[i] = make_float2(i < num ? x[i] : 0.f, 0.f);
    __syncthreads();
    lds_fft_dif(s, r, tw, tid, nth);  // X[k] at s[brev(k)]
    const float invM = 1.f / (float)M;
    for (int k = tid; k < M; k += nth) {
        const float2 c = s[brev(k, r)];
        const float mag = sqrtf(c.x * c.x + c.y * c.y);           // __vcabs
        t[k] = make_float2(mag, 0.f);                              // conj of a real value
    }
    __syncthreads();
    if (outPitch) {  // X / max(|X|, 1e-16), conjugated for the inverse transform
        for (int k = tid; k < M; k += nth) {
            const int b = brev(k, r);
            if (b >= k) {  // in-place bit-reversal permutation: swap pairs once
                const float2 ck = s[b], cb = s[k];
                float mk = t[k].x, mb = t[b].x;
                if (mk < 1e-16f) mk = 1e-16f;
                if (mb < 1e-16f) mb = 1e-16f;
                s[k] = make_float2(ck.x / mk, -(ck.y / mk));
                s[b] = make_float2(cb.x / mb, -(cb.y / mb));
            }
        }
        __syncthreads();
    }
    lds_fft_dif(t, r, tw, tid, nth);  // IFFT(mag)[n] = conj(FFT(mag))[n] / M: real part Re(.)/M
    if (outTimbre)
        for (int n = tid; n < num; n += nth) outTimbre[frame * num + n] = t[brev(n, r)].x * invM;
    if (outHc)
        for (int j = tid; j < hcNum; j += nth) outHc[frame * hcNum + j] = t[brev(hcIdx[j], r)].x * invM;
    if (outPitch) {
        lds_fft_dif(s, r, tw, tid, nth);
        for (int n = tid; n < num; n += nth) outPitch[frame * num + n] = s[brev(n, r)].x * invM;
    }
}

struct Taps32 {
    float h[32];
};

// 1024 outputs per workgroup, 4 consecutive outputs per thread.  The input window is staged in LDS split
// into even and odd samples, XE[k] = x[2 (i0-15+k)] and XO[k] = x[2 (i0-16+k) + 1] (zero outside the signal:
// a dropped tap and a zero tap add the same), so that x[2i -/+ j] for consecutive outputs are consecutive
// words and the first word a thread needs of either array sits at a 16-byte boundary: its 34 + 35 operands
// are 9 + 9 ds_read_b128 (a window whose first word is unused gets narrowed by the compiler into unaligned
// ds_read2 pieces at half the LDS rate).  Taps are applied in the reference's order (left taps j = 0..31 at
// x[2i - j], then right taps j = 1..31 at x[2i + j]) as one fma chain; the four results leave as one
// 16-byte store.
constexpr int DEC_OUT = 1024, DEC_LDS = DEC_OUT + 36;

template <bool PAIRS>
__global__ __launch_bounds__(256) void k_cqt_decimate(const float *__restrict__ x, int srcLen,
                                                      long long xStride, float *__restrict__ y,
                                                      int dstLen, long long yStride, Taps32 tp,
                                                      float sqrtRatio) {
    __shared__ __attribute__((aligned(16))) float XE[DEC_LDS];
    __shared__ __attribute__((aligned(16))) float XO[DEC_LDS];
    const int tid = threadIdx.x;
    const int i0 = blockIdx.x * DEC_OUT;
    x += (long long)blockIdx.y * xStride;
    y += (long long)blockIdx.y * yStride;
    auto at = [&](long long s) { return (s >= 0 && s < srcLen) ? x[s] : 0.f; };
    if (PAIRS) {
        // s_k = 2 (i0 - 15 + k): XE[k] = x[s_k], XO[k] = x[s_k - 1].  For odd k, s_k is a multiple of 4 (i0 is)
        // and one 16-byte load gives XE[k], XO[k+1], XE[k+1], XO[k+2]
        for (int q = tid; q <= (DEC_LDS - 2) / 2; q += 256) {
            if (q == 0) {
                const long long s0 = 2LL * (i0 - 15);
                XE[0] = at(s0);
                XO[0] = at(s0 - 1);
                XO[1] = at(s0 + 1);
            } else {
                const int k = 2 * q - 1;
                const long long s = 2LL * (i0 - 15 + k);
                float4 v;
                if (s >= 0 && s + 3 < srcLen) {
                    v = *reinterpret_cast<const float4 *>(x + s);
                } else {
                    v = make_float4(at(s), at(s + 1), at(s + 2), at(s + 3));
                }
                XE[k] = v.x;
                XE[k + 1] = v.z;
                if (k + 1 < DEC_LDS) XO[k + 1] = v.y;
                if (k + 2 < DEC_LDS) XO[k + 2] = v.w;
            }
        }
    } else {
        for (int k = tid; k < DEC_LDS; k += 256) {
            const long long s = 2LL * (i0 - 15 + k);
            XE[k] = at(s);
            XO[k] = at(s - 1);
        }
    }
    __syncthreads();
    // outputs i = i0 + 4 tid + q; x[2 (i + d)] is XE[4 tid + q + d + 15], x[2 (i + d) + 1] is XO[4 tid + q + d + 16]
    // (hand-issued: hipcc splits these float4 reads into ds_read2_b32 / ds_read2_b64 pieces -- half the LDS rate
    // and 2-way bank conflicts, SQ_LDS_BANK_CONFLICT was 65 % of the kernel's LDS cycles)
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 ev[9], ov[9];
    const float *pe = XE + 4 * tid, *po = XO + 4 * tid;
    // hand-issued aligned ds_read_b128 (hipcc narrows these reads to unaligned ds_read2 pieces: 65 % of the LDS
    // cycles were bank conflicts)
    RD128_P(ev[0], pe, 0);   RD128_P(ov[0], po, 0);   RD128_P(ev[1], pe, 16);  RD128_P(ov[1], po, 16);
    RD128_P(ev[2], pe, 32);  RD128_P(ov[2], po, 32);  RD128_P(ev[3], pe, 48);  RD128_P(ov[3], po, 48);
    RD128_P(ev[4], pe, 64);  RD128_P(ov[4], po, 64);  RD128_P(ev[5], pe, 80);  RD128_P(ov[5], po, 80);
    RD128_P(ev[6], pe, 96);  RD128_P(ov[6], po, 96);  RD128_P(ev[7], pe, 112); RD128_P(ov[7], po, 112);
    RD128_P(ev[8], pe, 128); RD128_P(ov[8], po, 128);
    LDS_WAIT_N(0);
    float E[36], O[36];
#pragma unroll
    for (int b = 0; b < 9; ++b) {
        PIN(ev[b]);  // the asm results are only defined after the wait above: pin the uses behind it
        PIN(ov[b]);
        E[4 * b] = ev[b].x; E[4 * b + 1] = ev[b].y; E[4 * b + 2] = ev[b].z; E[4 * b + 3] = ev[b].w;
        O[4 * b] = ov[b].x; O[4 * b + 1] = ov[b].y; O[4 * b + 2] = ov[b].z; O[4 * b + 3] = ov[b].w;
    }
    float r[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 32; ++j) {  // x[2i - j]: even j -> x[2 (i - j/2)], odd j -> x[2 (i - (j+1)/2) + 1]
            const float v = (j & 1) ? O[q + 16 - (j + 1) / 2] : E[q + 15 - j / 2];
            acc = __fmaf_rn(tp.h[j], v, acc);
        }
#pragma unroll
        for (int j = 1; j < 32; ++j) {  // x[2i + j]: even j -> x[2 (i + j/2)], odd j -> x[2 (i + (j-1)/2) + 1]
            const float v = (j & 1) ? O[q + 16 + (j - 1) / 2] : E[q + 15 + j / 2];
            acc = __fmaf_rn(tp.h[j], v, acc);
        }
        r[q] = acc / sqrtRatio;
    }
    const int i = i0 + 4 * tid;
    if (PAIRS && i + 3 < dstLen) {  // (PAIRS: the launcher checked that y rows are 16-byte aligned)
        *reinterpret_cast<float4 *>(y + i) = make_float4(r[0], r[1], r[2], r[3]);
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (i + q < dstLen) y[i + q] = r[q];
    }
}

constexpr int CH_FRAMES = 64;  // frames per workgroup

// ---- chroma (cqt_algorithm.c:484-597): |Q|^2 (or |Q|) -> 0/1 fold matrix product -> per-frame normalisation ----
// Size-generic form (num > 255 bins or > 64 classes, e.g. 36 bins / octave x 8 octaves): thread (frame, class)
// scans its row of the 0/1 matrix and adds the flagged bins in ascending order -- the order of the matrix
// product, so the same bits as the list kernel below -- then the frame's chroma vector is normalised in LDS.
__global__ __launch_bounds__(256) void k_cqt_chroma_scan(const float *__restrict__ re, const float *__restrict__ im,
                                                         long long rows, int num,
                                                         const unsigned char *__restrict__ fold, int chromaNum,
                                                         int isMag, int normType, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *cv = reinterpret_cast<float *>(smem_raw);          // [CH_FRAMES][chromaNum]
    const int tid = threadIdx.x;
    const long long f0 = (long long)blockIdx.x * CH_FRAMES;
    const int nf = rows - f0 < CH_FRAMES ? (int)(rows - f0) : CH_FRAMES;
    for (int it = tid; it < nf * chromaNum; it += 256) {
        const int f = it / chromaNum, c = it - f * chromaNum;
        const float *pr = re + (f0 + f) * num, *pi = im + (f0 + f) * num;
        const unsigned char *fl = fold + (long long)c * num;
        float v = 0.f;
        for (int j = 0; j < num; ++j)
            if (fl[j]) {
                float p = __fmaf_rn(pr[j], pr[j], pi[j] * pi[j]);
                if (isMag) p = sqrtf(p);
                v += p;
            }
        cv[it] = v;
    }
    __syncthreads();
    float *po = out + f0 * chromaNum;
    for (int it = tid; it < nf * chromaNum; it += 256) {
        const int f = it / chromaNum;
        float v = cv[it];
        if (normType != 0) {
            const float *c = cv + f * chromaNum;
            float red = normType == 2 ? 3.4e38f : 0.f;
            for (int k = 0; k < chromaNum; ++k) {             // 1 max, 2 min, 3 P2, 4 P1 (__mnormalize)
                const float av = fabsf(c[k]);
                if (normType == 1) red = fmaxf(red, av);
                else if (normType == 2) red = fminf(red, av);
                else if (normType == 3) red += av * av;
                else red += av;
            }
            if (normType == 3) red = sqrtf(red);
            if (red != 0.f) v = v / red;
        }
        po[it] = v;
    }
}

// The default form.  The per-class bin lists are built on the host (afx_cqt.c: afx_chroma_lists, CPU-tested) and
// travel as a kernel argument, copied to LDS once per workgroup; |Q|^2 of CH_FRAMES frames is staged in LDS with
// coalesced 16-byte loads at an odd row pitch (the 64 frames of a read fall on distinct banks); a thread owns
// (frame = tid & 63, classes wave, wave + 4, ...), so the class -- and with it the list walked -- is uniform per
// wave (broadcast LDS reads).  Sums in ascending bin order = the 0/1 matrix product of cqt_algorithm.c:553-560.
// Round 3, measured at 125 clips (profiles/r03_round_start.txt): 149 us per pass against 206 us for the round-2
// kernel, which built the lists per workgroup by scanning the matrix in global memory (removed).
__global__ __launch_bounds__(256) void k_cqt_chroma(const float *__restrict__ re, const float *__restrict__ im,
                                                       long long rows, int num, AfxChromaLists L, int chromaNum,
                                                       int isMag, int normType, float *__restrict__ out, int vec4) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int pitch = num | 1;                                // odd: frames of one bin on distinct banks
    float *p = reinterpret_cast<float *>(smem_raw);           // [CH_FRAMES][pitch]
    float *cv = p + CH_FRAMES * pitch;                        // [CH_FRAMES][chromaNum]
    const int tid = threadIdx.x;
    const long long f0 = (long long)blockIdx.x * CH_FRAMES;
    const int nf = rows - f0 < CH_FRAMES ? (int)(rows - f0) : CH_FRAMES;
    const float *pr = re + f0 * num, *pi = im + f0 * num;
    auto pw = [&](float a, float b) {
        float v = __fmaf_rn(a, a, b * b);
        if (isMag) v = sqrtf(v);
        return v;
    };
    if (vec4) {
        const float4 *pr4 = reinterpret_cast<const float4 *>(pr), *pi4 = reinterpret_cast<const float4 *>(pi);
        const int n4 = nf * num / 4, q4 = num / 4;
        for (int e0 = tid; e0 < n4; e0 += 4 * 256) {
            float4 a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + 256 * u;  // (in the last trip whole waves fall past the end: no load is issued for them)
                if (e < n4) {
                    a[u] = pr4[e];
                    b[u] = pi4[e];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + 256 * u;
                if (e < n4) {
                    const int f = e / q4, j = 4 * (e - f * q4);
                    float *d = p + f * pitch + j;
                    d[0] = pw(a[u].x, b[u].x);
                    d[1] = pw(a[u].y, b[u].y);
                    d[2] = pw(a[u].z, b[u].z);
                    d[3] = pw(a[u].w, b[u].w);
                }
            }
        }
    } else {
        for (int e = tid; e < nf * num; e += 256) {
            const int f = e / num, j = e - f * num;
            p[f * pitch + j] = pw(pr[e], pi[e]);
        }
    }
    // the lists: kernel argument -> LDS, one byte per thread and trip (indexing the argument itself in the walk below
    // compiles to a dependent global load per bin)
    unsigned char *sL = reinterpret_cast<unsigned char *>(cv + CH_FRAMES * chromaNum);
    for (int b = tid; b < (int)sizeof(AfxChromaLists); b += 256) sL[b] = reinterpret_cast<const unsigned char *>(&L)[b];
    __syncthreads();
    {
        const unsigned short *sStart = reinterpret_cast<const unsigned short *>(sL);
        const unsigned char *sBins = sL + sizeof(L.start);
        const int f = tid & (CH_FRAMES - 1);
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const float *row = p + f * pitch;
        for (int c = wave; c < chromaNum; c += 4) {           // uniform per wave
            const int b0 = sStart[c], b1 = sStart[c + 1];
            float v = 0.f;
            for (int q = b0; q < b1; ++q) v += row[sBins[q]];
            if (f < nf) cv[f * chromaNum + c] = v;
        }
    }
    __syncthreads();
    // per-frame normalisation (__mnormalize) and store: thread (frame, class) reads its frame's chromaNum values
    float *po = out + f0 * chromaNum;
    for (int it = tid; it < nf * chromaNum; it += 256) {
        const int f = it / chromaNum;
        float v = cv[it];
        if (normType != 0) {
            const float *c = cv + f * chromaNum;
            float red = normType == 2 ? 3.4e38f : 0.f;
            for (int k = 0; k < chromaNum; ++k) {             // 1 max, 2 min, 3 P2, 4 P1, in k_cqt_chroma's order
                const float av = fabsf(c[k]);
                if (normType == 1) red = fmaxf(red, av);
                else if (normType == 2) red = fminf(red, av);
                else if (normType == 3) red += av * av;
                else red += av;
            }
            if (normType == 3) red = sqrtf(red);
            if (red != 0.f) v = v / red;
        }
        po[it] = v;
    }
}

}  // namespace


template <int CT, int WAVES>
static int launch_cqt_mfma(const AfxCqtOctaveArgs *a, void *stream) {
    const int N = 1 << a->radix2Exp;
    int SH = 0;
    while (SH < 30 && !((a->hop >> SH) & 1)) ++SH;           // hop = odd * 2^SH
    const int S = 31 * a->hop + N;
    const int sigWords = ((SH ? S + (S >> SH) : 2 * S) + 67) & ~3;
    const size_t lds = sizeof(float) * ((size_t)sigWords + (size_t)WAVES * 16 * 64);
    if (lds > 150 * 1024) return AFX_ERR_UNSUPPORTED;
    const void *fn = reinterpret_cast<const void *>(k_cqt_octave_mfma<CT, WAVES>);
    if (lds > 48 * 1024) AFX_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int tilesPerClip = (a->timeLength + 31) / 32;
    const long long total = (long long)tilesPerClip * (a->batch > 0 ? a->batch : 1);
    if (total > 0x7fffffffLL) return AFX_ERR_UNSUPPORTED;
    // register prefetch of the next window needs 16-byte aligned rows and <= 5 float4 per thread
    const int pre = (reinterpret_cast<uintptr_t>(a->x) % 16 == 0) && (a->xStride % 4 == 0) &&
                    S <= 4 * 5 * (N / 2);
    // persistent workgroups (each keeps its slice of G in registers): two rounds of the 256 CUs
    const unsigned grid = (unsigned)(total < 512 ? total : 512);
    AfxCqtOctaveArgs b = *a;
    if (b.batch <= 0) b.batch = 1;
    hipLaunchKernelGGL((k_cqt_octave_mfma<CT, WAVES>), dim3(grid), dim3(64 * WAVES), lds,
                       (hipStream_t)stream, b, tilesPerClip, SH, sigWords, pre);
    AFX_LAUNCH_CHECK("k_cqt_octave_mfma");
    return AFX_OK;
}

template <int CT>
static int dispatch_cqt_mfma(const AfxCqtOctaveArgs *a, void *stream) {
    // the wave's slice of G lives in VGPRs (64 x CT), which caps the workgroup size (N/2 threads)
    switch (1 << a->radix2Exp) {
        case 256: return launch_cqt_mfma<CT, 2>(a, stream);
        case 512: return launch_cqt_mfma<CT, 4>(a, stream);
        case 1024: if (CT <= 2) return launch_cqt_mfma<CT, (CT <= 2 ? 8 : 4)>(a, stream); break;
        case 2048: if (CT == 1) return launch_cqt_mfma<CT, (CT == 1 ? 16 : 4)>(a, stream); break;
        default: break;
    }
    return AFX_ERR_UNSUPPORTED;
}


// independent-wave variant: needs G in LDS (N <= 512, one column tile), float4-aligned rows
template <int NBLK, int NV, int SHC>
static int launch_cqt_mfma_w(const AfxCqtOctaveArgs *a, int SH, int sigWords, int waves, void *stream) {
    const int N = NBLK * 16;
    const size_t lds = sizeof(float) * ((size_t)N * 32 + (size_t)waves * sigWords);
    const void *fn = reinterpret_cast<const void *>(k_cqt_octave_mfma_w<NBLK, NV, SHC>);
    if (lds > 48 * 1024) AFX_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int tilesPerClip = (a->timeLength + 31) / 32;
    const long long total = (long long)tilesPerClip * (a->batch > 0 ? a->batch : 1);
    if (total > 0x7fffffffLL) return AFX_ERR_UNSUPPORTED;
    long long wgs = (total + waves - 1) / waves;
    if (wgs > 256) wgs = 256;  // one persistent workgroup per CU
    AfxCqtOctaveArgs b = *a;
    if (b.batch <= 0) b.batch = 1;
    hipLaunchKernelGGL((k_cqt_octave_mfma_w<NBLK, NV, SHC>), dim3((unsigned)wgs), dim3(64 * waves), lds,
                       (hipStream_t)stream, b, tilesPerClip, SH, sigWords);
    AFX_LAUNCH_CHECK("k_cqt_octave_mfma_w");
    return AFX_OK;
}

static int try_cqt_mfma_w(const AfxCqtOctaveArgs *a, void *stream) {
    const int N = 1 << a->radix2Exp;
    if (a->colTiles != 1 || (N != 256 && N != 512)) return AFX_ERR_UNSUPPORTED;
    // top octave of the default ladder (hop = N/4 = 128): its 18 KB signal windows leave this kernel 4 waves
    // per CU; the K-split kernel (4 waves share one window, 16 waves per CU) measures 2 % faster on the whole
    // cfg-5 step (3.82 -> 3.74 ms, round 2).
    if (N == 512 && a->hop * 4 == N) return AFX_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(a->x) % 16) || (a->xStride % 4)) return AFX_ERR_UNSUPPORTED;
    int SH = 0;
    while (SH < 30 && !((a->hop >> SH) & 1)) ++SH;
    const int S = 31 * a->hop + N;
    const int sigWords = ((SH ? S + (S >> SH) : 2 * S) + 67) & ~3;
    const int nv = (S + 255) / 256;  // float4 loads per lane
    int waves = (int)((150 * 1024 - (size_t)N * 128) / (sizeof(float) * sigWords));
    if (waves > (nv > 10 ? 4 : 8)) waves = nv > 10 ? 4 : 8;  // 18 window registers x 4: 256-thread build
    if (waves < 4 || nv > 18) return AFX_ERR_UNSUPPORTED;
    if (N == 512) {
        // the octave ladder of the default plan (hop = N/4 halving down to 2): shift compiled in
        if (a->hop == 128) return launch_cqt_mfma_w<32, 18, 7>(a, SH, sigWords, waves, stream);
        if (a->hop == 64) return launch_cqt_mfma_w<32, 10, 6>(a, SH, sigWords, waves, stream);
        if (a->hop == 32) return launch_cqt_mfma_w<32, 10, 5>(a, SH, sigWords, waves, stream);
        if (a->hop == 16) return launch_cqt_mfma_w<32, 5, 4>(a, SH, sigWords, waves, stream);
        if (a->hop == 8) return launch_cqt_mfma_w<32, 5, 3>(a, SH, sigWords, waves, stream);
        if (a->hop == 4) return launch_cqt_mfma_w<32, 5, 2>(a, SH, sigWords, waves, stream);
        if (a->hop == 2) return launch_cqt_mfma_w<32, 5, 1>(a, SH, sigWords, waves, stream);
        if (nv <= 5) return launch_cqt_mfma_w<32, 5, 0>(a, SH, sigWords, waves, stream);
        if (nv <= 10) return launch_cqt_mfma_w<32, 10, 0>(a, SH, sigWords, waves, stream);
        return launch_cqt_mfma_w<32, 18, 0>(a, SH, sigWords, waves, stream);
    }
    if (nv <= 5) return launch_cqt_mfma_w<16, 5, 0>(a, SH, sigWords, waves, stream);
    if (nv <= 10) return launch_cqt_mfma_w<16, 10, 0>(a, SH, sigWords, waves, stream);
    return launch_cqt_mfma_w<16, 18, 0>(a, SH, sigWords, waves, stream);
}

extern "C" int afxk_cqt_octave(const AfxCqtOctaveArgs *a, void *stream) {
    if (a->radix2Exp < 1 || a->radix2Exp > 14) return AFX_ERR_UNSUPPORTED;
    if (a->timeLength <= 0) return AFX_OK;
    const int N = 1 << a->radix2Exp;
    if (a->timeKernel && a->colTiles >= 1 && a->colTiles <= 3) {
        int st = afxk_cqt_octave_f16(a, stream);
        if (st != AFX_ERR_UNSUPPORTED) return st;
        st = try_cqt_mfma_w(a, stream);
        if (st != AFX_ERR_UNSUPPORTED) return st;
        if (a->colTiles == 1) st = dispatch_cqt_mfma<1>(a, stream);
        else if (a->colTiles == 2) st = dispatch_cqt_mfma<2>(a, stream);
        else st = dispatch_cqt_mfma<3>(a, stream);
        if (st != AFX_ERR_UNSUPPORTED) return st;  // otherwise: FFT path below
    }
    int threads = N / 2;
    if (threads < 64) threads = 64;
    if (threads > 256) threads = 256;
    const size_t lds = (size_t)N * sizeof(float2);
    if (lds > 48 * 1024) {
        AFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_cqt_octave),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    hipLaunchKernelGGL(k_cqt_octave, dim3((unsigned)a->timeLength, (unsigned)(a->batch > 0 ? a->batch : 1)), dim3(threads), lds,
                       (hipStream_t)stream, *a);
    AFX_LAUNCH_CHECK("k_cqt_octave");
    return AFX_OK;
}

extern "C" int afxk_cqt_decimate(const float *x, int srcLen, long long xStride, float *y,
                                 int dstLen, long long yStride, int batch, const float *taps32,
                                 float sqrtRatio, void *stream) {
    if (dstLen <= 0 || batch <= 0) return AFX_OK;
    Taps32 tp;
    for (int i = 0; i < 32; ++i) tp.h[i] = taps32[i];
    const dim3 grid((unsigned)((dstLen + DEC_OUT - 1) / DEC_OUT), (unsigned)batch);
    // 16-byte loads and stores when every clip row of x and y starts 16-byte aligned
    const bool pairs = (reinterpret_cast<uintptr_t>(x) % 16 == 0) && (xStride % 4 == 0 || batch == 1) &&
                       (reinterpret_cast<uintptr_t>(y) % 16 == 0) && (yStride % 4 == 0 || batch == 1);
    if (pairs)
        hipLaunchKernelGGL(k_cqt_decimate<true>, grid, dim3(256), 0, (hipStream_t)stream, x, srcLen, xStride, y,
                           dstLen, yStride, tp, sqrtRatio);
    else
        hipLaunchKernelGGL(k_cqt_decimate<false>, grid, dim3(256), 0, (hipStream_t)stream, x, srcLen, xStride, y,
                           dstLen, yStride, tp, sqrtRatio);
    AFX_LAUNCH_CHECK("k_cqt_decimate");
    return AFX_OK;
}

extern "C" int afxk_cqt_deconv(const float *in, long long rows, int num, int radix2Exp,
                               const float *twiddle, const int *hcIdx, int hcNum, float *outTimbre,
                               float *outPitch, float *outHc, void *stream) {
    if (rows <= 0) return AFX_OK;
    if (radix2Exp < 1 || radix2Exp > 12 || rows > 0x7fffffffLL) return AFX_ERR_UNSUPPORTED;
    const int M = 1 << radix2Exp;
    int threads = M / 2;
    if (threads < 64) threads = 64;
    if (threads > 256) threads = 256;
    const size_t lds = (size_t)2 * M * sizeof(float2);
    if (lds > 48 * 1024)
        AFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_cqt_deconv),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_cqt_deconv, dim3((unsigned)rows), dim3(threads), lds, (hipStream_t)stream, in, rows,
                       num, radix2Exp, reinterpret_cast<const float2 *>(twiddle), hcIdx, hcNum, outTimbre,
                       outPitch, outHc);
    AFX_LAUNCH_CHECK("k_cqt_deconv");
    return AFX_OK;
}

extern "C" int afxk_cqt_chroma(const float *re, const float *im, long long rows, int num,
                               const unsigned char *fold, const AfxChromaLists *lists, int chromaNum, int isMag,
                               int normType, float *out, void *stream) {
    if (rows <= 0) return AFX_OK;
    if (num < 1 || chromaNum < 1) return AFX_ERR_ARG;
    const long long blocks = (rows + CH_FRAMES - 1) / CH_FRAMES;
    if (blocks > 0x7fffffffLL) {
        afxdev_set_error("cqt chroma: %lld rows in one launch", rows);
        return AFX_ERR_UNSUPPORTED;
    }
    if (lists && chromaNum <= 64 && num <= 255) {
        const int vec4 = (num % 4 == 0) && (reinterpret_cast<uintptr_t>(re) % 16 == 0) && (reinterpret_cast<uintptr_t>(im) % 16 == 0);
        const AfxChromaLists L = *lists;
        const size_t lds = sizeof(float) * ((size_t)CH_FRAMES * (num | 1) + (size_t)CH_FRAMES * chromaNum) +
                           ((sizeof(AfxChromaLists) + 15) & ~(size_t)15);
        if (lds > 48 * 1024)
            AFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_cqt_chroma),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_cqt_chroma, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, re, im, rows, num,
                           L, chromaNum, isMag, normType, out, vec4);
        AFX_LAUNCH_CHECK("k_cqt_chroma");
        return AFX_OK;
    }
    // byte-sized bin lists do not cover this plan (e.g. 36 bins / octave x 8 octaves = 288 bins): flag scan
    const size_t lds = sizeof(float) * (size_t)CH_FRAMES * chromaNum;
    if (!fold || lds > 150 * 1024) {
        afxdev_set_error("cqt chroma: %d classes of %d bins are outside the kernels' range", chromaNum, num);
        return AFX_ERR_UNSUPPORTED;
    }
    if (lds > 48 * 1024)
        AFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_cqt_chroma_scan),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_cqt_chroma_scan, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, re, im, rows, num,
                       fold, chromaNum, isMag, normType, out);
    AFX_LAUNCH_CHECK("k_cqt_chroma_scan");
    return AFX_OK;
}
