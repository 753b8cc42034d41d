// afx_melfused4k2.hip -- n_fft = 4096 (radix2Exp 12, the default of the reference wrapper), real results: framed
// STFT -> |S|^2 (or |S|, |S|^2p) -> banded filter bank, one 64-lane wave per frame, in the round-2 form of the
// headline kernel (afx_melfused2.hip): every LDS table laid out for 16-byte reads, exchanges written two rows at a
// time, hand-issued ds_read_b128 with explicit waits, no workgroup barrier in the frame loop.
//
// The 4096 real samples are 2048 complex points z[n] = (x[2n], x[2n+1]); their transform Z is built from the
// 1024-point transforms Ze, Zo of the even and odd z (decimation in time) and then split for real input:
//     ze[m] = (x[4m], x[4m+1]),  zo[m] = (x[4m+2], x[4m+3])        -- the two halves of a lane's float4
//     Z[k] = Ze[k] + W_2048^k Zo[k],  Z[k+1024] = Ze[k] - W_2048^k Zo[k]
//     Z[1024-k] = Ze[1024-k] - conj(W_2048^k) Zo[1024-k],  Z[2048-k] = Ze[1024-k] + conj(W_2048^k) Zo[1024-k]
//     X[k], conj(X[2048-k])      = E -+ ... of the pair (Z[k], Z[2048-k]) with w = W_4096^k / 2      (split_pair)
//     X[1024-k], conj(X[1024+k]) = the pair (Z[1024-k], Z[1024+k]) with w' = W_4096^(1024-k) / 2 = -i conj(w)
// Each half runs exactly the 16 x 16 x 4 pipeline of afx_melfused2.hip up to its last radix-4; a lane finishes the
// radix-4 of a base q and of its mirror 256 - q and so holds Zh[k] and Zh[1024-k] of both halves for its eight k:
// everything above is lane-local.  The first half's 16 values wait in registers while the second is transformed.
// Lane 0 carries the self-mirrored base q = 128 in place of its duplicate slots (as in afx_melfused2.hip) and the
// pair (512, 1536).  8 waves per workgroup (2 per SIMD): the frame's float4 image stays in registers and moves down by
// hop / 256 register pairs per frame (hop 1024: four new float4 per lane and frame).
//
// Real results (specMap 0 / 1 / 2) and complex results (3: S, 4: S^2 -- bftObj_setResultType(0), the wrapper's default for
// bft(): the imaginary parts wait in registers for a second pass of the bank; 246 VGPRs).  Round 1's k_stft_band_4k is gone.
// Per frame the same reference code: stft_algorithm.c:696-803 (frame, window, FFT), bft_algorithm.c:360-455
// (spectrum value, bank product).
#include <hip/hip_runtime.h>

#include <atomic>

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "afx_device.h"
#include "afx_hipcheck.h"
#include "afx_pkmath.h"
#include "afx_ccblock.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int NFFT = 4096;
constexpr int MC = 1024;                     // complex length of one half
constexpr int WAVES = 8;                     // one workgroup per CU, 2 waves per SIMD
constexpr int P1 = 72;                       // float2 per row of the exchange-1 image
constexpr int PROW_OFF = 1024;               // byte offset of the power row in a wave's region
constexpr int PROW_F = 2176;                 // 2049 bins + zero pad for the fixed-length band loops (afx_bft_fast.c: v4k)
constexpr int WAVE_LDS = PROW_OFF + PROW_F * 4;  // 9728: the pad [9220, 9728) lies behind both images
static_assert(16 * P1 * 8 <= PROW_OFF + 2049 * 4, "exchange image must end before the zero pad");
// table blob, byte offsets (built on the host by afxk_mel4k2_create, copied to LDS per workgroup)
constexpr int T_WIN = 0;                     // [2 halves][8][64] float4: (w[4m+2h], w[4m+2h+1]) of rows n1 = 2j, 2j + 1
constexpr int T_TW1 = 16384;                 // [8][64] float4: W_1024^(lane k1), k1 = 2j, 2j + 1
constexpr int T_TW2 = 24576;                 // [4] rows of 16 float2, TW2_PITCH bytes apart: W_64^(m2 j1)
constexpr int TW2_PITCH = 144;               // (conflict-free for the b128 lane groups; 128 is two-way: afx_melfused2.hip)
constexpr int T_TWC = 25152;                 // [2][64][4] float2: W_2048^bin of slot (s, lane, m)
constexpr int T_TWS = 29248;                 // [2][64][4] float2: 0.5 W_4096^bin of slot (s, lane, m)
constexpr int T_BAND = 33344;                // [64][WP] floats: lane-major band weights, A then B taps
__host__ __device__ constexpr int wpitch(int ta, int tb) { return ta + tb + 4; }
__host__ __device__ constexpr int tab_bytes(int ta, int tb) { return T_BAND + 64 * wpitch(ta, tb) * 4; }
__host__ __device__ constexpr int block_lds_bytes(int ta, int tb) { return tab_bytes(ta, tb) + WAVES * WAVE_LDS; }

struct KArgs4 {
    const float *x;
    long long clipStride;
    long long totalFrames;
    int timeLength, hop;
    int framesPerWave;
    int aligned;           // frame starts are 16-byte aligned -> float4 loads
    const float4 *tab;     // table blob
    const int *meta;       // [6][64]: startA, startB, rowA, rowB, segIdx lo / hi
    int specMap, postPow;
    float normValue;
    float *out;            // [totalFrames, num]
    float *outIm;          // complex results (specMap 3 / 4): imaginary parts, same shape
    int num;
    // STFT instantiations (afxk_stft4k): bins instead of bank rows
    const float *window;   // device [4096], natural order
    int mode;              // AFX_SPEC_*
    int binLo, binCount;   // bins binLo .. binLo + binCount - 1 are stored; above 2048: conjugate mirrors
    long long outPitch;    // floats between output rows
    // CC instantiations: cepstra of the rows in the same launch (afx_ccblock.h)
    const float *dct;      // device [num, num] orthonormal DCT-II
    int ccNum, ccCbrt;
    float *cc;             // [totalFrames, ccNum]
};

// orders this wave's LDS stores before its later LDS loads of other lanes' data (afx_melfused2.hip)
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
}

// wave priority by phase of the frame, as in afx_melfused2.hip (transform phases above window / band / store)
#ifndef AFX_PRIO4K_MASK
#define AFX_PRIO4K_MASK 0x1E
#endif
#define MEL4K_PHASE(p)                                                                               \
    do {                                                                                             \
        if (AFX_PRIO4K_MASK != 0) __builtin_amdgcn_s_setprio(((AFX_PRIO4K_MASK >> (p)) & 1) ? 1 : 0); \
    } while (0)

__device__ __forceinline__ v2 lo2(v4f q) { return v2{q.x, q.y}; }
__device__ __forceinline__ v2 hi2(v4f q) { return v2{q.z, q.w}; }

// |X|^2 of the conjugate pair (a, 2048-a) from A = Z[a], B = Z[2048-a], w = 0.5 W_4096^a
__device__ __forceinline__ void split_pair(v2 A, v2 B, v2 w, float &pa, float &pb) {
    const v2 e2 = pk_add_conj(A, B);   // 2 E
    const v2 d = pk_sub_conj(A, B);    // 2 i O
    const v2 wo = cmul_mi(d, w);       // W O
    const v2 x = e2 * 0.5f + wo;       // X[a]
    const v2 y = e2 * 0.5f - wo;       // conj(X[2048-a])
    pa = x.x * x.x + x.y * x.y;
    pb = y.x * y.x + y.y * y.y;
}
// the same for the pair (1024-k, 1024+k): A = Z[1024-k], B = Z[1024+k], its twiddle is -i conj(w), w = 0.5 W_4096^k
__device__ __forceinline__ void split_pair_q(v2 A, v2 B, v2 w, float &pa, float &pb) {
    const v2 e2 = pk_add_conj(A, B);
    const v2 d = pk_sub_conj(A, B);
    const v2 vv = cmul_conj(d, w);     // -(W' O)
    const v2 x = e2 * 0.5f - vv;       // X[1024-k]
    const v2 y = e2 * 0.5f + vv;       // conj(X[1024+k])
    pa = x.x * x.x + x.y * x.y;
    pb = y.x * y.x + y.y * y.y;
}

// complex results: the spectrum values themselves (x = X[a], y = conj(X[2048-a]); q variant: x = X[1024-k], y = conj(X[1024+k]))
__device__ __forceinline__ void split_pair_c(v2 A, v2 B, v2 w, v2 &x, v2 &y) {
    const v2 e2 = pk_add_conj(A, B);
    const v2 d = pk_sub_conj(A, B);
    const v2 wo = cmul_mi(d, w);
    x = e2 * 0.5f + wo;
    y = e2 * 0.5f - wo;
}
__device__ __forceinline__ void split_pair_qc(v2 A, v2 B, v2 w, v2 &x, v2 &y) {
    const v2 e2 = pk_add_conj(A, B);
    const v2 d = pk_sub_conj(A, B);
    const v2 vv = cmul_conj(d, w);
    x = e2 * 0.5f - vv;
    y = e2 * 0.5f + vv;
}
// (re, im) of the requested complex result from a spectrum value c: S (sq = false) or S^2 (bft_algorithm.c:457-485)
__device__ __forceinline__ void cplx_map(v2 c, bool sq, float &re, float &im) {
    re = sq ? c.x * c.x - c.y * c.y : c.x;
    im = sq ? 2.f * (c.x * c.y) : c.y;
}

// what an STFT instantiation stores for a spectrum value (the maps of afx_stft.hip)
__device__ __forceinline__ void stft_map(float re, float im, int mode, float normValue, float &v0, float &v1) {
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

// SHIFT: hop = 256 * SHIFT samples -> the next frame's register image is this one moved down by SHIFT float4,
//   only SHIFT new float4 per lane are fetched (0: every frame fetched whole)
// SPLIT: the plan's slots hold row SEGMENTS (afx_bandplan_build_split)
// CPLX: complex results (specMap 3: S, 4: S^2): the imaginary parts wait in registers for a second pass of the bank
// STFT: no bank -- the spectrum values themselves (CPLX form, specMap 3) go to memory through stft_map: the STFT object's
//   full complex spectrum, the linear-scale bin slices, the reassignment object's transforms (afxk_stft4k; frames inside the clip);
//   MAPPED: any AFX_SPEC_* map, otherwise the complex values as they are; FULL: all 4096 bins are stored (no range checks)
// CC: cepstra of the rows in the same launch (real results; afx_ccblock.h: every 16 frames the wave re-reads its rows from L2)
template <int TA, int TB, int SHIFT, bool SPLIT, bool CPLX, bool STFT = false, bool MAPPED = false, bool FULL = false, bool CC = false>
__global__ __launch_bounds__(WAVES * 64, 2) void k_stft_band_4k2(KArgs4 a) {
    static_assert(!CC || (!CPLX && !STFT), "cepstra: real bank rows");
    static_assert(!STFT || (CPLX && !SPLIT && TA == 0 && TB == 0), "STFT instantiations: complex values, no bank");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    constexpr int WP = wpitch(TA, TB);
    constexpr int TABB = tab_bytes(TA, TB);
    unsigned char *wreg = smem + TABB + wave * WAVE_LDS;
    float *prow = reinterpret_cast<float *>(wreg + PROW_OFF);

    // ---- workgroup-shared tables -> LDS (once) -------------------------------------------
    {
        float4 *s4 = reinterpret_cast<float4 *>(smem);
        for (int i = threadIdx.x + (STFT ? T_TW1 / 16 : 0); i < TABB / 16; i += WAVES * 64) s4[i] = a.tab[i];
        if constexpr (STFT) {
            // the window of each half in pair layout from the object's natural-order window (afxk_mel4k_create builds the same)
            v2 *winT = reinterpret_cast<v2 *>(smem + T_WIN);
            for (int i = threadIdx.x; i < 2048; i += WAVES * 64) {
                const int h = i >> 10, n1 = (i >> 6) & 15, l = i & 63;
                const int m = 64 * n1 + l;
                winT[1024 * h + 128 * (n1 >> 1) + 2 * l + (n1 & 1)] = v2{a.window[4 * m + 2 * h], a.window[4 * m + 2 * h + 1]};
            }
        }
        for (int i = 2049 + lane; i < PROW_F; i += 64) prow[i] = 0.f;  // zero pad, never overwritten
    }
    __syncthreads();

    // ---- per-lane constants (loop-invariant LDS byte addresses; afx_melfused2.hip) -----------
    const int k1 = lane >> 2, m2 = lane & 3;
    const unsigned T0 = lds_addr(smem), W0 = lds_addr(wreg);
    const unsigned aWin = T0 + T_WIN + 16 * lane;                      // + 1024 j + 8192 half; W_1024 at + T_TW1
    const unsigned aTw2 = T0 + T_TW2 + TW2_PITCH * m2;                       // + 16 j
    const unsigned aE1w = W0 + 8 * (8 * (k1 >> 1) + 2 * m2 + (k1 & 1));  // writer m1 = lane >> 2; row k: + 576 k
    const unsigned aE1r = W0 + 576 * k1 + 16 * m2;                     // pair jj: + 64 jj
    const unsigned aE2w = W0 + 32 * k1 + 16 * ((m2 >> 1) ^ ((k1 >> 3) & 1)) + 8 * (m2 & 1);  // j1: + 512 j1
    const int b3l = (lane >> 3) & 1;
    const unsigned aAlo = W0 + 32 * lane + 16 * b3l, aAhi = W0 + 32 * lane + 16 * (1 - b3l);  // s = 1: + 2048
    const int qm0 = lane == 0 ? 128 : 256 - lane, qm1 = 192 - lane;
    const unsigned aB0lo = W0 + 32 * qm0 + 16 * ((qm0 >> 3) & 1), aB0hi = W0 + 32 * qm0 + 16 * (1 - ((qm0 >> 3) & 1));
    const unsigned aB1lo = W0 + 32 * qm1 + 16 * ((qm1 >> 3) & 1), aB1hi = W0 + 32 * qm1 + 16 * (1 - ((qm1 >> 3) & 1));
    const unsigned aTclo = T0 + T_TWC + 32 * lane + 16 * b3l, aTchi = T0 + T_TWC + 32 * lane + 16 * (1 - b3l);  // T_TWS: + 4096
    const unsigned R = W0 + PROW_OFF;
    // power-row stores, offsets in units of 64 floats: bins k = lane + 64 s + 256 j and 1024 + k (+ 16 units) ...
    const unsigned aP01 = R + 4 * lane;
    const unsigned aP23 = R + 4 * (lane == 0 ? 128 : lane + 512);      // s = 0 slots 2, 3 | lane 0: (128, 384)
    // ... and 1024 - k, 2048 - k (+ 16 units)
    const unsigned aQs1 = R + 4 * (192 - lane);                        // s = 1 slots 3 .. 0 at + 0, 4, 8, 12; s = 0 slots 1, 0 at + 9, + 13
    const unsigned aQ23 = R + 4 * (lane == 0 ? 640 : 256 - lane);      // s = 0 slots 3, 2 | lane 0: (640, 896)
    const bool lane0 = (lane == 0);

    const int startA = STFT ? 0 : a.meta[lane], startB = STFT ? 0 : a.meta[64 + lane];
    const int rowA = STFT ? -1 : a.meta[128 + lane], rowB = STFT ? -1 : a.meta[192 + lane];
    const unsigned seg0 = SPLIT ? (unsigned)a.meta[256 + lane] : 0u, seg1 = SPLIT ? (unsigned)a.meta[320 + lane] : 0u;
    const unsigned apa = R + 4 * startA, apb = R + 4 * startB;
    const unsigned awr = T0 + T_BAND + 4 * WP * lane;

    const long long gw = (long long)blockIdx.x * WAVES + wave;
    long long f = gw * a.framesPerWave;
    long long fEnd = f + a.framesPerWave;
    if (fEnd > a.totalFrames) fEnd = a.totalFrames;
    if (f >= fEnd) return;
    int clip = (int)(f / a.timeLength);
    int t = (int)(f - (long long)clip * a.timeLength);
    int ccN = 0;  // CC: rows of this wave whose cepstra are still to be formed

    // rlo[n1] = ze[m] = (x[4m], x[4m+1]), rhi[n1] = zo[m] = (x[4m+2], x[4m+3]), m = 64 n1 + lane: two register images, each
    // moved down in place and refilled right behind its own window multiply by ONE asm statement (rows_shift_fetch /
    // rows_fetch_all, afx_asm.h: 8-byte loads, dword alignment is enough); the loads are waited for by hand at the end of
    // the frame, BEFORE its row stores are issued (so the wait never meets a store that was issued a moment ago)
    v2 rlo[16], rhi[16];
    {
        const float *px = a.x + (long long)clip * a.clipStride + (long long)t * a.hop + 4 * lane;
        rows_fetch_all(rlo, px);
        rows_fetch_all(rhi, px + 2);
        VM_WAIT_ALL();
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            PIN(rlo[n1]);
            PIN(rhi[n1]);
        }
    }

    for (; f < fEnd; ++f) {
        // first half's values at the lane's slots: EA[4 s + j] = Ze[k], EB[4 s + j] = Ze[1024 - k], k = lane + 64 s + 256 j
        // (lane 0, s = 0: k = 0, 256, 128, 384); ec = Ze[512] (lane 0)
        v2 EA[8], EB[8], ec;
        float pk[8], pn[8], pm[8], pq[8];  // |X|^2 (CPLX: real parts) at bins k, 2048 - k, 1024 - k, 1024 + k
        float p512, p1536;
        float ik[CPLX ? 8 : 1], in_[CPLX ? 8 : 1], im_[CPLX ? 8 : 1], iq[CPLX ? 8 : 1], i512 = 0.f, i1536 = 0.f;  // CPLX: imaginary parts
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            v2 v[16];
            MEL4K_PHASE(0);
            // ---- 1. window: 8 x 16 bytes per lane, the first half is used while the second lands ----
            {
                v4f wv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) RD128(wv[j], aWin, T_WIN + 8192 * half + 1024 * j);
                LDS_WAIT_N(4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    PIN(wv[j]);
                    v[2 * j] = (half ? rhi[2 * j] : rlo[2 * j]) * lo2(wv[j]);
                    v[2 * j + 1] = (half ? rhi[2 * j + 1] : rlo[2 * j + 1]) * hi2(wv[j]);
                }
                LDS_WAIT_N(0);
#pragma unroll
                for (int j = 4; j < 8; ++j) {
                    PIN(wv[j]);
                    v[2 * j] = (half ? rhi[2 * j] : rlo[2 * j]) * lo2(wv[j]);
                    v[2 * j + 1] = (half ? rhi[2 * j + 1] : rlo[2 * j + 1]) * hi2(wv[j]);
                }
            }
            // ---- 1b. this half's samples are consumed (pinned: the products must not sink below the statement that
            //      overwrites their operands -- the compiler would copy the whole image aside for them): move its image
            //      down and request the next frame's rows ----
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) PIN(v[n1]);
            if (f + 1 < fEnd) {
                int tn = t + 1, cn = clip;
                if (tn == a.timeLength) {
                    tn = 0;
                    ++cn;
                }
                const float *pnx = a.x + (long long)cn * a.clipStride + (long long)tn * a.hop + 4 * lane + 2 * half;
                bool whole = true;
                if constexpr (SHIFT > 0) {
                    if (tn != 0) {
                        rows_shift_fetch<SHIFT>(half ? rhi : rlo, pnx + 256 * (16 - SHIFT));
                        whole = false;
                    }
                }
                if (whole) rows_fetch_all(half ? rhi : rlo, pnx);
            }

            // ---- 2a. radix-16 over n1, twiddle W_1024^(lane k1), transpose through LDS -----------
            MEL4K_PHASE(1);
            dft16(v);
            {
                v4f tq[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) RD128(tq[j], aWin, T_TW1 + 1024 * j);
                LDS_WAIT_N(0);
#pragma unroll
                for (int j = 0; j < 8; ++j) PIN(tq[j]);
                v2 o[16];
                o[0] = v[0];
#pragma unroll
                for (int k = 1; k < 16; ++k) o[k] = cmul(v[rev4(k)], (k & 1) ? hi2(tq[k >> 1]) : lo2(tq[k >> 1]));
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const unsigned b = aE1w + 2304 * g;  // rows 4g .. 4g+3, 576 bytes = 72 units apart
                    WR2_64(b, o[4 * g], o[4 * g + 1], 0, 72);
                    WR2_64(b, o[4 * g + 2], o[4 * g + 3], 144, 216);
                }
            }
            wave_lds_sync();
            MEL4K_PHASE(2);
            {
                v4f rq[8];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) RD128(rq[jj], aE1r, 64 * jj);
                wave_lds_sync();
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    PIN(rq[jj]);
                    v[2 * jj] = lo2(rq[jj]);
                    v[2 * jj + 1] = hi2(rq[jj]);
                }
            }

            // ---- 2b. radix-16 over m1, twiddle W_64^(m2 j1) -> image V[q = k1 + 16 j1][m2] --------
            MEL4K_PHASE(3);
            dft16(v);
            {
                v4f tq[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) RD128(tq[j], aTw2, 16 * j);
                LDS_WAIT_N(0);
#pragma unroll
                for (int j = 0; j < 8; ++j) PIN(tq[j]);
                v2 o[16];
                o[0] = v[0];
#pragma unroll
                for (int j1 = 1; j1 < 16; ++j1) o[j1] = cmul(v[rev4(j1)], (j1 & 1) ? hi2(tq[j1 >> 1]) : lo2(tq[j1 >> 1]));
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const unsigned b = aE2w + 2048 * g;  // j1 = 4g .. 4g+3, 512 bytes = 64 units apart
                    WR2_64(b, o[4 * g], o[4 * g + 1], 0, 64);
                    WR2_64(b, o[4 * g + 2], o[4 * g + 3], 128, 192);
                }
            }
            wave_lds_sync();

            // ---- 3. last radix-4 of the base q and of its mirror; second half: combine + real-input split ----
            MEL4K_PHASE(4);
            {
                v4f zalo[2], zahi[2], zblo[2], zbhi[2], clo[2], chi[2], slo[2], shi[2];
                RD128(zalo[0], aAlo, 0);
                RD128(zahi[0], aAhi, 0);
                RD128(zblo[0], aB0lo, 0);
                RD128(zbhi[0], aB0hi, 0);
                if (half == 1) {
                    RD128(clo[0], aTclo, 0);
                    RD128(chi[0], aTchi, 0);
                    RD128(slo[0], aTclo, 4096);
                    RD128(shi[0], aTchi, 4096);
                }
                RD128(zalo[1], aAlo, 2048);
                RD128(zahi[1], aAhi, 2048);
                RD128(zblo[1], aB1lo, 0);
                RD128(zbhi[1], aB1hi, 0);
                if (half == 1) {
                    RD128(clo[1], aTclo, 2048);
                    RD128(chi[1], aTchi, 2048);
                    RD128(slo[1], aTclo, 4096 + 2048);
                    RD128(shi[1], aTchi, 4096 + 2048);
                }
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    if (s == 1) LDS_WAIT_N(0);
                    else if (half == 1) LDS_WAIT_N(8);
                    else LDS_WAIT_N(4);
                    PIN(zalo[s]); PIN(zahi[s]); PIN(zblo[s]); PIN(zbhi[s]);
                    v2 za0 = lo2(zalo[s]), za1 = hi2(zalo[s]), za2 = lo2(zahi[s]), za3 = hi2(zahi[s]);
                    v2 zb0 = lo2(zblo[s]), zb1 = hi2(zblo[s]), zb2 = lo2(zbhi[s]), zb3 = hi2(zbhi[s]);
                    dft4(za0, za1, za2, za3);  // Zh[q + 256 j]
                    dft4(zb0, zb1, zb2, zb3);  // Zh[q' + 256 j], q' the mirror base (lane 0, s = 0: 128)
                    v2 A[4] = {za0, za1, za2, za3};
                    v2 B[4] = {zb3, zb2, zb1, zb0};  // partner of Zh[q + 256 j] is Zh[q' + 256 (3 - j)]
                    v2 zc = za2;                     // Zh[512] in lane 0 (s = 0)
                    if (s == 0) {
                        // lane 0: q = 0 mirrors itself and q' = 128 mirrors itself:
                        // (Z0, Z0), (Z256, Z768), (Z128, Z896), (Z384, Z640); Z512 apart
                        A[2] = lane0 ? zb0 : za2;
                        A[3] = lane0 ? zb1 : za3;
                        B[0] = lane0 ? za0 : zb3;
                        B[1] = lane0 ? za3 : zb2;
                        B[2] = lane0 ? zb3 : zb1;
                        B[3] = lane0 ? zb2 : zb0;
                    }
                    if (half == 0) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            EA[4 * s + j] = A[j];
                            EB[4 * s + j] = B[j];
                        }
                        if (s == 0) ec = zc;
                    } else {
                        PIN(clo[s]); PIN(chi[s]); PIN(slo[s]); PIN(shi[s]);
                        const v2 wc[4] = {lo2(clo[s]), hi2(clo[s]), lo2(chi[s]), hi2(chi[s])};
                        const v2 ws[4] = {lo2(slo[s]), hi2(slo[s]), lo2(shi[s]), hi2(shi[s])};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int i = 4 * s + j;
                            const v2 to = cmul(A[j], wc[j]);       // W_2048^k Zo[k]
                            const v2 u = cmul_conj(B[j], wc[j]);   // conj(W_2048^k) Zo[1024-k]
                            const v2 zk = EA[i] + to, zk1 = EA[i] - to;  // Z[k], Z[k + 1024]
                            const v2 zm = EB[i] - u, zn = EB[i] + u;     // Z[1024 - k], Z[2048 - k]
                            if constexpr (CPLX) {
                                const bool sq = !STFT && a.specMap == 4;
                                v2 x, y;
                                split_pair_c(zk, zn, ws[j], x, y);
                                cplx_map(x, sq, pk[i], ik[CPLX ? i : 0]);
                                cplx_map(v2{y.x, -y.y}, sq, pn[i], in_[CPLX ? i : 0]);
                                split_pair_qc(zm, zk1, ws[j], x, y);
                                cplx_map(x, sq, pm[i], im_[CPLX ? i : 0]);
                                cplx_map(v2{y.x, -y.y}, sq, pq[i], iq[CPLX ? i : 0]);
                            } else {
                                split_pair(zk, zn, ws[j], pk[i], pn[i]);
                                split_pair_q(zm, zk1, ws[j], pm[i], pq[i]);
                            }
                        }
                        if (s == 0) {
                            // bins 512, 1536 (lane 0's values): Z[512] = Ze[512] - i Zo[512], Z[1536] = Ze[512] + i Zo[512]
                            constexpr float HH = 0.35355339059327376f;  // 0.5 W_4096^512 = 0.5 exp(-i pi / 4)
                            const v2 z5 = pk_add_mi(ec, zc), z15 = pk_add_pi(ec, zc);
                            if constexpr (CPLX) {
                                v2 x, y;
                                split_pair_c(z5, z15, v2{HH, -HH}, x, y);
                                cplx_map(x, !STFT && a.specMap == 4, p512, i512);
                                cplx_map(v2{y.x, -y.y}, !STFT && a.specMap == 4, p1536, i1536);
                            } else {
                                split_pair(z5, z15, v2{HH, -HH}, p512, p1536);
                            }
                        }
                    }
                }
            }
        }
        if constexpr (STFT) {
            // ---- 4'. the spectrum itself: lanes hold consecutive bins, every store instruction covers 256 contiguous bytes.
            //      The next frame's samples have landed (requested half a frame and a frame ago): waited for before the first store
            VM_WAIT_ALL();
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {
                PIN(rlo[n1]);
                PIN(rhi[n1]);
            }
            MEL4K_PHASE(6);
            // the frame's rows through wave-uniform base pointers (the frame number is the same in every lane, which the compiler
            // cannot see) + ONE byte-offset register per family of bins: 128 sixty-four-bit addresses per lane otherwise, and
            // 300 - 440 B of scratch.  ore / oim point at bin 0 of the row.
            const long long row = f * a.outPitch - a.binLo;
            auto uniform = [](const float *p) {
                const unsigned long long u = reinterpret_cast<unsigned long long>(p);
                const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
                return reinterpret_cast<const float *>(((unsigned long long)hi << 32) | lo);
            };
            const float *const ore = uniform(a.out + row), *const oim = uniform(a.outIm ? a.outIm + row : a.out + row);
            const bool two = !MAPPED || a.mode == AFX_SPEC_SQUARE;
            const int lo = a.binLo, hi = a.binLo + a.binCount;
            // bin = cb + (byte offset of the lane) / 4; the value, or for mirrors its conjugate (stftObj_stft keeps all 4096 bins)
            auto put = [&](bool pred, int bin, unsigned voff, int cb, float re, float im) {
                if (!FULL) pred = pred && bin >= lo && bin < hi;
                if (pred) {
                    float v0 = re, v1 = im;
                    if constexpr (MAPPED) stft_map(re, im, a.mode, a.normValue, v0, v1);
                    if (two) GST32X2_S(voff, v0, ore + cb, v1, oim + cb);
                    else GST32_S(voff, v0, ore + cb);
                }
            };
            const unsigned vUp = 4u * lane, vDn = 4u * (64 - lane);          // bins c + lane / c + 64 - lane
            const unsigned vUp23 = lane0 ? 4u * 128 : 4u * (lane + 512);      // slots 2, 3 of s = 0: lane 0 carries base 128 (bins 128, 384)
            const unsigned vDn23 = lane0 ? 4u * 448 : 4u * (64 - lane);       //   ... mirrored: c + 64 - 512 - lane | lane 0: c + 64 - 128
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bool sp = (i == 2 || i == 3);                          // the slots lane 0 uses for base 128
                const int c = 64 * (i >> 2) + 256 * (i & 3);                 // k = c + lane (sp, lane 0: c - 384)
                const int k = sp ? (lane0 ? c - 384 : c + lane) : c + lane;
                const unsigned up = sp ? vUp23 : vUp, dn = sp ? vDn23 : vDn;
                const int cu = sp ? c - 512 : c;                             // k = cu + up / 4
                const int cd = -64 - c;                                      // -k = cd + dn / 4   (sp, lane 0: 448 - 64 - c = -(c - 384))
                const bool kpos = !(i == 0) || !lane0;                       // k > 0
                const float re0 = pk[i], im0 = ik[CPLX ? i : 0], re1 = pn[i], im1 = in_[CPLX ? i : 0];
                const float re2 = pq[i], im2 = iq[CPLX ? i : 0], re3 = pm[i], im3 = im_[CPLX ? i : 0];
                put(true, k, up, cu, re0, im0);                     // X[k]
                put(kpos, 4096 - k, dn, 4096 + cd, re0, -im0);      //   mirror 4096 - k
                put(true, 2048 - k, dn, 2048 + cd, re1, im1);       // X[2048 - k]
                put(kpos, 2048 + k, up, 2048 + cu, re1, -im1);      //   mirror 2048 + k
                put(true, 1024 + k, up, 1024 + cu, re2, im2);       // X[1024 + k]
                put(true, 3072 - k, dn, 3072 + cd, re2, -im2);      //   mirror 3072 - k
                put(true, 1024 - k, dn, 1024 + cd, re3, im3);       // X[1024 - k]  (k = 0: bin 1024 twice, this one last -- as the power row has it)
                put(true, 3072 + k, up, 3072 + cu, re3, -im3);      //   mirror 3072 + k
            }
            put(lane0, 512, vUp, 512, p512, i512);
            put(lane0, 3584, vUp, 3584, p512, -i512);
            put(lane0, 1536, vUp, 1536, p1536, i1536);
            put(lane0, 2560, vUp, 2560, p1536, -i1536);
            wave_lds_sync();  // the next frame overwrites the images
        } else {
#pragma unroll
        for (int pass = 0; pass < (CPLX ? 2 : 1); ++pass) {
        // every read of the image has returned (lgkmcnt(0) above): the power row may overwrite it
        // (CPLX: the second pass writes the imaginary parts over the row the first pass has read)
        if (CPLX && pass == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                pk[i] = ik[CPLX ? i : 0];
                pn[i] = in_[CPLX ? i : 0];
                pm[i] = im_[CPLX ? i : 0];
                pq[i] = iq[CPLX ? i : 0];
            }
            p512 = i512;
            p1536 = i1536;
        }
        // bins k (s = 0: j 0, 1 | 2, 3; s = 1: j 0, 1 | 2, 3) and 1024 + k
        WR2ST_32(aP01, pk[0], pk[1], 0, 4);
        WR2ST_32(aP23, pk[2], pk[3], 0, 4);
        WR2ST_32(aP01, pk[4], pk[5], 1, 5);
        WR2ST_32(aP01, pk[6], pk[7], 9, 13);
        WR2ST_32(aP01, pq[0], pq[1], 16, 20);
        WR2ST_32(aP23, pq[2], pq[3], 16, 20);
        WR2ST_32(aP01, pq[4], pq[5], 17, 21);
        WR2ST_32(aP01, pq[6], pq[7], 25, 29);
        // bins 1024 - k and 2048 - k
        WR2ST_32(aQs1, pm[1], pm[0], 9, 13);
        WR2ST_32(aQ23, pm[3], pm[2], 0, 4);
        WR2ST_32(aQs1, pm[7], pm[6], 0, 4);
        WR2ST_32(aQs1, pm[5], pm[4], 8, 12);
        WR2ST_32(aQs1, pn[1], pn[0], 25, 29);
        WR2ST_32(aQ23, pn[3], pn[2], 16, 20);
        WR2ST_32(aQs1, pn[7], pn[6], 16, 20);
        WR2ST_32(aQs1, pn[5], pn[4], 24, 28);
        if (lane0) {
            prow[512] = p512;
            prow[1536] = p1536;
        }
        wave_lds_sync();
        if (!CPLX && a.specMap) {  // magnitude / norm exponent (rare modes): one pass over the row in LDS (afx_melfused2.hip)
            for (int k = lane; k < 2049; k += 64) {
                const float p = prow[k];
                prow[k] = a.specMap == 1 ? sqrtf(p) : powf(p, a.normValue);
            }
            wave_lds_sync();
        }

        MEL4K_PHASE(5);
        // ---- 4. banded filter bank (afx_melfused2.hip): weights by ds_read_b128, power row by immediate-offset
        //         ds_read_b64; the NEXT block of four quads is requested before this block's values are waited for ----
        float accA, accB;
        {
            constexpr int QA = TA / 4, QB = TB / 4, QT = QA + QB, BLK = 4, NB = (QT + BLK - 1) / BLK;
            v2 sA = {0.f, 0.f}, sB = {0.f, 0.f};
            v4f w[2][BLK];
            v2 p0[2][BLK], p1[2][BLK];
            auto request = [&](int blk, v4f (&wq)[BLK], v2 (&q0v)[BLK], v2 (&q1v)[BLK]) {
#pragma unroll
                for (int i = 0; i < BLK; ++i) {
                    const int q = blk * BLK + i;
                    if (q >= QT) continue;
                    RD128(wq[i], awr, 16 * q);
                    if (q < QA) {
                        RD64(q0v[i], apa, 16 * q);
                        RD64(q1v[i], apa, 16 * q + 8);
                    } else {
                        RD64(q0v[i], apb, 16 * (q - QA));
                        RD64(q1v[i], apb, 16 * (q - QA) + 8);
                    }
                }
            };
            request(0, w[0], p0[0], p1[0]);
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) {
                const int cur = blk & 1;
                const int nextQuads = (blk + 1 < NB) ? ((QT - (blk + 1) * BLK) < BLK ? (QT - (blk + 1) * BLK) : BLK) : 0;
                if (blk + 1 < NB) request(blk + 1, w[cur ^ 1], p0[cur ^ 1], p1[cur ^ 1]);
                if (nextQuads == 4) LDS_WAIT_N(12);
                else if (nextQuads == 3) LDS_WAIT_N(9);
                else if (nextQuads == 2) LDS_WAIT_N(6);
                else if (nextQuads == 1) LDS_WAIT_N(3);
                else LDS_WAIT_N(0);
#pragma unroll
                for (int i = 0; i < BLK; ++i) {
                    if (blk * BLK + i >= QT) continue;
                    PIN(w[cur][i]);
                    PIN(p0[cur][i]);
                    PIN(p1[cur][i]);
                    const int q = blk * BLK + i;
                    if (q < QA) {
                        sA += lo2(w[cur][i]) * p0[cur][i];
                        sA += hi2(w[cur][i]) * p1[cur][i];
                    } else {
                        sB += lo2(w[cur][i]) * p0[cur][i];
                        sB += hi2(w[cur][i]) * p1[cur][i];
                    }
                }
            }
            accA = hsum(sA);
            accB = hsum(sB);
        }
        if (!CPLX && !SPLIT && a.postPow) {
            accA = powf(accA, a.normValue);
            accB = powf(accB, a.normValue);
        }
        MEL4K_PHASE(6);
        // ---- 5. the next frame's samples have landed (requested half a frame and a frame ago); store ----
        if (pass == 0) {
            VM_WAIT_ALL();
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {
                PIN(rlo[n1]);
                PIN(rhi[n1]);
            }
        }
        if constexpr (CC && !SPLIT) {
            // the cepstra of the 16 rows stored BEFORE this one: their stores are a frame old, the block's wait finds them complete
            if (ccN == 16) {
                ccb_rows<2>(a.out, a.cc, a.dct, a.num, a.ccNum, a.ccCbrt, f - 16, 16, lane);
                ccN = 0;
            }
        }
        float *orow = ((CPLX && pass) ? a.outIm : a.out) + f * a.num;
        if constexpr (SPLIT) {
            // slot results -> LDS (start of the wave's region: the image there is dead since stage 3, the power row
            // starts behind it), then every row is the sum of its segments in ascending bins
            float *part = reinterpret_cast<float *>(wreg);
            part[lane] = accA;
            part[64 + lane] = accB;
            if (lane0) part[128] = 0.f;
            wave_lds_sync();
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const unsigned u = h ? seg1 : seg0;
                float sum = part[u & 255u] + part[(u >> 8) & 255u];
                sum += part[(u >> 16) & 255u];
                sum += part[u >> 24];
                if (!CPLX && a.postPow) sum = powf(sum, a.normValue);
                if (lane + 64 * h < a.num) orow[lane + 64 * h] = sum;
            }
        } else {
            if (rowA >= 0) orow[rowA] = accA;
            if (rowB >= 0) orow[rowB] = accB;
        }
        wave_lds_sync();  // the next frame overwrites the images / the power row
        if constexpr (CC) {
            // split plans: ONE call site, behind the row's stores where the band stage's values are dead (its wait then covers the
            // 16th row's stores); whole-row plans: only the wave's last rows here
            ++ccN;
            if ((SPLIT && ccN == 16) || f + 1 == fEnd) {
                ccb_rows<2>(a.out, a.cc, a.dct, a.num, a.ccNum, a.ccCbrt, f + 1 - ccN, ccN, lane);
                ccN = 0;
            }
        }

        }  // pass
        }  // !STFT

        if (++t == a.timeLength) {
            t = 0;
            ++clip;
        }
    }
}

// host: the transform's tables at their byte offsets in the blob (twiddles in double, rounded once); window == nullptr leaves
// the window part alone (STFT instantiations build it in the kernel from the object's device window)
void fill_transform_tables(float *tab, const float *hWindow) {
    const double PI = 3.14159265358979323846;
    // window of each half and W_1024^(lane k1) in pair layout: entry (n1, lane) at float2 index 128 (n1 >> 1) + 2 lane + (n1 & 1)
    float *win = tab + T_WIN / 4, *tw1 = tab + T_TW1 / 4, *tw2 = tab + T_TW2 / 4, *twc = tab + T_TWC / 4, *tws = tab + T_TWS / 4;
    for (int n1 = 0; n1 < 16; ++n1)
        for (int l = 0; l < 64; ++l) {
            const int at = 2 * (128 * (n1 >> 1) + 2 * l + (n1 & 1));
            const int m = 64 * n1 + l;
            for (int h = 0; h < 2 && hWindow; ++h) {
                win[2048 * h + at] = hWindow[4 * m + 2 * h];
                win[2048 * h + at + 1] = hWindow[4 * m + 2 * h + 1];
            }
            const double ang = -2.0 * PI * (double)(n1 * l) / MC;
            tw1[at] = (float)cos(ang);
            tw1[at + 1] = (float)sin(ang);
        }
    for (int m = 0; m < 4; ++m)
        for (int j = 0; j < 16; ++j) {
            const double ang = -2.0 * PI * (double)(m * j) / 64.0;
            tw2[(TW2_PITCH / 4) * m + 2 * j] = (float)cos(ang);
            tw2[(TW2_PITCH / 4) * m + 2 * j + 1] = (float)sin(ang);
        }
    // W_2048^bin and 0.5 W_4096^bin of slot (s, lane, m); 16-byte halves swapped when bit 3 of lane is set
    for (int s = 0; s < 2; ++s)
        for (int l = 0; l < 64; ++l)
            for (int m = 0; m < 4; ++m) {
                int bin = l + 64 * s + 256 * m;
                if (s == 0 && l == 0 && m >= 2) bin = m == 2 ? 128 : 384;  // lane 0 carries the self-mirrored base
                const int at = 2 * (4 * (64 * s + l) + 2 * ((m >> 1) ^ ((l >> 3) & 1)) + (m & 1));
                const double ac = -2.0 * PI * (double)bin / 2048.0, as = -2.0 * PI * (double)bin / NFFT;
                twc[at] = (float)cos(ac);
                twc[at + 1] = (float)sin(ac);
                tws[at] = (float)(0.5 * cos(as));
                tws[at + 1] = (float)(0.5 * sin(as));
            }
}

// the STFT instantiations' twiddle blob (bytes [T_TW1, tab_bytes(0, 0)) are read), one device copy per device, never freed
const float4 *stft_tables(void *stream) {
    static std::mutex mu;
    static float4 *dTab[AFX_MAX_DEVICES] = {};
    const int dev = afxdev_current_device();
    if (dev < 0 || dev >= AFX_MAX_DEVICES) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    if (!dTab[dev]) {
        const size_t bytes = (size_t)tab_bytes(0, 0);
        float *h = static_cast<float *>(calloc(bytes, 1));
        float4 *d = nullptr;
        if (!h) return nullptr;
        fill_transform_tables(h, nullptr);
        int st = afxdev_malloc(reinterpret_cast<void **>(&d), bytes);
        // (a synchronous copy, like wave_tables() of afx_stft.hip: the caller's stream is not waited for under this lock)
        if (st == AFX_OK && hipMemcpy(d, h, bytes, hipMemcpyHostToDevice) != hipSuccess) st = AFX_ERR_HIP;
        free(h);
        if (st != AFX_OK) {
            afxdev_free(d);
            return nullptr;
        }
        dTab[dev] = d;
    }
    return dTab[dev];
}

template <int SHIFT, bool MAPPED, bool FULL>
int launch_stft(const AfxStftArgs *a, const float4 *tab, void *stream) {
    const long long total = (long long)a->batch * a->timeLength;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    long long waves = (long long)cus * WAVES * 2;
    long long fpw = (total + waves - 1) / waves;
    if (fpw < 16) {  // (launch_variant: a call that cannot fill one round of workgroups is spread over all CUs)
        const long long oneRound = (total + (long long)cus * WAVES - 1) / ((long long)cus * WAVES);
        fpw = oneRound < 16 ? oneRound : 16;
    }
    const long long usedWaves = (total + fpw - 1) / fpw;
    const long long blocks = (usedWaves + WAVES - 1) / WAVES;
    KArgs4 k;
    memset(&k, 0, sizeof(k));
    k.x = a->x;
    k.clipStride = a->clipStride;
    k.totalFrames = total;
    k.timeLength = a->timeLength;
    k.hop = a->hop;
    k.framesPerWave = (int)fpw;
    k.tab = tab;
    k.specMap = 3;
    k.normValue = a->normValue;
    k.out = a->outRe;
    k.outIm = a->outIm;
    k.window = a->window;
    k.mode = a->mode;
    k.binLo = a->binLo;
    k.binCount = a->binCount;
    k.outPitch = a->outPitch ? a->outPitch : (long long)a->binCount;
    constexpr size_t lds = (size_t)block_lds_bytes(0, 0);
    static std::atomic<bool> attrSet[AFX_MAX_DEVICES];
    const int attrDev = afxdev_current_device() & (AFX_MAX_DEVICES - 1);
    if (!attrSet[attrDev].load(std::memory_order_acquire)) {  // (two threads may both set it: idempotent)
        AFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_stft_band_4k2<0, 0, SHIFT, false, true, true, MAPPED, FULL>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attrSet[attrDev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((k_stft_band_4k2<0, 0, SHIFT, false, true, true, MAPPED, FULL>), dim3((unsigned)blocks), dim3(WAVES * 64), lds,
                       (hipStream_t)stream, k);
    AFX_LAUNCH_CHECK("k_stft_band_4k2<stft>");
    return AFX_OK;
}

struct Plan4 {
    int variant;  // 200 + index into kVariants: FIRST field (afx_melfused.hip: variant >= 200 -> this file)
    int num, split;
    float4 *dTab;
    int *dMeta;
};
struct Variant {
    int tapsA, tapsB;
};
constexpr Variant kVariants[] = {{96, 32}, {128, 64}, {176, 8}};
static_assert(block_lds_bytes(96, 32) <= 163840 && block_lds_bytes(128, 64) <= 163840 && block_lds_bytes(176, 8) <= 163840,
              "tables + weights + 8 wave regions must fit the 160 KB LDS");

template <int TA, int TB, int SHIFT, bool SPLIT, bool CPLX, bool CC = false>
int launch_variant(const Plan4 *p, const AfxMelFusedArgs *a, void *stream) {
    const long long total = (long long)a->batch * a->timeLength;
    if (total <= 0) return AFX_OK;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    long long waves = (long long)cus * WAVES * 2;  // two rounds of workgroups (afx_melfused2.hip)
    long long fpw = (total + waves - 1) / waves;
    // long runs per wave (register re-use of the overlapping frames) once a round of workgroups is full; a call that
    // cannot fill one round -- the one-clip legacy entry points: 1000 frames -- is spread over all CUs instead
    // (16 frames in sequence per wave were 75 us of a 1000-frame call's 190, profiles/r05_legacy_phases.txt)
    if (fpw < 16) {
        const long long oneRound = (total + (long long)cus * WAVES - 1) / ((long long)cus * WAVES);
        fpw = oneRound < 16 ? oneRound : 16;
    }
    const long long usedWaves = (total + fpw - 1) / fpw;
    const long long blocks = (usedWaves + WAVES - 1) / WAVES;

    KArgs4 k;
    memset(&k, 0, sizeof(k));
    k.x = a->x;
    k.clipStride = a->clipStride;
    k.totalFrames = total;
    k.timeLength = a->timeLength;
    k.hop = a->hop;
    k.framesPerWave = (int)fpw;
    k.aligned = ((a->clipStride & 3) == 0) && ((a->hop & 3) == 0) && ((reinterpret_cast<uintptr_t>(a->x) & 15) == 0);
    k.tab = p->dTab;
    k.meta = p->dMeta;
    k.specMap = a->specMap;
    k.postPow = a->postPow;
    k.normValue = a->normValue;
    k.out = a->out;
    k.outIm = a->outIm;
    k.num = p->num;
    k.dct = a->dct;
    k.ccNum = a->ccNum;
    k.ccCbrt = a->ccRectify == 1;
    k.cc = a->cc;
    constexpr size_t lds = (size_t)block_lds_bytes(TA, TB);
    static std::atomic<bool> attrSet[AFX_MAX_DEVICES];  // per device: the attribute lives in the device's code object
    const int attrDev = afxdev_current_device() & (AFX_MAX_DEVICES - 1);
    if (!attrSet[attrDev].load(std::memory_order_acquire)) {  // (two threads may both set it: idempotent)
        AFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_stft_band_4k2<TA, TB, SHIFT, SPLIT, CPLX, false, false, false, CC>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attrSet[attrDev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((k_stft_band_4k2<TA, TB, SHIFT, SPLIT, CPLX, false, false, false, CC>), dim3((unsigned)blocks), dim3(WAVES * 64), lds,
                       (hipStream_t)stream, k);
    AFX_LAUNCH_CHECK("k_stft_band_4k2");
    return AFX_OK;
}

template <int TA, int TB, bool CPLX>
int launch_mode(const Plan4 *p, const AfxMelFusedArgs *a, void *stream) {
    // register re-use of the overlapping frames at the wrapper's default hop = N/4; other hops fetch every frame whole
    if (a->hop == 1024)
        return p->split ? launch_variant<TA, TB, 4, true, CPLX>(p, a, stream) : launch_variant<TA, TB, 4, false, CPLX>(p, a, stream);
    return p->split ? launch_variant<TA, TB, 0, true, CPLX>(p, a, stream) : launch_variant<TA, TB, 0, false, CPLX>(p, a, stream);
}

template <int TA, int TB>
int launch(const Plan4 *p, const AfxMelFusedArgs *a, void *stream) {
    if (a->specMap >= 3) {  // complex results: S (3) or S^2 (4)
        if (a->cc) return AFX_ERR_UNSUPPORTED;
        if (!a->outIm) return AFX_ERR_ARG;
        return launch_mode<TA, TB, true>(p, a, stream);
    }
    if (a->cc) {  // cepstra in the same launch (real results, every real mode)
        if (a->ccNum < 1 || a->ccNum > 16 || !a->dct || !a->out || p->num > 128 || (p->num & 3) || (a->ccRectify != 0 && a->ccRectify != 1) || a->energy)
            return AFX_ERR_UNSUPPORTED;
        if (a->hop == 1024)
            return p->split ? launch_variant<TA, TB, 4, true, false, true>(p, a, stream) : launch_variant<TA, TB, 4, false, false, true>(p, a, stream);
        return p->split ? launch_variant<TA, TB, 0, true, false, true>(p, a, stream) : launch_variant<TA, TB, 0, false, false, true>(p, a, stream);
    }
    return launch_mode<TA, TB, false>(p, a, stream);
}

}  // namespace

// ---- the n_fft 4096 entry points of the fused dispatcher (afx_melfused.hip reads Plan4.variant >= 200 as "this file") ----
extern "C" int afxk_mel4k_variant(int tapsA, int tapsB) {
    for (int i = 0; i < 3; ++i)
        if (tapsA <= kVariants[i].tapsA && tapsB <= kVariants[i].tapsB) return 200 + i;
    return -1;
}

extern "C" int afxk_mel4k_kind(const void *plan) {
    const Plan4 *p = static_cast<const Plan4 *>(plan);
    return !p ? 0 : (p->split ? 202 : 201);
}

extern "C" void afxk_mel4k_destroy(void *plan) {
    Plan4 *p = static_cast<Plan4 *>(plan);
    if (!p) return;
    afxdev_free(p->dTab);
    afxdev_free(p->dMeta);
    free(p);
}

extern "C" int afxk_mel4k_create(void **plan, const float *hWindow, const AfxBandPlan *band, void *stream) {
    *plan = nullptr;
    const int variant = afxk_mel4k_variant(band->tapsA, band->tapsB) - 200;  // index into {96+32, 128+64, 176+8} taps
    if (variant < 0 || variant > 2) return AFX_ERR_UNSUPPORTED;
    const int TA = kVariants[variant].tapsA, TB = kVariants[variant].tapsB;
    const int WP = wpitch(TA, TB);
    const size_t bytes = (size_t)tab_bytes(TA, TB);
    Plan4 *p = static_cast<Plan4 *>(calloc(1, sizeof(Plan4)));
    float *tab = static_cast<float *>(calloc(bytes, 1));
    if (!p || !tab) {
        free(p);
        free(tab);
        return AFX_ERR_NOMEM;
    }
    p->variant = 200 + variant;  // (first field: the dispatcher's tag)
    p->num = band->num;
    p->split = band->split;
    fill_transform_tables(tab, hWindow);
    float *wL = tab + T_BAND / 4;
    for (int l = 0; l < 64; ++l) {
        for (int t = 0; t < band->tapsA; ++t) wL[(size_t)l * WP + t] = band->wA[(size_t)t * 64 + l];
        for (int t = 0; t < band->tapsB; ++t) wL[(size_t)l * WP + TA + t] = band->wB[(size_t)t * 64 + l];
    }
    int meta[384];
    for (int l = 0; l < 64; ++l) {
        meta[l] = band->startA[l];
        meta[64 + l] = band->startB[l];
        meta[128 + l] = band->rowA[l];
        meta[192 + l] = band->rowB[l];
        meta[256 + l] = (int)band->segIdx[l];
        meta[320 + l] = (int)band->segIdx[64 + l];
    }
    int st = afxdev_malloc(reinterpret_cast<void **>(&p->dTab), bytes);
    if (st == AFX_OK) st = afxdev_h2d(p->dTab, tab, bytes, stream);
    if (st == AFX_OK) st = afxdev_malloc(reinterpret_cast<void **>(&p->dMeta), sizeof(meta));
    if (st == AFX_OK) st = afxdev_h2d(p->dMeta, meta, sizeof(meta), stream);
    if (st == AFX_OK) st = afxdev_stream_sync(stream);  // host staging buffers are freed below
    free(tab);
    if (st != AFX_OK) {
        afxk_mel4k_destroy(p);
        return st;
    }
    *plan = p;
    return AFX_OK;
}

// specMap 0 / 1 / 2: real results; 3 / 4: complex results (out + outIm)
extern "C" int afxk_mel4k_run(void *plan, const AfxMelFusedArgs *a, void *stream) {
    if (a->energy) return AFX_ERR_UNSUPPORTED;  // temporal features ride along at n_fft 2048 only (cepstra: every size, launch())
    const Plan4 *p = static_cast<const Plan4 *>(plan);
    if (!p || a->specMap > 4) return AFX_ERR_ARG;
    switch (p->variant) {
        case 200: return launch<96, 32>(p, a, stream);
        case 201: return launch<128, 64>(p, a, stream);
        case 202: return launch<176, 8>(p, a, stream);
        default: return AFX_ERR_UNSUPPORTED;
    }
}

// n_fft 4096 without a bank (afxk_stft, afx_stft.hip): every frame inside its clip (no padding), no temporal features.
// AFX_ERR_UNSUPPORTED: the caller runs the size-generic kernel.
extern "C" int afxk_stft4k(const AfxStftArgs *a, void *stream) {
    if (a->radix2Exp != 12 || a->bandStart || a->energy || a->binLo < 0 || a->binCount < 1 || a->binLo + a->binCount > NFFT ||
        a->padLeft != 0 || a->hop < 1 || (long long)(a->timeLength - 1) * a->hop + NFFT > a->dataLength)
        return AFX_ERR_UNSUPPORTED;
    const bool two = (a->mode == AFX_SPEC_COMPLEX || a->mode == AFX_SPEC_SQUARE);
    if (!a->outRe || (two && !a->outIm)) return AFX_ERR_ARG;
    if ((long long)a->batch * a->timeLength <= 0) return AFX_OK;
    const float4 *tab = stft_tables(stream);
    if (!tab) return AFX_ERR_UNSUPPORTED;
    const bool s4 = a->hop == 1024;  // register re-use of the overlapping frames at the wrapper's default hop
    if (a->mode == AFX_SPEC_COMPLEX) {
        if (a->binLo == 0 && a->binCount == NFFT) return s4 ? launch_stft<4, false, true>(a, tab, stream) : launch_stft<0, false, true>(a, tab, stream);
        return s4 ? launch_stft<4, false, false>(a, tab, stream) : launch_stft<0, false, false>(a, tab, stream);
    }
    return s4 ? launch_stft<4, true, false>(a, tab, stream) : launch_stft<0, true, false>(a, tab, stream);
}
