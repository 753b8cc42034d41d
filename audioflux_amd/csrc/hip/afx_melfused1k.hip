// afx_melfused1k.hip -- the fused STFT -> spectrum value -> banded filter bank kernel for
// n_fft = 1024 (radix2Exp 10): same design as afx_melfused.hip (one wave per frame, tables in
// LDS, register re-use of the overlapping frames, lane-owned bank rows), with the 512-point
// complex FFT of the packed real frame as 8 x 8 x 8 in eight registers per lane -- the
// transform of the CWT row pass (afx_cwt.hip, index algebra in tools/proto_fft512.py).
// Eight data registers and a 4.6 KB exchange image per wave leave room for 16 waves per CU.
//
// Replaces, per frame, the same reference code as afx_melfused.hip (stft_algorithm.c:696-803,
// fft_algorithm.c:450-519, flux_complex.c:254-286,469-503, bft_algorithm.c:457-529,
// flux_vector.c:55-86).
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

#ifndef AFX_CC_GROUPS  // whole-row plans: 16-band groups of the rows requested per trip to the L2, and whether the next trip is requested ahead
#define AFX_CC_GROUPS 2
#define AFX_CC_AHEAD true
#endif

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v2 lo2(v4f q) { return v2{q.x, q.y}; }
__device__ __forceinline__ v2 hi2(v4f q) { return v2{q.z, q.w}; }

constexpr int NFFT = 1024;
constexpr int MC = 512;            // complex FFT length
constexpr int RP = 10;             // exchange pitch (float2) of the 64 x 8 images: 80-byte rows are read as four ds_read_b128 free
                                   // of conflicts (pitch 9: eight half-rate ds_read2_b64) and the first exchange's stores lose
                                   // their two-way conflict (tools/proto_fft1024_v2.py: conflicts())
constexpr int EX_F2 = 64 * RP;     // 640 float2; also holds the 512-float2 natural image
constexpr int PROW_OFF = 3072;     // byte offset of the power row in a wave's region: bins 0..512 alias the images' tail,
constexpr int PROW_F = 640;        // the zero pad of the fixed-length band loops (bins 513..639) lies behind them
constexpr int WAVE_LDS_BYTES = PROW_OFF + PROW_F * 4;  // 5632
static_assert(EX_F2 * 8 <= PROW_OFF + 513 * 4, "the images must end before the zero pad");
constexpr int WAVES = 16;          // one workgroup per CU: 4 waves per SIMD
constexpr int TAB_WIN_F2 = 512;    // (w[2n], w[2n+1])
constexpr int TAB_TW1_F2 = 8 * 64; // W_512^(lane d0)
constexpr int TAB_TW2_F2 = 8 * 8;  // W_64^(c d1)
constexpr int TAB_TW3_F2 = 320;    // 0.5 * W_1024^k, k <= 256
constexpr int TAB_F2 = TAB_WIN_F2 + TAB_TW1_F2 + TAB_TW2_F2 + TAB_TW3_F2;
constexpr int TAB_BYTES = TAB_F2 * 8;
constexpr int T_WIN = 0, T_TW1 = T_WIN + 8 * TAB_WIN_F2, T_TW2 = T_TW1 + 8 * TAB_TW1_F2, T_TW3 = T_TW2 + 8 * TAB_TW2_F2;  // byte offsets
__host__ __device__ constexpr int wpitch(int ta, int tb) { return ta + tb + 4; }
__host__ __device__ constexpr int block_lds_bytes(int ta, int tb) {
    return TAB_BYTES + 64 * wpitch(ta, tb) * 4 + WAVES * WAVE_LDS_BYTES;
}

struct KArgs {
    const float *x;
    long long clipStride;
    long long totalFrames;
    int timeLength, hop;
    int framesPerWave;
    int aligned;
    const float2 *win2, *tw1, *tw2, *tw3;
    const float *wLane;
    const int *meta;
    int specMap, postPow;
    float normValue;
    float *out, *outIm;
    int num;
    // STFT instantiations (afxk_stft1k): bins instead of bank rows
    int mode;              // AFX_SPEC_*
    int binLo, binCount;   // bins binLo .. binLo + binCount - 1 are stored; above 512: conjugate mirrors
    long long outPitch;    // floats between output rows
    // CC instantiations: cepstra of the rows in the same launch (afx_ccblock.h)
    const float *dct;      // device [num, num] orthonormal DCT-II
    int ccNum, ccCbrt;
    float *cc;             // [totalFrames, ccNum]
};

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

// |X|^2 of the conjugate pair (k, 512-k) from A = Z[k], B = Z[512-k], w = 0.5 W_1024^k
__device__ __forceinline__ void split_pair(v2 A, v2 B, v2 w, float &pk, float &pq) {
    const v2 e2 = pk_add_conj(A, B);
    const v2 d = pk_sub_conj(A, B);
    const v2 wo = cmul_mi(d, w);
    const v2 x = e2 * 0.5f + wo;  // X[k]
    const v2 y = e2 * 0.5f - wo;  // conj(X[512-k])
    pk = x.x * x.x + x.y * x.y;
    pq = y.x * y.x + y.y * y.y;
}
__device__ __forceinline__ void split_pair_c(v2 A, v2 B, v2 w, bool sq, float &kr, float &ki, float &qr,
                                             float &qi) {
    const v2 e2 = pk_add_conj(A, B);
    const v2 d = pk_sub_conj(A, B);
    const v2 wo = cmul_mi(d, w);
    const v2 x = e2 * 0.5f + wo;
    const v2 y = e2 * 0.5f - wo;
    if (sq) {
        kr = x.x * x.x - x.y * x.y;
        ki = 2.f * (x.x * x.y);
        qr = y.x * y.x - y.y * y.y;
        qi = -2.f * (y.x * y.y);
    } else {
        kr = x.x;
        ki = x.y;
        qr = y.x;
        qi = -y.y;
    }
}

// STFT: no bank -- the spectrum values themselves (CPLX form) go to memory through stft_map (the STFT object's full complex
//   spectrum, linear-scale bin slices: afxk_stft1k, afx_melfused4k2.hip has the same at n_fft 4096); MAPPED: any AFX_SPEC_* map,
//   otherwise the complex values as they are; FULL: all 1024 bins are stored (no range checks)
// SPLIT: the plan's slots hold row SEGMENTS (afx_bandplan_build_split): banks whose rows are longer than the tap variants
// CC: cepstra of the rows in the same launch (real results; afx_ccblock.h: every 16 frames the wave re-reads its rows from L2)
template <int TA, int TB, bool GENERAL, int SHIFT, bool CPLX, bool STFT = false, bool MAPPED = false, bool FULL = false, bool SPLIT = false, bool CC = false>
__global__ __launch_bounds__(WAVES * 64) void k_stft_band_1k(KArgs a) {
    static_assert(!STFT || (CPLX && TA == 0 && TB == 0 && !SPLIT), "STFT instantiations: complex values, no bank");
    static_assert(!CC || (!CPLX && !STFT), "cepstra: real bank rows");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    constexpr int WP = wpitch(TA, TB);
    unsigned char *wreg = smem + TAB_BYTES + 64 * WP * 4 + wave * WAVE_LDS_BYTES;
    float *prow = reinterpret_cast<float *>(wreg + PROW_OFF);
    {
        v2 *tabWin = reinterpret_cast<v2 *>(smem + T_WIN);
        v2 *tabTw1 = reinterpret_cast<v2 *>(smem + T_TW1);
        v2 *tabTw2 = reinterpret_cast<v2 *>(smem + T_TW2);
        v2 *tabTw3 = reinterpret_cast<v2 *>(smem + T_TW3);
        float *tabW = reinterpret_cast<float *>(smem + TAB_BYTES);
        for (int i = threadIdx.x; i < TAB_WIN_F2; i += WAVES * 64) tabWin[i] = reinterpret_cast<const v2 *>(a.win2)[i];
        for (int i = threadIdx.x; i < TAB_TW1_F2; i += WAVES * 64) tabTw1[i] = reinterpret_cast<const v2 *>(a.tw1)[i];
        for (int i = threadIdx.x; i < TAB_TW3_F2; i += WAVES * 64) tabTw3[i] = reinterpret_cast<const v2 *>(a.tw3)[i];
        if constexpr (!STFT)
            for (int i = threadIdx.x; i < 64 * WP; i += WAVES * 64) tabW[i] = a.wLane[i];
        if (threadIdx.x < TAB_TW2_F2) tabTw2[threadIdx.x] = reinterpret_cast<const v2 *>(a.tw2)[threadIdx.x];
        if constexpr (CC)  // the DCT operand of the cepstrum block, behind the wave regions (afx_ccblock.h)
            ccb_table_fill(reinterpret_cast<float *>(smem + block_lds_bytes(TA, TB)), a.dct, a.num, a.ccNum, threadIdx.x, WAVES * 64);
        // zero pad behind bin 512 (the fixed-length band loops read it with zero weights): behind the images, written once
        for (int i = 513 + lane; i < PROW_F; i += 64) prow[i] = 0.f;
    }
    __syncthreads();

    // ---- per-lane constants: loop-invariant LDS byte addresses of the hand-issued reads and writes (afx_asm.h) ----
    const int b = lane >> 3, c = lane & 7;
    const bool lane0 = (lane == 0);
    const unsigned T0 = lds_addr(smem), W0 = lds_addr(wreg);
    const unsigned aWin = T0 + T_WIN + 8 * lane;       // window row r: + 512 r;  W_512^(lane d0): + T_TW1 + 512 d0
    const unsigned aTw2 = T0 + T_TW2 + 8 * c;          // W_64^(c d1): + 64 d1
    const unsigned aT3 = T0 + T_TW3 + 8 * lane;        // 0.5 W_1024^(lane + 64 j): + 512 j
    const unsigned aE1w = W0 + 8 * (c * RP + b);       // first exchange, row 8 d0 + c: + 64 RP d0
    const unsigned aE2w = W0 + 8 * (b * RP + c);       // second exchange (d0 = lane >> 3), row d0 + 8 d1: + 64 RP d1
    const unsigned aEr = W0 + 8 * RP * lane;           // a lane's row of either image: 4 x 16 bytes
    const unsigned aN = W0 + 8 * lane;                 // natural image Z[lane + 64 d2]: + 512 d2
    const unsigned aNm = W0 + 8 * (320 - lane);        // Z[512 - lane - 64 j]: + 512 (3 - j)   (lane 0, j = 0: see below)
    const unsigned aMid = W0 + 2048;                   // Z[256]
    const unsigned aT3m = T0 + T_TW3 + 2048;           // 0.5 W_1024^256
    const unsigned R = W0 + PROW_OFF;
    const unsigned aP = R + 4 * lane;                  // bins lane + 64 j: + 256 j
    const unsigned aQ = R + 4 * (320 - lane);          // bins 512 - lane - 64 j: + 256 (3 - j)

    const int startA = STFT ? 0 : a.meta[lane], startB = STFT ? 0 : a.meta[64 + lane];
    const int rowA = STFT ? -1 : a.meta[128 + lane], rowB = STFT ? -1 : a.meta[192 + lane];
    // (split plans with cepstra sit at the 128-register cap: they re-read their two segment words per frame -- vector cache --
    // instead of carrying them through the loop)
    const unsigned seg0 = (SPLIT && !CC) ? (unsigned)a.meta[256 + lane] : 0u, seg1 = (SPLIT && !CC) ? (unsigned)a.meta[320 + lane] : 0u;
    const unsigned apa = R + 4 * startA, apb = R + 4 * startB;
    const unsigned awr = T0 + TAB_BYTES + 4 * WP * lane;

    const long long gw = (long long)blockIdx.x * WAVES + wave;
    long long f = gw * a.framesPerWave;
    long long fEnd = f + a.framesPerWave;
    if (fEnd > a.totalFrames) fEnd = a.totalFrames;
    if (f >= fEnd) return;
    int clip = (int)(f / a.timeLength);
    int t = (int)(f - (long long)clip * a.timeLength);
    int ccN = 0;  // CC: rows of this wave whose cepstra are still to be formed
    const float *const ccTab = reinterpret_cast<const float *>(smem + block_lds_bytes(TA, TB));

    // raw[r] = (x[2n], x[2n+1]), n = 64 r + lane
    v2 raw[8];
    auto fetch = [&](const float *px, int first) {
        if (a.aligned) {
            const v2 *p2 = reinterpret_cast<const v2 *>(px);
#pragma unroll
            for (int r = 0; r < 8; ++r)
                if (r >= first) raw[r] = p2[64 * r + lane];
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r)
                if (r >= first) {
                    const int n = 64 * r + lane;
                    raw[r] = v2{px[2 * n], px[2 * n + 1]};
                }
        }
    };
    fetch(a.x + (long long)clip * a.clipStride + (long long)t * a.hop, 0);

    for (; f < fEnd; ++f) {
        v2 v[8];
        // ---- 1. window (the first half is used while the second lands); start fetching the next frame ----------
        {
            v2 wv[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) RD64(wv[r], aWin, T_WIN + 512 * r);
            LDS_WAIT_N(4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                PIN(wv[r]);
                v[r] = raw[r] * wv[r];
            }
            LDS_WAIT_N(0);
#pragma unroll
            for (int r = 4; r < 8; ++r) {
                PIN(wv[r]);
                v[r] = raw[r] * wv[r];
            }
        }
        if (f + 1 < fEnd) {
            int tn = t + 1, cn = clip;
            if (tn == a.timeLength) {
                tn = 0;
                ++cn;
            }
            const float *pn = a.x + (long long)cn * a.clipStride + (long long)tn * a.hop;
            bool whole = true;
            if constexpr (SHIFT > 0) {  // hop = 128 SHIFT samples = SHIFT registers, moved in place (afx_asm.h)
                if (tn != 0) {
                    shift_rows8_inplace<SHIFT>(raw);
                    fetch(pn, 8 - SHIFT);
                    whole = false;
                }
            }
            if (whole) fetch(pn, 0);
        }
        // ---- 2. 512-point complex FFT, 8 x 8 x 8: twiddles requested ahead of the butterflies that hide them ----
        {
            v2 tw[8];
#pragma unroll
            for (int d = 1; d < 8; ++d) RD64(tw[d], aWin, T_TW1 + 512 * d);
            dft8(v);  // v[rev8(d0)]
#pragma unroll
            for (int i = 0; i < 8; ++i) PIN(v[i]);
            LDS_WAIT_N(0);
#pragma unroll
            for (int d = 1; d < 8; ++d) PIN(tw[d]);
            v2 o[8];
            o[0] = v[0];
#pragma unroll
            for (int d0 = 1; d0 < 8; ++d0) o[d0] = cmul(v[rev8(d0)], tw[d0]);
#pragma unroll
            for (int g = 0; g < 4; ++g) WR2_64(aE1w, o[2 * g], o[2 * g + 1], 8 * RP * (2 * g), 8 * RP * (2 * g + 1));
        }
        wave_lds_order();
        {
            v4f rq[4];
            v2 tw[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) RD128(rq[i], aEr, 16 * i);
#pragma unroll
            for (int d = 1; d < 8; ++d) RD64(tw[d], aTw2, 64 * d);
            LDS_WAIT_N(7);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                PIN(rq[i]);
                v[2 * i] = lo2(rq[i]);
                v[2 * i + 1] = hi2(rq[i]);
            }
            dft8(v);  // v[rev8(d1)], lane = 8 d0 + c
#pragma unroll
            for (int i = 0; i < 8; ++i) PIN(v[i]);
            LDS_WAIT_N(0);
#pragma unroll
            for (int d = 1; d < 8; ++d) PIN(tw[d]);
            v2 o[8];
            o[0] = v[0];
#pragma unroll
            for (int d1 = 1; d1 < 8; ++d1) o[d1] = cmul(v[rev8(d1)], tw[d1]);
#pragma unroll
            for (int g = 0; g < 4; ++g) WR2_64(aE2w, o[2 * g], o[2 * g + 1], 8 * RP * (2 * g), 8 * RP * (2 * g + 1));
        }
        wave_lds_order();
        {
            v4f rq[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) RD128(rq[i], aEr, 16 * i);
            LDS_WAIT_N(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                PIN(rq[i]);
                v[2 * i] = lo2(rq[i]);
                v[2 * i + 1] = hi2(rq[i]);
            }
        }
        dft8(v);  // v[rev8(d2)] = Z[lane + 64 d2]
        // ---- 3. natural-order image, conjugate pairs (k, 512-k), spectrum values ---------
#pragma unroll
        for (int g = 0; g < 4; ++g) WR2_64(aN, v[rev8(2 * g)], v[rev8(2 * g + 1)], 64 * (2 * g), 64 * (2 * g + 1));
        wave_lds_order();
        float pk[5], pq[5];
        float pkI[CPLX ? 5 : 1], pqI[CPLX ? 5 : 1];
        {
            v2 za[4], zb[4], w3[4], zm, wm;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                RD64(za[j], aN, 512 * j);
                RD64(zb[j], aNm, 512 * (3 - j));  // Z[512 - k], k = lane + 64 j
                RD64(w3[j], aT3, 512 * j);
            }
            RD64(zm, aMid, 0);  // bin 256 pairs with itself
            RD64(wm, aT3m, 0);
            LDS_WAIT_N(8);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j == 2) LDS_WAIT_N(0);
                PIN(za[j]);
                PIN(zb[j]);
                PIN(w3[j]);
                // k = 0 pairs with itself (X[0] and X[512]); lane 0's read of slot 512 is past the image and discarded
                const v2 zbj = (j == 0) ? (lane0 ? za[0] : zb[0]) : zb[j];
                if (CPLX) split_pair_c(za[j], zbj, w3[j], !STFT && a.specMap == 4, pk[j], pkI[CPLX ? j : 0], pq[j], pqI[CPLX ? j : 0]);
                else split_pair(za[j], zbj, w3[j], pk[j], pq[j]);
            }
            PIN(zm);
            PIN(wm);
            if (CPLX) split_pair_c(zm, zm, wm, !STFT && a.specMap == 4, pk[4], pkI[CPLX ? 4 : 0], pq[4], pqI[CPLX ? 4 : 0]);
            else split_pair(zm, zm, wm, pk[4], pq[4]);
        }
        if (CPLX) {
        } else if (GENERAL && a.specMap == 1) {
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                pk[i] = sqrtf(pk[i]);
                pq[i] = sqrtf(pq[i]);
            }
        } else if (GENERAL && a.specMap == 2) {
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                pk[i] = powf(pk[i], a.normValue);
                pq[i] = powf(pq[i], a.normValue);
            }
        }
        if constexpr (STFT) {
            // ---- 4'. the spectrum itself: lanes hold consecutive bins; wave-uniform row bases in scalar registers + ONE byte-offset
            //      register per family of bins (afx_melfused4k2.hip).  ore / oim point at bin 0 of the row.
            const long long row = f * a.outPitch - a.binLo;
            auto uniform = [](const float *p) {
                const unsigned long long u = reinterpret_cast<unsigned long long>(p);
                const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
                return reinterpret_cast<const float *>(((unsigned long long)hi << 32) | lo);
            };
            const float *const ore = uniform(a.out + row), *const oim = uniform(a.outIm ? a.outIm + row : a.out + row);
            const bool two = !MAPPED || a.mode == AFX_SPEC_SQUARE;
            const int lo = a.binLo, hi = a.binLo + a.binCount;
            auto put = [&](bool pred, int bin, unsigned voff, int cb, float re, float im) {
                if (!FULL) pred = pred && bin >= lo && bin < hi;
                if (pred) {
                    float v0 = re, v1 = im;
                    if constexpr (MAPPED) stft_map(re, im, a.mode, a.normValue, v0, v1);
                    if (two) GST32X2_S(voff, v0, ore + cb, v1, oim + cb);
                    else GST32_S(voff, v0, ore + cb);
                }
            };
            const unsigned vUp = 4u * lane, vDn = 4u * (64 - lane);  // bins c + lane / c + 64 - lane
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = lane + 64 * j;
                const bool kpos = j > 0 || !lane0;  // k > 0: bins 0 and 512 have no mirror
                const float re0 = pk[j], im0 = pkI[CPLX ? j : 0], re1 = pq[j], im1 = pqI[CPLX ? j : 0];
                put(true, k, vUp, 64 * j, re0, im0);                    // X[k]
                put(kpos, NFFT - k, vDn, NFFT - 64 - 64 * j, re0, -im0);  //   mirror 1024 - k
                put(true, MC - k, vDn, MC - 64 - 64 * j, re1, im1);      // X[512 - k]
                put(kpos, MC + k, vUp, MC + 64 * j, re1, -im1);          //   mirror 512 + k
            }
            put(lane0, 256, vUp, 256, pk[4], pkI[CPLX ? 4 : 0]);
            put(lane0, 768, vUp, 768, pk[4], -pkI[CPLX ? 4 : 0]);
        } else {
        // every read of the image has returned (lgkmcnt(0) above): the power row may overwrite its tail
#pragma unroll
        for (int pass = 0; pass < (CPLX ? 2 : 1); ++pass) {
            if (CPLX && pass == 1) {
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    pk[i] = pkI[CPLX ? i : 0];
                    pq[i] = pqI[CPLX ? i : 0];
                }
            }
            WR2ST_32(aP, pk[0], pk[1], 0, 1);
            WR2ST_32(aP, pk[2], pk[3], 2, 3);
            WR2ST_32(aQ, pq[3], pq[2], 0, 1);
            WR2ST_32(aQ, pq[1], pq[0], 2, 3);
            if (lane0) prow[256] = pk[4];
            wave_lds_order();

            // ---- 4. banded filter bank: weights by ds_read_b128, power row by immediate-offset ds_read_b64; the NEXT
            //         block of four quads is requested before this block's values are waited for (see afx_melfused2.hip)
            float accA, accB;
            {
                constexpr int QA = TA / 4, QB = TB / 4, QT = QA + QB, BLK = CPLX ? 2 : 4, NB = (QT + BLK - 1) / BLK;  // (complex: ten values wait for the second pass)
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
                    if (nextQuads >= 4) LDS_WAIT_N(12);
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
                        if (blk * BLK + i < QA) {
                            sA += lo2(w[cur][i]) * p0[cur][i];
                            sA += hi2(w[cur][i]) * p1[cur][i];
                        } else {
                            sB += lo2(w[cur][i]) * p0[cur][i];
                            sB += hi2(w[cur][i]) * p1[cur][i];
                        }
                    }
                    // the sums of this block before the next block's requests: left free, the scheduler sinks every
                    // multiply-add behind the last request and keeps all the operands alive (212-532 bytes of scratch per lane)
                    PIN(sA);
                    PIN(sB);
                }
                accA = sA.x + sA.y;
                accB = sB.x + sB.y;
            }
            if (GENERAL && !CPLX && !SPLIT && a.postPow) {
                accA = powf(accA, a.normValue);
                accB = powf(accB, a.normValue);
            }
            // ---- 5. store --------------------------------------------------------------
            if constexpr (CC && !SPLIT) {
                // the cepstra of the 16 rows stored BEFORE this one: their stores are a frame old, the block's wait finds them complete
                if (ccN == 16) {
                    ccb_rows<4, AFX_CC_GROUPS, true, AFX_CC_AHEAD>(a.out, a.cc, a.dct, a.num, a.ccNum, a.ccCbrt, f - 16, 16, lane, ccTab);
                    ccN = 0;
                }
            }
            float *orow = ((CPLX && pass) ? a.outIm : a.out) + f * a.num;
            if constexpr (SPLIT) {
                // slot results -> LDS (start of the wave's region: the images there are dead, the power row starts behind), then every
                // row is the sum of its segments in ascending bins (afx_melfused2.hip)
                float *part = reinterpret_cast<float *>(wreg);
                part[lane] = accA;
                part[64 + lane] = accB;
                if (lane0) part[128] = 0.f;
                wave_lds_order();
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const unsigned u = CC ? (unsigned)a.meta[256 + 64 * h + lane] : (h ? seg1 : seg0);
                    float sum = part[u & 255u] + part[(u >> 8) & 255u];
                    sum += part[(u >> 16) & 255u];
                    sum += part[u >> 24];
                    if (GENERAL && !CPLX && a.postPow) sum = powf(sum, a.normValue);
                    if (lane + 64 * h < a.num) orow[lane + 64 * h] = sum;
                }
                wave_lds_order();  // before the next pass / frame writes there
            } else {
                if (rowA >= 0) orow[rowA] = accA;
                if (rowB >= 0) orow[rowB] = accB;
            }
            if constexpr (CC) {
                // split plans: ONE call site, behind the row's stores where the band stage's values are dead (its wait then covers the
                // 16th row's stores); whole-row plans: only the wave's last rows here
                ++ccN;
                if ((SPLIT && ccN == 16) || f + 1 == fEnd) {
                    ccb_rows<2, 1, true, false>(a.out, a.cc, a.dct, a.num, a.ccNum, a.ccCbrt, f + 1 - ccN, ccN, lane, ccTab);
                    ccN = 0;
                }
            }
        }
        }  // !STFT
        // (the band stage's reads have returned -- its last wait is lgkmcnt(0) -- before the next frame's images overwrite the row)

        if (++t == a.timeLength) {
            t = 0;
            ++clip;
        }
    }
}

// host: W_512^(lane d0) | W_64^(c d1) | 0.5 W_1024^k (k <= 256), in double, rounded once
void fill_twiddles(float *tw1, float *tw2, float *tw3) {
    const double PI = 3.14159265358979323846;
    for (int d = 0; d < 8; ++d)
        for (int l = 0; l < 64; ++l) {
            const double ang = -2.0 * PI * (double)(d * l) / MC;
            tw1[2 * (d * 64 + l)] = (float)cos(ang);
            tw1[2 * (d * 64 + l) + 1] = (float)sin(ang);
        }
    for (int d = 0; d < 8; ++d)
        for (int c = 0; c < 8; ++c) {
            const double ang = -2.0 * PI * (double)(d * c) / 64.0;
            tw2[2 * (d * 8 + c)] = (float)cos(ang);
            tw2[2 * (d * 8 + c) + 1] = (float)sin(ang);
        }
    for (int k = 0; k <= 256; ++k) {
        const double ang = -2.0 * PI * (double)k / NFFT;
        tw3[2 * k] = (float)(0.5 * cos(ang));
        tw3[2 * k + 1] = (float)(0.5 * sin(ang));
    }
}

struct Plan {
    int variant;  // >= 100: this file (afxk_melfused_* dispatches on it)
    int num, split;
    float2 *dWin2, *dTw1, *dTw2, *dTw3;
    float *dWLane;
    int *dMeta;
};
struct Variant {
    int tapsA, tapsB;
};
// ordered by cost; the plan's conflict-free lane assignment can stretch the short rows of a
// dense low band (mel-128 at n_fft 1024: 23 / 25 taps), hence the square variant
constexpr Variant kVariants[] = {{24, 8}, {32, 32}, {48, 16}, {72, 32}};
constexpr int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);

template <int TA, int TB, bool GENERAL, int SHIFT, bool CPLX, bool SPLIT = false, bool CC = false>
int launch_variant(const Plan *p, const AfxMelFusedArgs *a, void *stream) {
    const long long total = (long long)a->batch * a->timeLength;
    if (total <= 0) return AFX_OK;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    long long waves = (long long)cus * WAVES * 2;
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
    KArgs k;
    k.x = a->x;
    k.clipStride = a->clipStride;
    k.totalFrames = total;
    k.timeLength = a->timeLength;
    k.hop = a->hop;
    k.framesPerWave = (int)fpw;
    k.aligned = ((a->clipStride & 1) == 0) && ((a->hop & 1) == 0) && ((reinterpret_cast<uintptr_t>(a->x) & 7) == 0);
    k.win2 = p->dWin2;
    k.tw1 = p->dTw1;
    k.tw2 = p->dTw2;
    k.tw3 = p->dTw3;
    k.wLane = p->dWLane;
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
    constexpr size_t lds = (size_t)block_lds_bytes(TA, TB) + (CC ? CCB_BYTES : 0);  // (CC: the DCT operand table behind the wave regions)
    static_assert(lds <= 163840, "workgroup LDS budget");
    static std::atomic<bool> attrSet[AFX_MAX_DEVICES];  // per device: the attribute lives in the device's code object
    const int attrDev = afxdev_current_device() & (AFX_MAX_DEVICES - 1);
    if (!attrSet[attrDev].load(std::memory_order_acquire)) {  // (two threads may both set it: idempotent)
        AFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_stft_band_1k<TA, TB, GENERAL, SHIFT, CPLX, false, false, false, SPLIT, CC>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attrSet[attrDev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((k_stft_band_1k<TA, TB, GENERAL, SHIFT, CPLX, false, false, false, SPLIT, CC>), dim3((unsigned)blocks), dim3(WAVES * 64), lds,
                       (hipStream_t)stream, k);
    AFX_LAUNCH_CHECK("k_stft_band_1k");
    return AFX_OK;
}

template <int TA, int TB>
int launch(const Plan *p, const AfxMelFusedArgs *a, void *stream) {
    const bool general = (a->specMap != 0) || a->postPow;
    const bool shift2 = (a->hop == 256);  // hop = 128 * SHIFT
    if (a->cc) {  // cepstra in the same launch: real results, plain power rows on whole-row plans, every real mode on split plans
        if (a->specMap >= 3 || a->ccNum < 1 || a->ccNum > 16 || !a->dct || !a->out || p->num > 128 || (p->num & 3) ||
            (a->ccRectify != 0 && a->ccRectify != 1) || a->energy)
            return AFX_ERR_UNSUPPORTED;
        if (p->split)
            return shift2 ? launch_variant<TA, TB, true, 2, false, true, true>(p, a, stream) : launch_variant<TA, TB, true, 0, false, true, true>(p, a, stream);
        if (general) return AFX_ERR_UNSUPPORTED;
        return shift2 ? launch_variant<TA, TB, false, 2, false, false, true>(p, a, stream) : launch_variant<TA, TB, false, 0, false, false, true>(p, a, stream);
    }
    if (p->split) {  // segment plans: the general instantiations only (they take the plain modes too)
        if (a->specMap >= 3) {
            if (!a->outIm) return AFX_ERR_ARG;
            return shift2 ? launch_variant<TA, TB, true, 2, true, true>(p, a, stream) : launch_variant<TA, TB, true, 0, true, true>(p, a, stream);
        }
        return shift2 ? launch_variant<TA, TB, true, 2, false, true>(p, a, stream) : launch_variant<TA, TB, true, 0, false, true>(p, a, stream);
    }
    if (a->specMap >= 3) {
        if (!a->outIm) return AFX_ERR_ARG;
        return shift2 ? launch_variant<TA, TB, true, 2, true>(p, a, stream)
                      : launch_variant<TA, TB, true, 0, true>(p, a, stream);
    }
    if (general)
        return shift2 ? launch_variant<TA, TB, true, 2, false>(p, a, stream)
                      : launch_variant<TA, TB, true, 0, false>(p, a, stream);
    return shift2 ? launch_variant<TA, TB, false, 2, false>(p, a, stream)
                  : launch_variant<TA, TB, false, 0, false>(p, a, stream);
}

template <typename T>
int upload(T **dptr, const void *src, size_t bytes, void *stream) {
    int st = afxdev_malloc(reinterpret_cast<void **>(dptr), bytes);
    if (st != AFX_OK) return st;
    return afxdev_h2d(*dptr, src, bytes, stream);
}

}  // namespace

extern "C" int afxk_mel1k_variant(int tapsA, int tapsB) {
    for (int i = 0; i < kNumVariants; ++i)
        if (tapsA <= kVariants[i].tapsA && tapsB <= kVariants[i].tapsB) return 100 + i;
    return -1;
}

extern "C" int afxk_mel1k_kind(const void *plan) {
    const Plan *p = static_cast<const Plan *>(plan);
    return !p ? 0 : (p->split ? 102 : 101);
}

extern "C" void afxk_mel1k_destroy(void *plan) {
    Plan *p = static_cast<Plan *>(plan);
    if (!p) return;
    afxdev_free(p->dWin2);
    afxdev_free(p->dTw1);
    afxdev_free(p->dTw2);
    afxdev_free(p->dTw3);
    afxdev_free(p->dWLane);
    afxdev_free(p->dMeta);
    free(p);
}

extern "C" int afxk_mel1k_create(void **plan, const float *hWindow, const AfxBandPlan *band, void *stream) {
    *plan = nullptr;
    const int variant = afxk_mel1k_variant(band->tapsA, band->tapsB);
    if (variant < 0) return AFX_ERR_UNSUPPORTED;
    const int TA = kVariants[variant - 100].tapsA, TB = kVariants[variant - 100].tapsB;
    Plan *p = static_cast<Plan *>(calloc(1, sizeof(Plan)));
    if (!p) return AFX_ERR_NOMEM;
    p->variant = variant;
    p->num = band->num;
    p->split = band->split;
    const int WP = TA + TB + 4;
    float *tw1 = static_cast<float *>(malloc(sizeof(float) * 2 * TAB_TW1_F2));
    float *tw2 = static_cast<float *>(malloc(sizeof(float) * 2 * TAB_TW2_F2));
    float *tw3 = static_cast<float *>(calloc(2 * TAB_TW3_F2, sizeof(float)));
    float *wL = static_cast<float *>(calloc((size_t)64 * WP, sizeof(float)));
    int meta[384];  // startA | startB | rowA | rowB | segIdx[0..63] | segIdx[64..127]
    int st = (tw1 && tw2 && tw3 && wL) ? AFX_OK : AFX_ERR_NOMEM;
    if (st == AFX_OK) {
        fill_twiddles(tw1, tw2, tw3);
        for (int l = 0; l < 64; ++l) {
            for (int t = 0; t < band->tapsA; ++t) wL[(size_t)l * WP + t] = band->wA[(size_t)t * 64 + l];
            for (int t = 0; t < band->tapsB; ++t) wL[(size_t)l * WP + TA + t] = band->wB[(size_t)t * 64 + l];
            meta[l] = band->startA[l];
            meta[64 + l] = band->startB[l];
            meta[128 + l] = band->rowA[l];
            meta[192 + l] = band->rowB[l];
            meta[256 + l] = (int)band->segIdx[l];
            meta[320 + l] = (int)band->segIdx[64 + l];
        }
        st = upload(&p->dWin2, hWindow, sizeof(float) * NFFT, stream);
    }
    if (st == AFX_OK) st = upload(&p->dTw1, tw1, sizeof(float) * 2 * TAB_TW1_F2, stream);
    if (st == AFX_OK) st = upload(&p->dTw2, tw2, sizeof(float) * 2 * TAB_TW2_F2, stream);
    if (st == AFX_OK) st = upload(&p->dTw3, tw3, sizeof(float) * 2 * TAB_TW3_F2, stream);
    if (st == AFX_OK) st = upload(&p->dWLane, wL, sizeof(float) * (size_t)64 * WP, stream);
    if (st == AFX_OK) st = upload(&p->dMeta, meta, sizeof(meta), stream);
    if (st == AFX_OK) st = afxdev_stream_sync(stream);
    free(tw1);
    free(tw2);
    free(tw3);
    free(wL);
    if (st != AFX_OK) {
        afxk_mel1k_destroy(p);
        return st;
    }
    *plan = p;
    return AFX_OK;
}

extern "C" int afxk_mel1k_run(void *plan, const AfxMelFusedArgs *a, void *stream) {
    if (a->energy) return AFX_ERR_UNSUPPORTED;  // temporal features ride along at n_fft 2048 only (cepstra: every size, launch())
    const Plan *p = static_cast<const Plan *>(plan);
    if (!p) return AFX_ERR_ARG;
    switch (p->variant) {
        case 100: return launch<24, 8>(p, a, stream);
        case 101: return launch<32, 32>(p, a, stream);
        case 102: return launch<48, 16>(p, a, stream);
        case 103: return launch<72, 32>(p, a, stream);
        default: return AFX_ERR_UNSUPPORTED;
    }
}

// ---- n_fft 1024 without a bank (afxk_stft, afx_stft.hip): every frame inside its clip (no padding), no temporal features.
namespace {

// twiddle tables of the STFT instantiations, one device copy per device, never freed: tw1 | tw2 | tw3
const float2 *stft_tables(void *stream) {
    static std::mutex mu;
    static float2 *dTab[AFX_MAX_DEVICES] = {};
    const int dev = afxdev_current_device();
    if (dev < 0 || dev >= AFX_MAX_DEVICES) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    if (!dTab[dev]) {
        constexpr int NF2 = TAB_TW1_F2 + TAB_TW2_F2 + TAB_TW3_F2;
        float *h = static_cast<float *>(calloc(2 * NF2, sizeof(float)));
        if (!h) return nullptr;
        fill_twiddles(h, h + 2 * TAB_TW1_F2, h + 2 * (TAB_TW1_F2 + TAB_TW2_F2));
        float2 *d = nullptr;
        int st = afxdev_malloc(reinterpret_cast<void **>(&d), sizeof(float) * 2 * NF2);
        // (a synchronous copy, like wave_tables() of afx_stft.hip: the caller's stream is not waited for under this lock)
        if (st == AFX_OK && hipMemcpy(d, h, sizeof(float) * 2 * NF2, hipMemcpyHostToDevice) != hipSuccess) st = AFX_ERR_HIP;
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
int launch_stft(const AfxStftArgs *a, const float2 *tab, void *stream) {
    const long long total = (long long)a->batch * a->timeLength;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    long long waves = (long long)cus * WAVES * 2;
    long long fpw = (total + waves - 1) / waves;
    if (fpw < 16) {
        const long long oneRound = (total + (long long)cus * WAVES - 1) / ((long long)cus * WAVES);
        fpw = oneRound < 16 ? oneRound : 16;
    }
    const long long usedWaves = (total + fpw - 1) / fpw;
    const long long blocks = (usedWaves + WAVES - 1) / WAVES;
    KArgs k;
    memset(&k, 0, sizeof(k));
    k.x = a->x;
    k.clipStride = a->clipStride;
    k.totalFrames = total;
    k.timeLength = a->timeLength;
    k.hop = a->hop;
    k.framesPerWave = (int)fpw;
    k.aligned = ((a->clipStride & 1) == 0) && ((a->hop & 1) == 0) && ((reinterpret_cast<uintptr_t>(a->x) & 7) == 0);
    k.win2 = reinterpret_cast<const float2 *>(a->window);  // (w[2n], w[2n+1]) at [n]: the object's window as it lies
    k.tw1 = tab;
    k.tw2 = tab + TAB_TW1_F2;
    k.tw3 = tab + TAB_TW1_F2 + TAB_TW2_F2;
    k.specMap = 3;
    k.normValue = a->normValue;
    k.out = a->outRe;
    k.outIm = a->outIm;
    k.mode = a->mode;
    k.binLo = a->binLo;
    k.binCount = a->binCount;
    k.outPitch = a->outPitch ? a->outPitch : (long long)a->binCount;
    constexpr size_t lds = (size_t)block_lds_bytes(0, 0);
    static std::atomic<bool> attrSet[AFX_MAX_DEVICES];
    const int attrDev = afxdev_current_device() & (AFX_MAX_DEVICES - 1);
    if (!attrSet[attrDev].load(std::memory_order_acquire)) {  // (two threads may both set it: idempotent)
        AFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_stft_band_1k<0, 0, false, SHIFT, true, true, MAPPED, FULL>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attrSet[attrDev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((k_stft_band_1k<0, 0, false, SHIFT, true, true, MAPPED, FULL>), dim3((unsigned)blocks), dim3(WAVES * 64), lds,
                       (hipStream_t)stream, k);
    AFX_LAUNCH_CHECK("k_stft_band_1k<stft>");
    return AFX_OK;
}

}  // namespace

// AFX_ERR_UNSUPPORTED: the caller runs the size-generic kernel
extern "C" int afxk_stft1k(const AfxStftArgs *a, void *stream) {
    if (a->radix2Exp != 10 || a->bandStart || a->energy || a->binLo < 0 || a->binCount < 1 || a->binLo + a->binCount > NFFT ||
        a->padLeft != 0 || a->hop < 1 || (long long)(a->timeLength - 1) * a->hop + NFFT > a->dataLength ||
        (reinterpret_cast<uintptr_t>(a->window) & 7) != 0)
        return AFX_ERR_UNSUPPORTED;
    const bool two = (a->mode == AFX_SPEC_COMPLEX || a->mode == AFX_SPEC_SQUARE);
    if (!a->outRe || (two && !a->outIm)) return AFX_ERR_ARG;
    if ((long long)a->batch * a->timeLength <= 0) return AFX_OK;
    const float2 *tab = stft_tables(stream);
    if (!tab) return AFX_ERR_UNSUPPORTED;
    const bool s2 = a->hop == 256;  // register re-use of the overlapping frames
    if (a->mode == AFX_SPEC_COMPLEX) {
        if (a->binLo == 0 && a->binCount == NFFT) return s2 ? launch_stft<2, false, true>(a, tab, stream) : launch_stft<0, false, true>(a, tab, stream);
        return s2 ? launch_stft<2, false, false>(a, tab, stream) : launch_stft<0, false, false>(a, tab, stream);
    }
    return s2 ? launch_stft<2, true, false>(a, tab, stream) : launch_stft<0, true, false>(a, tab, stream);
}
