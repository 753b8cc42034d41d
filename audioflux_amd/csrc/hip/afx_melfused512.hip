// afx_melfused512.hip -- the fused STFT -> spectrum value -> banded filter bank kernel for n_fft = 512 (radix2Exp 9: 32 ms
// of 16 kHz speech): same design as afx_melfused1k.hip (one wave per frame, tables in LDS, register re-use of the overlapping
// frames at hop 128, lane-owned bank rows, every LDS access issued by hand -- afx_asm.h), with the 256-point complex FFT of the
// packed real frame as 4 x 4 x 4 x 4 in FOUR registers per lane: radix-4 in registers, three transposes through LDS (rows of
// 5 float2: every store and read free of bank conflicts), the last one lands in natural order -- lane L holds Z[L + 64 q2] --
// so only the mirror partners of the real-input split are read back.  Index algebra and conflict check: tools/proto_fft256.py.
//
// Replaces, per frame, the same reference code as afx_melfused.hip (stft_algorithm.c:696-803, fft_algorithm.c:450-519,
// flux_complex.c:254-286,469-503, bft_algorithm.c:457-529, flux_vector.c:55-86).
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

constexpr int NFFT = 512;
constexpr int MC = 256;            // complex FFT length
constexpr int RP = 5;              // float2 per row of the three transposes (40 bytes: conflict-free ds_write_b64 / ds_read_b64)
constexpr int EX_F2 = 64 * RP;     // 320 float2; also holds the 256-float2 natural image
constexpr int PROW_OFF = 1536;     // byte offset of the power row in a wave's region: bins 0..256 alias the images' tail,
constexpr int PROW_F = 384;        // the zero pad of the fixed-length band loops (bins 257..383) lies behind them
constexpr int WAVE_LDS_BYTES = PROW_OFF + PROW_F * 4;  // 3072
static_assert(EX_F2 * 8 <= PROW_OFF + 257 * 4, "the images must end before the zero pad");
constexpr int WAVES = 16;          // 4 waves per SIMD
// tables, byte offsets in LDS
constexpr int T_WIN = 0;           // [4][64] float2: (w[2n], w[2n+1]), n = 64 r + lane
constexpr int T_TW1 = 2048;        // [4][64] float2: W_256^(lane d0)
constexpr int T_TW2 = 4096;        // [16] rows of 5 float2: W_64^((4 b + c) q0), row 4 b + c
constexpr int T_TW3 = 4736;        // [4] rows of 5 float2: W_16^(c q1)
constexpr int T_TWS = 4896;        // [129] float2: 0.5 W_512^k
constexpr int TAB_BYTES = 5936;    // (16-byte multiple: the band weights behind are read as float4)
static_assert(T_TW2 + 16 * RP * 8 == T_TW3 && T_TW3 + 4 * RP * 8 == T_TWS && T_TWS + 129 * 8 <= TAB_BYTES && TAB_BYTES % 16 == 0, "table layout");
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
    const float *tab;      // table blob: TAB_BYTES, then [64][WP] band weights
    const int *meta;
    int specMap, postPow;
    float normValue;
    float *out, *outIm;
    int num;
    // STFT instantiations (afxk_stft512): bins instead of bank rows
    const float *window;   // device [512], natural order = (w[2n], w[2n+1]) at [n]
    int mode;              // AFX_SPEC_*
    int binLo, binCount;   // bins binLo .. binLo + binCount - 1 are stored; above 256: conjugate mirrors
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

// |X|^2 of the conjugate pair (k, 256-k) from A = Z[k], B = Z[256-k], w = 0.5 W_512^k
__device__ __forceinline__ void split_pair(v2 A, v2 B, v2 w, float &pk, float &pq) {
    const v2 e2 = pk_add_conj(A, B);
    const v2 d = pk_sub_conj(A, B);
    const v2 wo = cmul_mi(d, w);
    const v2 x = e2 * 0.5f + wo;  // X[k]
    const v2 y = e2 * 0.5f - wo;  // conj(X[256-k])
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

// GENERAL: magnitude / norm exponent / post power (real results); SHIFT: hop = 128 SHIFT samples = SHIFT registers;
// CPLX: complex results (specMap 3: S, 4: S^2), the bank runs over the real and the imaginary parts in turn
// STFT: no bank -- the spectrum values themselves (CPLX form) go to memory through stft_map (afxk_stft512; afx_melfused4k2.hip
//   has the same at n_fft 4096); MAPPED: any AFX_SPEC_* map; FULL: all 512 bins are stored (no range checks)
// SPLIT: the plan's slots hold row SEGMENTS (afx_bandplan_build_split): banks whose rows are longer than the tap variants
// CC: cepstra of the rows in the same launch (real results; afx_ccblock.h: every 16 frames the wave re-reads its rows from L2)
template <int TA, int TB, bool GENERAL, int SHIFT, bool CPLX, bool STFT = false, bool MAPPED = false, bool FULL = false, bool SPLIT = false, bool CC = false>
__global__ __launch_bounds__(WAVES * 64) void k_stft_band_512(KArgs a) {
    static_assert(!CC || (!CPLX && !STFT), "cepstra: real bank rows");
    static_assert(!STFT || (CPLX && TA == 0 && TB == 0 && !SPLIT), "STFT instantiations: complex values, no bank");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    constexpr int WP = wpitch(TA, TB);
    constexpr int TABB = TAB_BYTES + 64 * WP * 4;
    unsigned char *wreg = smem + TABB + wave * WAVE_LDS_BYTES;
    float *prow = reinterpret_cast<float *>(wreg + PROW_OFF);
    {
        float4 *s4 = reinterpret_cast<float4 *>(smem);
        const float4 *g4 = reinterpret_cast<const float4 *>(a.tab);
        for (int i = threadIdx.x + (STFT ? T_TW1 / 16 : 0); i < (STFT ? TAB_BYTES : TABB) / 16; i += WAVES * 64) s4[i] = g4[i];
        if constexpr (STFT)  // the object's own window; the blob holds the twiddles only
            for (int i = threadIdx.x; i < NFFT; i += WAVES * 64) reinterpret_cast<float *>(smem + T_WIN)[i] = a.window[i];
        for (int i = 257 + lane; i < PROW_F; i += 64) prow[i] = 0.f;  // zero pad, written once
        if constexpr (CC)  // the DCT operand of the cepstrum block, behind the wave regions (afx_ccblock.h)
            ccb_table_fill(reinterpret_cast<float *>(smem + block_lds_bytes(TA, TB)), a.dct, a.num, a.ccNum, threadIdx.x, WAVES * 64);
    }
    __syncthreads();

    // ---- per-lane constants: loop-invariant LDS byte addresses (tools/proto_fft256.py) ----
    const bool lane0 = (lane == 0);
    const int hi4 = lane >> 4, mid = (lane >> 2) & 3, low = lane & 3;
    const unsigned T0 = lds_addr(smem), W0 = lds_addr(wreg);
    const unsigned aWin = T0 + T_WIN + 8 * lane;                       // window row r: + 512 r;  W_256^(lane d0): + T_TW1 + 512 d0
    const unsigned aTw2 = T0 + T_TW2 + 8 * RP * (lane & 15);           // stage 2 (lane 16 d0 + 4 b + c): row 4 b + c, q0: + 8 q0
    const unsigned aTw3 = T0 + T_TW3 + 8 * RP * mid;                   // stage 3 (lane 16 d0 + q0 + 4 c): row c, q1: + 8 q1
    const unsigned aTs = T0 + T_TWS + 8 * lane;                        // 0.5 W_512^(lane + 64 j): + 512 j
    const unsigned aTsm = T0 + T_TWS + 8 * 128;
    const unsigned aE1w = W0 + 8 * ((lane & 15) * RP + hi4);           // lane (a, b, c) -> row 16 d0 + 4 b + c, column a: + 640 d0
    const unsigned aE2w = W0 + 8 * ((16 * hi4 + 4 * low) * RP + mid);  // lane (d0, b, c) -> row 16 d0 + q0 + 4 c, column b: + 40 q0
    const unsigned aE3w = W0 + 8 * ((hi4 + 4 * low) * RP + mid);       // lane (d0, c, q0) -> row d0 + 4 q0 + 16 q1, column c: + 640 q1
    const unsigned aEr = W0 + 8 * RP * lane;                           // a lane's row: 4 x 8 bytes
    const unsigned aN = W0 + 8 * lane;                                 // natural image Z[lane + 64 q2]: + 512 q2
    const unsigned aNm0 = W0 + 8 * ((256 - lane) & 255);               // Z[256 - lane] (lane 0: Z[0])
    const unsigned aNm1 = W0 + 8 * (192 - lane);                       // Z[256 - (lane + 64)]
    const unsigned aMid = W0 + 8 * 128;
    const unsigned R = W0 + PROW_OFF;
    const unsigned aP = R + 4 * lane;                                  // bins lane, lane + 64
    const unsigned aQ = R + 4 * (192 - lane);                          // bins 192 - lane, 256 - lane

    const int startA = STFT ? 0 : a.meta[lane], startB = STFT ? 0 : a.meta[64 + lane];
    const int rowA = STFT ? -1 : a.meta[128 + lane], rowB = STFT ? -1 : a.meta[192 + lane];
    const unsigned seg0 = SPLIT ? (unsigned)a.meta[256 + lane] : 0u, seg1 = SPLIT ? (unsigned)a.meta[320 + lane] : 0u;
    const unsigned apa = R + 4 * startA, apb = R + 4 * startB;
    const unsigned awr = T0 + TAB_BYTES + 4 * WP * lane;

    const long long gw = (long long)blockIdx.x * WAVES + wave;
    long long f = uniform64(gw * a.framesPerWave);  // (scalar registers: the frame counters are compared and advanced on the scalar unit)
    long long fEnd = f + a.framesPerWave;
    if (fEnd > a.totalFrames) fEnd = a.totalFrames;
    if (f >= fEnd) return;
    int clip = (int)(f / a.timeLength);
    int t = (int)(f - (long long)clip * a.timeLength);
    int ccN = 0;  // CC: rows of this wave whose cepstra are still to be formed
    const float *const ccTab = reinterpret_cast<const float *>(smem + block_lds_bytes(TA, TB));

    // raw[r] = (x[2n], x[2n+1]), n = 64 r + lane
    v2 raw[4];
    auto fetch = [&](const float *px, int first) {
        if (a.aligned) {
            const v2 *p2 = reinterpret_cast<const v2 *>(px);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r >= first) raw[r] = p2[64 * r + lane];
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r >= first) {
                    const int n = 64 * r + lane;
                    raw[r] = v2{px[2 * n], px[2 * n + 1]};
                }
        }
    };
    fetch(a.x + (long long)clip * a.clipStride + (long long)t * a.hop, 0);

    for (; f < fEnd; ++f) {
        v2 v[4];
        // ---- 1. window; start fetching the next frame ----------------------------------------
        {
            v2 wv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) RD64(wv[r], aWin, T_WIN + 512 * r);
            LDS_WAIT_N(0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
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
            if constexpr (SHIFT > 0) {  // hop = 128 SHIFT samples = SHIFT registers
                if (tn != 0) {
#pragma unroll
                    for (int r = 0; r + SHIFT < 4; ++r) raw[r] = raw[r + SHIFT];
                    fetch(pn, 4 - SHIFT);
                    whole = false;
                }
            }
            if (whole) fetch(pn, 0);
        }
        // ---- 2. 256-point complex FFT, 4 x 4 x 4 x 4 ------------------------------------------
        {   // stage 1: over r, twiddle W_256^(lane d0)
            v2 tw[4];
#pragma unroll
            for (int d = 1; d < 4; ++d) RD64(tw[d], aWin, T_TW1 + 512 * d);
            dft4(v[0], v[1], v[2], v[3]);
#pragma unroll
            for (int i = 0; i < 4; ++i) PIN(v[i]);
            LDS_WAIT_N(0);
            v2 o[4];
            o[0] = v[0];
#pragma unroll
            for (int d = 1; d < 4; ++d) {
                PIN(tw[d]);
                o[d] = cmul(v[d], tw[d]);
            }
            WR2_64(aE1w, o[0], o[1], 0, 16 * RP);
            WR2_64(aE1w, o[2], o[3], 32 * RP, 48 * RP);
        }
        wave_lds_order();
        {   // stage 2: over a, twiddle W_64^((4 b + c) q0)
            v2 tw[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) RD64(v[i], aEr, 8 * i);
#pragma unroll
            for (int q = 1; q < 4; ++q) RD64(tw[q], aTw2, 8 * q);
            LDS_WAIT_N(3);
#pragma unroll
            for (int i = 0; i < 4; ++i) PIN(v[i]);
            dft4(v[0], v[1], v[2], v[3]);
#pragma unroll
            for (int i = 0; i < 4; ++i) PIN(v[i]);
            LDS_WAIT_N(0);
            v2 o[4];
            o[0] = v[0];
#pragma unroll
            for (int q = 1; q < 4; ++q) {
                PIN(tw[q]);
                o[q] = cmul(v[q], tw[q]);
            }
            WR2_64(aE2w, o[0], o[1], 0, RP);
            WR2_64(aE2w, o[2], o[3], 2 * RP, 3 * RP);
        }
        wave_lds_order();
        {   // stage 3: over b, twiddle W_16^(c q1)
            v2 tw[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) RD64(v[i], aEr, 8 * i);
#pragma unroll
            for (int q = 1; q < 4; ++q) RD64(tw[q], aTw3, 8 * q);
            LDS_WAIT_N(3);
#pragma unroll
            for (int i = 0; i < 4; ++i) PIN(v[i]);
            dft4(v[0], v[1], v[2], v[3]);
#pragma unroll
            for (int i = 0; i < 4; ++i) PIN(v[i]);
            LDS_WAIT_N(0);
            v2 o[4];
            o[0] = v[0];
#pragma unroll
            for (int q = 1; q < 4; ++q) {
                PIN(tw[q]);
                o[q] = cmul(v[q], tw[q]);
            }
            WR2_64(aE3w, o[0], o[1], 0, 16 * RP);
            WR2_64(aE3w, o[2], o[3], 32 * RP, 48 * RP);
        }
        wave_lds_order();
#pragma unroll
        for (int i = 0; i < 4; ++i) RD64(v[i], aEr, 8 * i);
        LDS_WAIT_N(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) PIN(v[i]);
        dft4(v[0], v[1], v[2], v[3]);  // stage 4: v[q2] = Z[lane + 64 q2]
        // ---- 3. natural-order image for the mirror partners; conjugate pairs (k, 256-k), k = lane + 64 j --------
        WR2_64(aN, v[0], v[1], 0, 64);
        WR2_64(aN, v[2], v[3], 128, 192);
        wave_lds_order();
        float pk[3], pq[3];
        float pkI[CPLX ? 3 : 1], pqI[CPLX ? 3 : 1];
        {
            v2 zb[2], ws[2], zm, wm;
            RD64(zb[0], aNm0, 0);
            RD64(zb[1], aNm1, 0);
            RD64(ws[0], aTs, 0);
            RD64(ws[1], aTs, 512);
            RD64(zm, aMid, 0);  // bin 128 pairs with itself
            RD64(wm, aTsm, 0);
            LDS_WAIT_N(0);
            PIN(zb[0]); PIN(zb[1]); PIN(ws[0]); PIN(ws[1]); PIN(zm); PIN(wm);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (CPLX) split_pair_c(v[j], zb[j], ws[j], !STFT && a.specMap == 4, pk[j], pkI[CPLX ? j : 0], pq[j], pqI[CPLX ? j : 0]);
                else split_pair(v[j], zb[j], ws[j], pk[j], pq[j]);
            }
            if (CPLX) split_pair_c(zm, zm, wm, !STFT && a.specMap == 4, pk[2], pkI[CPLX ? 2 : 0], pq[2], pqI[CPLX ? 2 : 0]);
            else split_pair(zm, zm, wm, pk[2], pq[2]);
        }
        if (CPLX) {
        } else if (GENERAL && a.specMap == 1) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                pk[i] = sqrtf(pk[i]);
                pq[i] = sqrtf(pq[i]);
            }
        } else if (GENERAL && a.specMap == 2) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                pk[i] = powf(pk[i], a.normValue);
                pq[i] = powf(pq[i], a.normValue);
            }
        }
        if constexpr (STFT) {
            // ---- 4'. the spectrum itself: wave-uniform row bases in scalar registers + ONE byte-offset register per family of
            //      bins (afx_melfused4k2.hip).  ore / oim point at bin 0 of the row.
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
            for (int j = 0; j < 2; ++j) {
                const int k = lane + 64 * j;
                const bool kpos = j > 0 || !lane0;  // k > 0: bins 0 and 256 have no mirror
                const float re0 = pk[j], im0 = pkI[CPLX ? j : 0], re1 = pq[j], im1 = pqI[CPLX ? j : 0];
                put(true, k, vUp, 64 * j, re0, im0);                      // X[k]
                put(kpos, NFFT - k, vDn, NFFT - 64 - 64 * j, re0, -im0);  //   mirror 512 - k
                put(true, MC - k, vDn, MC - 64 - 64 * j, re1, im1);       // X[256 - k]
                put(kpos, MC + k, vUp, MC + 64 * j, re1, -im1);           //   mirror 256 + k
            }
            put(lane0, 128, vUp, 128, pk[2], pkI[CPLX ? 2 : 0]);
            put(lane0, 384, vUp, 384, pk[2], -pkI[CPLX ? 2 : 0]);
        } else {
        // every read of the image has returned (lgkmcnt(0) above): the power row may overwrite its tail
#pragma unroll
        for (int pass = 0; pass < (CPLX ? 2 : 1); ++pass) {
            if (CPLX && pass == 1) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    pk[i] = pkI[CPLX ? i : 0];
                    pq[i] = pqI[CPLX ? i : 0];
                }
            }
            WR2ST_32(aP, pk[0], pk[1], 0, 1);   // bins lane, lane + 64
            WR2ST_32(aQ, pq[1], pq[0], 0, 1);   // bins 192 - lane, 256 - lane
            if (lane0) prow[128] = pk[2];
            wave_lds_order();

            // ---- 4. banded filter bank (afx_melfused1k.hip) ----
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
                        if (blk * BLK + i < QA) {
                            sA += lo2(w[cur][i]) * p0[cur][i];
                            sA += hi2(w[cur][i]) * p1[cur][i];
                        } else {
                            sB += lo2(w[cur][i]) * p0[cur][i];
                            sB += hi2(w[cur][i]) * p1[cur][i];
                        }
                    }
                    PIN(sA);  // (this block's sums before the next block's requests: afx_melfused1k.hip)
                    PIN(sB);
                }
                accA = sA.x + sA.y;
                accB = sB.x + sB.y;
            }
            if (GENERAL && !CPLX && !SPLIT && a.postPow) {
                accA = powf(accA, a.normValue);
                accB = powf(accB, a.normValue);
            }
            // ---- 5. store ----
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
                    const unsigned u = h ? seg1 : seg0;
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
        // (the band stage's reads have returned before the next frame's images overwrite the row)

        if (++t == a.timeLength) {
            t = 0;
            ++clip;
        }
    }
}

// host: the transform's tables at their byte offsets (twiddles in double, rounded once); hWindow == nullptr leaves the window part alone
void fill_transform_tables(float *tab, const float *hWindow) {
    const double PI = 3.14159265358979323846;
    auto put = [&](int byteOff, int idx, double ang, double scale) {
        tab[byteOff / 4 + 2 * idx] = (float)(scale * cos(ang));
        tab[byteOff / 4 + 2 * idx + 1] = (float)(scale * sin(ang));
    };
    if (hWindow) memcpy(tab + T_WIN / 4, hWindow, sizeof(float) * NFFT);  // (w[2n], w[2n+1]) at [n], n = 64 r + lane
    for (int d = 0; d < 4; ++d)
        for (int l = 0; l < 64; ++l) put(T_TW1, 64 * d + l, -2.0 * PI * (double)(d * l) / MC, 1.0);
    for (int r = 0; r < 16; ++r)
        for (int q = 0; q < 4; ++q) put(T_TW2, RP * r + q, -2.0 * PI * (double)(r * q) / 64.0, 1.0);
    for (int c = 0; c < 4; ++c)
        for (int q = 0; q < 4; ++q) put(T_TW3, RP * c + q, -2.0 * PI * (double)(c * q) / 16.0, 1.0);
    for (int k = 0; k <= 128; ++k) put(T_TWS, k, -2.0 * PI * (double)k / NFFT, 0.5);
}

struct Plan {
    int variant;  // >= 300: this file (afxk_melfused_* dispatches on it)
    int num, split;
    float *dTab;
    int *dMeta;
};
struct Variant {
    int tapsA, tapsB;
};
// 257 bins: mel-128 needs 12-15 + 3 taps, mel-80 17-24, mel-64 20-28, mel-40 27-45, mel-26 41-61 (8 .. 44.1 kHz)
constexpr Variant kVariants[] = {{16, 4}, {32, 4}, {48, 4}, {64, 8}};
constexpr int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);

template <int TA, int TB, bool GENERAL, int SHIFT, bool CPLX, bool SPLIT = false, bool CC = false>
int launch_variant(const Plan *p, const AfxMelFusedArgs *a, void *stream) {
    const long long total = (long long)a->batch * a->timeLength;
    if (total <= 0) return AFX_OK;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    long long waves = (long long)cus * WAVES * 2;
    long long fpw = (total + waves - 1) / waves;
    if (fpw < 16) {  // (afx_melfused1k.hip: a call that cannot fill one round of workgroups is spread over all CUs)
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
    constexpr size_t lds = (size_t)block_lds_bytes(TA, TB) + (CC ? CCB_BYTES : 0);  // (CC: the DCT operand table behind the wave regions)
    static_assert(lds <= 163840, "workgroup LDS budget");
    static std::atomic<bool> attrSet[AFX_MAX_DEVICES];  // per device: the attribute lives in the device's code object
    const int attrDev = afxdev_current_device() & (AFX_MAX_DEVICES - 1);
    if (!attrSet[attrDev].load(std::memory_order_acquire)) {  // (two threads may both set it: idempotent)
        AFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_stft_band_512<TA, TB, GENERAL, SHIFT, CPLX, false, false, false, SPLIT, CC>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attrSet[attrDev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((k_stft_band_512<TA, TB, GENERAL, SHIFT, CPLX, false, false, false, SPLIT, CC>), dim3((unsigned)blocks), dim3(WAVES * 64), lds,
                       (hipStream_t)stream, k);
    AFX_LAUNCH_CHECK("k_stft_band_512");
    return AFX_OK;
}

template <int TA, int TB>
int launch(const Plan *p, const AfxMelFusedArgs *a, void *stream) {
    const bool general = (a->specMap != 0) || a->postPow;
    const bool shift1 = (a->hop == 128);  // hop = 128 * SHIFT
    if (a->cc) {  // cepstra in the same launch: real results, plain power rows on whole-row plans, every real mode on split plans
        if (a->specMap >= 3 || a->ccNum < 1 || a->ccNum > 16 || !a->dct || !a->out || p->num > 128 || (p->num & 3) ||
            (a->ccRectify != 0 && a->ccRectify != 1) || a->energy)
            return AFX_ERR_UNSUPPORTED;
        if (p->split)
            return shift1 ? launch_variant<TA, TB, true, 1, false, true, true>(p, a, stream) : launch_variant<TA, TB, true, 0, false, true, true>(p, a, stream);
        if (general) return AFX_ERR_UNSUPPORTED;
        return shift1 ? launch_variant<TA, TB, false, 1, false, false, true>(p, a, stream) : launch_variant<TA, TB, false, 0, false, false, true>(p, a, stream);
    }
    if (p->split) {  // segment plans: the general instantiations only (they take the plain modes too)
        if (a->specMap >= 3) {
            if (!a->outIm) return AFX_ERR_ARG;
            return shift1 ? launch_variant<TA, TB, true, 1, true, true>(p, a, stream) : launch_variant<TA, TB, true, 0, true, true>(p, a, stream);
        }
        return shift1 ? launch_variant<TA, TB, true, 1, false, true>(p, a, stream) : launch_variant<TA, TB, true, 0, false, true>(p, a, stream);
    }
    if (a->specMap >= 3) {
        if (!a->outIm) return AFX_ERR_ARG;
        return shift1 ? launch_variant<TA, TB, true, 1, true>(p, a, stream)
                      : launch_variant<TA, TB, true, 0, true>(p, a, stream);
    }
    if (general)
        return shift1 ? launch_variant<TA, TB, true, 1, false>(p, a, stream)
                      : launch_variant<TA, TB, true, 0, false>(p, a, stream);
    return shift1 ? launch_variant<TA, TB, false, 1, false>(p, a, stream)
                  : launch_variant<TA, TB, false, 0, false>(p, a, stream);
}

}  // namespace

extern "C" int afxk_mel512_variant(int tapsA, int tapsB) {
    for (int i = 0; i < kNumVariants; ++i)
        if (tapsA <= kVariants[i].tapsA && tapsB <= kVariants[i].tapsB) return 300 + i;
    return -1;
}

extern "C" int afxk_mel512_kind(const void *plan) {
    const Plan *p = static_cast<const Plan *>(plan);
    return !p ? 0 : (p->split ? 302 : 301);
}

extern "C" void afxk_mel512_destroy(void *plan) {
    Plan *p = static_cast<Plan *>(plan);
    if (!p) return;
    afxdev_free(p->dTab);
    afxdev_free(p->dMeta);
    free(p);
}

extern "C" int afxk_mel512_create(void **plan, const float *hWindow, const AfxBandPlan *band, void *stream) {
    *plan = nullptr;
    const int variant = afxk_mel512_variant(band->tapsA, band->tapsB);
    if (variant < 0) return AFX_ERR_UNSUPPORTED;
    const int TA = kVariants[variant - 300].tapsA, TB = kVariants[variant - 300].tapsB;
    const int WP = wpitch(TA, TB);
    const size_t bytes = (size_t)TAB_BYTES + (size_t)64 * WP * 4;
    Plan *p = static_cast<Plan *>(calloc(1, sizeof(Plan)));
    float *tab = static_cast<float *>(calloc(bytes, 1));
    if (!p || !tab) {
        free(p);
        free(tab);
        return AFX_ERR_NOMEM;
    }
    p->variant = variant;
    p->num = band->num;
    p->split = band->split;
    fill_transform_tables(tab, hWindow);
    float *wL = tab + TAB_BYTES / 4;
    int meta[384];  // startA | startB | rowA | rowB | segIdx[0..63] | segIdx[64..127]
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
    int st = afxdev_malloc(reinterpret_cast<void **>(&p->dTab), bytes);
    if (st == AFX_OK) st = afxdev_h2d(p->dTab, tab, bytes, stream);
    if (st == AFX_OK) st = afxdev_malloc(reinterpret_cast<void **>(&p->dMeta), sizeof(meta));
    if (st == AFX_OK) st = afxdev_h2d(p->dMeta, meta, sizeof(meta), stream);
    if (st == AFX_OK) st = afxdev_stream_sync(stream);  // host staging buffers are freed below
    free(tab);
    if (st != AFX_OK) {
        afxk_mel512_destroy(p);
        return st;
    }
    *plan = p;
    return AFX_OK;
}

extern "C" int afxk_mel512_run(void *plan, const AfxMelFusedArgs *a, void *stream) {
    if (a->energy) return AFX_ERR_UNSUPPORTED;  // temporal features ride along at n_fft 2048 only (cepstra: every size, launch())
    const Plan *p = static_cast<const Plan *>(plan);
    if (!p || a->specMap > 4) return AFX_ERR_ARG;
    switch (p->variant) {
        case 300: return launch<16, 4>(p, a, stream);
        case 301: return launch<32, 4>(p, a, stream);
        case 302: return launch<48, 4>(p, a, stream);
        case 303: return launch<64, 8>(p, a, stream);
        default: return AFX_ERR_UNSUPPORTED;
    }
}

// ---- n_fft 512 without a bank (afxk_stft, afx_stft.hip): every frame inside its clip (no padding), no temporal features.
namespace {

const float *stft_tables(void *stream) {  // one device copy of the twiddle blob per device, never freed
    static std::mutex mu;
    static float *dTab[AFX_MAX_DEVICES] = {};
    const int dev = afxdev_current_device();
    if (dev < 0 || dev >= AFX_MAX_DEVICES) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    if (!dTab[dev]) {
        float *h = static_cast<float *>(calloc(TAB_BYTES, 1));
        if (!h) return nullptr;
        fill_transform_tables(h, nullptr);
        float *d = nullptr;
        int st = afxdev_malloc(reinterpret_cast<void **>(&d), TAB_BYTES);
        // (a synchronous copy, like wave_tables() of afx_stft.hip: the caller's stream is not waited for under this lock)
        if (st == AFX_OK && hipMemcpy(d, h, TAB_BYTES, hipMemcpyHostToDevice) != hipSuccess) st = AFX_ERR_HIP;
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
int launch_stft(const AfxStftArgs *a, const float *tab, void *stream) {
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
        AFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_stft_band_512<0, 0, false, SHIFT, true, true, MAPPED, FULL>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attrSet[attrDev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((k_stft_band_512<0, 0, false, SHIFT, true, true, MAPPED, FULL>), dim3((unsigned)blocks), dim3(WAVES * 64), lds,
                       (hipStream_t)stream, k);
    AFX_LAUNCH_CHECK("k_stft_band_512<stft>");
    return AFX_OK;
}

}  // namespace

// AFX_ERR_UNSUPPORTED: the caller runs the size-generic kernel
extern "C" int afxk_stft512(const AfxStftArgs *a, void *stream) {
    if (a->radix2Exp != 9 || a->bandStart || a->energy || a->binLo < 0 || a->binCount < 1 || a->binLo + a->binCount > NFFT ||
        a->padLeft != 0 || a->hop < 1 || (long long)(a->timeLength - 1) * a->hop + NFFT > a->dataLength)
        return AFX_ERR_UNSUPPORTED;
    const bool two = (a->mode == AFX_SPEC_COMPLEX || a->mode == AFX_SPEC_SQUARE);
    if (!a->outRe || (two && !a->outIm)) return AFX_ERR_ARG;
    if ((long long)a->batch * a->timeLength <= 0) return AFX_OK;
    const float *tab = stft_tables(stream);
    if (!tab) return AFX_ERR_UNSUPPORTED;
    const bool s1 = a->hop == 128;  // register re-use of the overlapping frames
    if (a->mode == AFX_SPEC_COMPLEX) {
        if (a->binLo == 0 && a->binCount == NFFT) return s1 ? launch_stft<1, false, true>(a, tab, stream) : launch_stft<0, false, true>(a, tab, stream);
        return s1 ? launch_stft<1, false, false>(a, tab, stream) : launch_stft<0, false, false>(a, tab, stream);
    }
    return s1 ? launch_stft<1, true, false>(a, tab, stream) : launch_stft<0, true, false>(a, tab, stream);
}
