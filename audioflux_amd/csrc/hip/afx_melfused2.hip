// afx_melfused2.hip -- the headline kernel, round-2 form: framed STFT -> |S|^2 (or |S|,
// |S|^2p) -> banded filter bank (-> log10 -> DCT-II cepstra), one 64-lane wave per 2048-sample
// frame, real results.  Same algorithm and summation orders as k_stft_mel_banded
// (afx_melfused.hip, which keeps the complex-result instantiations); what changed is how the
// wave talks to the LDS and what it no longer computes twice:
//
//   * every constant table is laid out for 16-byte reads: window and W_1024 twiddles as
//     [(n1 >> 1)][lane][n1 & 1] (one ds_read_b128 = two rows), W_64 rows contiguous, the
//     W_2048 split twiddles per lane [s][lane][m]                       (46 -> 22 table reads)
//   * exchange 1: row pitch 72 float2, writer lane 4 m1 + m2 stores at column
//     8 (m1 >> 1) + 2 m2 + (m1 & 1): the reader takes (m1, m1 + 1) with one ds_read_b128;
//     exchange 2: image V[q][m2] (m2 fastest; 16-byte halves swapped when bit 3 of q is set:
//     conflict-free without padding), a lane's four radix-4 inputs are two ds_read_b128;
//     rows are written two at a time with ds_write2_b64            (64 -> 32 exchange ops)
//   * 513 conjugate pairs on 512 slots: lane 0's mirror side reads the self-mirrored base
//     q = 128 instead of q = 0, its slots 2 / 3 become the pairs (128, 896) / (384, 640) and
//     bin 512 is |Z[512]|^2 directly -- no lane runs the "centre" butterflies any more
//     (28 VALU + 6 LDS reads per frame that only lane 0's results were kept of)
//   * the power row goes out with ds_write2st64_b32 (bins k, k + 256 in one instruction), its
//     zero pad sits behind the exchange images and is written once per launch
//   * cepstra in the same launch (CC): every 16 frames the wave re-reads its own 16 mel rows
//     (L2-resident), takes log10 and runs 32 v_mfma_f32_16x16x4_f32 against the DCT rows held
//     in LDS -- the matrix pipe is otherwise idle in this kernel, the second launch and its
//     478 MB re-read of mel from HBM disappear (xxcc_algorithm.c:124-155).  CC = 1: the headline
//     form (num = 128, log10, whole-row plan); CC = 2: afx_ccblock.h's general form (any num <= 128
//     that is a multiple of 4, split plans, cube-root rectification; DCT operand from memory)
//   * temporal features (TEMPORAL): energy / rms / zero-crossing rate of the windowed frame
//     as wave reductions (temporal_algorithm.c:138-144), so isTemporal objects stay on this kernel
//
// Index algebra and LDS bank behaviour of every access class: tools/proto_fft1024_v2.py.
#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>

#include <cmath>
#include <cstdlib>
#include <cstring>

#include "afx_device.h"
#include "afx_hipcheck.h"
#include "afx_pkmath.h"
#include "afx_ccblock.h"

// Knock-out measurement builds (make EXTRA=-DAFX_KO=<mask>; results are WRONG, timing only): bit s drops the LDS traffic of
// site class s -- 0 exchange writes, 1 exchange / image reads, 2 table reads (window, twiddles), 3 band-stage reads, 4 power-row
// writes -- and leaves the arithmetic on whatever the registers hold; bit 5 drops the second radix-16 layer's arithmetic,
// bit 6 the band-stage multiply-adds.  What the step time does NOT lose says what does not bind it (profiles/r05_ab_headline.txt (c)).
#ifdef AFX_KO
#define KO_ON(s) (((AFX_KO) >> (s)) & 1)
#define RD128_S(s, dst, addr, off) do { if (KO_ON(s)) asm volatile("" : "=v"(dst)); else RD128(dst, addr, off); } while (0)
#define RD64_S(s, dst, addr, off) do { if (KO_ON(s)) asm volatile("" : "=v"(dst)); else RD64(dst, addr, off); } while (0)
#define WR2_64_S(s, addr, d0, d1, o0, o1) do { if (KO_ON(s)) asm volatile("" ::"v"(d0), "v"(d1)); else WR2_64(addr, d0, d1, o0, o1); } while (0)
#define WR2ST_32_S(s, addr, d0, d1, o0, o1) do { if (KO_ON(s)) asm volatile("" ::"v"(d0), "v"(d1)); else WR2ST_32(addr, d0, d1, o0, o1); } while (0)
#else
#define KO_ON(s) 0
#define RD128_S(s, dst, addr, off) RD128(dst, addr, off)
#define RD64_S(s, dst, addr, off) RD64(dst, addr, off)
#define WR2_64_S(s, addr, d0, d1, o0, o1) WR2_64(addr, d0, d1, o0, o1)
#define WR2ST_32_S(s, addr, d0, d1, o0, o1) WR2ST_32(addr, d0, d1, o0, o1)
#endif

// measurement switches of round 6 (profiles/r06_ab_headline.txt): AFX_V2_NTIN -- the samples are read with the streaming (nt) policy, so
// that the bank rows a wave re-reads for its cepstra 16 frames later are not pushed out of the L2 by them; AFX_V2_CCEVERY -- frames per
// cepstrum block (16: one full MFMA tile; 8: the rows are half as old when they are re-read)
#ifdef AFX_V2_NTIN
#define AFX_V2_LOAD(p) __builtin_nontemporal_load(p)
#else
#define AFX_V2_LOAD(p) (*(p))
#endif
#ifndef AFX_V2_CCEVERY
#define AFX_V2_CCEVERY 16
#endif

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int NFFT = 2048;
constexpr int MC = 1024;
constexpr int WAVES = 12;                    // one workgroup per CU, 3 waves per SIMD (8: - 5 %, profiles/r02_ab_headline.txt)
constexpr int WAVES_CPLX = 12;               // complex results (8 while the imaginary parts waited in registers for their pass)
__host__ __device__ constexpr int waves_of(bool cplx) { return cplx ? WAVES_CPLX : WAVES; }
constexpr int P1 = 72;                       // float2 per row of the exchange-1 image
constexpr int PROW_OFF = 5120;               // byte offset of the power row in a wave's region
constexpr int PROW_F = 1104;                 // 1025 bins + zero pad for the fixed-length band loops
constexpr int WAVE_LDS = PROW_OFF + PROW_F * 4;  // 9536: pad [9220, 9536) lies behind both images
static_assert(16 * P1 * 8 <= PROW_OFF + 1025 * 4, "exchange image must end before the zero pad");
// table blob, byte offsets (built on the host by afxk_mel2_create, copied to LDS per workgroup)
constexpr int T_WIN = 0;                     // [8][64] float4: (w[2n], w[2n+1]) of rows n1 = 2j, 2j + 1
constexpr int T_TW1 = 8192;                  // [8][64] float4: W_1024^(lane k1), k1 = 2j, 2j + 1
constexpr int T_TW2 = 16384;                 // [4] rows of 16 float2, TW2_PITCH bytes apart: W_64^(m2 j1)
constexpr int TW2_PITCH = 144;               // (128 put rows m2 and m2 + 2 on the same banks: every read of this table two-way
                                             //  conflicted -- the 6.4 % of the kernel's LDS cycles SQ_LDS_BANK_CONFLICT showed)
constexpr int T_TW3 = 16960;                 // [2][64][4] float2: 0.5 W_2048^bin of slot (s, lane, m)
constexpr int T_BAND = 21056;                // [64][WP] floats: lane-major band weights, A then B taps
__host__ __device__ constexpr int wpitch(int ta, int tb) { return ta + tb + 4; }
__host__ __device__ constexpr int tab_bytes(int ta, int tb) { return T_BAND + 64 * wpitch(ta, tb) * 4; }
constexpr int DCT_PITCH = 36;                // floats per lane of the DCT operand table (9 x 16 B: conflict-free b128)
__host__ __device__ constexpr int block_lds_bytes(int ta, int tb, bool cc) {
    return tab_bytes(ta, tb) + (cc ? 64 * DCT_PITCH * 4 : 0) + WAVES * WAVE_LDS;  // (complex results: fewer waves, less)
}

struct KArgs2 {
    const float *x;
    long long clipStride;
    long long totalFrames;
    int timeLength, hop;
    int framesPerWave;
    int aligned;           // frame starts are 8-byte aligned -> float2 loads
    const float4 *tab;     // table blob
    const int *meta;       // [6][64]: startA, startB, rowA, rowB, segIdx lo / hi
    int specMap, postPow;
    float normValue;
    float *out;            // [totalFrames, num]
    float *outIm;          // CPLX: imaginary parts, same shape
    int num;
    // CC
    const float *dct;      // device [num, num] orthonormal DCT-II (row = coefficient)
    int ccNum, ccCbrt;     // (ccCbrt: powf(x, 1/3) instead of log10, CC = 2 only)
    float *cc;             // [totalFrames, ccNum]
    // TEMPORAL
    float *energy, *rms, *zcr;  // [totalFrames]
    // STFT: the mapped spectrum rows themselves (afxk_stft2k): bins binLo .. binLo + binCount - 1 of every frame to out + f * outPitch
    const float *win;      // device window [2048], natural order, 8-byte aligned (the blob's window table is not used)
    int binLo, binCount;
    long long outPitch;
    int vecOut;            // binLo == 0, all 1025 bins, 16-byte aligned rows of >= 1028 floats: 16-byte stores (the pad gets zeros)
};

// (lds_addr, RD64 / RD128, WR2_64, WR2ST_32, PIN, LDS_WAIT_N and the L1-bypassing load: afx_asm.h)

// Orders this wave's LDS stores before its later LDS loads of other lanes' data: DS operations of
// one wave execute in issue order, lgkmcnt(0) drains them, the wave barrier pins the compiler.
// Deliberately NOT a fence (that would also drain vmcnt: the prefetch and the previous stores).
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
}

// Wave priority by phase of the frame (s_setprio): phases 0 window, 1 first radix-16 + twiddles + exchange writes, 2 exchange
// reads, 3 second radix-16, 4 last radix-4 + split, 5 band, 6 store / cepstra; bit p of AFX_PRIO_MASK raises phase p to
// AFX_PRIO_LEVEL.  The three waves of a SIMD are in different phases of their frames; with the transform's phases (1-4,
// dense packed arithmetic between short LDS exchanges) above the window / band / store phases (LDS- and memory-latency
// bound) a wave that can keep the vector unit busy wins the issue port: 1.53-1.56 -> 1.47-1.48 ms per step, measured
// interleaved against masks 0x0A, 0x1F, 0x3E, 0x20, 0x61 and levels 1 / 2 / 3 (profiles/r04_ab_headline.txt).
#ifndef AFX_PRIO_MASK
#define AFX_PRIO_MASK 0x1E
#endif
#ifndef AFX_PRIO_LEVEL
#define AFX_PRIO_LEVEL 1
#endif
#define MEL_PHASE(p)                                                                                       \
    do {                                                                                                   \
        if (AFX_PRIO_MASK != 0) __builtin_amdgcn_s_setprio(((AFX_PRIO_MASK >> (p)) & 1) ? AFX_PRIO_LEVEL : 0); \
    } while (0)

__device__ __forceinline__ v2 lo2(v4f q) { return v2{q.x, q.y}; }
__device__ __forceinline__ v2 hi2(v4f q) { return v2{q.z, q.w}; }

// |X|^2 of the conjugate pair (k, 1024-k) from A = Z[k], B = Z[1024-k], w = 0.5 W_2048^k
__device__ __forceinline__ void split_pair(v2 A, v2 B, v2 w, float &pk, float &pq) {
    const v2 e2 = pk_add_conj(A, B);   // 2 E
    const v2 d = pk_sub_conj(A, B);    // 2 i O
    const v2 wo = cmul_mi(d, w);       // W O
    const v2 x = e2 * 0.5f + wo;       // X[k]
    const v2 y = e2 * 0.5f - wo;       // conj(X[1024-k])
    pk = x.x * x.x + x.y * x.y;
    pq = y.x * y.x + y.y * y.y;
}

// complex results: the spectrum values themselves, x = X[k], y = conj(X[1024-k])
__device__ __forceinline__ void split_pair_c(v2 A, v2 B, v2 w, v2 &x, v2 &y) {
    const v2 e2 = pk_add_conj(A, B);
    const v2 d = pk_sub_conj(A, B);
    const v2 wo = cmul_mi(d, w);
    x = e2 * 0.5f + wo;
    y = e2 * 0.5f - wo;
}
// (re, im) of the requested complex result from a spectrum value c: S (sq = false) or S^2 (bft_algorithm.c:457-485)
__device__ __forceinline__ void cplx_map(v2 c, bool sq, float &re, float &im) {
    re = sq ? c.x * c.x - c.y * c.y : c.x;
    im = sq ? 2.f * (c.x * c.y) : c.y;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// SHIFT: hop = 128 * SHIFT samples -> the next frame's register image is this one moved down by
//   SHIFT registers, only SHIFT new float2 per lane are fetched (0: every frame fetched whole)
// SPLIT: the plan's slots hold row SEGMENTS (afx_bandplan_build_split)
// CC: cepstra of the rows in the same launch, ccNum <= 16 (1: num = 128, log10 rectification, DCT operand in LDS; 2: afx_ccblock.h)
// TEMPORAL: energy / rms / zcr of the windowed frame
// CPLX: complex results (specMap 3: S, 4: S^2): a second row in LDS holds the imaginary parts for a second pass of the bank
// STFT: no bank -- the frame's mapped spectrum row (|S|^2, |S|, |S|^2p) goes to memory as it stands in the LDS (afxk_stft2k: the
//   producer of the dense-bank route's [T, F] rows and the STFT object's real results at n_fft 2048)
template <int TA, int TB, int SHIFT, bool SPLIT, int CC, bool TEMPORAL, bool CPLX = false, bool STFT = false>
__global__ __launch_bounds__(waves_of(CPLX) * 64, 3) void k_stft_mel_v2(KArgs2 a) {
    constexpr int NWV = waves_of(CPLX);
    static_assert(!(CPLX && (CC || TEMPORAL)), "complex results: the bank only");
    static_assert(!STFT || (TA == 0 && TB == 0 && !SPLIT && CC == 0 && !TEMPORAL && !CPLX), "STFT instantiations: real rows, no bank");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    constexpr int WP = wpitch(TA, TB);
    constexpr int TABB = tab_bytes(TA, TB);
    constexpr int DCTB = CC == 1 ? 64 * DCT_PITCH * 4 : 0;
    unsigned char *wreg = smem + TABB + DCTB + wave * WAVE_LDS;
    float *prow = reinterpret_cast<float *>(wreg + PROW_OFF);

    // ---- workgroup-shared tables -> LDS (once) -------------------------------------------
    {
        float4 *s4 = reinterpret_cast<float4 *>(smem);
        for (int i = threadIdx.x + (STFT ? T_TW1 / 16 : 0); i < TABB / 16; i += NWV * 64) s4[i] = a.tab[i];
        if constexpr (STFT) {
            // the caller's window in the pair layout of afxk_mel2_create: entry (n1, lane) = (w[2n], w[2n+1]), n = 64 n1 + lane,
            // at float2 index 128 (n1 >> 1) + 2 lane + (n1 & 1)
            v2 *tw = reinterpret_cast<v2 *>(smem + T_WIN);
            const v2 *w2 = reinterpret_cast<const v2 *>(a.win);
            for (int at = threadIdx.x; at < 1024; at += NWV * 64) {
                const int n1 = 2 * (at >> 7) + (at & 1), l = (at & 127) >> 1;
                tw[at] = w2[64 * n1 + l];
            }
        }
        if constexpr (CC == 1) {
            // B operand of the cepstrum MFMAs: lane (coefficient fi = lane & 15, k-slot g = lane >> 4)
            // holds dct[fi][16 u + 4 g + c] at [lane][4 u + c]
            float *tabD = reinterpret_cast<float *>(smem + TABB);
            for (int i = threadIdx.x; i < 64 * 32; i += NWV * 64) {
                const int l = i >> 5, e = i & 31;
                const int fi = l & 15, g = l >> 4;
                tabD[l * DCT_PITCH + e] =
                    fi < a.ccNum ? a.dct[(long long)fi * a.num + 16 * (e >> 2) + 4 * g + (e & 3)] : 0.f;
            }
        }
        for (int i = 1025 + lane; i < PROW_F; i += 64) prow[i] = 0.f;  // zero pad, never overwritten
    }
    __syncthreads();

    // ---- per-lane constants (loop-invariant LDS byte addresses) -----------------------------
    const int k1 = lane >> 2, m2 = lane & 3;
    const unsigned T0 = lds_addr(smem), W0 = lds_addr(wreg);
    const unsigned aWin = T0 + T_WIN + 16 * lane;                      // + 1024 j; W_1024 at + T_TW1
    const unsigned aTw2 = T0 + T_TW2 + TW2_PITCH * m2;                       // + 16 j
    const unsigned aE1w = W0 + 8 * (8 * (k1 >> 1) + 2 * m2 + (k1 & 1));  // writer m1 = lane >> 2; row k: + 576 k
    const unsigned aE1r = W0 + 576 * k1 + 16 * m2;                     // pair jj: + 64 jj
    const unsigned aE2w = W0 + 32 * k1 + 16 * ((m2 >> 1) ^ ((k1 >> 3) & 1)) + 8 * (m2 & 1);  // j1: + 512 j1
    const int b3l = (lane >> 3) & 1;
    const unsigned aAlo = W0 + 32 * lane + 16 * b3l, aAhi = W0 + 32 * lane + 16 * (1 - b3l);  // s = 1: + 2048
    const int qm0 = lane == 0 ? 128 : 256 - lane, qm1 = 192 - lane;
    const unsigned aB0lo = W0 + 32 * qm0 + 16 * ((qm0 >> 3) & 1), aB0hi = W0 + 32 * qm0 + 16 * (1 - ((qm0 >> 3) & 1));
    const unsigned aB1lo = W0 + 32 * qm1 + 16 * ((qm1 >> 3) & 1), aB1hi = W0 + 32 * qm1 + 16 * (1 - ((qm1 >> 3) & 1));
    const unsigned aT3lo = T0 + T_TW3 + 32 * lane + 16 * b3l, aT3hi = T0 + T_TW3 + 32 * lane + 16 * (1 - b3l);
    const unsigned R = W0 + PROW_OFF;
    const unsigned aP01 = R + 4 * lane;                                // bins k, k + 256 | 64 + k ... by offsets
    const unsigned aP23 = R + 4 * (lane == 0 ? 128 : lane + 512);      // (k + 512, k + 768) | lane 0: (128, 384)
    const unsigned aQs1 = R + 4 * (192 - lane);                        // 1024 - bin of s = 1 slots; s = 0 slots 1, 0 at + 9, + 13
    const unsigned aQ23 = R + 4 * (lane == 0 ? 640 : 256 - lane);      // s = 0 slots 3, 2 | lane 0: (640, 896)
    const bool lane0 = (lane == 0);

    const int startA = STFT ? 0 : a.meta[lane], startB = STFT ? 0 : a.meta[64 + lane];
    const int rowA = STFT ? -1 : a.meta[128 + lane], rowB = STFT ? -1 : a.meta[192 + lane];
    const unsigned seg0 = SPLIT ? (unsigned)a.meta[256 + lane] : 0u, seg1 = SPLIT ? (unsigned)a.meta[320 + lane] : 0u;
    const unsigned apa = R + 4 * startA, apb = R + 4 * startB;
    const unsigned awr = T0 + T_BAND + 4 * WP * lane;

    const long long gw = (long long)blockIdx.x * NWV + wave;
    long long f = gw * a.framesPerWave;
    long long fEnd = f + a.framesPerWave;
    if (fEnd > a.totalFrames) fEnd = a.totalFrames;
    if (f >= fEnd) return;
    int clip = (int)(f / a.timeLength);
    int t = (int)(f - (long long)clip * a.timeLength);
    int ccN = 0;  // frames of this wave whose cepstra are still to be formed

    // raw samples of the frame about to be transformed: raw[n1] = (x[2n], x[2n+1]), n = 64 n1 + lane
    v2 raw[16];
    auto fetch = [&](const float *px, int first) {
        if (a.aligned) {
            const v2 *p2 = reinterpret_cast<const v2 *>(px);
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1)
                if (n1 >= first) raw[n1] = AFX_V2_LOAD(&p2[64 * n1 + lane]);
        } else {
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1)
                if (n1 >= first) {
                    const int n = 64 * n1 + lane;
                    raw[n1] = v2{px[2 * n], px[2 * n + 1]};
                }
        }
    };
    fetch(a.x + (long long)clip * a.clipStride + (long long)t * a.hop, 0);

    // ---- cepstra of `cnt` (<= 16) consecutive rows fb.. of this wave: C[16 frames, 16 coefficients] =
    //      log10(max(rows, 1e-8)) . D^T with v_mfma_f32_16x16x4_f32.  Lane (fi = lane & 15, g = lane >> 4)
    //      loads float4 row[fb + fi][16 u + 4 g ..] (k-slot g of MFMA (u, c) stands for band 16 u + 4 g + c),
    //      the matching DCT elements come from the LDS table.  Called one frame AFTER the 16th row was
    //      stored, so the s_waitcnt finds those stores long complete; reads bypass the CU's L1.
    auto cc_block = [&](long long fb, int cnt) {
        if constexpr (CC == 2) {  // the general form: runtime num / rectification, DCT operand from memory
            ccb_rows<SPLIT ? 2 : 4>(a.out, a.cc, a.dct, a.num, a.ccNum, a.ccCbrt, fb, cnt, lane);
        } else {
        VM_WAIT_ALL();  // own stores -> L2 (vmcnt counts stores on gfx9)
        int ln = lane;
        PIN(ln);  // keep this block's per-lane values out of the frame loop's registers
        const int fi = ln & 15, g = ln >> 4;
        const long long r = fb + (fi < cnt ? fi : cnt - 1);  // tail: duplicate the last row, not stored
        const v4f *src = reinterpret_cast<const v4f *>(a.out + r * 128) + g;
        const unsigned ad = T0 + TABB + 4 * DCT_PITCH * ln;
        v4f acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        constexpr float LOG10_2 = 0.30102999566398120f;
        // two halves of the 128 bands: 32 + 16 live registers instead of 64 + 16 (the frame loop
        // keeps the next frame's 32 prefetch registers alive across this block)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            v4f av[4], dv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) LOAD_SC1_B128(av[u], src + 4 * (4 * h + u));  // served by the L2, never by this CU's L1
#pragma unroll
            for (int u = 0; u < 4; ++u) RD128(dv[u], ad, 16 * (4 * h + u));
            VM_LGKM_WAIT_ALL();
#pragma unroll
            for (int u = 0; u < 4; ++u) PIN(av[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) PIN(dv[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    // log10f(max(x, 1e-8)) (xxcc_algorithm.c:131-137) as v_log_f32 * log10(2); four
                    // independent accumulator chains (a dependent f32 MFMA waits 40 cycles)
                    const float lg = __log2f(fmaxf(av[u][c], 1e-8f)) * LOG10_2;
                    acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(lg, dv[u][c], acc[c], 0, 0, 0);
                }
            }
        }
        const v4f sum = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        // C layout: column (coefficient) = lane & 15, row (frame) = 4 (lane >> 4) + reg
        if (fi < a.ccNum) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int rr = 4 * g + reg;
                if (rr < cnt) a.cc[(fb + rr) * a.ccNum + fi] = sum[reg];
            }
        }
        }  // CC == 1
        ccN -= cnt;
    };

    for (; f < fEnd; ++f) {
        v2 v[16];
        MEL_PHASE(0);
        // ---- 1. window: 8 x 16 bytes per lane, the first half is used while the second lands ----
        {
            v4f wv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) RD128_S(2, wv[j], aWin, T_WIN + 1024 * j);
            LDS_WAIT_N(4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                PIN(wv[j]);
                v[2 * j] = raw[2 * j] * lo2(wv[j]);
                v[2 * j + 1] = raw[2 * j + 1] * hi2(wv[j]);
            }
            LDS_WAIT_N(0);
#pragma unroll
            for (int j = 4; j < 8; ++j) {
                PIN(wv[j]);
                v[2 * j] = raw[2 * j] * lo2(wv[j]);
                v[2 * j + 1] = raw[2 * j + 1] * hi2(wv[j]);
            }
        }
        // ---- 1b. start fetching the next frame: in flight under the whole transform ---------
        if (f + 1 < fEnd) {
            int tn = t + 1, cn = clip;
            if (tn == a.timeLength) {
                tn = 0;
                ++cn;
            }
            const float *pn = a.x + (long long)cn * a.clipStride + (long long)tn * a.hop;
            bool whole = true;
            if constexpr (SHIFT > 0) {
                if (tn != 0) {
                    shift_rows_inplace<SHIFT>(raw);  // (in place: afx_asm.h)
                    fetch(pn, 16 - SHIFT);
                    whole = false;
                }
            }
            if (whole) fetch(pn, 0);
        }
        // ---- 1c. temporal features of the windowed frame (temporal_algorithm.c:138-144) -------
        if constexpr (TEMPORAL) {
            // sample order: n = 64 n1 + lane, (x, y) = samples 2n, 2n + 1; the sample before 2n is the
            // y of the previous lane (n1 unchanged) or, for lane 0, the y of lane 63 at n1 - 1
            float e = 0.f, z = 0.f;
            float prevTop = 0.f;  // y of lane 63 at n1 - 1, wave-uniform
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {
                e = fmaf(v[n1].x, v[n1].x, e);
                e = fmaf(v[n1].y, v[n1].y, e);
                float py = __shfl_up(v[n1].y, 1, 64);
                if (lane0) py = prevTop;
                const bool first = (n1 == 0) && lane0;  // sample 0 has no predecessor
                if (!first && v[n1].x * py < 0.f) z += 1.f;
                if (v[n1].y * v[n1].x < 0.f) z += 1.f;
                prevTop = __shfl(v[n1].y, 63, 64);
            }
            e = wave_sum(e);
            z = wave_sum(z);
            if (lane0) {
                a.energy[f] = e;
                a.rms[f] = sqrtf(e / (float)NFFT);
                a.zcr[f] = (float)((double)z / (double)NFFT);
            }
        }

        // ---- 2a. radix-16 over n1, twiddle W_1024^(lane k1), transpose through LDS -----------
        MEL_PHASE(1);
        dft16(v);
        {
            v4f tq[8];  // requested after the butterflies: held across them they would spill
#pragma unroll
            for (int j = 0; j < 8; ++j) RD128_S(2, tq[j], aWin, T_TW1 + 1024 * j);
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
                WR2_64_S(0, b, o[4 * g], o[4 * g + 1], 0, 72);
                WR2_64_S(0, b, o[4 * g + 2], o[4 * g + 3], 144, 216);
            }
        }
        wave_lds_sync();
        MEL_PHASE(2);
        {
            v4f rq[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) RD128_S(1, rq[jj], aE1r, 64 * jj);
            wave_lds_sync();
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                PIN(rq[jj]);
                v[2 * jj] = lo2(rq[jj]);
                v[2 * jj + 1] = hi2(rq[jj]);
            }
        }

        // ---- 2b. radix-16 over m1, twiddle W_64^(m2 j1) -> image V[q = k1 + 16 j1][m2] --------
        MEL_PHASE(3);
        if (!KO_ON(5)) dft16(v);
        {
            v4f tq[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) RD128_S(2, tq[j], aTw2, 16 * j);
            LDS_WAIT_N(0);
#pragma unroll
            for (int j = 0; j < 8; ++j) PIN(tq[j]);
            v2 o[16];
            o[0] = v[0];
#pragma unroll
            for (int j1 = 1; j1 < 16; ++j1) o[j1] = KO_ON(5) ? v[rev4(j1)] : cmul(v[rev4(j1)], (j1 & 1) ? hi2(tq[j1 >> 1]) : lo2(tq[j1 >> 1]));
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const unsigned b = aE2w + 2048 * g;  // j1 = 4g .. 4g+3, 512 bytes = 64 units apart
                WR2_64_S(0, b, o[4 * g], o[4 * g + 1], 0, 64);
                WR2_64_S(0, b, o[4 * g + 2], o[4 * g + 3], 128, 192);
            }
        }
        wave_lds_sync();

        // ---- 3. last radix-4 + real-input split -> spectrum values in registers --------------
        MEL_PHASE(4);
        float pk[2][4], pq[2][4], p512;  // |X|^2 (CPLX: real parts)
        float ik[CPLX ? 2 : 1][4], iq[CPLX ? 2 : 1][4], i512 = 0.f;  // CPLX: imaginary parts
        {
            v4f zalo[2], zahi[2], zblo[2], zbhi[2], wlo[2], whi[2];
            RD128_S(1, zalo[0], aAlo, 0);
            RD128_S(1, zahi[0], aAhi, 0);
            RD128_S(1, zblo[0], aB0lo, 0);
            RD128_S(1, zbhi[0], aB0hi, 0);
            RD128_S(2, wlo[0], aT3lo, 0);
            RD128_S(2, whi[0], aT3hi, 0);
            RD128_S(1, zalo[1], aAlo, 2048);
            RD128_S(1, zahi[1], aAhi, 2048);
            RD128_S(1, zblo[1], aB1lo, 0);
            RD128_S(1, zbhi[1], aB1hi, 0);
            RD128_S(2, wlo[1], aT3lo, 2048);
            RD128_S(2, whi[1], aT3hi, 2048);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if (s == 0) LDS_WAIT_N(6);
                else LDS_WAIT_N(0);
                PIN(zalo[s]); PIN(zahi[s]); PIN(zblo[s]); PIN(zbhi[s]); PIN(wlo[s]); PIN(whi[s]);
                v2 za0 = lo2(zalo[s]), za1 = hi2(zalo[s]), za2 = lo2(zahi[s]), za3 = hi2(zahi[s]);
                v2 zb0 = lo2(zblo[s]), zb1 = hi2(zblo[s]), zb2 = lo2(zbhi[s]), zb3 = hi2(zbhi[s]);
                dft4(za0, za1, za2, za3);  // Z[q + 256 j]
                dft4(zb0, zb1, zb2, zb3);  // Z[q' + 256 j], q' the mirror base (lane 0, s = 0: 128)
                v2 A0 = za0, A1 = za1, A2 = za2, A3 = za3;
                v2 B0 = zb3, B1 = zb2, B2 = zb1, B3 = zb0;  // partner of Z[q + 256 j] is Z[q' + 256 (3 - j)]
                if (s == 0) {
                    // lane 0: q = 0 mirrors itself and q' = 128 mirrors itself:
                    // (Z0, Z0) -> bins 0, 1024; (Z256, Z768); (Z128, Z896); (Z384, Z640); bin 512 below
                    if constexpr (CPLX) cplx_map(v2{za2.x, -za2.y}, a.specMap == 4, p512, i512);  // X[512] = conj(Z[512])
                    else p512 = za2.x * za2.x + za2.y * za2.y;
                    A2 = lane0 ? zb0 : za2;
                    A3 = lane0 ? zb1 : za3;
                    B0 = lane0 ? za0 : zb3;
                    B1 = lane0 ? za3 : zb2;
                    B2 = lane0 ? zb3 : zb1;
                    B3 = lane0 ? zb2 : zb0;
                }
                if constexpr (CPLX) {
                    const bool sq = a.specMap == 4;
                    const v2 AA[4] = {A0, A1, A2, A3}, BB[4] = {B0, B1, B2, B3};
                    const v2 ww[4] = {lo2(wlo[s]), hi2(wlo[s]), lo2(whi[s]), hi2(whi[s])};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v2 x, y;
                        split_pair_c(AA[j], BB[j], ww[j], x, y);
                        cplx_map(x, sq, pk[s][j], ik[CPLX ? s : 0][j]);
                        cplx_map(v2{y.x, -y.y}, sq, pq[s][j], iq[CPLX ? s : 0][j]);
                    }
                } else {
                    split_pair(A0, B0, lo2(wlo[s]), pk[s][0], pq[s][0]);
                    split_pair(A1, B1, hi2(wlo[s]), pk[s][1], pq[s][1]);
                    split_pair(A2, B2, lo2(whi[s]), pk[s][2], pq[s][2]);
                    split_pair(A3, B3, hi2(whi[s]), pk[s][3], pq[s][3]);
                }
            }
        }
        // every read of the images has returned (lgkmcnt(0) above): the power row may overwrite their tail.
        // CPLX: the imaginary parts' row goes to the START of the wave's region, dead until the next frame's first exchange
        // (held in registers for a second pass over one row they cost 17 registers across the first pass: two waves per
        // SIMD); its zero pad lies inside the images and is rewritten with it
        if constexpr (CPLX) {
            WR2ST_32(aP01 - PROW_OFF, ik[0][0], ik[0][1], 0, 4);
            WR2ST_32(aP23 - PROW_OFF, ik[0][2], ik[0][3], 0, 4);
            WR2ST_32(aP01 - PROW_OFF, ik[1][0], ik[1][1], 1, 5);
            WR2ST_32(aP01 - PROW_OFF, ik[1][2], ik[1][3], 9, 13);
            WR2ST_32(aQs1 - PROW_OFF, iq[0][1], iq[0][0], 9, 13);
            WR2ST_32(aQ23 - PROW_OFF, iq[0][3], iq[0][2], 0, 4);
            WR2ST_32(aQs1 - PROW_OFF, iq[1][3], iq[1][2], 0, 4);
            WR2ST_32(aQs1 - PROW_OFF, iq[1][1], iq[1][0], 8, 12);
            float *prowI = reinterpret_cast<float *>(wreg);
            if (lane0) prowI[512] = i512;
            prowI[1025 + lane] = 0.f;
            if (lane < PROW_F - 1025 - 64) prowI[1025 + 64 + lane] = 0.f;
        }
        WR2ST_32_S(4, aP01, pk[0][0], pk[0][1], 0, 4);
        WR2ST_32_S(4, aP23, pk[0][2], pk[0][3], 0, 4);
        WR2ST_32_S(4, aP01, pk[1][0], pk[1][1], 1, 5);
        WR2ST_32_S(4, aP01, pk[1][2], pk[1][3], 9, 13);
        WR2ST_32_S(4, aQs1, pq[0][1], pq[0][0], 9, 13);
        WR2ST_32_S(4, aQ23, pq[0][3], pq[0][2], 0, 4);
        WR2ST_32_S(4, aQs1, pq[1][3], pq[1][2], 0, 4);
        WR2ST_32_S(4, aQs1, pq[1][1], pq[1][0], 8, 12);
        if (lane0) prow[512] = p512;
        wave_lds_sync();
#pragma unroll
        for (int pass = 0; pass < (CPLX ? 2 : 1); ++pass) {
        const unsigned bpa = (CPLX && pass) ? apa - PROW_OFF : apa, bpb = (CPLX && pass) ? apb - PROW_OFF : apb;
        if (!CPLX && a.specMap) {
            // magnitude / norm exponent (rare modes): one pass over the row in LDS.  (Applied to the 17 register values
            // before the stores, the two branches' results met the plain path's in different registers and the plain
            // path paid 17 moves per frame for it.)
            for (int k = lane; k < 1025; k += 64) {
                const float p = prow[k];
                prow[k] = a.specMap == 1 ? sqrtf(p) : powf(p, a.normValue);
            }
            wave_lds_sync();
        }

        if constexpr (STFT) {
            // ---- 4'. the row itself: 16-byte reads of the natural-order row, 1 KB per store instruction of the wave ----
            MEL_PHASE(6);
            float *orow = a.out + f * a.outPitch;
            if (a.vecOut) {
                const float4 *p4 = reinterpret_cast<const float4 *>(prow);
                float4 *o4 = reinterpret_cast<float4 *>(orow);
                float4 q[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) q[i] = p4[lane + 64 * i];
#pragma unroll
                for (int i = 0; i < 4; ++i) o4[lane + 64 * i] = q[i];
                if (lane0) o4[256] = p4[256];  // bin 1024 and three words of the zero pad
            } else {
                for (int k = lane; k < a.binCount; k += 64) orow[k] = prow[a.binLo + k];
            }
        } else {
        MEL_PHASE(5);
        // ---- 4. banded filter bank: weights by ds_read_b128, power row by immediate-offset
        //         ds_read_b64 (conflict-free by the plan's bank-aware lane assignment); the NEXT
        //         block of four quads is requested before this block's values are waited for --------
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
                    RD128_S(3, wq[i], awr, 16 * q);
                    if (q < QA) {
                        RD64_S(3, q0v[i], bpa, 16 * q);
                        RD64_S(3, q1v[i], bpa, 16 * q + 8);
                    } else {
                        RD64_S(3, q0v[i], bpb, 16 * (q - QA));
                        RD64_S(3, q1v[i], bpb, 16 * (q - QA) + 8);
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
                    if (KO_ON(6)) {
                        asm volatile("" ::"v"(w[cur][i]), "v"(p0[cur][i]), "v"(p1[cur][i]));
                    } else if (q < QA) {
                        sA += lo2(w[cur][i]) * p0[cur][i];
                        sA += hi2(w[cur][i]) * p1[cur][i];
                    } else {
                        sB += lo2(w[cur][i]) * p0[cur][i];
                        sB += hi2(w[cur][i]) * p1[cur][i];
                    }
                }
                if constexpr (CPLX) {  // this block's sums before the next block's requests: left free, the multiply-adds of the
                    PIN(sA);           // complex instantiations sink behind the last request with every operand alive (240-256
                    PIN(sB);           // registers + 264 bytes of scratch -> 182)
                }
            }
            accA = sA.x + sA.y;
            accB = sB.x + sB.y;
        }
        if (!CPLX && !SPLIT && a.postPow) {
            accA = powf(accA, a.normValue);
            accB = powf(accB, a.normValue);
        }
        MEL_PHASE(6);
        // ---- 5. store (first the cepstra of the 16 rows stored before this one, if that many wait) ----
        if constexpr (CC != 0 && !SPLIT) {
            if (ccN == AFX_V2_CCEVERY) cc_block(f - AFX_V2_CCEVERY, AFX_V2_CCEVERY);
        }
        float *orow = ((CPLX && pass) ? a.outIm : a.out) + f * a.num;
        if constexpr (SPLIT) {
            // slot results -> LDS (start of the wave's region: the image there is dead since stage 3),
            // then every row is the sum of its segments in ascending bins
            float *part = reinterpret_cast<float *>(wreg + (CPLX ? PROW_F * 4 : 0));  // (CPLX: behind the imaginary parts' row)
            static_assert(PROW_F * 4 + 129 * 4 <= PROW_OFF, "segment sums must fit between the two rows");
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
        if constexpr (CC != 0) {
            ++ccN;
            // the wave's last rows (drains its last stores).  Split plans: ONE call site, behind the row's stores where the
            // band stage's values are dead (beside them the block spilled 336-656 bytes per lane); its wait then covers the
            // stores of the 16th row as well, once per 16 frames
            if (f + 1 == fEnd || (SPLIT && ccN == 16)) cc_block(f + 1 - ccN, ccN);
        }
        }  // !STFT
        wave_lds_sync();  // the next frame overwrites the images / the power row

        }  // pass

        if (++t == a.timeLength) {
            t = 0;
            ++clip;
        }
    }
}

// window (hWindow != nullptr) and twiddle tables of the blob, byte offsets T_WIN / T_TW1 / T_TW2 / T_TW3
void fill_tables(float *tab, const float *hWindow) {
    const double PI = 3.14159265358979323846;
    // window and W_1024^(lane k1) in pair layout: entry (n1, lane) at float2 index 128 (n1 >> 1) + 2 lane + (n1 & 1)
    float *win = tab + T_WIN / 4, *tw1 = tab + T_TW1 / 4, *tw2 = tab + T_TW2 / 4, *tw3 = tab + T_TW3 / 4;
    for (int n1 = 0; n1 < 16; ++n1)
        for (int l = 0; l < 64; ++l) {
            const int at = 2 * (128 * (n1 >> 1) + 2 * l + (n1 & 1));
            const int n = 64 * n1 + l;
            if (hWindow) {
                win[at] = hWindow[2 * n];
                win[at + 1] = hWindow[2 * n + 1];
            }
            const double ang = -2.0 * PI * (double)(n1 * l) / MC;  // twiddles in double, rounded once
            tw1[at] = (float)cos(ang);
            tw1[at + 1] = (float)sin(ang);
        }
    for (int m = 0; m < 4; ++m)
        for (int j = 0; j < 16; ++j) {
            const double ang = -2.0 * PI * (double)(m * j) / 64.0;
            tw2[(TW2_PITCH / 4) * m + 2 * j] = (float)cos(ang);
            tw2[(TW2_PITCH / 4) * m + 2 * j + 1] = (float)sin(ang);
        }
    // 0.5 W_2048^bin of the P-bin of slot (s, lane, m); 16-byte halves swapped when bit 3 of lane is set
    for (int s = 0; s < 2; ++s)
        for (int l = 0; l < 64; ++l)
            for (int m = 0; m < 4; ++m) {
                int bin = l + 64 * s + 256 * m;
                if (s == 0 && l == 0 && m >= 2) bin = m == 2 ? 128 : 384;  // lane 0 carries the self-mirrored base
                const double ang = -2.0 * PI * (double)bin / NFFT;
                const int at = 2 * (4 * (64 * s + l) + 2 * ((m >> 1) ^ ((l >> 3) & 1)) + (m & 1));
                tw3[at] = (float)(0.5 * cos(ang));
                tw3[at + 1] = (float)(0.5 * sin(ang));
            }
}

struct Plan2 {
    int variant, num, split;
    float4 *dTab;
    int *dMeta;
};
struct Variant {
    int tapsA, tapsB;
};
constexpr Variant kVariants[] = {{48, 16}, {72, 32}};

template <int TA, int TB, int SHIFT, bool SPLIT, int CC, bool TEMPORAL, bool CPLX = false>
int launch_variant(const Plan2 *p, const AfxMelFusedArgs *a, void *stream) {
    const long long total = (long long)a->batch * a->timeLength;
    if (total <= 0) return AFX_OK;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    // one 12-wave workgroup is resident per CU; two rounds of workgroups keep the tail short while each
    // wave still streams a long contiguous run of frames (and re-uses 3/4 of every frame from registers)
    constexpr int NWV = waves_of(CPLX);
    long long waves = (long long)cus * NWV * 2;  // 1 / 2 / 3 rounds measure the same (1.558 / 1.559 / 1.559 ms), 6: +1.2 %
    long long fpw = (total + waves - 1) / waves;
    // long runs per wave (register re-use of the overlapping frames) once a round of workgroups is full; a call that
    // cannot fill one round -- the one-clip legacy entry points: 1000 frames -- is spread over all CUs instead
    // (16 frames in sequence per wave were 75 us of a 1000-frame call's 190, profiles/r05_legacy_phases.txt)
    if (fpw < 16) {
        const long long oneRound = (total + (long long)cus * NWV - 1) / ((long long)cus * NWV);
        fpw = oneRound < 16 ? oneRound : 16;
    }
    const long long usedWaves = (total + fpw - 1) / fpw;
    const long long blocks = (usedWaves + NWV - 1) / NWV;

    KArgs2 k;
    memset(&k, 0, sizeof(k));
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
    k.energy = a->energy;
    k.rms = a->rms;
    k.zcr = a->zcr;
    constexpr size_t lds = (size_t)block_lds_bytes(TA, TB, CC == 1);
    static_assert(lds <= 163840, "workgroup LDS budget");
    static std::atomic<bool> attrSet[AFX_MAX_DEVICES];  // per device: the attribute lives in the device's code object
    const int attrDev = afxdev_current_device() & (AFX_MAX_DEVICES - 1);
    if (!attrSet[attrDev].load(std::memory_order_acquire)) {  // (two threads may both set it: idempotent)
        AFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_stft_mel_v2<TA, TB, SHIFT, SPLIT, CC, TEMPORAL, CPLX>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attrSet[attrDev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((k_stft_mel_v2<TA, TB, SHIFT, SPLIT, CC, TEMPORAL, CPLX>), dim3((unsigned)blocks), dim3(NWV * 64), lds,
                       (hipStream_t)stream, k);
    AFX_LAUNCH_CHECK("k_stft_mel_v2");
    return AFX_OK;
}

template <int TA, int TB, bool SPLIT, int CC, bool TEMPORAL>
int launch_hop(const Plan2 *p, const AfxMelFusedArgs *a, void *stream) {
    // register re-use of the overlapping frames for hop = 128 * SHIFT: N/8, N/4, N/2
#ifdef AFX_EXPERIMENTS  // measurement builds only (make EXTRA=-DAFX_EXPERIMENTS): AFX_EXP_MEL=noshift fetches every frame whole
    if (const char *e = getenv("AFX_EXP_MEL"))
        if (strstr(e, "noshift")) return launch_variant<TA, TB, 0, SPLIT, CC, TEMPORAL>(p, a, stream);
#endif
    switch (a->hop) {
        case 256: return launch_variant<TA, TB, 2, SPLIT, CC, TEMPORAL>(p, a, stream);
        case 512: return launch_variant<TA, TB, 4, SPLIT, CC, TEMPORAL>(p, a, stream);
        case 1024: return launch_variant<TA, TB, 8, SPLIT, CC, TEMPORAL>(p, a, stream);
        default: return launch_variant<TA, TB, 0, SPLIT, CC, TEMPORAL>(p, a, stream);
    }
}

template <int TA, int TB>
int launch(const Plan2 *p, const AfxMelFusedArgs *a, void *stream) {
    const bool cc = a->cc != nullptr, tmp = a->energy != nullptr;
    if (a->specMap >= 3) {  // complex results: S (3) or S^2 (4); hop N/4 with register re-use, any other hop plain
        if (cc || tmp) return AFX_ERR_UNSUPPORTED;
        if (!a->outIm) return AFX_ERR_ARG;
        if (a->hop == 512)
            return p->split ? launch_variant<TA, TB, 4, true, 0, false, true>(p, a, stream)
                            : launch_variant<TA, TB, 4, false, 0, false, true>(p, a, stream);
        return p->split ? launch_variant<TA, TB, 0, true, 0, false, true>(p, a, stream)
                        : launch_variant<TA, TB, 0, false, 0, false, true>(p, a, stream);
    }
    if (cc && tmp) return AFX_ERR_UNSUPPORTED;  // callers run the cepstra separately for temporal objects
    if (cc) {
        if (a->ccNum < 1 || a->ccNum > 16 || !a->dct || !a->out || p->num > 128 || (p->num & 3)) return AFX_ERR_UNSUPPORTED;
        if (a->ccRectify != 0 && a->ccRectify != 1) return AFX_ERR_UNSUPPORTED;
        if constexpr (block_lds_bytes(TA, TB, true) <= 163840)  // the headline form: DCT operand in LDS
            if (!p->split && p->num == 128 && a->ccRectify == 0) return launch_hop<TA, TB, false, 1, false>(p, a, stream);
        // the general form (afx_ccblock.h): hop N/4 with the register re-use of the overlapping frames, any other hop plain
        if (a->hop == 512)
            return p->split ? launch_variant<TA, TB, 4, true, 2, false>(p, a, stream) : launch_variant<TA, TB, 4, false, 2, false>(p, a, stream);
        return p->split ? launch_variant<TA, TB, 0, true, 2, false>(p, a, stream) : launch_variant<TA, TB, 0, false, 2, false>(p, a, stream);
    }
    if (tmp) {
        if (!a->rms || !a->zcr) return AFX_ERR_ARG;
        return p->split ? launch_hop<TA, TB, true, 0, true>(p, a, stream)
                        : launch_hop<TA, TB, false, 0, true>(p, a, stream);
    }
    return p->split ? launch_hop<TA, TB, true, 0, false>(p, a, stream)
                    : launch_hop<TA, TB, false, 0, false>(p, a, stream);
}

// twiddle tables of the STFT instantiations (the blob without window and bank), one device copy per device, never freed
const float4 *stft2k_tables() {
    static std::mutex mu;
    static float4 *dTab[AFX_MAX_DEVICES] = {};
    const int dev = afxdev_current_device();
    if (dev < 0 || dev >= AFX_MAX_DEVICES) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    if (!dTab[dev]) {
        const size_t bytes = (size_t)tab_bytes(0, 0);
        float *h = static_cast<float *>(calloc(bytes, 1));
        if (!h) return nullptr;
        fill_tables(h, nullptr);
        float4 *d = nullptr;
        int st = afxdev_malloc(reinterpret_cast<void **>(&d), bytes);
        // (a synchronous copy: the caller's stream is not waited for under this lock)
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

template <int SHIFT>
int launch_stft2k(const AfxStftArgs *a, const float4 *tab, void *stream) {
    const long long total = (long long)a->batch * a->timeLength;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    // the frame distribution of launch_variant: two rounds of 12-wave workgroups, short calls spread over every CU
    long long waves = (long long)cus * WAVES * 2;
    long long fpw = (total + waves - 1) / waves;
    if (fpw < 16) {
        const long long oneRound = (total + (long long)cus * WAVES - 1) / ((long long)cus * WAVES);
        fpw = oneRound < 16 ? oneRound : 16;
    }
    const long long usedWaves = (total + fpw - 1) / fpw;
    const long long blocks = (usedWaves + WAVES - 1) / WAVES;
    KArgs2 k;
    memset(&k, 0, sizeof(k));
    k.x = a->x;
    k.clipStride = a->clipStride;
    k.totalFrames = total;
    k.timeLength = a->timeLength;
    k.hop = a->hop;
    k.framesPerWave = (int)fpw;
    k.aligned = ((a->clipStride & 1) == 0) && ((a->hop & 1) == 0) && ((reinterpret_cast<uintptr_t>(a->x) & 7) == 0);
    k.tab = tab;
    k.specMap = a->mode == AFX_SPEC_POWER ? 0 : a->mode == AFX_SPEC_MAG ? 1 : 2;
    k.normValue = a->normValue;
    k.out = a->outRe;
    k.win = a->window;
    k.binLo = a->binLo;
    k.binCount = a->binCount;
    k.outPitch = a->outPitch ? a->outPitch : (long long)a->binCount;
    k.vecOut = a->binLo == 0 && a->binCount == NFFT / 2 + 1 && k.outPitch >= 1028 && (k.outPitch & 3) == 0 &&
               (reinterpret_cast<uintptr_t>(a->outRe) & 15) == 0;
    constexpr size_t lds = (size_t)block_lds_bytes(0, 0, false);
    static std::atomic<bool> attrSet[AFX_MAX_DEVICES];
    const int attrDev = afxdev_current_device() & (AFX_MAX_DEVICES - 1);
    if (!attrSet[attrDev].load(std::memory_order_acquire)) {
        AFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_stft_mel_v2<0, 0, SHIFT, false, 0, false, false, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attrSet[attrDev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((k_stft_mel_v2<0, 0, SHIFT, false, 0, false, false, true>), dim3((unsigned)blocks), dim3(WAVES * 64), lds,
                       (hipStream_t)stream, k);
    AFX_LAUNCH_CHECK("k_stft_mel_v2 (stft)");
    return AFX_OK;
}

}  // namespace

// The n_fft 2048 wave transform storing its mapped spectrum rows (real results: |S|^2, |S|, |S|^2p) -- the [T, F] rows of the
// dense-bank route and of the STFT / linear-scale objects (stft_algorithm.c:717-803, bft_algorithm.c:489-504); every frame inside
// its clip.  AFX_ERR_UNSUPPORTED: not this kernel's case (afxk_stft then runs k_stft_wave / the size-generic kernel)
extern "C" int afxk_stft2k(const AfxStftArgs *a, void *stream) {
    if (a->radix2Exp != 11 || a->bandStart || a->energy || a->padLeft != 0 || a->hop < 1 || a->binLo < 0 || a->binCount < 1 ||
        a->binLo + a->binCount > NFFT / 2 + 1 || (long long)(a->timeLength - 1) * a->hop + NFFT > a->dataLength ||
        (reinterpret_cast<uintptr_t>(a->window) & 7) != 0)
        return AFX_ERR_UNSUPPORTED;
    if (a->mode != AFX_SPEC_POWER && a->mode != AFX_SPEC_MAG && a->mode != AFX_SPEC_POWER_NORM) return AFX_ERR_UNSUPPORTED;
    if (!a->outRe) return AFX_ERR_ARG;
    const long long total = (long long)a->batch * a->timeLength;
    if (total <= 0) return AFX_OK;
    if (total > 0x7fffffffLL) return AFX_ERR_UNSUPPORTED;
    const float4 *tab = stft2k_tables();
    if (!tab) return AFX_ERR_UNSUPPORTED;
    switch (a->hop) {
        case 256: return launch_stft2k<2>(a, tab, stream);
        case 512: return launch_stft2k<4>(a, tab, stream);
        case 1024: return launch_stft2k<8>(a, tab, stream);
        default: return launch_stft2k<0>(a, tab, stream);
    }
}

extern "C" void afxk_mel2_destroy(void *plan) {
    Plan2 *p = static_cast<Plan2 *>(plan);
    if (!p) return;
    afxdev_free(p->dTab);
    afxdev_free(p->dMeta);
    free(p);
}

// variant: index into {48+16, 72+32} taps (afxk_melfused_variant for radix2Exp 11)
extern "C" int afxk_mel2_create(void **plan, int variant, const float *hWindow, const AfxBandPlan *band, void *stream) {
    *plan = nullptr;
    if (variant < 0 || variant > 1) return AFX_ERR_UNSUPPORTED;
    const int TA = kVariants[variant].tapsA, TB = kVariants[variant].tapsB;
    const int WP = wpitch(TA, TB);
    const size_t bytes = (size_t)tab_bytes(TA, TB);
    Plan2 *p = static_cast<Plan2 *>(calloc(1, sizeof(Plan2)));
    float *tab = static_cast<float *>(calloc(bytes, 1));
    if (!p || !tab) {
        free(p);
        free(tab);
        return AFX_ERR_NOMEM;
    }
    p->variant = variant;
    p->num = band->num;
    p->split = band->split;
    fill_tables(tab, hWindow);
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
        afxk_mel2_destroy(p);
        return st;
    }
    *plan = p;
    return AFX_OK;
}

// specMap 0 / 1 / 2: real results, 3 / 4: complex results (out + outIm); AFX_ERR_UNSUPPORTED when the requested fusion
// (cepstra / temporal features) does not apply to this plan
extern "C" int afxk_mel2_run(void *plan, const AfxMelFusedArgs *a, void *stream) {
    const Plan2 *p = static_cast<const Plan2 *>(plan);
    if (!p || a->specMap > 4) return AFX_ERR_ARG;
    return p->variant == 0 ? launch<48, 16>(p, a, stream) : launch<72, 32>(p, a, stream);
}
