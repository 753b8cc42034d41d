// afx_gemm_bf16.hip -- C[M,N] = A[M,K] B[N,K]^T on the bf16 matrix cores with float32-equivalent operands.
//
// The 128 x 128 tile kernel of afxk_gemm_nt (wide, 16-byte aligned shapes: the dense filter-bank products).  Written
// at the end of round 2, first run on the device in round 3: parity tests unchanged, the dense gammatone-128 route
// 5.40 -> 5.04 ms per 934 000 frames against the float32 matrix-core kernel it replaces (profiles/r03_round_start.txt).
//
// Round 2's k_gemm_nt128 ran the dense filter-bank products (gammatone / chroma banks x spectra) at
// 110 TFLOP/s on v_mfma_f32_32x32x2_f32 -- 70 % of a pipe that runs at the vector rate (64 cycles per 32x32x2).
// v_mfma_f32_32x32x16_bf16 does 8x the products in half the time.  Power spectra span ten decades inside one frame
// and feed a logarithm, so (unlike the CQT's f16 words, afx_cqt_f16.hip) no per-tile exponent will do: every float32
// value is split into THREE bf16 words (float32's exponent range; 8 + 8 + 8 significant bits)
//     a = a_h + a_m + a_l,   b = b_h + b_m + b_l
// and the product keeps the six terms down to 2^-24 of the leading one,
//     a_h b_h + (a_h b_m + a_m b_h) + (a_m b_m + a_h b_l + a_l b_h),
// accumulated in float32 by the matrix core: six MFMAs of 32 cycles per 16 k-steps = 192 cycles against 512 on the
// f32 pipe.  numpy model: elementwise relative error 8.5e-7, the f32 GEMM's 9.4e-7.
//
// Tiling: 128 x 128 outputs per workgroup, four waves own 64 x 64 quadrants (2 x 2 MFMA tiles each),
// K in steps of 16 (one MFMA k-step), double-buffered LDS.  The loader converts while it stages: a thread takes four
// consecutive k of two A rows and two B rows (16-byte loads), forms the three words (v_cvt_pk_bf16_f32, round to
// nearest even; the remainders are exact float32 subtractions) and stores 8 bytes per word plane.  A plane row is
// 16 bf16 = 32 bytes + a 16-byte pad: the fragment of lane (i = lane & 31, g = lane >> 5) is the 16 bytes at
// row 48 + 16 g, and 3 i + g covers the 16 bank quads of every ds_read_b128 lane group.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include <afx_asm.h>

#include "afx_device.h"
#include "afx_hipcheck.h"

#ifndef AFX_WAVES_PER_EU  // (tests/emu defines it away: the attribute is the device compiler's)
#define AFX_WAVES_PER_EU(lo, hi) __attribute__((amdgpu_waves_per_eu(lo, hi)))
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int TM = 128, TN = 128, TK = 16;
constexpr int ROW = 48;                       // bytes per plane row (32 of data + 16 of pad)
constexpr int PLANE = TM * ROW;               // one word plane of one operand
constexpr int OPER = 3 * PLANE;               // hi | mid | lo
constexpr int STAGE = 2 * OPER;               // A | B
constexpr int LDS_BYTES = 2 * STAGE;          // double buffer: 73 728 bytes

// four float32 -> the three bf16 words of each (round to nearest even; remainders exact)
__device__ __forceinline__ void split3(const f32x4 v, bf4 &h, bf4 &m, bf4 &l) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const __bf16 hh = (__bf16)v[c];
        const float r1 = v[c] - (float)hh;
        const __bf16 mm = (__bf16)r1;
        const float r2 = r1 - (float)mm;
        h[c] = hh;
        m[c] = mm;
        l[c] = (__bf16)r2;
    }
}

__global__ __launch_bounds__(256, 2) void k_gemm_nt128_bf16x3(const float *__restrict__ A, long long lda,
                                                              const float *__restrict__ B, int ldb,
                                                              float *__restrict__ C, long long ldc, long long M, int N,
                                                              int K, int post, float postArg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const long long m0 = (long long)blockIdx.x * TM;
    const int n0 = blockIdx.y * TN;
    const int lrow = tid >> 2, kq = tid & 3;  // loader: rows lrow, lrow + 64; k = k0 + 4 kq .. + 3

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto gload = [&](int k0, f32x4 (&ra)[2], f32x4 (&rb)[2]) {
        const int k = k0 + 4 * kq;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const long long m = m0 + lrow + 64 * p;
            const int n = n0 + lrow + 64 * p;
            f32x4 va = {0.f, 0.f, 0.f, 0.f}, vb = {0.f, 0.f, 0.f, 0.f};
            if (k < K) {
                if (m < M) va = *reinterpret_cast<const f32x4 *>(A + m * lda + k);
                if (n < N) vb = *reinterpret_cast<const f32x4 *>(B + (long long)n * ldb + k);
                if (k + 3 >= K) {  // tail: words at k >= K are padding
#pragma unroll
                    for (int c = 1; c < 4; ++c)
                        if (k + c >= K) {
                            va[c] = 0.f;
                            vb[c] = 0.f;
                        }
                }
            }
            ra[p] = va;
            rb[p] = vb;
        }
    };
    auto sstore = [&](int buf, const f32x4 (&ra)[2], const f32x4 (&rb)[2]) {
        unsigned char *base = smem + buf * STAGE;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int off = (lrow + 64 * p) * ROW + 8 * kq;
            bf4 h, m, l;
            split3(ra[p], h, m, l);
            *reinterpret_cast<bf4 *>(base + off) = h;
            *reinterpret_cast<bf4 *>(base + PLANE + off) = m;
            *reinterpret_cast<bf4 *>(base + 2 * PLANE + off) = l;
            split3(rb[p], h, m, l);
            *reinterpret_cast<bf4 *>(base + OPER + off) = h;
            *reinterpret_cast<bf4 *>(base + OPER + PLANE + off) = m;
            *reinterpret_cast<bf4 *>(base + OPER + 2 * PLANE + off) = l;
        }
    };

    const int nk = (K + TK - 1) / TK;
    f32x4 ra[2], rb[2];
    gload(0, ra, rb);
    sstore(0, ra, rb);
    __syncthreads();
    // fragment addresses of this lane inside a stage: A rows 64 wr + 32 ti + i, B rows 64 wc + 32 tj + i
    const int fragOff = (lane & 31) * ROW + 16 * (lane >> 5);
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * TK, ra, rb);
        const unsigned char *sa = smem + buf * STAGE + (64 * wr) * ROW + fragOff;
        const unsigned char *sb = smem + buf * STAGE + OPER + (64 * wc) * ROW + fragOff;
        bf8 a[2][3], b[2][3];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int w = 0; w < 3; ++w) {
                a[t][w] = *reinterpret_cast<const bf8 *>(sa + w * PLANE + 32 * t * ROW);
                b[t][w] = *reinterpret_cast<const bf8 *>(sb + w * PLANE + 32 * t * ROW);
            }
        // six terms, smallest first; the four tiles take turns, so an accumulator is reused every fourth MFMA
#define AFX_TERM(WA, WB)                                                                                               \
    do {                                                                                                               \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][WA], b[0][WB], acc[0][0], 0, 0, 0);                   \
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][WA], b[1][WB], acc[0][1], 0, 0, 0);                   \
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][WA], b[0][WB], acc[1][0], 0, 0, 0);                   \
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][WA], b[1][WB], acc[1][1], 0, 0, 0);                   \
    } while (0)
        AFX_TERM(2, 0);  // a_l b_h
        AFX_TERM(0, 2);  // a_h b_l
        AFX_TERM(1, 1);  // a_m b_m
        AFX_TERM(1, 0);  // a_m b_h
        AFX_TERM(0, 1);  // a_h b_m
        AFX_TERM(0, 0);  // a_h b_h
#undef AFX_TERM
        if (kt + 1 < nk) sstore(buf ^ 1, ra, rb);
        __syncthreads();
    }

    // C/D layout of the 32x32 MFMA: col j = lane&31, row i = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
            const int j = n0 + 64 * wc + 32 * tj + (lane & 31);
            if (j >= N) continue;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int i = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                const long long m = m0 + 64 * wr + 32 * ti + i;
                if (m < M) {
                    float v = acc[ti][tj][reg];
                    if (post == AFX_MAP_POW) v = powf(v, postArg);
                    C[m * ldc + j] = v;
                }
            }
        }
}


// ---- round 6: the dense FILTER-BANK product (afx_bft.c, dense branch) with the bank prepared once ------------------------
//
// What held k_gemm_nt128_bf16x3 at 35 % matrix-pipe utilisation on the gammatone-128 route (profiles/r04_rocprofv3_dense_gemm.txt):
// (1) the loader converts BOTH operands while it stages them, the bank's 128 x 16 tile again for every one of the 7 300 row
// tiles of a step -- as many vector instructions as matrix cycles; (2) the next tile is requested ONE k-step (0.37 us of matrix
// work) before it is converted: every k-step waits for memory; (3) both operands cross the LDS, whose pipe the two resident
// workgroups share: 73 KB per workgroup and k-step, all of it between two barriers.  Here
//   * the bank is split into its three bf16 word planes ONCE per object (k_bank_split -> "bank image") and laid out in
//     FRAGMENT order: per column tile and k-step, for each column half wc, 32-column MFMA tile tj and word w, the 64 lanes' 16
//     bytes in lane order.  A wave takes its B operands straight from memory into registers (six 1 KB loads per k-step,
//     L2-resident: 787 KB for 128 x 1025) -- the bank never touches the LDS;
//   * the float32 rows of A are requested a whole stage (two k-steps) ahead of their split -- three ahead of their MFMAs -- into a
//     ring of registers, the bank fragments one k-step ahead, both by hand (afx_asm.h) and waited for by count: the compiler's own
//     s_waitcnt placement drains every older load at the loop header (a prefetch distance of one k-step whatever the source says);
//   * A is split with integer arithmetic on the bit patterns (add / mask / subtract per word, half-word packs): exact,
//     8 + 8 + 8 bits cover float32's 24, so the six-term product below is the same expansion as the converting split's;
//   * the 24 MFMAs of a k-step keep their source order (the four tiles take turns: no accumulator is reused before its
//     fourth successor) and the split of the NEXT stage's rows is cut into 20 pieces of 3-5 vector instructions per k-step, one
//     behind each MFMA: the scheduler is held to that order (sched_barrier), left alone it gathers the MFMAs into chains on one
//     accumulator and the vector work behind them;
//   * tile, A fragments and epilogue as in k_gemm_nt128_bf16x3.
// Measured (profiles/r06_dense.txt, 233 500 x 128 x 1025 per launch): 384-404 us by box = 152-160 TF/s float32-equivalent,
// 0.91-0.96 PF/s of bf16 matrix work (37 % of the dense peak), matrix pipe 48 % busy at 2.0 GHz (round 4: 475 us, 35 %).  The
// knock-out table prices the A loads at 35 % and the fragment loads at 30 % of the launch (neither alone: 203 us = the MFMAs'
// own 190), LDS stores and the split at 9-11 % each, the barrier at nothing.  A 64 x 64 wave tile takes 12 KB of operands into
// registers per 24 MFMAs -- 62 bytes per cycle and CU against 64 from the vector memory pipe and 128 from the LDS, whichever
// mix serves them (fragments through the LDS measured the same: 417 us).  Two further forms measured slower and reverted
// (profiles/r06_dense.txt (c'), (c'')): 128 x 64 wave tiles at one wave per SIMD (474 us), loader waves beside multiplying waves (463 us).
// Knock-out measurement builds (tools/build_variant.sh kog<mask> -DAFX_KO_GEMM=<mask> afx_gemm_bf16; results WRONG, timing only):
// 1 A quads not loaded, 2 bank fragments not loaded, 4 no barrier, 8 A words not stored to the LDS, 16 split arithmetic dropped,
// 32 A fragments not read from the LDS, 64 no MFMAs, 128 every workgroup reads the rows of tile 0 (cache hits) (profiles/r06_dense.txt)
#ifndef AFX_KO_GEMM
#define AFX_KO_GEMM 0
#endif
constexpr int KO_A = (AFX_KO_GEMM >> 0) & 1 ? 0 : 4, KO_B = (AFX_KO_GEMM >> 1) & 1 ? 0 : 6;  // loads a stage half really issues
constexpr int KO_YOUNGER = KO_A + KO_B;       // 10: six fragments + four quads younger than the fragments about to be used
constexpr int IMG_U4 = 3 * TM * 2;            // u32x4 per (column tile, k-step) of the bank image: 12 fragments of 64 lanes
constexpr int LDS_BANK_BYTES = 2 * STAGE;     // A only: two k-steps (OPER each) per stage, double buffer: 73 728 bytes

__device__ __forceinline__ unsigned f2u(float x) { return __float_as_uint(x); }
__device__ __forceinline__ float u2f(unsigned x) { return __uint_as_float(x); }
// upper half-words of (x0, x1) as (low, high) half of one word
__device__ __forceinline__ unsigned pack_hi(unsigned x0, unsigned x1) { return (x1 & 0xffff0000u) | (x0 >> 16); }

// One float32 -> its three bf16 words, in two halves (the pieces the kernel places between its MFMAs).  hi and mid are ROUNDED
// (half up in magnitude: one integer add before the mask), lo takes what is left: |r1| <= 2^-9 |x| has at most 16 significant
// bits, |r2| <= 2^-9 |r1| at most 7 -- the split is exact.  (Plain truncation is exact as well and two instructions shorter,
// but leaves every word with the sign of x: the small terms of the product, all of one sign, are then rounded away one by one
// against the large accumulator -- a bias of -2.4e-6 of the result at K = 1025 under tests/emu, against 5e-7 for words of
// either sign.)
__device__ __forceinline__ void split_hi(float x, unsigned &hb, float &r1) {
    hb = (f2u(x) + 0x8000u) & 0xffff0000u;
    r1 = x - u2f(hb);
}
__device__ __forceinline__ void split_mid(float r1, unsigned &mb, float &r2) {
    mb = (f2u(r1) + 0x8000u) & 0xffff0000u;
    r2 = r1 - u2f(mb);
}

// bank [N, K] (row pitch ldb floats) -> image; one thread per 8 words of a row: row r = 64 wc + 32 tj + i of the tile, k-block g
// -> lane i + 32 g of fragment (wc, tj, word)
__global__ __launch_bounds__(256) void k_bank_split(const float *__restrict__ B, int ldb, int N, int K, u32x4 *__restrict__ img) {
    const int nk = (K + TK - 1) / TK;
    const int tile = blockIdx.x / nk, kt = blockIdx.x - tile * nk;
    const int row = threadIdx.x >> 1, g = threadIdx.x & 1;
    const int n = tile * TN + row, k0 = kt * TK + 8 * g;
    unsigned hb[8], mb[8];
    float r2[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int k = k0 + c;
        const float x = (n < N && k < K) ? B[(long long)n * ldb + k] : 0.f;
        float r1;
        split_hi(x, hb[c], r1);
        split_mid(r1, mb[c], r2[c]);
    }
    u32x4 *dst = img + (size_t)blockIdx.x * IMG_U4 + (size_t)(row >> 5) * 3 * 64 + (row & 31) + 32 * g;
    dst[0] = u32x4{hb[1] | (hb[0] >> 16), hb[3] | (hb[2] >> 16), hb[5] | (hb[4] >> 16), hb[7] | (hb[6] >> 16)};
    dst[64] = u32x4{mb[1] | (mb[0] >> 16), mb[3] | (mb[2] >> 16), mb[5] | (mb[4] >> 16), mb[7] | (mb[6] >> 16)};
    dst[128] = u32x4{pack_hi(f2u(r2[0]), f2u(r2[1])), pack_hi(f2u(r2[2]), f2u(r2[3])), pack_hi(f2u(r2[4]), f2u(r2[5])),
                     pack_hi(f2u(r2[6]), f2u(r2[7]))};
}

// (two workgroups per CU: two waves per SIMD, 256 registers each -- the rings live in registers)
__global__ __launch_bounds__(256) AFX_WAVES_PER_EU(2, 2) void k_gemm_bank_bf16x3(const float *__restrict__ A, long long lda,
                                                                                const u32x4 *__restrict__ img, float *__restrict__ C,
                                                                                long long ldc, long long M, int N, int K, int post,
                                                                                float postArg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const long long m0 = (long long)blockIdx.x * TM;
    const int n0 = blockIdx.y * TN;
    __builtin_assume(K >= 1);          // (the launcher's contract: the stage loop runs at least once)
    const int nk = (K + TK - 1) / TK;  // k-steps of the image
    const int ns = (nk + 1) / 2;       // stages: two k-steps of A between barriers
    // A loader: a thread takes quad `pq` (k = 32 s + 4 pq .. + 3) of rows lr + 32 j, j < 4 -- eight lanes cover one 128-byte line
    // of a row, a load instruction of the wave eight whole lines, a stage two k-steps between barriers.  (Measured equal to half
    // lines of 16 rows per instruction and one k-step per barrier, 395 against 392 us: what the loads cost is neither their shape
    // nor HBM -- with every workgroup reading the same 128 rows the launch still takes 348 us -- but the registers' fill rate,
    // see the knock-out table in profiles/r06_dense.txt.)
    const int lr = tid >> 3, pq = tid & 7, ph = pq >> 2, kq = pq & 3;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // rows behind M read row M - 1 again (their results are never stored)
    const float *pa[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long long m = m0 + lr + 32 * j < M ? m0 + lr + 32 * j : M - 1;
        pa[j] = A + ((AFX_KO_GEMM & 128) ? (long long)(lr + 32 * j) : m) * lda;  // (128: every workgroup reads the first tile's rows)
    }
    const u32x4 *pb = img + (size_t)blockIdx.y * nk * IMG_U4 + (size_t)wc * 6 * 64 + lane;  // this wave's six fragments: + 64 (3 tj + w)
    const int kLastQuad = ((K + 3) & ~3) - 4;  // quads behind the row's last one (last stage only) read that one again: masked below

    // Loads in flight.  A half-stage (one k-step of MFMAs) issues the bank fragments of the NEXT k-step, the first half of a
    // stage then the four A quads of stage s + 2 (behind the last k-step / stage: the last one again, never used).  Loads return
    // in order, so behind s_waitcnt vmcnt(10) -- six fragments and four quads younger than the fragments about to be used --
    // those fragments have arrived and so has every A quad of stage s + 1, requested a whole stage ago.
    f32x4 ra[2][4];   // [stage parity][row group]
    bf8 rb[2][2][3];  // [k-step parity][tj][word]
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) ra[s][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int w = 0; w < 3; ++w)
#pragma unroll
                for (int e = 0; e < 8; ++e) rb[s][t][w][e] = (__bf16)0.f;
    auto gload_a = [&](int sw, f32x4 (&a4)[4]) {
        const int st = sw < ns ? sw : ns - 1;
        const int k = 32 * st + 4 * pq < kLastQuad ? 32 * st + 4 * pq : kLastQuad;
        if (AFX_KO_GEMM & 1) return;
#pragma unroll
        for (int j = 0; j < 4; ++j) LOAD_B128_SLOT(a4[j], reinterpret_cast<const f32x4 *>(pa[j] + k));
    };
    auto gload_b = [&](int ktw, bf8 (&b)[2][3]) {
        const int kt = ktw < nk ? ktw : nk - 1;
        const u32x4 *s = pb + (size_t)kt * IMG_U4;
        if (AFX_KO_GEMM & 2) return;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int w = 0; w < 3; ++w) LOAD_B128_SLOT(b[t][w], s + 64 * (3 * t + w));
    };
    // this thread's words in a stage of the LDS: k-step half ph, rows lr + 32 j, 8 bytes per plane
    const int aoff = ph * OPER + lr * ROW + 8 * kq;
    // words at k >= K are padding in A (zeros in the image): NaNs there must not reach a product.  Only the last stage has
    // such words; the select is written without a branch so that the split stays in the block of the MFMAs
    bool tm[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) tm[c] = 32 * (ns - 1) + 4 * pq + c >= K;

    // the split of one quad, as numbered pieces (0-7: hi / mid halves of its four values, 8-9: packs + stores); state between them
    unsigned hb[4], mbw[4];
    float r1[4], r2[4];
    auto piece = [&](int jj, const f32x4 &q, bool last, unsigned char *dst) {
        if ((AFX_KO_GEMM & 16) && jj < 8) return;
        if ((AFX_KO_GEMM & 8) && jj >= 8) return;
        if (jj < 8) {
            const int c = jj >> 1;
            if ((jj & 1) == 0) {
                const float x = (last && tm[c]) ? 0.f : q[c];
                split_hi(x, hb[c], r1[c]);
            } else {
                split_mid(r1[c], mbw[c], r2[c]);
            }
        } else if (jj == 8) {
            *reinterpret_cast<u32x2 *>(dst) = u32x2{hb[1] | (hb[0] >> 16), hb[3] | (hb[2] >> 16)};
            *reinterpret_cast<u32x2 *>(dst + PLANE) = u32x2{mbw[1] | (mbw[0] >> 16), mbw[3] | (mbw[2] >> 16)};
        } else {
            *reinterpret_cast<u32x2 *>(dst + 2 * PLANE) = u32x2{pack_hi(f2u(r2[0]), f2u(r2[1])), pack_hi(f2u(r2[2]), f2u(r2[3]))};
        }
    };

    // prologue: fragments of k-step 0, A quads of stages 0 and 1; stage 0 split into the first buffer
    gload_b(0, rb[0]);
    gload_a(0, ra[0]);
    gload_a(1, ra[1]);
    VM_WAIT_N(KO_A);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        PIN(ra[0][j]);
#pragma unroll
        for (int jj = 0; jj < 10; ++jj) piece(jj, ra[0][j], ns == 1, smem + aoff + 32 * j * ROW);
    }
    __syncthreads();
    const int fragOff = (lane & 31) * ROW + 16 * (lane >> 5);
    for (int s0 = 0; s0 < ns; s0 += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int st = s0 + u;
            if (st >= ns) break;
            f32x4(&an)[4] = ra[u ^ 1];  // stage st + 1: split into the other buffer while this one is multiplied
            const bool last = st + 1 == ns - 1;
            unsigned char *nbase = smem + (u ^ 1) * STAGE + aoff;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int kt = 2 * st + h;
                gload_b(kt + 1, rb[h ^ 1]);
                if (h == 0) gload_a(st + 2, ra[u]);  // slot u held stage st, split one stage ago
                VM_WAIT_N(KO_YOUNGER);
                if (h == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) PIN(an[j]);
                }
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int w = 0; w < 3; ++w) PIN(rb[h][t][w]);
                const unsigned char *sa = smem + u * STAGE + h * OPER + (64 * wr) * ROW + fragOff;
                bf8 a[2][3];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int w = 0; w < 3; ++w) {
                        if (AFX_KO_GEMM & 32) asm volatile("" : "=v"(a[t][w]));
                        else a[t][w] = *reinterpret_cast<const bf8 *>(sa + w * PLANE + 32 * t * ROW);
                    }
                __builtin_amdgcn_sched_barrier(0);
                // six terms, smallest first: (a word, b word) = (l, h) (h, l) (m, m) (m, h) (h, m) (h, h); MFMA m = 4 term + tile
#pragma unroll
                for (int m = 0; m < 24; ++m) {
                    const int term = m >> 2, ti = (m >> 1) & 1, tj = m & 1;
                    const int wa = term == 0 ? 2 : (term == 2 || term == 3) ? 1 : 0;
                    const int wb = term == 1 ? 2 : (term == 2 || term == 4) ? 1 : 0;
                    if (AFX_KO_GEMM & 64) asm volatile("" ::"v"(a[ti][wa]), "v"(rb[h][tj][wb]));
                    else acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ti][wa], rb[h][tj][wb], acc[ti][tj], 0, 0, 0);
                    if (m < 20) {  // quads 2 h and 2 h + 1 of the next stage: ten pieces each
                        const int j = 2 * h + (m >= 10);
                        piece(m >= 10 ? m - 10 : m, an[j], last, nbase + 32 * j * ROW);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (!(AFX_KO_GEMM & 4)) __syncthreads();
        }
    }
    // The rings' last loads (behind the last k-step: clamped, never used) are still in flight, and nothing the compiler can see keeps
    // it from handing their registers to the epilogue's address arithmetic AHEAD of this hand-written wait -- met on the device: an
    // epilogue variant whose first instructions were scheduled before the wait returned errors of 0.3-0.6.  The pins behind the wait
    // keep every ring register alive up to it.
    VM_WAIT_N(0);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int j = 0; j < 4; ++j) PIN(ra[s][j]);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int w = 0; w < 3; ++w) PIN(rb[s][t][w]);
    }

#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
            const int j = n0 + 64 * wc + 32 * tj + (lane & 31);
            if (j >= N) continue;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int i = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                const long long m = m0 + 64 * wr + 32 * ti + i;
                if (m < M) {
                    float v = acc[ti][tj][reg];
                    if (post == AFX_MAP_POW) v = powf(v, postArg);
                    C[m * ldc + j] = v;
                }
            }
        }
}

}  // namespace

// contract (the wide branch of afxk_gemm_nt, pre == AFX_MAP_NONE): 16-byte aligned operands with row
// pitches that are multiples of 4 floats; AFX_ERR_UNSUPPORTED otherwise
extern "C" int afxk_gemm_nt128_bf16(const float *A, long long lda, const float *B, int ldb, float *C, long long ldc,
                                    long long M, int N, int K, int post, float postArg, void *stream) {
    if (M <= 0 || N <= 0 || K <= 0) return AFX_OK;
    if (lda % 4 || ldb % 4 || reinterpret_cast<uintptr_t>(A) % 16 || reinterpret_cast<uintptr_t>(B) % 16)
        return AFX_ERR_UNSUPPORTED;
    const long long gm = (M + TM - 1) / TM;
    const int gn = (N + TN - 1) / TN;
    if (gm > 0x7fffffffLL || gn > 65535) return AFX_ERR_UNSUPPORTED;
    AFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_nt128_bf16x3),
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    hipLaunchKernelGGL(k_gemm_nt128_bf16x3, dim3((unsigned)gm, (unsigned)gn), dim3(256), LDS_BYTES, (hipStream_t)stream, A,
                       lda, B, ldb, C, ldc, M, N, K, post, postArg);
    AFX_LAUNCH_CHECK("k_gemm_nt128_bf16x3");
    return AFX_OK;
}

// ---- the prepared-bank form (afx_device.h) ------------------------------------------------------------------------
extern "C" int afxk_gemm_bank_prepare(const float *B, int ldb, int N, int K, void **bankImage, void *stream) {
    *bankImage = nullptr;
    if (!B || N <= 0 || K <= 0 || ldb < K) return AFX_ERR_ARG;
    const int nk = (K + TK - 1) / TK, tiles = (N + TN - 1) / TN;
    void *img = nullptr;
    const int st = afxdev_malloc(&img, (size_t)tiles * nk * IMG_U4 * sizeof(u32x4));
    if (st != AFX_OK) return st;
    hipLaunchKernelGGL(k_bank_split, dim3((unsigned)(tiles * nk)), dim3(256), 0, (hipStream_t)stream, B, ldb, N, K,
                       static_cast<u32x4 *>(img));
    if (hipGetLastError() != hipSuccess) {
        afxdev_free(img);
        afxdev_set_error("k_bank_split: launch failed");
        return AFX_ERR_HIP;
    }
    *bankImage = img;
    return AFX_OK;
}

// C[M, N] = post(A[M, K] . bank[N, K]^T); A: 16-byte aligned rows at a pitch that is a multiple of 4 floats (>= K rounded up
// to 4: the last quad of a row is read whole); AFX_ERR_UNSUPPORTED otherwise (the caller then runs afxk_gemm_nt on the float bank)
extern "C" int afxk_gemm_nt_bank(const float *A, long long lda, const void *bankImage, int N, int K, float *C, long long ldc,
                                 long long M, int post, float postArg, void *stream) {
    if (M <= 0 || N <= 0 || K <= 0) return AFX_OK;
    if (!bankImage || lda % 4 || lda < ((K + 3) & ~3) || reinterpret_cast<uintptr_t>(A) % 16) return AFX_ERR_UNSUPPORTED;
    const long long gm = (M + TM - 1) / TM;
    const int gn = (N + TN - 1) / TN;
    if (gm > 0x7fffffffLL || gn > 65535) return AFX_ERR_UNSUPPORTED;
    AFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_bank_bf16x3), hipFuncAttributeMaxDynamicSharedMemorySize,
                                LDS_BANK_BYTES));
    hipLaunchKernelGGL(k_gemm_bank_bf16x3, dim3((unsigned)gm, (unsigned)gn), dim3(256), LDS_BANK_BYTES, (hipStream_t)stream, A, lda,
                       static_cast<const u32x4 *>(bankImage), C, ldc, M, N, K, post, postArg);
    AFX_LAUNCH_CHECK("k_gemm_bank_bf16x3");
    return AFX_OK;
}
