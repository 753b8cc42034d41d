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
// work) before it is converted: every k-step waits for memory.  Here
//   * the bank is split into its three bf16 word planes ONCE per object (k_bank_split -> "bank image": per column tile and
//     k-step the 3 x 128 x 16 words in the order the staging threads copy them, zero rows / zero words behind N and K), so
//     staging B is three 16-byte loads (L2-resident: 787 KB for 128 x 1025) and three ds_write_b128 per thread and k-step;
//   * the float32 rows of A are requested FOUR k-steps ahead into a ring of registers (the bank image likewise) and split
//     with integer arithmetic on the bit patterns (split3_words: add / mask / subtract per word, half-word packs): exact,
//     8 + 8 + 8 bits cover float32's 24, so the six-term product below is the same expansion as the converting split's;
//   * tile, fragments, MFMA order and epilogue as in k_gemm_nt128_bf16x3.
constexpr int RING = 4;                       // k-steps in flight in registers
constexpr int IMG_U4 = 3 * TM * 2;            // u32x4 per (column tile, k-step) of the bank image: 3 planes x 128 rows x 32 bytes

__device__ __forceinline__ unsigned f2u(float x) { return __float_as_uint(x); }
__device__ __forceinline__ float u2f(unsigned x) { return __uint_as_float(x); }
// upper half-words of (x0, x1) as (low, high) half of one word
__device__ __forceinline__ unsigned pack_hi(unsigned x0, unsigned x1) { return (x1 & 0xffff0000u) | (x0 >> 16); }

// four float32 -> 4 x 3 bf16 words.  hi and mid are ROUNDED (half up in magnitude: one integer add before the mask), lo takes
// what is left: |r1| <= 2^-9 |x| has at most 16 significant bits, |r2| <= 2^-9 |r1| at most 7 -- the split is exact.  (Plain
// truncation is exact as well and two instructions shorter, but leaves every word with the sign of x: the small terms of the
// product, all of one sign, are then rounded away one by one against the large accumulator -- a bias of -2.4e-6 of the
// result at K = 1025 under tests/emu, against 5e-7 for words of either sign.)
__device__ __forceinline__ void split3_words(const f32x4 v, u32x2 &h, u32x2 &m, u32x2 &l) {
    unsigned hb[4], mb[4];
    float r2[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        hb[c] = (f2u(v[c]) + 0x8000u) & 0xffff0000u;
        const float r1 = v[c] - u2f(hb[c]);
        mb[c] = (f2u(r1) + 0x8000u) & 0xffff0000u;
        r2[c] = r1 - u2f(mb[c]);
    }
    h = u32x2{hb[1] | (hb[0] >> 16), hb[3] | (hb[2] >> 16)};
    m = u32x2{mb[1] | (mb[0] >> 16), mb[3] | (mb[2] >> 16)};
    l = u32x2{pack_hi(f2u(r2[0]), f2u(r2[1])), pack_hi(f2u(r2[2]), f2u(r2[3]))};
}

// bank [N, K] (row pitch ldb floats) -> image [column tile][k-step][plane][128 rows][16 words]; one thread per 8 words of a row
__global__ __launch_bounds__(256) void k_bank_split(const float *__restrict__ B, int ldb, int N, int K, u32x4 *__restrict__ img) {
    const int nk = (K + TK - 1) / TK;
    const int tile = blockIdx.x / nk, kt = blockIdx.x - tile * nk;
    const int row = threadIdx.x >> 1, half = threadIdx.x & 1;
    const int n = tile * TN + row, k0 = kt * TK + 8 * half;
    f32x4 v[2];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int k = k0 + 4 * q + c;
            v[q][c] = (n < N && k < K) ? B[(long long)n * ldb + k] : 0.f;
        }
    u32x2 h0, m0, l0, h1, m1, l1;
    split3_words(v[0], h0, m0, l0);
    split3_words(v[1], h1, m1, l1);
    u32x4 *dst = img + (size_t)blockIdx.x * IMG_U4 + threadIdx.x;
    dst[0] = u32x4{h0.x, h0.y, h1.x, h1.y};
    dst[256] = u32x4{m0.x, m0.y, m1.x, m1.y};
    dst[512] = u32x4{l0.x, l0.y, l1.x, l1.y};
}

// (two workgroups per CU by their LDS: two waves per SIMD, 256 registers each -- the ring lives in registers)
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
    const int lrow = tid >> 2, kq = tid & 3;  // A loader: rows lrow, lrow + 64; k = k0 + 4 kq .. + 3
    const int nk = (K + TK - 1) / TK;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // rows behind M read row M - 1 again (their results are never stored)
    const long long ma = m0 + lrow < M ? m0 + lrow : M - 1, mb = m0 + lrow + 64 < M ? m0 + lrow + 64 : M - 1;
    const float *pa0 = A + ma * lda, *pa1 = A + mb * lda;
    const u32x4 *pb = img + (size_t)blockIdx.y * nk * IMG_U4 + tid;
    const int kLastQuad = ((K + 3) & ~3) - 4;  // quads behind the row's last one (last k-step only) read that one again: masked below

    // The ring's loads are issued by hand (afx_asm.h) and waited for by count: the compiler's own s_waitcnt placement drains
    // every load older than the current iteration's at the loop header (vmcnt(4) in the first of the four unrolled bodies: a
    // prefetch distance of ONE k-step, 35 % -> 44 % matrix-pipe utilisation only).  Every iteration issues exactly five loads
    // (behind the last k-step: the last k-step again, never staged), so "all but the youngest 15" = the group of k-step kt + 1.
    f32x4 ra[RING][2];
    u32x4 rb[RING][3];
#pragma unroll
    for (int s = 0; s < RING; ++s) {
        ra[s][0] = ra[s][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        rb[s][0] = rb[s][1] = rb[s][2] = u32x4{0u, 0u, 0u, 0u};
    }
    auto gload = [&](int ktw, f32x4 (&a2)[2], u32x4 (&b3)[3]) {
        const int kt = ktw < nk ? ktw : nk - 1;
        const int k = kt * TK + 4 * kq < kLastQuad ? kt * TK + 4 * kq : kLastQuad;
        const u32x4 *s = pb + (size_t)kt * IMG_U4;
        LOAD_B128_SLOT(a2[0], reinterpret_cast<const f32x4 *>(pa0 + k));
        LOAD_B128_SLOT(a2[1], reinterpret_cast<const f32x4 *>(pa1 + k));
        LOAD_B128_SLOT(b3[0], s);
        LOAD_B128_SLOT(b3[1], s + 256);
        LOAD_B128_SLOT(b3[2], s + 512);
    };
    auto arrived = [&](f32x4 (&a2)[2], u32x4 (&b3)[3]) {  // behind the wait: the slot's first uses stay behind it
        PIN(a2[0]);
        PIN(a2[1]);
        PIN(b3[0]);
        PIN(b3[1]);
        PIN(b3[2]);
    };
    const int aoff = lrow * ROW + 8 * kq;                  // A plane rows lrow, lrow + 64: 8 bytes per thread and plane
    const int boff = (tid >> 1) * ROW + 16 * (tid & 1);    // bank plane row tid >> 1: 16 bytes per thread and plane
    // words at k >= K are padding in A (zeros in the image): NaNs there must not reach a product.  Only the last k-step has
    // such words; the select is written without a branch so that the staging arithmetic stays in the block of the MFMAs
    bool tm[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) tm[c] = (nk - 1) * TK + 4 * kq + c >= K;
    auto sstore = [&](int kt, int buf, f32x4 (&a2)[2], const u32x4 (&b3)[3]) {
        unsigned char *base = smem + buf * STAGE;
        const bool last = kt == nk - 1;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
            for (int c = 0; c < 4; ++c) a2[p][c] = (last && tm[c]) ? 0.f : a2[p][c];
            u32x2 h, m, l;
            split3_words(a2[p], h, m, l);
            *reinterpret_cast<u32x2 *>(base + aoff + 64 * p * ROW) = h;
            *reinterpret_cast<u32x2 *>(base + PLANE + aoff + 64 * p * ROW) = m;
            *reinterpret_cast<u32x2 *>(base + 2 * PLANE + aoff + 64 * p * ROW) = l;
        }
#pragma unroll
        for (int w = 0; w < 3; ++w) *reinterpret_cast<u32x4 *>(base + OPER + w * PLANE + boff) = b3[w];
    };

#pragma unroll
    for (int s = 0; s < RING; ++s) gload(s, ra[s], rb[s]);
    VM_WAIT_N(15);
    arrived(ra[0], rb[0]);
    sstore(0, 0, ra[0], rb[0]);
    __syncthreads();
    const int fragOff = (lane & 31) * ROW + 16 * (lane >> 5);
    for (int kt0 = 0; kt0 < nk; kt0 += RING) {
#pragma unroll
        for (int u = 0; u < RING; ++u) {
            const int kt = kt0 + u;
            if (kt >= nk) break;
            const int buf = u & 1;  // (RING is even: kt & 1 == u & 1)
            // slot u held k-step kt, staged one iteration ago: k-step kt + RING takes it; then 15 loads are younger than
            // the group of k-step kt + 1, requested three iterations ago
            gload(kt + RING, ra[u], rb[u]);
            VM_WAIT_N(15);
            arrived(ra[(u + 1) % RING], rb[(u + 1) % RING]);
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
            // k-step kt + 1 -> the other buffer (behind the last k-step: stale registers into a buffer nobody reads).  One MFMA
            // (32 cycles of the matrix pipe) covers three vector instructions of the split
            sstore(kt + 1, buf ^ 1, ra[(u + 1) % RING], rb[(u + 1) % RING]);
#pragma unroll
            for (int i = 0; i < 24; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            }
            __syncthreads();
        }
    }

    VM_WAIT_N(0);  // the ring's last loads land in registers the epilogue may use
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
                                LDS_BYTES));
    hipLaunchKernelGGL(k_gemm_bank_bf16x3, dim3((unsigned)gm, (unsigned)gn), dim3(256), LDS_BYTES, (hipStream_t)stream, A, lda,
                       static_cast<const u32x4 *>(bankImage), C, ldc, M, N, K, post, postArg);
    AFX_LAUNCH_CHECK("k_gemm_bank_bf16x3");
    return AFX_OK;
}
