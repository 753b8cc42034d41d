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

#include "afx_device.h"
#include "afx_hipcheck.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));

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
