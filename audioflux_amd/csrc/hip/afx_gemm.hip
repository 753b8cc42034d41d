// afx_gemm.hip -- C[M,N] = post( pre(A)[M,K] . B[N,K]^T ) on the f32 matrix
// cores (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate, bitwise an fmaf
// chain in k order -- cdna_hip_programming.md section 3).
//
// This is the size-generic filter-bank apply ("K3" of SURVEY.md 2b) and, with
// pre = LOG10 / CBRT and B = the orthonormal DCT-II matrix, the cepstral
// transform ("K6").  It replaces the reference's naive A.B^T loops:
//   __mdot1   src/vector/flux_vector.c:55-86   (double accumulate on CPU)
//   __mcdot1  src/vector/flux_complex.c:53-87  (two real products: the bank's
//             imaginary plane is all zero, bft_algorithm.c:347,481-485)
//   xxccObj_xxcc rectify+DCT, src/feature/xxcc_algorithm.c:124-155
//
// Tiling: 64x64 output tile per 256-thread workgroup (4 waves in a 2x2
// arrangement, one 32x32 MFMA accumulator each), K stepped by 32 through LDS.
// A rows have an odd pitch in the hot case (K = 1025) so the global loads are
// scalar-per-lane but lane-contiguous along k (coalesced); the LDS tiles are
// pitched 33 floats so the MFMA operand reads (32 lanes = 32 rows, same k) hit
// 32 distinct banks.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "afx_device.h"
#include "afx_hipcheck.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 64, BN = 64, BK = 32, PITCH = BK + 1;

__device__ __forceinline__ float map_pre(float v, int pre) {
    if (pre == AFX_MAP_LOG10) {
        // xxcc_algorithm.c:130-139
        if (v < 1e-8f) v = 1e-8f;
        return log10f(v);
    }
    if (pre == AFX_MAP_CBRT) {
        // xxcc_algorithm.c:124-128: powf(x, 1.0/3) with the exponent rounded to float
        return powf(v, (float)(1.0 / 3));
    }
    return v;
}

__global__ __launch_bounds__(256) void k_gemm_nt(const float *__restrict__ A, long long lda,
                                                 const float *__restrict__ B, int ldb,
                                                 float *__restrict__ C, long long ldc,
                                                 long long M, int N, int K, int pre, int post,
                                                 float postArg) {
    __shared__ float As[BM * PITCH];
    __shared__ float Bs[BN * PITCH];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const long long m0 = (long long)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;

    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    // cooperative tile load: 256 threads, thread -> (row = tid/32 + 8*p, k = tid%32)
    const int lk = tid & 31, lr = tid >> 5;

    for (int k0 = 0; k0 < K; k0 += BK) {
        const int k = k0 + lk;
#pragma unroll
        for (int p = 0; p < BM / 8; ++p) {
            const int row = lr + 8 * p;
            const long long m = m0 + row;
            float v = 0.f;
            if (m < M && k < K) v = map_pre(A[m * lda + k], pre);
            As[row * PITCH + lk] = v;
        }
#pragma unroll
        for (int p = 0; p < BN / 8; ++p) {
            const int row = lr + 8 * p;
            const int n = n0 + row;
            float v = 0.f;
            if (n < N && k < K) v = B[(long long)n * ldb + k];
            Bs[row * PITCH + lk] = v;
        }
        __syncthreads();

        const float *ap = &As[(wr * 32 + (lane & 31)) * PITCH + (lane >> 5)];
        const float *bp = &Bs[(wc * 32 + (lane & 31)) * PITCH + (lane >> 5)];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            // A operand: A[i = lane&31][k = lane>>5]; B operand: B[k = lane>>5][j = lane&31]
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[kk], bp[kk], acc, 0, 0, 0);
        }
        __syncthreads();
    }

    // C/D layout of the 32x32 MFMA: col j = lane&31, row i = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const int j = n0 + wc * 32 + (lane & 31);
    if (j < N) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int i = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
            const long long m = m0 + wr * 32 + i;
            if (m < M) {
                float v = acc[reg];
                if (post == AFX_MAP_POW) v = powf(v, postArg);
                C[m * ldc + j] = v;
            }
        }
    }
}


// ---- 128 x 128 tile kernel for the dense filter-bank shape -----------------------------------
// C[M, N <= 128 per column block] = A[M, K] . B[N, K]^T with M = frames (10^5..10^6), K = bins
// (1025), N = bands: A is streamed once, every workgroup walks the whole B (L2-resident).
//   * 256 threads = 4 waves in a 2 x 2 arrangement, each wave owns a 64 x 64 block = 2 x 2
//     accumulators of v_mfma_f32_32x32x2_f32 (64 VGPRs): every LDS operand read feeds two MFMAs
//   * K stepped by 16 through DOUBLE-BUFFERED LDS tiles stored k-major ([k][row], pitch 132): the
//     MFMA operand read of a half-wave is 32 consecutive words; the next tile's global loads
//     (dwordx4 along k: both operands have 16-byte aligned rows, pitch % 4 == 0) are issued before
//     the current tile's 32 MFMAs per wave and land in registers under them; one barrier per tile
//   * the tail of K (1025 = 64 x 16 + 1) is masked element-wise: the pad words of the scratch rows
//     are not initialised
typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifndef AFX_HOST_EMULATION
#define GR32(dst, addr, off) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#else  // tests/emu (the kernel compiled for the host): the same read through the pointer the address came from
#define GR32(dst, addr, off) __builtin_memcpy(&(dst), reinterpret_cast<const char *>(addr##_p) + (off), 4)
#endif
constexpr int TM = 128, TN = 128, TK = 16, LP = TM + 4;
static_assert(TK == 16, "the k-step pipeline below is written out for 8 steps");

__global__ __launch_bounds__(256, 2) void k_gemm_nt128(const float *__restrict__ A, long long lda,
                                                       const float *__restrict__ B, int ldb,
                                                       float *__restrict__ C, long long ldc, long long M, int N,
                                                       int K, int post, float postArg) {
    __shared__ float As[2][TK * LP];
    __shared__ float Bs[2][TK * LP];
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
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                As[buf][(4 * kq + c) * LP + lrow + 64 * p] = ra[p][c];
                Bs[buf][(4 * kq + c) * LP + lrow + 64 * p] = rb[p][c];
            }
    };

    const int nk = (K + TK - 1) / TK;
    f32x4 ra[2], rb[2];
    gload(0, ra, rb);
    sstore(0, ra, rb);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * TK, ra, rb);
        const float *ap = &As[buf][(lane >> 5) * LP + 64 * wr + (lane & 31)];
        const float *bp = &Bs[buf][(lane >> 5) * LP + 64 * wc + (lane & 31)];
        // MFMA operands by hand-issued ds_read_b32 with immediate offsets, three k-steps ahead of
        // the MFMAs that use them (the compiler sinks plain loads next to their use and then waits a
        // full LDS round trip in front of every group of four MFMAs); partial waits: LDS returns in order
        float av[4][2], bv[4][2];
        const unsigned aa = (unsigned)(size_t)ap, ba = (unsigned)(size_t)bp;
#ifdef AFX_HOST_EMULATION
        const float *aa_p = ap, *ba_p = bp;
        (void)aa, (void)ba;
#endif
#define AFX_GEMM_REQ(S)                                   \
    do {                                                  \
        GR32(av[(S) & 3][0], aa, 8 * (S) * LP);           \
        GR32(av[(S) & 3][1], aa, 8 * (S) * LP + 128);     \
        GR32(bv[(S) & 3][0], ba, 8 * (S) * LP);           \
        GR32(bv[(S) & 3][1], ba, 8 * (S) * LP + 128);     \
    } while (0)
#ifndef AFX_HOST_EMULATION
#define AFX_GEMM_WAIT(n) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory")
#define AFX_GEMM_PIN(a0, a1, b0, b1) asm volatile("" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1))
#else
#define AFX_GEMM_WAIT(n) ((void)0)
#define AFX_GEMM_PIN(a0, a1, b0, b1) ((void)0)
#endif
#define AFX_GEMM_STEP(S, WAITN)                                                                          \
    do {                                                                                                 \
        if ((S) + 3 < TK / 2) AFX_GEMM_REQ(((S) + 3 < TK / 2 ? (S) + 3 : 0));                            \
        AFX_GEMM_WAIT(WAITN);                                                                            \
        float a0 = av[(S) & 3][0], a1 = av[(S) & 3][1], b0 = bv[(S) & 3][0], b1 = bv[(S) & 3][1];        \
        AFX_GEMM_PIN(a0, a1, b0, b1);                                                                    \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);                    \
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);                    \
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);                    \
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);                    \
    } while (0)
        AFX_GEMM_REQ(0);
        AFX_GEMM_REQ(1);
        AFX_GEMM_REQ(2);
        // wait count = 4 x the k-steps still in flight behind the one being consumed
        AFX_GEMM_STEP(0, 12);
        AFX_GEMM_STEP(1, 12);
        AFX_GEMM_STEP(2, 12);
        AFX_GEMM_STEP(3, 12);
        AFX_GEMM_STEP(4, 12);
        AFX_GEMM_STEP(5, 8);
        AFX_GEMM_STEP(6, 4);
        AFX_GEMM_STEP(7, 0);
#undef AFX_GEMM_REQ
#undef AFX_GEMM_STEP
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

extern "C" int afxk_gemm_nt(const float *A, long long lda, const float *B, int ldb, float *C,
                            long long ldc, long long M, int N, int K, int pre, int post,
                            float postArg, void *stream) {
    if (M <= 0 || N <= 0) return AFX_OK;
    // wide outputs with 16-byte aligned operand rows (the dense filter-bank route pads its scratch
    // spectrum and its bank copy to a pitch of 4 floats): the 128 x 128 double-buffered kernel
    if (N > 32 && pre == AFX_MAP_NONE && lda % 4 == 0 && ldb % 4 == 0 &&
        reinterpret_cast<uintptr_t>(A) % 16 == 0 && reinterpret_cast<uintptr_t>(B) % 16 == 0 && !getenv("AFX_GEMM_V1")) {
        if (getenv("AFX_GEMM_BF16")) {  // three-bf16-word operands on the bf16 matrix cores (afx_gemm_bf16.hip; off by default)
            const int st = afxk_gemm_nt128_bf16(A, lda, B, ldb, C, ldc, M, N, K, post, postArg, stream);
            if (st != AFX_ERR_UNSUPPORTED) return st;
        }
        const long long g = (M + TM - 1) / TM;
        const int gnn = (N + TN - 1) / TN;
        if (g <= 0x7fffffffLL && gnn <= 65535) {
            hipLaunchKernelGGL(k_gemm_nt128, dim3((unsigned)g, (unsigned)gnn), dim3(256), 0, (hipStream_t)stream, A, lda,
                               B, ldb, C, ldc, M, N, K, post, postArg);
            AFX_LAUNCH_CHECK("k_gemm_nt128");
            return AFX_OK;
        }
    }
    const long long gm = (M + BM - 1) / BM;
    const int gn = (N + BN - 1) / BN;
    if (gm > 0x7fffffffLL || gn > 65535) {
        afxdev_set_error("gemm: grid %lld x %d too large", gm, gn);
        return AFX_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(k_gemm_nt, dim3((unsigned)gm, (unsigned)gn), dim3(256), 0,
                       (hipStream_t)stream, A, lda, B, ldb, C, ldc, M, N, K, pre, post, postArg);
    AFX_LAUNCH_CHECK("k_gemm_nt");
    return AFX_OK;
}
