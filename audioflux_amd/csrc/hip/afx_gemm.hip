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


}  // namespace

extern "C" int afxk_gemm_nt(const float *A, long long lda, const float *B, int ldb, float *C,
                            long long ldc, long long M, int N, int K, int pre, int post,
                            float postArg, void *stream) {
    if (M <= 0 || N <= 0) return AFX_OK;
    // wide outputs with 16-byte aligned operand rows (the dense filter-bank route pads its scratch spectrum and its
    // bank copy to a pitch of 4 floats): 128 x 128 tiles on the bf16 matrix cores with three-word operands
    // (afx_gemm_bf16.hip).  Round 3, measured (profiles/r03_round_start.txt): the dense gammatone-128 route 5.40 ->
    // 5.04 ms per 934 000 frames against the 128 x 128 v_mfma_f32_32x32x2_f32 kernel of round 2 (removed).
    if (N > 32 && pre == AFX_MAP_NONE) {
        const int st = afxk_gemm_nt128_bf16(A, lda, B, ldb, C, ldc, M, N, K, post, postArg, stream);
        if (st != AFX_ERR_UNSUPPORTED) return st;
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
