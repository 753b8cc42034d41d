/* k_gemm_nt128_bf16x3 on the emulated library for ThreadSanitizer: its double-buffered LDS stages (store of stage k + 1
 * while stage k is read) must be separated by the kernel's own __syncthreads().  Exit status 0 and no report = pass. */
#include <stdio.h>
#include <stdlib.h>

int afxk_gemm_nt128_bf16(const float *A, long long lda, const float *B, int ldb, float *C, long long ldc, long long M, int N, int K,
                         int post, float postArg, void *stream);

int main(void) {
    const int M = 200, N = 130, K = 70, ld = 72;
    float *A = (float *)aligned_alloc(64, sizeof(float) * M * ld), *B = (float *)aligned_alloc(64, sizeof(float) * N * ld),
          *C = (float *)aligned_alloc(64, sizeof(float) * M * 136);
    if (!A || !B || !C) return 2;
    for (int i = 0; i < M * ld; i++) A[i] = (float)((i * 2654435761u) % 1000) * 1e-3f;
    for (int i = 0; i < N * ld; i++) B[i] = (float)((i * 40503u) % 1000) * 1e-3f;
    if (afxk_gemm_nt128_bf16(A, ld, B, ld, C, 136, M, N, K, 0, 0.f, NULL)) return 1;
    double s = 0;
    for (int m = 0; m < M; m++)
        for (int n = 0; n < N; n++) s += C[m * 136 + n];
    printf("sum %.3f\nOK\n", s);
    free(A), free(B), free(C);
    return 0;
}
