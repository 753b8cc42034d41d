// Calibration of SQ_VALU_MFMA_BUSY_CYCLES (tools/prof_compute.py: mfma_busy): one wave per SIMD issues nothing but
// v_mfma_f32_32x32x16_f16 with four independent accumulators -- the matrix pipe is as busy as it gets.  Run under
//   rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -- ./mfma_busy_cal
// and divide the counter by 1024 SIMDs x (SQ_BUSY_CYCLES / 32).   hipcc --offload-arch=gfx950 -O3 mfma_busy_cal.hip -o mfma_busy_cal
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_mfma_only(float *out, int iters) {
    f32x16 acc[4];
    for (int n = 0; n < 4; ++n)
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    h8 a, b;
    for (int e = 0; e < 8; ++e) a[e] = (_Float16)(0.001f * (threadIdx.x + e)), b[e] = (_Float16)(0.002f * (threadIdx.x - e));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[n], 0, 0, 0);
    }
    float s = 0.f;
    for (int n = 0; n < 4; ++n)
        for (int r = 0; r < 16; ++r) s += acc[n][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main(int argc, char **argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 256;  // workgroups = CUs kept busy
    float *out;
    hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_mfma_only<4>, dim3(wgs), dim3(256), 0, 0, out, 20000);  // one wave per SIMD
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double mf = (double)wgs * 4 * 20000 * 32;
        printf("%d workgroups, one wave per SIMD : %.3f ms = %.2f GHz at 32 cycles per MFMA, %.0f TF/s\n", wgs, ms, 20000.0 * 32 * 32 / (ms * 1e-3) / 1e9, mf * 32768 / (ms * 1e-3) / 1e12);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_mfma_only<8>, dim3(wgs), dim3(512), 0, 0, out, 10000);  // two waves per SIMD
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("two waves per SIMD: %.3f ms, %.0f TF/s\n", ms, mf * 32768 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
