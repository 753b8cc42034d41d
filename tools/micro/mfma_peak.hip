// Achievable v_mfma_f32_32x32x2_f32 rate on this part: independent accumulators, operands in
// registers, no memory traffic.  Build: hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n)
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[n], 0, 0, 0);
    }
    float s = 0.f;
    for (int n = 0; n < NACC; ++n)
        for (int r = 0; r < 16; ++r) s += acc[n][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// operands streamed from LDS (random data), two reads per MFMA, as in k_cqt_octave_mfma_w
template <int NACC, bool RANDOM>
__global__ __launch_bounds__(256) void kl(float *out, int iters, unsigned seed) {
    __shared__ float la[8192], lb[8192];
    unsigned h = seed + threadIdx.x * 2654435761u;
    for (int e = threadIdx.x; e < 8192; e += blockDim.x) {
        h = h * 1664525u + 1013904223u;
        la[e] = RANDOM ? (float)(h >> 8) * (1.f / 16777216.f) - 0.5f : 1.f;
        h = h * 1664525u + 1013904223u;
        lb[e] = RANDOM ? (float)(h >> 8) * (1.f / 16777216.f) - 0.5f : 2.f;
    }
    __syncthreads();
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n)
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
        const int base = (it & 7) * 1024 + lane;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const float a = la[base + 64 * u], b = lb[base + 64 * u];
            acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u % NACC], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int n = 0; n < NACC; ++n)
        for (int r = 0; r < 16; ++r) s += acc[n][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC, bool RANDOM>
void runl(const char *name, int wgs, int iters) {
    float *d;
    hipMalloc(&d, sizeof(float) * wgs * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    kl<NACC, RANDOM><<<wgs, 256>>>(d, iters, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kl<NACC, RANDOM><<<wgs, 256>>>(d, iters, 1u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)wgs * 4 * iters * 16.0 * 4096.0;
    printf("%s: wgs %d, NACC %d, random %d: %.3f ms, %.1f TFLOP/s\n", name, wgs, NACC, (int)RANDOM, ms,
           flops / ms / 1e9);
    hipFree(d);
}
template <int NACC>
void run(const char *name, int wgs, int threads, int iters) {
    float *d;
    hipMalloc(&d, sizeof(float) * wgs * threads);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<NACC><<<wgs, threads>>>(d, iters, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC><<<wgs, threads>>>(d, iters, 1.f, 2.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)wgs * (threads / 64) * iters * 8.0 * NACC * 4096.0;
    printf("%s: wgs %d x %d threads, NACC %d, %d iters: %.3f ms, %.1f TFLOP/s\n", name, wgs, threads, NACC,
           iters, ms, flops / ms / 1e9);
    hipFree(d);
}
int main() {
    run<1>("dependent chain, 1 wave/SIMD", 256, 256, 4000);
    run<2>("2 accumulators,  1 wave/SIMD", 256, 256, 2000);
    run<4>("4 accumulators,  1 wave/SIMD", 256, 256, 1000);
    run<1>("dependent chain, 2 waves/SIMD", 512, 256, 4000);
    run<2>("2 accumulators,  2 waves/SIMD", 512, 256, 2000);
    run<2>("2 accumulators,  4 waves/SIMD", 1024, 256, 2000);
    run<4>("4 accumulators, long run      ", 1024, 256, 20000);
    runl<2, false>("LDS-fed, constant data, 1 wave/SIMD", 256, 2000);
    runl<2, true>("LDS-fed, random data,   1 wave/SIMD", 256, 2000);
    runl<2, false>("LDS-fed, constant data, 2 waves/SIMD", 512, 2000);
    runl<2, true>("LDS-fed, random data,   2 waves/SIMD", 512, 2000);
    runl<2, true>("LDS-fed, random data,   2 waves/SIMD long", 512, 40000);
    return 0;
}
