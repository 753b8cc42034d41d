// ds_read_b64 x2 versus ds_read2_b64 (what the load/store optimizer makes of two float2 reads
// from one base): LDS read rate with 4 / 12 / 16 waves per CU, conflict-free lane-linear addresses.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(1024) void k(float *out, int iters) {
    __shared__ v2 lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = v2{(float)i, 1.f};
    __syncthreads();
    const unsigned base = (unsigned)(size_t)(lds + (threadIdx.x & 63));
    v2 acc = {0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        v2 r[16];
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 16; ++u) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r[u]) : "v"(base), "n"(u * 512));
        } else {
#pragma unroll
            for (int u = 0; u < 16; u += 2) {
                typedef float v4 __attribute__((ext_vector_type(4)));
                v4 t;
                asm volatile("ds_read2st64_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(t) : "v"(base), "n"(u), "n"(u + 1));
                r[u] = v2{t.x, t.y};
                r[u + 1] = v2{t.z, t.w};
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
        for (int u = 0; u < 16; ++u) asm volatile("" : "+v"(r[u]));
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += r[u];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y;
}
template <int MODE>
void run(const char *name, int threads) {
    float *d;
    hipMalloc(&d, sizeof(float) * 256 * threads);
    const int iters = 4000;
    k<MODE><<<256, threads>>>(d, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE><<<256, threads>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 256.0 * threads * iters * 16 * 8;
    printf("%s, %2d waves/CU: %.3f ms, %.1f TB/s aggregate, %.0f B/clk/CU at 2.4 GHz\n", name, threads / 64, ms,
           bytes / ms / 1e9, bytes / ms / 1e9 * 1e12 / 256 / 2.4e9 / 1e3 * 1e0);
    hipFree(d);
}
int main() {
    for (int t : {256, 768, 1024}) {
        run<0>("16 x ds_read_b64     ", t);
        run<1>(" 8 x ds_read2st64_b64", t);
    }
    return 0;
}
