// VALU issue/execute model of gfx950 for the wave-level FFT kernels: cycles per wave64 instruction
// of v_fma_f32 / v_pk_fma_f32 / v_pk_add_f32 (with op_sel modifiers) / v_mov_b32 / v_mov_b64, alone
// and mixed with ds_read_b64 traffic, at 1..4 waves per SIMD.  Answers: does a packed f32 op cost one
// or two passes, and do LDS reads issue beside VALU work of OTHER waves of the same SIMD?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o tools/micro/valu_rate && tools/micro/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2 __attribute__((ext_vector_type(2)));

template <int MODE, int LDSMIX>
__global__ __launch_bounds__(1024) void k(float *out, int iters, long long *cyc) {
    __shared__ v2 lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = v2{(float)i, 1.f};
    __syncthreads();
    const unsigned base = (unsigned)(size_t)(lds + (threadIdx.x & 63));
    v2 a[8];
    float s[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        a[u] = v2{(float)threadIdx.x * 1e-3f + u, 1.f};
        s[u] = (float)threadIdx.x * 1e-3f + u;
    }
    const v2 b = {0.999f, 1.001f}, c = {1e-3f, -1e-3f};
    const float bs = 0.999f, cs = 1e-3f;
    v2 r[4] = {};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
            if (LDSMIX && (rep & 1) == 0) {
#pragma unroll
                for (int u = 0; u < LDSMIX; ++u)
                    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r[u & 3]) : "v"(base), "n"((u & 7) * 512));
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[u]) : "v"(bs), "v"(cs));
                if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[u]) : "v"(b), "v"(c));
                if (MODE == 2) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "+v"(a[u]) : "v"(c));
                if (MODE == 3) asm volatile("v_mov_b32 %0, %1" : "=v"(s[u]) : "v"(s[(u + 1) & 7]));
                if (MODE == 4) asm volatile("v_mov_b64 %0, %1" : "=v"(a[u]) : "v"(a[(u + 1) & 7]));
                if (MODE == 5) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[u]) : "v"(b));
                if (MODE == 6) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[u]) : "v"(cs));
            }
        }
        if (LDSMIX) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    const long long t1 = __builtin_readcyclecounter();
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += a[u].x + a[u].y + s[u];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += r[u].x;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE, int LDSMIX>
void run(const char *name, int threads) {
    float *d;
    long long *dc, hc = 0;
    hipMalloc(&d, sizeof(float) * 256 * threads);
    hipMalloc(&dc, 8);
    const int iters = 2000;
    k<MODE, LDSMIX><<<256, threads>>>(d, iters, dc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE, LDSMIX><<<256, threads>>>(d, iters, dc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&hc, dc, 8, hipMemcpyDeviceToHost);
    const double instr = (double)iters * 64;                    // VALU instructions per wave
    const double wavesPerSimd = threads / 64.0 / 4.0;
    // s_memtime-style counter (100 MHz constant clock on gfx9): convert through the wall time
    const double clk = 2.4e9;                                   // nominal; relative numbers matter
    const double cycPerInstrPerSimd = ms * 1e-3 * clk / (instr * wavesPerSimd);
    printf("%-34s %4.1f waves/SIMD: %.3f ms  %.2f cyc/VALU-instr/SIMD (wall at 2.4 GHz)  %.2f (s_memtime of wave 0)\n", name,
           wavesPerSimd, ms, cycPerInstrPerSimd, (double)hc / (instr * wavesPerSimd));
    hipFree(d);
    hipFree(dc);
}

int main() {
    for (int t : {256, 512, 768, 1024}) {
        run<0, 0>("v_fma_f32", t);
        run<6, 0>("v_add_f32", t);
        run<1, 0>("v_pk_fma_f32", t);
        run<2, 0>("v_pk_add_f32 op_sel/neg", t);
        run<5, 0>("v_pk_mul_f32", t);
        run<3, 0>("v_mov_b32", t);
        run<4, 0>("v_mov_b64", t);
    }
    for (int t : {256, 768}) {
        run<1, 2>("v_pk_fma_f32 + 2 ds_read_b64/16", t);
        run<1, 4>("v_pk_fma_f32 + 4 ds_read_b64/16", t);
        run<1, 8>("v_pk_fma_f32 + 8 ds_read_b64/16", t);
        run<0, 4>("v_fma_f32 + 4 ds_read_b64/16", t);
    }
    return 0;
}
