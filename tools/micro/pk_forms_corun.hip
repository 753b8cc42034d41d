// Which modifier forms of the packed-f32 VALU instructions stay exact beside a kernel that streams v_mfma + ds_read_b128?
// One victim kernel per (instruction, op_sel / neg form) -- the forms found in the shipped kernels' ISA, hand-written and
// compiler-generated -- each compared bitwise with its solo run (see tools/micro/mfma_corun.hip, profiles/r03_pk_add_opsel.txt).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/pk_forms_corun.hip -o tools/micro/pk_forms_corun
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <vector>

typedef float v2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

template <int F>
__device__ __forceinline__ v2 step(v2 z, v2 w, v2 c) {
    v2 r = z;
    if (F == 0) { v2 u = z * 0.5f; asm("v_pk_add_f32 %0, %1, %2 " : "=v"(r) : "v"(u), "v"(w)); }
    if (F == 1) { v2 u = z * 0.5f; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(u), "v"(w)); }
    if (F == 2) { v2 u = z * 0.5f; asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(u), "v"(w)); }
    if (F == 3) { v2 u = z * 0.5f; asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(u), "v"(w)); }
    if (F == 4) { v2 u = z * 0.5f; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(u), "v"(w)); }
    if (F == 5) { v2 u = z * 0.5f; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(u), "v"(w)); }
    if (F == 6) { v2 u = z * 0.5f; asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(u), "v"(w)); }
    if (F == 7) { v2 u = z * 0.5f; asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(u), "v"(w)); }
    if (F == 8) { v2 u = z * 0.5f; asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(u), "v"(w)); }
    if (F == 9) { v2 u = z * 0.5f; asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(r) : "v"(u), "v"(w)); }
    if (F == 10) { asm("v_pk_mul_f32 %0, %1, %2 " : "=v"(r) : "v"(z), "v"(w)); r = r + c; }
    if (F == 11) { asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(z), "v"(w)); r = r + c; }
    if (F == 12) { asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(z), "v"(w)); r = r + c; }
    if (F == 13) { asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(z), "v"(w)); r = r + c; }
    if (F == 14) { asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(z), "v"(w)); r = r + c; }
    if (F == 15) { asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(r) : "v"(z), "v"(w)); r = r + c; }
    if (F == 16) { asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(z), "v"(w)); r = r + c; }
    if (F == 17) { asm("v_pk_fma_f32 %0, %1, %2, %3 " : "=v"(r) : "v"(z), "v"(w), "v"(c)); }
    if (F == 18) { asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(z), "v"(w), "v"(c)); }
    if (F == 19) { asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(z), "v"(w), "v"(c)); }
    if (F == 20) { asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(z), "v"(w), "v"(c)); }
    if (F == 21) { asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,1,0]" : "=v"(r) : "v"(z), "v"(w), "v"(c)); }
    if (F == 22) { asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(z), "v"(w), "v"(c)); }
    if (F == 23) { asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "=v"(r) : "v"(z), "v"(w), "v"(c)); }
    if (F == 24) { asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(z), "v"(w), "v"(c)); }
    if (F == 25) { asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(z), "v"(w), "v"(c)); }
    if (F == 26) { asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(r) : "v"(z), "v"(w)); r = r * 0.75f + c; }  // (z.hi, w.lo): the compiler's pair shuffle
    if (F == 27) { asm("v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(z), "v"(w)); r = r * 0.75f + c; }  // (z.lo, w.hi)
    return r;
}

template <int F>
__global__ __launch_bounds__(256) void v_form(float2 *out, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    v2 z[8];
    for (int j = 0; j < 8; ++j) z[j] = v2{0.9f + 1e-4f * (float)((t + 13 * j) % 1013), 0.1f + 1e-4f * (float)((t * 5 + j) % 499)};
    v2 w = {0.61f + 1e-3f * (float)(t & 31), 0.47f}, c = {0.031f, 0.017f};
    asm volatile("" : "+v"(w), "+v"(c));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = step<F>(z[j], w, c);
    }
    v2 s = {0.f, 0.f};
    for (int j = 0; j < 8; ++j) s = v2{s.x + z[j].x * (j + 1), s.y + z[j].y * (j + 1)};
    out[t] = make_float2(s.x, s.y);
}

__global__ __launch_bounds__(256) void p_lds_mfma(float *sink, int iters, int ldsBytes) {
    extern __shared__ unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    uint4 *v = reinterpret_cast<uint4 *>(smem);
    const int n16 = ldsBytes / 16;
    for (int e = tid; e < n16; e += 256) v[e] = make_uint4(0x2e662e66u, 0x2e662e66u, 0x2e662e66u, 0x2e662e66u);
    __syncthreads();
    f32x16 acc[4];
    for (int q = 0; q < 4; ++q)
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    const int span = n16 - 6 * 64 - 64;
    int base = lane;
    for (int it = 0; it < iters; ++it) {
        uint4 o[6];
#pragma unroll
        for (int u = 0; u < 6; ++u) o[u] = v[base + 64 * u];
        base += 64;
        if (base >= span) base = lane;
        const h8 a0 = __builtin_bit_cast(h8, o[0]), a1 = __builtin_bit_cast(h8, o[1]), a2 = __builtin_bit_cast(h8, o[2]),
                 a3 = __builtin_bit_cast(h8, o[3]), b0 = __builtin_bit_cast(h8, o[4]), b1 = __builtin_bit_cast(h8, o[5]);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[3], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b0, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a3, b0, acc[3], 0, 0, 0);
    }
    float s = 0.f;
    for (int q = 0; q < 4; ++q)
        for (int r = 0; r < 16; ++r) s += acc[q][r];
    if (sink) sink[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    const int VB = 4096;
    const size_t vn = (size_t)VB * 256;
    float2 *dOut;
    float *dSink;
    CK(hipMalloc(&dOut, vn * sizeof(float2)));
    CK(hipMalloc(&dSink, (size_t)2048 * 256 * sizeof(float)));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(p_lds_mfma), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipStream_t sv, sp;
    CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sp, hipStreamNonBlocking));
    std::vector<float2> ref(vn), got(vn);
    struct Victim { const char *name; void (*launch)(hipStream_t, float2 *, int); };
#define V(F, NAME) {NAME, [](hipStream_t s, float2 *o, int vb) { hipLaunchKernelGGL(v_form<F>, dim3(vb), dim3(256), 0, s, o, 3000); }}
    const Victim victims[] = {
        V(0, "v_pk_add_f32  (plain)"),
        V(1, "v_pk_add_f32 neg_lo:[0,1] neg_hi:[0,1] (sub)"),
        V(2, "v_pk_add_f32 op_sel_hi:[1,0] (src1 lo broadcast)"),
        V(3, "v_pk_add_f32 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1] (src1 lo broadcast, neg)"),
        V(4, "v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,1] (src1 hi broadcast)"),
        V(5, "v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0] (src1 swap)"),
        V(6, "v_pk_add_f32 op_sel:[1,0] op_sel_hi:[0,1] (src0 swap)"),
        V(7, "v_pk_add_f32 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1] (src0 swap, neg)"),
        V(8, "v_pk_add_f32 op_sel_hi:[0,1] (src0 lo broadcast)"),
        V(9, "v_pk_add_f32 op_sel:[1,0] op_sel_hi:[1,1] (src0 hi broadcast)"),
        V(10, "v_pk_mul_f32  (plain)"),
        V(11, "v_pk_mul_f32 op_sel:[0,0] op_sel_hi:[0,1] (src0 lo broadcast)"),
        V(12, "v_pk_mul_f32 op_sel_hi:[1,0] (src1 lo broadcast)"),
        V(13, "v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0] (src1 swap)"),
        V(14, "v_pk_mul_f32 op_sel:[1,0] op_sel_hi:[0,1] (src0 swap)"),
        V(15, "v_pk_mul_f32 op_sel:[1,0] op_sel_hi:[1,1] (src0 hi broadcast)"),
        V(16, "v_pk_mul_f32 op_sel:[1,1] op_sel_hi:[1,0] (src0 hi broadcast, src1 swap)"),
        V(17, "v_pk_fma_f32  (plain)"),
        V(18, "v_pk_fma_f32 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0] (cmul second half)"),
        V(19, "v_pk_fma_f32 op_sel:[0,0,0] op_sel_hi:[0,1,1] (src0 lo broadcast)"),
        V(20, "v_pk_fma_f32 op_sel_hi:[1,0,1] (src1 lo broadcast)"),
        V(21, "v_pk_fma_f32 op_sel_hi:[1,1,0] (src2 lo broadcast)"),
        V(22, "v_pk_fma_f32 op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_hi:[1,0,0] (cmul_mi second half)"),
        V(23, "v_pk_fma_f32 op_sel:[0,0,1] op_sel_hi:[1,1,0] (src2 swap)"),
        V(24, "v_pk_fma_f32 op_sel:[0,1,0] op_sel_hi:[1,0,1] (src1 swap)"),
        V(25, "v_pk_fma_f32 op_sel:[1,0,0] op_sel_hi:[0,1,1] (src0 swap)"),
        V(26, "v_pk_mov_b32 op_sel:[1,0] (src0 hi, src1 lo: the form the compiler emits)"),
        V(27, "v_pk_mov_b32 op_sel:[0,1] (src0 lo, src1 hi)"),
    };
    int bad = 0;
    for (const Victim &v : victims) {
        v.launch(sv, dOut, VB);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(ref.data(), dOut, vn * sizeof(float2), hipMemcpyDeviceToHost));
        long long wrong = 0, rows[4] = {0, 0, 0, 0};
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipMemset(dOut, 0xff, vn * sizeof(float2)));
            CK(hipDeviceSynchronize());
            hipLaunchKernelGGL(p_lds_mfma, dim3(2048), dim3(256), 65536, sp, dSink, 12000, 65536);
            v.launch(sv, dOut, VB);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(got.data(), dOut, vn * sizeof(float2), hipMemcpyDeviceToHost));
            const unsigned *g = reinterpret_cast<const unsigned *>(got.data()), *r = reinterpret_cast<const unsigned *>(ref.data());
            for (size_t i = 0; i < 2 * vn; ++i)
                if (g[i] != r[i]) { ++wrong; ++rows[((i / 2) & 63) >> 4]; }
        }
        printf("%-75s %9lld wrong words of %zu x 4 (16-lane rows: %lld %lld %lld %lld)  first value %.6g\n", v.name, wrong, 2 * vn, rows[0], rows[1], rows[2],
               rows[3], ref[0].x);
        fflush(stdout);
        bad += wrong != 0;
    }
    printf("forms with wrong results: %d\n", bad);
    return 0;
}
