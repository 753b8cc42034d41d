// Does a matrix-core-heavy kernel on one stream change the results of a vector-ALU kernel on another stream?
// (Round 3: the FFT-path CWT kernels -- packed-f32 butterflies, no MFMA -- came out wrong in 16-lane pieces whenever
// afx_cwt_td.hip's or afx_cqt_f16.hip's kernels really ran beside them; tools/gpu_concurrency*.py.)
// Victims: (a) compiler-generated scalar-f32 fma chains, (b) the packed-f32 complex helpers of afx_asm.h / afx_pkmath.h
// (dft16 + cmul: what the CWT column kernels execute), (c) compiler-generated float2 arithmetic (v_pk_* chosen by the
// compiler), each with and without a trip through LDS.  Partners: back-to-back MFMAs of one type, no memory traffic.
// Every victim result is compared bitwise with the same launch run alone.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I audioflux_amd/csrc/hip tools/micro/mfma_corun.hip -o tools/micro/mfma_corun
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <vector>

#include <afx_asm.h>
#include <afx_pkmath.h>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                             \
            return 2;                                                                  \
        }                                                                              \
    } while (0)

// ---- victims ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void v_plain(float *out, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    float a[16];
    for (int j = 0; j < 16; ++j) a[j] = 1.f + 1e-3f * (float)((t * 16 + j) % 977);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) a[j] = fmaf(a[j], 0.75f, 0.25f * a[(j + 5) & 15]);
    }
    float s = 0.f;
    for (int j = 0; j < 16; ++j) s += a[j] * (float)(j + 1);
    out[t] = s;
}

template <bool LDS>
__global__ __launch_bounds__(256) void v_dft16(float2 *out, int iters) {
    __shared__ v2 ex[LDS ? 4096 : 1];
    const int t = blockIdx.x * 256 + threadIdx.x, tid = threadIdx.x, c = tid & 15, gq = tid >> 4;
    v2 r[16];
    for (int j = 0; j < 16; ++j) r[j] = v2{1.f + 1e-3f * (float)((t + 31 * j) % 911), 0.5f - 1e-3f * (float)((t * 7 + j) % 499)};
    const v2 w = {0.2377f, -0.0711f};  // |w| = 0.2481: four passes of radix-16 stay in range
    for (int it = 0; it < iters; ++it) {
        dft16(r);
#pragma unroll
        for (int j = 0; j < 16; ++j) r[j] = cmul(r[j], w);
        if (LDS) {  // the exchange of k_cwt_inv_cols256: [p][g][c]
#pragma unroll
            for (int p = 0; p < 16; ++p) ex[(p * 16 + gq) * 16 + c] = r[p];
            __syncthreads();
#pragma unroll
            for (int gg = 0; gg < 16; ++gg) r[gg] = ex[(gq * 16 + gg) * 16 + c];
            __syncthreads();
        }
    }
    v2 s = {0.f, 0.f};
    for (int j = 0; j < 16; ++j) s = cfma(r[j], v2{1.f + j, 0.5f}, s);
    out[t] = make_float2(s.x, s.y);
}

__global__ __launch_bounds__(256) void v_float2(float2 *out, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    float2 a[16];
    for (int j = 0; j < 16; ++j) a[j] = make_float2(1.f + 1e-3f * (float)((t + j) % 977), 0.3f + 1e-3f * (float)((t * 3 + j) % 503));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float2 b = a[(j + 3) & 15];
            a[j] = make_float2(fmaf(a[j].x, 0.75f, 0.25f * b.x), fmaf(a[j].y, 0.75f, 0.25f * b.y));
        }
    }
    float2 s = make_float2(0.f, 0.f);
    for (int j = 0; j < 16; ++j) s = make_float2(s.x + a[j].x * (j + 1), s.y + a[j].y * (j + 1));
    out[t] = s;
}

// ---- which instruction pairs are at risk?  one dependent chain per thread, 8 independent chains -----------------------
// MODE 0: v_mul_f32 -> v_fma_f32 (scalar f32), dependent, adjacent, ONE asm statement
// MODE 1: v_pk_mul_f32 -> v_pk_fma_f32, straight operand selects, adjacent, one asm statement
// MODE 2: afx_asm.h cmul (cross-half op_sel), as shipped
// MODE 3: cmul with s_nop 0 between the two instructions
// MODE 4: cmul with s_nop 1 between
// MODE 5: single-instruction asm statements only (pk_add_mi / pk_add_pi of afx_asm.h; the compiler pads them itself)
// MODE 6: compiler-generated serial scalar chain x = fma(x, a, b) (adjacent dependent v_fma_f32, no asm)
// MODE 7: compiler-generated serial float2 chain (adjacent dependent v_pk_fma_f32, no asm)
// MODE 8: two cmuls interleaved in one asm statement (dependent instructions two apart)
template <int MODE>
__global__ __launch_bounds__(256) void v_pairs(float2 *out, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    v2 z[8];
    for (int j = 0; j < 8; ++j) z[j] = v2{0.9f + 1e-4f * (float)((t + 13 * j) % 1013), 0.1f + 1e-4f * (float)((t * 5 + j) % 499)};
    const v2 w = {0.99995f, 0.01f};  // |w| = 1 - 3e-9: the chains neither grow nor vanish
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v2 a = z[j], tt, r;
            if (MODE == 0) {
                float t0, r0;
                asm("v_mul_f32 %0, %2, %3\n\tv_fma_f32 %1, %4, %5, %0" : "=&v"(t0), "=v"(r0) : "v"(a.x), "v"(w.x), "v"(a.y), "v"(-w.y));
                float t1, r1;
                asm("v_mul_f32 %0, %2, %3\n\tv_fma_f32 %1, %4, %5, %0" : "=&v"(t1), "=v"(r1) : "v"(a.x), "v"(w.y), "v"(a.y), "v"(w.x));
                r = v2{r0, r1};
            } else if (MODE == 1) {
                asm("v_pk_mul_f32 %0, %2, %3\n\tv_pk_fma_f32 %1, %2, %4, %0" : "=&v"(tt), "=v"(r) : "v"(a), "v"(w), "v"(v2{0.01f, -0.01f}));
            } else if (MODE == 2) {
                r = cmul(a, w);
            } else if (MODE == 3) {
                asm("v_pk_mul_f32 %0, %2, %3 op_sel:[0,0] op_sel_hi:[0,1]\n\ts_nop 0\n\t"
                    "v_pk_fma_f32 %1, %2, %3, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=&v"(tt), "=v"(r) : "v"(a), "v"(w));
            } else if (MODE == 4) {
                asm("v_pk_mul_f32 %0, %2, %3 op_sel:[0,0] op_sel_hi:[0,1]\n\ts_nop 1\n\t"
                    "v_pk_fma_f32 %1, %2, %3, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=&v"(tt), "=v"(r) : "v"(a), "v"(w));
            } else if (MODE == 5) {
                const v2 u = pk_add_mi(a, w);       // a - i w
                const v2 q = pk_add_pi(u, w);       // + i w: back to a (up to rounding)
                r = pk_add_mi(q, v2{0.f, 0.f});
            } else if (MODE == 6) {
                float x = a.x, y = a.y;
                x = fmaf(x, 0.99995f, 1e-6f);
                x = fmaf(x, 1.00005f, -1e-6f);
                y = fmaf(y, 0.99995f, 1e-6f);
                y = fmaf(y, 1.00005f, -1e-6f);
                r = v2{x, y};
            } else if (MODE == 7) {
                float2 x = make_float2(a.x, a.y);
                x = make_float2(fmaf(x.x, 0.99995f, 1e-6f), fmaf(x.y, 0.99995f, 1e-6f));
                x = make_float2(fmaf(x.x, 1.00005f, -1e-6f), fmaf(x.y, 1.00005f, -1e-6f));
                r = v2{x.x, x.y};
            } else {
                r = a;
            }
            z[j] = r;
        }
        if (MODE == 8) {
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                v2 t0, t1, r0, r1;
                asm("v_pk_mul_f32 %0, %4, %6 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
                    "v_pk_mul_f32 %1, %5, %6 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
                    "v_pk_fma_f32 %2, %4, %6, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"
                    "v_pk_fma_f32 %3, %5, %6, %1 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
                    : "=&v"(t0), "=&v"(t1), "=&v"(r0), "=&v"(r1) : "v"(z[j]), "v"(z[j + 1]), "v"(w));
                z[j] = r0;
                z[j + 1] = r1;
            }
        }
    }
    v2 s = {0.f, 0.f};
    for (int j = 0; j < 8; ++j) s = v2{s.x + z[j].x * (j + 1), s.y + z[j].y * (j + 1)};
    out[t] = make_float2(s.x, s.y);
}

// ---- which FORM of the (a -+ i b) packed add misbehaves?  each step: u = a - i w (form under test), a' = u + i w (same form)
template <int F>
__device__ __forceinline__ v2 form_mi(v2 a, v2 b, v2 cpm, float sone, float smone) {  // a - i b = (a.x + b.y, a.y - b.x)
    v2 r;
    if (F == 1) return pk_add_mi(a, b);
    if (F == 11) { asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
    if (F == 5) { asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=&v"(r) : "v"(a), "v"(b)); return r; }
    if (F == 6) { asm("v_pk_fma_f32 %0, %2, %3, %1 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "v"(b), "v"(cpm)); return r; }
    if (F == 9) { float x, y; asm("v_add_f32 %0, %2, %5\n\tv_sub_f32 %1, %3, %4" : "=&v"(x), "=v"(y) : "v"(a.x), "v"(a.y), "v"(b.x), "v"(b.y)); return v2{x, y}; }
    if (F == 10) return v2{a.x + b.y, a.y - b.x};
    (void)sone; (void)smone;
    return a;
}
template <int F>
__device__ __forceinline__ v2 form_pi(v2 a, v2 b, v2 cmp, float sone, float smone) {  // a + i b = (a.x - b.y, a.y + b.x)
    v2 r;
    if (F == 1) return pk_add_pi(a, b);
    if (F == 11) { asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
    if (F == 5) { asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=&v"(r) : "v"(a), "v"(b)); return r; }
    if (F == 6) { asm("v_pk_fma_f32 %0, %2, %3, %1 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "v"(b), "v"(cmp)); return r; }
    if (F == 9) { float x, y; asm("v_sub_f32 %0, %2, %5\n\tv_add_f32 %1, %3, %4" : "=&v"(x), "=v"(y) : "v"(a.x), "v"(a.y), "v"(b.x), "v"(b.y)); return v2{x, y}; }
    if (F == 10) return v2{a.x - b.y, a.y + b.x};
    (void)sone; (void)smone;
    return a;
}
template <int F>
__global__ __launch_bounds__(256) void v_forms(float2 *out, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    v2 z[8];
    for (int j = 0; j < 8; ++j) z[j] = v2{0.9f + 1e-4f * (float)((t + 13 * j) % 1013), 0.1f + 1e-4f * (float)((t * 5 + j) % 499)};
    const v2 w = {0.25f + 1e-3f * (float)(t & 31), 0.125f};
    v2 cpm = {1.f, -1.f}, cmp = {-1.f, 1.f};
    asm volatile("" : "+v"(cpm), "+v"(cmp));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v2 a = z[j], r;
            if (F == 2) {  // swap only: (a.x + w.y, a.y + w.x) then subtract it again with the same form on -w
                v2 u, nw = {-w.x, -w.y};
                asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(u) : "v"(a), "v"(w));
                asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(u), "v"(nw));
            } else if (F == 3) {  // negation only
                v2 u;
                asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(u) : "v"(a), "v"(w));
                asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(u), "v"(w));
            } else if (F == 4) {  // no modifier
                v2 u, nw = {-w.x, -w.y};
                asm("v_pk_add_f32 %0, %1, %2" : "=v"(u) : "v"(a), "v"(w));
                asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(u), "v"(nw));
            } else if (F == 8) {  // v_pk_mul_f32 with the swap (mul_mi of afx_asm.h): four quarter turns
                r = mul_mi(mul_mi(mul_mi(mul_mi(a))));
            } else {
                const v2 u = form_mi<F>(a, w, cpm, 1.f, -1.f);
                r = form_pi<F>(u, w, cmp, 1.f, -1.f);
            }
            z[j] = r;
        }
    }
    v2 s = {0.f, 0.f};
    for (int j = 0; j < 8; ++j) s = v2{s.x + z[j].x * (j + 1), s.y + z[j].y * (j + 1)};
    out[t] = make_float2(s.x, s.y);
}

// ---- partners: four independent accumulators, MFMAs back to back ------------------------------------------------------
template <int KIND>
__global__ __launch_bounds__(256) void p_mfma(float *sink, int iters) {
    f32x16 acc[4];
    for (int q = 0; q < 4; ++q)
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    const float seed = 1.f + 1e-3f * (float)(threadIdx.x & 63);
    h8 ah, bh;
    b8 ab, bb;
    for (int e = 0; e < 8; ++e) {
        ah[e] = (_Float16)(seed * 0.01f * (e + 1));
        bh[e] = (_Float16)(0.02f * (e + 1));
        ab[e] = (__bf16)(seed * 0.01f * (e + 1));
        bb[e] = (__bf16)(0.02f * (e + 1));
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (KIND == 0) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[q], 0, 0, 0);
            if (KIND == 1) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[q], 0, 0, 0);
            if (KIND == 2) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(seed, 0.02f, acc[q], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int q = 0; q < 4; ++q)
        for (int r = 0; r < 16; ++r) s += acc[q][r];
    if (sink) sink[blockIdx.x * 256 + threadIdx.x] = s;
}

// partner like the K loop of afx_cwt_td.hip / afx_cqt_f16.hip: six ds_read_b128 per six MFMAs out of a large LDS image
// (MFMA = false: the same LDS reads, operands folded with integer adds instead)
template <bool MFMA>
__global__ __launch_bounds__(256) void p_lds_mfma(float *sink, int iters, int ldsBytes) {
    extern __shared__ unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    uint4 *v = reinterpret_cast<uint4 *>(smem);
    const int n16 = ldsBytes / 16;
    for (int e = tid; e < n16; e += 256) v[e] = make_uint4(0x2e662e66u, 0x2e662e66u, 0x2e662e66u, 0x2e662e66u);  // 0.1 in f16
    __syncthreads();
    f32x16 acc[4];
    for (int q = 0; q < 4; ++q)
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    unsigned fold = 0;
    const int span = n16 - 6 * 64 - 64;
    int base = lane;
    for (int it = 0; it < iters; ++it) {
        uint4 o[6];
#pragma unroll
        for (int u = 0; u < 6; ++u) o[u] = v[base + 64 * u];
        base += 64;
        if (base >= span) base = lane;
        if (MFMA) {
            const h8 a0 = __builtin_bit_cast(h8, o[0]), a1 = __builtin_bit_cast(h8, o[1]), a2 = __builtin_bit_cast(h8, o[2]),
                     a3 = __builtin_bit_cast(h8, o[3]), b0 = __builtin_bit_cast(h8, o[4]), b1 = __builtin_bit_cast(h8, o[5]);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[3], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b0, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a3, b0, acc[3], 0, 0, 0);
        } else {
#pragma unroll
            for (int u = 0; u < 6; ++u) fold += o[u].x + o[u].y + o[u].z + o[u].w;
        }
    }
    float s = (float)fold;
    for (int q = 0; q < 4; ++q)
        for (int r = 0; r < 16; ++r) s += acc[q][r];
    if (sink) sink[blockIdx.x * 256 + threadIdx.x] = s;
}

// partner without matrix cores: scalar-f32 fma chains (control)
__global__ __launch_bounds__(256) void p_valu(float *sink, int iters) {
    float a[8];
    for (int j = 0; j < 8; ++j) a[j] = 1.f + 1e-3f * (float)(threadIdx.x + j);
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = fmaf(a[j], 0.75f, 0.25f * a[(j + 3) & 7]);
    if (sink) sink[blockIdx.x * 256 + threadIdx.x] = a[0] + a[3] + a[7];
}

int main() {
    const int VB = 4096, PB = 2048;  // workgroups of 256
    const size_t vn = (size_t)VB * 256;
    float2 *dOut;
    float *dSink;
    CK(hipMalloc(&dOut, vn * sizeof(float2)));
    CK(hipMalloc(&dSink, (size_t)PB * 256 * sizeof(float)));
    hipStream_t sv, sp;
    CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sp, hipStreamNonBlocking));
    std::vector<float2> ref(vn), got(vn);

    struct Victim {
        const char *name;
        int words;  // floats per thread written
        void (*launch)(hipStream_t, void *, int);
    };
#define PAIRV(M, NAME) {NAME, 2, [](hipStream_t s, void *o, int vb) { hipLaunchKernelGGL(v_pairs<M>, dim3(vb), dim3(256), 0, s, (float2 *)o, 4000); }}
#define FORMV(M, NAME) {NAME, 2, [](hipStream_t s, void *o, int vb) { hipLaunchKernelGGL(v_forms<M>, dim3(vb), dim3(256), 0, s, (float2 *)o, 4000); }}
    const Victim victims[] = {
        FORMV(1, "shipped: pk_add_mi / pk_add_pi of afx_asm.h (v_pk_fma_f32 with 1.0, swapped src0, neg)"),
        FORMV(11, "form: v_pk_add_f32 swap + neg (pk_add_mi / pk_add_pi until round 3)"),
        FORMV(2, "form: v_pk_add_f32 swap only"),
        FORMV(3, "form: v_pk_add_f32 neg only"),
        FORMV(4, "form: v_pk_add_f32 no modifier"),
        FORMV(5, "form: v_pk_add_f32 swap + neg, early-clobber destination"),
        FORMV(6, "form: v_pk_fma_f32 with (1,-1) constant pair, swapped src0"),
        FORMV(8, "shipped: mul_mi of afx_asm.h (v_pk_mul_f32 with the swap)"),
        FORMV(9, "form: v_add_f32 + v_sub_f32"),
        FORMV(10, "form: compiler-generated (a.x + b.y, a.y - b.x)"),
        PAIRV(0, "asm: v_mul_f32 -> v_fma_f32 adjacent (scalar f32)"),
        PAIRV(1, "asm: v_pk_mul_f32 -> v_pk_fma_f32 adjacent, straight selects"),
        PAIRV(2, "shipped: cmul of afx_asm.h (adjacent, cross-half op_sel)"),
        PAIRV(3, "asm: cmul with s_nop 0 between"),
        PAIRV(4, "asm: cmul with s_nop 1 between"),
        PAIRV(5, "shipped: pk_add_mi / pk_add_pi chains, one statement each (compiler-padded)"),
        PAIRV(6, "compiler: serial v_fma_f32 chain"),
        PAIRV(7, "compiler: serial float2 fma chain"),
        PAIRV(8, "asm: two cmuls interleaved (dependent instructions two apart)"),
        {"scalar f32 fma chains (compiler)", 1, [](hipStream_t s, void *o, int vb) { hipLaunchKernelGGL(v_plain, dim3(vb), dim3(256), 0, s, (float *)o, 3000); }},
        {"float2 arithmetic (compiler-chosen v_pk_*)", 2, [](hipStream_t s, void *o, int vb) { hipLaunchKernelGGL(v_float2, dim3(vb), dim3(256), 0, s, (float2 *)o, 1500); }},
        {"shipped: dft16 + cmul (afx_pkmath.h), registers only", 2, [](hipStream_t s, void *o, int vb) { hipLaunchKernelGGL(v_dft16<false>, dim3(vb), dim3(256), 0, s, (float2 *)o, 400); }},
        {"shipped: dft16 + cmul + LDS exchange", 2, [](hipStream_t s, void *o, int vb) { hipLaunchKernelGGL(v_dft16<true>, dim3(vb), dim3(256), 0, s, (float2 *)o, 300); }},
    };
    struct Partner {
        const char *name;
        void (*launch)(hipStream_t, float *, int);
    };
    const Partner partners[] = {
        {"none", nullptr},
        {"scalar f32 fma chains (no MFMA)", [](hipStream_t s, float *k, int pb) { hipLaunchKernelGGL(p_valu, dim3(pb), dim3(256), 0, s, k, 60000); }},
        {"v_mfma_f32_32x32x16_f16 back to back", [](hipStream_t s, float *k, int pb) { hipLaunchKernelGGL(p_mfma<0>, dim3(pb), dim3(256), 0, s, k, 12000); }},
        {"v_mfma_f32_32x32x16_bf16 back to back", [](hipStream_t s, float *k, int pb) { hipLaunchKernelGGL(p_mfma<1>, dim3(pb), dim3(256), 0, s, k, 12000); }},
        {"v_mfma_f32_32x32x2f32 back to back", [](hipStream_t s, float *k, int pb) { hipLaunchKernelGGL(p_mfma<2>, dim3(pb), dim3(256), 0, s, k, 6000); }},
        {"6 ds_read_b128 + 6 f16 MFMAs per step, 150,272 B of LDS per workgroup", [](hipStream_t s, float *k, int) { hipLaunchKernelGGL(p_lds_mfma<true>, dim3(1024), dim3(256), 150272, s, k, 12000, 150272); }},
        {"6 ds_read_b128 + 6 f16 MFMAs per step, 65,536 B of LDS per workgroup", [](hipStream_t s, float *k, int) { hipLaunchKernelGGL(p_lds_mfma<true>, dim3(2048), dim3(256), 65536, s, k, 12000, 65536); }},
        {"6 ds_read_b128 per step (no MFMA), 65,536 B of LDS per workgroup", [](hipStream_t s, float *k, int) { hipLaunchKernelGGL(p_lds_mfma<false>, dim3(2048), dim3(256), 65536, s, k, 24000, 65536); }},
    };
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(p_lds_mfma<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(p_lds_mfma<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    int total = 0;
    for (const Victim &v : victims) {
        const size_t bytes = vn * v.words * sizeof(float);
        CK(hipMemset(dOut, 0xff, vn * sizeof(float2)));
        CK(hipDeviceSynchronize());  // (the victim's stream is non-blocking: it does not wait for the null stream's memset)
        v.launch(sv, dOut, VB);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(ref.data(), dOut, bytes, hipMemcpyDeviceToHost));
        for (const Partner &p : partners) {
            long long wrong = 0, wrongRows[4] = {0, 0, 0, 0};
            double tv = 0, tp = 0;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipMemset(dOut, 0xff, vn * sizeof(float2)));
                CK(hipDeviceSynchronize());
                hipEvent_t e0, e1, p0, p1;
                CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&p0)); CK(hipEventCreate(&p1));
                CK(hipEventRecord(p0, sp));
                if (p.launch) p.launch(sp, dSink, PB);
                CK(hipEventRecord(p1, sp));
                CK(hipEventRecord(e0, sv));
                v.launch(sv, dOut, VB);
                CK(hipEventRecord(e1, sv));
                CK(hipDeviceSynchronize());
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1)); tv += ms;
                CK(hipEventElapsedTime(&ms, p0, p1)); tp += ms;
                CK(hipMemcpy(got.data(), dOut, bytes, hipMemcpyDeviceToHost));
                const unsigned *g = reinterpret_cast<const unsigned *>(got.data()), *r = reinterpret_cast<const unsigned *>(ref.data());
                for (size_t i = 0; i < vn * v.words; ++i)
                    if (g[i] != r[i]) {
                        ++wrong;
                        const size_t thread = i / v.words;
                        ++wrongRows[(thread & 63) >> 4];
                        if (wrong <= 3) {
                            float a, b;
                            memcpy(&a, &g[i], 4); memcpy(&b, &r[i], 4);
                            printf("      thread %zu (wave lane %zu) word %zu: got %.9g want %.9g\n", thread, thread & 63, i % v.words, a, b);
                        }
                    }
            }
            printf("victim [%s] beside [%s]: %lld wrong words of %zu x 4 runs (by 16-lane row of the wave: %lld %lld %lld %lld); victim %.2f ms, partner %.2f ms per run\n",
                   v.name, p.name, wrong, vn * v.words, wrongRows[0], wrongRows[1], wrongRows[2], wrongRows[3], tv / 4, tp / 4);
            fflush(stdout);
            total += wrong != 0;
        }
    }
    printf("combinations with wrong results: %d\n", total);
    return 0;
}
