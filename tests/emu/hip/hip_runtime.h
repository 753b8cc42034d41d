// tests/emu/hip/hip_runtime.h -- a HOST EMULATION of the slice of HIP / gfx950 that the CQT wave kernels use, so that
// their DEVICE code (audioflux_amd/csrc/hip/afx_cqt_f16.hip, afx_gemm_bf16.hip, ..., included unchanged) can be compiled for
// x86 and run on the CPU: tests/emu/cqt_emulated_*.cpp + tests/test_emulated_kernels.py.
//
// One host thread per lane.  Per-lane code runs as written; every cross-lane operation (MFMA, DPP, readlane,
// readfirstlane, the wave barrier that orders LDS traffic, __syncthreads) is a rendezvous of the wave's / the
// workgroup's threads through an exchange buffer:
//   * v_mfma_f32_32x32x16_f16: A operand lane l = row l & 31, k = 8 (l >> 5) + e; B operand lane l = column l & 31,
//     same k; D register r of lane l = row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31 (MI355X_MICROARCH.md);
//     products exact in float32, accumulated in float32 in k order;
//   * raw buffer loads / stores: per-dword bounds check against num_records, out-of-range loads return 0,
//     out-of-range stores are dropped (the behaviour tools/micro/buffer_oob.hip checks on the device);
//   * DPP controls quad_perm / row_mirror / row_half_mirror, all lanes enabled.
// What it cannot show: timing, register pressure, the hardware's own MFMA rounding order.  Test infrastructure,
// never linked into the product.
#ifndef AFX_EMU_HIP_RUNTIME_H
#define AFX_EMU_HIP_RUNTIME_H
#define AFX_HOST_EMULATION 1

#include <pthread.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

// ---- language
#define __global__
#define __device__
#define __host__
#define __shared__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define AFX_WAVES_PER_EU(lo, hi)  // (a register-allocation hint of the device build)

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
struct EmuIdx {
    unsigned x, y, z;
};
extern thread_local EmuIdx threadIdx, blockIdx, blockDim, gridDim;

struct alignas(8) float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) uint2 { unsigned x, y; };
struct alignas(8) int2 { int x, y; };
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
static inline float2 make_float2(float a, float b) { return float2{a, b}; }
static inline uint2 make_uint2(unsigned a, unsigned b) { return uint2{a, b}; }
static inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
static inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }

// ---- the runtime calls the launchers make (results do not matter here)
typedef int hipError_t;
typedef void *hipStream_t;
enum { hipSuccess = 0 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2 , hipMemcpyDeviceToDevice };
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char *hipGetErrorString(hipError_t) { return "emulation"; }
static inline hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
template <class T>
static inline hipError_t hipMalloc(T **p, size_t n) { *p = static_cast<T *>(calloc(1, n)); return hipSuccess; }
static inline hipError_t hipMemset(void *p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMallocAsync(void **p, size_t n, hipStream_t) { *p = malloc(n ? n : 1); return hipSuccess; }
static inline hipError_t hipFreeAsync(void *p, hipStream_t) { free(p); return hipSuccess; }
static inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 1 };
static inline hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t, int) {  // few "CUs": few workgroups (AFX_EMU_CUS: long runs of frames per wave)
    const char *e = getenv("AFX_EMU_CUS");
    *v = e && atoi(e) > 0 ? atoi(e) : 4;
    return hipSuccess;
}

// ---- launch: workgroups one after the other, one thread per lane
namespace emu {
void launch(const char *kernel, dim3 grid, dim3 block, const std::function<void()> &body);
void wave_barrier();   // all 64 lanes of the calling lane's wave
void block_barrier();  // all threads of the workgroup
unsigned *exchange();  // the wave's exchange buffer: [64][32] dwords
int lane();
}  // namespace emu
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) emu::launch(#kernel, (grid), (block), [&] { kernel(__VA_ARGS__); })
static inline void __syncthreads() { emu::block_barrier(); }
#define __builtin_amdgcn_s_barrier() emu::block_barrier()  // (k_cqt_pyramid: one per step, every wave of the workgroup)

// ---- cross-lane operations
static inline int emu_readlane(int v, int src) {
    unsigned *x = emu::exchange();
    x[emu::lane() * 32] = (unsigned)v;
    emu::wave_barrier();
    const int r = (int)x[src * 32];
    emu::wave_barrier();
    return r;
}
static inline int emu_mov_dpp(int v, int ctrl) {
    const int l = emu::lane();
    int src;
    if (ctrl < 0x100) src = (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3);       // quad_perm
    else if (ctrl == 0x140) src = (l & ~15) | (15 - (l & 15));                // row_mirror
    else if (ctrl == 0x141) src = (l & ~7) | (7 - (l & 7));                   // row_half_mirror
    else { fprintf(stderr, "emu: DPP control %#x not modelled\n", ctrl); abort(); }
    return emu_readlane(v, src);
}
#define __builtin_amdgcn_readlane(v, l) emu_readlane((v), (l))
#define __builtin_amdgcn_readfirstlane(v) emu_readlane((v), 0)
#define __builtin_amdgcn_mov_dpp(v, ctrl, rowmask, bankmask, bc) emu_mov_dpp((v), (ctrl))
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#ifdef AFX_EMU_NO_LDS_ORDER  // the race check's own test: with the ordering points gone ThreadSanitizer must complain
#define __builtin_amdgcn_wave_barrier() ((void)0)
#else
#define __builtin_amdgcn_wave_barrier() emu::wave_barrier()
#endif
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(a, b, c) ((void)0)
#define __builtin_amdgcn_s_memrealtime() 0ull
#define __builtin_amdgcn_s_memtime() 0ull
#define __builtin_amdgcn_s_setprio(p) ((void)0)  // wave priorities order the issue port, not the results

typedef float emu_f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 emu_h8 __attribute__((ext_vector_type(8)));
static inline emu_f32x16 emu_mfma_32x32x16_f16(emu_h8 a, emu_h8 b, emu_f32x16 c) {
    unsigned *x = emu::exchange();
    const int l = emu::lane();
    memcpy(x + l * 32, &a, 16);
    memcpy(x + l * 32 + 4, &b, 16);
    emu::wave_barrier();
    const int col = l & 31, g = l >> 5;
    emu_f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * g;
        float acc = c[r];
        for (int k = 0; k < 16; ++k) {
            _Float16 av, bv;
            memcpy(&av, reinterpret_cast<const char *>(x + (row + 32 * (k >> 3)) * 32) + 2 * (k & 7), 2);
            memcpy(&bv, reinterpret_cast<const char *>(x + (col + 32 * (k >> 3)) * 32 + 4) + 2 * (k & 7), 2);
            acc += (float)av * (float)bv;
        }
        d[r] = acc;
    }
    emu::wave_barrier();
    return d;
}
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) emu_mfma_32x32x16_f16((a), (b), (c))

// v_mfma_f32_16x16x32_f16: A operand lane l = row l & 15, k = 8 (l >> 4) + e; B operand lane l = column l & 15, same k;
// D register r of lane l = row 4 (l >> 4) + r, column l & 15
typedef float emu_f32x4h __attribute__((ext_vector_type(4)));
static inline emu_f32x4h emu_mfma_16x16x32_f16(emu_h8 a, emu_h8 b, emu_f32x4h c) {
    unsigned *x = emu::exchange();
    const int l = emu::lane();
    memcpy(x + l * 32, &a, 16);
    memcpy(x + l * 32 + 4, &b, 16);
    emu::wave_barrier();
    const int col = l & 15;
    emu_f32x4h d = c;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * (l >> 4) + r;
        float acc = c[r];
        for (int k = 0; k < 32; ++k) {
            _Float16 av, bv;
            memcpy(&av, reinterpret_cast<const char *>(x + (row + 16 * (k >> 3)) * 32) + 2 * (k & 7), 2);
            memcpy(&bv, reinterpret_cast<const char *>(x + (col + 16 * (k >> 3)) * 32 + 4) + 2 * (k & 7), 2);
            acc += (float)av * (float)bv;
        }
        d[r] = acc;
    }
    emu::wave_barrier();
    return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, x, y, z) emu_mfma_16x16x32_f16((a), (b), (c))

// v_mfma_f32_32x32x2_f32: A operand lane l = row l & 31, k = l >> 5; B operand lane l = column l & 31, k = l >> 5
static inline emu_f32x16 emu_mfma_32x32x2_f32(float a, float b, emu_f32x16 c) {
    unsigned *x = emu::exchange();
    const int l = emu::lane();
    memcpy(x + l * 32, &a, 4);
    memcpy(x + l * 32 + 1, &b, 4);
    emu::wave_barrier();
    const int col = l & 31, g = l >> 5;
    emu_f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * g;
        float acc = c[r];
        for (int k = 0; k < 2; ++k) acc = fmaf(__uint_as_float(x[(row + 32 * k) * 32]), __uint_as_float(x[(col + 32 * k) * 32 + 1]), acc);
        d[r] = acc;
    }
    emu::wave_barrier();
    return d;
}
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) emu_mfma_32x32x2_f32((a), (b), (c))
static inline unsigned __brev(unsigned v) {
    unsigned r = 0;
    for (int i = 0; i < 32; ++i) r |= ((v >> i) & 1u) << (31 - i);
    return r;
}

// v_mfma_f32_32x32x16_bf16: the same operand / result layout with bf16 words
typedef __bf16 emu_bf8 __attribute__((ext_vector_type(8)));
static inline float emu_bf16_value(unsigned short w) { return __uint_as_float((unsigned)w << 16); }
static inline emu_f32x16 emu_mfma_32x32x16_bf16(emu_bf8 a, emu_bf8 b, emu_f32x16 c) {
    unsigned *x = emu::exchange();
    const int l = emu::lane();
    memcpy(x + l * 32, &a, 16);
    memcpy(x + l * 32 + 4, &b, 16);
    emu::wave_barrier();
    const int col = l & 31, g = l >> 5;
    emu_f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * g;
        float acc = c[r];
        for (int k = 0; k < 16; ++k) {
            unsigned short av, bv;
            memcpy(&av, reinterpret_cast<const char *>(x + (row + 32 * (k >> 3)) * 32) + 2 * (k & 7), 2);
            memcpy(&bv, reinterpret_cast<const char *>(x + (col + 32 * (k >> 3)) * 32 + 4) + 2 * (k & 7), 2);
            acc += emu_bf16_value(av) * emu_bf16_value(bv);
        }
        d[r] = acc;
    }
    emu::wave_barrier();
    return d;
}
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) emu_mfma_32x32x16_bf16((a), (b), (c))

// ---- raw buffer resources
struct __amdgpu_buffer_rsrc_t {
    char *base;
    unsigned records;
};
typedef unsigned emu_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned emu_u32x3 __attribute__((ext_vector_type(3)));
static inline __amdgpu_buffer_rsrc_t emu_make_rsrc(const void *p, short, int records, int) {
    return __amdgpu_buffer_rsrc_t{const_cast<char *>(static_cast<const char *>(p)), (unsigned)records};
}
static inline unsigned emu_buf_load(const __amdgpu_buffer_rsrc_t &r, unsigned off) {
    unsigned v = 0;
    if ((unsigned long long)off + 4 <= r.records) memcpy(&v, r.base + off, 4);
    return v;
}
static inline void emu_buf_store(const __amdgpu_buffer_rsrc_t &r, unsigned off, unsigned v) {
    if ((unsigned long long)off + 4 <= r.records) memcpy(r.base + off, &v, 4);
}
static inline emu_u32x4 emu_load_b128(const __amdgpu_buffer_rsrc_t &r, int voff, int soff) {
    const unsigned o = (unsigned)voff + (unsigned)soff;
    return emu_u32x4{emu_buf_load(r, o), emu_buf_load(r, o + 4), emu_buf_load(r, o + 8), emu_buf_load(r, o + 12)};
}
#define __builtin_amdgcn_make_buffer_rsrc(p, stride, num, flags) emu_make_rsrc((p), (stride), (num), (flags))
#define __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, aux) emu_load_b128((r), (voff), (soff))
#define __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, aux) emu_buf_load((r), (unsigned)(voff) + (unsigned)(soff))
#define __builtin_amdgcn_raw_buffer_store_b32(v, r, voff, soff, aux) emu_buf_store((r), (unsigned)(voff) + (unsigned)(soff), (v))
#define __builtin_amdgcn_raw_buffer_store_b96(v, r, voff, soff, aux)                      \
    do {                                                                                   \
        const emu_u32x3 _v = (v);                                                          \
        const unsigned _o = (unsigned)(voff) + (unsigned)(soff);                           \
        emu_buf_store((r), _o, _v.x), emu_buf_store((r), _o + 4, _v.y), emu_buf_store((r), _o + 8, _v.z); \
    } while (0)
#define __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, aux)                     \
    do {                                                                                   \
        const emu_u32x4 _v = (v);                                                          \
        const unsigned _o = (unsigned)(voff) + (unsigned)(soff);                           \
        emu_buf_store((r), _o, _v.x), emu_buf_store((r), _o + 4, _v.y), emu_buf_store((r), _o + 8, _v.z), \
            emu_buf_store((r), _o + 12, _v.w);                                             \
    } while (0)

// ds_bpermute_b32: lane l receives the value of lane (byte address / 4) & 63
static inline int emu_ds_bpermute(int addr, int v) {
    unsigned *x = emu::exchange();
    x[emu::lane() * 32] = (unsigned)v;
    emu::wave_barrier();
    const int r = (int)x[((addr >> 2) & 63) * 32];
    emu::wave_barrier();
    return r;
}
#define __builtin_amdgcn_ds_bpermute(addr, v) emu_ds_bpermute((addr), (v))

// ---- wave shuffles, fast-math intrinsics
template <class T>
static inline T emu_shfl_from(T v, int src) {
    static_assert(sizeof(T) == 4, "32-bit values");
    int u;
    memcpy(&u, &v, 4);
    u = emu_readlane(u, src);
    T r;
    memcpy(&r, &u, 4);
    return r;
}
template <class T>
static inline T __shfl(T v, int src, int = 64) { return emu_shfl_from(v, src & 63); }
template <class T>
static inline T __shfl_xor(T v, int mask, int = 64) { return emu_shfl_from(v, emu::lane() ^ mask); }
template <class T>
static inline T __shfl_down(T v, int delta, int = 64) { return emu_shfl_from(v, emu::lane() + delta < 64 ? emu::lane() + delta : emu::lane()); }
template <class T>
static inline T __shfl_up(T v, int delta, int = 64) { return emu_shfl_from(v, emu::lane() >= delta ? emu::lane() - delta : emu::lane()); }
#define __log2f(x) log2f(x)
#define __builtin_amdgcn_exp2f(x) exp2f(x)

// v_mfma_f32_16x16x4_f32: A operand lane l = row l & 15, k = l >> 4; B operand lane l = column l & 15, k = l >> 4;
// D register r of lane l = row 4 (l >> 4) + r, column l & 15
typedef float emu_f32x4 __attribute__((ext_vector_type(4)));
static inline emu_f32x4 emu_mfma_16x16x4_f32(float a, float b, emu_f32x4 c) {
    unsigned *x = emu::exchange();
    const int l = emu::lane();
    memcpy(x + l * 32, &a, 4);
    memcpy(x + l * 32 + 1, &b, 4);
    emu::wave_barrier();
    emu_f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * (l >> 4) + r, col = l & 15;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc = fmaf(__uint_as_float(x[(row + 16 * k) * 32]), __uint_as_float(x[(col + 16 * k) * 32 + 1]), acc);
        d[r] = acc;
    }
    emu::wave_barrier();
    return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) emu_mfma_16x16x4_f32((a), (b), (c))

#endif
