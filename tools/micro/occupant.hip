// Co-runner kernels for tools/gpu_concurrency.py: each stresses ONE resource the time-domain CWT kernel
// (csrc/hip/afx_cwt_td.hip) uses, so that a concurrent run of the FFT-path CWT kernels can tell which of them
// (if any) disturbs it.  Launchers take a HIP stream; nothing here touches the library's buffers unless the
// caller passes one (occ_oob: out-of-range buffer stores against a live output buffer).
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/micro/occupant.hip -o tools/micro/libocc.so
#include <hip/hip_runtime.h>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int RSRC_RAW = 0x00020000;
constexpr unsigned OOR = 0x80000000u;

// 1. LDS + matrix cores, no global traffic: a workgroup that holds `ldsBytes` of a CU's LDS, sweeps all of it with
//    writes and reads and keeps the MFMA pipe busy.
__global__ __launch_bounds__(256) void k_occ_lds(int ldsBytes, int iters, float *sink) {
    extern __shared__ unsigned char smem[];
    const int tid = threadIdx.x;
    uint4 *v = reinterpret_cast<uint4 *>(smem);
    const int n16 = ldsBytes / 16;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    unsigned s = 0;
    for (int it = 0; it < iters; ++it) {
        for (int e = tid; e < n16; e += 256) v[e] = make_uint4(e + it, e ^ it, 0x3c003c00u, 0x3c003c00u);
        __syncthreads();
        for (int e = tid; e < n16; e += 256) {
            const uint4 q = v[(e * 7 + it) % n16];
            s += q.x ^ q.y;
            const h8 a = __builtin_bit_cast(h8, q);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    if (sink) sink[blockIdx.x * 256 + tid] = acc[0] + (float)s;
}

// 2. HBM streaming with 16-byte raw buffer loads and stores (private buffers), small LDS footprint
__global__ __launch_bounds__(256) void k_occ_mem(const float *src, float *dst, long long n4, int iters) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, (int)(n4 * 16), RSRC_RAW);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dst, 0, (int)(n4 * 16), RSRC_RAW);
    for (int it = 0; it < iters; ++it)
        for (long long e = blockIdx.x * 256LL + threadIdx.x; e < n4; e += (long long)gridDim.x * 256) {
            u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)e * 16u, 0, 0);
            q.x += it;
            __builtin_amdgcn_raw_buffer_store_b128(q, rd, (unsigned)e * 16u, 0, 0);
        }
}

// 3. out-of-range (dropped) raw buffer stores and loads against `base` (e.g. a live output buffer of another stream)
__global__ __launch_bounds__(256) void k_occ_oob(float *base, int iters, float *sink) {
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(base, 0, 4, RSRC_RAW);
    const u32x4 z = {0x7fc00000u, 0x7fc00000u, 0x7fc00000u, 0x7fc00000u};  // NaNs: visible if one ever lands
    unsigned s = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) __builtin_amdgcn_raw_buffer_store_b128(z, rd, OOR + 16u * (r + 8 * threadIdx.x), 0, 0);
        const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rd, OOR + 16u * threadIdx.x, 0, 0);
        s += q.x;
    }
    if (sink) sink[blockIdx.x * 256 + threadIdx.x] = (float)s;
}

extern "C" int occ_lds(void *stream, int blocks, int ldsBytes, int iters, float *sink) {
    static bool set = false;
    if (!set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_occ_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return 1;
        set = true;
    }
    hipLaunchKernelGGL(k_occ_lds, dim3(blocks), dim3(256), ldsBytes, (hipStream_t)stream, ldsBytes, iters, sink);
    return hipGetLastError() != hipSuccess;
}
extern "C" int occ_mem(void *stream, int blocks, const float *src, float *dst, long long n4, int iters) {
    hipLaunchKernelGGL(k_occ_mem, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, dst, n4, iters);
    return hipGetLastError() != hipSuccess;
}
extern "C" int occ_oob(void *stream, int blocks, float *base, int iters, float *sink) {
    hipLaunchKernelGGL(k_occ_oob, dim3(blocks), dim3(256), 0, (hipStream_t)stream, base, iters, sink);
    return hipGetLastError() != hipSuccess;
}
