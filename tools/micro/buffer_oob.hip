// What k_cqt_octave_f16 / k_cqt_all_f16 rely on, checked in isolation on the device:
//   1. raw buffer loads (stride 0) are bounds-checked PER DWORD: a buffer_load_dwordx4 that straddles num_records
//      returns the in-range dwords and zeros for the rest (the zero padding behind the last framed sample when
//      validLength is not a multiple of 4: the lowest CQT octave, hop 2);
//   2. a negative offset (as unsigned: >= 2^31) is out of range and reads 0 (the zero padding in front of frame 0);
//   3. out-of-range buffer stores (dword / dwordx3 / dwordx4) are dropped, also when only the row is beyond the
//      buffer, and a dwordx3 store whose last dword would fall outside is dropped per dword as well;
//   4. v_mfma_f32_32x32x16_f16: A[i = lane & 31][k = 8 (lane >> 5) + e] x B[k][j = lane & 31], the D layout
//      col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5), against a host product.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/buffer_oob.hip -o tools/micro/buffer_oob && tools/micro/buffer_oob
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int RSRC_RAW = 0x00020000;

__global__ void k_loads(const float *x, int valid, float *out /* [64][4] */, int base) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x), 0, valid * 4, RSRC_RAW);
    const int lane = threadIdx.x;
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (base + 4 * lane) * 4, 0, 0);
    out[4 * lane + 0] = __uint_as_float(v.x);
    out[4 * lane + 1] = __uint_as_float(v.y);
    out[4 * lane + 2] = __uint_as_float(v.z);
    out[4 * lane + 3] = __uint_as_float(v.w);
}

__global__ void k_stores(float *y, int valid /* floats inside the buffer */) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(y, 0, valid * 4, RSRC_RAW);
    const int lane = threadIdx.x;
    // lanes walk across the end of the buffer in steps of 3 floats
    const u32x3 v3 = {__float_as_uint(1.f + lane), __float_as_uint(2.f + lane), __float_as_uint(3.f + lane)};
    __builtin_amdgcn_raw_buffer_store_b96(v3, r, (valid - 96 + 3 * lane) * 4, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(-7.f), r, 0x80000000u + 4u * lane, 0, 0);
}

__global__ void k_mfma(const _Float16 *A /* [32][16] */, const _Float16 *B /* [16][32] */, float *D /* [32][32] */) {
    const int lane = threadIdx.x, i = lane & 31, g = lane >> 5;
    h8 a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = A[i * 16 + 8 * g + e];
        b[e] = B[(8 * g + e) * 32 + i];
    }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * g) * 32 + i] = acc[r];
}

#define CK(x)                                                                 \
    do {                                                                      \
        hipError_t e = (x);                                                   \
        if (e != hipSuccess) {                                                \
            printf("%s: %s\n", #x, hipGetErrorString(e));                     \
            return 2;                                                         \
        }                                                                     \
    } while (0)

int main() {
    int bad = 0;
    // ---- loads
    const int n = 1024;
    std::vector<float> hx(n);
    for (int i = 0; i < n; ++i) hx[i] = 1.f + i;
    float *dx, *dout;
    CK(hipMalloc(&dx, n * 4));
    CK(hipMalloc(&dout, 256 * 4));
    CK(hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice));
    for (int valid : {200, 201, 202, 203}) {
        for (int base : {0, -8}) {
            hipLaunchKernelGGL(k_loads, dim3(1), dim3(64), 0, 0, dx, valid, dout, base);
            std::vector<float> ho(256);
            CK(hipMemcpy(ho.data(), dout, 256 * 4, hipMemcpyDeviceToHost));
            for (int q = 0; q < 256; ++q) {
                const int s = base + q;
                const float want = (s >= 0 && s < valid) ? hx[s] : 0.f;
                if (ho[q] != want) {
                    if (bad < 10) printf("load: valid %d base %d sample %d: got %g want %g\n", valid, base, s, ho[q], want);
                    ++bad;
                }
            }
        }
    }
    printf("per-dword bounds check of buffer_load_dwordx4, negative offsets read 0: %s\n", bad ? "FAILED" : "ok");
    // ---- stores
    int bad2 = 0;
    float *dy;
    CK(hipMalloc(&dy, n * 4));
    for (int valid : {400, 401, 402}) {
        CK(hipMemset(dy, 0, n * 4));
        hipLaunchKernelGGL(k_stores, dim3(1), dim3(64), 0, 0, dy, valid);
        std::vector<float> hy(n);
        CK(hipMemcpy(hy.data(), dy, n * 4, hipMemcpyDeviceToHost));
        for (int p = 0; p < n; ++p) {
            float want = 0.f;
            const int rel = p - (valid - 96);
            if (p < valid && rel >= 0 && rel < 192) want = (float)(1 + rel % 3) + (float)(rel / 3);
            if (hy[p] != want) {
                if (bad2 < 10) printf("store: valid %d position %d: got %g want %g\n", valid, p, hy[p], want);
                ++bad2;
            }
        }
    }
    printf("out-of-range stores dropped per dword: %s\n", bad2 ? "FAILED" : "ok");
    // ---- MFMA layout
    int bad3 = 0;
    std::vector<_Float16> hA(32 * 16), hB(16 * 32);
    for (int i = 0; i < 32; ++i)
        for (int k = 0; k < 16; ++k) hA[i * 16 + k] = (_Float16)(float)((i * 7 + k * 3) % 11 - 5);
    for (int k = 0; k < 16; ++k)
        for (int j = 0; j < 32; ++j) hB[k * 32 + j] = (_Float16)(float)((k * 5 + j * 2) % 9 - 4);   // asymmetric
    _Float16 *dA, *dB;
    float *dD;
    CK(hipMalloc(&dA, hA.size() * 2));
    CK(hipMalloc(&dB, hB.size() * 2));
    CK(hipMalloc(&dD, 32 * 32 * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    std::vector<float> hD(32 * 32);
    CK(hipMemcpy(hD.data(), dD, 32 * 32 * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            float want = 0.f;
            for (int k = 0; k < 16; ++k) want += (float)hA[i * 16 + k] * (float)hB[k * 32 + j];
            if (hD[i * 32 + j] != want) {
                if (bad3 < 10) printf("mfma: D[%d][%d] got %g want %g\n", i, j, hD[i * 32 + j], want);
                ++bad3;
            }
        }
    printf("v_mfma_f32_32x32x16_f16 operand / result layout: %s\n", bad3 ? "FAILED" : "ok");
    return (bad || bad2 || bad3) ? 1 : 0;
}
