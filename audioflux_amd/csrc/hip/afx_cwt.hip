// afx_cwt.hip -- continuous wavelet transform kernels ("K9/K10" of SURVEY.md 2b).
//
// The reference (__cwtObj_cwt, src/cwt_algorithm.c:361-483) reflect-pads the signal,
// takes one FFT of length L, multiplies the spectrum by each of `num` real
// frequency-domain wavelets and runs `num` inverse FFTs of length L, then crops.  L is
// 2^r or 2^(r+1) -- far beyond what one CU's LDS holds -- so the transforms here are
// "four-step" FFTs over the factorisation L = L1 * L2:
//
//   forward   pass 1  columns (size L1, stride L2), tiled C columns per workgroup, the
//                     reflect padding is applied while loading, twiddle W_L^(k1 n2)
//             pass 2  rows (size L2, contiguous) -> spectrum in the transposed layout
//                     Xt[k1][k2] (frequency k = k1 + L1 k2)
//   inverse   pass 1  rows: Xt * wavelet_j (bank stored in the same transposed layout),
//                     conjugated, FFT over k2, twiddle            (grid: L1 x num)
//             pass 2  columns: FFT over k1, conjugate, 1/L, and ONLY the cropped window
//                     [pad, pad+2^r) is stored, in natural time order (grid: tiles x num)
//
// so every global access is contiguous or 128-byte-tiled, the inverse needs no transpose,
// and the padded halves are never written.  IFFT(z) = conj(FFT(conj z))/L as in
// fftObj_ifft (src/dsp/fft_algorithm.c:559-623).  The d/dt variant (cwtObj_cwtDet)
// multiplies by j*omega*wavelet: (re,im) -> (-b*im, b*re), cwt_algorithm.c:432-435.
#include <hip/hip_runtime.h>

#include "afx_device.h"
#include "afx_hipcheck.h"

namespace {

__device__ __forceinline__ int brev(int k, int r) {
    return r == 0 ? 0 : (int)(__brev((unsigned)k) >> (32 - r));
}

// `cnt` FFTs of size 2^r held in LDS; element i of FFT f at s[f*fs + i*es].  In-place
// radix-2 DIF: X[k] ends up at element brev(k).  W_n^m = tw[m * twStride].
__device__ __forceinline__ void lds_fft(float2 *s, int r, int cnt, int fs, int es, const float2 *tw,
                                        int twStride, int tid, int nth) {
    const int n = 1 << r, halfn = n >> 1;
    for (int st = 0; st < r; ++st) {
        const int half = n >> (st + 1);
        for (int idx = tid; idx < cnt * halfn; idx += nth) {
            const int f = idx % cnt, j = idx / cnt;
            const int pos = j & (half - 1);
            const int i0 = ((j - pos) << 1) + pos;
            const int i1 = i0 + half;
            float2 *p0 = s + f * fs + i0 * es, *p1 = s + f * fs + i1 * es;
            const float2 u = *p0, v = *p1;
            const float2 w = tw[(long long)(pos << st) * twStride];
            const float dx = u.x - v.x, dy = u.y - v.y;
            *p0 = make_float2(u.x + v.x, u.y + v.y);
            *p1 = make_float2(dx * w.x - dy * w.y, dx * w.y + dy * w.x);
        }
        __syncthreads();
    }
}

// W_L^m for 0 <= m < L from the half table tw[0..L/2)
__device__ __forceinline__ float2 twl(const float2 *tw, long long m, long long halfL) {
    if (m >= halfL) {
        const float2 w = tw[m - halfL];
        return make_float2(-w.x, -w.y);
    }
    return tw[m];
}

struct CwtGeom {
    int r1, r2;        // L1 = 2^r1 (columns FFT), L2 = 2^r2 (rows FFT)
    int dataLength;    // 2^r samples
    int pad;
    int C;             // columns per tile
    const float2 *tw;  // W_L^m, m < L/2
};

// forward pass 1: reflect-padded real input -> A[k1][n2] * W_L^(k1 n2)
__global__ void k_cwt_fwd_cols(CwtGeom g, const float *__restrict__ x, float2 *__restrict__ A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *s = reinterpret_cast<float2 *>(smem_raw);
    const int L1 = 1 << g.r1, L2 = 1 << g.r2;
    const long long L = (long long)L1 * L2;
    const int c0 = blockIdx.x * g.C;
    const int tid = threadIdx.x, nth = blockDim.x;
    const int D = g.dataLength, P = g.pad;
    for (int idx = tid; idx < L1 * g.C; idx += nth) {
        const int c = idx % g.C, n1 = idx / g.C;
        const long long n = (long long)n1 * L2 + c0 + c;
        float v;  // cwt_algorithm.c:404-414
        if (n < P) v = x[P - 1 - n];
        else if (n < P + D) v = x[n - P];
        else v = x[D - 1 - (n - P - D)];
        s[n1 * g.C + c] = make_float2(v, 0.f);
    }
    __syncthreads();
    lds_fft(s, g.r1, g.C, 1, g.C, g.tw, L2, tid, nth);  // W_L1^m = W_L^(m L2)
    for (int idx = tid; idx < L1 * g.C; idx += nth) {
        const int c = idx % g.C, k1 = idx / g.C;
        const float2 a = s[brev(k1, g.r1) * g.C + c];
        const float2 w = twl(g.tw, (long long)k1 * (c0 + c), L >> 1);
        A[(long long)k1 * L2 + c0 + c] = make_float2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x);
    }
}

// forward pass 2: rows -> Xt[k1][k2]
__global__ void k_cwt_fwd_rows(CwtGeom g, const float2 *__restrict__ A, float2 *__restrict__ Xt) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *s = reinterpret_cast<float2 *>(smem_raw);
    const int L1 = 1 << g.r1, L2 = 1 << g.r2;
    const long long row = (long long)blockIdx.x * L2;
    const int tid = threadIdx.x, nth = blockDim.x;
    for (int i = tid; i < L2; i += nth) s[i] = A[row + i];
    __syncthreads();
    lds_fft(s, g.r2, 1, 0, 1, g.tw, L1, tid, nth);  // W_L2^m = W_L^(m L1)
    for (int k2 = tid; k2 < L2; k2 += nth) Xt[row + k2] = s[brev(k2, g.r2)];
}

// inverse pass 1: (Xt * wavelet)^* rows -> B[j][k1][m1] * W_L^(m1 k1)
__global__ void k_cwt_inv_rows(CwtGeom g, const float2 *__restrict__ Xt,
                               const float *__restrict__ bankT, int isDet,
                               float2 *__restrict__ B) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *s = reinterpret_cast<float2 *>(smem_raw);
    const int L1 = 1 << g.r1, L2 = 1 << g.r2;
    const long long L = (long long)L1 * L2;
    const int k1 = blockIdx.x, j = blockIdx.y;
    const long long row = (long long)k1 * L2;
    const float *bank = bankT + (long long)j * L + row;
    const int tid = threadIdx.x, nth = blockDim.x;
    for (int i = tid; i < L2; i += nth) {
        const float2 xv = Xt[row + i];
        const float b = bank[i];
        float zr, zi;
        if (!isDet) {  // cwt_algorithm.c:428-431
            zr = b * xv.x;
            zi = b * xv.y;
        } else {       // :432-435
            zr = -b * xv.y;
            zi = b * xv.x;
        }
        s[i] = make_float2(zr, -zi);  // conjugate: IFFT through a forward FFT
    }
    __syncthreads();
    lds_fft(s, g.r2, 1, 0, 1, g.tw, L1, tid, nth);
    float2 *out = B + (long long)j * L + row;
    for (int m1 = tid; m1 < L2; m1 += nth) {
        const float2 a = s[brev(m1, g.r2)];
        const float2 w = twl(g.tw, (long long)m1 * k1, L >> 1);
        out[m1] = make_float2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x);
    }
}

// inverse pass 2: columns over k1 -> time sample n = m1 + L2 m2; conj, 1/L, crop, store
__global__ void k_cwt_inv_cols(CwtGeom g, const float2 *__restrict__ B, float *__restrict__ outRe,
                               float *__restrict__ outIm) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *s = reinterpret_cast<float2 *>(smem_raw);
    const int L1 = 1 << g.r1, L2 = 1 << g.r2;
    const long long L = (long long)L1 * L2;
    const int c0 = blockIdx.x * g.C, j = blockIdx.y;
    const float2 *in = B + (long long)j * L;
    const int tid = threadIdx.x, nth = blockDim.x;
    for (int idx = tid; idx < L1 * g.C; idx += nth) {
        const int c = idx % g.C, k1 = idx / g.C;
        s[k1 * g.C + c] = in[(long long)k1 * L2 + c0 + c];
    }
    __syncthreads();
    lds_fft(s, g.r1, g.C, 1, g.C, g.tw, L2, tid, nth);
    const float invL = 1.f / (float)L;
    const long long D = g.dataLength, P = g.pad;
    for (int idx = tid; idx < L1 * g.C; idx += nth) {
        const int c = idx % g.C, m2 = idx / g.C;
        const long long n = (long long)m2 * L2 + c0 + c;
        if (n >= P && n < P + D) {
            const float2 a = s[brev(m2, g.r1) * g.C + c];
            outRe[(long long)j * D + n - P] = a.x * invL;
            outIm[(long long)j * D + n - P] = -a.y * invL;
        }
    }
}

int lds_opt_in(const void *fn, size_t lds) {
    if (lds > 48 * 1024) AFX_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return AFX_OK;
}

CwtGeom make_geom(const AfxCwtPlanDims *d, const float *tw) {
    CwtGeom g;
    g.r1 = d->r1;
    g.r2 = d->r2;
    g.dataLength = d->dataLength;
    g.pad = d->pad;
    g.C = d->tileCols;
    g.tw = reinterpret_cast<const float2 *>(tw);
    return g;
}

}  // namespace

extern "C" int afxk_cwt_forward(const AfxCwtPlanDims *d, const float *tw, const float *x,
                                float *scratchA, float *Xt, void *stream) {
    const CwtGeom g = make_geom(d, tw);
    const int L1 = 1 << d->r1, L2 = 1 << d->r2;
    const size_t ldsC = (size_t)L1 * d->tileCols * sizeof(float2), ldsR = (size_t)L2 * sizeof(float2);
    int st = lds_opt_in(reinterpret_cast<const void *>(k_cwt_fwd_cols), ldsC);
    if (st == AFX_OK) st = lds_opt_in(reinterpret_cast<const void *>(k_cwt_fwd_rows), ldsR);
    if (st != AFX_OK) return st;
    hipLaunchKernelGGL(k_cwt_fwd_cols, dim3(L2 / d->tileCols), dim3(256), ldsC, (hipStream_t)stream,
                       g, x, reinterpret_cast<float2 *>(scratchA));
    AFX_LAUNCH_CHECK("k_cwt_fwd_cols");
    hipLaunchKernelGGL(k_cwt_fwd_rows, dim3(L1), dim3(256), ldsR, (hipStream_t)stream, g,
                       reinterpret_cast<const float2 *>(scratchA), reinterpret_cast<float2 *>(Xt));
    AFX_LAUNCH_CHECK("k_cwt_fwd_rows");
    return AFX_OK;
}

extern "C" int afxk_cwt_inverse(const AfxCwtPlanDims *d, const float *tw, const float *Xt,
                                const float *bankT, int num, int isDet, float *scratchB,
                                float *outRe, float *outIm, void *stream) {
    const CwtGeom g = make_geom(d, tw);
    const int L1 = 1 << d->r1, L2 = 1 << d->r2;
    const size_t ldsC = (size_t)L1 * d->tileCols * sizeof(float2), ldsR = (size_t)L2 * sizeof(float2);
    int st = lds_opt_in(reinterpret_cast<const void *>(k_cwt_inv_cols), ldsC);
    if (st == AFX_OK) st = lds_opt_in(reinterpret_cast<const void *>(k_cwt_inv_rows), ldsR);
    if (st != AFX_OK) return st;
    hipLaunchKernelGGL(k_cwt_inv_rows, dim3(L1, num), dim3(256), ldsR, (hipStream_t)stream, g,
                       reinterpret_cast<const float2 *>(Xt), bankT, isDet,
                       reinterpret_cast<float2 *>(scratchB));
    AFX_LAUNCH_CHECK("k_cwt_inv_rows");
    hipLaunchKernelGGL(k_cwt_inv_cols, dim3(L2 / d->tileCols, num), dim3(256), ldsC,
                       (hipStream_t)stream, g, reinterpret_cast<const float2 *>(scratchB), outRe, outIm);
    AFX_LAUNCH_CHECK("k_cwt_inv_cols");
    return AFX_OK;
}
