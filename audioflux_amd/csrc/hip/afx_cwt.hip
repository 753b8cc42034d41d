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

#include <cstdlib>

#include "afx_device.h"
#include "afx_hipcheck.h"
#include "afx_ldsfft.h"
#include "afx_pkmath.h"

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
    const float2 *fastTw;
    const int *support;  // [num][2] non-zero k2 range per scale, or NULL
    const int *order;    // [num] scale handled by blockIdx.y (+ list base), or NULL = identity
    const int *orderLo;  // [num][2] (scale, first support row) in the same order, narrow-band kernels
    int num;           // scales (chunk stride of the per-scale buffers = num * L)
};

// forward pass 1: reflect-padded real input -> A[k1][n2] * W_L^(k1 n2)
__global__ void k_cwt_fwd_cols(CwtGeom g, const float *__restrict__ x, long long xStride,
                               float2 *__restrict__ A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *s = reinterpret_cast<float2 *>(smem_raw);
    const int L1 = 1 << g.r1, L2 = 1 << g.r2;
    const long long L = (long long)L1 * L2;
    x += (long long)blockIdx.y * xStride;
    A += (long long)blockIdx.y * L;
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
    const long long row = (long long)blockIdx.y * L1 * L2 + (long long)blockIdx.x * L2;
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
    Xt += (long long)blockIdx.z * L;
    B += (long long)blockIdx.z * g.num * L;
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
    const float2 *in = B + ((long long)blockIdx.z * g.num + j) * L;
    outRe += (long long)blockIdx.z * g.num * g.dataLength;
    outIm += (long long)blockIdx.z * g.num * g.dataLength;
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


// ---- register-FFT inverse for L = 2^17 (L1 = 256 columns, L2 = 512 rows) -----------------
// Same two passes and the same intermediate B[j][k1][m1] as the generic kernels, with the
// transforms held in VGPRs (packed-f32 butterflies, afx_pkmath.h) instead of radix-2 passes
// through LDS with a workgroup barrier per stage:
//   rows : one wave per row.  512 = 8 x 8 x 8: lane l holds z[64 a + l], a < 8; radix-8 over a,
//          twiddle W_512^(l d0), exchange, radix-8, twiddle W_64^(c d1), exchange, radix-8.
//          Output m1 = lam + 64 d2 sits in lane lam, register d2: stores are lane-contiguous.
//          (index algebra: tools/proto_fft512.py)
//   cols : 16 columns per workgroup, thread (c = tid & 15, g = tid >> 4) holds B[16 a + g][c],
//          a < 16; 256 = 16 x 16: radix-16 over a, twiddle W_256^(g p), exchange, radix-16.
//          Output m2 = p + 16 q sits in thread (c, p), register q: each store covers 16
//          consecutive time samples (64 B) of 4 rows.
constexpr int RP = 9;  // exchange pitch (float2) of the 64 x 8 images of the row transform

__device__ __forceinline__ v2 ld2(const float2 *p) {
    const float2 t = *p;
    v2 r = {t.x, t.y};
    return r;
}

constexpr int ROWS_PER_WAVE = 4;  // 1 and 2 measure the same (profiles/r01_cwt_narrowband.txt)

__global__ __launch_bounds__(256) void k_cwt_inv_rows512(CwtGeom g, const float2 *__restrict__ Xt,
                                                         const float *__restrict__ bankT, int isDet,
                                                         float2 *__restrict__ B) {
    __shared__ v2 ex[4][64 * RP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int L2 = 512;
    constexpr long long L = 1LL << 17;
    const int j = g.order ? g.order[blockIdx.y] : (int)blockIdx.y;
    // wave w of workgroup b owns rows k1 = 16 b + w + 4 it, it < ROWS_PER_WAVE
    const int k1base = blockIdx.x * (4 * ROWS_PER_WAVE) + wave;
    const float2 *xc = Xt + (long long)blockIdx.z * L;
    const float *bankj = bankT + (long long)j * L;
    float2 *outj = B + ((long long)blockIdx.z * g.num + j) * L;
    v2 *e = ex[wave];

    // The wavelet is zero outside k2 in [lo, hi): a 64-wide block outside that range reads ONE
    // known-zero entry of the row instead of its own 256 bytes (the products are exact zeros
    // either way; the select keeps the code branch-free) -- the bank read of a chunk shrinks
    // from 44 MB to the support of the wavelets.
    const int lo = g.support ? g.support[2 * j] : 0, hi = g.support ? g.support[2 * j + 1] : L2;
    const int zeroAt = lo > 0 ? 0 : (hi < L2 ? L2 - 1 : -1);
    int boff[8];
#pragma unroll
    for (int a = 0; a < 8; ++a)
        boff[a] = (zeroAt >= 0 && (64 * a + 64 <= lo || 64 * a >= hi)) ? zeroAt : 64 * a + lane;

    // twiddle tables of the transform: once per wave
    float2 t1[8], t2[8];
#pragma unroll
    for (int d = 1; d < 8; ++d) {
        t1[d] = g.fastTw[64 * d + lane];                 // W_512^(lane d)
        t2[d] = g.fastTw[8 * 64 + 8 * d + (lane & 7)];   // W_64^(c d)
    }
    // operands of one row: data, wavelet, four-step twiddles (one gathered + eight wave-uniform
    // table values instead of eight gathers); the NEXT row's are requested before this row's
    // butterflies start, so a wave never waits on memory after its first row
    float2 xv[8], wus[8], wlv, xvN[8], wusN[8], wlvN;
    float bw[8], bwN[8];
    auto request = [&](int k1, float2 (&x)[8], float (&bq)[8], float2 (&wu)[8], float2 &wl) {
        const long long row = (long long)k1 * L2;
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            x[a] = xc[row + 64 * a + lane];
            bq[a] = bankj[row + boff[a]];
        }
        wl = g.tw[lane * k1];  // W_L^(lane k1), lane k1 < 2^14 < L/2
#pragma unroll
        for (int d2 = 0; d2 < 8; ++d2) wu[d2] = g.tw[(64 * k1 * d2) & ((1 << 16) - 1)];  // W_L^(m mod L/2)
    };
    request(k1base, xvN, bwN, wusN, wlvN);
#pragma unroll
    for (int it = 0; it < ROWS_PER_WAVE; ++it) {
        const int k1 = k1base + 4 * it;
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            xv[a] = xvN[a];
            bw[a] = bwN[a];
            wus[a] = wusN[a];
        }
        wlv = wlvN;
        if (it + 1 < ROWS_PER_WAVE) request(k1 + 4, xvN, bwN, wusN, wlvN);
        __builtin_amdgcn_sched_barrier(0);

        v2 r[8];
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            // conj(X * wavelet): IFFT through a forward FFT (cwt_algorithm.c:428-435)
            if (!isDet) r[a] = v2{bw[a] * xv[a].x, -(bw[a] * xv[a].y)};
            else r[a] = v2{-bw[a] * xv[a].y, -(bw[a] * xv[a].x)};
        }
        dft8(r);  // r[rev8(d0)] = sum_a z[64 a + l] W_8^(a d0)
#pragma unroll
        for (int d0 = 1; d0 < 8; ++d0) r[rev8(d0)] = cmul(r[rev8(d0)], v2{t1[d0].x, t1[d0].y});
        // exchange 1: lane l = 8 b + c, register d0  ->  lane 8 d0 + c, register b
        {
            const int b = lane >> 3, c = lane & 7;
            wave_lds_order();  // the previous row's reads of the image are done
#pragma unroll
            for (int d0 = 0; d0 < 8; ++d0) e[(8 * d0 + c) * RP + b] = r[rev8(d0)];
            wave_lds_order();
#pragma unroll
            for (int bb = 0; bb < 8; ++bb) r[bb] = e[lane * RP + bb];
        }
        dft8(r);  // r[rev8(d1)], lane = 8 d0 + c
#pragma unroll
        for (int d1 = 1; d1 < 8; ++d1) r[rev8(d1)] = cmul(r[rev8(d1)], v2{t2[d1].x, t2[d1].y});
        // exchange 2: lane 8 d0 + c, register d1  ->  lane d0 + 8 d1, register c
        {
            const int d0 = lane >> 3, c = lane & 7;
            wave_lds_order();  // the reads of exchange 1 are done before the image is overwritten
#pragma unroll
            for (int d1 = 0; d1 < 8; ++d1) e[(d0 + 8 * d1) * RP + c] = r[rev8(d1)];
            wave_lds_order();
#pragma unroll
            for (int cc = 0; cc < 8; ++cc) r[cc] = e[lane * RP + cc];
        }
        dft8(r);  // r[rev8(d2)] = Z[m1 = lane + 64 d2]
        // four-step twiddle W_L^(m1 k1) = W_L^(lane k1) W_L^(64 k1 d2)
        const v2 wl = {wlv.x, wlv.y};
        float2 *out = outj + (long long)k1 * L2;
#pragma unroll
        for (int d2 = 0; d2 < 8; ++d2) {
            const float sgn = ((64 * k1 * d2) >> 16) & 1 ? -1.f : 1.f;  // W_L^(m + L/2) = -W_L^m
            const v2 w = cmul(wl, v2{wus[d2].x * sgn, wus[d2].y * sgn});
            const v2 o = cmul(r[rev8(d2)], w);
            out[64 * d2 + lane] = make_float2(o.x, o.y);
        }
    }
}

// forward pass 2 at L = 2^17 (rows of 512): the same wave-level 8 x 8 x 8 transform as k_cwt_inv_rows512 in place of the
// size-generic radix-2 passes through LDS (k_cwt_fwd_rows: a workgroup barrier per stage, 48 % of its LDS cycles bank
// conflicts -- profiles/r05_ab_cwt.txt).  One wave per row, four rows per wave; X[k2 = lane + 64 d2] leaves lane-contiguous.
__global__ __launch_bounds__(256) void k_cwt_fwd_rows512(CwtGeom g, const float2 *__restrict__ A, float2 *__restrict__ Xt) {
    __shared__ v2 ex[4][64 * RP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int L2 = 512;
    constexpr long long L = 1LL << 17;
    const int k1base = blockIdx.x * (4 * ROWS_PER_WAVE) + wave;
    const float2 *ac = A + (long long)blockIdx.y * L;
    float2 *xo = Xt + (long long)blockIdx.y * L;
    v2 *e = ex[wave];
    float2 t1[8], t2[8];
#pragma unroll
    for (int d = 1; d < 8; ++d) {
        t1[d] = g.fastTw[64 * d + lane];                 // W_512^(lane d)
        t2[d] = g.fastTw[8 * 64 + 8 * d + (lane & 7)];   // W_64^(c d)
    }
    float2 xv[8], xvN[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) xvN[a] = ac[(long long)k1base * L2 + 64 * a + lane];
#pragma unroll
    for (int it = 0; it < ROWS_PER_WAVE; ++it) {
        const int k1 = k1base + 4 * it;
#pragma unroll
        for (int a = 0; a < 8; ++a) xv[a] = xvN[a];
        if (it + 1 < ROWS_PER_WAVE) {
#pragma unroll
            for (int a = 0; a < 8; ++a) xvN[a] = ac[(long long)(k1 + 4) * L2 + 64 * a + lane];
        }
        __builtin_amdgcn_sched_barrier(0);
        v2 r[8];
#pragma unroll
        for (int a = 0; a < 8; ++a) r[a] = v2{xv[a].x, xv[a].y};
        dft8(r);
#pragma unroll
        for (int d0 = 1; d0 < 8; ++d0) r[rev8(d0)] = cmul(r[rev8(d0)], v2{t1[d0].x, t1[d0].y});
        {
            const int b = lane >> 3, c = lane & 7;
            wave_lds_order();
#pragma unroll
            for (int d0 = 0; d0 < 8; ++d0) e[(8 * d0 + c) * RP + b] = r[rev8(d0)];
            wave_lds_order();
#pragma unroll
            for (int bb = 0; bb < 8; ++bb) r[bb] = e[lane * RP + bb];
        }
        dft8(r);
#pragma unroll
        for (int d1 = 1; d1 < 8; ++d1) r[rev8(d1)] = cmul(r[rev8(d1)], v2{t2[d1].x, t2[d1].y});
        {
            const int d0 = lane >> 3, c = lane & 7;
            wave_lds_order();
#pragma unroll
            for (int d1 = 0; d1 < 8; ++d1) e[(d0 + 8 * d1) * RP + c] = r[rev8(d1)];
            wave_lds_order();
#pragma unroll
            for (int cc = 0; cc < 8; ++cc) r[cc] = e[lane * RP + cc];
        }
        dft8(r);  // r[rev8(d2)] = X[k2 = lane + 64 d2]
        float2 *out = xo + (long long)k1 * L2;
#pragma unroll
        for (int d2 = 0; d2 < 8; ++d2) out[64 * d2 + lane] = make_float2(r[rev8(d2)].x, r[rev8(d2)].y);
    }
}

// second half of the 256-point column transform, shared by the two column kernels: r[a] holds
// B[16 a + g][c] of thread (c, g); radix-16, twiddle, exchange, radix-16, conj, 1/L, crop, store
__device__ __forceinline__ void cols256_twiddles(const CwtGeom &g, int gq, float2 (&t3)[16]) {
#pragma unroll
    for (int p = 1; p < 16; ++p) t3[p] = g.fastTw[8 * 64 + 8 * 8 + 16 * p + gq];  // W_256^(g p)
}

// Column block of a 256-thread column-pass workgroup (grid.x = 512 / 16 = 32).  Workgroups go to the eight XCDs round-robin in
// launch order, x fastest: blocks x and x + 1 -- the two 64-byte halves of every 128-byte line of an output row -- would be
// written through two different L2s, each evicting a partial line.  XCD k (x = k, k + 8, k + 16, k + 24) takes the four ADJACENT
// blocks 4 k .. 4 k + 3 instead: 256 contiguous bytes of every row pass through one L2 (+ 1.3 % on cfg 4, profiles/r06_cwt_phases.txt).
__device__ __forceinline__ int col_block16() {
#ifdef AFX_CWT_NO_XCDMAP
    return (int)blockIdx.x;
#else
    const int x = (int)blockIdx.x;
    return 4 * (x & 7) + (x >> 3);
#endif
}

__device__ __forceinline__ void cols256_finish(const CwtGeom &g, v2 (&r)[16], const float2 (&t3)[16], v2 *ex,
                                               int c, int gq, int c0, float *__restrict__ oRe,
                                               float *__restrict__ oIm) {
    constexpr int L2 = 512;
    constexpr long long L = 1LL << 17;
    dft16(r);  // r[rev4(p)] = sum_a B[16 a + g] W_16^(a p)
#pragma unroll
    for (int p = 1; p < 16; ++p) r[rev4(p)] = cmul(r[rev4(p)], v2{t3[p].x, t3[p].y});
#pragma unroll
    for (int p = 0; p < 16; ++p) ex[(p * 16 + gq) * 16 + c] = r[rev4(p)];
    __syncthreads();
    const int p = gq;  // this thread now owns outputs m2 = p + 16 q of column c
#pragma unroll
    for (int gg = 0; gg < 16; ++gg) r[gg] = ex[(p * 16 + gg) * 16 + c];
    dft16(r);  // r[rev4(q)] = Y[m2 = p + 16 q]
    const float invL = 1.f / (float)L;
    // time sample n = (p + 16 q) 512 + column; kept if pad <= n < pad + dataLength (one unsigned
    // compare on n - pad: every quantity is below 2^17)
    const int n0 = p * L2 + c0 + c - g.pad;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int n = n0 + q * (16 * L2);
#ifdef AFX_KO_CWT_STORES  // knock-out measurement build (results wrong): no row is stored (the compiler cannot know: pad >= 0)
        if ((unsigned)n < (unsigned)(g.pad < 0 ? g.dataLength : 0)) {
#else
        if ((unsigned)n < (unsigned)g.dataLength) {  // conj, 1/L, crop (cwt_algorithm.c:449-458)
#endif
            const v2 a = r[rev4(q)];
            oRe[n] = a.x * invL;
            oIm[n] = -a.y * invL;
        }
    }
}

__global__ __launch_bounds__(256) void k_cwt_inv_cols256(CwtGeom g, const float2 *__restrict__ B,
                                                         float *__restrict__ outRe,
                                                         float *__restrict__ outIm) {
    __shared__ v2 ex[16 * 16 * 16];  // [p][g][c]
    constexpr int L2 = 512;
    constexpr long long L = 1LL << 17;
    const int tid = threadIdx.x, c = tid & 15, gq = tid >> 4;
    const int c0 = col_block16() * 16;
    const int j = g.order ? g.order[blockIdx.y] : (int)blockIdx.y;
    const float2 *in = B + ((long long)blockIdx.z * g.num + j) * L + c0 + c;
    v2 r[16];
    float2 t3[16];
#pragma unroll
    for (int a = 0; a < 16; ++a) r[a] = ld2(in + (long long)(16 * a + gq) * L2);
    cols256_twiddles(g, gq, t3);
    __builtin_amdgcn_sched_barrier(0);  // all 31 loads in flight before the first butterfly
    const long long D = g.dataLength;
    cols256_finish(g, r, t3, ex, c, gq, c0, outRe + ((long long)blockIdx.z * g.num + j) * D,
                   outIm + ((long long)blockIdx.z * g.num + j) * D);
}

// forward pass 1 at L = 2^17 (columns of 256): thread (c = tid & 15, g = tid >> 4) holds the reflect-padded samples
// n1 = 16 a + g of column c0 + c; 256 = 16 x 16 in registers with one exchange (the forward half of cols256_finish), then
// the four-step twiddle W_L^(k1 n2).  Replaces the size-generic k_cwt_fwd_cols (eight radix-2 passes through LDS).
__global__ __launch_bounds__(256) void k_cwt_fwd_cols256(CwtGeom g, const float *__restrict__ x, long long xStride,
                                                         float2 *__restrict__ A) {
    __shared__ v2 ex[16 * 16 * 16];  // [p][g][c]
    constexpr int L2 = 512;
    constexpr long long L = 1LL << 17;
    const int tid = threadIdx.x, c = tid & 15, gq = tid >> 4;
    const int c0 = col_block16() * 16, m = c0 + c;
    x += (long long)blockIdx.y * xStride;
    A += (long long)blockIdx.y * L;
    const int D = g.dataLength, P = g.pad;
    float2 t3[16];
    cols256_twiddles(g, gq, t3);
    v2 r[16];
#pragma unroll
    for (int a = 0; a < 16; ++a) {
        const long long n = (long long)(16 * a + gq) * L2 + m;
        float v;  // cwt_algorithm.c:404-414
        if (n < P) v = x[P - 1 - n];
        else if (n < P + D) v = x[n - P];
        else v = x[D - 1 - (n - P - D)];
        r[a] = v2{v, 0.f};
    }
    dft16(r);  // r[rev4(p)] = sum_a s[16 a + g] W_16^(a p)
#pragma unroll
    for (int p = 1; p < 16; ++p) r[rev4(p)] = cmul(r[rev4(p)], v2{t3[p].x, t3[p].y});
#pragma unroll
    for (int p = 0; p < 16; ++p) ex[(p * 16 + gq) * 16 + c] = r[rev4(p)];
    __syncthreads();
    const int p = gq;  // this thread now owns k1 = p + 16 q of column c
#pragma unroll
    for (int gg = 0; gg < 16; ++gg) r[gg] = ex[(p * 16 + gg) * 16 + c];
    dft16(r);  // r[rev4(q)] = S[k1 = p + 16 q]
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int k1 = p + 16 * q;
        const float2 w = twl(g.tw, (long long)k1 * m, L >> 1);
        const v2 o = cmul(r[rev4(q)], v2{w.x, w.y});
        A[(long long)k1 * L2 + m] = make_float2(o.x, o.y);
    }
}

// Four-step twiddles W_L^(m1 (16 a + g)), a = 0 .. 15, from the two LDS tables (W_L^m = tlo[m & 255] thi[m >> 8]) in two
// levels: A[a >> 2] = W_L^(m1 (g + 64 (a >> 2))) and B[a & 3] = W_L^(16 m1 (a & 3)) -- 14 gathered reads and 19 products
// instead of 32 and 16 (these kernels keep the LDS pipe 86 % busy, profiles/r05_ab_cwt.txt); one more rounding per value.
__device__ __forceinline__ void nb_fourstep_twiddles(const v2 *tlo, const v2 *thi, int m1, int gq, v2 (&wl)[16]) {
    v2 A[4], B[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int m = m1 * (gq + 64 * q);  // < 2^17
        A[q] = cmul(tlo[m & 255], thi[m >> 8]);
    }
#pragma unroll
    for (int r = 1; r < 4; ++r) {
        const int m = 16 * m1 * r;
        B[r] = cmul(tlo[m & 255], thi[m >> 8]);
    }
#pragma unroll
    for (int a = 0; a < 16; ++a) wl[a] = (a & 3) ? cmul(A[a >> 2], B[a & 3]) : A[a >> 2];
}

// staged products z[0 .. R) of one row, read two at a time as 16 bytes (ds_read_b128: the rows start 16-byte aligned; read as
// float2 the compiler pairs them into ds_read2_b64, which the LDS serves at HALF the rate -- these kernels are LDS-bound)
typedef float nb_v4 __attribute__((ext_vector_type(4)));
template <int RR>
__device__ __forceinline__ void nb_row(const v2 *z, v2 (&out)[RR]) {
    static_assert(RR % 2 == 0, "pairs");
    const nb_v4 *z4 = reinterpret_cast<const nb_v4 *>(__builtin_assume_aligned(z, 16));
#pragma unroll
    for (int i = 0; i < RR / 2; ++i) {
        const nb_v4 q = z4[i];
        out[2 * i] = v2{q.x, q.y};
        out[2 * i + 1] = v2{q.z, q.w};
    }
}

// Phase clocks of the narrow-band kernels (measurement builds, make EXTRA=-DAFX_EXPERIMENTS; tools/cwt_phases.py): thread 0 of every
// workgroup stamps s_memtime at the phase boundaries and adds the differences to g_nbPhase[log2 R][phase]; [..][7] counts workgroups.
#ifdef AFX_EXPERIMENTS
__device__ unsigned long long g_nbPhase[5][8];
#define NB_CLOCK_DECL unsigned long long nbT[8]; int nbI = 0
#define NB_STAMP()                                                         \
    do {                                                                   \
        if (threadIdx.x == 0) nbT[nbI] = __builtin_amdgcn_s_memtime();     \
        ++nbI;                                                             \
    } while (0)
#define NB_CLOCK_END(R)                                                                                             \
    do {                                                                                                            \
        if (threadIdx.x == 0) {                                                                                     \
            constexpr int lr = (R) == 2 ? 1 : (R) == 4 ? 2 : (R) == 8 ? 3 : 4;                                      \
            for (int i = 0; i + 1 < nbI; ++i) atomicAdd(&g_nbPhase[lr][i], nbT[i + 1] - nbT[i]);                    \
            atomicAdd(&g_nbPhase[lr][7], 1ull);                                                                     \
        }                                                                                                           \
    } while (0)
#else
#define NB_CLOCK_DECL
#define NB_STAMP() ((void)0)
#define NB_CLOCK_END(R) ((void)0)
#endif

// ---- the R-term sums on the f32 matrix pipe (round 6).  Counters of the narrow-band launches: vector unit 55-74 % busy on the CUs
// they own, LDS 35-59 %, matrix pipe idle -- and the sums are 17 (R = 2) ... 60 % (R = 16) of a thread's vector instructions.  As a real
// product per wave: D[row][col] = sum_kk A[row][kk] B[kk][col], A = the staged products (2 R floats per row: re, im interleaved), B =
// the twiddles W_512^((lo + k2) m1) of the 16 columns as [wr, -wi] / [wi, wr] (a "re" and an "im" column tile), v_mfma_f32_16x16x4_f32:
// true float32 products, float32 accumulation (no conversions: the f16 (hi, lo) form of round 4 paid for them on the vector unit).
// Tile T of wave w holds rows k1 = 16 (4 T + (i & 3)) + 4 w + (i >> 2), i = 0 .. 15: the D layout (row 4 (lane >> 4) + reg, column
// lane & 15) then leaves thread (c = lane & 15, g = 4 w + (lane >> 4)) with its own rows 16 a + g, a = 4 T + reg -- no exchange.
// K slot q = lane >> 4 of step s stands for float kk = F q + s of the row (F = R / 2: a lane's A operands are F consecutive floats of
// its row -- one or two 16-byte reads for R = 8 / 16 -- and its B operands need only F / 2 gathered twiddles instead of R).
// LDS layout of the staged rows for R >= 8: the row's 16-byte pieces rotated by rot(k1), so that the 16 lanes of a read group (four
// rows 16 apart x four adjacent rows) hit 16 distinct bank quads without padding.
template <int R>
__device__ __forceinline__ int nb_rot(int k1) {
    return R == 16 ? (((k1 >> 4) & 3) + 4 * ((k1 >> 1) & 1)) : R == 8 ? ((k1 >> 4) & 3) : 0;
}
// float2 index of element (k1, k2) in the staging buffer
template <int R>
__device__ __forceinline__ int nb_slot(int k1, int k2) {
    if (R < 8) return k1 * R + k2;
    constexpr int P = R / 2;  // 16-byte pieces per row
    return k1 * R + 2 * (((k2 >> 1) + nb_rot<R>(k1)) & (P - 1)) + (k2 & 1);
}
typedef float nb_f4 __attribute__((ext_vector_type(4)));
template <int R>
__device__ __forceinline__ void nb_mfma_acc(const v2 *zs, const v2 *thi, int lo, int m1, int lane, int wave, nb_f4 (&accRe)[4], nb_f4 (&accIm)[4]) {
    constexpr int F = R / 2;  // floats of a row per lane = MFMA steps
    const int i = lane & 15, q = lane >> 4;
    // B operands: step s <-> float kk = F q + s of a row <-> (k2 = kk >> 1, re / im)
    float bRe[F], bIm[F];
#pragma unroll
    for (int s = 0; s < F; ++s) {
        const int kk = F * q + s, k2 = kk >> 1;
        const v2 w = thi[((lo + k2) * m1) & 511];  // W_512^((lo + k2) m1)
        bRe[s] = (kk & 1) ? -w.y : w.x;
        bIm[s] = (kk & 1) ? w.x : w.y;
    }
    float A[4][F];
#pragma unroll
    for (int T = 0; T < 4; ++T) {
        const int k1 = 16 * (4 * T + (i & 3)) + 4 * wave + (i >> 2);
        const float *row = reinterpret_cast<const float *>(zs + k1 * R);
        if constexpr (R == 16) {
            const int rot = nb_rot<R>(k1);
            const nb_f4 p0 = *reinterpret_cast<const nb_f4 *>(row + 4 * ((2 * q + rot) & 7));
            const nb_f4 p1 = *reinterpret_cast<const nb_f4 *>(row + 4 * ((2 * q + 1 + rot) & 7));
            A[T][0] = p0.x; A[T][1] = p0.y; A[T][2] = p0.z; A[T][3] = p0.w;
            A[T][4] = p1.x; A[T][5] = p1.y; A[T][6] = p1.z; A[T][7] = p1.w;
        } else if constexpr (R == 8) {
            const nb_f4 p0 = *reinterpret_cast<const nb_f4 *>(row + 4 * ((q + nb_rot<R>(k1)) & 3));
            A[T][0] = p0.x; A[T][1] = p0.y; A[T][2] = p0.z; A[T][3] = p0.w;
        } else if constexpr (R == 4) {
            const v2 p0 = *reinterpret_cast<const v2 *>(row + 2 * q);
            A[T][0] = p0.x; A[T][1] = p0.y;
        } else {
            A[T][0] = row[q];
        }
    }
#pragma unroll
    for (int s = 0; s < F; ++s) {  // eight independent accumulator chains per step
#pragma unroll
        for (int T = 0; T < 4; ++T) {
            accRe[T] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[T][s], bRe[s], accRe[T], 0, 0, 0);
            accIm[T] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[T][s], bIm[s], accIm[T], 0, 0, 0);
        }
    }
}
__device__ __forceinline__ void nb_mfma_zero(nb_f4 (&accRe)[4], nb_f4 (&accIm)[4]) {
#pragma unroll
    for (int T = 0; T < 4; ++T) accRe[T] = accIm[T] = nb_f4{0.f, 0.f, 0.f, 0.f};
}
// accumulators -> the thread's own rows 16 a + g, a = 4 T + reg
__device__ __forceinline__ void nb_mfma_rows(const nb_f4 (&accRe)[4], const nb_f4 (&accIm)[4], v2 (&out)[16]) {
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) out[4 * T + reg] = v2{accRe[T][reg], accIm[T][reg]};
}
template <int R>
__device__ __forceinline__ void nb_sums_mfma(const v2 *zs, const v2 *thi, int lo, int m1, int lane, int wave, v2 (&out)[16]) {
    nb_f4 accRe[4], accIm[4];
    nb_mfma_zero(accRe, accIm);
    nb_mfma_acc<R>(zs, thi, lo, m1, lane, wave, accRe, accIm);
    nb_mfma_rows(accRe, accIm, out);
}

// Narrow-band scales: every non-zero of the wavelet lies in R rows k2 in [lo, lo + R) of the
// transposed spectrum (frequencies k = k1 + 256 k2), so the 512-point row transform of the first
// pass is an R-term sum,
//     B[k1][m1] = W_L^(m1 k1) * sum_{k2} conj(Xt[k1][k2] wavelet[k1][k2]) W_512^(k2 m1),
// which this kernel evaluates on the fly for its 16 columns m1 and feeds straight into the column
// transform: such a scale runs NO row pass and its 1 MB intermediate is neither written nor read
// (HBM traffic per scale and chunk: 2.5 MB -> the 0.5 MB of output).  The R x 256 products are
// staged in LDS with row-contiguous loads; they are the same for the 32 workgroups of a scale and
// are served by L2.  The terms outside a wavelet's support that R rounds up to are exact zeros.
template <int R>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_cwt_inv_cols256_nb(CwtGeom g, const float2 *__restrict__ Xt,
                                                            const float *__restrict__ bankT, int isDet,
                                                            int listBase, float *__restrict__ outRe,
                                                            float *__restrict__ outIm) {
    __shared__ __attribute__((aligned(16))) v2 ex[16 * 16 * 16];  // [p][g][c]
    static_assert(256 * R <= 16 * 16 * 16, "the staged products share the exchange buffer");
    v2 *zs = ex;                       // [k1][k2 - lo], consumed before the exchange starts
    __shared__ v2 tlo[256], thi[512];  // W_L^m (m < 256), W_L^(256 q) (q < 512)
    constexpr int L2 = 512;
    constexpr long long L = 1LL << 17;
    const int tid = threadIdx.x, c = tid & 15, gq = tid >> 4;
    NB_CLOCK_DECL;
    NB_STAMP();  // 0 -> 1: operand and twiddle loads issued and ARRIVED (the LDS stores below wait for them)
    const int c0 = col_block16() * 16, m1 = c0 + c;
    // (scale, first row of its support): one dependent read ahead of the operand loads
    const int2 jl = reinterpret_cast<const int2 *>(g.orderLo)[listBase + blockIdx.y];
    const int j = jl.x;
    int lo = jl.y;
    if (lo + R > L2) lo = L2 - R;
    const float2 *xc = Xt + (long long)blockIdx.z * L + lo;
    const float *bankj = bankT + (long long)j * L + lo;
    float2 xv[R], t3[16];
    float bw[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int idx = tid + 256 * i, k2 = idx & (R - 1), k1 = idx / R;
        xv[i] = xc[k1 * L2 + k2];
        bw[i] = bankj[k1 * L2 + k2];
    }
    // twiddles from two small LDS tables instead of scattered gathers over the 512 KB table
    // (each would touch 64 cache lines per wave): W_L^m = W_L^(m & 255) * W_L^(256 (m >> 8))
    const float2 ta = g.tw[256 * tid], tb = g.tw[tid];  // W_L^(256 i) = W_512^i; i >= 256: -W_512^(i - 256)
    cols256_twiddles(g, gq, t3);
    __builtin_amdgcn_sched_barrier(0);  // every global read of the workgroup is in flight from here
#ifdef AFX_EXPERIMENTS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    NB_STAMP();  // 1 -> 2: products staged in LDS, barrier
    thi[tid] = v2{ta.x, ta.y};
    thi[tid + 256] = v2{-ta.x, -ta.y};
    tlo[tid] = v2{tb.x, tb.y};
#pragma unroll
    for (int i = 0; i < R; ++i) {
        // conj(X * wavelet) (cwt_algorithm.c:428-435): IFFT through a forward FFT
#ifdef AFX_CWT_NB_VALU
        const int slot = tid + 256 * i;
#else
        const int slot = nb_slot<R>((tid + 256 * i) / R, (tid + 256 * i) & (R - 1));
#endif
        zs[slot] = isDet ? v2{-bw[i] * xv[i].y, -(bw[i] * xv[i].x)} : v2{bw[i] * xv[i].x, -(bw[i] * xv[i].y)};
    }
    __syncthreads();
    NB_STAMP();  // 2 -> 3: gathered twiddles + the R-term sums of 16 rows
    v2 wl[16];
    v2 r[16];
#ifdef AFX_CWT_NB_VALU  // (round 5's form, kept for A/B: every thread sums its 16 rows on the vector unit)
    v2 w5[R];
#pragma unroll
    for (int k2 = 0; k2 < R; ++k2) w5[k2] = thi[((lo + k2) * m1) & (L2 - 1)];  // W_512^((lo + k2) m1)
    nb_fourstep_twiddles(tlo, thi, m1, gq, wl);  // W_L^(m1 k1), m1 k1 < L
#pragma unroll
    for (int a = 0; a < 16; ++a) {
        v2 z[R];
        nb_row<R>(zs + (16 * a + gq) * R, z);
        v2 acc = cmul(z[0], w5[0]);
#pragma unroll
        for (int k2 = 1; k2 < R; ++k2) acc = cfma(z[k2], w5[k2], acc);
        r[a] = cmul(acc, wl[a]);
    }
#else
    nb_sums_mfma<R>(zs, thi, lo, m1, tid & 63, __builtin_amdgcn_readfirstlane(tid >> 6), r);
    nb_fourstep_twiddles(tlo, thi, m1, gq, wl);  // W_L^(m1 k1), m1 k1 < L
#pragma unroll
    for (int a = 0; a < 16; ++a) r[a] = cmul(r[a], wl[a]);
#endif
    NB_STAMP();  // 3 -> 4: barrier
    __syncthreads();  // every thread is done with zs before the exchange buffer is written
    NB_STAMP();  // 4 -> 5: column transform (exchange through LDS) and the row stores issued
    const long long D = g.dataLength;
    cols256_finish(g, r, t3, ex, c, gq, c0, outRe + ((long long)blockIdx.z * g.num + j) * D,
                   outIm + ((long long)blockIdx.z * g.num + j) * D);
    NB_STAMP();
#ifdef AFX_EXPERIMENTS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    NB_STAMP();  // 5 -> 6: the stores acknowledged
    NB_CLOCK_END(R);
}

// Scales whose support spans 17 ... 32 rows: the same kernel with the R-term sum taken in two blocks of rows
// (16, then R2 = 4 / 8 / 16) through the one staging buffer -- such a scale has neither a short time kernel (it is
// not a candidate of afx_cwt_td.hip) nor a band narrow enough for the single-block kernel, and its row pass +
// 1 MB intermediate + column pass cost 2.7 x what the R-term sums cost here (profiles/r03_cwt_nb2.txt).
template <int R2>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(R2 >= 8 ? 3 : 4, 4))) void k_cwt_inv_cols256_nb2(CwtGeom g, const float2 *__restrict__ Xt,
                                                             const float *__restrict__ bankT, int isDet,
                                                             int listBase, float *__restrict__ outRe,
                                                             float *__restrict__ outIm) {
    constexpr int R = 16;
    __shared__ __attribute__((aligned(16))) v2 ex[16 * 16 * 16];  // [p][g][c]; before that the staged products of one block of rows
    v2 *zs = ex;
    __shared__ v2 tlo[256], thi[512];  // W_L^m (m < 256), W_L^(256 q) (q < 512)
    constexpr int L2 = 512;
    constexpr long long L = 1LL << 17;
    const int tid = threadIdx.x, c = tid & 15, gq = tid >> 4;
    const int c0 = col_block16() * 16, m1 = c0 + c;
    const int2 jl = reinterpret_cast<const int2 *>(g.orderLo)[listBase + blockIdx.y];
    const int j = jl.x;
    int lo = jl.y;
    if (lo + R + R2 > L2) lo = L2 - R - R2;
    const float2 *xc = Xt + (long long)blockIdx.z * L + lo;
    const float *bankj = bankT + (long long)j * L + lo;
    float2 xv[R], t3[16];
    float bw[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int idx = tid + 256 * i, k2 = idx & (R - 1), k1 = idx / R;
        xv[i] = xc[k1 * L2 + k2];
        bw[i] = bankj[k1 * L2 + k2];
    }
    const float2 ta = g.tw[256 * tid], tb = g.tw[tid];
    cols256_twiddles(g, gq, t3);
    __builtin_amdgcn_sched_barrier(0);
    thi[tid] = v2{ta.x, ta.y};
    thi[tid + 256] = v2{-ta.x, -ta.y};
    tlo[tid] = v2{tb.x, tb.y};
#pragma unroll
    for (int i = 0; i < R; ++i) {
#ifdef AFX_CWT_NB_VALU
        const int slot = tid + 256 * i;
#else
        const int slot = nb_slot<R>((tid + 256 * i) / R, (tid + 256 * i) & (R - 1));
#endif
        zs[slot] = isDet ? v2{-bw[i] * xv[i].y, -(bw[i] * xv[i].x)} : v2{bw[i] * xv[i].x, -(bw[i] * xv[i].y)};
    }
    // second block of rows: requested now, staged once the first block's sums are taken
    float2 xw[R2];
    float bv[R2];
#pragma unroll
    for (int i = 0; i < R2; ++i) {
        const int idx = tid + 256 * i, k2 = idx & (R2 - 1), k1 = idx / R2;
        xw[i] = xc[k1 * L2 + R + k2];
        bv[i] = bankj[k1 * L2 + R + k2];
    }
    __syncthreads();
#ifndef AFX_CWT_NB_VALU  // both blocks of rows into the same accumulators of the f32 matrix pipe (nb_mfma_acc above)
    nb_f4 accRe[4], accIm[4];
    nb_mfma_zero(accRe, accIm);
    nb_mfma_acc<R>(zs, thi, lo, m1, tid & 63, __builtin_amdgcn_readfirstlane(tid >> 6), accRe, accIm);
    __syncthreads();  // every thread is done with the first block
#pragma unroll
    for (int i = 0; i < R2; ++i) {
        const int slot = nb_slot<R2>((tid + 256 * i) / R2, (tid + 256 * i) & (R2 - 1));
        zs[slot] = isDet ? v2{-bv[i] * xw[i].y, -(bv[i] * xw[i].x)} : v2{bv[i] * xw[i].x, -(bv[i] * xw[i].y)};
    }
    __syncthreads();
    v2 r[16];
    {
        v2 wl[16];
        nb_mfma_acc<R2>(zs, thi, lo + R, m1, tid & 63, __builtin_amdgcn_readfirstlane(tid >> 6), accRe, accIm);
        nb_mfma_rows(accRe, accIm, r);
        nb_fourstep_twiddles(tlo, thi, m1, gq, wl);  // W_L^(m1 k1), m1 k1 < L
#pragma unroll
        for (int a = 0; a < 16; ++a) r[a] = cmul(r[a], wl[a]);
    }
#else
    v2 acc[16];
    {
        v2 w5[R];
#pragma unroll
        for (int k2 = 0; k2 < R; ++k2) w5[k2] = thi[((lo + k2) * m1) & (L2 - 1)];  // W_512^((lo + k2) m1)
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            v2 z[R];
            nb_row<R>(zs + (16 * a + gq) * R, z);
            acc[a] = cmul(z[0], w5[0]);
#pragma unroll
            for (int k2 = 1; k2 < R; ++k2) acc[a] = cfma(z[k2], w5[k2], acc[a]);
        }
    }
    __syncthreads();  // every thread is done with the first block
#pragma unroll
    for (int i = 0; i < R2; ++i)
        zs[tid + 256 * i] = isDet ? v2{-bv[i] * xw[i].y, -(bv[i] * xw[i].x)} : v2{bv[i] * xw[i].x, -(bv[i] * xw[i].y)};
    __syncthreads();
    v2 r[16];
    {
        v2 w5[R2], wl[16];
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) w5[k2] = thi[((lo + R + k2) * m1) & (L2 - 1)];
        nb_fourstep_twiddles(tlo, thi, m1, gq, wl);  // W_L^(m1 k1), m1 k1 < L
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            v2 z[R2];
            nb_row<R2>(zs + (16 * a + gq) * R2, z);
#pragma unroll
            for (int k2 = 0; k2 < R2; ++k2) acc[a] = cfma(z[k2], w5[k2], acc[a]);
            r[a] = cmul(acc[a], wl[a]);
        }
    }
#endif
    __syncthreads();  // every thread is done with zs before the exchange buffer is written
    const long long D = g.dataLength;
    cols256_finish(g, r, t3, ex, c, gq, c0, outRe + ((long long)blockIdx.z * g.num + j) * D,
                   outIm + ((long long)blockIdx.z * g.num + j) * D);
}

// ---- transforms that fit one CU's LDS (L <= 16384: the reference wrapper's default sizes) ----
// No four-step split, no HBM intermediate: the forward kernel transforms one reflect-padded chunk
// per workgroup into the natural-order spectrum X[chunk][k]; the inverse kernel takes one
// (scale, chunk) per workgroup -- conj(X * wavelet) in LDS, forward FFT, conjugate, 1/L, crop.
__global__ __launch_bounds__(512) void k_cwt_small_fwd(const float *__restrict__ x, long long xStride, int D,
                                                       int P, int rL, const float2 *__restrict__ tw,
                                                       float2 *__restrict__ X) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *s = reinterpret_cast<float2 *>(smem_raw);
    const int L = 1 << rL, tid = threadIdx.x, nth = blockDim.x;
    x += (long long)blockIdx.x * xStride;
    for (int n = tid; n < L; n += nth) {
        float v;  // cwt_algorithm.c:404-414
        if (n < P) v = x[P - 1 - n];
        else if (n < P + D) v = x[n - P];
        else v = x[D - 1 - (n - P - D)];
        s[afx_lds_pad(n)] = make_float2(v, 0.f);
    }
    __syncthreads();
    afx_lds_fft_dif_t<true>(s, rL, tw, 1, tid, nth);
    float2 *out = X + (long long)blockIdx.x * L;
    for (int k = tid; k < L; k += nth) out[k] = s[afx_lds_pad(brev(k, rL))];
}

__global__ __launch_bounds__(512) void k_cwt_small_inv(const float2 *__restrict__ X,
                                                       const float *__restrict__ bank, int isDet, int D, int P,
                                                       int rL, const float2 *__restrict__ tw, int num,
                                                       float *__restrict__ outRe, float *__restrict__ outIm) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *s = reinterpret_cast<float2 *>(smem_raw);
    const int L = 1 << rL, tid = threadIdx.x, nth = blockDim.x;
    const int j = blockIdx.x, c = blockIdx.y;
    const float2 *xc = X + (long long)c * L;
    const float *bj = bank + (long long)j * L;
    for (int k = tid; k < L; k += nth) {
        const float2 xv = xc[k];
        const float b = bj[k];
        // conj(X * wavelet) (cwt_algorithm.c:428-435): IFFT through a forward FFT
        s[afx_lds_pad(k)] = isDet ? make_float2(-b * xv.y, -(b * xv.x)) : make_float2(b * xv.x, -(b * xv.y));
    }
    __syncthreads();
    afx_lds_fft_dif_t<true>(s, rL, tw, 1, tid, nth);
    const float invL = 1.f / (float)L;
    float *oRe = outRe + ((long long)c * num + j) * D, *oIm = outIm + ((long long)c * num + j) * D;
    for (int n = tid; n < D; n += nth) {
        const float2 a = s[afx_lds_pad(brev(n + P, rL))];
        oRe[n] = a.x * invL;
        oIm[n] = -a.y * invL;
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
    g.fastTw = reinterpret_cast<const float2 *>(d->fastTw);
    g.support = d->support;
    g.order = d->order;
    g.orderLo = d->orderLo;
    g.num = 0;
    return g;
}

}  // namespace


extern "C" int afxk_cwt_small(const AfxCwtPlanDims *d, const float *tw, const float *x, long long xStride,
                              int chunks, const float *bankNatural, int num, int isDet, float *X,
                              float *outRe, float *outIm, void *stream) {
    const int rL = d->r1 + d->r2;
    if (rL > 14 || chunks <= 0) return rL > 14 ? AFX_ERR_UNSUPPORTED : AFX_OK;
    if (chunks > 65535 || num > 0x7fffffff) return AFX_ERR_UNSUPPORTED;
    const int L = 1 << rL;
    const size_t lds = sizeof(float2) * (size_t)afx_lds_padded_size(L);
    int st = lds_opt_in(reinterpret_cast<const void *>(k_cwt_small_fwd), lds);
    if (st == AFX_OK) st = lds_opt_in(reinterpret_cast<const void *>(k_cwt_small_inv), lds);
    if (st != AFX_OK) return st;
    int threads = L / 4;
    if (threads < 64) threads = 64;
    if (threads > 512) threads = 512;
    const float2 *tw2 = reinterpret_cast<const float2 *>(tw);
    if (x) {
        hipLaunchKernelGGL(k_cwt_small_fwd, dim3(chunks), dim3(threads), lds, (hipStream_t)stream, x, xStride,
                           d->dataLength, d->pad, rL, tw2, reinterpret_cast<float2 *>(X));
        AFX_LAUNCH_CHECK("k_cwt_small_fwd");
    }
    hipLaunchKernelGGL(k_cwt_small_inv, dim3(num, chunks), dim3(threads), lds, (hipStream_t)stream,
                       reinterpret_cast<const float2 *>(X), bankNatural, isDet, d->dataLength, d->pad, rL, tw2,
                       num, outRe, outIm);
    AFX_LAUNCH_CHECK("k_cwt_small_inv");
    return AFX_OK;
}

extern "C" int afxk_cwt_forward(const AfxCwtPlanDims *d, const float *tw, const float *x,
                                long long xStride, int chunks, float *scratchA, float *Xt,
                                void *stream) {
    if (chunks <= 0) return AFX_OK;
    if (chunks > 65535) return AFX_ERR_UNSUPPORTED;
    const CwtGeom g = make_geom(d, tw);
    const int L1 = 1 << d->r1, L2 = 1 << d->r2;
    const size_t ldsC = (size_t)L1 * d->tileCols * sizeof(float2), ldsR = (size_t)L2 * sizeof(float2);
    int st = lds_opt_in(reinterpret_cast<const void *>(k_cwt_fwd_cols), ldsC);
    if (st == AFX_OK) st = lds_opt_in(reinterpret_cast<const void *>(k_cwt_fwd_rows), ldsR);
    if (st != AFX_OK) return st;
    if (d->fastTw && d->r1 == 8 && d->r2 == 9 && !afxdev_no_fused()) {
        hipLaunchKernelGGL(k_cwt_fwd_cols256, dim3(L2 / 16, chunks), dim3(256), 0, (hipStream_t)stream, g, x, xStride,
                           reinterpret_cast<float2 *>(scratchA));
        AFX_LAUNCH_CHECK("k_cwt_fwd_cols256");
    } else {
        hipLaunchKernelGGL(k_cwt_fwd_cols, dim3(L2 / d->tileCols, chunks), dim3(256), ldsC,
                           (hipStream_t)stream, g, x, xStride, reinterpret_cast<float2 *>(scratchA));
        AFX_LAUNCH_CHECK("k_cwt_fwd_cols");
    }
    if (d->fastTw && d->r1 == 8 && d->r2 == 9 && !afxdev_no_fused()) {
        hipLaunchKernelGGL(k_cwt_fwd_rows512, dim3(L1 / (4 * ROWS_PER_WAVE), chunks), dim3(256), 0, (hipStream_t)stream, g,
                           reinterpret_cast<const float2 *>(scratchA), reinterpret_cast<float2 *>(Xt));
        AFX_LAUNCH_CHECK("k_cwt_fwd_rows512");
        return AFX_OK;
    }
    hipLaunchKernelGGL(k_cwt_fwd_rows, dim3(L1, chunks), dim3(256), ldsR, (hipStream_t)stream, g,
                       reinterpret_cast<const float2 *>(scratchA), reinterpret_cast<float2 *>(Xt));
    AFX_LAUNCH_CHECK("k_cwt_fwd_rows");
    return AFX_OK;
}

extern "C" int afxk_cwt_inverse(const AfxCwtPlanDims *d, const float *tw, const float *Xt,
                                const float *bankT, int num, int isDet, int chunks, float *scratchB,
                                float *outRe, float *outIm, int parts, void *stream) {
    if (chunks <= 0) return AFX_OK;
    if (chunks > 65535 || num > 65535) return AFX_ERR_UNSUPPORTED;
    CwtGeom g = make_geom(d, tw);
    g.num = num;
    const int L1 = 1 << d->r1, L2 = 1 << d->r2;
    if (d->fastTw && d->r1 == 8 && d->r2 == 9 && !afxdev_no_fused()) {
        const float2 *Xt2 = reinterpret_cast<const float2 *>(Xt);
        float2 *B2 = reinterpret_cast<float2 *>(scratchB);
        hipStream_t s = (hipStream_t)stream;
        // wide scales: row pass -> intermediate -> column pass.  The first nTd entries of the order list are the
        // short-kernel scales the host runs through afxk_cwt_td (afx_cwt_td.hip); a derivative bank (isDet) without
        // time-domain images of its own (d->tdDet) takes both passes here for them.
        const int nTd = d->order ? d->nTd : 0;
        const int skip = (isDet ? d->tdDet : d->td) ? nTd : 0;
        const int nWide = d->order ? (nTd - skip) + d->nWide : num;
        if (nWide > 0 && (parts & AFX_CWT_WIDE)) {
            CwtGeom gw = g;
            if (gw.order) gw.order += skip;
            hipLaunchKernelGGL(k_cwt_inv_rows512, dim3(L1 / (4 * ROWS_PER_WAVE), nWide, chunks), dim3(256), 0, s,
                               gw, Xt2, bankT, isDet, B2);
            AFX_LAUNCH_CHECK("k_cwt_inv_rows512");
            hipLaunchKernelGGL(k_cwt_inv_cols256, dim3(L2 / 16, nWide, chunks), dim3(256), 0, s, gw, B2, outRe,
                               outIm);
            AFX_LAUNCH_CHECK("k_cwt_inv_cols256");
        }
        // narrow-band scales: column pass only, straight from the spectrum
        if (d->order && (parts & AFX_CWT_NARROW)) {
            int base = nTd + d->nWide;
            if (d->nNarrow[0] > 0) {
                hipLaunchKernelGGL(k_cwt_inv_cols256_nb<2>, dim3(L2 / 16, d->nNarrow[0], chunks), dim3(256), 0, s,
                                   g, Xt2, bankT, isDet, base, outRe, outIm);
                AFX_LAUNCH_CHECK("k_cwt_inv_cols256_nb<2>");
            }
            base += d->nNarrow[0];
            if (d->nNarrow[1] > 0) {
                hipLaunchKernelGGL(k_cwt_inv_cols256_nb<4>, dim3(L2 / 16, d->nNarrow[1], chunks), dim3(256), 0, s,
                                   g, Xt2, bankT, isDet, base, outRe, outIm);
                AFX_LAUNCH_CHECK("k_cwt_inv_cols256_nb<4>");
            }
            base += d->nNarrow[1];
            if (d->nNarrow[2] > 0) {
                hipLaunchKernelGGL(k_cwt_inv_cols256_nb<8>, dim3(L2 / 16, d->nNarrow[2], chunks), dim3(256), 0, s,
                                   g, Xt2, bankT, isDet, base, outRe, outIm);
                AFX_LAUNCH_CHECK("k_cwt_inv_cols256_nb<8>");
            }
            base += d->nNarrow[2];
            if (d->nNarrow[3] > 0) {
                hipLaunchKernelGGL(k_cwt_inv_cols256_nb<16>, dim3(L2 / 16, d->nNarrow[3], chunks), dim3(256), 0, s,
                                   g, Xt2, bankT, isDet, base, outRe, outIm);
                AFX_LAUNCH_CHECK("k_cwt_inv_cols256_nb<16>");
            }
            base += d->nNarrow[3];
            if (d->nNarrow[4] > 0) {
                hipLaunchKernelGGL(k_cwt_inv_cols256_nb2<4>, dim3(L2 / 16, d->nNarrow[4], chunks), dim3(256), 0, s,
                                   g, Xt2, bankT, isDet, base, outRe, outIm);
                AFX_LAUNCH_CHECK("k_cwt_inv_cols256_nb2<4>");
            }
            base += d->nNarrow[4];
            if (d->nNarrow[5] > 0) {
                hipLaunchKernelGGL(k_cwt_inv_cols256_nb2<8>, dim3(L2 / 16, d->nNarrow[5], chunks), dim3(256), 0, s,
                                   g, Xt2, bankT, isDet, base, outRe, outIm);
                AFX_LAUNCH_CHECK("k_cwt_inv_cols256_nb2<8>");
            }
            base += d->nNarrow[5];
            if (d->nNarrow[6] > 0) {
                hipLaunchKernelGGL(k_cwt_inv_cols256_nb2<16>, dim3(L2 / 16, d->nNarrow[6], chunks), dim3(256), 0, s,
                                   g, Xt2, bankT, isDet, base, outRe, outIm);
                AFX_LAUNCH_CHECK("k_cwt_inv_cols256_nb2<16>");
            }
        }
        return AFX_OK;
    }
    if (!(parts & AFX_CWT_WIDE)) return AFX_OK;  // the size-generic kernels take every scale in the wide part
    const size_t ldsC = (size_t)L1 * d->tileCols * sizeof(float2), ldsR = (size_t)L2 * sizeof(float2);
    int st = lds_opt_in(reinterpret_cast<const void *>(k_cwt_inv_cols), ldsC);
    if (st == AFX_OK) st = lds_opt_in(reinterpret_cast<const void *>(k_cwt_inv_rows), ldsR);
    if (st != AFX_OK) return st;
    hipLaunchKernelGGL(k_cwt_inv_rows, dim3(L1, num, chunks), dim3(256), ldsR, (hipStream_t)stream, g,
                       reinterpret_cast<const float2 *>(Xt), bankT, isDet,
                       reinterpret_cast<float2 *>(scratchB));
    AFX_LAUNCH_CHECK("k_cwt_inv_rows");
    hipLaunchKernelGGL(k_cwt_inv_cols, dim3(L2 / d->tileCols, num, chunks), dim3(256), ldsC,
                       (hipStream_t)stream, g, reinterpret_cast<const float2 *>(scratchB), outRe, outIm);
    AFX_LAUNCH_CHECK("k_cwt_inv_cols");
    return AFX_OK;
}

#ifdef AFX_EXPERIMENTS
// measurement builds: read and clear the phase clocks of the narrow-band kernels (tools/cwt_phases.py); out[5][8]
extern "C" int afx_cwt_nb_phases(unsigned long long *out) {
    unsigned long long zero[5][8] = {};
    AFX_HIP(hipDeviceSynchronize());
    AFX_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_nbPhase), sizeof(zero)));
    AFX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_nbPhase), zero, sizeof(zero)));
    return AFX_OK;
}
#endif
