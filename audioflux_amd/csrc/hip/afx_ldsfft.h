// afx_ldsfft.h -- workgroup-level in-place FFT of 2^r complex points held in LDS, shared by the
// size-generic kernels (afx_stft.hip, afx_cepstrogram.hip, afx_cqt.hip).  Decimation in
// frequency, two radix-2 stages per LDS pass (a radix-4 butterfly computed in registers but
// written back in the radix-2 layout): X[k] ends at s[bitrev_r(k)], with half the barriers and
// half the LDS round trips of one-stage-per-pass.  tw[j * twStride] = W_n^j, n = 2^r.
#ifndef AFX_LDSFFT_H
#define AFX_LDSFFT_H

#include <hip/hip_runtime.h>

// Skewed addressing for the transform buffer: one float2 of padding per 32 keeps the
// power-of-two strides of the late passes and the bit-reversed read-out off a single bank group
// (k_stft_generic at n_fft 4096: 75 % of its LDS cycles were bank conflicts without it).
__device__ __forceinline__ int afx_lds_pad(int i) { return i + (i >> 5); }
__host__ __device__ constexpr int afx_lds_padded_size(int n) { return n + (n >> 5) + 1; }

template <bool PAD>
__device__ __forceinline__ void afx_lds_fft_dif_t(float2 *s, int r, const float2 *tw, int twStride, int tid,
                                                  int nth) {
#define AFX_IX(i) (PAD ? afx_lds_pad(i) : (i))
    const int n = 1 << r;
    int st = 0;
    for (; st + 1 < r; st += 2) {  // stages st and st+1 in one pass
        const int half = n >> (st + 1), half2 = half >> 1;
        for (int j = tid; j < (n >> 2); j += nth) {
            const int p = j & (half2 - 1);
            const int i0 = ((j - p) << 2) + p;  // block base (size 2 half) + p
            const int ia = AFX_IX(i0), ib = AFX_IX(i0 + half2), ic = AFX_IX(i0 + half), id = AFX_IX(i0 + half + half2);
            const float2 va = s[ia], vb = s[ib], vc = s[ic], vd = s[id];
            const int ta = (p << st) * twStride;  // < n/4 * twStride: 32-bit index arithmetic
            const float2 wa = tw[ta];       // W_n^(p << st)
            const float2 w2 = tw[2 * ta];   // W_n^(p << (st+1))
            // stage st: pair (a,c) with twiddle wa, pair (b,d) with twiddle -i wa
            const float2 a1 = make_float2(va.x + vc.x, va.y + vc.y);
            const float2 dc = make_float2(va.x - vc.x, va.y - vc.y);
            const float2 c1 = make_float2(dc.x * wa.x - dc.y * wa.y, dc.x * wa.y + dc.y * wa.x);
            const float2 b1 = make_float2(vb.x + vd.x, vb.y + vd.y);
            const float2 dd = make_float2(vb.x - vd.x, vb.y - vd.y);
            const float2 d0 = make_float2(dd.x * wa.x - dd.y * wa.y, dd.x * wa.y + dd.y * wa.x);
            const float2 d1 = make_float2(d0.y, -d0.x);  // times -i = W_n^(n/4)
            // stage st+1: pairs (a1,b1) and (c1,d1), twiddle w2
            s[ia] = make_float2(a1.x + b1.x, a1.y + b1.y);
            const float2 e1 = make_float2(a1.x - b1.x, a1.y - b1.y);
            s[ib] = make_float2(e1.x * w2.x - e1.y * w2.y, e1.x * w2.y + e1.y * w2.x);
            s[ic] = make_float2(c1.x + d1.x, c1.y + d1.y);
            const float2 e2 = make_float2(c1.x - d1.x, c1.y - d1.y);
            s[id] = make_float2(e2.x * w2.x - e2.y * w2.y, e2.x * w2.y + e2.y * w2.x);
        }
        __syncthreads();
    }
    if (st < r) {  // odd stage count: the last stage (half = 1, twiddle 1)
        for (int j = tid; j < (n >> 1); j += nth) {
            const int i0 = AFX_IX(2 * j), i1 = AFX_IX(2 * j + 1);
            const float2 u = s[i0], v = s[i1];
            s[i0] = make_float2(u.x + v.x, u.y + v.y);
            s[i1] = make_float2(u.x - v.x, u.y - v.y);
        }
        __syncthreads();
    }
#undef AFX_IX
}

__device__ __forceinline__ void afx_lds_fft_dif(float2 *s, int r, const float2 *tw, int twStride, int tid,
                                                int nth) {
    afx_lds_fft_dif_t<false>(s, r, tw, twStride, tid, nth);
}

#endif /* AFX_LDSFFT_H */
