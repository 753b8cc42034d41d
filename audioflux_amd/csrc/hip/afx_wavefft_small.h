// afx_wavefft_small.h -- one 64-lane wave transforms one real sequence of 1024 or 512 samples: the transforms of the fused
// STFT -> filter-bank kernels at those sizes (afx_melfused1k.hip: 512 complex points = 8 x 8 x 8 in eight registers per
// lane; afx_melfused512.hip: 256 = 4 x 4 x 4 x 4 in four; index algebra: tools/proto_fft512.py, tools/proto_fft256.py) as
// device functions with plain LDS accesses, for the kernels that run several transforms per frame (afx_cepstrogram.hip).
// Same contract as afxw::rfft2048 (afx_wavefft2048.h).  gfx950 only.
#ifndef AFX_WAVEFFT_SMALL_H
#define AFX_WAVEFFT_SMALL_H

#include <cmath>

#include "afx_pkmath.h"

namespace afxws {

// S[k] = E + W_N^k O,  S[N/2 - k] = conj(E - W_N^k O) from A = Z[k], B = Z[N/2 - k], w = 0.5 W_N^k
__device__ __forceinline__ void split(v2 A, v2 B, v2 w, v2 &x, v2 &y) {
    const v2 e2 = pk_add_conj(A, B);
    const v2 d = pk_sub_conj(A, B);
    const v2 wo = cmul_mi(d, w);
    x = e2 * 0.5f + wo;  // S[k]
    y = e2 * 0.5f - wo;  // conj(S[N/2 - k])
}

// Spectrum of the wave's sequence, spread over the lanes: for j < NJ and k = lane + 64 j: x[j] = S[k], y[j] =
// conj(S[N/2 - k]) (lane 0, j = 0: S[0] and conj(S[N/2])); xm = S[N/4] in every lane.  Every bin 0 .. N/2 exactly once.
template <int NJ>
struct Bins {
    v2 x[NJ], y[NJ], xm;
};

// ---- N = 1024: 512 complex points, 8 x 8 x 8 ------------------------------------------------------------------------
struct Fft1k {
    static constexpr int N = 1024, M = 512, NR = 8, NJ = 4;
    static constexpr int RP = 10;                 // float2 per row of the two 64 x 8 exchange images
    static constexpr int EX_F2 = 64 * RP;         // 640 float2 per wave; also holds the 512-float2 natural image
    static constexpr int TW1_F2 = 8 * 64;         // W_512^(lane d0) at [d0][lane]
    static constexpr int TW2_F2 = 8 * 8;          // W_64^(c d1) at [d1][c]
    static constexpr int TW3_F2 = 257;            // 0.5 W_1024^k, k <= 256
    static constexpr int TAB_F2 = TW1_F2 + TW2_F2 + TW3_F2 + 1;  // (even count: 16-byte multiple)
    typedef Bins<NJ> B;

    static __device__ __forceinline__ const v2 *tw3_of(const v2 *tab) { return tab + TW1_F2 + TW2_F2; }

    // v[r] = (s[2n], s[2n+1]), n = 64 r + lane.  `ex` (EX_F2 float2, private to the wave) must be free on entry and is free on return.
    static __device__ __forceinline__ void rfft(v2 (&v)[8], v2 *ex, const v2 *tab, int lane, B &o) {
        const v2 *tw1 = tab, *tw2 = tab + TW1_F2, *tw3 = tab + TW1_F2 + TW2_F2;
        const int b = lane >> 3, c = lane & 7;
        __builtin_amdgcn_s_setprio(1);
        dft8(v);  // v[rev8(d0)]
        ex[c * RP + b] = v[0];
#pragma unroll
        for (int d0 = 1; d0 < 8; ++d0) ex[(8 * d0 + c) * RP + b] = cmul(v[rev8(d0)], tw1[d0 * 64 + lane]);
        wave_lds_order();
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = ex[lane * RP + i];
        wave_lds_order();
        dft8(v);  // v[rev8(d1)], lane = 8 d0 + c
        v2 t[8];
        t[0] = v[0];
#pragma unroll
        for (int d1 = 1; d1 < 8; ++d1) t[d1] = cmul(v[rev8(d1)], tw2[d1 * 8 + c]);
#pragma unroll
        for (int d1 = 0; d1 < 8; ++d1) ex[(b + 8 * d1) * RP + c] = t[d1];
        wave_lds_order();
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = ex[lane * RP + i];
        wave_lds_order();
        dft8(v);  // v[rev8(d2)] = Z[lane + 64 d2]
#pragma unroll
        for (int d2 = 0; d2 < 8; ++d2) ex[lane + 64 * d2] = v[rev8(d2)];
        wave_lds_order();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const v2 za = ex[lane + 64 * j];
            v2 zb = ex[(512 - lane - 64 * j) & 511];  // (lane 0, j = 0: Z[0] pairs with itself: S[0] and S[512])
            split(za, zb, tw3[lane + 64 * j], o.x[j], o.y[j]);
        }
        {
            const v2 zm = ex[256];
            v2 ym;
            split(zm, zm, tw3[256], o.xm, ym);
        }
        wave_lds_order();  // every lane has its bins in registers: ex may be overwritten
        __builtin_amdgcn_s_setprio(0);
    }

    static inline void fill_tables(float *tab) {  // tab[2 * TAB_F2] floats, in double, rounded once
        const double PI = 3.14159265358979323846;
        float *tw1 = tab, *tw2 = tab + 2 * TW1_F2, *tw3 = tw2 + 2 * TW2_F2;
        for (int d = 0; d < 8; ++d)
            for (int l = 0; l < 64; ++l) {
                const double ang = -2.0 * PI * (double)(d * l) / 512.0;
                tw1[2 * (d * 64 + l)] = (float)cos(ang);
                tw1[2 * (d * 64 + l) + 1] = (float)sin(ang);
            }
        for (int d = 0; d < 8; ++d)
            for (int c = 0; c < 8; ++c) {
                const double ang = -2.0 * PI * (double)(d * c) / 64.0;
                tw2[2 * (d * 8 + c)] = (float)cos(ang);
                tw2[2 * (d * 8 + c) + 1] = (float)sin(ang);
            }
        for (int k = 0; k <= 256; ++k) {
            const double ang = -2.0 * PI * (double)k / 1024.0;
            tw3[2 * k] = (float)(0.5 * cos(ang));
            tw3[2 * k + 1] = (float)(0.5 * sin(ang));
        }
    }
};

// ---- N = 512: 256 complex points, 4 x 4 x 4 x 4 ---------------------------------------------------------------------
struct Fft512 {
    static constexpr int N = 512, M = 256, NR = 4, NJ = 2;
    static constexpr int RP = 5;                  // float2 per row of the three 64 x 4 transposes
    static constexpr int EX_F2 = 64 * RP;         // 320 float2 per wave; also holds the 256-float2 natural image
    static constexpr int TW1_F2 = 4 * 64;         // W_256^(lane d0) at [d0][lane]
    static constexpr int TW2_F2 = 16 * 4;         // W_64^((4 b + c) q0) at [4 b + c][q0]
    static constexpr int TW3_F2 = 4 * 4;          // W_16^(c q1) at [c][q1]
    static constexpr int TWS_F2 = 129;            // 0.5 W_512^k, k <= 128
    static constexpr int TAB_F2 = TW1_F2 + TW2_F2 + TW3_F2 + TWS_F2 + 1;
    typedef Bins<NJ> B;

    static __device__ __forceinline__ const v2 *tw3_of(const v2 *tab) { return tab + TW1_F2 + TW2_F2 + TW3_F2; }

    // the 256-point complex transform alone: v[r] = z[64 r + lane] -> v[q] = Z[lane + 64 q], and the natural-order image Z[k] in
    // ex[0 .. 255] (still needed by the caller: wave_lds_order() before `ex` is written again).  rfft = cfft + the real-input split;
    // afx_stft256.hip packs TWO real frames into z (re = frame a, im = frame b) and separates their spectra from Z[k], Z[256 - k]
    static __device__ __forceinline__ void cfft(v2 (&v)[4], v2 *ex, const v2 *tab, int lane) {
        const v2 *tw1 = tab, *tw2 = tab + TW1_F2, *tw3 = tw2 + TW2_F2;
        const int hi4 = lane >> 4, mid = (lane >> 2) & 3, low = lane & 3;
        __builtin_amdgcn_s_setprio(1);
        // stage 1: over r, twiddle W_256^(lane d0); lane (a, b, c) -> row 16 d0 + 4 b + c, column a
        dft4(v[0], v[1], v[2], v[3]);
        ex[(lane & 15) * RP + hi4] = v[0];
#pragma unroll
        for (int d0 = 1; d0 < 4; ++d0) ex[(16 * d0 + (lane & 15)) * RP + hi4] = cmul(v[d0], tw1[d0 * 64 + lane]);
        wave_lds_order();
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = ex[lane * RP + i];
        wave_lds_order();
        // stage 2: over a, twiddle W_64^((4 b + c) q0); lane (d0, b, c) -> row 16 d0 + q0 + 4 c, column b
        dft4(v[0], v[1], v[2], v[3]);
        v2 t[4];
        t[0] = v[0];
#pragma unroll
        for (int q0 = 1; q0 < 4; ++q0) t[q0] = cmul(v[q0], tw2[(lane & 15) * 4 + q0]);
#pragma unroll
        for (int q0 = 0; q0 < 4; ++q0) ex[(16 * hi4 + 4 * low + q0) * RP + mid] = t[q0];
        wave_lds_order();
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = ex[lane * RP + i];
        wave_lds_order();
        // stage 3: over b, twiddle W_16^(c q1); lane (d0, c, q0) -> row d0 + 4 q0 + 16 q1, column c
        dft4(v[0], v[1], v[2], v[3]);
        t[0] = v[0];
#pragma unroll
        for (int q1 = 1; q1 < 4; ++q1) t[q1] = cmul(v[q1], tw3[mid * 4 + q1]);
#pragma unroll
        for (int q1 = 0; q1 < 4; ++q1) ex[(hi4 + 4 * low + 16 * q1) * RP + mid] = t[q1];
        wave_lds_order();
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = ex[lane * RP + i];
        wave_lds_order();
        dft4(v[0], v[1], v[2], v[3]);  // stage 4: v[q2] = Z[lane + 64 q2]
#pragma unroll
        for (int q2 = 0; q2 < 4; ++q2) ex[lane + 64 * q2] = v[q2];
        wave_lds_order();
    }

    static __device__ __forceinline__ void rfft(v2 (&v)[4], v2 *ex, const v2 *tab, int lane, B &o) {
        const v2 *tws = tab + TW1_F2 + TW2_F2 + TW3_F2;
        cfft(v, ex, tab, lane);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const v2 zb = ex[(256 - lane - 64 * j) & 255];  // (lane 0, j = 0: Z[0] pairs with itself)
            split(v[j], zb, tws[lane + 64 * j], o.x[j], o.y[j]);
        }
        {
            const v2 zm = ex[128];
            v2 ym;
            split(zm, zm, tws[128], o.xm, ym);
        }
        wave_lds_order();
        __builtin_amdgcn_s_setprio(0);
    }

    static inline void fill_tables(float *tab) {
        const double PI = 3.14159265358979323846;
        float *tw1 = tab, *tw2 = tab + 2 * TW1_F2, *tw3 = tw2 + 2 * TW2_F2, *tws = tw3 + 2 * TW3_F2;
        for (int d = 0; d < 4; ++d)
            for (int l = 0; l < 64; ++l) {
                const double ang = -2.0 * PI * (double)(d * l) / 256.0;
                tw1[2 * (d * 64 + l)] = (float)cos(ang);
                tw1[2 * (d * 64 + l) + 1] = (float)sin(ang);
            }
        for (int r = 0; r < 16; ++r)
            for (int q = 0; q < 4; ++q) {
                const double ang = -2.0 * PI * (double)(r * q) / 64.0;
                tw2[2 * (r * 4 + q)] = (float)cos(ang);
                tw2[2 * (r * 4 + q) + 1] = (float)sin(ang);
            }
        for (int c = 0; c < 4; ++c)
            for (int q = 0; q < 4; ++q) {
                const double ang = -2.0 * PI * (double)(c * q) / 16.0;
                tw3[2 * (c * 4 + q)] = (float)cos(ang);
                tw3[2 * (c * 4 + q) + 1] = (float)sin(ang);
            }
        for (int k = 0; k <= 128; ++k) {
            const double ang = -2.0 * PI * (double)k / 512.0;
            tws[2 * k] = (float)(0.5 * cos(ang));
            tws[2 * k + 1] = (float)(0.5 * sin(ang));
        }
    }
};

}  // namespace afxws

#endif /* AFX_WAVEFFT_SMALL_H */
