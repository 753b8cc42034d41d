// afx_wavefft2048.h -- one 64-lane wave transforms one real sequence of 2048 samples.
//
// The decomposition of the fused STFT -> filter-bank kernel (afx_melfused.hip, section 4.1 of
// DESIGN.md; index algebra: tools/proto_fft1024.py), as a device function with plain LDS
// accesses, for the kernels that run several transforms per frame (afx_cepstrogram.hip):
//   2048 real samples = 1024 complex z[n] = (s[2n], s[2n+1]);  1024 = 16 x 16 x 4
//   radix-16 over n1 in registers -> twiddle W_1024^(lane k1) -> LDS transpose (pitch 68 float2)
//   -> radix-16 over m1 -> twiddle W_64^(m2 j1) -> LDS image V[m2][q] (pitch 260)
//   -> per lane: the radix-4 of bases q = lane, lane + 64 and of their mirrors 256 - q, folded
//      into the real-input split  S[k] = E + W_2048^k O,  S[1024 - k] = conj(E - W_2048^k O).
// gfx950 only.
#ifndef AFX_WAVEFFT2048_H
#define AFX_WAVEFFT2048_H

#include <cmath>

#include "afx_pkmath.h"

namespace afxw {

constexpr int EX_PITCH = 68;            // float2 per k1 row of the first exchange image
constexpr int EX_F2 = 16 * EX_PITCH;    // 1088 float2 per wave; the second image needs 3*260 + 256
constexpr int TAB_TW1_F2 = 16 * 64;     // W_1024^(lane k1)   at [k1][lane]
constexpr int TW2_ROW = 18;             // float2 per row: 16 + 2 (144 bytes, 16-byte aligned: rows m2 and m2 + 2 on different banks;
                                        // 128 bytes is a two-way conflict on every read of the table)
constexpr int TAB_TW2_F2 = 4 * TW2_ROW; // W_64^(m2 j1)       at [m2][j1]
constexpr int TAB_TW3_F2 = 1024;        // 0.5 W_2048^k
constexpr int TAB_F2 = TAB_TW1_F2 + TAB_TW2_F2 + TAB_TW3_F2;

struct Tables {  // LDS-resident
    const v2 *tw1, *tw2, *tw3;
};

// Spectrum of the wave's sequence, spread over the lanes: for s < 2, j < 4 and
// k = lane + 64 s + 256 j:  x[s][j] = S[k],  y[s][j] = conj(S[1024 - k]);
// xc[i] = S[128 + 256 i], yc[i] = conj(S[896 - 256 i]) in every lane (base 128 mirrors itself).
// Lane 0, s = 0 holds k = 0, 256, 512, 768 with partners 1024, 768, 512, 256 (bins 256, 512, 768
// therefore appear twice, equal up to rounding).
struct Bins {
    v2 x[2][4], y[2][4];
    v2 xc[2], yc[2];
};

__device__ __forceinline__ void split(v2 A, v2 B, v2 w /* 0.5 W_2048^k */, v2 &x, v2 &y) {
    const v2 e2 = pk_add_conj(A, B);  // 2 E
    const v2 d = pk_sub_conj(A, B);   // 2 i O
    const v2 wo = cmul_mi(d, w);      // W O
    x = e2 * 0.5f + wo;               // S[k]
    y = e2 * 0.5f - wo;               // conj(S[1024 - k])
}

// v[n1] = (s[2n], s[2n+1]), n = 64 n1 + lane.  `ex` (EX_F2 float2, private to the wave) must be
// free on entry; on return every lane has its values in registers and `ex` is free again.
__device__ __forceinline__ void rfft2048(v2 (&v)[16], v2 *ex, const Tables &t, int lane, Bins &o) {
    const int k1 = lane >> 2, m2 = lane & 3;
    // the transform (dense packed arithmetic between short LDS exchanges) runs above the latency-bound phases of the
    // SIMD's other waves -- window, logarithm / lifters, stores: cepstrogram n_fft 4096 0.497 -> 0.403 ms, n_fft 2048
    // 0.412 -> 0.390 ms, STFT 0.313 -> 0.303 ms per call, same bits (profiles/r04_ab_headline.txt (5))
    __builtin_amdgcn_s_setprio(1);
    dft16(v);
    ex[lane] = v[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) ex[k * EX_PITCH + lane] = cmul(v[rev4(k)], t.tw1[k * 64 + lane]);
    wave_lds_order();
#pragma unroll
    for (int m1 = 0; m1 < 16; ++m1) v[m1] = ex[k1 * EX_PITCH + 4 * m1 + m2];
    wave_lds_order();
    dft16(v);
    ex[m2 * 260 + k1] = v[0];
#pragma unroll
    for (int j1 = 1; j1 < 16; ++j1) ex[m2 * 260 + k1 + 16 * j1] = cmul(v[rev4(j1)], t.tw2[m2 * TW2_ROW + j1]);
    wave_lds_order();
    const int qm = (256 - lane) & 255;  // mirror base of q = lane (lane 0 mirrors itself)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int q = lane + 64 * s;
        const int qp = s == 0 ? qm : 192 - lane;  // (256 - q) & 255
        v2 za0 = ex[q], za1 = ex[260 + q], za2 = ex[520 + q], za3 = ex[780 + q];
        v2 zb0 = ex[qp], zb1 = ex[260 + qp], zb2 = ex[520 + qp], zb3 = ex[780 + qp];
        const v2 w0 = t.tw3[q], w1 = t.tw3[q + 256], w2 = t.tw3[q + 512], w3 = t.tw3[q + 768];
        dft4(za0, za1, za2, za3);  // Z[q + 256 j]
        dft4(zb0, zb1, zb2, zb3);  // Z[qp + 256 j]
        // partner of Z[q + 256 j] is Z[qp + 256 (3 - j)]; for q = 0 it is Z[256 ((4 - j) & 3)]
        v2 b0 = zb3, b1 = zb2, b2 = zb1, b3 = zb0;
        if (s == 0) {
            const bool self = (lane == 0);
            b0 = self ? zb0 : zb3;
            b1 = self ? zb3 : zb2;
            b2 = self ? zb2 : zb1;
            b3 = self ? zb1 : zb0;
        }
        split(za0, b0, w0, o.x[s][0], o.y[s][0]);
        split(za1, b1, w1, o.x[s][1], o.y[s][1]);
        split(za2, b2, w2, o.x[s][2], o.y[s][2]);
        split(za3, b3, w3, o.x[s][3], o.y[s][3]);
    }
    {
        v2 zc0 = ex[128], zc1 = ex[260 + 128], zc2 = ex[520 + 128], zc3 = ex[780 + 128];
        dft4(zc0, zc1, zc2, zc3);
        split(zc0, zc3, t.tw3[128], o.xc[0], o.yc[0]);  // bins 128, 896
        split(zc1, zc2, t.tw3[384], o.xc[1], o.yc[1]);  // bins 384, 640
    }
    wave_lds_order();  // every lane has its bins in registers: ex may be overwritten
    __builtin_amdgcn_s_setprio(0);
}

// ---- N = 4096 ---------------------------------------------------------------------------------
// Spectrum of a 4096-sample real sequence from the transforms of its even and odd samples.  For
// every position of the afxw::Bins layout (k' = lane + 64 s + 256 j, its partner 1024 - k', and
// the base-128 extras) four bins come out: k', 1024 - k', 1024 + k', 2048 - k'.  emit(slot, X)
// receives the spectrum value itself (not conjugated); slots: 4 (4 s + j) + {0: k', 1: 2048 - k',
// 2: 1024 - k', 3: 1024 + k'}, extras 32 + 4 i + {0..3} for k' = 128 + 256 i.
template <typename Emit>
__device__ __forceinline__ void combine4096(const Bins &e, const Bins &o, const v2 *w4, int lane,
                                            Emit emit) {
    auto position = [&](int slot, int kp, v2 xE, v2 xO, v2 yE, v2 yO) {
        const v2 wk = w4[kp], wp = w4[1024 - kp];
        const v2 t = cmul(xO, wk);                    // W^k' O[k']
        const v2 u = cmul(yO, v2{wp.x, -wp.y});       // conj(W^(1024-k')) conj(O[1024-k'])
        const v2 s0 = xE + t, d0 = xE - t, s1 = yE + u, d1 = yE - u;
        emit(slot + 0, s0);                 // X[k']          = E + W O
        emit(slot + 1, v2{d0.x, -d0.y});    // X[2048 - k']   = conj(E - W O)
        emit(slot + 2, v2{s1.x, -s1.y});    // X[1024 - k']   = conj(yE + u)
        emit(slot + 3, d1);                 // X[1024 + k']   = yE - u
    };
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            position(4 * (4 * s + j), lane + 64 * s + 256 * j, e.x[s][j], o.x[s][j], e.y[s][j], o.y[s][j]);
#pragma unroll
    for (int i = 0; i < 2; ++i) position(32 + 4 * i, 128 + 256 * i, e.xc[i], o.xc[i], e.yc[i], o.yc[i]);
}

// bin index of a slot of combine4096 for this lane
__device__ __forceinline__ int bin4096(int slot, int lane) {
    const int p = slot >> 2, r = slot & 3;
    const int kp = p < 8 ? lane + 64 * (p >> 2) + 256 * (p & 3) : 128 + 256 * (p - 8);
    return r == 0 ? kp : r == 1 ? 2048 - kp : r == 2 ? 1024 - kp : 1024 + kp;
}

// host: the three twiddle tables, evaluated in double and rounded once; tab[2 * TAB_F2] floats
inline void fill_tables(float *tab) {
    const double PI = 3.14159265358979323846;
    float *tw1 = tab, *tw2 = tab + 2 * TAB_TW1_F2, *tw3 = tw2 + 2 * TAB_TW2_F2;
    for (int k = 0; k < 16; ++k)
        for (int l = 0; l < 64; ++l) {
            const double ang = -2.0 * PI * (double)(k * l) / 1024.0;
            tw1[2 * (k * 64 + l)] = (float)cos(ang);
            tw1[2 * (k * 64 + l) + 1] = (float)sin(ang);
        }
    for (int m = 0; m < 4; ++m)
        for (int j = 0; j < 16; ++j) {
            const double ang = -2.0 * PI * (double)(m * j) / 64.0;
            tw2[2 * (m * TW2_ROW + j)] = (float)cos(ang);
            tw2[2 * (m * TW2_ROW + j) + 1] = (float)sin(ang);
        }
    for (int k = 0; k < 1024; ++k) {
        const double ang = -2.0 * PI * (double)k / 2048.0;
        tw3[2 * k] = (float)(0.5 * cos(ang));
        tw3[2 * k + 1] = (float)(0.5 * sin(ang));
    }
}

}  // namespace afxw

#endif /* AFX_WAVEFFT2048_H */
