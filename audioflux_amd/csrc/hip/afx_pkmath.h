// afx_pkmath.h -- packed-f32 complex arithmetic and small in-register DFTs shared by the
// wave-level FFT kernels (afx_melfused.hip, afx_cwt.hip).  gfx950 only.
#ifndef AFX_PKMATH_H
#define AFX_PKMATH_H

#include <hip/hip_runtime.h>


// ---- packed-f32 complex primitives -------------------------------------------------------
// VOP3P operand modifiers: op_sel[i] / op_sel_hi[i] pick the half of source i that feeds the
// low / high result lane, neg_lo / neg_hi negate it.  hipcc does not fold a swap+negate into
// these modifiers (it emits v_xor + v_mov per complex multiply), so the three patterns that
// need them are spelled out.  Plain VALU->VALU dependences are hardware-interlocked on gfx9.

// (the eight primitives themselves -- inline assembly -- live in afx_asm.h)
#include <afx_asm.h>

// forward 4-point DFT in place: (p0,p1,p2,p3) -> (X0,X1,X2,X3); 8 v_pk_add_f32
__device__ __forceinline__ void dft4(v2 &p0, v2 &p1, v2 &p2, v2 &p3) {
    const v2 s0 = p0 + p2, s1 = p0 - p2, s2 = p1 + p3, s3 = p1 - p3;
    p0 = s0 + s2;
    p2 = s0 - s2;
    p1 = pk_add_mi(s1, s3);  // s1 - i s3
    p3 = pk_add_pi(s1, s3);  // s1 + i s3
}

// forward 16-point DFT in place, radix-4 x radix-4.  Input x[n] = v[n];
// output X[k] = v[4*(k&3) + (k>>2)]  (base-4 digit reversal).
__device__ __forceinline__ void dft16(v2 (&v)[16]) {
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, H = 0.70710678118654752f;
    const v2 w1 = {C1, -S1}, w3 = {S1, -C1}, w9 = {-C1, S1};
#pragma unroll
    for (int b = 0; b < 4; ++b) dft4(v[b], v[4 + b], v[8 + b], v[12 + b]);
    // t[b][c] sits at v[4c+b]; multiply by W16^(b*c), W16 = exp(-2 pi i / 16)
    v[5] = cmul(v[5], w1);                 // W^1
    v[9] = pk_add_mi(v[9], v[9]) * H;      // W^2 = H(1 - i):  H (x + y, y - x)
    v[13] = cmul(v[13], w3);               // W^3
    v[6] = pk_add_mi(v[6], v[6]) * H;      // W^2
    v[10] = mul_mi(v[10]);                 // W^4 = -i
    v[14] = pk_add_pi(v[14], v[14]) * -H;  // W^6 = -H(1 + i): -H (x - y, x + y)
    v[7] = cmul(v[7], w3);                 // W^3
    v[11] = pk_add_pi(v[11], v[11]) * -H;  // W^6
    v[15] = cmul(v[15], w9);               // W^9
#pragma unroll
    for (int c = 0; c < 4; ++c) dft4(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
}

__host__ __device__ constexpr int rev4(int k) { return 4 * (k & 3) + (k >> 2); }


// forward 8-point DFT in place (two radix-4 halves + one radix-2 layer, 29 packed ops).
// Input x[n] = v[n]; output X[k] = v[rev8(k)], rev8(k) = 2*(k&3) + (k>>2).
__device__ __forceinline__ void dft8(v2 (&v)[8]) {
    constexpr float H = 0.70710678118654752f;
    dft4(v[0], v[2], v[4], v[6]);  // E_k at v[2k]
    dft4(v[1], v[3], v[5], v[7]);  // O_k at v[2k+1]
    v[3] = pk_add_mi(v[3], v[3]) * H;   // W8^1 = H(1 - i)
    v[5] = mul_mi(v[5]);                // W8^2 = -i
    v[7] = pk_add_pi(v[7], v[7]) * -H;  // W8^3 = -H(1 + i)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const v2 e = v[2 * k], o = v[2 * k + 1];
        v[2 * k] = e + o;      // X[k]
        v[2 * k + 1] = e - o;  // X[k + 4]
    }
}
__host__ __device__ constexpr int rev8(int k) { return 2 * (k & 3) + (k >> 2); }

// Orders a wave's own LDS stores before its later LDS loads of other lanes' data.  DS
// operations of one wave execute in issue order; lgkmcnt(0) drains them and the wave barrier
// pins the compiler.  Deliberately NOT a fence: a wavefront-scope fence also emits vmcnt(0),
// which would drain global prefetches and stores at every exchange.
__device__ __forceinline__ void wave_lds_order() {
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0), vmcnt/expcnt untouched
    __builtin_amdgcn_wave_barrier();
}

#endif /* AFX_PKMATH_H */
