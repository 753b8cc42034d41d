// tests/emu/hip/afx_asm.h -- the CPU lane emulator's version of audioflux_amd/csrc/hip/afx_asm.h: the same operations
// in C (one host thread per lane, tests/emu/hip/hip_runtime.h).  The emulated builds put this directory ahead of the
// product's on the include path, so the kernels' <afx_asm.h> resolves here.  Test infrastructure.
#ifndef AFX_ASM_H
#define AFX_ASM_H

#include <hip/hip_runtime.h>

typedef float v2 __attribute__((ext_vector_type(2)));

// a + (-i) b = (a.x + b.y, a.y - b.x)
__device__ __forceinline__ v2 pk_add_mi(v2 a, v2 b) {
    v2 r;
    r = v2{a.x + b.y, a.y - b.x};
    return r;
}
// a + i b = (a.x - b.y, a.y + b.x)
__device__ __forceinline__ v2 pk_add_pi(v2 a, v2 b) {
    v2 r;
    r = v2{a.x - b.y, a.y + b.x};
    return r;
}
// a + conj(b) = (a.x + b.x, a.y - b.y)
__device__ __forceinline__ v2 pk_add_conj(v2 a, v2 b) {
    v2 r;
    r = v2{a.x + b.x, a.y - b.y};
    return r;
}
// a - conj(b) = (a.x - b.x, a.y + b.y)
__device__ __forceinline__ v2 pk_sub_conj(v2 a, v2 b) {
    v2 r;
    r = v2{a.x - b.x, a.y + b.y};
    return r;
}
// complex product a * b
// (both instructions in ONE asm statement: the compiler pads every inline-asm VALU result with an
// s_nop before its first use -- it cannot see that the hardware interlocks the dependence)
__device__ __forceinline__ v2 cmul(v2 a, v2 b) {
    v2 t, r;
    t = v2{a.x * b.x, a.x * b.y};
    r = v2{__builtin_fmaf(-a.y, b.y, t.x), __builtin_fmaf(a.y, b.x, t.y)};
    return r;
}
// a * conj(b) = (ax bx + ay by, ay bx - ax by)
__device__ __forceinline__ v2 cmul_conj(v2 a, v2 b) {
    v2 t, r;
    t = v2{a.x * b.x, a.x * b.y};
    r = v2{__builtin_fmaf(a.y, b.y, t.x), __builtin_fmaf(a.y, b.x, -t.y)};
    return r;
}
// complex multiply-accumulate c + a * b: two packed fmas
__device__ __forceinline__ v2 cfma(v2 a, v2 b, v2 c) {
    v2 t, r;
    t = v2{__builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(a.x, b.y, c.y)};
    r = v2{__builtin_fmaf(-a.y, b.y, t.x), __builtin_fmaf(a.y, b.x, t.y)};
    return r;
}
// w * (-i d):  real = w.x d.y + w.y d.x,  imag = w.y d.y - w.x d.x
__device__ __forceinline__ v2 cmul_mi(v2 d, v2 w) {
    v2 t, r;
    t = v2{d.y * w.x, d.y * w.y};
    r = v2{__builtin_fmaf(d.x, w.y, t.x), __builtin_fmaf(-d.x, w.x, t.y)};
    return r;
}
// a.x + a.y
__device__ __forceinline__ float hsum(v2 a) { return a.x + a.y; }
// (-i) a = (a.y, -a.x) as one multiply by the constant pair (1, -1)
__device__ __forceinline__ v2 mul_mi(v2 a) {
    v2 r;
    const v2 c = {1.f, -1.f};
    r = v2{a.y * c.x, a.x * c.y};
    return r;
}


// r[n] = r[n + S] for n + S < 16 (the product's version is one asm statement that moves the registers in place)
template <int S>
__device__ __forceinline__ void shift_rows_inplace(v2 (&r)[16]) {
    for (int n = 0; n + S < 16; ++n) r[n] = r[n + S];
}


template <int S>
__device__ __forceinline__ void shift_rows8_inplace(v2 (&r)[8]) {
    for (int n = 0; n + S < 8; ++n) r[n] = r[n + S];
}

// rows 1024 bytes apart in memory (n_fft 4096): move down by S and refill the last S from p / the whole image from p
template <int S>
__device__ __forceinline__ void rows_shift_fetch(v2 (&r)[16], const float *p) {
    for (int n = 0; n + S < 16; ++n) r[n] = r[n + S];
    for (int i = 0; i < S; ++i) r[16 - S + i] = v2{p[256 * i], p[256 * i + 1]};
}
__device__ __forceinline__ void rows_fetch_all(v2 (&r)[16], const float *p) {
    for (int n = 0; n < 16; ++n) r[n] = v2{p[256 * n], p[256 * n + 1]};
}


// (x0 up, x1 up) -> f16 pair `hi` (round to nearest even) and f16 pair `lo` = f16(x up - hi): four mixed-precision
// fmas (x up is exact: up is a power of two; the subtraction of the f16 word happens inside the fma, one rounding).
// One asm statement: VALU->VALU dependences are interlocked, and hipcc's own form of this costs 7 instructions.
__device__ __forceinline__ void split_pair(float x0, float x1, float up, unsigned &hi, unsigned &lo) {
    const _Float16 h0 = (_Float16)(x0 * up), h1 = (_Float16)(x1 * up);
    const _Float16 l0 = (_Float16)(x0 * up - (float)h0), l1 = (_Float16)(x1 * up - (float)h1);
    unsigned short b[4];
    __builtin_memcpy(&b[0], &h0, 2), __builtin_memcpy(&b[1], &h1, 2), __builtin_memcpy(&b[2], &l0, 2), __builtin_memcpy(&b[3], &l1, 2);
    hi = (unsigned)b[0] | ((unsigned)b[1] << 16);
    lo = (unsigned)b[2] | ((unsigned)b[3] << 16);
}



// ---- LDS: addresses are offsets into the emulation's LDS array of the translation unit (afx_emu_lds); every DS
// instruction is a rendezvous of the wave, as on the device where a wave's DS operations execute in issue order
#define lds_addr(p) ((unsigned)(static_cast<const unsigned char *>(static_cast<const void *>(p)) - afx_emu_lds))
#define RD64(dst, addr, off) (afx_emu_ds(), __builtin_memcpy(&(dst), afx_emu_lds + (addr) + (off), 8), afx_emu_ds())
#define RD128(dst, addr, off) (afx_emu_ds(), __builtin_memcpy(&(dst), afx_emu_lds + (addr) + (off), 16), afx_emu_ds())
#define RD128_P(dst, ptr, off) __builtin_memcpy(&(dst), reinterpret_cast<const char *>(ptr) + (off), 16)
#define RD64_P(dst, ptr, off) __builtin_memcpy(&(dst), reinterpret_cast<const char *>(ptr) + (off), 8)
#define WR2_64(addr, d0, d1, o0, o1) \
    (__builtin_memcpy(afx_emu_lds + (addr) + 8 * (o0), &(d0), 8), __builtin_memcpy(afx_emu_lds + (addr) + 8 * (o1), &(d1), 8))
#define WR2ST_32(addr, d0, d1, o0, o1) \
    (__builtin_memcpy(afx_emu_lds + (addr) + 256 * (o0), &(d0), 4), __builtin_memcpy(afx_emu_lds + (addr) + 256 * (o1), &(d1), 4))
#define PIN(x) ((void)0)
#define LDS_WAIT_N(n) ((void)0)
#define VM_WAIT_ALL() afx_emu_ds()  // rows stored by the other lanes' threads
#define VM_LGKM_WAIT_ALL() ((void)0)
#define LOAD_SC1_B128(dst, ptr) ((dst) = *(ptr))
#define LOAD_B128(dst, ptr) ((dst) = *(ptr))
#define LOAD_B128_SLOT(dst, ptr) __builtin_memcpy(&(dst), (ptr), 16)
static inline long long uniform64(long long x) { return x; }
#define GLD128_S(dst, voff, sbase, off) __builtin_memcpy(&(dst), reinterpret_cast<const char *>(sbase) + (voff) + (off), 16)
#define VM_WAIT_N(n) ((void)0)
#define GST32_S(voff, data, sbase) \
    (*reinterpret_cast<float *>(const_cast<char *>(reinterpret_cast<const char *>(sbase)) + (voff)) = (data))
#define GST32X2_S(voff, d0, sbase0, d1, sbase1) (GST32_S(voff, d0, sbase0), GST32_S(voff, d1, sbase1))

#endif /* AFX_ASM_H */
