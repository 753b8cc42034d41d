// afx_f16split.h -- float32 operands as (hi, lo) binary16 words for the f16 matrix cores, shared by the CQT octave
// kernels (afx_cqt_f16.hip) and the time-domain CWT kernel (afx_cwt_td.hip):  x 2^e = xh + xl  with e chosen per
// tile so that the tile's peak sits in [2^13, 2^14).  binary16 is a floating-point format: the two words carry
// >= 22 significant bits of EVERY sample down to 2^-17 of the tile's peak (below that the f16 subnormal step,
// 2^-38 of the peak, takes over), products of two 11-bit significands are exact in float32.
#ifndef AFX_F16SPLIT_H
#define AFX_F16SPLIT_H

#include <hip/hip_runtime.h>

#include <afx_asm.h>

namespace {

__device__ __forceinline__ float dpp_f(float v, int ctrl) {
    // all lanes read a lane of their own row: row_mask / bank_mask 0xf, bound_ctrl on
    switch (ctrl) {
        case 0xB1: return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xf, 0xf, true));
        case 0x4E: return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xf, 0xf, true));
        case 0x141: return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xf, 0xf, true));
        default: return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xf, 0xf, true));
    }
}

// (split_pair -- four v_fma_mix in one asm statement -- lives in afx_asm.h)

// wave maximum of a non-negative float without LDS traffic: four DPP steps give every row of 16 lanes its maximum,
// the four rows meet on the scalar unit (non-negative floats order like their bit patterns).  Returns the bits.
__device__ __forceinline__ unsigned wave_max_bits(float peak) {
    peak = fmaxf(peak, dpp_f(peak, 0xB1));   // quad_perm [1,0,3,2]
    peak = fmaxf(peak, dpp_f(peak, 0x4E));   // quad_perm [2,3,0,1]
    peak = fmaxf(peak, dpp_f(peak, 0x141));  // row_half_mirror
    peak = fmaxf(peak, dpp_f(peak, 0x140));  // row_mirror
    const unsigned pk = __float_as_uint(peak);
    const unsigned p01 = max((unsigned)__builtin_amdgcn_readlane((int)pk, 0), (unsigned)__builtin_amdgcn_readlane((int)pk, 16));
    const unsigned p23 = max((unsigned)__builtin_amdgcn_readlane((int)pk, 32), (unsigned)__builtin_amdgcn_readlane((int)pk, 48));
    return max(p01, p23);
}

// exponent e that puts a peak with these bits into [2^13, 2^14) (0 for a zero / subnormal peak; capped at 126)
__device__ __forceinline__ int split_exponent(unsigned peakBits) {
    const int pe = (int)((peakBits >> 23) & 0xff) - 127;  // floor(log2 peak) of a normal
    int e = 13 - pe;
    if (pe == -127) e = 0;
    return e > 126 ? 126 : e;
}

}  // namespace

#endif /* AFX_F16SPLIT_H */
