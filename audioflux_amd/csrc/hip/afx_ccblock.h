// afx_ccblock.h -- cepstra inside the fused STFT -> filter-bank kernels (afx_melfused{512,1k,2,4k2}.hip): every 16 frames
// a wave re-reads the 16 bank rows it has just stored (L2), rectifies them and multiplies by the first ccNum rows of the
// orthonormal DCT-II with v_mfma_f32_16x16x4_f32 -- xxccObj_xxcc's rectify + DCT + crop (src/feature/xxcc_algorithm.c:
// 95-156: log10f(max(x, 1e-8)) or powf(x, 1/3); fftObj_dct / dctObj_dct, src/dsp/fft_algorithm.c:625-674) without a
// second launch and without the second trip of the rows through HBM.
//
// Two forms.  The headline form (afx_melfused2.hip, num = 128, log) keeps the DCT operand in LDS and is written out
// there.  ccb_rows below is the GENERAL form: any num <= 128 that is a multiple of 4 (rows are read as 16-byte pieces),
// either rectification, whole-row or split band plans; its DCT operand comes from memory (the first 16 rows of the
// [num, num] matrix are <= 8 KB and stay in the vector cache), so the kernels' LDS budgets do not change.
#ifndef AFX_CCBLOCK_H
#define AFX_CCBLOCK_H

#include <hip/hip_runtime.h>

#include <afx_asm.h>

namespace {

typedef float ccb_v4 __attribute__((ext_vector_type(4)));

// rectify: 0 = log10f(max(x, 1e-8)) as v_log_f32 * log10(2); 1 = powf(x, 1/3) as v_exp_f32(v_log_f32(x) / 3): 0 -> 0 and
// negative -> NaN like powf, relative error <= 5e-7 over the range of a power spectrogram (the library call costs ~40
// instructions and, inlined into the frame loops, their registers)
__device__ __forceinline__ float ccb_rect(float x, int cbrt) {
    constexpr float LOG10_2 = 0.30102999566398120f;
    const float lg = __log2f(cbrt ? x : fmaxf(x, 1e-8f));
    return cbrt ? __builtin_amdgcn_exp2f(lg * (float)(1.0 / 3)) : lg * LOG10_2;
}

// cepstra of `cnt` (<= 16) consecutive rows fb .. of this wave: C[16 frames, 16 coefficients] = rect(rows) . D^T.
// Lane (fi = lane & 15, g = lane >> 4) loads row[fb + fi][16 u + 4 g .. + 3] (k-slot g of MFMA (u, c) stands for band
// 16 u + 4 g + c) and the matching elements of DCT row fi.  Bands >= num: the lane re-reads the row's first piece (no
// access past the last row) against zeros of the operand.  Call it one frame AFTER the last of the rows was stored.
// CHAINS: independent accumulator chains (4: a dependent f32 MFMA never waits; 2: eight registers less, for the split-plan
// instantiations that sit at their register cap).
// GROUPS: 16-band groups requested per trip (2: 8 + 8 live registers; 1: 4 + 4).
template <int CHAINS = 4, int GROUPS = 2>
__device__ __forceinline__ void ccb_rows(const float *out, float *cc, const float *dct, int num, int ccNum, int cbrt,
                                         long long fb, int cnt, int lane) {
    VM_WAIT_ALL();  // own stores -> L2 (vmcnt counts stores on gfx9)
    int ln = lane;
    PIN(ln);  // keep this block's per-lane values out of the frame loop's registers
    const int fi = ln & 15, g = ln >> 4;
    const long long r = fb + (fi < cnt ? fi : cnt - 1);  // tail: duplicate the last row, not stored
    const float *src = out + r * num;
    const bool dOn = fi < ccNum;
    const float *drow = dct + (long long)(dOn ? fi : 0) * num;
    static_assert(CHAINS == 2 || CHAINS == 4, "accumulator chains");
    ccb_v4 acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = ccb_v4{0.f, 0.f, 0.f, 0.f};
    const int nu = (num + 15) >> 4;
#pragma unroll 1
    for (int u0 = 0; u0 < nu; u0 += GROUPS) {
        ccb_v4 av[GROUPS], dv[GROUPS];
#pragma unroll
        for (int j = 0; j < GROUPS; ++j) {
            const int band = 16 * (u0 + j) + 4 * g;
            const bool on = band < num;
            LOAD_SC1_B128(av[j], src + (on ? band : 0));  // served by the L2, never by this CU's L1
            const float *dp = drow + (on ? band : 0);
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dv[j]) : "v"(dp) : "memory");
        }
        VM_WAIT_ALL();
#pragma unroll
        for (int j = 0; j < GROUPS; ++j) {
            PIN(av[j]);
            PIN(dv[j]);
            const bool on = dOn && (16 * (u0 + j) + 4 * g) < num;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float lg = ccb_rect(av[j][c], cbrt);
                acc[c % CHAINS] = __builtin_amdgcn_mfma_f32_16x16x4f32(lg, on ? dv[j][c] : 0.f, acc[c % CHAINS], 0, 0, 0);
            }
        }
    }
    const ccb_v4 sum = CHAINS == 4 ? (acc[0] + acc[1]) + (acc[2 % CHAINS] + acc[3 % CHAINS]) : acc[0] + acc[1];
    // C layout: column (coefficient) = lane & 15, row (frame) = 4 (lane >> 4) + reg
    if (dOn) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int rr = 4 * g + reg;
            if (rr < cnt) cc[(fb + rr) * ccNum + fi] = sum[reg];
        }
    }
}

}  // namespace

#endif /* AFX_CCBLOCK_H */
