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

// knock-out measurement builds of the block (make EXTRA=-DAFX_KO_CC=<mask>; results WRONG, timing only; profiles/r06_ab_mfcc.txt):
// bit 0 the row loads (registers as they are), 1 the MFMAs (one v_add per product), 2 the rectification, 3 the wait for the stores
#ifdef AFX_KO_CC
#define CCB_KO(b) (((AFX_KO_CC) >> (b)) & 1)
#else
#define CCB_KO(b) 0
#endif

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

// DCT operand table in LDS (optional): lane (coefficient fi = lane & 15, k-slot g = lane >> 4) holds dct[fi][16 u + 4 g + c] at
// [lane][4 u + c], CCB_PITCH floats per lane (9 x 16 bytes: conflict-free ds_read_b128); zeros for fi >= ccNum and bands >= num.
#ifdef AFX_CC_PLAINLOAD  // (measurement, profiles/r06_ab_headline.txt (b))
constexpr int CCB_ROW_AUX = 0;
#else
constexpr int CCB_ROW_AUX = 17;  // sc0 sc1
#endif
constexpr int CCB_PITCH = 36;
constexpr int CCB_BYTES = 64 * CCB_PITCH * 4;  // 9216
__device__ __forceinline__ void ccb_table_fill(float *tabD, const float *dct, int num, int ccNum, int tid, int nthreads) {
    for (int i = tid; i < 64 * 32; i += nthreads) {
        const int l = i >> 5, e = i & 31;
        const int fi = l & 15, g = l >> 4;
        const int band = 16 * (e >> 2) + 4 * g + (e & 3);
        tabD[l * CCB_PITCH + e] = (fi < ccNum && band < num) ? dct[(long long)fi * num + band] : 0.f;
    }
}

// cepstra of `cnt` (<= 16) consecutive rows fb .. of this wave: C[16 frames, 16 coefficients] = rect(rows) . D^T.
// Lane (fi = lane & 15, g = lane >> 4) loads row[fb + fi][16 u + 4 g .. + 3] (k-slot g of MFMA (u, c) stands for band
// 16 u + 4 g + c) and the matching elements of DCT row fi -- from the LDS table `ldsTab` (LDSD) or from memory.  The rows come
// through a raw buffer over exactly these `cnt` rows with the L2-scope cache policy (served by the L2, never by this CU's
// L1: another lane of the wave stored them a moment ago): compiler-tracked loads -- no hand-placed waits, so the pieces of
// the next trip can be requested before this trip's arithmetic -- and out-of-range pieces (bands >= num of the last row,
// rows past cnt) read as zeros against zeros of the operand.  Call it with the rows' stores issued: the wait in front
// covers them.
// CHAINS: independent accumulator chains (4: a dependent f32 MFMA never waits; 2: eight registers less, for the split-plan
// instantiations that sit at their register cap).  GROUPS: 16-band groups per trip.  AHEAD: the next trip's pieces are requested
// before this trip's arithmetic (4 GROUPS registers more).
template <int CHAINS = 4, int GROUPS = 2, bool LDSD = false, bool AHEAD = true>
__device__ __forceinline__ void ccb_rows(const float *out, float *cc, const float *dct, int num, int ccNum, int cbrt,
                                         long long fb, int cnt, int lane, const float *ldsTab = nullptr) {
    static_assert(CHAINS == 2 || CHAINS == 4, "accumulator chains");
    static_assert(GROUPS == 1 || GROUPS == 2 || GROUPS == 4 || GROUPS == 8, "groups per trip");
    if (!CCB_KO(3)) VM_WAIT_ALL();  // own stores -> L2 (vmcnt counts stores on gfx9)
    int ln = lane;
    PIN(ln);  // keep this block's per-lane values out of the frame loop's registers
    const int fi = ln & 15, g = ln >> 4;
    // (the block is wave-uniform: say so, or the resource is built in vector registers and every load becomes a waterfall loop)
    const unsigned long long base = reinterpret_cast<unsigned long long>(out + fb * num);
    float *const ubase = reinterpret_cast<float *>(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) |
                                                   (unsigned)__builtin_amdgcn_readfirstlane((int)base));
    const __amdgpu_buffer_rsrc_t rows = __builtin_amdgcn_make_buffer_rsrc(ubase, 0, __builtin_amdgcn_readfirstlane(cnt * num * 4), 0x00020000);
    const unsigned rowOff = (unsigned)((fi < cnt ? fi : cnt - 1) * num + 4 * g) * 4u;  // tail: duplicate the last row, not stored
    const bool dOn = fi < ccNum;
    const float *drow = dct + (long long)(dOn ? fi : 0) * num + 4 * g;
    const ccb_v4 *dl = reinterpret_cast<const ccb_v4 *>(ldsTab + CCB_PITCH * ln);
    ccb_v4 acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = ccb_v4{0.f, 0.f, 0.f, 0.f};
    const int nu = (num + 15) >> 4;
    typedef unsigned ccb_u4 __attribute__((ext_vector_type(4)));
    auto row_piece = [&](int u) {  // bands 16 u + 4 g .. + 3 of the lane's row (zeros past the row's end when the row is the last)
        const bool in = 16 * u + 4 * g < num;
        if (CCB_KO(0)) {
            ccb_v4 q;
            asm volatile("" : "=v"(q));
            return q;
        }
        return __builtin_bit_cast(ccb_v4, (ccb_u4)__builtin_amdgcn_raw_buffer_load_b128(rows, in ? (int)(rowOff + 64u * (unsigned)u) : (int)0x80000000, 0, CCB_ROW_AUX));
    };
    ccb_v4 nxt[GROUPS];
    if constexpr (AHEAD) {
#pragma unroll
        for (int j = 0; j < GROUPS; ++j) nxt[j] = row_piece(j);
    }
#pragma unroll 1
    for (int u0 = 0; u0 < nu; u0 += GROUPS) {
        ccb_v4 av[GROUPS], dv[GROUPS];
#pragma unroll
        for (int j = 0; j < GROUPS; ++j) {
            const int u = u0 + j;
            if constexpr (AHEAD) av[j] = nxt[j];
            else av[j] = row_piece(u);
            if constexpr (LDSD) dv[j] = dl[u];
            else dv[j] = (dOn && 16 * u + 4 * g < num) ? *reinterpret_cast<const ccb_v4 *>(drow + 16 * u) : ccb_v4{0.f, 0.f, 0.f, 0.f};
        }
        if constexpr (AHEAD) {
#pragma unroll
            for (int j = 0; j < GROUPS; ++j) nxt[j] = row_piece(u0 + GROUPS + j);  // (past the last group: out of range, zeros)
        }
#pragma unroll
        for (int j = 0; j < GROUPS; ++j) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float lg = CCB_KO(2) ? av[j][c] : ccb_rect(av[j][c], cbrt);
                if (CCB_KO(1)) acc[c % CHAINS][0] += lg * dv[j][c];
                else acc[c % CHAINS] = __builtin_amdgcn_mfma_f32_16x16x4f32(lg, dv[j][c], acc[c % CHAINS], 0, 0, 0);
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
