// afx_cqt_f16.hip -- one CQT octave on the f16 matrix cores with float32-equivalent operands.
//
// Same linear map as k_cqt_octave_mfma_w (afx_cqt.hip): Q[t][j] = sum_n x_t[n] G_j[n], the octave's
// frames (a Toeplitz view of the signal) times the time-domain image G of the thresholded spectral
// kernels (reference: FFT + sparse spectral product, src/cqt_algorithm.c:951-1041).  The f32 MFMA
// runs at the vector rate (64 cycles per 32x32x2); v_mfma_f32_32x32x16_f16 does 8x the products in
// half the time.  Both operands are therefore split into two f16 words,
//     x 2^e  = xh + xl,   G_j 2^s_j = gh + gl      (power-of-two scaling: exact)
// and the product is accumulated as xh gh + xh gl + xl gh in three float32 accumulators:
//   * e (per 32-frame tile, from the tile's own peak) and s_j (per column, host) put the operand peaks
//     at [2^13, 2^14), so xh + xl carries >= 22 significant bits of every sample that matters and
//     the f16 subnormal step sits 2^-38 below the peak;
//   * products of two 11-bit significands are exact in float32, accumulation is float32;
//   * the dropped term xl gl is 2^-22 of the product.
// Measured against the compiled reference this is as close as the f32 MFMA kernel (1-3e-6 peak-relative:
// both are dominated by the reference's own float32 FFT rounding); tests/test_cqt_gpu.py.
//
// Layout.  B (the image, both words) is prepared by the host in fragment order [word][step][lane][8]
// (afx_cqt.c: afx_cqt_time_kernel_f16) and copied into LDS once per persistent workgroup (64 KB at N = 512).
// Every WAVE owns its 32-frame tiles: it converts its signal window to (xh, xl) in its private LDS
// region and reads A fragments with ds_read_b128: lane (i = lane & 31, g = lane >> 5) of step ks takes
// the 8 samples (t0 + i) hop + 16 ks + 8 g ... + 7.
//   hop >= 16: one 16-byte pad per hop samples puts the 16 rows of every b128 lane group on 16 distinct
//              bank quads (row stride in quads = hop/8 + 1, odd);
//   hop  = 8 : rows are 16 bytes apart, no pad;
//   hop  < 8 : 8/hop copies of the window, copy c shifted by c hop samples, so that row i reads a 16-byte
//              aligned fragment from copy i mod (8/hop); copies are 64 (4 copies) / 128 (2 copies) bytes
//              mod 256 apart -> conflict-free (index algebra and bank check: tools/proto_cqt_f16.py,
//              tests/test_prototypes.py).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "afx_device.h"
#include "afx_hipcheck.h"
#include "afx_pkmath.h"
#include "afx_f16split.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

template <int H>
struct CqF16 {
    static constexpr int N = 512, KS = N / 16;
    static constexpr int COPIES = H >= 8 ? 1 : 8 / H;
    static constexpr bool PAD = H >= 16;
    static constexpr int S = 31 * H + N;             // samples of one tile's window
    static constexpr int NV = (S + 255) / 256;       // float4 loads per lane
    static constexpr int MARGIN = 16;                // bytes in front of a copy (shifted copies start below 0)
    static constexpr int RAW = MARGIN + 2 * (S + 8) + (PAD ? 16 * (S / H + 1) : 0);
    static constexpr int CS = COPIES == 1 ? ((RAW + 15) & ~15)
                                          : ((RAW + 255) & ~255) + (COPIES == 4 ? 64 : 128);  // copy stride
    static constexpr int PART = COPIES * CS;         // bytes of one word plane (xh or xl)
    // the transposed epilogue reuses the region: [32 frames][2 planes][4 pieces][4 floats] = 4 KB
    static constexpr int WAVE_BYTES = 2 * PART > 4096 ? 2 * PART : 4096;
    static constexpr int B_BYTES = 2 * KS * 64 * 16; // both word planes of the image
    // byte offset of sample s inside copy c (s >= 0; multiple of 4 where it is used for stores)
    __host__ __device__ static constexpr int at(int s, int c) {
        return MARGIN + 2 * (s - c * H) + (PAD ? 16 * (s / H) : 0);
    }
    // fragment offset of step ks relative to the lane's base (compile-time immediates)
    __host__ __device__ static constexpr int step(int ks) { return 32 * ks + (PAD ? 16 * ((16 * ks) / H) : 0); }
};

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
constexpr int RSRC_RAW = 0x00020000;  // raw buffer, 32-bit data format (cdna_hip_programming.md T8)

// ---- the pieces of one 32-frame tile, shared by the per-octave kernel and the pyramid kernel ----

// exponent e of a tile: its window's peak -> [2^13, 2^14)
template <int H>
__device__ __forceinline__ int cq_window_exponent(const u32x4 (&wnd)[CqF16<H>::NV], int lane) {
    using C = CqF16<H>;
    float peak = 0.f;
#pragma unroll
    for (int u = 0; u < C::NV; ++u) {
        float4 v = __builtin_bit_cast(float4, wnd[u]);
        if (256 * (u + 1) > C::S) {  // the last register reaches past the window: those samples are not the tile's
            const int s = 4 * (lane + 64 * u);
            if (s >= C::S) v.x = 0.f;
            if (s + 1 >= C::S) v.y = 0.f;
            if (s + 2 >= C::S) v.z = 0.f;
            if (s + 3 >= C::S) v.w = 0.f;
        }
        peak = fmaxf(peak, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    // wave maximum without LDS traffic: four DPP steps give every row of 16 lanes its maximum, the four rows meet
    // on the scalar unit (non-negative floats order like their bit patterns)
    return split_exponent(wave_max_bits(peak));
}

// window registers (lane l, register u: samples 4 (l + 64 u) .. + 3) -> (xh, xl) planes, every copy.  One address
// register: sample s = 4 l + 256 u sits at at(4 l, 0) + u (512 + 16 * 256 / H) (the pad count of 256 u samples is a
// constant), so every store has an immediate offset -- as per-u addresses the compiler keeps 18 registers alive across
// the tile and spills them.
template <int H>
__device__ __forceinline__ void cq_convert_window(const u32x4 (&wnd)[CqF16<H>::NV], float up, unsigned char *sig, int lane) {
    using C = CqF16<H>;
    constexpr int USTEP = 512 + (C::PAD ? 16 * (256 / H) : 0);
    unsigned char *d0 = sig + C::MARGIN + 8 * lane + (C::PAD ? 16 * ((4 * lane) / H) : 0);
#pragma unroll
    for (int u = 0; u < C::NV; ++u) {
        const int s = 4 * (lane + 64 * u);
        if (s < C::S) {
            const float4 v = __builtin_bit_cast(float4, wnd[u]);
            unsigned hi0, hi1, lo0, lo1;
            split_pair(v.x, v.y, up, hi0, lo0);
            split_pair(v.z, v.w, up, hi1, lo1);
#pragma unroll
            for (int c = 0; c < C::COPIES; ++c) {
                unsigned char *d = d0 + (u * USTEP + c * C::CS - 2 * c * H);
                if ((2 * c * H) % 8 == 0) {
                    *reinterpret_cast<uint2 *>(d) = make_uint2(hi0, hi1);
                    *reinterpret_cast<uint2 *>(d + C::PART) = make_uint2(lo0, lo1);
                } else {
                    reinterpret_cast<unsigned *>(d)[0] = hi0;
                    reinterpret_cast<unsigned *>(d)[1] = hi1;
                    reinterpret_cast<unsigned *>(d + C::PART)[0] = lo0;
                    reinterpret_cast<unsigned *>(d + C::PART)[1] = lo1;
                }
            }
        }
    }
}

// K loop: 32 steps x (xh gh, xh gl, xl gh), operands two steps ahead; nothing but MFMAs and DS reads at
// immediate offsets between the two scheduling barriers
template <int H>
__device__ __forceinline__ void cq_kloop(const unsigned char *aHi, const unsigned char *aLo, const unsigned char *bHi,
                                         const unsigned char *bLo, f32x16 &hh, f32x16 &hl, f32x16 &lh) {
    using C = CqF16<H>;
#pragma unroll
    for (int r = 0; r < 16; ++r) hh[r] = hl[r] = lh[r] = 0.f;
    h8 ah[3], al[3], bh[3], bl[3];
    auto load = [&](int ks, int slot) {
        ah[slot] = *reinterpret_cast<const h8 *>(aHi + C::step(ks));
        al[slot] = *reinterpret_cast<const h8 *>(aLo + C::step(ks));
        bh[slot] = *reinterpret_cast<const h8 *>(bHi + 1024 * ks);
        bl[slot] = *reinterpret_cast<const h8 *>(bLo + 1024 * ks);
    };
    load(0, 0);
    load(1, 1);
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
        __builtin_amdgcn_sched_barrier(0);
        if (ks + 2 < C::KS) load(ks + 2, (ks + 2) % 3);
        const int sl = ks % 3;
        hh = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[sl], bh[sl], hh, 0, 0, 0);
        hl = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[sl], bl[sl], hl, 0, 0, 0);
        lh = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[sl], bh[sl], lh, 0, 0, 0);
        if (ks + 2 < C::KS) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // DS read
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

// What a lane needs of an octave to scale and place its results (12 bins per octave or the generic column form)
struct CqLane {
    float colMul;                   // 2^-s_j sqrt(2^k) / sqrt(len_j) of the lane's image column
    unsigned voffRe, voffIm;        // generic epilogue: lane = column, 16 rows
    unsigned char *epiW;            // transposed epilogue (12 bins per octave): where the lane's 16 values go ...
    const unsigned char *epiR;      // ... and where its four 12-byte pieces come from
    unsigned voff12;
    unsigned rowBytes, planeBytes;
};

__device__ __forceinline__ CqLane cq_lane_setup(int lane, unsigned char *sig, int rows, int colBase, int num, int timeLength,
                                                const float *colMul, const float *scale, float octScale) {
    const int i = lane & 31, g = lane >> 5;
    CqLane L;
    const bool colOk = i < 2 * rows, colIm = i >= rows;
    const int colOff = colBase + (colOk ? (colIm ? i - rows : i) : 0);
    L.colMul = colMul[i] * (octScale / scale[colOff]);
    const unsigned OOR = 0x80000000u;
    L.planeBytes = (unsigned)timeLength * (unsigned)num * 4u;
    L.rowBytes = (unsigned)num * 4u;
    const unsigned laneOff = (unsigned)(4 * g * num + colOff) * 4u;
    L.voffRe = (colOk && !colIm) ? laneOff : OOR;
    L.voffIm = (colOk && colIm) ? laneOff : OOR;
    // the lane's 16 values go to epi[frame][plane][piece][word] ...
    const int jj = i < 12 ? i : i - 12;
    L.epiW = sig + (i < 24 ? (i >= 12 ? 64 : 0) + (jj / 3) * 16 + (jj % 3) * 4 : (i - 24) * 16 + 12) + 4 * g * 128;
    // ... and leave as: store q, lane L -> frame 16 (q >> 1) + (L >> 2), plane q & 1, bins 3 (L & 3) .. + 2
    L.epiR = sig + (lane >> 2) * 128 + (lane & 3) * 16;
    L.voff12 = (unsigned)(lane >> 2) * L.rowBytes + (unsigned)(colBase + 3 * (lane & 3)) * 4u;
    return L;
}

// results of one tile -> memory.  D layout: col = lane & 31, row = (r&3) + 8 (r>>2) + 4 (lane>>5).  Output: per clip one
// raw buffer per plane, T x num floats; rows past timeLength and the padding columns fall out of range and are dropped
// by the bounds check, so every tile issues the same number of stores.
template <bool R12>
__device__ __forceinline__ void cq_store_tile(const f32x16 &hh, const f32x16 &hl, const f32x16 &lh, float down, const CqLane &L,
                                              float *outRe, float *outIm, int t0) {
    const __amdgpu_buffer_rsrc_t rRe = __builtin_amdgcn_make_buffer_rsrc(outRe, 0, (int)L.planeBytes, RSRC_RAW);
    const __amdgpu_buffer_rsrc_t rIm = __builtin_amdgcn_make_buffer_rsrc(outIm, 0, (int)L.planeBytes, RSRC_RAW);
    const float mul = down * L.colMul;
    const unsigned tileOff = (unsigned)t0 * L.rowBytes;
    if (R12) {
        wave_lds_order();  // the last fragment reads are done: the window region is free
#pragma unroll
        for (int r = 0; r < 16; ++r)
            *reinterpret_cast<float *>(L.epiW + ((r & 3) + 8 * (r >> 2)) * 128) = (hh[r] + (hl[r] + lh[r])) * mul;
        wave_lds_order();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const u32x4 v = *reinterpret_cast<const u32x4 *>(L.epiR + (q >> 1) * 2048 + (q & 1) * 64);
            const u32x3 v3 = {v.x, v.y, v.z};
            __builtin_amdgcn_raw_buffer_store_b96(v3, (q & 1) ? rIm : rRe, L.voff12 + tileOff + (unsigned)(q >> 1) * 16u * L.rowBytes, 0, 0);
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned ro = tileOff + (unsigned)((r & 3) + 8 * (r >> 2)) * L.rowBytes;  // scalar
            const unsigned v = __float_as_uint((hh[r] + (hl[r] + lh[r])) * mul);
            __builtin_amdgcn_raw_buffer_store_b32(v, rRe, L.voffRe + ro, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(v, rIm, L.voffIm + ro, 0, 0);
        }
    }
}

// image -> LDS: all of a thread's 16-byte loads in flight, then the stores
__device__ __forceinline__ void cq_image_to_lds(const unsigned short *timeKernelH, unsigned char *Bl, int tid, int nth) {
    const float4 *src = reinterpret_cast<const float4 *>(timeKernelH);
    float4 *dstl = reinterpret_cast<float4 *>(Bl);
    constexpr int Q = CqF16<128>::B_BYTES / 16;  // 4096
    for (int e0 = tid; e0 < Q; e0 += 8 * nth) {
        float4 tq[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (e0 + u * nth < Q) tq[u] = src[e0 + u * nth];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (e0 + u * nth < Q) dstl[e0 + u * nth] = tq[u];
    }
}

// R12: 12 bins per octave (the default ladder) -> the tile's results are transposed through LDS and leave as four
// 12-byte-per-lane stores (one lane = 3 consecutive bins of one frame and plane); otherwise 32 dword stores.
template <int H, bool R12>
// hop 128: 19 KB of window planes per wave leave room for four waves (one per SIMD, up to 512 VGPRs)
__global__ __launch_bounds__(H >= 128 ? 256 : 512) void k_cqt_octave_f16(AfxCqtOctaveArgs a, int tilesPerClip) {
    using C = CqF16<H>;
    constexpr int NSTORE = R12 ? 4 : 32;  // memory stores per tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, nth = blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), waves = nth >> 6;
    unsigned char *Bl = smem_raw;                                  // [2][KS][64] x 16 bytes
    unsigned char *sig = smem_raw + C::B_BYTES + wave * C::WAVE_BYTES;
    const int i = lane & 31, g = lane >> 5;
    cq_image_to_lds(a.timeKernelH, Bl, tid, nth);
    __syncthreads();

    const CqLane L = cq_lane_setup(lane, sig, a.rows, a.colBase, a.num, a.timeLength, a.colMul, a.scale, a.octScale);
    const unsigned OOR = 0x80000000u;
    // A fragment base of this lane: row i of copy i mod COPIES, first step
    const int cpy = i % C::COPIES;
    const unsigned char *aHi = sig + cpy * C::CS + C::at(i * H + 8 * g, cpy);
    const unsigned char *aLo = aHi + C::PART;
    const unsigned char *bHi = Bl + lane * 16;
    const unsigned char *bLo = bHi + C::KS * 64 * 16;

    const int totalTiles = tilesPerClip * a.batch;
    const int stride = gridDim.x * waves;
    // Input: one raw buffer per clip holding the validLength framed samples; the zero padding in front of
    // the first frame (negative positions wrap to huge offsets) and everything past validLength read as 0
    // (src/stft_algorithm.c:650-653: samples past validLength are dropped by the reference's padded framing).
    u32x4 wnd[C::NV];
    auto fetch = [&](int t) {
        const int clip = t / tilesPerClip, t0 = (t - clip * tilesPerClip) * 32;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(a.x + (long long)clip * a.xStride), 0, a.validLength * 4, RSRC_RAW);
        const int p0 = t0 * H - (a.rightPad ? 0 : (C::N >> 1));
#pragma unroll
        for (int u = 0; u < C::NV; ++u)
            wnd[u] = __builtin_amdgcn_raw_buffer_load_b128(rx, (p0 + 4 * (lane + 64 * u)) * 4, 0, 0);
    };
    int t = blockIdx.x * waves + wave;  // wave-uniform (SGPR)
    if (t < totalTiles) fetch(t);
    // NSTORE out-of-range (dropped) stores behind the first window: the loop is then entered with the same
    // count of memory operations in flight as on its back edge (window loads, then a tile's stores), and
    // the compiler's wait for the window becomes vmcnt(NSTORE + ...) instead of a wait for the previous tile's
    // stores to be acknowledged
    {
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(a.outRe, 0, 4, RSRC_RAW);
#pragma unroll
        for (int r = 0; r < NSTORE; ++r) __builtin_amdgcn_raw_buffer_store_b32(0u, rd, OOR + 4u * r, 0, 0);  // distinct: not merged
    }
    for (; t < totalTiles; t += stride) {
        const int clip = t / tilesPerClip, t0 = (t - clip * tilesPerClip) * 32;
        // ---- tile exponent: peak of the window -> [2^13, 2^14)
        const int e = cq_window_exponent<H>(wnd, lane);
        const float up = __uint_as_float((unsigned)(e + 127) << 23);      // 2^e
        const float down = __uint_as_float((unsigned)(127 - e) << 23);    // 2^-e
        // ---- window -> (xh, xl) planes (every copy)
        wave_lds_order();  // the fragment / epilogue reads of the previous tile are done
        cq_convert_window<H>(wnd, up, sig, lane);
        wave_lds_order();
        if (t + stride < totalTiles) fetch(t + stride);
        f32x16 hh, hl, lh;
        cq_kloop<H>(aHi, aLo, bHi, bLo, hh, hl, lh);
        const long long po = (long long)clip * a.outStride;
        cq_store_tile<R12>(hh, hl, lh, down, L, a.outRe + po, a.outIm + po, t0);
    }
}

// ================================================================================================================
// k_cqt_pyramid: the default ladder (N = 512, 12 bins per octave, seven octaves, hop 128 ... 2) in ONE persistent
// launch.  Reference: the octave recursion of _cqtObj_cqt (src/cqt_algorithm.c:951-1048) -- per octave frame + FFT +
// sparse kernel product, then the 2:1 "Fast" resampler (src/dsp/resample_algorithm.c:430-521) makes the next level.
//
// A workgroup of eleven waves walks a run of 32-frame tiles of one clip, one STEP per tile:
//   * waves 0-6, CONSUMERS: wave k owns octave level k (hop 128 >> k).  Per step: the level's window -> (xh, xl) planes
//     in the wave's own LDS region -> 96 MFMAs -> the tile's 12 bins of 32 rows to memory; the next window is already
//     on its way into registers while the matrix cores work.  Exactly the tile of k_cqt_octave_f16 above.
//   * waves 7-10, PRODUCERS: the decimation chain, level k -> k+1 in BLOCKS of 32 hop_(k+1) samples (what a tile of
//     level k+1 advances by): the 63-tap FIR of k_cqt_decimate (afx_cqt.hip), same taps in the same order, 256 outputs
//     per round from even/odd copies of the input in the wave's LDS region.  Stage 0 (clip -> level 1) takes two
//     waves, stage 1 one, stages 2-5 share the last.
// The level signals live in per-workgroup RINGS in global memory (8192 ... 1024 samples per level, 68 KB per
// workgroup): written once, read a step or two later by the same CU, overwritten four blocks on -- they stay in the
// L2 and never reach HBM, a clip is read from HBM once.  Ring loads bypass the CU's L1 (sc0 sc1), ring stores are
// drained (vmcnt(0)) before the step's s_barrier: ONE barrier per step is the only synchronisation.
//
// Schedule (step s; "block b of level k" = samples [32 b hop_k, 32 (b+1) hop_k)):
//   stage k writes block s - 2k of level k+1; it reads blocks b-1 .. b+1 of level k (the clip for k = 0), whose last
//   one was written in step s - 1;
//   consumer k >= 1 computes tile s - 2k - e_k (e = 1,1,1,1,2,4: the blocks a window reaches past its tile), whose
//   window it fetched during step s - 1 from blocks written up to step s - 2; consumer 0 reads the clip and follows
//   consumer 1.  A ring of 4 blocks (8 / 16 for the two smallest levels) holds everything alive at a step.
// A run [t0, t1) therefore takes steps t0 - 9 ... t1 + 15: nine blocks of lead-in for the dependency cone of the
// lowest octave, fifteen steps for the pipeline to drain (tools/proto_cqt_pyramid.py checks every read of the
// schedule against the writes).  Positions outside a level's signal are written as zeros, so no load needs a mask
// except the consumer's validLength rule.
// Arithmetic is the per-octave path's, operation for operation: results are bit-identical to it.

namespace pyr {
constexpr int CONSUMERS = 7, PRODUCERS = 4, WAVES = CONSUMERS + PRODUCERS;
constexpr int LEAD = 9, DRAIN = 15;
__host__ __device__ constexpr int ring_size(int k) { return k >= 4 ? 1024 : 16384 >> k; }  // 8192, 4096, 2048, 1024 x 3
__host__ __device__ constexpr int ring_off(int k) { return k <= 4 ? 16384 - (32768 >> k) : 14336 + (k - 4) * 1024; }
static_assert(ring_off(1) == 0 && ring_off(2) == 8192 && ring_off(3) == 12288 && ring_off(4) == 14336 && ring_off(5) == 15360, "ring layout");
static_assert(ring_off(6) + ring_size(6) == AFX_CQT_PYR_RING_FLOATS, "ring layout");
__host__ __device__ constexpr int win_ahead(int k) { return k == 5 ? 2 : k == 6 ? 4 : 1; }
__host__ __device__ constexpr int lag(int k) { return k == 0 ? 3 : 2 * k + win_ahead(k); }
__host__ __device__ constexpr int plane_bytes(int k) {
    return k == 0 ? CqF16<128>::WAVE_BYTES : k == 1 ? CqF16<64>::WAVE_BYTES : k == 2 ? CqF16<32>::WAVE_BYTES
         : k == 3 ? CqF16<16>::WAVE_BYTES : k == 4 ? CqF16<8>::WAVE_BYTES : k == 5 ? CqF16<4>::WAVE_BYTES : CqF16<2>::WAVE_BYTES;
}
__host__ __device__ constexpr int plane_off(int k) { return k == 0 ? 0 : plane_off(k - 1) + plane_bytes(k - 1); }
constexpr int FIR_WORDS = 296;                       // even / odd input copies of a PAIR of rounds, float2 each
constexpr int FIR_BYTES = 2 * FIR_WORDS * 8;
constexpr int LDS_BYTES = CqF16<128>::B_BYTES + plane_off(7) + PRODUCERS * FIR_BYTES;
static_assert(LDS_BYTES <= 160 * 1024, "LDS");
constexpr int AUX_L2 = 17;                           // sc0 sc1: served by the L2, never by this CU's L1
// blocks of level k (1..6) a run [t0, t1) needs: [t0 - need_back(k), t1 + need_ahead(k)]
__host__ __device__ constexpr int need_back(int k) { return 10 - k; }
__host__ __device__ constexpr int need_ahead(int k) { return 9 - k; }
}  // namespace pyr

// phase stamps of the TIMING instantiation (tools/pyr_phases.py): shader cycles per role and phase, summed over the steps
template <bool TIMING>
struct PyrClock {
    unsigned long long t, acc[8];
    __device__ __forceinline__ PyrClock() {
        if (TIMING) {
            for (int i = 0; i < 8; ++i) acc[i] = 0;
            t = __builtin_amdgcn_s_memtime();
        }
    }
    __device__ __forceinline__ void lap(int slot) {
        if (TIMING) {
            __builtin_amdgcn_s_waitcnt(0xc07f);  // the stamp is an SMEM result
            const unsigned long long n = __builtin_amdgcn_s_memtime();
            acc[slot] += n - t;
            t = n;
        }
    }
    __device__ __forceinline__ void flush(unsigned long long *dst, int lane) {
        if (TIMING && dst && lane == 0)
            for (int i = 0; i < 8; ++i) dst[i] += acc[i];
    }
};

__device__ __forceinline__ void pyr_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// ---- consumer: octave level K of the run [t0c, t1c) of `clip`
template <int K, bool TIMING>
__device__ __forceinline__ void pyr_consumer(const AfxCqtPyramidArgs &a, unsigned char *smem, int lane, float *wgRing, int clip,
                                             int t0c, int t1c, unsigned long long *tim) {  // one run
    constexpr int H = 128 >> K;
    using C = CqF16<H>;
    constexpr int NVA = K == 0 ? C::NV / 2 : C::NV;  // level 0: half of the next window before the K loop, half behind it
    constexpr unsigned RMASK = (unsigned)pyr::ring_size(K) - 1u;
    unsigned char *Bl = smem;
    unsigned char *sig = smem + C::B_BYTES + pyr::plane_off(K);
    const int i = lane & 31, g = lane >> 5;
    const CqLane L = cq_lane_setup(lane, sig, 12, (6 - K) * 12, a.num, a.timeLength, a.colMul, a.scale, a.octScale[K]);
    const int cpy = i % C::COPIES;
    const unsigned char *aHi = sig + cpy * C::CS + C::at(i * H + 8 * g, cpy);
    const unsigned char *aLo = aHi + C::PART;
    const unsigned char *bHi = Bl + lane * 16;
    const unsigned char *bLo = bHi + C::KS * 64 * 16;
    const int valid = a.valid[K];
    // level 0: the clip, framed samples only (the bounds check supplies the zero padding on both sides); other
    // levels: the ring (zeros outside the signal are IN the ring; samples in [valid, len) are masked below)
    const __amdgpu_buffer_rsrc_t rsrc =
        K == 0 ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x + (long long)clip * a.xStride), 0, valid * 4, RSRC_RAW)
               : __builtin_amdgcn_make_buffer_rsrc(wgRing + pyr::ring_off(K), 0, pyr::ring_size(K) * 4, RSRC_RAW);
    float *outRe = a.outRe + (long long)clip * a.outStride, *outIm = a.outIm + (long long)clip * a.outStride;
    u32x4 wnd[C::NV];
    auto fetch = [&](int t, int u0, int u1) {
        const int p0 = t * 32 * H - (C::N >> 1);
#pragma unroll
        for (int u = 0; u < C::NV; ++u) {
            if (u < u0 || u >= u1) continue;
            const int pos = p0 + 4 * (lane + 64 * u);
            if (K == 0) wnd[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, pos * 4, 0, 0);
            else wnd[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(((unsigned)pos & RMASK) * 4u), 0, pyr::AUX_L2);
        }
    };
    const int s0 = t0c - pyr::LEAD, s1 = t1c + pyr::DRAIN;
    PyrClock<TIMING> clk;
    for (int s = s0; s <= s1; ++s) {
        const int t = s - pyr::lag(K);
        const bool act = t >= t0c && t < t1c, actNext = t + 1 >= t0c && t + 1 < t1c;  // wave-uniform
        float down = 0.f;
        if (act) {
            const int p0 = t * 32 * H - (C::N >> 1);
            if (K != 0 && p0 + C::S > valid) {  // stft_algorithm.c:838-843: samples past validLength are not framed
#pragma unroll
                for (int u = 0; u < C::NV; ++u) {
                    const int pos = p0 + 4 * (lane + 64 * u);
                    if (pos >= valid) wnd[u].x = 0u;
                    if (pos + 1 >= valid) wnd[u].y = 0u;
                    if (pos + 2 >= valid) wnd[u].z = 0u;
                    if (pos + 3 >= valid) wnd[u].w = 0u;
                }
            }
            const int e = cq_window_exponent<H>(wnd, lane);
            const float up = __uint_as_float((unsigned)(e + 127) << 23);
            down = __uint_as_float((unsigned)(127 - e) << 23);
            if (TIMING) {  // time the wait for the window apart from the conversion
#pragma unroll
                for (int u = 0; u < C::NV; ++u) PIN(wnd[u]);
                clk.lap(0);
            }
            wave_lds_order();
            cq_convert_window<H>(wnd, up, sig, lane);
            wave_lds_order();
            clk.lap(1);
        }
        if (actNext) fetch(t + 1, 0, NVA);
        clk.lap(2);
        f32x16 hh, hl, lh;
        if (act) cq_kloop<H>(aHi, aLo, bHi, bLo, hh, hl, lh);
        clk.lap(3);
        // the prefetched window is in its registers BEFORE the step's barrier (its ring blocks are overwritten three
        // steps on; nothing younger is in flight here, so this wait costs nothing) ...
#pragma unroll
        for (int u = 0; u < NVA; ++u) PIN(wnd[u]);
        clk.lap(4);
        if (act) cq_store_tile<true>(hh, hl, lh, down, L, outRe, outIm, t * 32);
        // ... level 0 reads the clip, nothing overwrites it: the second half follows the tile's stores
        if (K == 0 && actNext) fetch(t + 1, NVA, C::NV);
        clk.lap(5);
        pyr_barrier();
        clk.lap(6);
    }
    clk.flush(tim, lane);
}

// ---- producers: the 2:1 resampler in rounds of 256 outputs i0 ... i0 + 255 of level k+1 (the first nOut stored), four per
// lane, TWO ROUNDS AT A TIME (A, B: any two rounds, also of different stages -- the taps are the same): their inputs are
// staged in LDS interleaved, E2[m] = (XE_A[m], XE_B[m]) with XE[m] = x[2 (i0 - 15 + m)], O2 likewise with
// XO[m] = x[2 (i0 - 16 + m) + 1], so that a 16-byte read returns aligned register pairs and one v_pk_fma_f32 applies a tap
// to both rounds.  k_cqt_decimate's arithmetic per output: left taps j = 0..31 at x[2i - j], then right taps j = 1..31 at
// x[2i + j], one fma chain, divided by sqrt(ratio) -- the same bits.
//
// Producer PW, pair p of a step (stage, round of the block): 0: (0,0|0,1) (0,2|0,3); 1: (0,4|0,5) (0,6|0,7); 2: (1,0|1,1) (1,2|1,3);
// 3: (2,0|2,1) (3,0|4,0) (5,0|-).  Stage k writes block s - 2k of level k+1.
namespace pyr {
__host__ __device__ constexpr int pair_stage(int pw, int p, int half) {  // -1: no round in this half
    return pw < 2 ? 0 : pw == 2 ? 1 : p == 0 ? 2 : p == 1 ? 3 + half : half ? -1 : 5;
}
__host__ __device__ constexpr int pair_round(int pw, int p, int half) {
    return pw < 2 ? 4 * pw + 2 * p + half : pw == 2 ? 2 * p + half : p == 0 ? half : 0;
}
}  // namespace pyr

struct PyrRound {
    int i0;         // first output
    bool on, zero;  // due in this step; the block lies outside the signal (zeros, no input needed)
};

template <int K>
__device__ __forceinline__ PyrRound pyr_round_of(int s, int r, int t0c, int t1c, const AfxCqtPyramidArgs &a) {
    PyrRound R;
    if (K < 0) {
        R.i0 = 0;
        R.on = false;
        R.zero = true;
        return R;
    }
    constexpr int KK = K < 0 ? 0 : K, blockOut = 2048 >> KK;
    const int b = s - 2 * KK;
    R.i0 = b * blockOut + 256 * r;
    R.on = b >= t0c - pyr::need_back(KK + 1) && b <= t1c + pyr::need_ahead(KK + 1);  // the run needs that block
    R.zero = R.i0 >= a.len[KK + 1] || R.i0 + 256 <= 0;
    return R;
}

// the round's input on its way: 16 bytes at s_q = 2 i0 - 32 + 4 q hold XE[2q-1], XO[2q], XE[2q], XO[2q+1]; q = 0 ... 146.
// No control flow around the loads (a load in a branch ends in a copy of its result, i.e. a wait right behind it): a
// lane or a round with nothing to fetch loads from an out-of-range offset (zeros, no memory access).  Level 0 is the
// clip -- its bounds check supplies the zeros on both sides --, the others a ring with the zeros inside.
template <int K>
__device__ __forceinline__ void pyr_fir_issue(const PyrRound &R, const AfxCqtPyramidArgs &a, float *wgRing, int clip, int lane,
                                              u32x4 (&v)[3]) {
    constexpr int KK = K < 0 ? 0 : K;
    const bool live = K >= 0 && R.on && !R.zero;
    const __amdgpu_buffer_rsrc_t src =
        KK == 0 ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x + (long long)clip * a.xStride), 0, a.len[0] * 4, RSRC_RAW)
                : __builtin_amdgcn_make_buffer_rsrc(wgRing + pyr::ring_off(KK), 0, pyr::ring_size(KK) * 4, RSRC_RAW);
    constexpr unsigned mask = KK == 0 ? 0xffffffffu : (unsigned)pyr::ring_size(KK) - 1u;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int q = lane + 64 * u;
        unsigned off = ((unsigned)(2 * R.i0 - 32 + 4 * q) & mask) * 4u;
        if (!live || q >= 147) off = 0x80000000u;
        v[u] = __builtin_amdgcn_raw_buffer_load_b128(src, (int)off, 0, pyr::AUX_L2);
    }
}

// the four outputs of a lane -> its ring, zeros outside the signal
template <int K>
__device__ __forceinline__ void pyr_fir_store(const PyrRound &R, const float (&r)[4], const AfxCqtPyramidArgs &a, float *wgRing, int lane) {
    if (K < 0) return;
    constexpr int KK = K < 0 ? 0 : K, blockOut = 2048 >> KK, nOut = blockOut < 256 ? blockOut : 256;
    if (!R.on || 4 * lane >= nOut) return;
    const int i = R.i0 + 4 * lane, dstLen = a.len[KK + 1];
    u32x4 o;
    o.x = (i >= 0 && i < dstLen) ? __float_as_uint(r[0]) : 0u;
    o.y = (i + 1 >= 0 && i + 1 < dstLen) ? __float_as_uint(r[1]) : 0u;
    o.z = (i + 2 >= 0 && i + 2 < dstLen) ? __float_as_uint(r[2]) : 0u;
    o.w = (i + 3 >= 0 && i + 3 < dstLen) ? __float_as_uint(r[3]) : 0u;
    const __amdgpu_buffer_rsrc_t dst =
        __builtin_amdgcn_make_buffer_rsrc(wgRing + pyr::ring_off(KK + 1), 0, pyr::ring_size(KK + 1) * 4, RSRC_RAW);
    constexpr unsigned dstMask = (unsigned)pyr::ring_size(KK + 1) - 1u;
    __builtin_amdgcn_raw_buffer_store_b128(o, dst, (int)(((unsigned)i & dstMask) * 4u), 0, 0);
}

typedef float f4 __attribute__((ext_vector_type(4)));

template <int KA, int KB, bool TIMING>
__device__ __forceinline__ void pyr_fir_pair(const PyrRound &RA, const PyrRound &RB, u32x4 (&va)[3], u32x4 (&vb)[3], const v2 (&hp)[16],
                                             float sqrtRatio, const AfxCqtPyramidArgs &a, float *wgRing, v2 *E2, v2 *O2, int lane,
                                             PyrClock<TIMING> &clk) {
    const bool liveA = KA >= 0 && RA.on && !RA.zero, liveB = KB >= 0 && RB.on && !RB.zero;
    float rA[4] = {0.f, 0.f, 0.f, 0.f}, rB[4] = {0.f, 0.f, 0.f, 0.f};
    if (liveA || liveB) {
        if (TIMING) {
            clk.lap(0);
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                PIN(va[u]);
                PIN(vb[u]);
            }
            clk.lap(2);  // waiting for the rounds' input
        }
        __builtin_amdgcn_wave_barrier();  // (the previous pair's reads were waited for before its taps)
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int q = lane + 64 * u;
            if (q < 147) {
                const float4 fa = __builtin_bit_cast(float4, va[u]), fb = __builtin_bit_cast(float4, vb[u]);
                const int m = 2 * q - 1;
                if (m >= 0) E2[m] = v2{fa.x, fb.x};
                O2[m + 1] = v2{fa.y, fb.y};
                E2[m + 1] = v2{fa.z, fb.z};
                O2[m + 2] = v2{fa.w, fb.w};
            }
        }
        wave_lds_order();
        // entries e (two per 16-byte read) of the lane's window: E2[4 lane + e], O2[4 lane + e]; the left taps need
        // entries 0 ... 18 of both, the right taps 16 ... 34: two register sets of 20, one after the other
        const v2 *pe = E2 + 4 * lane, *po = O2 + 4 * lane;
        v2 acc[4] = {v2{0.f, 0.f}, v2{0.f, 0.f}, v2{0.f, 0.f}, v2{0.f, 0.f}};
        {
            f4 ev[10], ov[10];
            RD128_P(ev[0], pe, 0);    RD128_P(ov[0], po, 0);    RD128_P(ev[1], pe, 16);   RD128_P(ov[1], po, 16);
            RD128_P(ev[2], pe, 32);   RD128_P(ov[2], po, 32);   RD128_P(ev[3], pe, 48);   RD128_P(ov[3], po, 48);
            RD128_P(ev[4], pe, 64);   RD128_P(ov[4], po, 64);   RD128_P(ev[5], pe, 80);   RD128_P(ov[5], po, 80);
            RD128_P(ev[6], pe, 96);   RD128_P(ov[6], po, 96);   RD128_P(ev[7], pe, 112);  RD128_P(ov[7], po, 112);
            RD128_P(ev[8], pe, 128);  RD128_P(ov[8], po, 128);  RD128_P(ev[9], pe, 144);  RD128_P(ov[9], po, 144);
            LDS_WAIT_N(0);
            clk.lap(3);  // staging through LDS
#pragma unroll
            for (int b = 0; b < 10; ++b) {
                PIN(ev[b]);
                PIN(ov[b]);
            }
            auto E = [&](int e) { return (e & 1) ? v2{ev[e >> 1].z, ev[e >> 1].w} : v2{ev[e >> 1].x, ev[e >> 1].y}; };
            auto O = [&](int e) { return (e & 1) ? v2{ov[e >> 1].z, ov[e >> 1].w} : v2{ov[e >> 1].x, ov[e >> 1].y}; };
#pragma unroll
            for (int j = 0; j < 32; ++j) {  // x[2i - j]: even j -> x[2 (i - j/2)], odd j -> x[2 (i - (j+1)/2) + 1]
                if (j & 1) {
                    const int e = 16 - (j + 1) / 2;
                    pk_tap4_hi(acc[0], acc[1], acc[2], acc[3], hp[j >> 1], O(e), O(e + 1), O(e + 2), O(e + 3));
                } else {
                    const int e = 15 - j / 2;
                    pk_tap4_lo(acc[0], acc[1], acc[2], acc[3], hp[j >> 1], E(e), E(e + 1), E(e + 2), E(e + 3));
                }
            }
        }
        {
            f4 ev[10], ov[10];  // entries 16 ... 35
            RD128_P(ev[0], pe, 128);  RD128_P(ov[0], po, 128);  RD128_P(ev[1], pe, 144);  RD128_P(ov[1], po, 144);
            RD128_P(ev[2], pe, 160);  RD128_P(ov[2], po, 160);  RD128_P(ev[3], pe, 176);  RD128_P(ov[3], po, 176);
            RD128_P(ev[4], pe, 192);  RD128_P(ov[4], po, 192);  RD128_P(ev[5], pe, 208);  RD128_P(ov[5], po, 208);
            RD128_P(ev[6], pe, 224);  RD128_P(ov[6], po, 224);  RD128_P(ev[7], pe, 240);  RD128_P(ov[7], po, 240);
            RD128_P(ev[8], pe, 256);  RD128_P(ov[8], po, 256);  RD128_P(ev[9], pe, 272);  RD128_P(ov[9], po, 272);
            LDS_WAIT_N(0);
#pragma unroll
            for (int b = 0; b < 10; ++b) {
                PIN(ev[b]);
                PIN(ov[b]);
            }
            auto E = [&](int e) { return ((e - 16) & 1) ? v2{ev[(e - 16) >> 1].z, ev[(e - 16) >> 1].w} : v2{ev[(e - 16) >> 1].x, ev[(e - 16) >> 1].y}; };
            auto O = [&](int e) { return ((e - 16) & 1) ? v2{ov[(e - 16) >> 1].z, ov[(e - 16) >> 1].w} : v2{ov[(e - 16) >> 1].x, ov[(e - 16) >> 1].y}; };
#pragma unroll
            for (int j = 1; j < 32; ++j) {  // x[2i + j]: even j -> x[2 (i + j/2)], odd j -> x[2 (i + (j-1)/2) + 1]
                if (j & 1) {
                    const int e = 16 + (j - 1) / 2;
                    pk_tap4_hi(acc[0], acc[1], acc[2], acc[3], hp[j >> 1], O(e), O(e + 1), O(e + 2), O(e + 3));
                } else {
                    const int e = 15 + j / 2;
                    pk_tap4_lo(acc[0], acc[1], acc[2], acc[3], hp[j >> 1], E(e), E(e + 1), E(e + 2), E(e + 3));
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            rA[q] = acc[q].x / sqrtRatio;
            rB[q] = acc[q].y / sqrtRatio;
        }
        if (TIMING) {
            PIN(rA[0]); PIN(rA[3]); PIN(rB[0]); PIN(rB[3]);
            clk.lap(4);  // the taps
        }
    }
    pyr_fir_store<KA>(RA, rA, a, wgRing, lane);
    pyr_fir_store<KB>(RB, rB, a, wgRing, lane);
}

template <int PW, bool TIMING>
__device__ __forceinline__ void pyr_producer(const AfxCqtPyramidArgs &a, unsigned char *smem, int lane, float *wgRing, int clip,
                                             int t0c, int t1c, unsigned long long *tim) {
    v2 *E2 = reinterpret_cast<v2 *>(smem + CqF16<128>::B_BYTES + pyr::plane_off(7) + PW * pyr::FIR_BYTES);
    v2 *O2 = E2 + pyr::FIR_WORDS;
    const int s0 = t0c - pyr::LEAD, s1 = t1c + pyr::DRAIN;
    v2 hp[16];  // the taps in VECTOR registers, pairs (h[2n], h[2n+1]): as scalars next to the descriptors they spill
#pragma unroll
    for (int n = 0; n < 16; ++n) {
        hp[n] = v2{a.taps[2 * n], a.taps[2 * n + 1]};
        PIN(hp[n]);
    }
    const float sqrtRatio = a.sqrtRatio;
    PyrClock<TIMING> clk;
    constexpr int K0A = pyr::pair_stage(PW, 0, 0), K0B = pyr::pair_stage(PW, 0, 1), K1A = pyr::pair_stage(PW, 1, 0), K1B = pyr::pair_stage(PW, 1, 1);
    constexpr int K2A = PW == 3 ? pyr::pair_stage(PW, 2, 0) : -1, K2B = -1;
    constexpr int R0A = pyr::pair_round(PW, 0, 0), R0B = pyr::pair_round(PW, 0, 1), R1A = pyr::pair_round(PW, 1, 0), R1B = pyr::pair_round(PW, 1, 1);
    // the input of a pair is requested one pair ahead, and across the step's barrier when it comes from the clip
    // (producers 0 and 1; a ring block is only complete behind the barrier): two register sets, used in turn
    constexpr bool AHEAD = PW < 2;
    u32x4 a0[3], b0[3], a1[3], b1[3];
    if (AHEAD) {
        pyr_fir_issue<K0A>(pyr_round_of<K0A>(s0, R0A, t0c, t1c, a), a, wgRing, clip, lane, a0);
        pyr_fir_issue<K0B>(pyr_round_of<K0B>(s0, R0B, t0c, t1c, a), a, wgRing, clip, lane, b0);
    }
    for (int s = s0; s <= s1; ++s) {
        const PyrRound P0A = pyr_round_of<K0A>(s, R0A, t0c, t1c, a), P0B = pyr_round_of<K0B>(s, R0B, t0c, t1c, a);
        const PyrRound P1A = pyr_round_of<K1A>(s, R1A, t0c, t1c, a), P1B = pyr_round_of<K1B>(s, R1B, t0c, t1c, a);
        if (!AHEAD) {
            pyr_fir_issue<K0A>(P0A, a, wgRing, clip, lane, a0);
            pyr_fir_issue<K0B>(P0B, a, wgRing, clip, lane, b0);
        }
        pyr_fir_issue<K1A>(P1A, a, wgRing, clip, lane, a1);
        pyr_fir_issue<K1B>(P1B, a, wgRing, clip, lane, b1);
        pyr_fir_pair<K0A, K0B, TIMING>(P0A, P0B, a0, b0, hp, sqrtRatio, a, wgRing, E2, O2, lane, clk);
        if (AHEAD) {  // (harmless behind the run's last step)
            pyr_fir_issue<K0A>(pyr_round_of<K0A>(s + 1, R0A, t0c, t1c, a), a, wgRing, clip, lane, a0);
            pyr_fir_issue<K0B>(pyr_round_of<K0B>(s + 1, R0B, t0c, t1c, a), a, wgRing, clip, lane, b0);
        }
        const PyrRound P2A = pyr_round_of<K2A>(s, 0, t0c, t1c, a), P2B = pyr_round_of<K2B>(s, 0, t0c, t1c, a);
        if (PW == 3) {
            pyr_fir_issue<K2A>(P2A, a, wgRing, clip, lane, a0);
            pyr_fir_issue<K2B>(P2B, a, wgRing, clip, lane, b0);
        }
        pyr_fir_pair<K1A, K1B, TIMING>(P1A, P1B, a1, b1, hp, sqrtRatio, a, wgRing, E2, O2, lane, clk);
        if (PW == 3) pyr_fir_pair<K2A, K2B, TIMING>(P2A, P2B, a0, b0, hp, sqrtRatio, a, wgRing, E2, O2, lane, clk);
        clk.lap(5);
        VM_WAIT_ALL();  // the blocks are in the L2 before the other waves pass the barrier
        clk.lap(1);
        pyr_barrier();
        clk.lap(6);
    }
    clk.flush(tim, lane);
}

// the runs of this workgroup, one after the other; ROLE 0-6: consumer of that level, 7-10: producer ROLE - 7
template <int ROLE, bool TIMING>
__device__ __forceinline__ void pyr_role(const AfxCqtPyramidArgs &a, unsigned char *smem, int lane, float *wgRing, int items) {
    unsigned long long *tim = TIMING && a.timing ? a.timing + ((size_t)blockIdx.x * pyr::WAVES + (threadIdx.x >> 6)) * 8 : nullptr;
    const int nT = (a.timeLength + 31) / 32;
    for (int it = blockIdx.x; it < items; it += gridDim.x) {
        const int clip = it / a.chunksPerClip, chunk = it - clip * a.chunksPerClip;
        const int t0c = chunk * a.tilesPerChunk;
        const int t1c = t0c + a.tilesPerChunk < nT ? t0c + a.tilesPerChunk : nT;
        if (t0c >= t1c) continue;  // (uniform over the workgroup)
        if (ROLE < pyr::CONSUMERS) pyr_consumer<(ROLE < pyr::CONSUMERS ? ROLE : 0), TIMING>(a, smem, lane, wgRing, clip, t0c, t1c, tim);
        else pyr_producer<(ROLE >= pyr::CONSUMERS ? ROLE - pyr::CONSUMERS : 0), TIMING>(a, smem, lane, wgRing, clip, t0c, t1c, tim);
    }
}

template <bool TIMING>
__global__ __launch_bounds__(64 * pyr::WAVES) void k_cqt_pyramid(AfxCqtPyramidArgs a, int items) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    cq_image_to_lds(a.timeKernelH, smem_raw, tid, 64 * pyr::WAVES);
    __syncthreads();
    float *wgRing = a.ring + (size_t)blockIdx.x * AFX_CQT_PYR_RING_FLOATS;
    // (the loop over the workgroup's runs sits INSIDE every role: around the switch, the compiler hoists the per-lane
    // constants of all eight roles in front of it and spills them)
    switch (wave) {
        case 0: pyr_role<0, TIMING>(a, smem_raw, lane, wgRing, items); break;
        case 1: pyr_role<1, TIMING>(a, smem_raw, lane, wgRing, items); break;
        case 2: pyr_role<2, TIMING>(a, smem_raw, lane, wgRing, items); break;
        case 3: pyr_role<3, TIMING>(a, smem_raw, lane, wgRing, items); break;
        case 4: pyr_role<4, TIMING>(a, smem_raw, lane, wgRing, items); break;
        case 5: pyr_role<5, TIMING>(a, smem_raw, lane, wgRing, items); break;
        case 6: pyr_role<6, TIMING>(a, smem_raw, lane, wgRing, items); break;
        case 7: pyr_role<7, TIMING>(a, smem_raw, lane, wgRing, items); break;
        case 8: pyr_role<8, TIMING>(a, smem_raw, lane, wgRing, items); break;
        case 9: pyr_role<9, TIMING>(a, smem_raw, lane, wgRing, items); break;
        default: pyr_role<10, TIMING>(a, smem_raw, lane, wgRing, items); break;
    }
}

template <int H, bool R12>
int launch_f16(const AfxCqtOctaveArgs *a, void *stream) {
    using C = CqF16<H>;
    int waves = (160 * 1024 - C::B_BYTES) / C::WAVE_BYTES;
    if (waves > 8) waves = 8;
    if (waves >= 4) waves &= ~3;  // the same number of waves on every SIMD
    if (waves < 1) return AFX_ERR_UNSUPPORTED;
    const size_t lds = (size_t)C::B_BYTES + (size_t)waves * C::WAVE_BYTES;
    const void *fn = reinterpret_cast<const void *>(k_cqt_octave_f16<H, R12>);
    AFX_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int tilesPerClip = (a->timeLength + 31) / 32;
    const long long total = (long long)tilesPerClip * (a->batch > 0 ? a->batch : 1);
    if (total > 0x7fffffffLL) return AFX_ERR_UNSUPPORTED;
    long long wgs = (total + waves - 1) / waves;
    if (wgs > 256) wgs = 256;  // one persistent workgroup per CU
    AfxCqtOctaveArgs b = *a;
    if (b.batch <= 0) b.batch = 1;
    hipLaunchKernelGGL((k_cqt_octave_f16<H, R12>), dim3((unsigned)wgs), dim3(64 * waves), lds, (hipStream_t)stream, b,
                       tilesPerClip);
    AFX_LAUNCH_CHECK("k_cqt_octave_f16");
    return AFX_OK;
}

template <int H>
int dispatch_f16(const AfxCqtOctaveArgs *a, void *stream) {
    // 12 bins per octave: the tile is transposed through LDS and leaves as 12-byte-per-lane stores
    return a->rows == 12 ? launch_f16<H, true>(a, stream) : launch_f16<H, false>(a, stream);
}

}  // namespace

// N = 512, one column tile, power-of-two hop <= 128; anything else returns
// AFX_ERR_UNSUPPORTED and the caller (afxk_cqt_octave) takes the float32 kernels.
extern "C" int afxk_cqt_octave_f16(const AfxCqtOctaveArgs *a, void *stream) {
    if (!a->timeKernelH || !a->colMul || a->colTiles != 1 || a->radix2Exp != 9) return AFX_ERR_UNSUPPORTED;
    // (no alignment condition on the clip rows: buffer loads need dword alignment only)
    // 32-bit byte offsets inside one clip's signal and one clip's output plane
    if (a->validLength > (1 << 28) || (long long)a->timeLength * a->num > (1LL << 28)) return AFX_ERR_UNSUPPORTED;
    switch (a->hop) {
        case 128: return dispatch_f16<128>(a, stream);
        case 64: return dispatch_f16<64>(a, stream);
        case 32: return dispatch_f16<32>(a, stream);
        case 16: return dispatch_f16<16>(a, stream);
        case 8: return dispatch_f16<8>(a, stream);
        case 4: return dispatch_f16<4>(a, stream);
        case 2: return dispatch_f16<2>(a, stream);
        default: return AFX_ERR_UNSUPPORTED;
    }
}

// ---- the pyramid launch ----
extern "C" int afxk_cqt_pyramid_plan(int batch, int timeLength, int *chunksPerClip, int *tilesPerChunk) {
    if (batch <= 0 || timeLength <= 0) return 0;
    const int nT = (timeLength + 31) / 32;
    // runs of one clip: as many as fill the CUs, but long enough that the 24 steps of lead-in and drain stay small
    int cpc = AFX_CQT_PYR_MAX_WGS / batch;
    if (cpc > nT / 48) cpc = nT / 48;
    if (cpc < 1) cpc = 1;
    int tpc = (nT + cpc - 1) / cpc;
    if (const char *e = getenv("AFX_CQT_PYR_TILES"))  // tests: short runs, so that small inputs cross run boundaries
        if (atoi(e) > 0 && atoi(e) < tpc) tpc = atoi(e);
    cpc = (nT + tpc - 1) / tpc;  // no empty runs
    if (chunksPerClip) *chunksPerClip = cpc;
    if (tilesPerChunk) *tilesPerChunk = tpc;
    const long long items = (long long)batch * cpc;
    return (int)(items < AFX_CQT_PYR_MAX_WGS ? items : AFX_CQT_PYR_MAX_WGS);
}

extern "C" int afxk_cqt_pyramid(const AfxCqtPyramidArgs *a, void *stream) {
    if (!a->x || !a->timeKernelH || !a->colMul || !a->scale || !a->outRe || !a->outIm || !a->ring) return AFX_ERR_ARG;
    if (a->batch <= 0 || a->timeLength <= 0 || a->chunksPerClip <= 0 || a->tilesPerChunk <= 0) return AFX_ERR_ARG;
    // 32-bit byte offsets inside one clip's signal and one clip's output plane
    if (a->len[0] > (1 << 28) || (long long)a->timeLength * a->num > (1LL << 28)) return AFX_ERR_UNSUPPORTED;
    const long long items = (long long)a->batch * a->chunksPerClip;
    if (items > 0x7fffffffLL) return AFX_ERR_UNSUPPORTED;
    const unsigned grid = (unsigned)(items < AFX_CQT_PYR_MAX_WGS ? items : AFX_CQT_PYR_MAX_WGS);
    if (a->timing) {  // the instrumented instantiation (tools/pyr_phases.py)
        AFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_cqt_pyramid<true>), hipFuncAttributeMaxDynamicSharedMemorySize, pyr::LDS_BYTES));
        hipLaunchKernelGGL(k_cqt_pyramid<true>, dim3(grid), dim3(64 * pyr::WAVES), pyr::LDS_BYTES, (hipStream_t)stream, *a, (int)items);
        AFX_LAUNCH_CHECK("k_cqt_pyramid<timing>");
        return AFX_OK;
    }
    AFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_cqt_pyramid<false>), hipFuncAttributeMaxDynamicSharedMemorySize, pyr::LDS_BYTES));
    hipLaunchKernelGGL(k_cqt_pyramid<false>, dim3(grid), dim3(64 * pyr::WAVES), pyr::LDS_BYTES, (hipStream_t)stream, *a, (int)items);
    AFX_LAUNCH_CHECK("k_cqt_pyramid");
    return AFX_OK;
}
