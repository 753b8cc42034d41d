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

// Knock-out measurement builds (make EXTRA=-DAFX_KO_CQT=<mask>; results are WRONG, timing only; tools/gpu_ko_cqt.sh,
// profiles/r06_ko_cqt.txt): bit 0 row stores, 1 chroma loads + stores, 2 ring stores, 3 A-operand LDS reads, 4 B-operand (image / tap)
// LDS reads, 5 window conversion, 6 resampler products, 7 the two correction products of the K loop, 8 clip loads, 9 ring loads,
// 11 row pieces as aligned 64-byte writes.  "Price" bits ADD work on the real data (knocking a class out changes the operands, and
// the clock of this kernel depends on them): 12 every ring store twice (second copy into the chroma plane), 13 the K loop's first
// product twice (+ 224 MFMAs per step), 14 the K loop's A-operand reads twice
#ifdef AFX_KO_CQT
#define CQ_KO(s) (((AFX_KO_CQT) >> (s)) & 1)
#else
#define CQ_KO(s) 0
#endif

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

// (knock-out builds: an operand the compiler cannot fold)
__device__ __forceinline__ h8 cq_ko_operand(int salt) {
    u32x4 v;
    asm volatile("v_mov_b32 %0, %1" : "=v"(v.x) : "s"(salt));
    v.y = v.z = v.w = v.x;
    return __builtin_bit_cast(h8, v);
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
        if (CQ_KO(3)) {
            ah[slot] = al[slot] = cq_ko_operand(0x3c003c00);
        } else {
            ah[slot] = *reinterpret_cast<const h8 *>(aHi + C::step(ks));
            al[slot] = *reinterpret_cast<const h8 *>(aLo + C::step(ks));
        }
        if (CQ_KO(4)) {
            bh[slot] = bl[slot] = cq_ko_operand(0x38003800);
        } else {
            bh[slot] = *reinterpret_cast<const h8 *>(bHi + 1024 * ks);
            bl[slot] = *reinterpret_cast<const h8 *>(bLo + 1024 * ks);
        }
    };
    load(0, 0);
    load(1, 1);
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
        __builtin_amdgcn_sched_barrier(0);
        if (ks + 2 < C::KS) load(ks + 2, (ks + 2) % 3);
        const int sl = ks % 3;
        hh = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[sl], bh[sl], hh, 0, 0, 0);
        if (CQ_KO(13)) lh = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[sl], bh[sl], lh, 0, 0, 0);
        if (CQ_KO(14) && ks + 2 < C::KS) {
            h8 d0 = *reinterpret_cast<const volatile h8 *>(aHi + C::step(ks + 2)), d1 = *reinterpret_cast<const volatile h8 *>(aLo + C::step(ks + 2));
            asm volatile("" ::"v"(d0), "v"(d1));
        }
        if (!CQ_KO(7)) {
            hl = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[sl], bl[sl], hl, 0, 0, 0);
            lh = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[sl], bh[sl], lh, 0, 0, 0);
        }
        if (ks + 2 < C::KS && !CQ_KO(3) && !CQ_KO(4) && !CQ_KO(7) && !CQ_KO(13) && !CQ_KO(14)) {
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
template <bool R12, int AUX = 0>
__device__ __forceinline__ void cq_store_tile(const f32x16 &hh, const f32x16 &hl, const f32x16 &lh, float down, const CqLane &L,
                                              float *outRe, float *outIm, int t0, u32x3 *pieces = nullptr) {
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
            if (pieces) pieces[q] = v3;  // (the pyramid's chroma: bins 3 (lane & 3) .. + 2 of frame 16 (q >> 1) + (lane >> 2), plane q & 1)
#if defined(AFX_KO_CQT) && (AFX_KO_CQT & 2048)  // bit 11 (timing only): every row piece as an ALIGNED 64-byte write (4 lanes x 16 bytes) instead of 48 bytes
            __builtin_amdgcn_raw_buffer_store_b128(v, (q & 1) ? rIm : rRe, ((L.voff12 + tileOff + (unsigned)(q >> 1) * 16u * L.rowBytes) & ~63u) + 16u * (threadIdx.x & 3), 0, AUX);
#else
            __builtin_amdgcn_raw_buffer_store_b96(v3, (q & 1) ? rIm : rRe, CQ_KO(0) ? 0x80000000u + 16u * q : L.voff12 + tileOff + (unsigned)(q >> 1) * 16u * L.rowBytes, 0, AUX);
#endif
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
// A workgroup of seven waves walks a run of 32-frame tiles of one clip, one STEP per tile; wave k owns octave level k
// (hop H = 128 >> k).  Per step: the level's window -> (xh, xl) planes in the wave's own LDS region -> 96 MFMAs ->
// the tile's 12 bins of 32 rows to memory (the tile of k_cqt_octave_f16 above) -- and then, FROM THE SAME PLANES, the
// next level's samples: row t' of the tile holds the 63 inputs of the H/2 resampler outputs (32 t + t') H/2 + c',
// y = sum_m h[|m|] x[2i + m], at window positions n = 2c' + 256 + m, so the 2:1 resampler is one more product of the
// Toeplitz operand with a banded matrix W[n][c'] = h[|n - 256 - 2c'|]: 5-8 K steps of three MFMAs per 32 columns
// (+ 15 ... 48 MFMAs on the tile's 96), no second pass over the signal, no vector-unit filter.  W is never stored: a
// lane's eight taps of a K step are eight consecutive entries of the tap table T[d] = h[|d|] 2^15 (f16 (hi, lo) words,
// zeros outside |d| <= 31) at d = 16 ks + 8 g - 256 - 2c'; four copies of the table shifted by 2 (c' mod 4) make that
// a 16-byte aligned read at an immediate offset, the copies 704 bytes apart put the 16 lanes of a read group on 16
// distinct bank quads.  [A vector-unit resampler in four more waves was measured first: 13-17 k cycles per step
// against 5-6 k for the octave waves -- one wave issues a vector instruction every ~5.5 cycles, and packed f32
// arithmetic beside MFMA waves is slower than scalar (profiles/r04_cqt_pyramid.txt).]
// The arithmetic of the resampler is therefore the octave product's: operands as (hi, lo) binary16 words of the
// tile-scaled samples (>= 22 bits of every sample) and of the taps, three products, float32 accumulation -- as close
// to the exact filter as the reference's float32 chain, not the same bits (tests: both against the reference).
//
// The level signals live in per-workgroup RINGS in global memory (8192 ... 1024 samples per level, 68 KB per
// workgroup): written once, read a few steps later by the same CU, overwritten four blocks on -- they stay in the
// L2 and never reach HBM, a clip is read from HBM once.  Ring loads bypass the CU's L1 (sc0 sc1), ring stores are
// drained (vmcnt(0)) before the step's s_barrier: ONE barrier per step is the only synchronisation.
//
// Schedule (step s; "block b of level k" = samples [32 b hop_k, 32 (b+1) hop_k) = what tile b of level k-1 produces):
// wave k works on tile s - lag_k, lag = 0, 3, 6, 9, 12, 16, 22: its window reaches e_k = 1,1,1,1,2,4 blocks past its
// tile, those were written by wave k-1 in step (t + e_k) + lag_(k-1), are visible one step later, when the window is
// requested, and used one step after that.  A ring of 4 blocks (8 / 16 for the two smallest levels) holds what is
// alive at a step.  A run [t0, t1) of tiles takes steps t0 - 9 ... t1 + 21: wave k also runs the resampler alone on
// the tiles [t0 - (9 - k), t0) and [t1, t1 + (8 - k)] whose blocks the lower octaves of the run reach into
// (tools/proto_cqt_pyramid.py checks every read of the schedule against the writes).  Positions outside a level's
// signal are written as zeros; the framing rule (samples in [validLength, length) are not framed but ARE resampled,
// stft_algorithm.c:838-843) costs the few tiles at a clip's end a second conversion.

namespace pyr {
constexpr int WAVES = 8;
constexpr int LEAD = 9, DRAIN = 22;
__host__ __device__ constexpr int ring_size(int k) { return k >= 4 ? 1024 : 16384 >> k; }  // 8192, 4096, 2048, 1024 x 3
__host__ __device__ constexpr int ring_off(int k) { return k <= 4 ? 16384 - (32768 >> k) : 14336 + (k - 4) * 1024; }
static_assert(ring_off(1) == 0 && ring_off(2) == 8192 && ring_off(3) == 12288 && ring_off(4) == 14336 && ring_off(5) == 15360, "ring layout");
static_assert(ring_off(6) + ring_size(6) == AFX_CQT_PYR_RING_FLOATS, "ring layout");
__host__ __device__ constexpr int lag(int k) { return k <= 4 ? 3 * k + 1 : k == 5 ? 17 : 23; }  // (level 0 is PREPARED a step before)
__host__ __device__ constexpr int plane_bytes(int k) {
    return k == 0 ? CqF16<128>::WAVE_BYTES : k == 1 ? CqF16<64>::WAVE_BYTES : k == 2 ? CqF16<32>::WAVE_BYTES
         : k == 3 ? CqF16<16>::WAVE_BYTES : k == 4 ? CqF16<8>::WAVE_BYTES : k == 5 ? CqF16<4>::WAVE_BYTES : CqF16<2>::WAVE_BYTES;
}
__host__ __device__ constexpr int plane_off(int k) { return k == 0 ? 0 : plane_off(k - 1) + plane_bytes(k - 1); }
// the tap table: 2 word planes x 4 shifted copies
constexpr int TAB_COPY = AFX_CQT_PYR_TAB_COPY, TAB_PLANE = 4 * TAB_COPY, TAB_BYTES = 2 * TAB_PLANE;
static_assert(TAB_BYTES == AFX_CQT_PYR_TAB_HALFS * 2, "tap table");
// level 0's second plane buffer (tiles alternate), and the 2^-e of the two
constexpr int ALT_BYTES = CqF16<128>::WAVE_BYTES;
constexpr int LDS_BYTES = CqF16<128>::B_BYTES + plane_off(7) + TAB_BYTES + ALT_BYTES + 16;
static_assert(LDS_BYTES <= 160 * 1024, "LDS");
constexpr int AUX_L2 = 17;                           // sc0 sc1: served by the L2, never by this CU's L1
constexpr int AUX_STREAM = 2;                        // nt: the clip and the rows pass through once -- they must not push the rings out of the L2
// (measurement builds: cache policy of the row stores / the ring stores / the chroma stores)
#ifndef AFX_CQ_ROW_AUX
#define AFX_CQ_ROW_AUX 2
#endif
#ifndef AFX_CQ_RING_AUX
#define AFX_CQ_RING_AUX 0
#endif
#ifndef AFX_CQ_CHROMA_AUX
#define AFX_CQ_CHROMA_AUX 0
#endif
// tiles of level k whose resampler output (block of level k+1) a run [t0, t1) needs: [t0 - need_back(k), t1 + need_ahead(k)]
__host__ __device__ constexpr int need_back(int k) { return 9 - k; }
__host__ __device__ constexpr int need_ahead(int k) { return 8 - k; }
// K steps of the resampler product for column tile CT of hop H: positions 2c' + 225 ... 2c' + 287 of the valid columns
__host__ __device__ constexpr int dec_cols(int H) { return H / 2 < 32 ? H / 2 : 32; }
__host__ __device__ constexpr int dec_ks0(int ct) { return (225 + 64 * ct) / 16; }
__host__ __device__ constexpr int dec_ks1(int H, int ct) { return (2 * (dec_cols(H) - 1 + 32 * ct) + 287) / 16; }
}  // namespace pyr

// phase stamps of the TIMING instantiation (tools/pyr_phases.py): shader cycles per wave and phase, summed over the steps
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

// the resampler product of the tile in the planes: outputs c' = 32 CT + (row) of every frame (column); operands two steps ahead
template <int H, int CT>
__device__ __forceinline__ void pyr_dec_loop(const unsigned char *aHi, const unsigned char *aLo, const unsigned char *tHi, const unsigned char *tLo,
                                             f32x16 &hh, f32x16 &hl, f32x16 &lh) {
    using C = CqF16<H>;
    constexpr int KS0 = pyr::dec_ks0(CT), KS1 = pyr::dec_ks1(H, CT);
#pragma unroll
    for (int r = 0; r < 16; ++r) hh[r] = hl[r] = lh[r] = 0.f;
    h8 ah[3], al[3], th[3], tl[3];
    auto load = [&](int ks, int slot) {
        if (CQ_KO(3)) {
            ah[slot] = al[slot] = cq_ko_operand(0x3c003c00);
        } else {
            ah[slot] = *reinterpret_cast<const h8 *>(aHi + C::step(ks));
            al[slot] = *reinterpret_cast<const h8 *>(aLo + C::step(ks));
        }
        if (CQ_KO(4)) {
            th[slot] = tl[slot] = cq_ko_operand(0x38003800);
        } else {
            th[slot] = *reinterpret_cast<const h8 *>(tHi + 32 * ks);
            tl[slot] = *reinterpret_cast<const h8 *>(tLo + 32 * ks);
        }
    };
    load(KS0, 0);
    load(KS0 + 1, 1);
#pragma unroll
    for (int ks = KS0; ks <= KS1; ++ks) {
        __builtin_amdgcn_sched_barrier(0);
        if (ks + 2 <= KS1) load(ks + 2, (ks + 2 - KS0) % 3);
        const int sl = (ks - KS0) % 3;
        // the taps as the A operand: result ROW = output c', COLUMN = frame -- a lane then holds runs of four consecutive
        // outputs of its frame (16-byte stores)
        hh = __builtin_amdgcn_mfma_f32_32x32x16_f16(th[sl], ah[sl], hh, 0, 0, 0);
        hl = __builtin_amdgcn_mfma_f32_32x32x16_f16(tl[sl], ah[sl], hl, 0, 0, 0);
        lh = __builtin_amdgcn_mfma_f32_32x32x16_f16(th[sl], al[sl], lh, 0, 0, 0);
        if (ks + 2 <= KS1 && !CQ_KO(3) && !CQ_KO(4)) {
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

// The framing rule (stft_algorithm.c:838-843: samples in [validLength, length) are not framed) on planes that hold the
// whole signal -- the resampler reads those samples --: window samples first ... last - 1 become zeros in both word planes
// and every copy, two bytes per lane and trip (a clip's last tiles only; at most hop - 1 samples)
template <int H>
__device__ __forceinline__ void cq_zero_samples(unsigned char *sig, int first, int last, int lane) {
    using C = CqF16<H>;
    for (int s = first + lane; s < last; s += 64) {
#pragma unroll
        for (int c = 0; c < C::COPIES; ++c) {
            unsigned char *d = sig + c * C::CS + C::at(s, c);
            *reinterpret_cast<unsigned short *>(d) = 0;
            *reinterpret_cast<unsigned short *>(d + C::PART) = 0;
        }
    }
}

// ... and its results -> the ring of the next level.  Lane (frame t' = lane & 31, g = lane >> 5) holds outputs
// c' = 32 CT + 8 q + 4 g + j (register 4 q + j) of sample row (32 t + t') H/2: four 16-byte stores.  A block is a
// ring slot (ring sizes are multiples of the block), so the address is a lane constant + immediates; zeros outside the
// signal (only the first and the last blocks of a clip take the masked path).
template <int H, int CT>
__device__ __forceinline__ void pyr_dec_store(const f32x16 &hh, const f32x16 &hl, const f32x16 &lh, float mul, int t, int lane,
                                              const __amdgpu_buffer_rsrc_t &ring, unsigned ringMask, int dstLen,
                                              const __amdgpu_buffer_rsrc_t &ringDup) {
    constexpr int W = H / 2;                      // outputs per frame
    constexpr int NQ = W - 32 * CT >= 32 ? 4 : W >= 8 ? W / 8 : 1;  // 16-byte pieces per lane that hold valid outputs
    const int tf = lane & 31, g = lane >> 5;
    const int i0 = (32 * t + tf) * W + 32 * CT + 4 * g;
    const bool laneOk = W >= 8 || g == 0;         // W = 4, 2: outputs 0 ... W-1 sit in the g = 0 half only
    const unsigned base = (laneOk && !CQ_KO(2)) ? ((unsigned)i0 & ringMask) * 4u : 0x80000000u;
    const int blockLo = 32 * t * W, blockHi = blockLo + 32 * W;
    const bool inside = blockLo >= 0 && blockHi <= dstLen;  // wave-uniform
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] = (hh[4 * q + j] + (hl[4 * q + j] + lh[4 * q + j])) * mul;
            if (!inside) {
                const int i = i0 + 8 * q + j;
                if (i < 0 || i >= dstLen) v[j] = 0.f;
            }
        }
        if (W >= 4) {
            const u32x4 o = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
            __builtin_amdgcn_raw_buffer_store_b128(o, ring, (int)(base + 32u * q), 0, AFX_CQ_RING_AUX);
            if (CQ_KO(12)) __builtin_amdgcn_raw_buffer_store_b128(o, ringDup, (int)(base + 32u * q), 0, AFX_CQ_RING_AUX);
        } else {  // hop 4: two outputs per frame
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[0]), ring, (int)base, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[1]), ring, (int)(base + 4u), 0, 0);
        }
    }
}

// ---- chroma in the same launch (cqt_algorithm.c:484-597 for chromaNum = 12 = bins per octave: class c sums one bin of
// every octave, then the frame's 12 values are normalised).  The octaves of a frame are finished 1 ... 23 steps apart, so
// the sums travel through memory: the output rows themselves hold the partial sums -- level 0 writes its 12 powers, every
// further level adds its own (the partials of the ~23 tiles in flight stay in the L2), level 6 adds, normalises and
// writes the result.  Summation order is therefore highest octave first (the reference and k_cqt_chroma: lowest
// first): the same terms, last-bit differences.  A lane holds bins 3 (lane & 3) .. + 2 of frame 16 h + (lane >> 2) after the
// rows' transposition: the partials are requested before the K loop and arrive during it.
struct PyrChroma {
    __amdgpu_buffer_rsrc_t rows;  // the clip's [T][12] output
    unsigned off[3];              // byte offsets of the lane's three classes inside a row
    float part[2][3];             // partial sums of the two half tiles
};

template <int K>
__device__ __forceinline__ void pyr_chroma_request(PyrChroma &ch, int t, int lane) {
    if (K == 0) return;  // (the first level stores)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            ch.part[h][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                ch.rows, CQ_KO(1) ? (int)(0x80000000u + 4u * (3 * h + j)) : (int)((unsigned)(32 * t + 16 * h + (lane >> 2)) * 48u + ch.off[j]), 0, pyr::AUX_L2));
}

template <int K>
__device__ __forceinline__ void pyr_chroma_add(PyrChroma &ch, const u32x3 (&pieces)[4], int t, int lane, int isMag, int normType) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const u32x3 re = pieces[2 * h], im = pieces[2 * h + 1];
        float v[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float x = __uint_as_float(re[j]), y = __uint_as_float(im[j]);
            float p = __fmaf_rn(x, x, y * y);
            if (isMag) p = sqrtf(p);
            v[j] = K == 0 ? p : ch.part[h][j] + p;
        }
        if (K == AFX_CQT_PYR_LEVELS - 1 && normType != 0) {  // the frame's 12 values sit in the four lanes of a quad
            float red = normType == 2 ? 3.4e38f : 0.f;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float av = fabsf(v[j]);
                if (normType == 1) red = fmaxf(red, av);
                else if (normType == 2) red = fminf(red, av);
                else if (normType == 3) red += av * av;
                else red += av;
            }
            const float r1 = dpp_f(red, 0xB1);  // quad_perm [1,0,3,2]
            red = normType == 1 ? fmaxf(red, r1) : normType == 2 ? fminf(red, r1) : red + r1;
            const float r2 = dpp_f(red, 0x4E);  // quad_perm [2,3,0,1]
            red = normType == 1 ? fmaxf(red, r2) : normType == 2 ? fminf(red, r2) : red + r2;
            if (normType == 3) red = sqrtf(red);
            if (red != 0.f) {
#pragma unroll
                for (int j = 0; j < 3; ++j) v[j] = v[j] / red;
            }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[j]), ch.rows,
                                                  CQ_KO(1) ? (int)(0x80000000u + 4u * (3 * h + j)) : (int)((unsigned)(32 * t + 16 * h + (lane >> 2)) * 48u + ch.off[j]), 0, AFX_CQ_CHROMA_AUX);
    }
}

// ---- one wave's part of the run [t0c, t1c) of `clip`.  PART 0: level K whole (levels 1-6).  Level 0 is shared by two
// waves -- its window is 18 x 16 bytes per lane, too much beside a K loop --: PART 2 (wave 7) PREPARES tile s: window
// -> planes (two buffers, by the tile's parity) and 2^-e in LDS; PART 1 (wave 0) works on tile s - 1 from the planes
// of the step before: resampler, K loop, rows.
template <int K, int PART, bool TIMING>
__device__ __forceinline__ void pyr_wave(const AfxCqtPyramidArgs &a, unsigned char *smem, int lane, float *wgRing, int clip,
                                         int t0c, int t1c, unsigned long long *tim) {
    constexpr int H = 128 >> K;
    using C = CqF16<H>;
    constexpr bool PREP = PART != 1, WORK = PART != 2;
    // The two waves of a SIMD share its matrix pipe: one of them converts first and multiplies behind it, the other (EARLY:
    // levels 4-6) multiplies first -- on planes it prepared at the end of the step before -- and converts behind that.
    constexpr bool EARLY = PART == 0 && K >= 4;
    // ... and a convert-first level wave (levels 1-3) keeps a tile's accumulators over the barrier and stores its rows at the
    // START of the next step, while its partner is already multiplying: at the end of a step both waves of a SIMD used to sit
    // in their epilogues with the matrix pipe idle (phase clocks: 72 % busy on the busiest SIMD).  Same values, one step later.
    constexpr bool LATE = PART == 0 && K >= 1 && K < 4;
    constexpr unsigned RMASK = K == 0 ? 0xffffffffu : (unsigned)pyr::ring_size(K) - 1u;
    constexpr bool DEC = K < AFX_CQT_PYR_LEVELS - 1;  // the last level feeds nobody
    constexpr int KN = K + 1 < AFX_CQT_PYR_LEVELS ? K + 1 : K;
    constexpr int LAG = PART == 2 ? 0 : pyr::lag(K);
    constexpr int ALT = K == 0 ? pyr::plane_off(7) + pyr::TAB_BYTES - pyr::plane_off(0) : 0;  // level 0's second plane buffer
    unsigned char *Bl = smem;
    unsigned char *sig0 = smem + C::B_BYTES + pyr::plane_off(K);
    float *downSlot = reinterpret_cast<float *>(smem + C::B_BYTES + pyr::plane_off(7) + pyr::TAB_BYTES + pyr::ALT_BYTES);
    const unsigned char *tab = smem + C::B_BYTES + pyr::plane_off(7);
    const int i = lane & 31, g = lane >> 5;
    const CqLane L0 = cq_lane_setup(lane, sig0, 12, (6 - K) * 12, a.num, a.timeLength, a.colMul, a.scale, a.octScale[K]);
    const int cpy = i % C::COPIES;
    const unsigned char *aHi0 = sig0 + cpy * C::CS + C::at(i * H + 8 * g, cpy);
    const unsigned char *bHi = Bl + lane * 16;
    const unsigned char *bLo = bHi + C::KS * 64 * 16;
    // tap-table fragment of column c' = i (+ 32 per column tile: 128 bytes down): copy c' mod 4, entry 8 g - 8 (c' >> 2) - 96 (+ 16 ks)
    const unsigned char *tHi = tab + (i & 3) * pyr::TAB_COPY + 2 * (8 * g - 8 * (i >> 2) - 96);
    const unsigned char *tLo = tHi + pyr::TAB_PLANE;
    const int valid = a.valid[K], len = a.len[K];
    // level 0: the clip (the bounds check supplies the zeros on both sides); other levels: the ring, zeros inside
    const __amdgpu_buffer_rsrc_t rsrc =
        K == 0 ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x + (long long)clip * a.xStride), 0, len * 4, RSRC_RAW)
               : __builtin_amdgcn_make_buffer_rsrc(wgRing + pyr::ring_off(K), 0, pyr::ring_size(K) * 4, RSRC_RAW);
    const __amdgpu_buffer_rsrc_t ringNext =
        __builtin_amdgcn_make_buffer_rsrc(wgRing + pyr::ring_off(KN), 0, pyr::ring_size(KN) * 4, RSRC_RAW);
    constexpr unsigned NMASK = (unsigned)pyr::ring_size(KN) - 1u;
    // (price build, bit 12: a second copy of every ring store, into this workgroup's slice of the chroma plane)
    const __amdgpu_buffer_rsrc_t ringDup = __builtin_amdgcn_make_buffer_rsrc(
        (CQ_KO(12) && a.chroma) ? a.chroma + (size_t)blockIdx.x * AFX_CQT_PYR_RING_FLOATS + pyr::ring_off(KN) : wgRing, 0,
        (CQ_KO(12) && a.chroma) ? pyr::ring_size(KN) * 4 : 0, RSRC_RAW);
    float *outRe = a.outRe + (long long)clip * a.outStride, *outIm = a.outIm + (long long)clip * a.outStride;
    constexpr int NW = PREP ? C::NV : 1;
    u32x4 wnd[NW];
    const bool chromaOn = WORK && a.chroma != nullptr;
    PyrChroma ch;
    ch.rows = __builtin_amdgcn_make_buffer_rsrc(a.chroma ? a.chroma + (long long)clip * a.timeLength * 12 : a.outRe, 0,
                                                a.chroma ? a.timeLength * 48 : 0, RSRC_RAW);
#pragma unroll
    for (int j = 0; j < 3; ++j) ch.off[j] = 4u * (unsigned)a.chromaClass[3 * (lane & 3) + j];
    auto fetch = [&](int t) {  // the window of a tile, requested one step ahead
        const int p0 = t * 32 * H - (C::N >> 1);
#pragma unroll
        for (int u = 0; u < (PREP ? C::NV : 0); ++u) {
            const int pos = p0 + 4 * (lane + 64 * u);
            // (the last register reaches past the window: those 16-byte pieces are not requested -- zeros, like every use of them
            //  already assumes; in a ring they would be the block the level above is writing in this very step: a read nobody uses,
            //  but a race in the lane emulation under ThreadSanitizer, tests/test_emulated_kernels.py)
            const bool past = 4 * (lane + 64 * u) >= C::S;
            wnd[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (CQ_KO(K == 0 ? 8 : 9) || past) ? (int)(0x80000000u + 16u * u) : (int)(((unsigned)pos & RMASK) * 4u), 0, K == 0 ? pyr::AUX_STREAM : pyr::AUX_L2);
        }
    };
    // the wave works on tile s - LAG: the octave's rows for the tiles of the run, the resampler also for the tiles
    // whose block a lower octave of the run reaches into
    auto has_oct = [&](int t) { return t >= t0c && t < t1c; };
    auto has_dec = [&](int t) { return DEC && t >= t0c - pyr::need_back(K) && t <= t1c + pyr::need_ahead(K); };
    auto is_empty = [&](int t) {  // nothing of the signal in the tile's window: zeros out
        const int p0 = t * 32 * H - (C::N >> 1);
        return p0 + C::S <= 0 || p0 >= len;
    };
    auto is_live = [&](int t) { return has_oct(t) || (has_dec(t) && !is_empty(t)); };
    // window registers -> planes, the WHOLE signal (the resampler reads what the framing rule drops); returns 2^-e
    auto prepare = [&](int t, unsigned char *sig) {
        const int p0 = t * 32 * H - (C::N >> 1);
        if (!has_oct(t)) {
            // a tile resampled but not transformed: row t' reads window positions t' H + 225 ... t' H + H + 285, i.e.
            // the level's samples 32 t H - 31 ... 32 (t+1) H + 29, and only their blocks are sure to be written (what
            // else the window covers is not this run's: out of the tile exponent with it)
            const int lo = p0 + 224, hi = p0 + 32 * H + 288;
#pragma unroll
            for (int u = 0; u < NW; ++u) {
                const int pos = p0 + 4 * (lane + 64 * u);
                if (pos < lo || pos >= hi) wnd[u].x = 0u;
                if (pos + 1 < lo || pos + 1 >= hi) wnd[u].y = 0u;
                if (pos + 2 < lo || pos + 2 >= hi) wnd[u].z = 0u;
                if (pos + 3 < lo || pos + 3 >= hi) wnd[u].w = 0u;
            }
        }
        const int e = cq_window_exponent<H>(reinterpret_cast<const u32x4(&)[C::NV]>(wnd), lane);
        const float up = __uint_as_float((unsigned)(e + 127) << 23);
        wave_lds_order();
        if (!CQ_KO(5)) cq_convert_window<H>(reinterpret_cast<const u32x4(&)[C::NV]>(wnd), up, sig, lane);
        wave_lds_order();
        return __uint_as_float((unsigned)(127 - e) << 23);
    };
    const int s0 = t0c - pyr::LEAD, s1 = t1c + pyr::DRAIN;
    PyrClock<TIMING> clk;
    // the two waves of a SIMD want its matrix pipe at the same time for most of a step, and the arbiter serves the OLDER wave
    // first: the multiply-first wave -- the one with the longer chain behind its products (rows, then the next tile's planes)
    // -- got what the other left (phase clocks: 65 cycles per MFMA against 35).  It goes first instead.
    __builtin_amdgcn_s_setprio(EARLY ? 2 : 0);
    float downNext = 0.f;  // (EARLY waves: 2^-e of the planes prepared at the end of the step before)
    f32x16 hh, hl, lh;
    bool pend = false;     // (LATE waves: the tile of the step before still has to leave)
    float pendDown = 0.f;
    int pendT = 0;
    auto rows_out = [&](float dn, int tt, int bf) {
        CqLane L = L0;
        L.epiW += bf * ALT;
        L.epiR += bf * ALT;
        u32x3 pieces[4];
        cq_store_tile<true, AFX_CQ_ROW_AUX>(hh, hl, lh, dn, L, outRe, outIm, tt * 32, pieces);
        if (chromaOn) pyr_chroma_add<K>(ch, pieces, tt, lane, a.chromaMag, a.chromaNorm);
    };
    // a convert-first wave finds the window of step s in registers requested during step s - 1; the run's first step has
    // no step before it.  Only level 0's preparer is live there (LAG 0, need_back(0) == LEAD: tile t0c - 9 of a run that
    // starts mid-clip is resampled); its window comes from the clip itself, never from a ring not written yet.
    static_assert(K == 0 || pyr::need_back(K) < pyr::LEAD + LAG, "a level wave must not be live at the first step of a run");
    if (PREP && !EARLY && is_live(s0 - LAG)) fetch(s0 - LAG);
    for (int s = s0; s <= s1; ++s) {
        const int t = s - LAG;
        const bool oct = has_oct(t), dec = has_dec(t), empty = is_empty(t), live = is_live(t), next = is_live(t + 1);  // wave-uniform
        const int buf = K == 0 ? (t & 1) : 0;
        unsigned char *sig = sig0 + buf * ALT;
        const unsigned char *aHi = aHi0 + buf * ALT;
        const int p0 = t * 32 * H - (C::N >> 1);
        float down = downNext;
        if (LATE && pend) {  // the rows of the tile before: through the planes (still that tile's), before they are converted anew
            rows_out(pendDown, pendT, 0);
            pend = false;
            clk.lap(7);
        }
        if (EARLY && next) fetch(t + 1);  // (the same blocks a convert-first wave asks for in this step: all written by step s - 1)
        if (PREP && !EARLY && live) {
            if (TIMING) {  // time the wait for the window apart from the conversion
#pragma unroll
                for (int u = 0; u < NW; ++u) PIN(wnd[u]);
                clk.lap(0);
            }
            down = prepare(t, sig);
            if (PART == 2 && lane == 0) downSlot[buf] = down;
            clk.lap(1);
        }
        if (PART == 1 && live) down = downSlot[buf];
        if (PREP && !EARLY && next) fetch(t + 1);
        clk.lap(2);
        // the resampler first: its stores have the whole K loop to reach the L2 (the wait behind the loop covers them: no
        // wait at the barrier), and the octave's epilogue transposes through the planes
        if (WORK && DEC && dec) {
            f32x16 dh, dl, dm;
            if (empty || CQ_KO(6)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) dh[r] = dl[r] = dm[r] = 0.f;
            } else {
                pyr_dec_loop<H, 0>(aHi, aHi + C::PART, tHi, tLo, dh, dl, dm);
            }
            pyr_dec_store<H, 0>(dh, dl, dm, down * a.decMul, t, lane, ringNext, NMASK, a.len[KN], ringDup);
            if (H / 2 > 32) {
                if (!empty && !CQ_KO(6)) pyr_dec_loop<H, 1>(aHi, aHi + C::PART, tHi - 128, tLo - 128, dh, dl, dm);  // outputs 32 ... 63: 64 table entries down
                pyr_dec_store<H, 1>(dh, dl, dm, down * a.decMul, t, lane, ringNext, NMASK, a.len[KN], ringDup);
            }
        }
        clk.lap(5);
        if (chromaOn && oct) pyr_chroma_request<K>(ch, t, lane);
        if (WORK && oct) {
            if (p0 + C::S > valid) cq_zero_samples<H>(sig, valid - p0 > 0 ? valid - p0 : 0, (len < p0 + C::S ? len : p0 + C::S) - p0, lane);
            wave_lds_order();
            cq_kloop<H>(aHi, aHi + C::PART, bHi, bLo, hh, hl, lh);
        }
        clk.lap(3);
        // the prefetched window is in its registers, the block in the L2, BEFORE the step's barrier (the window's ring
        // blocks are overwritten three steps on)
        if (PREP && K != 0) {
#pragma unroll
            for (int u = 0; u < NW; ++u) PIN(wnd[u]);
        }
        if (WORK && DEC) VM_WAIT_ALL();
        clk.lap(4);
        if (WORK && oct) {
            if (LATE) {
                pend = true;
                pendDown = down;
                pendT = t;
            } else {
                rows_out(down, t, buf);
            }
        }
        clk.lap(7);
        if (EARLY && next) {  // the next tile's planes now: the matrix-core work of the next step starts at its barrier
            downNext = prepare(t + 1, sig);
            clk.lap(1);
        }
        pyr_barrier();
        clk.lap(6);
        if (TIMING && K == 0 && PART == 1 && blockIdx.x == 0 && tim && lane == 0) {
            // the step's end as an absolute stamp: entry i of the series lives in the spare rows 9, 10 of block i / 16
            const int i = s - s0;
            if (i < 4096) (tim - (size_t)(threadIdx.x >> 6) * 8)[((size_t)(i >> 4) * 11 + 9 + ((i >> 3) & 1)) * 8 + (i & 7)] = __builtin_amdgcn_s_memtime();
        }
    }
    if (LATE && pend) rows_out(pendDown, pendT, 0);  // (not reached: a run ends with DRAIN steps without rows)
    clk.flush(tim, lane);
}

// the runs of this workgroup, one after the other
template <int K, int PART, bool TIMING>
__device__ __forceinline__ void pyr_role(const AfxCqtPyramidArgs &a, unsigned char *smem, int lane, float *wgRing, int items) {
    unsigned long long *tim = TIMING && a.timing ? a.timing + ((size_t)blockIdx.x * 11 + (threadIdx.x >> 6)) * 8 : nullptr;
    const int nT = (a.timeLength + 31) / 32;
    for (int it = blockIdx.x; it < items; it += gridDim.x) {
        const int clip = it / a.chunksPerClip, chunk = it - clip * a.chunksPerClip;
        const int t0c = chunk * a.tilesPerChunk;
        const int t1c = t0c + a.tilesPerChunk < nT ? t0c + a.tilesPerChunk : nT;
        if (t0c >= t1c) continue;  // (uniform over the workgroup)
        pyr_wave<K, PART, TIMING>(a, smem, lane, wgRing, clip, t0c, t1c, tim);
    }
}

template <bool TIMING>
__global__ __launch_bounds__(64 * pyr::WAVES) void k_cqt_pyramid(AfxCqtPyramidArgs a, int items) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned long long tKernel0 = TIMING ? __builtin_amdgcn_s_memtime() : 0ull;
    cq_image_to_lds(a.timeKernelH, smem_raw, tid, 64 * pyr::WAVES);
    {
        const unsigned *src = reinterpret_cast<const unsigned *>(a.decTab);
        unsigned *dst = reinterpret_cast<unsigned *>(smem_raw + CqF16<128>::B_BYTES + pyr::plane_off(7));
        for (int e = tid; e < pyr::TAB_BYTES / 4; e += 64 * pyr::WAVES) dst[e] = src[e];
    }
    __syncthreads();
    float *wgRing = a.ring + (size_t)blockIdx.x * AFX_CQT_PYR_RING_FLOATS;
    // (the loop over the workgroup's runs sits INSIDE every role: around the switch, the compiler hoists the per-lane
    // constants of all eight roles in front of it and spills them)
    // wave ids w and w + 4 share a SIMD: (level 0's multiplier, its preparer), (1, 6), (2, 5), (3, 4) -- 144 + 0, 120 + 96,
    // 114 + 111, 111 + 111 MFMAs per step, one convert-first and one multiply-first wave each
    switch (wave) {
        case 0: pyr_role<0, 1, TIMING>(a, smem_raw, lane, wgRing, items); break;
        case 4: pyr_role<0, 2, TIMING>(a, smem_raw, lane, wgRing, items); break;
        case 1: pyr_role<1, 0, TIMING>(a, smem_raw, lane, wgRing, items); break;
        case 5: pyr_role<6, 0, TIMING>(a, smem_raw, lane, wgRing, items); break;
        case 2: pyr_role<2, 0, TIMING>(a, smem_raw, lane, wgRing, items); break;
        case 6: pyr_role<5, 0, TIMING>(a, smem_raw, lane, wgRing, items); break;
        case 3: pyr_role<3, 0, TIMING>(a, smem_raw, lane, wgRing, items); break;
        default: pyr_role<4, 0, TIMING>(a, smem_raw, lane, wgRing, items); break;
    }
    if (TIMING && a.timing && tid == 0) {  // row 8 of the block: kernel cycles (summed over launches), start / end stamps of the last one
        unsigned long long *r8 = a.timing + ((size_t)blockIdx.x * 11 + 8) * 8;
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        r8[0] += t1 - tKernel0;
        r8[1] = tKernel0;
        r8[2] = t1;
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
extern "C" int afxk_cqt_pyramid_plan(int batch, int timeLength, int maxTiles, int *chunksPerClip, int *tilesPerChunk) {
    if (batch <= 0 || timeLength <= 0) return 0;
    const int nT = (timeLength + 31) / 32;
    // runs of one clip: as many as fill the CUs, but long enough that the 30 steps of lead-in and drain stay small
    int cpc = AFX_CQT_PYR_MAX_WGS / batch;
    if (cpc > nT / 48) cpc = nT / 48;
    if (cpc < 1) cpc = 1;
    int tpc = (nT + cpc - 1) / cpc;
    if (maxTiles > 0 && maxTiles < tpc) tpc = maxTiles;  // (tests: short runs, so that small inputs cross run boundaries)
    cpc = (nT + tpc - 1) / tpc;  // no empty runs
    if (chunksPerClip) *chunksPerClip = cpc;
    if (tilesPerChunk) *tilesPerChunk = tpc;
    const long long items = (long long)batch * cpc;
    return (int)(items < AFX_CQT_PYR_MAX_WGS ? items : AFX_CQT_PYR_MAX_WGS);
}

extern "C" int afxk_cqt_pyramid(const AfxCqtPyramidArgs *a, void *stream) {
    if (!a->x || !a->timeKernelH || !a->colMul || !a->scale || !a->outRe || !a->outIm || !a->ring || !a->decTab) return AFX_ERR_ARG;
    if (a->batch <= 0 || a->timeLength <= 0 || a->chunksPerClip <= 0 || a->tilesPerChunk <= 0) return AFX_ERR_ARG;
    // 32-bit byte offsets inside one clip's signal and one clip's output plane
    if (a->len[0] > (1 << 28) || (long long)a->timeLength * a->num > (1LL << 28)) return AFX_ERR_UNSUPPORTED;
    const long long items = (long long)a->batch * a->chunksPerClip;
    if (items > 0x7fffffffLL) return AFX_ERR_UNSUPPORTED;
    const unsigned grid = (unsigned)(items < AFX_CQT_PYR_MAX_WGS ? items : AFX_CQT_PYR_MAX_WGS);
#ifdef AFX_EXPERIMENTS
    if (a->timing) {  // the instrumented instantiation (tools/pyr_phases.py; measurement builds only: make EXTRA=-DAFX_EXPERIMENTS)
        AFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_cqt_pyramid<true>), hipFuncAttributeMaxDynamicSharedMemorySize, pyr::LDS_BYTES));
        hipLaunchKernelGGL(k_cqt_pyramid<true>, dim3(grid), dim3(64 * pyr::WAVES), pyr::LDS_BYTES, (hipStream_t)stream, *a, (int)items);
        AFX_LAUNCH_CHECK("k_cqt_pyramid<timing>");
        return AFX_OK;
    }
#else
    if (a->timing) return AFX_ERR_UNSUPPORTED;
#endif
    AFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_cqt_pyramid<false>), hipFuncAttributeMaxDynamicSharedMemorySize, pyr::LDS_BYTES));
    hipLaunchKernelGGL(k_cqt_pyramid<false>, dim3(grid), dim3(64 * pyr::WAVES), pyr::LDS_BYTES, (hipStream_t)stream, *a, (int)items);
    AFX_LAUNCH_CHECK("k_cqt_pyramid");
    return AFX_OK;
}
