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
    {   // image -> LDS: all of a thread's 16-byte loads in flight, then the stores
        const float4 *src = reinterpret_cast<const float4 *>(a.timeKernelH);
        float4 *dstl = reinterpret_cast<float4 *>(Bl);
        constexpr int Q = C::B_BYTES / 16;  // 4096
        for (int e0 = tid; e0 < Q; e0 += 8 * nth) {
            float4 tq[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) tq[u] = src[(e0 + u * nth) & (Q - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) dstl[(e0 + u * nth) & (Q - 1)] = tq[u];  // Q is a multiple of 8 nth
        }
    }
    __syncthreads();

    // Output: per clip one raw buffer per plane, T x num floats; rows past timeLength and the padding columns
    // fall out of range and are dropped by the bounds check, so every tile issues the same NSTORE stores
    // (the compiler can then count them: waiting for the prefetched window does not wait for stores).
    const bool colOk = i < 2 * a.rows, colIm = i >= a.rows;
    const int colOff = a.colBase + (colOk ? (colIm ? i - a.rows : i) : 0);
    const float colMul = a.colMul[i] * (a.octScale / a.scale[colOff]);  // 2^-s_j sqrt(2^k) / sqrt(len_j)
    const unsigned OOR = 0x80000000u;
    const unsigned planeBytes = (unsigned)a.timeLength * (unsigned)a.num * 4u;
    const unsigned rowBytes = (unsigned)a.num * 4u;
    // generic epilogue: lane = column, 16 rows
    const unsigned laneOff = (unsigned)(4 * g * a.num + colOff) * 4u;
    const unsigned voffRe = (colOk && !colIm) ? laneOff : OOR;
    const unsigned voffIm = (colOk && colIm) ? laneOff : OOR;
    // transposed epilogue (R12): the lane's 16 values go to epi[frame][plane][piece][word] ...
    const int jj = i < 12 ? i : i - 12;
    unsigned char *epiW = sig + (i < 24 ? (i >= 12 ? 64 : 0) + (jj / 3) * 16 + (jj % 3) * 4 : (i - 24) * 16 + 12) + 4 * g * 128;
    // ... and leave as: store q, lane L -> frame 16 (q >> 1) + (L >> 2), plane q & 1, bins 3 (L & 3) .. + 2
    const unsigned char *epiR = sig + (lane >> 2) * 128 + (lane & 3) * 16;
    const unsigned voff12 = (unsigned)(lane >> 2) * rowBytes + (unsigned)(a.colBase + 3 * (lane & 3)) * 4u;
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
        float peak = 0.f;
#pragma unroll
        for (int u = 0; u < C::NV; ++u) {
            const float4 v = __builtin_bit_cast(float4, wnd[u]);
            peak = fmaxf(peak, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        }
        // wave maximum without LDS traffic: four DPP steps give every row of 16 lanes its maximum, the four
        // rows meet on the scalar unit (non-negative floats order like their bit patterns)
        peak = fmaxf(peak, dpp_f(peak, 0xB1));   // quad_perm [1,0,3,2]
        peak = fmaxf(peak, dpp_f(peak, 0x4E));   // quad_perm [2,3,0,1]
        peak = fmaxf(peak, dpp_f(peak, 0x141));  // row_half_mirror
        peak = fmaxf(peak, dpp_f(peak, 0x140));  // row_mirror
        const unsigned pk = __float_as_uint(peak);
        const unsigned p01 = max((unsigned)__builtin_amdgcn_readlane((int)pk, 0), (unsigned)__builtin_amdgcn_readlane((int)pk, 16));
        const unsigned p23 = max((unsigned)__builtin_amdgcn_readlane((int)pk, 32), (unsigned)__builtin_amdgcn_readlane((int)pk, 48));
        const int pe = (int)((max(p01, p23) >> 23) & 0xff) - 127;  // floor(log2 peak) of a normal
        int e = 13 - pe;
        if (pe == -127) e = 0;  // zero / subnormal window
        e = e > 126 ? 126 : e;
        const float up = __uint_as_float((unsigned)(e + 127) << 23);      // 2^e
        const float down = __uint_as_float((unsigned)(127 - e) << 23);    // 2^-e
        // ---- window -> (xh, xl) planes (every copy)
        wave_lds_order();  // the fragment / epilogue reads of the previous tile are done
#pragma unroll
        for (int u = 0; u < C::NV; ++u) {
            const int s = 4 * (lane + 64 * u);
            if (s < C::S) {
                const float4 v = __builtin_bit_cast(float4, wnd[u]);
                unsigned hi0, hi1, lo0, lo1;
                split_pair(v.x, v.y, up, hi0, lo0);
                split_pair(v.z, v.w, up, hi1, lo1);
                const int base = C::MARGIN + 2 * s + (C::PAD ? 16 * (s / H) : 0);
#pragma unroll
                for (int c = 0; c < C::COPIES; ++c) {
                    unsigned char *d = sig + c * C::CS + base - 2 * c * H;
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
        wave_lds_order();
        if (t + stride < totalTiles) fetch(t + stride);

        // ---- K loop: 32 steps x (xh gh, xh gl, xl gh), operands two steps ahead
        f32x16 hh, hl, lh;
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
        // ---- D layout: col = lane & 31, row = (r&3) + 8 (r>>2) + 4 (lane>>5)
        {
            const long long po = (long long)clip * a.outStride;
            const __amdgpu_buffer_rsrc_t rRe = __builtin_amdgcn_make_buffer_rsrc(a.outRe + po, 0, (int)planeBytes, RSRC_RAW);
            const __amdgpu_buffer_rsrc_t rIm = __builtin_amdgcn_make_buffer_rsrc(a.outIm + po, 0, (int)planeBytes, RSRC_RAW);
            const float mul = down * colMul;
            const unsigned tileOff = (unsigned)t0 * rowBytes;
            if (R12) {
                wave_lds_order();  // the last fragment reads are done: the window region is free
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    *reinterpret_cast<float *>(epiW + ((r & 3) + 8 * (r >> 2)) * 128) = (hh[r] + (hl[r] + lh[r])) * mul;
                wave_lds_order();
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const u32x4 v = *reinterpret_cast<const u32x4 *>(epiR + (q >> 1) * 2048 + (q & 1) * 64);
                    const u32x3 v3 = {v.x, v.y, v.z};
                    __builtin_amdgcn_raw_buffer_store_b96(v3, (q & 1) ? rIm : rRe, voff12 + tileOff + (unsigned)(q >> 1) * 16u * rowBytes, 0, 0);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned ro = tileOff + (unsigned)((r & 3) + 8 * (r >> 2)) * rowBytes;  // scalar
                    const unsigned v = __float_as_uint((hh[r] + (hl[r] + lh[r])) * mul);
                    __builtin_amdgcn_raw_buffer_store_b32(v, rRe, voffRe + ro, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(v, rIm, voffIm + ro, 0, 0);
                }
            }
        }
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
