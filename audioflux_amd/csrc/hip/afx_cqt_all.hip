// afx_cqt_all.hip -- ALL seven octaves of the default CQT plan (+ its chroma) in one launch.
//
// STATUS: written at the end of round 2 without hardware access.  Compiled, never run: switched OFF by default
// (AFX_CQT_FUSED=1 enables it).  The per-tile body is the one of k_cqt_octave_f16
// (afx_cqt_f16.hip: split-f16 operands on the f16 matrix cores, measured and parity-tested); what is new here is
// the loop around it.
//
// Why: in the recursive CQT every octave uses the SAME kernel image (cqt_algorithm.c:999-1041 -- the top octave's
// spectral kernels applied to the decimated signal), so one persistent workgroup can keep the 64 KB image in LDS
// and walk a 32-frame block through all seven levels without a barrier:
//   * the seven 48-byte pieces of an output row are written by one wave within a few tens of microseconds and
//     merge in its XCD's write-back L2 -- no partial-line traffic to HBM (the per-octave launches need passes of
//     <= 448 MB of output for that), and seven launch tails become one;
//   * the next level's window is prefetched under the current level's K loop;
//   * the chroma of the block is accumulated as the octaves go by (ascending octave = ascending bin: the summation
//     order of k_cqt_chroma, so the result is the same bit for bit) -- k_cqt_chroma and its re-read of the CQT
//     rows disappear.
// Levels: level l = signal decimated l times (device pointers from the host's decimation chain), hop 128 >> l,
// octave 6 - l, output columns 12 (6 - l) ... + 11.  Processed lowest octave first.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "afx_device.h"
#include "afx_hipcheck.h"
#include "afx_pkmath.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
constexpr int RSRC_RAW = 0x00020000;

// window layout of one level (same as CqF16 in afx_cqt_f16.hip; tools/proto_cqt_f16.py pins the algebra)
template <int H>
struct Lay {
    static constexpr int N = 512, KS = N / 16;
    static constexpr int COPIES = H >= 8 ? 1 : 8 / H;
    static constexpr bool PAD = H >= 16;
    static constexpr int S = 31 * H + N;
    static constexpr int NV = (S + 255) / 256;
    static constexpr int MARGIN = 16;
    static constexpr int RAW = MARGIN + 2 * (S + 8) + (PAD ? 16 * (S / H + 1) : 0);
    static constexpr int CS = COPIES == 1 ? ((RAW + 15) & ~15) : ((RAW + 255) & ~255) + (COPIES == 4 ? 64 : 128);
    static constexpr int PART = COPIES * CS;
    __host__ __device__ static constexpr int at(int s, int c) { return MARGIN + 2 * (s - c * H) + (PAD ? 16 * (s / H) : 0); }
    __host__ __device__ static constexpr int step(int ks) { return 32 * ks + (PAD ? 16 * ((16 * ks) / H) : 0); }
};
constexpr int SIG_BYTES = 2 * Lay<128>::PART;  // the largest window (hop 128); the 4 KB epilogue image reuses it
constexpr int ACC_BYTES = 32 * 12 * 4;         // chroma accumulator of the block
constexpr int WAVE_BYTES = SIG_BYTES + ACC_BYTES;
constexpr int B_BYTES = 2 * 32 * 64 * 16;
static_assert(2 * Lay<2>::PART <= SIG_BYTES && 2 * Lay<64>::PART <= SIG_BYTES && SIG_BYTES % 16 == 0, "window region");

__device__ __forceinline__ float dpp_mov(float v, int ctrl) {
    switch (ctrl) {
        case 0xB1: return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xf, 0xf, true));
        case 0x4E: return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xf, 0xf, true));
        case 0x141: return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xf, 0xf, true));
        default: return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xf, 0xf, true));
    }
}

__device__ __forceinline__ void split_pair(float x0, float x1, float up, unsigned &hi, unsigned &lo) {
#ifndef AFX_HOST_EMULATION
    asm("v_fma_mixlo_f16 %0, %2, %4, 0\n\t"
        "v_fma_mixhi_f16 %0, %3, %4, 0\n\t"
        "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %3, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(hi), "=&v"(lo)
        : "v"(x0), "v"(x1), "s"(up));
#else  // tests/emu (the kernel compiled for the host): the same four roundings in C
    const _Float16 h0 = (_Float16)(x0 * up), h1 = (_Float16)(x1 * up);
    const _Float16 l0 = (_Float16)(x0 * up - (float)h0), l1 = (_Float16)(x1 * up - (float)h1);
    unsigned short b[4];
    __builtin_memcpy(&b[0], &h0, 2), __builtin_memcpy(&b[1], &h1, 2), __builtin_memcpy(&b[2], &l0, 2), __builtin_memcpy(&b[3], &l1, 2);
    hi = (unsigned)b[0] | ((unsigned)b[1] << 16);
    lo = (unsigned)b[2] | ((unsigned)b[3] << 16);
#endif
}

// per-lane constants of the kernel
struct Lane {
    int lane, i, g;
    unsigned char *sig;          // the wave's window / epilogue region
    float *acc;                  // the wave's chroma accumulator [32][12]
    const unsigned char *bHi, *bLo;
    unsigned char *epiW;         // transposed epilogue: where this lane's accumulator words go
    const unsigned char *epiR;   // ... and where its 12-byte store pieces come from
    unsigned rowBytes, planeBytes;
    float mul[7];                // per level: 2^-s_j sqrt(2^level) / sqrt(len_j) of this lane's column
};

template <int H>
__device__ __forceinline__ void fetch_window(const AfxCqtAllArgs &a, const Lane &L, int clip, int t0, u32x4 (&w)[Lay<H>::NV]) {
    constexpr int level = H == 128 ? 0 : H == 64 ? 1 : H == 32 ? 2 : H == 16 ? 3 : H == 8 ? 4 : H == 4 ? 5 : 6;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.x[level] + (long long)clip * a.xStride[level]), 0, a.validLength[level] * 4, RSRC_RAW);
    const int p0 = t0 * H - 256;
#pragma unroll
    for (int u = 0; u < Lay<H>::NV; ++u)
        w[u] = __builtin_amdgcn_raw_buffer_load_b128(rx, (p0 + 4 * (L.lane + 64 * u)) * 4, 0, 0);
}

// One 32-frame tile of level `H` (window already in w), results to the clip's output rows; with CHROMA the
// tile's |Q|^2 (or |Q|) is added to the block's chroma accumulator.  `prefetch()` issues the loads of the tile
// that follows (next level, or the next block's lowest level) under this tile's K loop.
template <int H, bool CHROMA, class Prefetch>
__device__ __forceinline__ void tile(const AfxCqtAllArgs &a, const Lane &L, int clip, int t0,
                                     u32x4 (&w)[Lay<H>::NV], Prefetch prefetch) {
    using C = Lay<H>;
    constexpr int level = H == 128 ? 0 : H == 64 ? 1 : H == 32 ? 2 : H == 16 ? 3 : H == 8 ? 4 : H == 4 ? 5 : 6;
    constexpr int octave = 6 - level;
    // ---- tile exponent
    float peak = 0.f;
#pragma unroll
    for (int u = 0; u < C::NV; ++u) {
        const float4 v = __builtin_bit_cast(float4, w[u]);
        peak = fmaxf(peak, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    peak = fmaxf(peak, dpp_mov(peak, 0xB1));
    peak = fmaxf(peak, dpp_mov(peak, 0x4E));
    peak = fmaxf(peak, dpp_mov(peak, 0x141));
    peak = fmaxf(peak, dpp_mov(peak, 0x140));
    const unsigned pk = __float_as_uint(peak);
    const unsigned p01 = max((unsigned)__builtin_amdgcn_readlane((int)pk, 0), (unsigned)__builtin_amdgcn_readlane((int)pk, 16));
    const unsigned p23 = max((unsigned)__builtin_amdgcn_readlane((int)pk, 32), (unsigned)__builtin_amdgcn_readlane((int)pk, 48));
    const int pe = (int)((max(p01, p23) >> 23) & 0xff) - 127;
    int e = 13 - pe;
    if (pe == -127) e = 0;
    e = e > 126 ? 126 : e;
    const float up = __uint_as_float((unsigned)(e + 127) << 23);
    const float down = __uint_as_float((unsigned)(127 - e) << 23);
    // ---- window -> (xh, xl) planes
    wave_lds_order();  // the previous tile's fragment / epilogue / chroma reads are done
#pragma unroll
    for (int u = 0; u < C::NV; ++u) {
        const int s = 4 * (L.lane + 64 * u);
        if (s < C::S) {
            const float4 v = __builtin_bit_cast(float4, w[u]);
            unsigned hi0, hi1, lo0, lo1;
            split_pair(v.x, v.y, up, hi0, lo0);
            split_pair(v.z, v.w, up, hi1, lo1);
            const int base = C::MARGIN + 2 * s + (C::PAD ? 16 * (s / H) : 0);
#pragma unroll
            for (int c = 0; c < C::COPIES; ++c) {
                unsigned char *d = L.sig + c * C::CS + base - 2 * c * H;
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
    prefetch();
    // ---- K loop
    const int cpy = L.i % C::COPIES;
    const unsigned char *aHi = L.sig + cpy * C::CS + C::at(L.i * H + 8 * L.g, cpy);
    const unsigned char *aLo = aHi + C::PART;
    f32x16 hh, hl, lh;
#pragma unroll
    for (int r = 0; r < 16; ++r) hh[r] = hl[r] = lh[r] = 0.f;
    h8 ah[3], al[3], bh[3], bl[3];
    auto load = [&](int ks, int slot) {
        ah[slot] = *reinterpret_cast<const h8 *>(aHi + C::step(ks));
        al[slot] = *reinterpret_cast<const h8 *>(aLo + C::step(ks));
        bh[slot] = *reinterpret_cast<const h8 *>(L.bHi + 1024 * ks);
        bl[slot] = *reinterpret_cast<const h8 *>(L.bLo + 1024 * ks);
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
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- transposed epilogue: epi[frame][plane][piece][4 words], four 12-byte stores per lane
    const long long po = (long long)clip * a.outStride;
    const __amdgpu_buffer_rsrc_t rRe = __builtin_amdgcn_make_buffer_rsrc(a.outRe + po, 0, (int)L.planeBytes, RSRC_RAW);
    const __amdgpu_buffer_rsrc_t rIm = __builtin_amdgcn_make_buffer_rsrc(a.outIm + po, 0, (int)L.planeBytes, RSRC_RAW);
    const float mul = down * L.mul[level];
    wave_lds_order();
#pragma unroll
    for (int r = 0; r < 16; ++r)
        *reinterpret_cast<float *>(L.epiW + ((r & 3) + 8 * (r >> 2)) * 128) = (hh[r] + (hl[r] + lh[r])) * mul;
    wave_lds_order();
    const unsigned voff = (unsigned)(L.lane >> 2) * L.rowBytes + (unsigned)(12 * octave + 3 * (L.lane & 3)) * 4u +
                          (unsigned)t0 * L.rowBytes;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const u32x4 v = *reinterpret_cast<const u32x4 *>(L.epiR + (q >> 1) * 2048 + (q & 1) * 64);
        const u32x3 v3 = {v.x, v.y, v.z};
        __builtin_amdgcn_raw_buffer_store_b96(v3, (q & 1) ? rIm : rRe, voff + (unsigned)(q >> 1) * 16u * L.rowBytes, 0, 0);
    }
    // ---- chroma: lane f < 32 owns frame f; its 12 bins in ascending order, class from the host's table
    if (CHROMA && L.lane < 32) {
        const unsigned char *row = L.sig + L.lane * 128;
        float re[12], im[12];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const float4 a4 = *reinterpret_cast<const float4 *>(row + 16 * p);
            const float4 b4 = *reinterpret_cast<const float4 *>(row + 64 + 16 * p);
            re[3 * p] = a4.x; re[3 * p + 1] = a4.y; re[3 * p + 2] = a4.z;
            im[3 * p] = b4.x; im[3 * p + 1] = b4.y; im[3 * p + 2] = b4.z;
        }
        float *accRow = L.acc + L.lane * 12;
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const int c = a.cls[12 * octave + j];  // uniform
            float v = __fmaf_rn(re[j], re[j], im[j] * im[j]);  // k_cqt_chroma's expression
            if (a.isMag) v = sqrtf(v);
            accRow[c] += v;
        }
    }
}

template <bool CHROMA>
__global__ __launch_bounds__(256) void k_cqt_all_f16(AfxCqtAllArgs a, int tilesPerClip) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, nth = blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), waves = nth >> 6;
    unsigned char *Bl = smem_raw;
    Lane L;
    L.lane = tid & 63;
    L.i = L.lane & 31;
    L.g = L.lane >> 5;
    L.sig = smem_raw + B_BYTES + wave * WAVE_BYTES;
    L.acc = reinterpret_cast<float *>(L.sig + SIG_BYTES);
    {   // image -> LDS
        const float4 *src = reinterpret_cast<const float4 *>(a.imageH);
        float4 *dstl = reinterpret_cast<float4 *>(Bl);
        constexpr int Q = B_BYTES / 16;
        for (int e0 = tid; e0 < Q; e0 += 8 * nth) {
            float4 tq[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) tq[u] = src[(e0 + u * nth) & (Q - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) dstl[(e0 + u * nth) & (Q - 1)] = tq[u];
        }
    }
    __syncthreads();
    L.bHi = Bl + L.lane * 16;
    L.bLo = L.bHi + 32 * 64 * 16;
    const int jj = L.i < 12 ? L.i : L.i - 12;
    L.epiW = L.sig + (L.i < 24 ? (L.i >= 12 ? 64 : 0) + (jj / 3) * 16 + (jj % 3) * 4 : (L.i - 24) * 16 + 12) + 4 * L.g * 128;
    L.epiR = L.sig + (L.lane >> 2) * 128 + (L.lane & 3) * 16;
    L.rowBytes = (unsigned)a.num * 4u;
    L.planeBytes = (unsigned)a.timeLength * (unsigned)a.num * 4u;
    {
        const float cm = a.colMul[L.i];
#pragma unroll
        for (int level = 0; level < 7; ++level) {
            const int col = 12 * (6 - level) + (L.i < 24 ? jj : 0);
            L.mul[level] = cm * (a.octScale[level] / a.scale[col]);
        }
    }
    const unsigned OOR = 0x80000000u;
    const int totalBlocks = tilesPerClip * a.batch;
    const int stride = gridDim.x * waves;
    int blk = blockIdx.x * waves + wave;  // wave-uniform
    u32x4 w2[Lay<2>::NV], w4[Lay<4>::NV], w8[Lay<8>::NV], w16[Lay<16>::NV], w32[Lay<32>::NV], w64[Lay<64>::NV],
        w128[Lay<128>::NV];
    if (blk < totalBlocks) {
        const int clip = blk / tilesPerClip;
        fetch_window<2>(a, L, clip, (blk - clip * tilesPerClip) * 32, w2);
    }
    {   // as many dropped stores as a block's last tile (+ its chroma rows) leaves in flight: see afx_cqt_f16.hip
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(a.outRe, 0, 4, RSRC_RAW);
#pragma unroll
        for (int r = 0; r < (CHROMA ? 7 : 4); ++r) __builtin_amdgcn_raw_buffer_store_b32(0u, rd, OOR + 4u * r, 0, 0);
    }
    for (; blk < totalBlocks; blk += stride) {
        const int clip = blk / tilesPerClip, t0 = (blk - clip * tilesPerClip) * 32;
        if (CHROMA) {
            wave_lds_order();  // (outside the lane condition: the host emulation of tests/emu makes it a rendezvous of the wave)
            if (L.lane < 32) {
                float4 *z = reinterpret_cast<float4 *>(L.acc + L.lane * 12);
                z[0] = z[1] = z[2] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        tile<2, CHROMA>(a, L, clip, t0, w2, [&] { fetch_window<4>(a, L, clip, t0, w4); });
        tile<4, CHROMA>(a, L, clip, t0, w4, [&] { fetch_window<8>(a, L, clip, t0, w8); });
        tile<8, CHROMA>(a, L, clip, t0, w8, [&] { fetch_window<16>(a, L, clip, t0, w16); });
        tile<16, CHROMA>(a, L, clip, t0, w16, [&] { fetch_window<32>(a, L, clip, t0, w32); });
        tile<32, CHROMA>(a, L, clip, t0, w32, [&] { fetch_window<64>(a, L, clip, t0, w64); });
        tile<64, CHROMA>(a, L, clip, t0, w64, [&] { fetch_window<128>(a, L, clip, t0, w128); });
        tile<128, CHROMA>(a, L, clip, t0, w128, [&] {
            const int nb = blk + stride;
            if (nb < totalBlocks) {
                const int nc = nb / tilesPerClip;
                fetch_window<2>(a, L, nc, (nb - nc * tilesPerClip) * 32, w2);
            }
        });
        if (CHROMA) {
            // per-frame normalisation (__mnormalize, k_cqt_chroma's order) and the frame's 12 values as three
            // 16-byte stores; rows past timeLength fall out of the clip's chroma buffer and are dropped
            wave_lds_order();
            const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(
                a.chroma + (long long)clip * a.chromaStride, 0, a.timeLength * 12 * 4, RSRC_RAW);
            float v[12];
            const float *c = L.acc + (L.lane & 31) * 12;
#pragma unroll
            for (int k = 0; k < 12; ++k) v[k] = c[k];
            if (a.normType != 0) {
                float red = a.normType == 2 ? 3.4e38f : 0.f;
#pragma unroll
                for (int k = 0; k < 12; ++k) {
                    const float av = fabsf(v[k]);
                    if (a.normType == 1) red = fmaxf(red, av);
                    else if (a.normType == 2) red = fminf(red, av);
                    else if (a.normType == 3) red += av * av;
                    else red += av;
                }
                if (a.normType == 3) red = sqrtf(red);
                if (red != 0.f) {
#pragma unroll
                    for (int k = 0; k < 12; ++k) v[k] = v[k] / red;
                }
            }
            const unsigned vo = L.lane < 32 ? (unsigned)(t0 + L.lane) * 48u : OOR;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const u32x4 o4 = {__float_as_uint(v[4 * q]), __float_as_uint(v[4 * q + 1]), __float_as_uint(v[4 * q + 2]),
                                  __float_as_uint(v[4 * q + 3])};
                __builtin_amdgcn_raw_buffer_store_b128(o4, rc, vo + 16u * q, 0, 0);
            }
        }
    }
}

template <bool CHROMA>
int launch_all(const AfxCqtAllArgs *a, void *stream) {
    const int waves = 4;
    const size_t lds = (size_t)B_BYTES + (size_t)waves * WAVE_BYTES;
    const void *fn = reinterpret_cast<const void *>(k_cqt_all_f16<CHROMA>);
    AFX_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int tilesPerClip = (a->timeLength + 31) / 32;
    const long long total = (long long)tilesPerClip * a->batch;
    if (total > 0x7fffffffLL) return AFX_ERR_UNSUPPORTED;
    long long wgs = (total + waves - 1) / waves;
    if (wgs > 256) wgs = 256;
    hipLaunchKernelGGL((k_cqt_all_f16<CHROMA>), dim3((unsigned)wgs), dim3(64 * waves), lds, (hipStream_t)stream, *a,
                       tilesPerClip);
    AFX_LAUNCH_CHECK("k_cqt_all_f16");
    return AFX_OK;
}

}  // namespace

// Seven octaves of 12 bins, N = 512, hop ladder 128 ... 2, one shared image; a->chroma != NULL adds the chroma
// (12 classes).  AFX_ERR_UNSUPPORTED for anything else -- the caller keeps the per-octave launches.
extern "C" int afxk_cqt_all_f16(const AfxCqtAllArgs *a, void *stream) {
    if (!a->imageH || !a->colMul || a->num != 84 || a->batch <= 0 || a->timeLength <= 0) return AFX_ERR_UNSUPPORTED;
    for (int l = 0; l < 7; ++l)
        if (!a->x[l] || a->validLength[l] > (1 << 28) || a->validLength[l] < 0) return AFX_ERR_UNSUPPORTED;
    if ((long long)a->timeLength * a->num > (1LL << 28)) return AFX_ERR_UNSUPPORTED;
    return a->chroma ? launch_all<true>(a, stream) : launch_all<false>(a, stream);
}
