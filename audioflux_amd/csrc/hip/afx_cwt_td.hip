// afx_cwt_td.hip -- the WIDE scales of the continuous wavelet transform in the time domain, on the f16 matrix cores.
//
// The reference (__cwtObj_cwt, src/cwt_algorithm.c:361-483) forms W_j = IFFT(psi_j . FFT(xp)) for every scale j with
// one transform of length L (xp = the chunk, reflect padded).  That is the circular convolution
//     W_j[n] = sum_t g_j[t] xp[pad + n - t],          g_j = IFFT(psi_j)                       (the same linear map)
// and a wavelet that is WIDE in frequency is SHORT in time: for morlet at BASELINE cfg 4 the 40 scales whose support
// spans 17..157 rows of the transposed spectrum -- the ones that take both four-step passes and write + re-read a
// 1 MB intermediate per scale and chunk (82 of the 141 MB of traffic per chunk, 70 % of the step's time in round 2)
// -- have |g_j| < 1e-7 max|g_j| beyond 60..580 samples.  The host plan (afx_cwt.c: cwt_td_plan) evaluates g_j in
// double from the bank's own float32 row, truncates it there (truncation error 4e-8 of the peak, tools/proto_cwt_td.py)
// and hands pairs of scales to this kernel, which never forms an intermediate: HBM traffic per scale and chunk =
// the 0.5 MB of output (the input chunk is read from L2).
//
// Formulation = the CQT octave product (afx_cqt_f16.hip) at hop 8: rows of the A operand are Toeplitz views of the
// signal, 8 samples apart; the 8 output phases inside a row are 8 shifted copies of the kernel among the B columns:
//     y[n0 + 8 i + p] = sum_m win[8 i + m] h_p[m],    win[k] = xp[n0 - kh + k],   h_p[m] = g[p + kh - m]
//   * 32 columns = 2 scales x (re, im) x 8 phases, 32 rows = 256 consecutive outputs per MFMA tile;
//   * both operands as (hi, lo) binary16 words under power-of-two scaling (afx_f16split.h), three products
//     xh gh + xh gl + xl gh on v_mfma_f32_32x32x16_f16 with float32 accumulation: measured against the reference
//     (tests) 3-6e-7 of the peak -- and 1e-6 of every 512-sample block's OWN peak after a level step, where the
//     reference's float32 transform of 2^17 points carries 4e-4 (tools/proto_cwt_td.py);
//   * a persistent workgroup keeps ONE pair's image (2 KB per K step of 16 taps, <= 128 KB) in LDS and walks tiles of
//     512 outputs (two row tiles share every B fragment): per K step 6 MFMAs on 4 accumulators, 6 ds_read_b128;
//   * the window comes in by 16-byte loads (chunk edges: per-sample loads through the reflect / wrap index map,
//     cwt_algorithm.c:404-414), is scaled by 2^e from its own peak, split and stored as two f16 planes; results are
//     transposed through LDS and leave as 1 KB runs per (scale, plane).
#include <hip/hip_runtime.h>

#include <atomic>

#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "afx_device.h"
#include "afx_hipcheck.h"
#include "afx_pkmath.h"
#include "afx_f16split.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int RSRC_RAW = 0x00020000;  // raw buffer, 32-bit data format (cdna_hip_programming.md T8)
constexpr unsigned OOR = 0x80000000u; // an offset the bounds check rejects: loads return 0, stores are dropped

constexpr int WAVES = 4;             // one wave per SIMD: the image leaves room for no more at the longest kernels
constexpr int TILE = 512;            // outputs per wave iteration: two row tiles of 32 rows x 8 phases
constexpr int SLAB = 8192;           // outputs per work unit (a workgroup's share of one chunk at a time)
constexpr int EPI_PITCH = 264;       // floats per (scale, plane) row of the epilogue buffer: banks 8 q + p distinct
// Two classes of pairs, one instantiation each: MAXK = the taps of the longest kernel of the class.  The long class
// (<= 1024 taps: up to 128 KB of image) leaves room for one workgroup per CU; the short class (<= SHORTK taps: <= 48 KB
// of image, 3.7 KB of window planes per wave) fits TWO workgroups per CU, so that one wave's conversion / epilogue
// runs under the other's K loop -- with 8 .. 24 K steps per tile those phases are as long as the loop itself.
constexpr int SHORTK = 384;
template <int MAXK>
struct TdGeom {
    static constexpr int NV = (TILE + MAXK + 255) / 256;     // 16-byte window loads per lane
    static constexpr int PLANE = 2 * (TILE + MAXK) + 96;     // bytes of one f16 plane of the window (+ slack: the K
                                                             // loop requests two steps past the end, never used)
    static constexpr int WAVE_BYTES = (2 * PLANE > 4 * EPI_PITCH * 4 ? 2 * PLANE : 4 * EPI_PITCH * 4);  // hi | lo;
                                                             // re-used by the transposed epilogue (4224 B)
};

constexpr int MAXPAIRS = AFX_CWT_TD_MAXPAIRS;
struct TdArgs {
    const float *x;
    long long xStride;
    int chunks, dataLength, num, wrap;
    const AfxCwtTdPair *pairs;
    const unsigned char *image;
    int nPairs;                // pairs of this launch (a class of the plan: pairs[0 .. nPairs))
    float *outRe, *outIm;
    int wgBase[MAXPAIRS + 1];  // workgroups [wgBase[p], wgBase[p + 1]) serve pair p (per call: exact shares of its units)
};

// position q of the (conceptually padded) chunk -> sample index: reflect (cwt_algorithm.c:404-414) or wrap
__device__ __forceinline__ int td_index(int q, int D, int wrap) {
    if (q >= 0 && q < D) return q;
    if (wrap) return q < 0 ? q + D : q - D;  // (|kernel half length| < D: one turn)
    return q < 0 ? -1 - q : 2 * D - 1 - q;
}

template <int MAXK>
__global__ __launch_bounds__(WAVES * 64) void k_cwt_td(TdArgs a) {
    constexpr int NV = TdGeom<MAXK>::NV, PLANE = TdGeom<MAXK>::PLANE, WAVE_BYTES = TdGeom<MAXK>::WAVE_BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, g = lane >> 5;

    // ---- which pair this workgroup serves, and which of its work units (host: workgroups in proportion to K)
    int p = 0;
    while (p + 1 < a.nPairs && (int)blockIdx.x >= a.wgBase[p + 1]) ++p;
    const AfxCwtTdPair *pr = a.pairs + p;  // (fields are read one by one: a local copy of the struct lands in scratch)
    const int wgCount = a.wgBase[p + 1] - a.wgBase[p];
    const int local = (int)blockIdx.x - a.wgBase[p];
    const int KS = pr->ks, kh = pr->kh, Kt = 16 * KS;
    const int imgBytes = 2 * KS * 1024;
    unsigned char *Bl = smem_raw;
    unsigned char *sig = smem_raw + imgBytes + 2048 + wave * WAVE_BYTES;  // (+ 2 KB: the K loop requests two steps ahead)
    {   // image -> LDS
        const float4 *src = reinterpret_cast<const float4 *>(a.image + pr->img);
        float4 *dst = reinterpret_cast<float4 *>(Bl);
        for (int e = tid; e < imgBytes / 16; e += WAVES * 64) dst[e] = src[e];
    }
    __syncthreads();

    const int D = a.dataLength;
    const int slabs = D / SLAB;                       // D is a power of two >= SLAB
    const int units = a.chunks * slabs;
    const int itersPerUnit = SLAB / (TILE * WAVES);   // 4
    const int myUnits = local < units ? (units - local + wgCount - 1) / wgCount : 0;
    const int tiles = myUnits * itersPerUnit;
    const int nv4 = (TILE + Kt) / 4;                  // 16-byte vectors of one window

    // tile tau of this wave -> (chunk, first output n0)
    auto where = [&](int tau, int &chunk, int &n0) {
        const int u = local + (tau / itersPerUnit) * wgCount;
        chunk = u / slabs;
        n0 = (u - chunk * slabs) * SLAB + ((tau % itersPerUnit) * WAVES + wave) * TILE;
    };
    // Window loads.  Interior tiles (the whole window inside the chunk: all but the first and last few of a chunk):
    // NV unconditional 16-byte buffer loads per lane -- a fixed count, so that the wait for the window at the top of
    // the next tile is vmcnt(<the 8 stores issued behind them>) and not a wait for those stores (the same device as in
    // afx_cqt_f16.hip); lanes past the window read out of range (zeros).  Edge tiles (wave-uniform branch): per-sample
    // loads through the reflect / wrap index map.
    u32x4 wnd[NV];
    auto fetch = [&](int tau) {
        int chunk, n0;
        where(tau, chunk, n0);
        const float *xc = a.x + (long long)chunk * a.xStride;
        const int q00 = n0 - kh;
        if (q00 >= 0 && q00 + TILE + Kt <= D) {
            const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(xc), 0, D * 4, RSRC_RAW);
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const int v = lane + 64 * u;
                wnd[u] = __builtin_amdgcn_raw_buffer_load_b128(rx, v < nv4 ? (unsigned)(q00 + 4 * v) * 4u : OOR, 0, 0);
            }
        } else {
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const int v = lane + 64 * u;
                const int q0 = q00 + 4 * v;
                float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
                if (v < nv4) {
                    r.x = xc[td_index(q0, D, a.wrap)];
                    r.y = xc[td_index(q0 + 1, D, a.wrap)];
                    r.z = xc[td_index(q0 + 2, D, a.wrap)];
                    r.w = xc[td_index(q0 + 3, D, a.wrap)];
                }
                wnd[u] = __builtin_bit_cast(u32x4, r);
            }
        }
    };

    const float colMul = pr->colMul[i], colSum = pr->colSum[i];
    // A fragments: row i of row tile rt, half g, K step ks = the 16 bytes at 16 (32 rt + i + g + 2 ks)
    const unsigned char *aHi0 = sig + 16 * (i + g);
    const unsigned char *bHi0 = Bl + 16 * lane;
    const int bLoOff = KS * 1024;
    // epilogue: column c = lane & 31 -> q = c >> 3 in (scale, plane) order, phase c & 7
    float *epi = reinterpret_cast<float *>(sig);
    float *epiW = epi + (i >> 3) * EPI_PITCH + (i & 7) + 32 * g;
    const int scaleA = pr->scale[0], scaleB = pr->scale[1];

    if (tiles > 0) fetch(0);
    {   // eight out-of-range (dropped) stores behind the first window: the tile loop is then entered with the same
        // count of memory operations in flight as on its back edge (window loads, then a tile's eight stores), and the
        // compiler's wait for the window becomes vmcnt(8 + ...) instead of a wait for the previous tile's stores
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(a.outRe, 0, 4, RSRC_RAW);
        const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int r = 0; r < 8; ++r) __builtin_amdgcn_raw_buffer_store_b128(z, rd, OOR + 16u * r, 0, 0);  // distinct: not merged
    }
    for (int tau = 0; tau < tiles; ++tau) {
        int chunk, n0;
        where(tau, chunk, n0);
        // ---- tile exponent from the window's own peak.  A window that sits on a constant (all samples within a quarter
        //      of the extreme one, m) is split as x - m: the (hi, lo) words carry 22 bits of every SAMPLE, so under an
        //      offset they would round the offset, not the signal, and a kernel that sums to ~0 cancels the offset but
        //      not its rounding (measured: rows 2.8e-5 of their peak from float64 on a clip at 0.5 +- 0.05, the reference
        //      4e-6).  x - m is exact (x / m in [3/4, 5/4]); the constant's own response m sum_k g[k] comes back in the
        //      epilogue from the kernel's bin 0 (AfxCwtTdPair.colSum).  m is then moved to the middle of
        //      the samples, which halves what is left to split.
        float ppos = 0.f, pneg = 0.f;
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const float4 w4 = __builtin_bit_cast(float4, wnd[u]);
            ppos = fmaxf(ppos, fmaxf(fmaxf(w4.x, w4.y), fmaxf(w4.z, w4.w)));
            pneg = fmaxf(pneg, fmaxf(fmaxf(-w4.x, -w4.y), fmaxf(-w4.z, -w4.w)));
        }
        const unsigned bpos = wave_max_bits(ppos), bneg = wave_max_bits(pneg);
        unsigned peakBits = bpos > bneg ? bpos : bneg;
        float base = 0.f;
        if ((bpos == 0u || bneg == 0u) && peakBits != 0u && peakBits < 0x7f800000u) {  // one sign throughout (wave-uniform)
            const float m = bneg == 0u ? __uint_as_float(bpos) : -__uint_as_float(bneg);
            float dev = 0.f;
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const float4 w4 = __builtin_bit_cast(float4, wnd[u]);
                if (lane + 64 * u < nv4)  // (registers beyond the window hold zeros)
                    dev = fmaxf(dev, fmaxf(fmaxf(fabsf(w4.x - m), fabsf(w4.y - m)), fmaxf(fabsf(w4.z - m), fabsf(w4.w - m))));
            }
            const float range = __uint_as_float(wave_max_bits(dev));  // extreme sample to the opposite extreme
            if (range <= 0.25f * fabsf(m)) {
                base = m - copysignf(0.5f * range, m);  // the middle of the samples: every x / base in [3/4, 8/7], x - base exact
                peakBits = __float_as_uint(0.5f * range + fabsf(m) * 0x1p-22f);  // >= max |x - base| (the middle is rounded)
            }
        }
        const int e = split_exponent(peakBits);
        const float up = __uint_as_float((unsigned)(e + 127) << 23);      // 2^e
        const float down = __uint_as_float((unsigned)(127 - e) << 23);    // 2^-e
        // ---- window -> (xh, xl) planes
        wave_lds_order();  // the previous tile's epilogue reads are done
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int v = lane + 64 * u;
            if (v < nv4) {
                const float4 w4 = __builtin_bit_cast(float4, wnd[u]);
                unsigned hi0, hi1, lo0, lo1;
                split_pair(w4.x - base, w4.y - base, up, hi0, lo0);
                split_pair(w4.z - base, w4.w - base, up, hi1, lo1);
                *reinterpret_cast<uint2 *>(sig + 8 * v) = make_uint2(hi0, hi1);
                *reinterpret_cast<uint2 *>(sig + PLANE + 8 * v) = make_uint2(lo0, lo1);
            }
        }
        wave_lds_order();
        if (tau + 1 < tiles) fetch(tau + 1);

        // ---- K loop: per step of 16 taps 6 MFMAs (two row tiles x {xh gh, xh gl, xl gh}), operands two steps ahead.
        //      The first step takes a zero C operand (an inline constant): no pass over the 64 accumulator registers.
        f32x16 hh0, hh1, x0, x1;
        const f32x16 Z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        h8 ah0[4], al0[4], ah1[4], al1[4], bh[4], bl[4];
        const unsigned char *pa = aHi0, *pb = bHi0;
        auto load = [&](int slot, int off) {  // off: K step relative to the current base (compile-time immediates)
            ah0[slot] = *reinterpret_cast<const h8 *>(pa + 32 * off);
            al0[slot] = *reinterpret_cast<const h8 *>(pa + PLANE + 32 * off);
            ah1[slot] = *reinterpret_cast<const h8 *>(pa + 512 + 32 * off);
            al1[slot] = *reinterpret_cast<const h8 *>(pa + PLANE + 512 + 32 * off);
            bh[slot] = *reinterpret_cast<const h8 *>(pb + 1024 * off);
            bl[slot] = *reinterpret_cast<const h8 *>(pb + bLoOff + 1024 * off);
        };
        auto block = [&](auto first) {  // four K steps; first: the accumulators start here
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                __builtin_amdgcn_sched_barrier(0);
                load((s + 2) & 3, s + 2);  // (the last block requests two steps past the end: slack above, never used)
                const bool init = decltype(first)::value && s == 0;
                hh0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0[s], bh[s], init ? Z : hh0, 0, 0, 0);
                hh1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1[s], bh[s], init ? Z : hh1, 0, 0, 0);
                x0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0[s], bl[s], init ? Z : x0, 0, 0, 0);
                x1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1[s], bl[s], init ? Z : x1, 0, 0, 0);
                x0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al0[s], bh[s], x0, 0, 0, 0);
                x1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al1[s], bh[s], x1, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read
                }
            }
            pa += 128;
            pb += 4096;
        };
        load(0, 0);
        load(1, 1);
        block(std::true_type{});
        for (int kb = 4; kb + 4 <= KS; kb += 4) block(std::false_type{});  // KS is even and >= 4 (host)
        if (KS & 2) {  // taps are rounded to 32: a last half block, its operands were requested by the block before
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                __builtin_amdgcn_sched_barrier(0);
                hh0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0[s], bh[s], hh0, 0, 0, 0);
                hh1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1[s], bh[s], hh1, 0, 0, 0);
                x0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0[s], bl[s], x0, 0, 0, 0);
                x1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1[s], bl[s], x1, 0, 0, 0);
                x0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al0[s], bh[s], x0, 0, 0, 0);
                x1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al1[s], bh[s], x1, 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);

        // ---- epilogue: D layout col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 g -> output n0 + 256 rt + 8 row + phase
        const float mul = down * colMul;
        const float back = base * colSum;  // the response to the constant that was taken out of the window
        // one raw buffer per plane and chunk ([num][D] floats); a single scale in the pair: its second half is dropped
        const __amdgpu_buffer_rsrc_t rRe = __builtin_amdgcn_make_buffer_rsrc(a.outRe + (long long)chunk * a.num * D, 0, a.num * D * 4, RSRC_RAW);
        const __amdgpu_buffer_rsrc_t rIm = __builtin_amdgcn_make_buffer_rsrc(a.outIm + (long long)chunk * a.num * D, 0, a.num * D * 4, RSRC_RAW);
        const unsigned offA = (unsigned)(scaleA * D + n0 + 4 * lane) * 4u;
        const unsigned offB = scaleB >= 0 ? (unsigned)(scaleB * D + n0 + 4 * lane) * 4u : OOR;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            wave_lds_order();  // fragment reads (rt 0) / the previous row tile's epilogue reads are done
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = rt == 0 ? (hh0[r] + x0[r]) : (hh1[r] + x1[r]);
                epiW[8 * ((r & 3) + 8 * (r >> 2))] = fmaf(v, mul, back);
            }
            wave_lds_order();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const u32x4 v = *reinterpret_cast<const u32x4 *>(epi + q * EPI_PITCH + 4 * lane);
                const unsigned off = (q >= 2 ? offB : offA) + (scaleB < 0 && q >= 2 ? 0u : 1024u * rt);
                __builtin_amdgcn_raw_buffer_store_b128(v, (q & 1) ? rIm : rRe, off, 0, 0);
            }
        }
    }
}

}  // namespace

template <int MAXK>
static int launch_td(const AfxCwtTdPlan *p, int first, int count, int maxKs, double wgTarget, TdArgs a, void *stream) {
    if (count <= 0) return AFX_OK;
    const size_t lds = (size_t)2 * maxKs * 1024 + 2048 + (size_t)WAVES * TdGeom<MAXK>::WAVE_BYTES;
    if (lds > 160 * 1024) return AFX_ERR_UNSUPPORTED;
    static std::atomic<bool> attrSet[AFX_MAX_DEVICES];
    const int dev = afxdev_current_device() & (AFX_MAX_DEVICES - 1);
    if (!attrSet[dev].load(std::memory_order_acquire)) {  // (two threads may both set it: idempotent)
        AFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_cwt_td<MAXK>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024));
        attrSet[dev].store(true, std::memory_order_release);
    }
    a.pairs = p->pairs + first;
    a.nPairs = count;
    // Workgroups per pair in proportion to its K steps (every workgroup then issues about the same number of matrix
    // instructions) and EXACT shares: every workgroup of a pair takes the same number of (chunk, slab) units
    const long long units = (long long)a.chunks * (a.dataLength / SLAB);
    long long ksSum = 0;
    for (int q = 0; q < count; ++q) ksSum += p->hostKs[first + q];
    int base = 0;
    for (int q = 0; q < count; ++q) {
        double ideal = wgTarget * (double)p->hostKs[first + q] / (double)ksSum;
        if (ideal < 1.0) ideal = 1.0;
        long long per = (long long)((double)units / ideal + 0.999);  // units per workgroup
        if (per < 1) per = 1;
        a.wgBase[q] = base;
        base += (int)((units + per - 1) / per);
    }
    a.wgBase[count] = base;
    hipLaunchKernelGGL(k_cwt_td<MAXK>, dim3((unsigned)base), dim3(WAVES * 64), lds, (hipStream_t)stream, a);
    AFX_LAUNCH_CHECK("k_cwt_td");
    return AFX_OK;
}

// every precondition of the two launches below (the host plans with it: afx_cwt.c: cwt_td_plan)
extern "C" int afxk_cwt_td_fits(const AfxCwtTdPlan *p, int dataLength, int num) {
    if (!p || p->nPairs <= 0 || !p->hostKs) return AFX_ERR_UNSUPPORTED;
    if (dataLength < SLAB || (dataLength & (dataLength - 1))) return AFX_ERR_UNSUPPORTED;
    if (p->maxKs < 4 || 16 * p->maxKs > AFX_CWT_TD_MAXK) return AFX_ERR_UNSUPPORTED;
    for (int q = 0; q < p->nPairs; ++q)
        if (p->hostKs[q] < 4 || (p->hostKs[q] & 1)) return AFX_ERR_UNSUPPORTED;  // whole blocks of four K steps + at most one half block
    if ((long long)num * dataLength * 4 > 0x7fffffffLL) return AFX_ERR_UNSUPPORTED;  // 32-bit offsets inside one chunk's planes
    int nLong = 0;  // pairs are sorted longest first; each class is one launch of <= MAXPAIRS pairs
    while (nLong < p->nPairs && 16 * p->hostKs[nLong] > SHORTK) ++nLong;
    if (nLong > MAXPAIRS || p->nPairs - nLong > MAXPAIRS) return AFX_ERR_UNSUPPORTED;
    if (nLong > 0 && (size_t)2 * p->maxKs * 1024 + 2048 + (size_t)WAVES * TdGeom<AFX_CWT_TD_MAXK>::WAVE_BYTES > 160 * 1024) return AFX_ERR_UNSUPPORTED;
    if (nLong < p->nPairs && (size_t)2 * p->hostKs[nLong] * 1024 + 2048 + (size_t)WAVES * TdGeom<SHORTK>::WAVE_BYTES > 160 * 1024)
        return AFX_ERR_UNSUPPORTED;
    return AFX_OK;
}

// streamShort (NULL: `stream`): where the short-kernel class is launched -- beside the long class when it is another
// stream (two workgroups of it fit a CU, so it fills the long class's tail); the caller joins the two
extern "C" int afxk_cwt_td(const AfxCwtTdPlan *p, const float *x, long long xStride, int chunks, int dataLength, int num,
                           float *outRe, float *outIm, void *stream, void *streamShort) {
    if (!p || p->nPairs <= 0 || chunks <= 0) return AFX_OK;
    if (afxk_cwt_td_fits(p, dataLength, num) != AFX_OK) return AFX_ERR_UNSUPPORTED;
    TdArgs a;
    a.x = x;
    a.xStride = xStride;
    a.chunks = chunks;
    a.dataLength = dataLength;
    a.num = num;
    a.wrap = p->wrap;
    a.image = p->image;
    a.outRe = outRe;
    a.outIm = outIm;
    int nLong = 0;  // pairs are sorted longest first
    while (nLong < p->nPairs && 16 * p->hostKs[nLong] > SHORTK) ++nLong;
    long long ksLong = 0, ksAll = 0;
    for (int q = 0; q < p->nPairs; ++q) {
        ksAll += p->hostKs[q];
        if (q < nLong) ksLong += p->hostKs[q];
    }
    // ~4 rounds of the 256 CUs in all, split between the classes by their matrix work
    const double wgAll = 1024.0;
    int st = launch_td<AFX_CWT_TD_MAXK>(p, 0, nLong, p->maxKs, wgAll * (double)ksLong / (double)ksAll, a, stream);
    if (st == AFX_OK && nLong < p->nPairs)
        st = launch_td<SHORTK>(p, nLong, p->nPairs - nLong, p->hostKs[nLong], wgAll * (double)(ksAll - ksLong) / (double)ksAll + 256.0,
                               a, streamShort ? streamShort : stream);
    return st;
}
