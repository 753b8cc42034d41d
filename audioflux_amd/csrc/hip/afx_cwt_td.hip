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

#include <cstdint>
#include <cstdlib>

#include "afx_device.h"
#include "afx_hipcheck.h"
#include "afx_pkmath.h"
#include "afx_f16split.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

constexpr int WAVES = 4;             // one wave per SIMD: the image leaves room for no more at the longest kernels
constexpr int TILE = 512;            // outputs per wave iteration: two row tiles of 32 rows x 8 phases
constexpr int SLAB = 8192;           // outputs per work unit (a workgroup's share of one chunk at a time)
constexpr int NV = (TILE + AFX_CWT_TD_MAXK) / 256;   // 16-byte window loads per lane (6)
constexpr int PLANE = 2 * (TILE + AFX_CWT_TD_MAXK) + 64;  // bytes of one f16 plane of the window (+ slack: the K loop
                                                          // requests one step past the end, never used)
constexpr int WAVE_BYTES = 2 * PLANE;  // hi | lo; re-used by the transposed epilogue (4 x 264 floats = 4224 B)
constexpr int EPI_PITCH = 264;       // floats per (scale, plane) row of the epilogue buffer: banks 8 q + p distinct
static_assert(4 * EPI_PITCH * 4 <= WAVE_BYTES, "epilogue buffer must fit the window region");

struct TdArgs {
    const float *x;
    long long xStride;
    int chunks, dataLength, num, wrap, aligned;
    const AfxCwtTdPair *pairs;
    const unsigned char *image;
    int nPairs;
    float *outRe, *outIm;
};

// position q of the (conceptually padded) chunk -> sample index: reflect (cwt_algorithm.c:404-414) or wrap
__device__ __forceinline__ int td_index(int q, int D, int wrap) {
    if (q >= 0 && q < D) return q;
    if (wrap) {
        q %= D;
        return q < 0 ? q + D : q;
    }
    return q < 0 ? -1 - q : 2 * D - 1 - q;
}

__global__ __launch_bounds__(WAVES * 64) void k_cwt_td(TdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, g = lane >> 5;

    // ---- which pair this workgroup serves, and which of its work units (host: workgroups in proportion to K)
    int p = 0;
    while (p + 1 < a.nPairs && (int)blockIdx.x >= a.pairs[p].wgBase + a.pairs[p].wgCount) ++p;
    const AfxCwtTdPair *pr = a.pairs + p;  // (fields are read one by one: a local copy of the struct lands in scratch)
    const int wgCount = pr->wgCount;
    const int local = (int)blockIdx.x - pr->wgBase;
    const int KS = pr->ks, kh = pr->kh, Kt = 16 * KS;
    const int imgBytes = 2 * KS * 1024;
    unsigned char *Bl = smem_raw;
    unsigned char *sig = smem_raw + imgBytes + 1024 + wave * WAVE_BYTES;  // (+ 1 KB: the K loop reads one step ahead)
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
    float4 wnd[NV];
    auto fetch = [&](int tau) {
        int chunk, n0;
        where(tau, chunk, n0);
        const float *xc = a.x + (long long)chunk * a.xStride;
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int v = lane + 64 * u;
            const int q0 = n0 - kh + 4 * v;
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
            if (v < nv4) {
                if (a.aligned && q0 >= 0 && q0 + 3 < D) {
                    r = *reinterpret_cast<const float4 *>(xc + q0);
                } else {
                    r.x = xc[td_index(q0, D, a.wrap)];
                    r.y = xc[td_index(q0 + 1, D, a.wrap)];
                    r.z = xc[td_index(q0 + 2, D, a.wrap)];
                    r.w = xc[td_index(q0 + 3, D, a.wrap)];
                }
            }
            wnd[u] = r;
        }
    };

    const float colMul = pr->colMul[i];
    // A fragments: row i of row tile rt, half g, K step ks = the 16 bytes at 16 (32 rt + i + g + 2 ks)
    const unsigned char *aHi0 = sig + 16 * (i + g);
    const unsigned char *bHi0 = Bl + 16 * lane;
    const int bLoOff = KS * 1024;
    // epilogue: column c = lane & 31 -> q = c >> 3 in (scale, plane) order, phase c & 7
    float *epi = reinterpret_cast<float *>(sig);
    float *epiW = epi + (i >> 3) * EPI_PITCH + (i & 7) + 32 * g;
    const int scaleA = pr->scale[0], scaleB = pr->scale[1];

    if (tiles > 0) fetch(0);
    for (int tau = 0; tau < tiles; ++tau) {
        int chunk, n0;
        where(tau, chunk, n0);
        // ---- tile exponent from the window's own peak
        float peak = 0.f;
#pragma unroll
        for (int u = 0; u < NV; ++u)
            peak = fmaxf(peak, fmaxf(fmaxf(fabsf(wnd[u].x), fabsf(wnd[u].y)), fmaxf(fabsf(wnd[u].z), fabsf(wnd[u].w))));
        const int e = split_exponent(wave_max_bits(peak));
        const float up = __uint_as_float((unsigned)(e + 127) << 23);      // 2^e
        const float down = __uint_as_float((unsigned)(127 - e) << 23);    // 2^-e
        // ---- window -> (xh, xl) planes
        wave_lds_order();  // the previous tile's epilogue reads are done
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int v = lane + 64 * u;
            if (v < nv4) {
                unsigned hi0, hi1, lo0, lo1;
                split_pair(wnd[u].x, wnd[u].y, up, hi0, lo0);
                split_pair(wnd[u].z, wnd[u].w, up, hi1, lo1);
                *reinterpret_cast<uint2 *>(sig + 8 * v) = make_uint2(hi0, hi1);
                *reinterpret_cast<uint2 *>(sig + PLANE + 8 * v) = make_uint2(lo0, lo1);
            }
        }
        wave_lds_order();
        if (tau + 1 < tiles) fetch(tau + 1);

        // ---- K loop: per step of 16 taps 6 MFMAs (two row tiles x {xh gh, xh gl, xl gh}), operands one step ahead
        f32x16 hh0, hh1, x0, x1;
#pragma unroll
        for (int r = 0; r < 16; ++r) hh0[r] = hh1[r] = x0[r] = x1[r] = 0.f;
        h8 ah0[2], al0[2], ah1[2], al1[2], bh[2], bl[2];
        const unsigned char *pa = aHi0, *pb = bHi0;
        auto load = [&](int slot, int off) {  // off: K step relative to the current base (compile-time immediates)
            ah0[slot] = *reinterpret_cast<const h8 *>(pa + 32 * off);
            al0[slot] = *reinterpret_cast<const h8 *>(pa + PLANE + 32 * off);
            ah1[slot] = *reinterpret_cast<const h8 *>(pa + 512 + 32 * off);
            al1[slot] = *reinterpret_cast<const h8 *>(pa + PLANE + 512 + 32 * off);
            bh[slot] = *reinterpret_cast<const h8 *>(pb + 1024 * off);
            bl[slot] = *reinterpret_cast<const h8 *>(pb + bLoOff + 1024 * off);
        };
        load(0, 0);
        for (int kb = 0; kb < KS; kb += 4) {  // KS is a multiple of 4 (host)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                __builtin_amdgcn_sched_barrier(0);
                load((s + 1) & 1, s + 1);  // (the last step of the last block reads one step past the end: slack above)
                const int sl = s & 1;
                hh0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0[sl], bh[sl], hh0, 0, 0, 0);
                hh1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1[sl], bh[sl], hh1, 0, 0, 0);
                x0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0[sl], bl[sl], x0, 0, 0, 0);
                x1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1[sl], bl[sl], x1, 0, 0, 0);
                x0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al0[sl], bh[sl], x0, 0, 0, 0);
                x1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al1[sl], bh[sl], x1, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read
                }
            }
            pa += 128;
            pb += 4096;
        }
        __builtin_amdgcn_sched_barrier(0);

        // ---- epilogue: D layout col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 g -> output n0 + 256 rt + 8 row + phase
        const float mul = down * colMul;
        const long long planeA = ((long long)chunk * a.num + scaleA) * D + n0;
        const long long planeB = ((long long)chunk * a.num + (scaleB >= 0 ? scaleB : scaleA)) * D + n0;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            wave_lds_order();  // fragment reads (rt 0) / the previous row tile's epilogue reads are done
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = rt == 0 ? (hh0[r] + x0[r]) : (hh1[r] + x1[r]);
                epiW[8 * ((r & 3) + 8 * (r >> 2))] = v * mul;
            }
            wave_lds_order();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q >= 2 && scaleB < 0) continue;  // an odd scale count: the last pair's second half is empty
                const float4 v = *reinterpret_cast<const float4 *>(epi + q * EPI_PITCH + 4 * lane);
                float *dst = ((q & 1) ? a.outIm : a.outRe) + (q >= 2 ? planeB : planeA) + 256 * rt + 4 * lane;
                if (a.aligned) {
                    *reinterpret_cast<float4 *>(dst) = v;
                } else {
                    dst[0] = v.x, dst[1] = v.y, dst[2] = v.z, dst[3] = v.w;
                }
            }
        }
    }
}

}  // namespace

extern "C" int afxk_cwt_td(const AfxCwtTdPlan *p, const float *x, long long xStride, int chunks, int dataLength, int num,
                           float *outRe, float *outIm, void *stream) {
    if (!p || p->nPairs <= 0 || chunks <= 0) return AFX_OK;
    if (dataLength < SLAB || (dataLength & (dataLength - 1))) return AFX_ERR_UNSUPPORTED;
    if (p->maxKs < 4 || 16 * p->maxKs > AFX_CWT_TD_MAXK || p->wgTotal <= 0) return AFX_ERR_UNSUPPORTED;
    const size_t lds = (size_t)2 * p->maxKs * 1024 + 1024 + (size_t)WAVES * WAVE_BYTES;
    if (lds > 160 * 1024) return AFX_ERR_UNSUPPORTED;
    static bool attrSet[AFX_MAX_DEVICES] = {};
    const int dev = afxdev_current_device() & (AFX_MAX_DEVICES - 1);
    if (!attrSet[dev]) {
        AFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_cwt_td), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024));
        attrSet[dev] = true;
    }
    TdArgs a;
    a.x = x;
    a.xStride = xStride;
    a.chunks = chunks;
    a.dataLength = dataLength;
    a.num = num;
    a.wrap = p->wrap;
    a.aligned = (reinterpret_cast<uintptr_t>(x) % 16 == 0) && (xStride % 4 == 0 || chunks == 1) &&
                (reinterpret_cast<uintptr_t>(outRe) % 16 == 0) && (reinterpret_cast<uintptr_t>(outIm) % 16 == 0);
    a.pairs = p->pairs;
    a.image = p->image;
    a.nPairs = p->nPairs;
    a.outRe = outRe;
    a.outIm = outIm;
    hipLaunchKernelGGL(k_cwt_td, dim3((unsigned)p->wgTotal), dim3(WAVES * 64), lds, (hipStream_t)stream, a);
    AFX_LAUNCH_CHECK("k_cwt_td");
    return AFX_OK;
}
