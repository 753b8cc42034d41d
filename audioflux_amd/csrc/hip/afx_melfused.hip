// afx_melfused.hip -- the headline kernel: framed STFT -> |S|^2 (or |S|,
// |S|^2p) -> banded filter bank, one 64-lane wave per 2048-sample frame, with
// no HBM round trip between the stages ("K1+K2+K3" of SURVEY.md 2b fused).
//
// What it replaces in the reference (per frame): window multiply + N-point
// radix-2 FFT (src/stft_algorithm.c:696-715, src/dsp/fft_algorithm.c:450-519),
// crop to N/2+1 bins (src/vector/flux_complex.c:254-286), re^2+im^2 / sqrt / pow
// (flux_complex.c:469-503, src/bft_algorithm.c:489-504) and the filter-bank
// product (src/vector/flux_vector.c:55-86) -- the [T,N] complex scratch and the
// [T,F] power spectrum the reference materialises never exist here.
//
// Per frame, all inside one wave (no workgroup barriers; waves are independent):
//   1. 16 coalesced float2 loads per lane of the hop-overlapped frame, times the
//      window: z[n] = (x[2n] w[2n], x[2n+1] w[2n+1]), n = 64*n1 + lane
//   2. 1024-point complex FFT of z (1024 = 16 x 16 x 4): radix-16 DFT in registers
//      over n1, twiddle W_1024^(lane*k1), transpose through LDS (pitch 68 float2:
//      conflict-free ds_read_b64), radix-16 DFT in registers, twiddle W_64, second
//      LDS image V[m2][q]
//   3. last radix-4 + real-input split, fused: lane takes the bases q = lane,
//      64+lane (and 128), reads V[0..3][q] and V[0..3][256-q], finishes both
//      radix-4 butterflies in registers -> Z[q+256j], Z[(256-q)+256j], and forms the
//      conjugate pairs (k, 1024-k): X[k] = E + W_2048^k O, X[1024-k] = conj(E - W O);
//      the spectrum value of both bins goes to the wave's power row in LDS
//      (complex arithmetic is written on float2 vectors with v_pk_{add,mul,fma}_f32
//      operand swizzles/negations, see pk_* helpers: 2 instructions per complex
//      multiply, 8 per radix-4 butterfly)
//   4. banded filter bank: lane i owns a long row A and a short row B of the bank
//      (AfxBandPlan); its weights stay in VGPRs for the life of the kernel, the
//      power row is read from LDS with immediate offsets, two bins per ds_read_b64:
//      (acc0, acc1) += (w[2t], w[2t+1]) * (P[s+2t], P[s+2t+1]), ascending bins
//   5. two dword stores per lane: out[frame, rowA], out[frame, rowB]
//
// HBM traffic per frame = 4*hop bytes in (each sample once; the 4x frame overlap
// is served by L1/L2) + 4*num bytes out.  Index algebra validated by
// tools/proto_fft1024.py.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>

#include "afx_device.h"
#include "afx_hipcheck.h"
#include "afx_pkmath.h"

#ifndef AFX_V
// Compile-time experiment switch for within-probe A/B runs (tools/variants.sh builds variants,
// tools/ab.py runs them interleaved on the GPU).  0 = shipped.  1: fence-based wave sync;
// 512: compiler-generated LDS reads instead of the hand-issued ones.  Earlier experiments and
// their timings (W_64 / W_1024 / window tables through L1, band-loop block sizes, operand
// batching, software pipelining across frames) are recorded in profiles/r01_ab_variants.txt.
#define AFX_V 0
#endif

namespace {

constexpr int NFFT = 2048;
constexpr int MC = 1024;        // complex FFT length
constexpr int EX_PITCH = 68;    // float2 per k1 row of the exchange image
constexpr int EX_F2 = 16 * EX_PITCH;  // 1088 float2; also holds the 1040-float2 natural image
constexpr int PROW_F = 1104;    // 1025 bins + zero pad for the fixed-length band loops (>= 1025 + 72)
constexpr int WAVE_LDS_BYTES = EX_F2 * 8;  // 8704; the power row (4416 B) aliases the exchange image
constexpr int WAVES = 12;       // one workgroup per CU: 3 waves per SIMD
// workgroup-shared constant tables staged in LDS once per workgroup
constexpr int TAB_TW1_F2 = 16 * 64;  // W_1024^(lane*k1)
constexpr int TAB_TW2_F2 = 64;       // W_64^(m2*j1)
constexpr int TAB_TW3_F2 = 1024;     // 0.5 * W_2048^k, k < 1024
constexpr int TAB_WIN_F2 = 1024;  // (w[2n], w[2n+1])
constexpr int TAB_F2 = TAB_WIN_F2 + TAB_TW1_F2 + TAB_TW2_F2 + TAB_TW3_F2;
constexpr int TAB_BYTES = TAB_F2 * 8;  // 20992
// band weights: one row of TA+TB floats per lane, row pitch TA+TB+4 floats (pitch/4 odd:
// the ds_read_b128 of any 16 consecutive lanes touch 64 distinct banks)
__host__ __device__ constexpr int wpitch(int ta, int tb) { return ta + tb + 4; }
__host__ __device__ constexpr int block_lds_bytes(int ta, int tb) {
    return TAB_BYTES + 64 * wpitch(ta, tb) * 4 + WAVES * WAVE_LDS_BYTES;
}

struct KArgs {
    const float *x;
    long long clipStride;
    long long totalFrames;
    int timeLength, hop;
    int framesPerWave;
    int aligned;  // frame starts are 8-byte aligned -> float2 loads
    const float2 *win2;  // [1024]  (w[2n], w[2n+1])
    const float2 *tw1;   // [16][64] W_1024^(lane*k1)
    const float2 *tw2;   // [4][16]  W_64^(m2*j1)
    const float2 *tw3;   // [1024]   0.5 * W_2048^k
    const float *wLane;    // [64][wpitch]: lane-major band weights, A taps then B taps
    const int *meta;       // [4][64]: startA, startB, rowA, rowB
    int specMap, postPow;
    float normValue;
    float *out;
    float *outIm;  // complex result mode only
    int num;
};

// ---- un-paired LDS reads ----------------------------------------------------------------
// hipcc's load/store optimizer fuses two float2 reads from one base into ds_read2_b64 /
// ds_read2st64_b64, which the LDS serves at 128 B/clk; two plain ds_read_b64 run at 256 B/clk
// (MI355X_MICROARCH.md LDS table; tools/micro/lds_read_rate.hip: 107 vs 68 TB/s aggregate at
// 12 waves per CU).  The wide read sites therefore issue ds_read_b64 themselves: RD64 requests
// (immediate offset, no wait), lds_wait() drains, PIN ties each value to the drained state so
// that no use is scheduled above the wait.  AFX_V & 512 restores the compiler's reads.
__device__ __forceinline__ unsigned lds_addr(const void *p) { return (unsigned)(size_t)p; }
#define RD64(dst, addr, off) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#define RD128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#define PIN(x) asm volatile("" : "+v"(x))
#define LDS_WAIT_N(n) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory")
__device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__device__ __forceinline__ void wave_lds_sync() {
    // Orders this wave's LDS stores before its later LDS loads of other lanes' data.  DS
    // operations of one wave execute in issue order; lgkmcnt(0) drains them and the wave
    // barrier pins the compiler.  Deliberately NOT a fence: a wavefront-scope fence also
    // emits vmcnt(0), which would drain the next frame's prefetch and the previous frame's
    // stores at every exchange.
#if AFX_V & 1
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#else
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0), vmcnt/expcnt untouched
    __builtin_amdgcn_wave_barrier();
#endif
}

// |X|^2 (optionally mapped) of the conjugate pair (k, 1024-k) from A = Z[k], B = Z[1024-k]
__device__ __forceinline__ void split_pair(v2 A, v2 B, v2 w /* 0.5 W_2048^k */, float &pk, float &pq) {
    const v2 e2 = pk_add_conj(A, B);   // 2 E
    const v2 d = pk_sub_conj(A, B);    // 2 i O  ->  2 O = -i d
    const v2 wo = cmul_mi(d, w);       // W O   (w carries the 1/2)
    const v2 x = e2 * 0.5f + wo;       // X[k]
    const v2 y = e2 * 0.5f - wo;       // conj(X[1024-k])
    pk = x.x * x.x + x.y * x.y;
    pq = y.x * y.x + y.y * y.y;
}

// complex result mode (bftObj_setResultType 0): the spectrum value itself (sq = false:
// data type MAG) or its complex square (sq = true: POWER, __mcsquare, flux_complex.c:469-503)
// of bins k and 1024-k; the bank is real, so real and imaginary parts go through it separately
__device__ __forceinline__ void split_pair_c(v2 A, v2 B, v2 w, bool sq, float &kr, float &ki, float &qr,
                                             float &qi) {
    const v2 e2 = pk_add_conj(A, B);
    const v2 d = pk_sub_conj(A, B);
    const v2 wo = cmul_mi(d, w);
    const v2 x = e2 * 0.5f + wo;  // X[k]
    const v2 y = e2 * 0.5f - wo;  // conj(X[1024-k])
    if (sq) {
        kr = x.x * x.x - x.y * x.y;
        ki = 2.f * (x.x * x.y);
        qr = y.x * y.x - y.y * y.y;
        qi = -2.f * (y.x * y.y);
    } else {
        kr = x.x;
        ki = x.y;
        qr = y.x;
        qi = -y.y;
    }
}

// GENERAL = false: plain |S|^2 (the hot configuration; no sqrt/pow code in the loop)
// CPLX (with GENERAL): complex result, two passes of the filter-bank stage (real, imaginary)
// SHIFT: consecutive frames of a clip overlap; with hop = 128*SHIFT samples the next
// frame's register image is the current one moved down by SHIFT registers, so only SHIFT
// new float2 per lane are fetched per frame (SHIFT = 0: every frame is fetched whole)
// SPLIT: the plan's slots hold row SEGMENTS (afx_bandplan_build_split): the slot results go
// through 129 floats of the (by then dead) exchange buffer and lane l adds up rows l and l + 64
template <int TA, int TB, bool GENERAL, int SHIFT, bool CPLX = false, bool SPLIT = false>
__global__ __launch_bounds__(WAVES * 64, 3) void k_stft_mel_banded(KArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    constexpr int WP = wpitch(TA, TB);
    v2 *tabWin = reinterpret_cast<v2 *>(smem);
    v2 *tabTw1 = tabWin + TAB_WIN_F2;
    v2 *tabTw2 = tabTw1 + TAB_TW1_F2;
    v2 *tabTw3 = tabTw2 + TAB_TW2_F2;
    float *tabW = reinterpret_cast<float *>(smem + TAB_BYTES);
    v2 *ex = reinterpret_cast<v2 *>(smem + TAB_BYTES + 64 * WP * 4 + wave * WAVE_LDS_BYTES);
    float *prow = reinterpret_cast<float *>(ex);  // aliases ex: written only after the pair reads

    // ---- workgroup-shared tables -> LDS (once) ----------------------------------
    {
        const v2 *gTw1 = reinterpret_cast<const v2 *>(a.tw1);
        const v2 *gTw2 = reinterpret_cast<const v2 *>(a.tw2), *gTw3 = reinterpret_cast<const v2 *>(a.tw3);
        // pair layout for ds_read_b128: entry (n1, lane) of a [16][64] table sits at
        // [(n1 >> 1)][lane][n1 & 1], so one 16-byte read returns the rows 2j and 2j + 1 of a lane
        auto pairIdx = [](int i) { return ((i >> 7) << 7) + ((i & 63) << 1) + ((i >> 6) & 1); };
        for (int i = threadIdx.x; i < TAB_WIN_F2; i += WAVES * 64)
            tabWin[((AFX_V & 2) && !CPLX) ? pairIdx(i) : i] = reinterpret_cast<const v2 *>(a.win2)[i];
        for (int i = threadIdx.x; i < TAB_TW1_F2; i += WAVES * 64) tabTw1[((AFX_V & 4) && !CPLX) ? pairIdx(i) : i] = gTw1[i];
        for (int i = threadIdx.x; i < TAB_TW3_F2; i += WAVES * 64) tabTw3[i] = gTw3[i];
        for (int i = threadIdx.x; i < 64 * WP; i += WAVES * 64) tabW[i] = a.wLane[i];
        if (threadIdx.x < TAB_TW2_F2) tabTw2[threadIdx.x] = gTw2[threadIdx.x];
    }
    __syncthreads();

    // ---- per-lane constants -----------------------------------------------------
    const int k1 = lane >> 2, m2 = lane & 3;
    const int startA = a.meta[lane], startB = a.meta[64 + lane];
    const int rowA = a.meta[128 + lane], rowB = a.meta[192 + lane];
    // split plans: the (up to four) slots whose results make up rows lane and lane + 64
    const unsigned seg0 = SPLIT ? (unsigned)a.meta[256 + lane] : 0u, seg1 = SPLIT ? (unsigned)a.meta[320 + lane] : 0u;
    const float4 *wrow = reinterpret_cast<const float4 *>(tabW + lane * WP);
    const int qm = (256 - lane) & 255;  // mirror base of q = lane (lane 0 mirrors itself)

    const long long gw = (long long)blockIdx.x * WAVES + wave;
    long long f = gw * a.framesPerWave;
    long long fEnd = f + a.framesPerWave;
    if (fEnd > a.totalFrames) fEnd = a.totalFrames;
    if (f >= fEnd) return;
    int clip = (int)(f / a.timeLength);
    int t = (int)(f - (long long)clip * a.timeLength);

    // raw samples of the frame about to be transformed: raw[n1] = (x[2n], x[2n+1]), n = 64 n1 + lane
    v2 raw[16];
    auto fetch = [&](const float *px, int first) {
        if (a.aligned) {
            const v2 *p2 = reinterpret_cast<const v2 *>(px);
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1)
                if (n1 >= first) raw[n1] = p2[64 * n1 + lane];
        } else {
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1)
                if (n1 >= first) {
                    const int n = 64 * n1 + lane;
                    raw[n1] = v2{px[2 * n], px[2 * n + 1]};
                }
        }
    };
    fetch(a.x + (long long)clip * a.clipStride + (long long)t * a.hop, 0);
    // hand-issued LDS reads need register headroom: the complex-result instantiation (two extra
    // 20-float arrays) keeps the compiler's reads
    constexpr bool HAND = !(AFX_V & 512) && !CPLX;

    for (; f < fEnd; ++f) {
        v2 v[16];
        // ---- 1. window (samples were fetched during the previous frame) ---------------
        if constexpr (HAND && (AFX_V & 2)) {
            typedef float v4f __attribute__((ext_vector_type(4)));
            const unsigned aw = lds_addr(tabWin + 2 * lane);
            v4f wv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) RD128(wv[j], aw, 1024 * j);
            LDS_WAIT_N(4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                PIN(wv[j]);
                v[2 * j] = raw[2 * j] * v2{wv[j].x, wv[j].y};
                v[2 * j + 1] = raw[2 * j + 1] * v2{wv[j].z, wv[j].w};
            }
            LDS_WAIT_N(0);
#pragma unroll
            for (int j = 4; j < 8; ++j) {
                PIN(wv[j]);
                v[2 * j] = raw[2 * j] * v2{wv[j].x, wv[j].y};
                v[2 * j + 1] = raw[2 * j + 1] * v2{wv[j].z, wv[j].w};
            }
        } else if constexpr (HAND) {
            const unsigned aw = lds_addr(tabWin + lane);
            v2 wv[16];
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) RD64(wv[n1], aw, 512 * n1);
            LDS_WAIT_N(8);  // in-order return: the first eight have landed
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1) {
                PIN(wv[n1]);
                v[n1] = raw[n1] * wv[n1];
            }
            LDS_WAIT_N(0);
#pragma unroll
            for (int n1 = 8; n1 < 16; ++n1) {
                PIN(wv[n1]);
                v[n1] = raw[n1] * wv[n1];
            }
        } else {
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) v[n1] = raw[n1] * tabWin[64 * n1 + lane];
        }
        // ---- 1b. start fetching the next frame: in flight under the whole transform ---
        if (f + 1 < fEnd) {
            int tn = t + 1, cn = clip;
            if (tn == a.timeLength) {
                tn = 0;
                ++cn;
            }
            const float *pn = a.x + (long long)cn * a.clipStride + (long long)tn * a.hop;
            if (SHIFT > 0 && tn != 0) {
#pragma unroll
                for (int n1 = 0; n1 + SHIFT < 16; ++n1) raw[n1] = raw[n1 + SHIFT];
                fetch(pn, 16 - SHIFT);
            } else {
                fetch(pn, 0);
            }
        }

        // ---- 2a. radix-16 over n1, twiddle, transpose through LDS ---------------
        v2 t1[16];
        if constexpr (!HAND) {
#pragma unroll
            for (int k = 1; k < 16; ++k) t1[k] = tabTw1[k * 64 + lane];
        }
        dft16(v);
        if constexpr (HAND && (AFX_V & 4)) {
            typedef float v4f __attribute__((ext_vector_type(4)));
            const unsigned a1 = lds_addr(tabTw1 + 2 * lane);
            v4f tq[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) RD128(tq[j], a1, 1024 * j);
            lds_wait();
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                PIN(tq[j]);
                t1[2 * j] = v2{tq[j].x, tq[j].y};
                t1[2 * j + 1] = v2{tq[j].z, tq[j].w};
            }
        } else if constexpr (HAND) {
            // requested after the butterflies: held across them they cost 30 live VGPRs and spill
            const unsigned a1 = lds_addr(tabTw1 + lane);
#pragma unroll
            for (int k = 1; k < 16; ++k) RD64(t1[k], a1, 512 * k);
            lds_wait();
#pragma unroll
            for (int k = 1; k < 16; ++k) PIN(t1[k]);
        }
        ex[lane] = v[0];
#pragma unroll
        for (int k = 1; k < 16; ++k) ex[k * EX_PITCH + lane] = cmul(v[rev4(k)], t1[k]);
        wave_lds_sync();
        if constexpr (HAND) {
            const unsigned ae = lds_addr(ex + k1 * EX_PITCH + m2);
#pragma unroll
            for (int m1 = 0; m1 < 16; ++m1) RD64(v[m1], ae, 32 * m1);
            wave_lds_sync();
#pragma unroll
            for (int m1 = 0; m1 < 16; ++m1) PIN(v[m1]);
        } else {
#pragma unroll
            for (int m1 = 0; m1 < 16; ++m1) v[m1] = ex[k1 * EX_PITCH + 4 * m1 + m2];
            wave_lds_sync();
        }

        // ---- 2b. radix-16 over m1, twiddle W_64^(m2*j1) -> image V[m2][q = k1 + 16 j1] ----
        // all twiddles are read in one batch before the butterflies (LDS reads interleaved with
        // the image writes would be serialised one round trip at a time: same array, may alias)
        dft16(v);
        if constexpr (HAND && (AFX_V & 8)) {
            typedef float v4f __attribute__((ext_vector_type(4)));
            const unsigned a2 = lds_addr(tabTw2 + m2 * 16);
            v4f tq[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) RD128(tq[j], a2, 16 * j);
            lds_wait();
#pragma unroll
            for (int j = 0; j < 8; ++j) PIN(tq[j]);
            ex[m2 * 260 + k1] = v[0];
#pragma unroll
            for (int j1 = 1; j1 < 16; ++j1) {
                const v2 tw = (j1 & 1) ? v2{tq[j1 >> 1].z, tq[j1 >> 1].w} : v2{tq[j1 >> 1].x, tq[j1 >> 1].y};
                ex[m2 * 260 + k1 + 16 * j1] = cmul(v[rev4(j1)], tw);
            }
        } else {
        ex[m2 * 260 + k1] = v[0];
#pragma unroll
        for (int j1 = 1; j1 < 16; ++j1) ex[m2 * 260 + k1 + 16 * j1] = cmul(v[rev4(j1)], tabTw2[m2 * 16 + j1]);
        }
        wave_lds_sync();

        // ---- 3. last radix-4 + real-input split -> spectrum values in registers -------
        float pk[20], pq[20];
        float pkI[CPLX ? 20 : 1], pqI[CPLX ? 20 : 1];  // imaginary parts (complex result mode)
        // every LDS operand of this stage is requested up front (24 + 6 reads in flight)
        v2 zin[2][8], w3[2][4];
        v2 zc0, zc1, zc2, zc3, wc0, wc1;
        if constexpr (HAND) {
            const unsigned aq = lds_addr(ex + lane), aq0 = lds_addr(ex + qm), aq1 = lds_addr(ex + 192 - lane);
            const unsigned a3 = lds_addr(tabTw3 + lane), ac = lds_addr(ex), a3c = lds_addr(tabTw3);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                RD64(zin[0][m], aq, 2080 * m);
                RD64(zin[0][4 + m], aq0, 2080 * m);
                RD64(w3[0][m], a3, 2048 * m);
            }
            RD64(zc0, ac, 8 * 128);
            RD64(zc1, ac, 8 * (260 + 128));
            RD64(zc2, ac, 8 * (520 + 128));
            RD64(zc3, ac, 8 * (780 + 128));
            RD64(wc0, a3c, 8 * 128);
            RD64(wc1, a3c, 8 * 384);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                RD64(zin[1][m], aq, 2080 * m + 512);
                RD64(zin[1][4 + m], aq1, 2080 * m);
                RD64(w3[1][m], a3, 2048 * m + 512);
            }
            lds_wait();
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                PIN(zin[0][m]);
                PIN(zin[1][m]);
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                PIN(w3[0][m]);
                PIN(w3[1][m]);
            }
            PIN(zc0); PIN(zc1); PIN(zc2); PIN(zc3); PIN(wc0); PIN(wc1);
        } else {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int q = lane + 64 * s;
                const int qp = s == 0 ? qm : 192 - lane;  // (256 - q) & 255
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    zin[s][m] = ex[260 * m + q];
                    zin[s][4 + m] = ex[260 * m + qp];
                    w3[s][m] = tabTw3[q + 256 * m];
                }
            }
            zc0 = ex[128]; zc1 = ex[260 + 128]; zc2 = ex[520 + 128]; zc3 = ex[780 + 128];
            wc0 = tabTw3[128]; wc1 = tabTw3[384];
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            v2 za0 = zin[s][0], za1 = zin[s][1], za2 = zin[s][2], za3 = zin[s][3];
            v2 zb0 = zin[s][4], zb1 = zin[s][5], zb2 = zin[s][6], zb3 = zin[s][7];
            dft4(za0, za1, za2, za3);  // Z[q + 256 j]
            dft4(zb0, zb1, zb2, zb3);  // Z[qp + 256 j]
            // partner of Z[q + 256 j] is Z[qp + 256 (3 - j)]; for q = 0 it is Z[256 ((4 - j) & 3)]
            v2 b0 = zb3, b1 = zb2, b2 = zb1, b3 = zb0;
            if (s == 0) {
                const bool self = (lane == 0);
                b0 = self ? zb0 : zb3;
                b1 = self ? zb3 : zb2;
                b2 = self ? zb2 : zb1;
                b3 = self ? zb1 : zb0;
            }
            if (CPLX) {
                const bool sq = a.specMap == 4;
                split_pair_c(za0, b0, w3[s][0], sq, pk[8 * s + 0], pkI[CPLX ? 8 * s + 0 : 0], pq[8 * s + 0], pqI[CPLX ? 8 * s + 0 : 0]);
                split_pair_c(za1, b1, w3[s][1], sq, pk[8 * s + 1], pkI[CPLX ? 8 * s + 1 : 0], pq[8 * s + 1], pqI[CPLX ? 8 * s + 1 : 0]);
                split_pair_c(za2, b2, w3[s][2], sq, pk[8 * s + 2], pkI[CPLX ? 8 * s + 2 : 0], pq[8 * s + 2], pqI[CPLX ? 8 * s + 2 : 0]);
                split_pair_c(za3, b3, w3[s][3], sq, pk[8 * s + 3], pkI[CPLX ? 8 * s + 3 : 0], pq[8 * s + 3], pqI[CPLX ? 8 * s + 3 : 0]);
            } else {
                split_pair(za0, b0, w3[s][0], pk[8 * s + 0], pq[8 * s + 0]);
                split_pair(za1, b1, w3[s][1], pk[8 * s + 1], pq[8 * s + 1]);
                split_pair(za2, b2, w3[s][2], pk[8 * s + 2], pq[8 * s + 2]);
                split_pair(za3, b3, w3[s][3], pk[8 * s + 3], pq[8 * s + 3]);
            }
        }
        {   // base 128 mirrors itself: bins 128, 384 and their partners 896, 640 (every lane
            // computes them, lane 0 stores them)
            dft4(zc0, zc1, zc2, zc3);
            if (CPLX) {
                const bool sq = a.specMap == 4;
                split_pair_c(zc0, zc3, wc0, sq, pk[16], pkI[CPLX ? 16 : 0], pq[16], pqI[CPLX ? 16 : 0]);
                split_pair_c(zc1, zc2, wc1, sq, pk[17], pkI[CPLX ? 17 : 0], pq[17], pqI[CPLX ? 17 : 0]);
            } else {
                split_pair(zc0, zc3, wc0, pk[16], pq[16]);
                split_pair(zc1, zc2, wc1, pk[17], pq[17]);
            }
        }
        if (CPLX) {
        } else if (GENERAL && a.specMap == 1) {
#pragma unroll
            for (int i = 0; i < 18; ++i) {
                pk[i] = sqrtf(pk[i]);
                pq[i] = sqrtf(pq[i]);
            }
        } else if (GENERAL && a.specMap == 2) {
#pragma unroll
            for (int i = 0; i < 18; ++i) {
                pk[i] = powf(pk[i], a.normValue);
                pq[i] = powf(pq[i], a.normValue);
            }
        }
        wave_lds_sync();  // every lane has its bins in registers; ex becomes the power row
#pragma unroll
        for (int pass = 0; pass < (CPLX ? 2 : 1); ++pass) {
        if (pass == 1) wave_lds_sync();  // the real pass has read the row; now the imaginary parts
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = lane + 64 * s + 256 * j;
                prow[k] = (CPLX && pass) ? pkI[CPLX ? 8 * s + j : 0] : pk[8 * s + j];
                prow[MC - k] = (CPLX && pass) ? pqI[CPLX ? 8 * s + j : 0] : pq[8 * s + j];
            }
        }
        if (lane == 0) {
            prow[128] = (CPLX && pass) ? pkI[CPLX ? 16 : 0] : pk[16];
            prow[896] = (CPLX && pass) ? pqI[CPLX ? 16 : 0] : pq[16];
            prow[384] = (CPLX && pass) ? pkI[CPLX ? 17 : 0] : pk[17];
            prow[640] = (CPLX && pass) ? pqI[CPLX ? 17 : 0] : pq[17];
        }
        // zero pad behind bin 1024: the fixed-length band loops read it with zero weights
        prow[1025 + lane] = 0.f;
        if (lane < PROW_F - 1025 - 64) prow[1025 + 64 + lane] = 0.f;
        wave_lds_sync();

        // ---- 4. banded filter bank: weights by ds_read_b128, power row by immediate-offset
        //         ds_read_b64 (conflict-free by the plan's bank-aware lane assignment) ------
        float accA, accB;
        {
            // startA / startB are even: the power row is read as pairs (ds_read_b64)
            const v2 *pa = reinterpret_cast<const v2 *>(prow + startA);
            v2 sA = {0.f, 0.f}, sB = {0.f, 0.f};
            const v2 *pb = reinterpret_cast<const v2 *>(prow + startB);
            // operands are requested in blocks of 4 quads (12 LDS reads in flight) so that one
            // LDS round trip is paid per block instead of per quad
            if constexpr (HAND) {
            // every operand by hand-issued reads, the NEXT block of four quads requested before
            // this block's values are waited for (in-order return: lgkmcnt(12) = "all but the
            // 12 reads of the next block"), so one LDS round trip overlaps the previous block's FMAs
            typedef float v4f __attribute__((ext_vector_type(4)));
            constexpr int QA = TA / 4, QB = TB / 4, QT = QA + QB, BLK = 4, NB = (QT + BLK - 1) / BLK;
            const unsigned apa = lds_addr(pa), apb = lds_addr(pb), aw = lds_addr(wrow);
            v4f w[2][BLK];
            v2 p0[2][BLK], p1[2][BLK];
            auto request = [&](int blk, v4f (&wq)[BLK], v2 (&q0v)[BLK], v2 (&q1v)[BLK]) {
#pragma unroll
                for (int i = 0; i < BLK; ++i) {
                    const int q = blk * BLK + i;
                    if (q >= QT) continue;
                    RD128(wq[i], aw, 16 * q);
                    if (q < QA) {
                        RD64(q0v[i], apa, 16 * q);
                        RD64(q1v[i], apa, 16 * q + 8);
                    } else {
                        RD64(q0v[i], apb, 16 * (q - QA));
                        RD64(q1v[i], apb, 16 * (q - QA) + 8);
                    }
                }
            };
            request(0, w[0], p0[0], p1[0]);
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) {
                const int cur = blk & 1;
                const int nextQuads = (blk + 1 < NB) ? ((QT - (blk + 1) * BLK) < BLK ? (QT - (blk + 1) * BLK) : BLK) : 0;
                if (blk + 1 < NB) request(blk + 1, w[cur ^ 1], p0[cur ^ 1], p1[cur ^ 1]);
                if (nextQuads == 4) LDS_WAIT_N(12);
                else if (nextQuads == 3) LDS_WAIT_N(9);
                else if (nextQuads == 2) LDS_WAIT_N(6);
                else if (nextQuads == 1) LDS_WAIT_N(3);
                else LDS_WAIT_N(0);
#pragma unroll
                for (int i = 0; i < BLK; ++i) {
                    if (blk * BLK + i >= QT) continue;
                    PIN(w[cur][i]);
                    PIN(p0[cur][i]);
                    PIN(p1[cur][i]);
                    const int q = blk * BLK + i;
                    if (q < QA) {
                        sA += v2{w[cur][i].x, w[cur][i].y} * p0[cur][i];
                        sA += v2{w[cur][i].z, w[cur][i].w} * p1[cur][i];
                    } else {
                        sB += v2{w[cur][i].x, w[cur][i].y} * p0[cur][i];
                        sB += v2{w[cur][i].z, w[cur][i].w} * p1[cur][i];
                    }
                }
            }
            } else {
            constexpr int QA = TA / 4, QB = TB / 4, QT = QA + QB, BLK = 4;
#pragma unroll
            for (int q0 = 0; q0 < QT; q0 += BLK) {
                float4 w[BLK];
                v2 p0[BLK], p1[BLK];
#pragma unroll
                for (int i = 0; i < BLK; ++i) {
                    const int q = q0 + i;
                    if (q < QT) {
                        w[i] = wrow[q];
                        const v2 *src = q < QA ? pa + 2 * q : pb + 2 * (q - QA);
                        p0[i] = src[0];
                        p1[i] = src[1];
                    }
                }
#pragma unroll
                for (int i = 0; i < BLK; ++i) {
                    const int q = q0 + i;
                    if (q < QT) {
                        if (q < QA) {
                            sA += v2{w[i].x, w[i].y} * p0[i];
                            sA += v2{w[i].z, w[i].w} * p1[i];
                        } else {
                            sB += v2{w[i].x, w[i].y} * p0[i];
                            sB += v2{w[i].z, w[i].w} * p1[i];
                        }
                    }
                }
            }
            }
            accA = sA.x + sA.y;
            accB = sB.x + sB.y;
        }
        if (GENERAL && !CPLX && !SPLIT && a.postPow) {
            accA = powf(accA, a.normValue);
            accB = powf(accB, a.normValue);
        }
        // ---- 5. store ---------------------------------------------------------------
        float *orow = ((CPLX && pass) ? a.outIm : a.out) + f * a.num;
        if constexpr (SPLIT) {
            // slot results -> LDS (behind the power row: that part of the exchange buffer is
            // dead since stage 3), then every row is the sum of its segments in ascending bins
            float *part = prow + PROW_F;
            part[lane] = accA;
            part[64 + lane] = accB;
            if (lane == 0) part[128] = 0.f;
            wave_lds_sync();
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const unsigned u = h ? seg1 : seg0;
                float sum = part[u & 255u] + part[(u >> 8) & 255u];
                sum += part[(u >> 16) & 255u];
                sum += part[u >> 24];
                if (GENERAL && !CPLX && a.postPow) sum = powf(sum, a.normValue);
                if (lane + 64 * h < a.num) orow[lane + 64 * h] = sum;
            }
        } else {
            if (rowA >= 0) orow[rowA] = accA;
            if (rowB >= 0) orow[rowB] = accB;
        }
        }  // pass
        wave_lds_sync();  // the next frame overwrites ex / prow

        if (++t == a.timeLength) {
            t = 0;
            ++clip;
        }
    }
}

// ---- two frames per wave -------------------------------------------------------------------
// k_stft_mel_banded is bound by dependency latency at 12 waves per CU, not by
// LDS bandwidth or VALU throughput.  This variant (hop 512, real results) lets one wave carry
// TWO consecutive frames t, t+1 of a clip through every stage: two independent butterfly
// chains for the scheduler to interleave, and every table read (window, W_1024, W_64, W_2048)
// and every filter-bank weight read serves both frames.  The frames overlap by 1536 samples:
// their union is 20 float2 registers per lane, the next pair re-uses 12 of them (8 fetched per
// pair = 4 per frame, as before).  The exchange image is used by frame A, then by frame B; the
// two power rows fit the same 8.8 KB.  8 waves per CU (2 per SIMD, 256-VGPR budget), 16 frames
// in flight per CU instead of 12.
constexpr int PWAVES = 8;
constexpr int PAIR_LDS_BYTES = 2 * PROW_F * 4;  // 8832 >= the 8704-byte exchange image
__host__ __device__ constexpr int pair_block_lds_bytes(int ta, int tb) {
    return TAB_BYTES + 64 * wpitch(ta, tb) * 4 + PWAVES * PAIR_LDS_BYTES;
}

template <int TA, int TB, bool GENERAL>
__global__ __launch_bounds__(PWAVES * 64, 2) void k_stft_mel_pair(KArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    constexpr int WP = wpitch(TA, TB);
    v2 *tabWin = reinterpret_cast<v2 *>(smem);
    v2 *tabTw1 = tabWin + TAB_WIN_F2;
    v2 *tabTw2 = tabTw1 + TAB_TW1_F2;
    v2 *tabTw3 = tabTw2 + TAB_TW2_F2;
    float *tabW = reinterpret_cast<float *>(smem + TAB_BYTES);
    v2 *ex = reinterpret_cast<v2 *>(smem + TAB_BYTES + 64 * WP * 4 + wave * PAIR_LDS_BYTES);
    float *prow = reinterpret_cast<float *>(ex);  // [2][PROW_F], aliases the exchange image

    {
        const v2 *gTw1 = reinterpret_cast<const v2 *>(a.tw1);
        const v2 *gTw2 = reinterpret_cast<const v2 *>(a.tw2), *gTw3 = reinterpret_cast<const v2 *>(a.tw3);
        for (int i = threadIdx.x; i < TAB_WIN_F2; i += PWAVES * 64) tabWin[i] = reinterpret_cast<const v2 *>(a.win2)[i];
        for (int i = threadIdx.x; i < TAB_TW1_F2; i += PWAVES * 64) tabTw1[i] = gTw1[i];
        for (int i = threadIdx.x; i < TAB_TW3_F2; i += PWAVES * 64) tabTw3[i] = gTw3[i];
        for (int i = threadIdx.x; i < 64 * WP; i += PWAVES * 64) tabW[i] = a.wLane[i];
        if (threadIdx.x < TAB_TW2_F2) tabTw2[threadIdx.x] = gTw2[threadIdx.x];
    }
    __syncthreads();

    const int k1 = lane >> 2, m2 = lane & 3;
    const int startA = a.meta[lane], startB = a.meta[64 + lane];
    const int rowA = a.meta[128 + lane], rowB = a.meta[192 + lane];
    const float *wrow = tabW + lane * WP;
    const int qm = (256 - lane) & 255;

    const long long gw = (long long)blockIdx.x * PWAVES + wave;
    long long f = gw * a.framesPerWave;
    long long fEnd = f + a.framesPerWave;
    if (fEnd > a.totalFrames) fEnd = a.totalFrames;
    if (f >= fEnd) return;
    int clip = (int)(f / a.timeLength);
    int t = (int)(f - (long long)clip * a.timeLength);

    // raw[r] = (x[2n], x[2n+1]), n = 64 r + lane, r < 20: frame t is r 0..15, frame t+1 is r 4..19
    v2 raw[20];
    auto fetch = [&](const float *px, int first, bool hasNext) {
        const v2 *p2 = reinterpret_cast<const v2 *>(px);  // launch requires float2-aligned frames
#pragma unroll
        for (int r = 0; r < 20; ++r)
            if (r >= first) raw[r] = (r < 16 || hasNext) ? p2[64 * r + lane] : v2{0.f, 0.f};
    };
    fetch(a.x + (long long)clip * a.clipStride + (long long)t * a.hop, 0, t + 1 < a.timeLength);

    while (f < fEnd) {
        const bool two = (f + 1 < fEnd) && (t + 1 < a.timeLength);
        v2 v[2][16];
        // ---- 1. window: one read per sample position serves both frames -------------------
        {
            const unsigned aw = lds_addr(tabWin + lane);
            v2 wv[16];
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) RD64(wv[n1], aw, 512 * n1);
            lds_wait();
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {
                PIN(wv[n1]);
                v[0][n1] = raw[n1] * wv[n1];
                v[1][n1] = raw[n1 + 4] * wv[n1];
            }
        }
        // ---- 1b. fetch for the next pair -----------------------------------------------
        const int adv = two ? 2 : 1;
        {
            int tn = t + adv, cn = clip;
            if (tn >= a.timeLength) {
                tn = 0;
                ++cn;
            }
            if (f + adv < fEnd) {
                const float *pn = a.x + (long long)cn * a.clipStride + (long long)tn * a.hop;
                const bool hasNext = tn + 1 < a.timeLength;
                if (two && tn != 0) {  // same clip, two frames on: registers 8..19 become 0..11
#pragma unroll
                    for (int r = 0; r < 12; ++r) raw[r] = raw[r + 8];
                    fetch(pn, 12, hasNext);
                } else {
                    fetch(pn, 0, hasNext);
                }
            }
            t = tn;
            clip = cn;
        }
        // ---- 2a. radix-16 over n1, twiddle, transpose through LDS (A, then B) -------------
        {
            dft16(v[0]);
            dft16(v[1]);
            v2 t1[16];
            const unsigned a1 = lds_addr(tabTw1 + lane);
#pragma unroll
            for (int k = 1; k < 16; ++k) RD64(t1[k], a1, 512 * k);
            lds_wait();
#pragma unroll
            for (int k = 1; k < 16; ++k) {
                PIN(t1[k]);
                v[0][rev4(k)] = cmul(v[0][rev4(k)], t1[k]);
                v[1][rev4(k)] = cmul(v[1][rev4(k)], t1[k]);
            }
        }
        const unsigned ae = lds_addr(ex + k1 * EX_PITCH + m2);
#pragma unroll
        for (int fr = 0; fr < 2; ++fr) {
            ex[lane] = v[fr][0];
#pragma unroll
            for (int k = 1; k < 16; ++k) ex[k * EX_PITCH + lane] = v[fr][rev4(k)];
            wave_lds_sync();
#pragma unroll
            for (int m1 = 0; m1 < 16; ++m1) RD64(v[fr][m1], ae, 32 * m1);
            wave_lds_sync();
#pragma unroll
            for (int m1 = 0; m1 < 16; ++m1) PIN(v[fr][m1]);
        }
        // ---- 2b. radix-16 over m1, twiddle W_64^(m2 j1) ------------------------------------
        dft16(v[0]);
        dft16(v[1]);
        {
            v2 t2[16];
#pragma unroll
            for (int j1 = 1; j1 < 16; ++j1) t2[j1] = tabTw2[m2 * 16 + j1];
#pragma unroll
            for (int j1 = 1; j1 < 16; ++j1) {
                v[0][rev4(j1)] = cmul(v[0][rev4(j1)], t2[j1]);
                v[1][rev4(j1)] = cmul(v[1][rev4(j1)], t2[j1]);
            }
        }
        // ---- 3. image V[m2][q], last radix-4 + real-input split (A, then B) ----------------
        float pk[2][20], pq[2][20];
        v2 w3[2][4], wc0, wc1;
        {
            const unsigned a3 = lds_addr(tabTw3 + lane), a3c = lds_addr(tabTw3);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                RD64(w3[0][m], a3, 2048 * m);
                RD64(w3[1][m], a3, 2048 * m + 512);
            }
            RD64(wc0, a3c, 8 * 128);
            RD64(wc1, a3c, 8 * 384);
        }
#pragma unroll
        for (int fr = 0; fr < 2; ++fr) {
            ex[m2 * 260 + k1] = v[fr][0];
#pragma unroll
            for (int j1 = 1; j1 < 16; ++j1) ex[m2 * 260 + k1 + 16 * j1] = v[fr][rev4(j1)];
            wave_lds_sync();
            v2 zin[2][8], zc0, zc1, zc2, zc3;
            {
                const unsigned aq = lds_addr(ex + lane), aq0 = lds_addr(ex + qm), aq1 = lds_addr(ex + 192 - lane);
                const unsigned ac = lds_addr(ex);
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    RD64(zin[0][m], aq, 2080 * m);
                    RD64(zin[0][4 + m], aq0, 2080 * m);
                    RD64(zin[1][m], aq, 2080 * m + 512);
                    RD64(zin[1][4 + m], aq1, 2080 * m);
                }
                RD64(zc0, ac, 8 * 128);
                RD64(zc1, ac, 8 * (260 + 128));
                RD64(zc2, ac, 8 * (520 + 128));
                RD64(zc3, ac, 8 * (780 + 128));
                wave_lds_sync();  // drains the reads (and the table reads above on the first pass)
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    PIN(zin[0][m]);
                    PIN(zin[1][m]);
                }
                PIN(zc0); PIN(zc1); PIN(zc2); PIN(zc3);
                if (fr == 0) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        PIN(w3[0][m]);
                        PIN(w3[1][m]);
                    }
                    PIN(wc0);
                    PIN(wc1);
                }
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                v2 za0 = zin[s][0], za1 = zin[s][1], za2 = zin[s][2], za3 = zin[s][3];
                v2 zb0 = zin[s][4], zb1 = zin[s][5], zb2 = zin[s][6], zb3 = zin[s][7];
                dft4(za0, za1, za2, za3);
                dft4(zb0, zb1, zb2, zb3);
                v2 b0 = zb3, b1 = zb2, b2 = zb1, b3 = zb0;
                if (s == 0) {
                    const bool self = (lane == 0);
                    b0 = self ? zb0 : zb3;
                    b1 = self ? zb3 : zb2;
                    b2 = self ? zb2 : zb1;
                    b3 = self ? zb1 : zb0;
                }
                split_pair(za0, b0, w3[s][0], pk[fr][8 * s + 0], pq[fr][8 * s + 0]);
                split_pair(za1, b1, w3[s][1], pk[fr][8 * s + 1], pq[fr][8 * s + 1]);
                split_pair(za2, b2, w3[s][2], pk[fr][8 * s + 2], pq[fr][8 * s + 2]);
                split_pair(za3, b3, w3[s][3], pk[fr][8 * s + 3], pq[fr][8 * s + 3]);
            }
            dft4(zc0, zc1, zc2, zc3);
            split_pair(zc0, zc3, wc0, pk[fr][16], pq[fr][16]);
            split_pair(zc1, zc2, wc1, pk[fr][17], pq[fr][17]);
        }
        if (GENERAL && a.specMap == 1) {
#pragma unroll
            for (int fr = 0; fr < 2; ++fr)
#pragma unroll
                for (int i = 0; i < 18; ++i) {
                    pk[fr][i] = sqrtf(pk[fr][i]);
                    pq[fr][i] = sqrtf(pq[fr][i]);
                }
        } else if (GENERAL && a.specMap == 2) {
#pragma unroll
            for (int fr = 0; fr < 2; ++fr)
#pragma unroll
                for (int i = 0; i < 18; ++i) {
                    pk[fr][i] = powf(pk[fr][i], a.normValue);
                    pq[fr][i] = powf(pq[fr][i], a.normValue);
                }
        }
        // ---- power rows of both frames (the image is no longer needed) ----------------------
#pragma unroll
        for (int fr = 0; fr < 2; ++fr) {
            float *pr = prow + fr * PROW_F;
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = lane + 64 * s + 256 * j;
                    pr[k] = pk[fr][8 * s + j];
                    pr[MC - k] = pq[fr][8 * s + j];
                }
            if (lane == 0) {
                pr[128] = pk[fr][16];
                pr[896] = pq[fr][16];
                pr[384] = pk[fr][17];
                pr[640] = pq[fr][17];
            }
            pr[1025 + lane] = 0.f;
            if (lane < PROW_F - 1025 - 64) pr[1025 + 64 + lane] = 0.f;
        }
        wave_lds_sync();
        // ---- 4. banded filter bank: every weight read serves both frames -------------------
        float acc[2][2];
        {
            typedef float v4f __attribute__((ext_vector_type(4)));
            constexpr int QA = TA / 4, QB = TB / 4, QT = QA + QB, BLK = 2, NB = (QT + BLK - 1) / BLK;
            const unsigned apa = lds_addr(prow + startA), apb = lds_addr(prow + startB), awr = lds_addr(wrow);
            v2 sA[2] = {{0.f, 0.f}, {0.f, 0.f}}, sB[2] = {{0.f, 0.f}, {0.f, 0.f}};
            v4f w[2][BLK];
            v2 p0[2][2][BLK], p1[2][2][BLK];  // [buffer][frame][quad]
            auto request = [&](int blk, int buf) {
#pragma unroll
                for (int i = 0; i < BLK; ++i) {
                    const int q = blk * BLK + i;
                    if (q >= QT) continue;
                    RD128(w[buf][i], awr, 16 * q);
#pragma unroll
                    for (int fr = 0; fr < 2; ++fr) {
                        if (q < QA) {
                            RD64(p0[buf][fr][i], apa, 16 * q + 4 * PROW_F * fr);
                            RD64(p1[buf][fr][i], apa, 16 * q + 8 + 4 * PROW_F * fr);
                        } else {
                            RD64(p0[buf][fr][i], apb, 16 * (q - QA) + 4 * PROW_F * fr);
                            RD64(p1[buf][fr][i], apb, 16 * (q - QA) + 8 + 4 * PROW_F * fr);
                        }
                    }
                }
            };
            request(0, 0);
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) {
                const int cur = blk & 1;
                const int nextQuads = (blk + 1 < NB) ? ((QT - (blk + 1) * BLK) < BLK ? (QT - (blk + 1) * BLK) : BLK) : 0;
                if (blk + 1 < NB) request(blk + 1, cur ^ 1);
                if (nextQuads == 2) LDS_WAIT_N(10);  // 5 reads per quad
                else if (nextQuads == 1) LDS_WAIT_N(5);
                else LDS_WAIT_N(0);
#pragma unroll
                for (int i = 0; i < BLK; ++i) {
                    const int q = blk * BLK + i;
                    if (q >= QT) continue;
                    PIN(w[cur][i]);
#pragma unroll
                    for (int fr = 0; fr < 2; ++fr) {
                        PIN(p0[cur][fr][i]);
                        PIN(p1[cur][fr][i]);
                        if (q < QA) {
                            sA[fr] += v2{w[cur][i].x, w[cur][i].y} * p0[cur][fr][i];
                            sA[fr] += v2{w[cur][i].z, w[cur][i].w} * p1[cur][fr][i];
                        } else {
                            sB[fr] += v2{w[cur][i].x, w[cur][i].y} * p0[cur][fr][i];
                            sB[fr] += v2{w[cur][i].z, w[cur][i].w} * p1[cur][fr][i];
                        }
                    }
                }
            }
#pragma unroll
            for (int fr = 0; fr < 2; ++fr) {
                acc[fr][0] = sA[fr].x + sA[fr].y;
                acc[fr][1] = sB[fr].x + sB[fr].y;
            }
        }
        if (GENERAL && a.postPow) {
#pragma unroll
            for (int fr = 0; fr < 2; ++fr) {
                acc[fr][0] = powf(acc[fr][0], a.normValue);
                acc[fr][1] = powf(acc[fr][1], a.normValue);
            }
        }
        // ---- 5. store -----------------------------------------------------------------
        {
            float *orow = a.out + f * a.num;
            if (rowA >= 0) orow[rowA] = acc[0][0];
            if (rowB >= 0) orow[rowB] = acc[0][1];
            if (two) {
                if (rowA >= 0) orow[a.num + rowA] = acc[1][0];
                if (rowB >= 0) orow[a.num + rowB] = acc[1][1];
            }
        }
        wave_lds_sync();  // the next pair overwrites the rows
        f += adv;
    }
}

struct Plan {
    int variant;
    int num;
    int split;  // slots hold row segments (AfxBandPlan.split)
    void *v2;   // real-result kernel of afx_melfused2.hip (this file keeps the complex-result modes)
    float2 *dWin2, *dTw1, *dTw2, *dTw3;
    float *dWLane;
    int *dMeta;
};

struct Variant {
    int tapsA, tapsB;
};
constexpr Variant kVariants[] = {{48, 16}, {72, 32}};
constexpr int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);

template <int TA, int TB, bool GENERAL, int SHIFT, bool CPLX = false, bool SPLIT = false>
int launch_variant(const Plan *p, const AfxMelFusedArgs *a, void *stream) {
    const long long total = (long long)a->batch * a->timeLength;
    if (total <= 0) return AFX_OK;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    }
    // one 12-wave workgroup is resident per CU (LDS ~140 KB); two rounds of workgroups keep
    // the tail short while each wave still streams a long contiguous, L2-friendly run of
    // frames (and re-uses 3/4 of every frame from registers)
    long long waves = (long long)cus * WAVES * 2;
    long long fpw = (total + waves - 1) / waves;
    if (fpw < 16) fpw = 16;
    const long long usedWaves = (total + fpw - 1) / fpw;
    const long long blocks = (usedWaves + WAVES - 1) / WAVES;

    KArgs k;
    k.x = a->x;
    k.clipStride = a->clipStride;
    k.totalFrames = total;
    k.timeLength = a->timeLength;
    k.hop = a->hop;
    k.framesPerWave = (int)fpw;
    k.aligned = ((a->clipStride & 1) == 0) && ((a->hop & 1) == 0) &&
                ((reinterpret_cast<uintptr_t>(a->x) & 7) == 0);
    k.win2 = p->dWin2;
    k.tw1 = p->dTw1;
    k.tw2 = p->dTw2;
    k.tw3 = p->dTw3;
    k.wLane = p->dWLane;
    k.meta = p->dMeta;
    k.specMap = a->specMap;
    k.postPow = a->postPow;
    k.normValue = a->normValue;
    k.out = a->out;
    k.outIm = a->outIm;
    k.num = p->num;
    constexpr size_t lds = (size_t)block_lds_bytes(TA, TB);
    static bool attrSet[AFX_MAX_DEVICES] = {};  // per device: the attribute lives in the device's code object
    const int attrDev = afxdev_current_device() & (AFX_MAX_DEVICES - 1);
    if (!attrSet[attrDev]) {
        AFX_HIP(hipFuncSetAttribute(
            reinterpret_cast<const void *>(k_stft_mel_banded<TA, TB, GENERAL, SHIFT, CPLX, SPLIT>),
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attrSet[attrDev] = true;
    }
    hipLaunchKernelGGL((k_stft_mel_banded<TA, TB, GENERAL, SHIFT, CPLX, SPLIT>), dim3((unsigned)blocks),
                       dim3(WAVES * 64), lds, (hipStream_t)stream, k);
    AFX_LAUNCH_CHECK("k_stft_mel_banded");
    return AFX_OK;
}


template <int TA, int TB>
int launch_pair(const Plan *p, const AfxMelFusedArgs *a, void *stream) {
    const long long total = (long long)a->batch * a->timeLength;
    if (total <= 0) return AFX_OK;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    long long waves = (long long)cus * PWAVES * 2;
    long long fpw = (total + waves - 1) / waves;
    if (fpw < 16) fpw = 16;
    fpw = (fpw + 1) & ~1LL;  // pairs
    const long long usedWaves = (total + fpw - 1) / fpw;
    const long long blocks = (usedWaves + PWAVES - 1) / PWAVES;
    KArgs k;
    k.x = a->x;
    k.clipStride = a->clipStride;
    k.totalFrames = total;
    k.timeLength = a->timeLength;
    k.hop = a->hop;
    k.framesPerWave = (int)fpw;
    k.aligned = 1;
    k.win2 = p->dWin2;
    k.tw1 = p->dTw1;
    k.tw2 = p->dTw2;
    k.tw3 = p->dTw3;
    k.wLane = p->dWLane;
    k.meta = p->dMeta;
    k.specMap = a->specMap;
    k.postPow = a->postPow;
    k.normValue = a->normValue;
    k.out = a->out;
    k.outIm = nullptr;
    k.num = p->num;
    constexpr size_t lds = (size_t)pair_block_lds_bytes(TA, TB);
    static bool attrSet[AFX_MAX_DEVICES] = {};  // per device: the attribute lives in the device's code object
    const int attrDev = afxdev_current_device() & (AFX_MAX_DEVICES - 1);
    if (!attrSet[attrDev]) {
        AFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_stft_mel_pair<TA, TB, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attrSet[attrDev] = true;
    }
    hipLaunchKernelGGL((k_stft_mel_pair<TA, TB, true>), dim3((unsigned)blocks), dim3(PWAVES * 64), lds,
                       (hipStream_t)stream, k);
    AFX_LAUNCH_CHECK("k_stft_mel_pair");
    return AFX_OK;
}

template <int TA, int TB, bool SPLIT = false>
int launch(const Plan *p, const AfxMelFusedArgs *a, void *stream) {
    // The GENERAL instantiation also serves plain |S|^2 (its map branches cost two scalar
    // compares per frame): with the hand-issued LDS reads the branch-free instantiation is
    // scheduled into 148 B/lane of scratch, the branched one into 150 VGPRs and none.
    const bool shift4 = (a->hop == 512);  // hop = 128 * SHIFT
    if (a->specMap >= 3) {  // complex result: S (3) or S^2 (4), real and imaginary planes
        if (!a->outIm) return AFX_ERR_ARG;
        return shift4 ? launch_variant<TA, TB, true, 4, true, SPLIT>(p, a, stream)
                      : launch_variant<TA, TB, true, 0, true, SPLIT>(p, a, stream);
    }
    // hop 512, float2-aligned frames, real results: two frames per wave
    const bool alignedFrames = ((a->clipStride & 1) == 0) && ((reinterpret_cast<uintptr_t>(a->x) & 7) == 0);
    if (!SPLIT && shift4 && alignedFrames && a->dataLength >= 2048 && getenv("AFX_PAIR"))
        return launch_pair<TA, TB>(p, a, stream);
    // register re-use of the overlapping frames for hop = 128 * SHIFT: N/8, N/4, N/2
    switch (a->hop) {
        case 256: return launch_variant<TA, TB, true, 2, false, SPLIT>(p, a, stream);
        case 512: return launch_variant<TA, TB, true, 4, false, SPLIT>(p, a, stream);
        case 1024: return launch_variant<TA, TB, true, 8, false, SPLIT>(p, a, stream);
        default: return launch_variant<TA, TB, true, 0, false, SPLIT>(p, a, stream);
    }
}

template <typename T>
int upload(T **dptr, const void *src, size_t bytes, void *stream) {
    int st = afxdev_malloc(reinterpret_cast<void **>(dptr), bytes);
    if (st != AFX_OK) return st;
    return afxdev_h2d(*dptr, src, bytes, stream);
}

}  // namespace

// real-result modes at n_fft = 2048: afx_melfused2.hip
extern "C" int afxk_mel2_create(void **plan, int variant, const float *hWindow, const AfxBandPlan *band, void *stream);
extern "C" int afxk_mel2_run(void *plan, const AfxMelFusedArgs *a, void *stream);
extern "C" void afxk_mel2_destroy(void *plan);

// n_fft = 1024 lives in afx_melfused1k.hip; its plans carry variant numbers >= 100
extern "C" int afxk_mel1k_variant(int tapsA, int tapsB);
extern "C" int afxk_mel1k_create(void **plan, const float *hWindow, const AfxBandPlan *band, void *stream);
extern "C" int afxk_mel1k_run(void *plan, const AfxMelFusedArgs *a, void *stream);
extern "C" void afxk_mel1k_destroy(void *plan);

// n_fft = 4096 lives in afx_melfused4k.hip (variant numbers >= 200)
extern "C" int afxk_mel4k_variant(int tapsA, int tapsB);
extern "C" int afxk_mel4k_create(void **plan, const float *hWindow, const AfxBandPlan *band, void *stream);
extern "C" int afxk_mel4k_run(void *plan, const AfxMelFusedArgs *a, void *stream);
extern "C" void afxk_mel4k_destroy(void *plan);
extern "C" int afxk_mel4k_kind(const void *plan);

extern "C" int afxk_melfused_variant(int radix2Exp, int tapsA, int tapsB) {
    if (getenv("AFX_NO_FUSED")) return -1;
    if (radix2Exp == 10) return afxk_mel1k_variant(tapsA, tapsB);
    if (radix2Exp == 12) return afxk_mel4k_variant(tapsA, tapsB);
    if (radix2Exp != 11) return -1;
    for (int i = 0; i < kNumVariants; ++i) {
        if (tapsA <= kVariants[i].tapsA && tapsB <= kVariants[i].tapsB) return i;
    }
    return -1;
}

extern "C" int afxk_melfused_kind(const void *plan) {
    const Plan *p = static_cast<const Plan *>(plan);
    if (!p) return 0;
    if (p->variant >= 200) return afxk_mel4k_kind(plan);
    if (p->variant >= 100) return 101;
    return p->split ? 2 : 1;
}

extern "C" void afxk_melfused_destroy(void *plan) {
    Plan *p = static_cast<Plan *>(plan);
    if (!p) return;
    if (p->variant >= 200) {
        afxk_mel4k_destroy(plan);
        return;
    }
    if (p->variant >= 100) {
        afxk_mel1k_destroy(plan);
        return;
    }
    afxk_mel2_destroy(p->v2);
    afxdev_free(p->dWin2);
    afxdev_free(p->dTw1);
    afxdev_free(p->dTw2);
    afxdev_free(p->dTw3);
    afxdev_free(p->dWLane);
    afxdev_free(p->dMeta);
    free(p);
}

extern "C" int afxk_melfused_create(void **plan, int radix2Exp, const float *hWindow,
                                    const AfxBandPlan *band, void *stream) {
    *plan = nullptr;
    const int variant = afxk_melfused_variant(radix2Exp, band->tapsA, band->tapsB);
    if (variant < 0) return AFX_ERR_UNSUPPORTED;
    if (variant >= 200) return afxk_mel4k_create(plan, hWindow, band, stream);
    if (variant >= 100) return afxk_mel1k_create(plan, hWindow, band, stream);
    const int TA = kVariants[variant].tapsA, TB = kVariants[variant].tapsB;
    Plan *p = static_cast<Plan *>(calloc(1, sizeof(Plan)));
    if (!p) return AFX_ERR_NOMEM;
    p->variant = variant;
    p->num = band->num;
    p->split = band->split;

    // twiddle tables in double, rounded once
    float *tw1 = static_cast<float *>(malloc(sizeof(float) * 2 * 16 * 64));
    float *tw2 = static_cast<float *>(malloc(sizeof(float) * 2 * 4 * 16));
    float *tw3 = static_cast<float *>(malloc(sizeof(float) * 2 * 1024));
    const int WP = TA + TB + 4;
    float *wL = static_cast<float *>(calloc((size_t)64 * WP, sizeof(float)));
    int meta[384];  // startA | startB | rowA | rowB | segIdx[0..63] | segIdx[64..127]
    int st = (tw1 && tw2 && tw3 && wL) ? AFX_OK : AFX_ERR_NOMEM;
    if (st == AFX_OK) {
        const double PI = 3.14159265358979323846;
        for (int k = 0; k < 16; ++k)
            for (int l = 0; l < 64; ++l) {
                const double ang = -2.0 * PI * (double)(k * l) / MC;
                tw1[2 * (k * 64 + l)] = (float)cos(ang);
                tw1[2 * (k * 64 + l) + 1] = (float)sin(ang);
            }
        for (int m = 0; m < 4; ++m)
            for (int j = 0; j < 16; ++j) {
                const double ang = -2.0 * PI * (double)(m * j) / 64.0;
                tw2[2 * (m * 16 + j)] = (float)cos(ang);
                tw2[2 * (m * 16 + j) + 1] = (float)sin(ang);
            }
        for (int k = 0; k < 1024; ++k) {
            const double ang = -2.0 * PI * (double)k / NFFT;
            tw3[2 * k] = (float)(0.5 * cos(ang));
            tw3[2 * k + 1] = (float)(0.5 * sin(ang));
        }
        for (int l = 0; l < 64; ++l) {
            for (int t = 0; t < band->tapsA; ++t) wL[(size_t)l * WP + t] = band->wA[(size_t)t * 64 + l];
            for (int t = 0; t < band->tapsB; ++t) wL[(size_t)l * WP + TA + t] = band->wB[(size_t)t * 64 + l];
        }
        for (int l = 0; l < 64; ++l) {
            meta[l] = band->startA[l];
            meta[64 + l] = band->startB[l];
            meta[128 + l] = band->rowA[l];
            meta[192 + l] = band->rowB[l];
            meta[256 + l] = (int)band->segIdx[l];
            meta[320 + l] = (int)band->segIdx[64 + l];
        }
        st = upload(&p->dWin2, hWindow, sizeof(float) * NFFT, stream);
    }
    if (st == AFX_OK) st = upload(&p->dTw1, tw1, sizeof(float) * 2 * 16 * 64, stream);
    if (st == AFX_OK) st = upload(&p->dTw2, tw2, sizeof(float) * 2 * 4 * 16, stream);
    if (st == AFX_OK) st = upload(&p->dTw3, tw3, sizeof(float) * 2 * 1024, stream);
    if (st == AFX_OK) st = upload(&p->dWLane, wL, sizeof(float) * (size_t)64 * WP, stream);
    if (st == AFX_OK) st = upload(&p->dMeta, meta, sizeof(meta), stream);
    if (st == AFX_OK) st = afxdev_stream_sync(stream);  // host staging buffers are freed below
    if (st == AFX_OK) st = afxk_mel2_create(&p->v2, variant, hWindow, band, stream);
    free(tw1);
    free(tw2);
    free(tw3);
    free(wL);
    if (st != AFX_OK) {
        afxk_melfused_destroy(p);
        return st;
    }
    *plan = p;
    return AFX_OK;
}

extern "C" int afxk_melfused_run(void *plan, const AfxMelFusedArgs *a, void *stream) {
    const Plan *p = static_cast<const Plan *>(plan);
    if (!p) return AFX_ERR_ARG;
    if (p->variant >= 200) return afxk_mel4k_run(plan, a, stream);
    if (p->variant >= 100) return afxk_mel1k_run(plan, a, stream);
    if (a->specMap < 3 && !getenv("AFX_MEL_V1")) return afxk_mel2_run(p->v2, a, stream);
    if (a->cc || a->energy) return AFX_ERR_UNSUPPORTED;
    switch (p->variant) {
        case 0:
            return p->split ? launch<48, 16, true>(p, a, stream) : launch<48, 16>(p, a, stream);
        case 1:
            return p->split ? launch<72, 32, true>(p, a, stream) : launch<72, 32>(p, a, stream);
        default:
            return AFX_ERR_UNSUPPORTED;
    }
}
