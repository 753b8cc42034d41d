// afx_melfused4k.hip -- the fused STFT -> spectrum value -> banded filter bank kernel for
// n_fft = 4096 (radix2Exp 12, the default of the reference wrapper), one wave per frame.
//
// A 4096-point real transform is split by decimation in time into the 2048-point real
// transforms E, O of its even and odd samples: X[k] = E[k] + W_4096^k O[k], and because both
// are spectra of real sequences, X[2048-k] = conj(E[k] - W_4096^k O[k]).  E and O each go
// through exactly the pipeline of afx_melfused.hip (2048 real samples packed as 1024 complex,
// 16 x 16 x 4 in registers, last radix-4 folded into the real-input split); E's bins stay in 40
// VGPRs while O is transformed.  A lane loads float4 x[4m..4m+3], m = 64 n1 + lane: (.x,.z) is
// the packed even sample pair z_E[m], (.y,.w) the odd one, and with hop = 1024 the next frame's
// register image is this one moved down by four float4 (only four are fetched per frame).
// 8 waves per CU (2 per SIMD): 218 VGPRs, 8.7 KB of exchange image / power row per wave.
//
// Replaces, per frame, the same reference code as afx_melfused.hip.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "afx_device.h"
#include "afx_hipcheck.h"
#include "afx_pkmath.h"

namespace {

constexpr int NFFT = 4096;
constexpr int MC = 1024;          // complex FFT length of one half (2048 real samples)
constexpr int EX_PITCH = 68;
constexpr int EX_F2 = 16 * EX_PITCH;       // 1088 float2 = 8704 B
constexpr int PROW_F = 2176;               // 2049 bins + zero pad (>= 2049 + 127), 8704 B
constexpr int WAVE_LDS_BYTES = EX_F2 * 8;  // the power row aliases the exchange image
constexpr int WAVES = 8;
constexpr int TAB_WIN_F4 = 1024;  // as two float2 tables: (w[4m], w[4m+2]) | (w[4m+1], w[4m+3])
constexpr int TAB_TW1_F2 = 16 * 64;
constexpr int TAB_TW2_F2 = 64;
constexpr int TAB_TW3_F2 = 1024;  // 0.5 * W_2048^k
constexpr int TAB_TW4_F2 = 1032;  // W_4096^k, k <= 1024
constexpr int TAB_BYTES = TAB_WIN_F4 * 16 + (TAB_TW1_F2 + TAB_TW2_F2 + TAB_TW3_F2 + TAB_TW4_F2) * 8;
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
    int aligned;  // frame starts 16-byte aligned -> float4 loads
    const float4 *win4;
    const float2 *tw1, *tw2, *tw3, *tw4;
    const float *wLane;
    const int *meta;
    int specMap, postPow;
    float normValue;
    float *out, *outIm;
    int num;
};

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
}

// half spectrum values of the conjugate pair (k, 1024-k): H[k] and H[1024-k]
__device__ __forceinline__ void split_pair_h(v2 A, v2 B, v2 w /* 0.5 W_2048^k */, v2 &hk, v2 &hq) {
    const v2 e2 = pk_add_conj(A, B);
    const v2 d = pk_sub_conj(A, B);
    const v2 wo = cmul_mi(d, w);
    hk = e2 * 0.5f + wo;            // H[k]
    const v2 y = e2 * 0.5f - wo;    // conj(H[1024-k])
    hq = v2{y.x, -y.y};
}

// SPLIT: the plan's slots hold row SEGMENTS (afx_bandplan_build_split); lane l adds up rows l and
// l + 64 from the slot results of other lanes (ds_bpermute: no LDS memory, no extra barrier)
template <int TA, int TB, int SHIFT, bool CPLX, bool SPLIT = false>
__global__ __launch_bounds__(WAVES * 64, 2) void k_stft_band_4k(KArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    constexpr int WP = wpitch(TA, TB);
    v2 *tabWinE = reinterpret_cast<v2 *>(smem);  // even-sample pairs, then odd-sample pairs
    v2 *tabWinO = tabWinE + 1024;
    v2 *tabTw1 = reinterpret_cast<v2 *>(smem + TAB_WIN_F4 * 16);
    v2 *tabTw2 = tabTw1 + TAB_TW1_F2;
    v2 *tabTw3 = tabTw2 + TAB_TW2_F2;
    v2 *tabTw4 = tabTw3 + TAB_TW3_F2;
    float *tabW = reinterpret_cast<float *>(smem + TAB_BYTES);
    v2 *ex = reinterpret_cast<v2 *>(smem + TAB_BYTES + 64 * WP * 4 + wave * WAVE_LDS_BYTES);
    float *prow = reinterpret_cast<float *>(ex);

    for (int i = threadIdx.x; i < 2048; i += WAVES * 64) tabWinE[i] = reinterpret_cast<const v2 *>(a.win4)[i];
    for (int i = threadIdx.x; i < TAB_TW1_F2; i += WAVES * 64) tabTw1[i] = reinterpret_cast<const v2 *>(a.tw1)[i];
    for (int i = threadIdx.x; i < TAB_TW3_F2; i += WAVES * 64) tabTw3[i] = reinterpret_cast<const v2 *>(a.tw3)[i];
    for (int i = threadIdx.x; i < TAB_TW4_F2; i += WAVES * 64) tabTw4[i] = reinterpret_cast<const v2 *>(a.tw4)[i];
    for (int i = threadIdx.x; i < 64 * WP; i += WAVES * 64) tabW[i] = a.wLane[i];
    if (threadIdx.x < TAB_TW2_F2) tabTw2[threadIdx.x] = reinterpret_cast<const v2 *>(a.tw2)[threadIdx.x];
    __syncthreads();

    const int k1 = lane >> 2, m2 = lane & 3;
    const int startA = a.meta[lane], startB = a.meta[64 + lane];
    const int rowA = a.meta[128 + lane], rowB = a.meta[192 + lane];
    // split plans: the (up to four) slots whose results make up rows lane and lane + 64
    const unsigned seg0 = SPLIT ? (unsigned)a.meta[256 + lane] : 0u, seg1 = SPLIT ? (unsigned)a.meta[320 + lane] : 0u;
    const float4 *wrow = reinterpret_cast<const float4 *>(tabW + lane * WP);
    const int qm = (256 - lane) & 255;

    const long long gw = (long long)blockIdx.x * WAVES + wave;
    long long f = gw * a.framesPerWave;
    long long fEnd = f + a.framesPerWave;
    if (fEnd > a.totalFrames) fEnd = a.totalFrames;
    if (f >= fEnd) return;
    int clip = (int)(f / a.timeLength);
    int t = (int)(f - (long long)clip * a.timeLength);

    // raw[n1] = x[4m .. 4m+3], m = 64 n1 + lane
    float4 raw[16];
    auto fetch = [&](const float *px, int first) {
        if (a.aligned) {
            const float4 *p4 = reinterpret_cast<const float4 *>(px);
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1)
                if (n1 >= first) raw[n1] = p4[64 * n1 + lane];
        } else {
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1)
                if (n1 >= first) {
                    const int m = 64 * n1 + lane;
                    raw[n1] = make_float4(px[4 * m], px[4 * m + 1], px[4 * m + 2], px[4 * m + 3]);
                }
        }
    };
    fetch(a.x + (long long)clip * a.clipStride + (long long)t * a.hop, 0);

    for (; f < fEnd; ++f) {
        // spectrum of the even samples at the lane's bins: hk[i] = E[k_i], hq[i] = E[1024 - k_i],
        // i = 8 s + j <-> k = lane + 64 s + 256 j; [16], [17]: k = 128, 384 (every lane)
        v2 ek[18], eq[18];
        float imv[CPLX ? 18 : 1][4];  // imaginary parts of the lane's bins (complex results)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            v2 v[16];
            // ---- 1. window: the packed pair of this half -------------------------------
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {
                const v2 w = (half == 0 ? tabWinE : tabWinO)[64 * n1 + lane];
                v[n1] = (half == 0 ? v2{raw[n1].x, raw[n1].z} : v2{raw[n1].y, raw[n1].w}) * w;
            }
            if (half == 1 && f + 1 < fEnd) {
                // ---- 1b. start fetching the next frame (raw is consumed) -------------------
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
            // ---- 2a. radix-16 over n1, twiddle, transpose through LDS -------------------
            {
                v2 t1[16];
#pragma unroll
                for (int k = 1; k < 16; ++k) t1[k] = tabTw1[k * 64 + lane];
                dft16(v);
                ex[lane] = v[0];
#pragma unroll
                for (int k = 1; k < 16; ++k) ex[k * EX_PITCH + lane] = cmul(v[rev4(k)], t1[k]);
                wave_lds_sync();
#pragma unroll
                for (int m1 = 0; m1 < 16; ++m1) v[m1] = ex[k1 * EX_PITCH + 4 * m1 + m2];
                wave_lds_sync();
            }
            // ---- 2b. radix-16 over m1, twiddle W_64^(m2 j1) -> image V[m2][q] ----------
            dft16(v);
            ex[m2 * 260 + k1] = v[0];
#pragma unroll
            for (int j1 = 1; j1 < 16; ++j1) ex[m2 * 260 + k1 + 16 * j1] = cmul(v[rev4(j1)], tabTw2[m2 * 16 + j1]);
            wave_lds_sync();
            // ---- 3. last radix-4 + real-input split of this half ---------------------------
            v2 hk[18], hq[18];
            {
                v2 zin[2][8], w3[2][4];
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int q = lane + 64 * s;
                    const int qp = s == 0 ? qm : 192 - lane;
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        zin[s][m] = ex[260 * m + q];
                        zin[s][4 + m] = ex[260 * m + qp];
                        w3[s][m] = tabTw3[q + 256 * m];
                    }
                }
                v2 zc0 = ex[128], zc1 = ex[260 + 128], zc2 = ex[520 + 128], zc3 = ex[780 + 128];
                const v2 wc0 = tabTw3[128], wc1 = tabTw3[384];
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
                    split_pair_h(za0, b0, w3[s][0], hk[8 * s + 0], hq[8 * s + 0]);
                    split_pair_h(za1, b1, w3[s][1], hk[8 * s + 1], hq[8 * s + 1]);
                    split_pair_h(za2, b2, w3[s][2], hk[8 * s + 2], hq[8 * s + 2]);
                    split_pair_h(za3, b3, w3[s][3], hk[8 * s + 3], hq[8 * s + 3]);
                }
                dft4(zc0, zc1, zc2, zc3);
                split_pair_h(zc0, zc3, wc0, hk[16], hq[16]);
                split_pair_h(zc1, zc2, wc1, hk[17], hq[17]);
            }
            wave_lds_sync();  // image reads done: the next half (or the power row) may overwrite it
            if (half == 0) {
#pragma unroll
                for (int i = 0; i < 18; ++i) {
                    if ((i & 7) >= 4 && i < 16) continue;  // only j < 4 exist
                    ek[i] = hk[i];
                    eq[i] = hq[i];
                }
            } else {
                // ---- combine: X[k] = E + W_4096^k O, X[2048-k] = conj(E - W O); the spectrum
                //      values go straight to the power row (the image is no longer needed) -------
                // complex results (CPLX): S (specMap 3) or S^2 (4); the imaginary parts wait in
                // registers (E's are free now) for the second pass of the filter-bank stage
                auto mapv = [&](v2 c, float &im) {
                    if (!CPLX) return c.x * c.x + c.y * c.y;
                    if (a.specMap == 4) {
                        im = 2.f * (c.x * c.y);
                        return c.x * c.x - c.y * c.y;
                    }
                    im = c.y;
                    return c.x;
                };
#pragma unroll
                for (int i = 0; i < 18; ++i) {
                    if ((i & 7) >= 4 && i < 16) continue;
                    const int k = i < 16 ? lane + 64 * (i >> 3) + 256 * (i & 7) : (i == 16 ? 128 : 384);
                    const v2 tk = cmul(hk[i], tabTw4[k]);         // bins k and 2048 - k
                    const v2 tq = cmul(hq[i], tabTw4[MC - k]);    // bins 1024 - k and 1024 + k
                    const v2 xa = ek[i] + tk, xb = ek[i] - tk, ya = eq[i] + tq, yb = eq[i] - tq;
                    // X[2048-k] = conj(E - T), X[1024+k] = conj(E' - T')
                    float i0 = 0.f, i1 = 0.f, i2 = 0.f, i3 = 0.f;
                    const float r0 = mapv(xa, i0), r1 = mapv(v2{xb.x, -xb.y}, i1);
                    const float r2 = mapv(ya, i2), r3 = mapv(v2{yb.x, -yb.y}, i3);
                    if (CPLX) {
                        imv[CPLX ? i : 0][0] = i0;
                        imv[CPLX ? i : 0][1] = i1;
                        imv[CPLX ? i : 0][2] = i2;
                        imv[CPLX ? i : 0][3] = i3;
                    }
                    if (i < 16 || lane == 0) {
                        prow[k] = r0;
                        prow[2048 - k] = r1;
                        prow[MC - k] = r2;
                        prow[MC + k] = r3;
                    }
                }
            }
        }
#pragma unroll
        for (int pass = 0; pass < (CPLX ? 2 : 1); ++pass) {
        if (CPLX && pass == 1) {
            wave_lds_sync();  // the real pass has read the row
#pragma unroll
            for (int i = 0; i < 18; ++i) {
                if ((i & 7) >= 4 && i < 16) continue;
                const int k = i < 16 ? lane + 64 * (i >> 3) + 256 * (i & 7) : (i == 16 ? 128 : 384);
                if (i < 16 || lane == 0) {
                    prow[k] = imv[CPLX ? i : 0][0];
                    prow[2048 - k] = imv[CPLX ? i : 0][1];
                    prow[MC - k] = imv[CPLX ? i : 0][2];
                    prow[MC + k] = imv[CPLX ? i : 0][3];
                }
            }
        }
        prow[2049 + lane] = 0.f;
        if (lane < PROW_F - 2049 - 64) prow[2049 + 64 + lane] = 0.f;
        wave_lds_sync();
        if (!CPLX && a.specMap) {  // magnitude / norm exponent: one compact pass over the row (rare path)
            for (int k = lane; k < 2049; k += 64) {
                const float p = prow[k];
                prow[k] = a.specMap == 1 ? sqrtf(p) : powf(p, a.normValue);
            }
            wave_lds_sync();
        }

        // ---- 4. banded filter bank (see afx_melfused.hip) ------------------------------------
        float accA, accB;
        {
            const v2 *pa = reinterpret_cast<const v2 *>(prow + startA);
            const v2 *pb = reinterpret_cast<const v2 *>(prow + startB);
            v2 sA = {0.f, 0.f}, sB = {0.f, 0.f};
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
            accA = hsum(sA);
            accB = hsum(sB);
        }
        if (!CPLX && !SPLIT && a.postPow) {
            accA = powf(accA, a.normValue);
            accB = powf(accB, a.normValue);
        }
        float *orow = ((CPLX && pass) ? a.outIm : a.out) + f * a.num;
        if constexpr (SPLIT) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const unsigned u = h ? seg1 : seg0;
                float sum = 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) {  // ascending bin order
                    const unsigned slot = (u >> (8 * c)) & 255u;  // lane | 64 (B slot) | 128 (none)
                    const int src = (int)(slot & 63u) * 4;
                    const float va = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(accA)));
                    const float vb = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(accB)));
                    sum += (slot & 128u) ? 0.f : ((slot & 64u) ? vb : va);
                }
                if (!CPLX && a.postPow) sum = powf(sum, a.normValue);
                if (lane + 64 * h < a.num) orow[lane + 64 * h] = sum;
            }
        } else {
            if (rowA >= 0) orow[rowA] = accA;
            if (rowB >= 0) orow[rowB] = accB;
        }
        }  // pass
        wave_lds_sync();

        if (++t == a.timeLength) {
            t = 0;
            ++clip;
        }
    }
}

struct Plan {
    int variant;  // >= 200: this file
    int num;
    int split;    // slots hold row segments (AfxBandPlan.split)
    void *v2;     // the real-result kernel's plan (afx_melfused4k2.hip)
    float4 *dWin4;
    float2 *dTw1, *dTw2, *dTw3, *dTw4;
    float *dWLane;
    int *dMeta;
};
struct Variant {
    int tapsA, tapsB;
};
constexpr Variant kVariants[] = {{96, 32}, {128, 64}, {176, 8}};
constexpr int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);
static_assert(block_lds_bytes(96, 32) <= 160 * 1024 && block_lds_bytes(128, 64) <= 160 * 1024 &&
                  block_lds_bytes(176, 8) <= 160 * 1024,
              "tables + weights + 8 wave images must fit the 160 KB LDS");

template <int TA, int TB, int SHIFT, bool CPLX, bool SPLIT = false>
int launch_variant(const Plan *p, const AfxMelFusedArgs *a, void *stream) {
    const long long total = (long long)a->batch * a->timeLength;
    if (total <= 0) return AFX_OK;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    long long waves = (long long)cus * WAVES * 2;
    long long fpw = (total + waves - 1) / waves;
    // long runs per wave (register re-use of the overlapping frames) once a round of workgroups is full; a call that
    // cannot fill one round -- the one-clip legacy entry points: 1000 frames -- is spread over all CUs instead
    // (16 frames in sequence per wave were 75 us of a 1000-frame call's 190, profiles/r05_legacy_phases.txt)
    if (fpw < 16) {
        const long long oneRound = (total + (long long)cus * WAVES - 1) / ((long long)cus * WAVES);
        fpw = oneRound < 16 ? oneRound : 16;
    }
    const long long usedWaves = (total + fpw - 1) / fpw;
    const long long blocks = (usedWaves + WAVES - 1) / WAVES;
    KArgs k;
    k.x = a->x;
    k.clipStride = a->clipStride;
    k.totalFrames = total;
    k.timeLength = a->timeLength;
    k.hop = a->hop;
    k.framesPerWave = (int)fpw;
    k.aligned = ((a->clipStride & 3) == 0) && ((a->hop & 3) == 0) && ((reinterpret_cast<uintptr_t>(a->x) & 15) == 0);
    k.win4 = p->dWin4;
    k.tw1 = p->dTw1;
    k.tw2 = p->dTw2;
    k.tw3 = p->dTw3;
    k.tw4 = p->dTw4;
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
        AFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_stft_band_4k<TA, TB, SHIFT, CPLX, SPLIT>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attrSet[attrDev] = true;
    }
    hipLaunchKernelGGL((k_stft_band_4k<TA, TB, SHIFT, CPLX, SPLIT>), dim3((unsigned)blocks), dim3(WAVES * 64), lds,
                       (hipStream_t)stream, k);
    AFX_LAUNCH_CHECK("k_stft_band_4k");
    return AFX_OK;
}

// complex results (specMap 3 / 4); real results run k_stft_band_4k2 (afx_melfused4k2.hip)
template <int TA, int TB, bool SPLIT = false>
int launch(const Plan *p, const AfxMelFusedArgs *a, void *stream) {
    if (a->specMap < 3) return AFX_ERR_ARG;
    if (!a->outIm) return AFX_ERR_ARG;
    return a->hop == 1024 ? launch_variant<TA, TB, 4, true, SPLIT>(p, a, stream)
                          : launch_variant<TA, TB, 0, true, SPLIT>(p, a, stream);
}

template <typename T>
int upload(T **dptr, const void *src, size_t bytes, void *stream) {
    int st = afxdev_malloc(reinterpret_cast<void **>(dptr), bytes);
    if (st != AFX_OK) return st;
    return afxdev_h2d(*dptr, src, bytes, stream);
}

}  // namespace

// afx_melfused4k2.hip
extern "C" int afxk_mel4k2_create(void **plan, int variant, const float *hWindow, const AfxBandPlan *band, void *stream);
extern "C" int afxk_mel4k2_run(void *plan, const AfxMelFusedArgs *a, void *stream);
extern "C" void afxk_mel4k2_destroy(void *plan);

extern "C" int afxk_mel4k_variant(int tapsA, int tapsB) {
    for (int i = 0; i < kNumVariants; ++i)
        if (tapsA <= kVariants[i].tapsA && tapsB <= kVariants[i].tapsB) return 200 + i;
    return -1;
}

extern "C" int afxk_mel4k_kind(const void *plan) {
    const Plan *p = static_cast<const Plan *>(plan);
    return !p ? 0 : (p->split ? 202 : 201);
}

extern "C" void afxk_mel4k_destroy(void *plan) {
    Plan *p = static_cast<Plan *>(plan);
    if (!p) return;
    afxk_mel4k2_destroy(p->v2);
    afxdev_free(p->dWin4);
    afxdev_free(p->dTw1);
    afxdev_free(p->dTw2);
    afxdev_free(p->dTw3);
    afxdev_free(p->dTw4);
    afxdev_free(p->dWLane);
    afxdev_free(p->dMeta);
    free(p);
}

extern "C" int afxk_mel4k_create(void **plan, const float *hWindow, const AfxBandPlan *band, void *stream) {
    *plan = nullptr;
    const int variant = afxk_mel4k_variant(band->tapsA, band->tapsB);
    if (variant < 0) return AFX_ERR_UNSUPPORTED;
    const int TA = kVariants[variant - 200].tapsA, TB = kVariants[variant - 200].tapsB;
    Plan *p = static_cast<Plan *>(calloc(1, sizeof(Plan)));
    if (!p) return AFX_ERR_NOMEM;
    p->variant = variant;
    p->num = band->num;
    p->split = band->split;
    const int WP = TA + TB + 4;
    float *tw1 = static_cast<float *>(malloc(sizeof(float) * 2 * TAB_TW1_F2));
    float *tw2 = static_cast<float *>(malloc(sizeof(float) * 2 * TAB_TW2_F2));
    float *tw3 = static_cast<float *>(malloc(sizeof(float) * 2 * TAB_TW3_F2));
    float *tw4 = static_cast<float *>(calloc(2 * TAB_TW4_F2, sizeof(float)));
    float *wL = static_cast<float *>(calloc((size_t)64 * WP, sizeof(float)));
    int meta[384];  // startA | startB | rowA | rowB | segIdx[0..63] | segIdx[64..127]
    int st = (tw1 && tw2 && tw3 && tw4 && wL) ? AFX_OK : AFX_ERR_NOMEM;
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
            const double ang = -2.0 * PI * (double)k / 2048.0;
            tw3[2 * k] = (float)(0.5 * cos(ang));
            tw3[2 * k + 1] = (float)(0.5 * sin(ang));
        }
        for (int k = 0; k <= 1024; ++k) {
            const double ang = -2.0 * PI * (double)k / NFFT;
            tw4[2 * k] = (float)cos(ang);
            tw4[2 * k + 1] = (float)sin(ang);
        }
        for (int l = 0; l < 64; ++l) {
            for (int t = 0; t < band->tapsA; ++t) wL[(size_t)l * WP + t] = band->wA[(size_t)t * 64 + l];
            for (int t = 0; t < band->tapsB; ++t) wL[(size_t)l * WP + TA + t] = band->wB[(size_t)t * 64 + l];
            meta[l] = band->startA[l];
            meta[64 + l] = band->startB[l];
            meta[128 + l] = band->rowA[l];
            meta[192 + l] = band->rowB[l];
            meta[256 + l] = (int)band->segIdx[l];
            meta[320 + l] = (int)band->segIdx[64 + l];
        }
        float *w2 = static_cast<float *>(malloc(sizeof(float) * NFFT));
        if (!w2) st = AFX_ERR_NOMEM;
        if (st == AFX_OK) {
            for (int m = 0; m < 1024; ++m) {
                w2[2 * m] = hWindow[4 * m];
                w2[2 * m + 1] = hWindow[4 * m + 2];
                w2[2048 + 2 * m] = hWindow[4 * m + 1];
                w2[2048 + 2 * m + 1] = hWindow[4 * m + 3];
            }
            st = upload(&p->dWin4, w2, sizeof(float) * NFFT, stream);
            if (st == AFX_OK) st = afxdev_stream_sync(stream);
        }
        free(w2);
    }
    if (st == AFX_OK) st = upload(&p->dTw1, tw1, sizeof(float) * 2 * TAB_TW1_F2, stream);
    if (st == AFX_OK) st = upload(&p->dTw2, tw2, sizeof(float) * 2 * TAB_TW2_F2, stream);
    if (st == AFX_OK) st = upload(&p->dTw3, tw3, sizeof(float) * 2 * TAB_TW3_F2, stream);
    if (st == AFX_OK) st = upload(&p->dTw4, tw4, sizeof(float) * 2 * TAB_TW4_F2, stream);
    if (st == AFX_OK) st = upload(&p->dWLane, wL, sizeof(float) * (size_t)64 * WP, stream);
    if (st == AFX_OK) st = upload(&p->dMeta, meta, sizeof(meta), stream);
    if (st == AFX_OK) st = afxdev_stream_sync(stream);
    if (st == AFX_OK) st = afxk_mel4k2_create(&p->v2, variant - 200, hWindow, band, stream);
    free(tw1);
    free(tw2);
    free(tw3);
    free(tw4);
    free(wL);
    if (st != AFX_OK) {
        afxk_mel4k_destroy(p);
        return st;
    }
    *plan = p;
    return AFX_OK;
}

extern "C" int afxk_mel4k_run(void *plan, const AfxMelFusedArgs *a, void *stream) {
    if (a->cc || a->energy) return AFX_ERR_UNSUPPORTED;  // fusions exist at n_fft 2048 only
    const Plan *p = static_cast<const Plan *>(plan);
    if (!p) return AFX_ERR_ARG;
    if (a->specMap < 3) return afxk_mel4k2_run(p->v2, a, stream);  // real results: afx_melfused4k2.hip
    switch (p->variant) {
        case 200: return p->split ? launch<96, 32, true>(p, a, stream) : launch<96, 32>(p, a, stream);
        case 201: return p->split ? launch<128, 64, true>(p, a, stream) : launch<128, 64>(p, a, stream);
        case 202: return p->split ? launch<176, 8, true>(p, a, stream) : launch<176, 8>(p, a, stream);
        default: return AFX_ERR_UNSUPPORTED;
    }
}
