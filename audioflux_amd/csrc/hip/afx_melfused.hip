// afx_melfused.hip -- fused STFT -> filter bank, one 64-lane wave per 2048-sample frame: plan
// management for n_fft 2048 and the COMPLEX-RESULT kernel (bftObj_setResultType(0), the
// reference wrapper's default: bft_algorithm.c:457-485).  Real results -- the headline path --
// run k_stft_mel_v2 (afx_melfused2.hip); n_fft 1024 / 4096 live in afx_melfused1k/4k.hip.
//
// Per frame, all inside one wave (no workgroup barriers; waves are independent):
//   1. 16 coalesced float2 loads per lane of the hop-overlapped frame, times the window
//   2. 1024-point complex FFT (16 x 16 x 4): radix-16 in registers, twiddle W_1024^(lane*k1),
//      transpose through LDS (pitch 68 float2), radix-16, twiddle W_64, second image V[m2][q]
//   3. last radix-4 + real-input split fused: X[k] = E + W_2048^k O, X[1024-k] = conj(E - W O);
//      S (data type MAG) or S^2 (POWER, __mcsquare flux_complex.c:469-503) of both bins
//   4. the bank is real, so real and imaginary parts go through the banded filter-bank stage
//      one after the other (two passes over the wave's spectrum row in LDS)
//   5. two dword stores per lane and plane
// Index algebra: tools/proto_fft1024.py.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>

#include "afx_device.h"
#include "afx_hipcheck.h"
#include "afx_pkmath.h"


#ifdef AFX_NO_PRIO  // measurement builds only
#define AFX_TRANSFORM_PRIO(p) ((void)0)
#else
#define AFX_TRANSFORM_PRIO(p) __builtin_amdgcn_s_setprio(p)
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

__device__ __forceinline__ void wave_lds_sync() {
    // Orders this wave's LDS stores before its later LDS loads of other lanes' data.  DS
    // operations of one wave execute in issue order; lgkmcnt(0) drains them and the wave
    // barrier pins the compiler.  Deliberately NOT a fence: a wavefront-scope fence also
    // emits vmcnt(0), which would drain the next frame's prefetch and the previous frame's
    // stores at every exchange.
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0), vmcnt/expcnt untouched
    __builtin_amdgcn_wave_barrier();
}

// the spectrum value itself (sq = false: data type MAG) or its complex square (sq = true: POWER,
// __mcsquare, flux_complex.c:469-503) of bins k and 1024-k from A = Z[k], B = Z[1024-k]; the bank
// is real, so real and imaginary parts go through it separately
__device__ __forceinline__ void split_pair_c(v2 A, v2 B, v2 w /* 0.5 W_2048^k */, bool sq, float &kr, float &ki,
                                             float &qr, float &qi) {
    const v2 e2 = pk_add_conj(A, B);   // 2 E
    const v2 d = pk_sub_conj(A, B);    // 2 i O  ->  2 O = -i d
    const v2 wo = cmul_mi(d, w);       // W O   (w carries the 1/2)
    const v2 x = e2 * 0.5f + wo;       // X[k]
    const v2 y = e2 * 0.5f - wo;       // conj(X[1024-k])
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

// SHIFT: consecutive frames of a clip overlap; with hop = 128*SHIFT samples the next frame's
// register image is the current one moved down by SHIFT registers, so only SHIFT new float2 per
// lane are fetched per frame (SHIFT = 0: every frame is fetched whole)
// SPLIT: the plan's slots hold row SEGMENTS (afx_bandplan_build_split): the slot results go
// through 129 floats of the (by then dead) exchange buffer and lane l adds up rows l and l + 64
template <int TA, int TB, int SHIFT, bool SPLIT>
__global__ __launch_bounds__(WAVES * 64, 3) void k_stft_mel_cplx(KArgs a) {
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
        for (int i = threadIdx.x; i < TAB_WIN_F2; i += WAVES * 64) tabWin[i] = reinterpret_cast<const v2 *>(a.win2)[i];
        for (int i = threadIdx.x; i < TAB_TW1_F2; i += WAVES * 64) tabTw1[i] = gTw1[i];
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
    const bool sq = a.specMap == 4;

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

    for (; f < fEnd; ++f) {
        v2 v[16];
        // ---- 1. window (samples were fetched during the previous frame) ---------------
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) v[n1] = raw[n1] * tabWin[64 * n1 + lane];
        // ---- 1b. start fetching the next frame: in flight under the whole transform ---
        if (f + 1 < fEnd) {
            int tn = t + 1, cn = clip;
            if (tn == a.timeLength) {
                tn = 0;
                ++cn;
            }
            const float *pn = a.x + (long long)cn * a.clipStride + (long long)tn * a.hop;
            bool whole = true;
            if constexpr (SHIFT > 0) {
                if (tn != 0) {
                    shift_rows_inplace<SHIFT>(raw);  // in place (afx_asm.h: the compiler's own form keeps two images of the frame)
                    fetch(pn, 16 - SHIFT);
                    whole = false;
                }
            }
            if (whole) fetch(pn, 0);
        }

        AFX_TRANSFORM_PRIO(1);  // (afx_melfused2.hip: the transform's phases above the window / band / store phases of the SIMD's other waves)
        // ---- 2a. radix-16 over n1, twiddle, transpose through LDS ---------------
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

        // ---- 2b. radix-16 over m1, twiddle W_64^(m2*j1) -> image V[m2][q = k1 + 16 j1] ----
        dft16(v);
        ex[m2 * 260 + k1] = v[0];
#pragma unroll
        for (int j1 = 1; j1 < 16; ++j1) ex[m2 * 260 + k1 + 16 * j1] = cmul(v[rev4(j1)], tabTw2[m2 * 16 + j1]);
        wave_lds_sync();

        // ---- 3. last radix-4 + real-input split -> spectrum values in registers -------
        float pk[20], pq[20], pkI[20], pqI[20];
        v2 zin[2][8], w3[2][4];
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
        v2 zc0 = ex[128], zc1 = ex[260 + 128], zc2 = ex[520 + 128], zc3 = ex[780 + 128];
        const v2 wc0 = tabTw3[128], wc1 = tabTw3[384];
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
            split_pair_c(za0, b0, w3[s][0], sq, pk[8 * s + 0], pkI[8 * s + 0], pq[8 * s + 0], pqI[8 * s + 0]);
            split_pair_c(za1, b1, w3[s][1], sq, pk[8 * s + 1], pkI[8 * s + 1], pq[8 * s + 1], pqI[8 * s + 1]);
            split_pair_c(za2, b2, w3[s][2], sq, pk[8 * s + 2], pkI[8 * s + 2], pq[8 * s + 2], pqI[8 * s + 2]);
            split_pair_c(za3, b3, w3[s][3], sq, pk[8 * s + 3], pkI[8 * s + 3], pq[8 * s + 3], pqI[8 * s + 3]);
        }
        // base 128 mirrors itself: bins 128, 384 and their partners 896, 640 (every lane computes
        // them, lane 0 stores them)
        dft4(zc0, zc1, zc2, zc3);
        split_pair_c(zc0, zc3, wc0, sq, pk[16], pkI[16], pq[16], pqI[16]);
        split_pair_c(zc1, zc2, wc1, sq, pk[17], pkI[17], pq[17], pqI[17]);
        wave_lds_sync();  // every lane has its bins in registers; ex becomes the spectrum row
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1) wave_lds_sync();  // the real pass has read the row; now the imaginary parts
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = lane + 64 * s + 256 * j;
                    prow[k] = pass ? pkI[8 * s + j] : pk[8 * s + j];
                    prow[MC - k] = pass ? pqI[8 * s + j] : pq[8 * s + j];
                }
            }
            if (lane == 0) {
                prow[128] = pass ? pkI[16] : pk[16];
                prow[896] = pass ? pqI[16] : pq[16];
                prow[384] = pass ? pkI[17] : pk[17];
                prow[640] = pass ? pqI[17] : pq[17];
            }
            // zero pad behind bin 1024: the fixed-length band loops read it with zero weights
            prow[1025 + lane] = 0.f;
            if (lane < PROW_F - 1025 - 64) prow[1025 + 64 + lane] = 0.f;
            wave_lds_sync();

            AFX_TRANSFORM_PRIO(0);
            // ---- 4. banded filter bank: weights by ds_read_b128, the row by ds_read_b64 (starts are
            //         even; conflict-free by the plan's bank-aware lane assignment), operands in
            //         blocks of 4 quads so that one LDS round trip is paid per block -------------
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
                accA = sA.x + sA.y;
                accB = sB.x + sB.y;
            }
            // ---- 5. store ---------------------------------------------------------------
            float *orow = (pass ? a.outIm : a.out) + f * a.num;
            if constexpr (SPLIT) {
                // slot results -> LDS (behind the row: that part of the exchange buffer is dead
                // since stage 3), then every row is the sum of its segments in ascending bins
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
                    if (lane + 64 * h < a.num) orow[lane + 64 * h] = sum;
                }
            } else {
                if (rowA >= 0) orow[rowA] = accA;
                if (rowB >= 0) orow[rowB] = accB;
            }
        }  // pass
        wave_lds_sync();  // the next frame overwrites ex / the row

        if (++t == a.timeLength) {
            t = 0;
            ++clip;
        }
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

template <int TA, int TB, int SHIFT, bool SPLIT>
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
        AFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_stft_mel_cplx<TA, TB, SHIFT, SPLIT>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attrSet[attrDev] = true;
    }
    hipLaunchKernelGGL((k_stft_mel_cplx<TA, TB, SHIFT, SPLIT>), dim3((unsigned)blocks), dim3(WAVES * 64), lds,
                       (hipStream_t)stream, k);
    AFX_LAUNCH_CHECK("k_stft_mel_cplx");
    return AFX_OK;
}

// complex result: S (specMap 3) or S^2 (4), real and imaginary planes
template <int TA, int TB, bool SPLIT = false>
int launch(const Plan *p, const AfxMelFusedArgs *a, void *stream) {
    if (!a->outIm) return AFX_ERR_ARG;
    return a->hop == 512 ? launch_variant<TA, TB, 4, SPLIT>(p, a, stream)  // hop = 128 * SHIFT
                         : launch_variant<TA, TB, 0, SPLIT>(p, a, stream);
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

// n_fft = 4096 lives in afx_melfused4k2.hip (variant numbers >= 200)
extern "C" int afxk_mel4k_variant(int tapsA, int tapsB);
extern "C" int afxk_mel4k_create(void **plan, const float *hWindow, const AfxBandPlan *band, void *stream);
extern "C" int afxk_mel4k_run(void *plan, const AfxMelFusedArgs *a, void *stream);
extern "C" void afxk_mel4k_destroy(void *plan);
extern "C" int afxk_mel4k_kind(const void *plan);

extern "C" int afxk_melfused_variant(int radix2Exp, int tapsA, int tapsB) {
    if (afxdev_no_fused()) return -1;
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
    if (a->specMap < 3) return afxk_mel2_run(p->v2, a, stream);  // real results: afx_melfused2.hip
    // complex results: afx_melfused2.hip too (round 5), except the whole-row plan of the wide variant, whose instantiation spills there
    if (!(a->cc || a->energy) && (p->variant == 0 || p->split)) return afxk_mel2_run(p->v2, a, stream);
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
