// afx_melfused.hip -- fused STFT -> filter bank, one 64-lane wave per frame: plan management and dispatch by transform
// size.  n_fft 2048 runs k_stft_mel_v2 (afx_melfused2.hip) for real AND complex results (bftObj_setResultType(0), the
// reference wrapper's default: bft_algorithm.c:457-485; round 1's separate complex-result kernel of this file is gone since
// the complex instantiations of k_stft_mel_v2 keep three waves per SIMD); n_fft 1024 / 4096 live in afx_melfused1k.hip /
// afx_melfused4k2.hip, n_fft 512 in afx_melfused512.hip.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "afx_device.h"
#include "afx_hipcheck.h"

namespace {

struct Plan {
    int variant;  // first field of every size's plan: < 100 this file (n_fft 2048), >= 100 afx_melfused1k, >= 200 afx_melfused4k2, >= 300 afx_melfused512
    int num;
    int split;  // slots hold row segments (AfxBandPlan.split)
    void *v2;   // plan of afx_melfused2.hip
};

struct Variant {
    int tapsA, tapsB;
};
constexpr Variant kVariants[] = {{48, 16}, {72, 32}};
constexpr int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);

}  // namespace

// n_fft = 2048: afx_melfused2.hip
extern "C" int afxk_mel2_create(void **plan, int variant, const float *hWindow, const AfxBandPlan *band, void *stream);
extern "C" int afxk_mel2_run(void *plan, const AfxMelFusedArgs *a, void *stream);
extern "C" void afxk_mel2_destroy(void *plan);

// n_fft = 1024 lives in afx_melfused1k.hip; its plans carry variant numbers >= 100
extern "C" int afxk_mel1k_variant(int tapsA, int tapsB);
extern "C" int afxk_mel1k_create(void **plan, const float *hWindow, const AfxBandPlan *band, void *stream);
extern "C" int afxk_mel1k_run(void *plan, const AfxMelFusedArgs *a, void *stream);
extern "C" void afxk_mel1k_destroy(void *plan);
extern "C" int afxk_mel1k_kind(const void *plan);

// n_fft = 512 lives in afx_melfused512.hip (variant numbers >= 300)
extern "C" int afxk_mel512_variant(int tapsA, int tapsB);
extern "C" int afxk_mel512_create(void **plan, const float *hWindow, const AfxBandPlan *band, void *stream);
extern "C" int afxk_mel512_run(void *plan, const AfxMelFusedArgs *a, void *stream);
extern "C" void afxk_mel512_destroy(void *plan);
extern "C" int afxk_mel512_kind(const void *plan);

// n_fft = 4096 lives in afx_melfused4k2.hip (variant numbers 200 .. 299)
extern "C" int afxk_mel4k_variant(int tapsA, int tapsB);
extern "C" int afxk_mel4k_create(void **plan, const float *hWindow, const AfxBandPlan *band, void *stream);
extern "C" int afxk_mel4k_run(void *plan, const AfxMelFusedArgs *a, void *stream);
extern "C" void afxk_mel4k_destroy(void *plan);
extern "C" int afxk_mel4k_kind(const void *plan);

extern "C" int afxk_melfused_variant(int radix2Exp, int tapsA, int tapsB) {
    if (afxdev_no_fused()) return -1;
    if (radix2Exp == 9) return afxk_mel512_variant(tapsA, tapsB);
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
    if (p->variant >= 300) return afxk_mel512_kind(plan);
    if (p->variant >= 200) return afxk_mel4k_kind(plan);
    if (p->variant >= 100) return afxk_mel1k_kind(plan);
    return p->split ? 2 : 1;
}

extern "C" void afxk_melfused_destroy(void *plan) {
    Plan *p = static_cast<Plan *>(plan);
    if (!p) return;
    if (p->variant >= 300) {
        afxk_mel512_destroy(plan);
        return;
    }
    if (p->variant >= 200) {
        afxk_mel4k_destroy(plan);
        return;
    }
    if (p->variant >= 100) {
        afxk_mel1k_destroy(plan);
        return;
    }
    afxk_mel2_destroy(p->v2);
    free(p);
}

extern "C" int afxk_melfused_create(void **plan, int radix2Exp, const float *hWindow,
                                    const AfxBandPlan *band, void *stream) {
    *plan = nullptr;
    const int variant = afxk_melfused_variant(radix2Exp, band->tapsA, band->tapsB);
    if (variant < 0) return AFX_ERR_UNSUPPORTED;
    if (variant >= 300) return afxk_mel512_create(plan, hWindow, band, stream);
    if (variant >= 200) return afxk_mel4k_create(plan, hWindow, band, stream);
    if (variant >= 100) return afxk_mel1k_create(plan, hWindow, band, stream);
    Plan *p = static_cast<Plan *>(calloc(1, sizeof(Plan)));
    if (!p) return AFX_ERR_NOMEM;
    p->variant = variant;
    p->num = band->num;
    p->split = band->split;
    const int st = afxk_mel2_create(&p->v2, variant, hWindow, band, stream);
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
    if (p->variant >= 300) return afxk_mel512_run(plan, a, stream);
    if (p->variant >= 200) return afxk_mel4k_run(plan, a, stream);
    if (p->variant >= 100) return afxk_mel1k_run(plan, a, stream);
    return afxk_mel2_run(p->v2, a, stream);  // real and complex results
}
