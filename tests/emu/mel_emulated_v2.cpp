// the DEVICE code of audioflux_amd/csrc/hip/afx_melfused2.hip (k_stft_mel_v2: the headline kernel -- STFT -> mel -> log10 ->
// DCT-II in one launch) compiled for the host against tests/emu/hip/hip_runtime.h, with a minimal afxk_melfused_*
// dispatcher for n_fft 2048 (the product's own dispatcher, afx_melfused.hip, also owns the 1k / 4k / complex kernels)
#include "hip/hip_runtime.h"
namespace {
alignas(16) unsigned char smem[160 * 1024];
}
static unsigned char *const afx_emu_lds = smem;
static inline void afx_emu_ds() { emu::wave_barrier(); }
#include "../../audioflux_amd/csrc/hip/afx_melfused2.hip"

namespace {
struct EmuPlan {
    void *v2;
    int split;
};
const int kTaps[2][2] = {{48, 16}, {72, 32}};
}  // namespace

extern "C" int afxk_melfused_variant(int radix2Exp, int tapsA, int tapsB) {
    if (radix2Exp != 11 || getenv("AFX_NO_FUSED")) return -1;
    for (int i = 0; i < 2; ++i)
        if (tapsA <= kTaps[i][0] && tapsB <= kTaps[i][1]) return i;
    return -1;
}
extern "C" int afxk_melfused_create(void **plan, int radix2Exp, const float *hWindow, const AfxBandPlan *band, void *stream) {
    *plan = nullptr;
    const int variant = afxk_melfused_variant(radix2Exp, band->tapsA, band->tapsB);
    if (variant < 0) return AFX_ERR_UNSUPPORTED;
    EmuPlan *p = static_cast<EmuPlan *>(calloc(1, sizeof(EmuPlan)));
    if (!p) return AFX_ERR_NOMEM;
    p->split = band->split;
    const int st = afxk_mel2_create(&p->v2, variant, hWindow, band, stream);
    if (st != AFX_OK) {
        free(p);
        return st;
    }
    *plan = p;
    return AFX_OK;
}
extern "C" int afxk_melfused_run(void *plan, const AfxMelFusedArgs *a, void *stream) {
    const EmuPlan *p = static_cast<const EmuPlan *>(plan);
    if (!p) return AFX_ERR_ARG;
    if (a->specMap >= 3) return AFX_ERR_UNSUPPORTED;  // complex results: afx_melfused.hip's kernel, not emulated
    return afxk_mel2_run(p->v2, a, stream);
}
extern "C" void afxk_melfused_destroy(void *plan) {
    EmuPlan *p = static_cast<EmuPlan *>(plan);
    if (!p) return;
    afxk_mel2_destroy(p->v2);
    free(p);
}
extern "C" int afxk_melfused_kind(const void *plan) { return plan ? (static_cast<const EmuPlan *>(plan)->split ? 2 : 1) : 0; }
