// placeholder: fused-kernel hooks (filled in by the register-resident kernel)
#include "afx_device.h"
#include "afx_objects.h"

extern "C" int afx_bft_plan_fast(struct OpaqueBFT *, const float *, const float *) { return AFX_OK; }
extern "C" int afx_bft_try_fast(struct OpaqueBFT *, const float *, int, int, long long, float *,
                                float *, void *, int *used) {
    *used = 0;
    return AFX_OK;
}
extern "C" int afx_bft_try_fast_cc(struct OpaqueBFT *, struct OpaqueXXCC *, const float *, int, int,
                                   long long, int, CepstralRectifyType *, float *, float *, void *,
                                   int *used) {
    *used = 0;
    return AFX_OK;
}
extern "C" void afx_bft_free_fast(struct OpaqueBFT *) {}
