/* afx_api.c -- additive library-level entry points of include/afx_batch.h
 * that are not tied to one object type. */
#include <stdlib.h>

#include "afx_batch.h"
#include "afx_device.h"
#include "afx_objects.h"

int afx_runtime_status(void) { return afxdev_ensure(); }
const char *afx_last_error(void) { return afxdev_last_error(); }
int afx_error_count(void) { return afxdev_error_count(); }
int afx_device_count(void) { return afxdev_device_count(); }
int afx_set_device(int ordinal) { return afxdev_set_device(ordinal); }
const char *afx_version(void) { return "audioflux_mi355x 0.1.0 gfx950"; }

int bftObj_fusedPlanKind(BFTObj bft) { return bft ? afxk_melfused_kind(bft->fast) : 0; }

/* calls of afx_bftXxccBatchDevice on this thread that ran as ONE launch (diagnostic: tests assert the route) */
static __thread long long t_oneLaunchCalls = 0;
long long afx_bftXxccOneLaunchCount(void) { return t_oneLaunchCalls; }

int afx_bftXxccBatchDevice(BFTObj bft, XXCCObj xxcc, const float *dData, int batch,
                           int dataLength, long long clipStride, int ccNum,
                           CepstralRectifyType *rectifyType, float *dMel, float *dCc,
                           void *hipStream) {
    if (!bft || !xxcc || !dData || !dCc) return AFX_ERR_ARG;
    AFX_ENTER(bft);
    if (!bft->resultType || xxcc->num != bft->num || ccNum < 1 || ccNum > xxcc->num) {
        afxdev_set_error("afx_bftXxccBatchDevice: needs a real-result BFT and an XXCC of the same num");
        return AFX_ERR_ARG;
    }
    void *stream = hipStream;
    const int T = bftObj_calTimeLength(bft, dataLength);
    if (T <= 0 || batch <= 0) return AFX_OK;
    const long long frames = (long long)batch * T;

    /* one launch (STFT -> bank -> rectify -> DCT-II inside the size's fused kernel) when the plan supports it */
    int used = 0;
    int st = afx_bft_try_fast_cc(bft, xxcc, dData, batch, dataLength, clipStride, ccNum,
                                 rectifyType, dMel, dCc, stream, &used);
    if (st == AFX_OK && used) t_oneLaunchCalls++;
    if (st != AFX_OK || used) return st;

    /* otherwise: bank output (kept in scratch when the caller does not want it) + DCT GEMM */
    float *mel = dMel;
    if (!mel) {
        st = afxdev_reserve((void **)&bft->dOut, &bft->capOut,
                            sizeof(float) * (size_t)frames * bft->num);
        if (st != AFX_OK) return st;
        mel = bft->dOut;
    }
    st = afx_bft_run_device(bft, dData, batch, dataLength, clipStride, mel, NULL, NULL, stream);
    if (st != AFX_OK) return st;
    return xxccObj_xxccDevice(xxcc, mel, frames, ccNum, rectifyType, dCc, stream);
}
