// the DEVICE code of audioflux_amd/csrc/hip/afx_melfused2.hip (k_stft_mel_v2: the headline kernel -- STFT -> mel -> log10 ->
// DCT-II in one launch) compiled for the host against tests/emu/hip/hip_runtime.h; exports afxk_mel2_*
#include "hip/hip_runtime.h"
namespace {
alignas(16) unsigned char smem[160 * 1024];
}
static unsigned char *const afx_emu_lds = smem;
static inline void afx_emu_ds() { emu::wave_barrier(); }
#include "../../audioflux_amd/csrc/hip/afx_melfused2.hip"

