// the DEVICE code of audioflux_amd/csrc/hip/afx_melfused512.hip (k_stft_band_512 (n_fft 512)) compiled for the host against
// tests/emu/hip/hip_runtime.h
#include "hip/hip_runtime.h"
namespace {
alignas(16) unsigned char smem[160 * 1024];
}
static unsigned char *const afx_emu_lds = smem;
static inline void afx_emu_ds() { emu::wave_barrier(); }
#include "../../audioflux_amd/csrc/hip/afx_melfused512.hip"
