// the DEVICE code of audioflux_amd/csrc/hip/afx_melfused4k.hip (k_stft_band_4k (n_fft 4096)) compiled for the host against
// tests/emu/hip/hip_runtime.h
#include "hip/hip_runtime.h"
namespace {
alignas(16) unsigned char smem[160 * 1024];
}
#include "../../audioflux_amd/csrc/hip/afx_melfused4k.hip"
