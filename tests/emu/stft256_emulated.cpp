// the DEVICE code of audioflux_amd/csrc/hip/afx_stft256.hip (k_stft_256: two frames per 256-point complex wave transform) compiled
// for the host against tests/emu/hip/hip_runtime.h; exports afxk_stft256
#include "hip/hip_runtime.h"
namespace {
alignas(16) unsigned char smem_raw[160 * 1024];
}
#include "../../audioflux_amd/csrc/hip/afx_stft256.hip"
