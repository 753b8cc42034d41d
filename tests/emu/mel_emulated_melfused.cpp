// the DEVICE code of audioflux_amd/csrc/hip/afx_melfused.hip (the afxk_melfused_* dispatcher and the n_fft 2048 kernel for complex results) compiled for the host against
// tests/emu/hip/hip_runtime.h
#include "hip/hip_runtime.h"
namespace {
alignas(16) unsigned char smem[160 * 1024];
}
#include "../../audioflux_amd/csrc/hip/afx_melfused.hip"
