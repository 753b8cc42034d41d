// the DEVICE code of audioflux_amd/csrc/hip/afx_gemm_bf16.hip (k_gemm_nt128_bf16x3 and its launcher) compiled for the
// host against tests/emu/hip/hip_runtime.h; exports afxk_gemm_nt128_bf16
#include "hip/hip_runtime.h"
namespace {
alignas(16) unsigned char smem[160 * 1024];
}
#include "../../audioflux_amd/csrc/hip/afx_gemm_bf16.hip"
