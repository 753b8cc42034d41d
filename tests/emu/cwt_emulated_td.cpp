// the DEVICE code of audioflux_amd/csrc/hip/afx_cwt_td.hip (k_cwt_td<MAXK> and its launcher) compiled for the host
// against tests/emu/hip/hip_runtime.h; exports afxk_cwt_td
#include "hip/hip_runtime.h"
namespace {
alignas(16) unsigned char smem_raw[160 * 1024];
}
#include "../../audioflux_amd/csrc/hip/afx_cwt_td.hip"
