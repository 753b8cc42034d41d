// the DEVICE code of audioflux_amd/csrc/hip/afx_cqt_f16.hip (k_cqt_octave_f16 and its launcher) compiled for the host
// against tests/emu/hip/hip_runtime.h; exports afxk_cqt_octave_f16
#include "hip/hip_runtime.h"
namespace {
alignas(16) unsigned char smem_raw[160 * 1024];
inline void afx_emu_ds() { emu::wave_barrier(); }  // VM_WAIT_ALL of k_cqt_pyramid
}
#include "../../audioflux_amd/csrc/hip/afx_cqt_f16.hip"
