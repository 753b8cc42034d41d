// the DEVICE code of audioflux_amd/csrc/hip/afx_istft.hip (the one-launch inverse STFT: k_istft_w4096 / _w2048 / _wsmall / _w256, and the
// two size-generic kernels) compiled for the host against tests/emu/hip/hip_runtime.h; exports afxk_istft_fused / afxk_istft
#include "hip/hip_runtime.h"
namespace {
alignas(16) unsigned char smem_raw[160 * 1024];
}
#include "../../audioflux_amd/csrc/hip/afx_istft.hip"

// the wave kernels' twiddle tables (afx_stft.hip's wave_tables() on the device): afxw tables + W_4096^k, k <= 1024
extern "C" const void *afxk_wave_tables(void) {
    static float *tab = nullptr;
    if (!tab) {
        tab = static_cast<float *>(calloc(2 * ((size_t)afxw::TAB_F2 + 1032), sizeof(float)));
        afxw::fill_tables(tab);
        const double PI = 3.14159265358979323846;
        for (int k = 0; k <= 1024; ++k) {
            tab[2 * (afxw::TAB_F2 + k)] = (float)cos(-2.0 * PI * (double)k / 4096.0);
            tab[2 * (afxw::TAB_F2 + k) + 1] = (float)sin(-2.0 * PI * (double)k / 4096.0);
        }
    }
    return tab;
}
