// afx_hipcheck.h -- error-check macros shared by the .hip translation units
#ifndef AFX_HIPCHECK_H
#define AFX_HIPCHECK_H

#include <hip/hip_runtime.h>

#include "afx_device.h"

#define AFX_HIP(call)                                                                      \
    do {                                                                                   \
        hipError_t _e = (call);                                                            \
        if (_e != hipSuccess) {                                                            \
            afxdev_set_error("%s failed at %s:%d: %s", #call, __FILE__, __LINE__,          \
                             hipGetErrorString(_e));                                       \
            return AFX_ERR_HIP;                                                            \
        }                                                                                  \
    } while (0)

// after a kernel launch
#define AFX_LAUNCH_CHECK(name)                                                             \
    do {                                                                                   \
        hipError_t _e = hipGetLastError();                                                 \
        if (_e != hipSuccess) {                                                            \
            afxdev_set_error("launch of %s failed: %s", name, hipGetErrorString(_e));      \
            return AFX_ERR_HIP;                                                            \
        }                                                                                  \
    } while (0)

#endif
