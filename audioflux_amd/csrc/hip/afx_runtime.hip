// afx_runtime.hip -- HIP runtime plumbing behind afx_device.h: device
// selection, device memory, copies, streams, error text.  No compute here.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>

#include "afx_device.h"
#include "afx_hipcheck.h"

namespace {
thread_local char g_err[512] = "";
std::once_flag g_once;
int g_init_status = AFX_ERR_NODEVICE;
char g_init_why[256] = "HIP runtime not initialised";
}  // namespace

extern "C" void afxdev_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    if (getenv("AFX_QUIET") == nullptr) fprintf(stderr, "[audioflux_mi355x] %s\n", g_err);
}

extern "C" const char *afxdev_last_error(void) { return g_err; }

extern "C" int afxdev_ensure(void) {
    std::call_once(g_once, [] {
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess || n <= 0) {
            snprintf(g_init_why, sizeof(g_init_why), "hipGetDeviceCount: %s, %d device(s)", hipGetErrorString(e), n);
            g_init_status = AFX_ERR_NODEVICE;
            return;
        }
        int dev = 0;
        if (const char *s = getenv("AFX_DEVICE")) dev = atoi(s);
        if (dev < 0 || dev >= n) dev = 0;
        e = hipSetDevice(dev);
        if (e != hipSuccess) {
            snprintf(g_init_why, sizeof(g_init_why), "hipSetDevice(%d): %s", dev, hipGetErrorString(e));
            g_init_status = AFX_ERR_NODEVICE;
            return;
        }
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) {
            if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
                // kernels are built for gfx950 only; anything else cannot run them
                snprintf(g_init_why, sizeof(g_init_why), "device %d is %s, this library targets gfx950", dev,
                         prop.gcnArchName);
                g_init_status = AFX_ERR_NODEVICE;
                return;
            }
        }
        g_init_status = AFX_OK;
    });
    if (g_init_status != AFX_OK) {
        afxdev_set_error("no usable MI355X (gfx950) HIP device (%s): this backend has no CPU fallback", g_init_why);
    }
    return g_init_status;
}

extern "C" int afxdev_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int afxdev_set_device(int ordinal) {
    int st = afxdev_ensure();
    if (st != AFX_OK) return st;
    AFX_HIP(hipSetDevice(ordinal));
    return AFX_OK;
}

extern "C" int afxdev_malloc(void **dptr, size_t bytes) {
    *dptr = nullptr;
    int st = afxdev_ensure();
    if (st != AFX_OK) return st;
    if (bytes == 0) bytes = 4;
    hipError_t e = hipMalloc(dptr, bytes);
    if (e != hipSuccess) {
        *dptr = nullptr;
        afxdev_set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return AFX_ERR_NOMEM;
    }
    return AFX_OK;
}

extern "C" void afxdev_free(void *dptr) {
    if (dptr) (void)hipFree(dptr);
}

extern "C" int afxdev_reserve(void **dptr, size_t *capacity, size_t bytes) {
    if (*dptr && *capacity >= bytes) return AFX_OK;
    if (*dptr) {
        (void)hipFree(*dptr);
        *dptr = nullptr;
        *capacity = 0;
    }
    int st = afxdev_malloc(dptr, bytes);
    if (st == AFX_OK) *capacity = bytes;
    return st;
}

extern "C" int afxdev_memset(void *dptr, int value, size_t bytes, void *stream) {
    AFX_HIP(hipMemsetAsync(dptr, value, bytes, (hipStream_t)stream));
    return AFX_OK;
}

extern "C" int afxdev_h2d(void *dst, const void *src, size_t bytes, void *stream) {
    AFX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return AFX_OK;
}

extern "C" int afxdev_d2h(void *dst, const void *src, size_t bytes, void *stream) {
    AFX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return AFX_OK;
}

extern "C" int afxdev_d2d(void *dst, const void *src, size_t bytes, void *stream) {
    AFX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return AFX_OK;
}

extern "C" int afxdev_stream_create(void **stream) {
    *stream = nullptr;
    int st = afxdev_ensure();
    if (st != AFX_OK) return st;
    hipStream_t s;
    AFX_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void *)s;
    return AFX_OK;
}

extern "C" void afxdev_stream_destroy(void *stream) {
    if (stream) (void)hipStreamDestroy((hipStream_t)stream);
}

extern "C" int afxdev_stream_sync(void *stream) {
    AFX_HIP(hipStreamSynchronize((hipStream_t)stream));
    return AFX_OK;
}
