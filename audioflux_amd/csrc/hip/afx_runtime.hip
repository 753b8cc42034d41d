// afx_runtime.hip -- HIP runtime plumbing behind afx_device.h: device
// selection, device memory, copies, streams, error text.  No compute here.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <mutex>

#include "afx_device.h"
#include "afx_hipcheck.h"

namespace {
thread_local char g_err[512] = "";
thread_local int t_err_count = 0;
std::once_flag g_once;
int g_init_status = AFX_ERR_NODEVICE;
char g_init_why[256] = "HIP runtime not initialised";
int g_device_count = 0;
// HIP's current device is PER THREAD: the library's device (AFX_DEVICE, afx_set_device) is a
// process-wide default that every thread adopts on its first call, and every object is tied to
// the device its stream was created on (afxdev_bind_stream at each entry point).
std::atomic<int> g_default_device{0};

bool device_is_gfx950(int dev, char *why, size_t n) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return true;  // cannot tell: let launches decide
    if (strncmp(prop.gcnArchName, "gfx950", 6) == 0) return true;
    // kernels are built for gfx950 only; anything else cannot run them
    snprintf(why, n, "device %d is %s, this library targets gfx950", dev, prop.gcnArchName);
    return false;
}
}  // namespace

// AFX_QUIET: read once per process
static bool quiet() {
    static const bool q = getenv("AFX_QUIET") != nullptr;
    return q;
}

extern "C" void afxdev_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    ++t_err_count;
    if (!quiet()) fprintf(stderr, "[audioflux_mi355x] %s\n", g_err);
}

static void stage_drop_pending(void);  // a failed call never reaches its sync: its staged downloads must not be delivered later
extern "C" void afxdev_report_failure(const char *who, int st) {
    stage_drop_pending();
    ++t_err_count;
    if (!quiet()) fprintf(stderr, "[audioflux_mi355x] %s failed (%d): %s\n", who, st, g_err);
}

extern "C" int afxdev_no_fused(void) {
    static const int v = getenv("AFX_NO_FUSED") != nullptr;
    return v;
}
extern "C" int afxdev_cqt_f32(void) {
    static const int v = getenv("AFX_CQT_F32") != nullptr;
    return v;
}

extern "C" const char *afxdev_last_error(void) { return g_err; }
extern "C" int afxdev_error_count(void) { return t_err_count; }

static int init_status(void) {
    std::call_once(g_once, [] {
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess || n <= 0) {
            snprintf(g_init_why, sizeof(g_init_why), "hipGetDeviceCount: %s, %d device(s)", hipGetErrorString(e), n);
            g_init_status = AFX_ERR_NODEVICE;
            return;
        }
        g_device_count = n;
        int dev = 0;
        if (const char *s = getenv("AFX_DEVICE")) dev = atoi(s);
        if (dev < 0 || dev >= n) dev = 0;
        e = hipSetDevice(dev);
        if (e != hipSuccess) {
            snprintf(g_init_why, sizeof(g_init_why), "hipSetDevice(%d): %s", dev, hipGetErrorString(e));
            g_init_status = AFX_ERR_NODEVICE;
            return;
        }
        if (!device_is_gfx950(dev, g_init_why, sizeof(g_init_why))) {
            g_init_status = AFX_ERR_NODEVICE;
            return;
        }
        g_default_device.store(dev);
        g_init_status = AFX_OK;
    });
    if (g_init_status != AFX_OK) {
        afxdev_set_error("no usable MI355X (gfx950) HIP device (%s): this backend has no CPU fallback", g_init_why);
    }
    return g_init_status;
}

// Called first by every constructor (and by afx_runtime_status): initialise once, then -- per
// thread, outside the call_once -- make the library's default device current, so that a new
// object's stream, plan constants and scratch all land there whichever thread builds it.
extern "C" int afxdev_ensure(void) {
    int st = init_status();
    if (st != AFX_OK) return st;
    const int want = g_default_device.load();
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != want) {
        hipError_t e = hipSetDevice(want);
        if (e != hipSuccess) {
            afxdev_set_error("hipSetDevice(%d): %s", want, hipGetErrorString(e));
            return AFX_ERR_NODEVICE;
        }
    }
    return AFX_OK;
}

extern "C" int afxdev_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int afxdev_set_device(int ordinal) {
    int st = afxdev_ensure();
    if (st != AFX_OK) return st;
    if (ordinal < 0 || ordinal >= g_device_count) {
        afxdev_set_error("afx_set_device(%d): %d device(s) visible", ordinal, g_device_count);
        return AFX_ERR_ARG;
    }
    char why[256];
    if (!device_is_gfx950(ordinal, why, sizeof(why))) {
        afxdev_set_error("%s", why);
        return AFX_ERR_NODEVICE;
    }
    AFX_HIP(hipSetDevice(ordinal));
    g_default_device.store(ordinal);  // objects created from now on (by any thread) live there
    return AFX_OK;
}

extern "C" int afxdev_current_device(void) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    return dev;
}

// Make the device that owns `stream` (an object's own stream, created by its constructor) the
// calling thread's current device.  Called at every compute entry point: allocations, table
// lookups and launches of the call then all refer to the object's device, whatever the thread
// (or torch, or another object) selected before.
extern "C" int afxdev_bind_stream(void *stream) {
    if (!stream) return afxdev_ensure();
    hipDevice_t d = 0;
    hipError_t e = hipStreamGetDevice((hipStream_t)stream, &d);
    if (e != hipSuccess) {
        afxdev_set_error("hipStreamGetDevice: %s", hipGetErrorString(e));
        return AFX_ERR_HIP;
    }
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != (int)d) AFX_HIP(hipSetDevice((int)d));
    return AFX_OK;
}

extern "C" int afxdev_malloc(void **dptr, size_t bytes) {
    *dptr = nullptr;
    int st = init_status();  // on the CURRENT device: the constructor's or the bound object's
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

// ---- small host-pointer copies through pinned staging --------------------------------------------------------
// The legacy entry points (one clip per call: bftObj_bft, spectrogramObj_spectrogram, cqtObj_cqt, ...) hand over
// pageable host arrays, usually ones the runtime has never seen: hipMemcpyAsync then has the operating system pin
// the pages first (227 us per 1.9 MB against 44 us on the wire, profiles/r04_hostabi.txt).  With AFX_STAGING=1 copies of
// at most AFX_STAGE_MAX bytes go through a pinned slab that belongs to the stream: up = host memcpy in growing
// pieces, each piece's DMA running under the memcpy of the next; down = DMA into the slab, the memcpy to the
// caller's array deferred to afxdev_stream_sync (the ONLY synchronisation point of the host code -- every
// afxdev_d2h is followed by one before its function returns).  Slab space is handed out by a bump pointer that the
// same sync resets; what does not fit takes the plain copy.  Larger transfers (the batch entry points) are
// unchanged: for them the runtime's own pin cache measured as good as a hand-made ring.
namespace {
constexpr size_t AFX_STAGE_MAX = 4u << 20;  // slab bytes per direction and stream
constexpr int AFX_STAGE_PEND = 24;
struct Stage {
    std::atomic<void *> stream{nullptr};  // (read outside the table's mutex by the per-thread shortcut)
    int dev = -1;
    unsigned char *up = nullptr, *down = nullptr;
    size_t upUsed = 0, downUsed = 0;
    struct {
        void *dst;
        const unsigned char *src;
        size_t bytes;
    } pend[AFX_STAGE_PEND];
    int nPend = 0;
};
constexpr int AFX_STAGE_SLOTS = 64;
std::mutex g_stageMu;
Stage g_stage[AFX_STAGE_SLOTS];  // (a stream's slabs go back to this table's free entries when the stream is destroyed)
// OFF unless AFX_STAGING is set (read once): in the reference's published protocol the caller's arrays keep their
// addresses, the runtime's pin cache serves them, and the staged path measured 0.233 ms per call against 0.205
// (profiles/r05_legacy_phases.txt); it pays for callers that bring new pages every call.
bool stage_off() {
    static const bool off = getenv("AFX_STAGING") == nullptr;
    return off;
}
// the stage of `stream` (created on first use); nullptr: table full or no pinned memory -> plain copies
thread_local Stage *t_lastStage = nullptr;  // the stage this thread used last
// every stage this thread has staged transfers on (an object with several streams, a nested object): a failing call drops the
// pending downloads of ALL of them, not only of the last one (a later sync would copy into caller arrays that may be gone).
// A stream -- and so its stage -- belongs to ONE thread at a time, like the object that owns it (include/afx_batch.h).
constexpr int AFX_STAGE_TOUCHED = 8;
thread_local Stage *t_touched[AFX_STAGE_TOUCHED] = {};
Stage *stage_note(Stage *s) {
    if (s) {
        for (Stage *&t : t_touched) {
            if (t == s) return s;
            if (!t) {
                t = s;
                return s;
            }
        }
        t_touched[0] = s;  // (more than eight streams per thread: the oldest note is given up)
    }
    return s;
}
Stage *stage_find(void *stream, bool create);
Stage *stage_of(void *stream, bool create) { return stage_note(stage_find(stream, create)); }
Stage *stage_find(void *stream, bool create) {
    Stage *&last = t_lastStage;
    if (last && last->stream.load(std::memory_order_relaxed) == stream) return last;
    std::lock_guard<std::mutex> g(g_stageMu);
    Stage *freeSlot = nullptr;
    for (Stage &s : g_stage) {
        if (s.stream.load(std::memory_order_relaxed) == stream && stream) return last = &s;
        if (!s.stream.load(std::memory_order_relaxed) && !freeSlot) freeSlot = &s;
    }
    if (!create || !freeSlot || !stream) return nullptr;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    if (freeSlot->up && freeSlot->dev != dev) {  // slabs of another device's context: start over
        (void)hipHostFree(freeSlot->up);
        (void)hipHostFree(freeSlot->down);
        freeSlot->up = freeSlot->down = nullptr;
    }
    if (!freeSlot->up) {
        void *u = nullptr, *d = nullptr;
        if (hipHostMalloc(&u, AFX_STAGE_MAX, hipHostMallocDefault) != hipSuccess) return nullptr;
        if (hipHostMalloc(&d, AFX_STAGE_MAX, hipHostMallocDefault) != hipSuccess) {
            (void)hipHostFree(u);
            return nullptr;
        }
        freeSlot->up = static_cast<unsigned char *>(u);
        freeSlot->down = static_cast<unsigned char *>(d);
        freeSlot->dev = dev;
    }
    freeSlot->upUsed = freeSlot->downUsed = 0;
    freeSlot->nPend = 0;
    freeSlot->stream.store(stream, std::memory_order_relaxed);
    return last = freeSlot;
}
}  // namespace

static void stage_drop_pending(void) {
    for (Stage *&t : t_touched) {
        if (t) t->nPend = 0;
        t = nullptr;
    }
    if (t_lastStage) t_lastStage->nPend = 0;
}

extern "C" int afxdev_h2d(void *dst, const void *src, size_t bytes, void *stream) {
    if (bytes == 0) return AFX_OK;
    Stage *s = (bytes <= AFX_STAGE_MAX && !stage_off()) ? stage_of(stream, true) : nullptr;
    if (s && s->upUsed + bytes <= AFX_STAGE_MAX) {
        unsigned char *slab = s->up + s->upUsed;
        s->upUsed += (bytes + 255) & ~(size_t)255;
        // pieces of 128 K, 256 K, 512 K, then 1 M: the first DMA starts early, later ones amortise their submission
        size_t off = 0, piece = 128u << 10;
        while (off < bytes) {
            const size_t n = bytes - off < piece + (piece >> 1) ? bytes - off : piece;
            memcpy(slab + off, static_cast<const unsigned char *>(src) + off, n);
            AFX_HIP(hipMemcpyAsync(static_cast<unsigned char *>(dst) + off, slab + off, n, hipMemcpyHostToDevice,
                                   (hipStream_t)stream));
            off += n;
            if (piece < (1u << 20)) piece <<= 1;
        }
        return AFX_OK;
    }
    AFX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return AFX_OK;
}

extern "C" int afxdev_d2h(void *dst, const void *src, size_t bytes, void *stream) {
    if (bytes == 0) return AFX_OK;
    Stage *s = (bytes <= AFX_STAGE_MAX && !stage_off()) ? stage_of(stream, true) : nullptr;
    if (s && s->nPend < AFX_STAGE_PEND && s->downUsed + bytes <= AFX_STAGE_MAX) {
        unsigned char *slab = s->down + s->downUsed;
        s->downUsed += (bytes + 255) & ~(size_t)255;
        AFX_HIP(hipMemcpyAsync(slab, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
        s->pend[s->nPend].dst = dst;
        s->pend[s->nPend].src = slab;
        s->pend[s->nPend].bytes = bytes;
        ++s->nPend;
        return AFX_OK;
    }
    AFX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return AFX_OK;
}

extern "C" int afxdev_d2d(void *dst, const void *src, size_t bytes, void *stream) {
    AFX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return AFX_OK;
}

extern "C" int afxdev_stream_create(void **stream) {
    *stream = nullptr;
    int st = init_status();
    if (st != AFX_OK) return st;
    hipStream_t s;
    AFX_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void *)s;
    return AFX_OK;
}

// `waiter` waits (on the device) for everything enqueued on `signaler` so far.  The multi-stream schedules (CQT
// decimations under the octave products, the CWT chains) call this two or three times per launch group: the
// events come from a small per-thread, per-device ring instead of a create / destroy pair per call (a wait
// captures the record that precedes it, so an event may be recorded again while an earlier wait is pending).
#define AFX_EVENT_RING 16  /* (a wait is bound to the record that precedes it at enqueue time: re-recording a slot later is harmless) */
namespace {
struct EventRing {
    hipEvent_t ev[AFX_EVENT_RING];
    int dev = -1, next = 0;  // (no destructor: a thread may end after the HIP runtime has shut down)
};
thread_local EventRing t_events;
}  // namespace

extern "C" int afxdev_stream_wait_stream(void *waiter, void *signaler) {
    int dev = -1;
    AFX_HIP(hipGetDevice(&dev));
    EventRing &r = t_events;
    if (r.dev != dev) {
        if (r.dev >= 0)
            for (hipEvent_t e : r.ev) (void)hipEventDestroy(e);
        r.dev = -1;
        for (int i = 0; i < AFX_EVENT_RING; ++i) {
            if (hipEventCreateWithFlags(&r.ev[i], hipEventDisableTiming) != hipSuccess) {
                for (int j = 0; j < i; ++j) (void)hipEventDestroy(r.ev[j]);
                afxdev_set_error("stream wait: hipEventCreate failed");
                return AFX_ERR_HIP;
            }
        }
        r.dev = dev;
        r.next = 0;
    }
    hipEvent_t ev = r.ev[r.next];
    r.next = (r.next + 1) % AFX_EVENT_RING;
    hipError_t e = hipEventRecord(ev, (hipStream_t)signaler);
    if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)waiter, ev, 0);
    if (e != hipSuccess) {
        afxdev_set_error("stream wait: %s", hipGetErrorString(e));
        return AFX_ERR_HIP;
    }
    return AFX_OK;
}

extern "C" void afxdev_stream_destroy(void *stream) {
    if (!stream) return;
    (void)hipStreamDestroy((hipStream_t)stream);
    std::lock_guard<std::mutex> g(g_stageMu);
    for (Stage &s : g_stage)
        if (s.stream.load(std::memory_order_relaxed) == stream) {  // the slabs stay with the table entry for the next stream
            s.nPend = 0;
            s.stream.store(nullptr, std::memory_order_relaxed);
        }
}

extern "C" int afxdev_stream_sync(void *stream) {
    const hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    if (Stage *s = stage_off() ? nullptr : stage_of(stream, false)) {
        // staged downloads reach the caller's arrays here; the slabs are free again
        if (e == hipSuccess)
            for (int i = 0; i < s->nPend; ++i) memcpy(s->pend[i].dst, s->pend[i].src, s->pend[i].bytes);
        s->nPend = 0;
        s->upUsed = s->downUsed = 0;
    }
    AFX_HIP(e);
    return AFX_OK;
}

// ---- clock probe (include/afx_batch.h: afx_clock_probe_start / _stop) ----------------------------------------
// One wave that sleeps and, every few microseconds, writes (shader cycles, wall ticks) since its start to memory:
// s_memtime counts shader-clock cycles, wall_clock64 the constant reference clock, so their ratio over a stretch in
// which the OTHER 255 CUs run the measured kernels is the clock the chip held under that load.  It ends when the
// caller's stop word becomes non-zero (or after maxWallTicks: it can never outlive a forgotten stop by more).
namespace {
__global__ __launch_bounds__(64) void k_clock_probe(unsigned long long *out, const volatile unsigned *stop, unsigned long long maxWallTicks) {
    const unsigned long long s0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
    for (;;) {
        __builtin_amdgcn_s_sleep(127);
        __builtin_amdgcn_s_sleep(127);
        const unsigned long long s1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
        const unsigned st = __hip_atomic_load(const_cast<const unsigned *>(stop), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (threadIdx.x == 0) {
            out[0] = s1 - s0;
            out[1] = w1 - w0;
        }
        if (st != 0u || w1 - w0 >= maxWallTicks) break;
    }
}
}  // namespace

extern "C" int afx_clock_probe_start(void *stream, unsigned long long *dOut2, const unsigned *dStop, double maxSeconds, int *wallClockKHz) {
    if (!dOut2 || !dStop || maxSeconds <= 0.0 || maxSeconds > 60.0) return AFX_ERR_ARG;
    if (afxdev_ensure() != AFX_OK) return AFX_ERR_NODEVICE;
    int dev = 0, khz = 0;
    AFX_HIP(hipGetDevice(&dev));
    AFX_HIP(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev));
    if (khz <= 0) return AFX_ERR_UNSUPPORTED;
    if (wallClockKHz) *wallClockKHz = khz;
    hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, (hipStream_t)stream, dOut2, dStop, (unsigned long long)(maxSeconds * 1e3 * khz));
    AFX_LAUNCH_CHECK("k_clock_probe");
    return AFX_OK;
}
