// A stand-in HIP runtime for the CPU-only launch audit (tests/test_hoststub.py::test_launch_audit): the product's
// REAL host side -- the C objects AND the launchers of audioflux_amd/csrc/hip/*.hip, compiled --cuda-host-only with
// UBSan + clang's integer checks -- linked against this file instead of libamdhip64.  One "gfx950" device with 256
// CUs and 288 GB; allocations are address ranges without memory behind them, copies and memsets do nothing, and
// hipLaunchKernel CHECKS every launch configuration instead of launching:
//   * block size 1..1024 threads, grid dimensions within the HIP limits (x < 2^31, y / z < 65536, threads per
//     dimension < 2^32 -- that one is rejected with hipErrorInvalidConfiguration, as the real runtime does, and
//     must come back from the launcher as a status), nothing zero;
//   * dynamic LDS <= 64 KB, or <= what hipFuncSetAttribute(MaxDynamicSharedMemorySize) raised it to for that kernel,
//     and never above the 160 KB of a gfx950 CU.
// It also keeps the HAPPENS-BEFORE relation of the streams (launch order per stream, hipEventRecord /
// hipStreamWaitEvent edges, host synchronisation) and, for the CQT kernels -- whose argument lists it decodes -- the
// address ranges every launch reads and writes: two launches that touch overlapping ranges, at least one writing,
// without an ordering between them are reported as "FAKEHIP RACE".  That checks the multi-stream schedules of
// afx_cqt.c (decimations on a side stream under the octave products) and afx_cwt.c (three chains) on the CPU;
// A decoded range that starts inside one of the library's own allocations and ends past it is a "FAKEHIP OVERRUN".
// FAKEHIP_ORDER=1 switches this on; FAKEHIP_DROP_WAIT=<k> ignores the k-th hipStreamWaitEvent (the detector's own
// test).
// With it the launch arithmetic of every kernel runs at the BASELINE sizes and far beyond them without a GPU.  It says nothing about what the kernels compute.  Test infrastructure, never linked into the product.
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "afx_device.h"

namespace {
// constructed on first use: clang's registration hooks run from static constructors of the kernel translation units
typedef std::map<const void *, unsigned long long> VClock;  // stream -> operations of it that are known to be complete
struct Access {
    uintptr_t lo, hi;
    bool write;
    const void *stream;
    unsigned long long stamp;
    std::string what;
};
struct State {
    std::mutex mu;
    std::map<const void *, std::string> names;
    std::map<const void *, int> maxDynLds;
    std::map<std::string, unsigned long long> launches;
    std::map<const void *, VClock> streams, events;
    VClock host;
    std::vector<Access> log;
    std::map<uintptr_t, size_t> allocs;  // device allocations of the library: base -> bytes
    std::map<uintptr_t, std::vector<char>> uploads;  // small host-to-device copies, kept: index lists the decoders need
    std::map<std::string, int> raceKinds;
    int waits = 0, races = 0;
};
State &st() {
    static State *s = new State;
    return *s;
}
#define g_mu st().mu
#define g_names st().names
#define g_maxDynLds st().maxDynLds
#define g_launches st().launches
unsigned long long g_next = 1ull << 44;
unsigned long long g_allocated = 0;
int g_violations = 0;
thread_local dim3 t_grid, t_block;
thread_local size_t t_shmem;
thread_local hipStream_t t_stream;
thread_local hipError_t t_lastError = hipSuccess;
int g_rejected = 0;
constexpr unsigned long long DEVICE_BYTES = 288ull << 30;

void *dry_alloc(size_t bytes) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (bytes > DEVICE_BYTES) return nullptr;
    void *p = reinterpret_cast<void *>(static_cast<uintptr_t>(g_next));
    g_next += ((bytes + 4095) & ~4095ull) + 4096;  // a guard page between allocations: an overrun never lands in the next one
    g_allocated += bytes;
    st().allocs[reinterpret_cast<uintptr_t>(p)] = bytes;
    return p;
}

void merge(VClock &into, const VClock &from) {
    for (const auto &kv : from) {
        unsigned long long &v = into[kv.first];
        if (kv.second > v) v = kv.second;
    }
}

// a new operation on `stream` (caller holds the lock): its clock, with its own stamp
VClock &begin_op(const void *stream, unsigned long long *stamp) {
    VClock &v = st().streams[stream];
    merge(v, st().host);
    *stamp = ++v[stream];
    return v;
}

// FAKEHIP_ORDER=1 switches the range bookkeeping on (quadratic in the number of launches: for the small drivers)
bool order_on() {
    static const bool on = getenv("FAKEHIP_ORDER") != nullptr;
    return on;
}

void touch(const VClock &vc, const void *stream, unsigned long long stamp, const std::string &what, const void *p, long long floats,
           bool write) {
    if (!p || floats <= 0 || !order_on()) return;
    State &S = st();
    const uintptr_t lo = reinterpret_cast<uintptr_t>(p), hi = lo + 4ull * (unsigned long long)floats;
    {   // a range that starts inside one of the library's own allocations must end inside it
        auto al = S.allocs.upper_bound(lo);
        if (al != S.allocs.begin()) {
            --al;
            if (lo < al->first + al->second + 4096 && hi > al->first + al->second) {
                ++S.races;
                fprintf(stderr, "FAKEHIP OVERRUN %s %s %llu bytes past an allocation of %zu\n", what.c_str(), write ? "writes" : "reads",
                        (unsigned long long)(hi - (al->first + al->second)), al->second);
            }
        }
    }
    for (const Access &a : S.log) {
        if (a.hi <= lo || hi <= a.lo || !(a.write || write)) continue;
        if (a.stream == stream) continue;  // same stream: in order
        auto it = vc.find(a.stream);
        if (it != vc.end() && it->second >= a.stamp) continue;  // ordered by an event or a host synchronisation
        ++S.races;
        const std::string kind = a.what + (a.write ? " (write)" : " (read)") + "  <->  " + what + (write ? " (write)" : " (read)");
        if (S.raceKinds[kind]++ == 0) fprintf(stderr, "FAKEHIP RACE %s: overlapping ranges, no ordering between the streams\n", kind.c_str());
    }
    if (S.log.size() > 200000) S.log.erase(S.log.begin(), S.log.begin() + 100000);
    S.log.push_back(Access{lo, hi, write, stream, stamp, what});
}

// what was uploaded to [p, p + bytes), or nullptr
const void *uploaded(const void *p, size_t bytes) {
    State &S = st();
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    auto it = S.uploads.upper_bound(a);
    if (it == S.uploads.begin()) return nullptr;
    --it;
    if (a + bytes > it->first + it->second.size()) return nullptr;
    return it->second.data() + (a - it->first);
}

// afx_cwt.hip: struct CwtGeom (kept in step by hand; a mismatch shows up as wild ranges, i.e. as reported races)
struct CwtGeomView {
    int r1, r2, dataLength, pad, C;
    const void *tw, *fastTw;
    const int *support, *order, *orderLo;
    int num;
};

const char *short_name(const std::string &mangled) {
    static const char *known[] = {"k_cwt_fwd_cols", "k_cwt_fwd_rows", "k_cwt_inv_rows512", "k_cwt_inv_cols256_nb", "k_cwt_inv_cols256",
                                  "k_cqt_decimate", "k_cqt_octave_f16", "k_cqt_octave_mfma_w", "k_cqt_octave_mfma", "k_cqt_octave",
                                  "k_cqt_chroma"};
    for (const char *k : known)
        if (mangled.find(k) != std::string::npos) return k;
    return nullptr;
}

// the CQT kernels' argument lists (afx_cqt.hip, afx_cqt_f16.hip) -> the ranges a launch reads / writes
void record_accesses(const std::string &mangled, dim3 g, void **args, const void *stream) {
    const char *k = order_on() ? short_name(mangled) : nullptr;
    if (!k) return;
    unsigned long long stamp;
    const VClock vc = begin_op(stream, &stamp);
    const std::string name = k;
    if (name.rfind("k_cwt_", 0) == 0) {
        // the four-step CWT kernels (afx_cwt.hip): chunk = blockIdx.y (forward) / blockIdx.z (inverse), scale slot =
        // blockIdx.y of the inverse kernels, scale = order[slot] (wide) or orderLo[listBase + slot].x (narrow-band)
        const CwtGeomView &q = *static_cast<const CwtGeomView *>(args[0]);
        const long long L = 1LL << (q.r1 + q.r2), D = q.dataLength;
        if (name == "k_cwt_fwd_cols") {
            const float *x = *static_cast<const float **>(args[1]);
            const long long xs = *static_cast<long long *>(args[2]);
            float *A = *static_cast<float **>(args[3]);
            for (unsigned c = 0; c < g.y; ++c) {
                touch(vc, stream, stamp, name, x + c * xs, D, false);
                touch(vc, stream, stamp, name, A + 2 * L * c, 2 * L, true);
            }
        } else if (name == "k_cwt_fwd_rows") {
            const float *A = *static_cast<const float **>(args[1]);
            float *Xt = *static_cast<float **>(args[2]);
            for (unsigned c = 0; c < g.y; ++c) {
                touch(vc, stream, stamp, name, A + 2 * L * c, 2 * L, false);
                touch(vc, stream, stamp, name, Xt + 2 * L * c, 2 * L, true);
            }
        } else {
            const bool nb = name == "k_cwt_inv_cols256_nb", rows = name == "k_cwt_inv_rows512";
            const int listBase = nb ? *static_cast<int *>(args[4]) : 0;
            const float *in = *static_cast<const float **>(args[1]);  // Xt (rows512, nb) or B (cols256)
            float *B = rows ? *static_cast<float **>(args[4]) : nullptr;
            float *outRe = rows ? nullptr : *static_cast<float **>(args[nb ? 5 : 2]);
            float *outIm = rows ? nullptr : *static_cast<float **>(args[nb ? 6 : 3]);
            for (unsigned slot = 0; slot < g.y; ++slot) {
                long long j = slot;
                if (nb) {
                    const int *e = static_cast<const int *>(uploaded(q.orderLo + 2 * (listBase + slot), 8));
                    if (!e) { ++st().races; fprintf(stderr, "FAKEHIP RACE check: %s reads an index list that was never uploaded\n", k); return; }
                    j = e[0];
                } else if (q.order) {
                    const int *e = static_cast<const int *>(uploaded(q.order + slot, 4));
                    if (!e) { ++st().races; fprintf(stderr, "FAKEHIP RACE check: %s reads an index list that was never uploaded\n", k); return; }
                    j = e[0];
                }
                if (j < 0 || j >= q.num) { ++st().races; fprintf(stderr, "FAKEHIP RACE check: %s scale %lld of %d\n", k, j, q.num); return; }
                for (unsigned c = 0; c < g.z; ++c) {
                    const long long row = (long long)c * q.num + j;
                    if (rows) {
                        touch(vc, stream, stamp, name, in + 2 * L * c, 2 * L, false);
                        touch(vc, stream, stamp, name, B + 2 * L * row, 2 * L, true);
                    } else {
                        if (nb) touch(vc, stream, stamp, name, in + 2 * L * c, 2 * L, false);
                        else touch(vc, stream, stamp, name, in + 2 * L * row, 2 * L, false);
                        touch(vc, stream, stamp, name, outRe + D * row, D, true);
                        touch(vc, stream, stamp, name, outIm + D * row, D, true);
                    }
                }
            }
        }
    } else if (name == "k_cqt_decimate") {
        const float *x = *static_cast<const float **>(args[0]);
        const int srcLen = *static_cast<int *>(args[1]), dstLen = *static_cast<int *>(args[4]);
        const long long xs = *static_cast<long long *>(args[2]), ys = *static_cast<long long *>(args[5]);
        float *y = *static_cast<float **>(args[3]);
        for (unsigned b = 0; b < g.y; ++b) {
            touch(vc, stream, stamp, name, x + b * xs, srcLen, false);
            touch(vc, stream, stamp, name, y + b * ys, dstLen, true);
        }
    } else if (name.rfind("k_cqt_octave", 0) == 0) {
        const AfxCqtOctaveArgs &a = *static_cast<const AfxCqtOctaveArgs *>(args[0]);
        for (int b = 0; b < (a.batch > 0 ? a.batch : 1); ++b) {
            touch(vc, stream, stamp, name, a.x + b * a.xStride, a.validLength, false);
            touch(vc, stream, stamp, name, a.outRe + b * a.outStride, (long long)a.timeLength * a.num, true);
            touch(vc, stream, stamp, name, a.outIm + b * a.outStride, (long long)a.timeLength * a.num, true);
        }
    } else {  // k_cqt_chroma, k_cqt_chroma_scan: (re, im, rows, num, lists | fold, chromaNum, isMag, normType, out[, vec4])
        const float *re = *static_cast<const float **>(args[0]), *im = *static_cast<const float **>(args[1]);
        const long long rows = *static_cast<long long *>(args[2]);
        const int num = *static_cast<int *>(args[3]), cn = *static_cast<int *>(args[5]);
        touch(vc, stream, stamp, name, re, rows * num, false);
        touch(vc, stream, stamp, name, im, rows * num, false);
        touch(vc, stream, stamp, name, *static_cast<float **>(args[8]), rows * cn, true);
    }
}

void violation(const std::string &name, const char *what, dim3 g, dim3 b, size_t lds) {
    ++g_violations;
    fprintf(stderr, "FAKEHIP VIOLATION %s: %s (grid %u x %u x %u, block %u x %u x %u, dynamic LDS %zu)\n", name.c_str(), what,
            g.x, g.y, g.z, b.x, b.y, b.z, lds);
}
}  // namespace

extern "C" {
// ---- what the audit driver reads
int fakehip_violations(void) { return g_violations; }
int fakehip_rejected(void) { return g_rejected; }
int fakehip_races(void) { return st().races; }
void fakehip_report(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (const auto &kv : g_launches) printf("  launched %8llu x %s\n", kv.second, kv.first.c_str());
}
int fakehip_launched(const char *substr) {
    std::lock_guard<std::mutex> lk(g_mu);
    unsigned long long n = 0;
    for (const auto &kv : g_launches)
        if (kv.first.find(substr) != std::string::npos) n += kv.second;
    return n > 0x7fffffff ? 0x7fffffff : (int)n;
}

// ---- registration hooks emitted by clang for every translation unit with kernels
extern const char fakehip_fatbin_placeholder[16] = {0};
void **__hipRegisterFatBinary(const void *) {
    static void *handle[1];
    return handle;
}
void __hipUnregisterFatBinary(void **) {}
void __hipRegisterFunction(void **, const void *hostFunction, char *, const char *deviceName, unsigned, void *, void *, void *, void *,
                           int *) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_names[hostFunction] = deviceName;
}
void __hipRegisterVar(void **, void *, char *, char *, int, size_t, int, int) {}
hipError_t __hipPushCallConfiguration(dim3 gridDim, dim3 blockDim, size_t sharedMem, hipStream_t stream) {
    t_grid = gridDim;
    t_block = blockDim;
    t_shmem = sharedMem;
    t_stream = stream;
    return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3 *gridDim, dim3 *blockDim, size_t *sharedMem, hipStream_t *stream) {
    *gridDim = t_grid;
    *blockDim = t_block;
    *sharedMem = t_shmem;
    *stream = t_stream;
    return hipSuccess;
}

hipError_t hipLaunchKernel(const void *f, dim3 g, dim3 b, void **args, size_t lds, hipStream_t stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_names.find(f);
    const std::string name = it == g_names.end() ? std::string("<unregistered kernel>") : it->second;
    ++g_launches[name];
    record_accesses(name, g, args, stream);
    if (it == g_names.end()) violation(name, "launch of a function that was never registered", g, b, lds);
    const unsigned long long threads = (unsigned long long)b.x * b.y * b.z;
    if (threads == 0 || threads > 1024) violation(name, "block size outside 1..1024", g, b, lds);
    if (g.x == 0 || g.y == 0 || g.z == 0) violation(name, "empty grid", g, b, lds);
    if (g.x > 0x7fffffffu || g.y > 65535u || g.z > 65535u) violation(name, "grid dimension beyond the HIP limits", g, b, lds);
    if ((unsigned long long)g.x * b.x >= (1ull << 32) || (unsigned long long)g.y * b.y >= (1ull << 32) ||
        (unsigned long long)g.z * b.z >= (1ull << 32)) {
        // HIP rejects this launch (hipErrorInvalidConfiguration): the launcher's check after the launch must turn it
        // into a status.  Reached only by batches whose buffers exceed the device several times over.
        fprintf(stderr, "fakehip: %s with 2^32 or more threads in one dimension (grid %u, block %u): rejected as HIP would\n",
                name.c_str(), g.x, b.x);
        ++g_rejected;
        t_lastError = hipErrorInvalidConfiguration;
        return hipErrorInvalidConfiguration;
    }
    if (lds > 160 * 1024) violation(name, "dynamic LDS beyond the 160 KB of a CU", g, b, lds);
    else if (lds > 64 * 1024) {
        auto a = g_maxDynLds.find(f);
        if (a == g_maxDynLds.end() || (size_t)a->second < lds)
            violation(name, "dynamic LDS above 64 KB without a matching hipFuncSetAttribute", g, b, lds);
    }
    return hipSuccess;
}

hipError_t hipFuncSetAttribute(const void *func, hipFuncAttribute attr, int value) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (attr == hipFuncAttributeMaxDynamicSharedMemorySize) {
        if (value < 0 || value > 160 * 1024) {
            ++g_violations;
            fprintf(stderr, "FAKEHIP VIOLATION hipFuncSetAttribute: MaxDynamicSharedMemorySize %d\n", value);
            return hipErrorInvalidValue;
        }
        g_maxDynLds[func] = value;
    }
    return hipSuccess;
}

// ---- device, memory, streams, events
hipError_t hipGetDeviceCount(int *count) { *count = 1; return hipSuccess; }
hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidDevice; }
hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) {
    memset(p, 0, sizeof(*p));
    snprintf(p->name, sizeof(p->name), "stand-in MI355X");
    snprintf(p->gcnArchName, sizeof(p->gcnArchName), "gfx950:sramecc+:xnack-");
    p->totalGlobalMem = DEVICE_BYTES;
    p->multiProcessorCount = 256;
    p->warpSize = 64;
    p->maxThreadsPerBlock = 1024;
    p->sharedMemPerBlock = 64 * 1024;
    p->maxSharedMemoryPerMultiProcessor = 160 * 1024;
    return hipSuccess;
}
hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t attr, int) {
    if (attr == hipDeviceAttributeMultiprocessorCount) *v = 256;
    else if (attr == hipDeviceAttributeWarpSize) *v = 64;
    else *v = 0;
    return hipSuccess;
}
hipError_t hipMalloc(void **p, size_t bytes) { *p = dry_alloc(bytes ? bytes : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipMallocAsync(void **p, size_t bytes, hipStream_t) { return hipMalloc(p, bytes); }
hipError_t hipFree(void *) { return hipSuccess; }
// pinned host memory (the staging slabs of afx_runtime.hip): plain zeroed host memory, kept until process end
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned) { *p = calloc(bytes ? bytes : 1, 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
hipError_t hipFreeAsync(void *, hipStream_t) { return hipSuccess; }
hipError_t hipMemcpy(void *dst, const void *src, size_t bytes, hipMemcpyKind kind) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (kind == hipMemcpyHostToDevice && bytes <= (1u << 20) && order_on())
        st().uploads[reinterpret_cast<uintptr_t>(dst)] = std::vector<char>(static_cast<const char *>(src), static_cast<const char *>(src) + bytes);
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    unsigned long long stamp;
    const VClock vc = begin_op(stream, &stamp);
    if (kind == hipMemcpyHostToDevice && bytes <= (1u << 20) && order_on())
        st().uploads[reinterpret_cast<uintptr_t>(dst)] = std::vector<char>(static_cast<const char *>(src), static_cast<const char *>(src) + bytes);
    if (kind == hipMemcpyHostToDevice || kind == hipMemcpyDeviceToDevice) touch(vc, stream, stamp, "copy", dst, (long long)(bytes / 4), true);
    if (kind == hipMemcpyDeviceToHost || kind == hipMemcpyDeviceToDevice) touch(vc, stream, stamp, "copy", src, (long long)(bytes / 4), false);
    return hipSuccess;
}
hipError_t hipMemset(void *, int, size_t) { return hipSuccess; }
hipError_t hipMemsetAsync(void *dst, int, size_t bytes, hipStream_t stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    unsigned long long stamp;
    const VClock vc = begin_op(stream, &stamp);
    touch(vc, stream, stamp, "memset", dst, (long long)(bytes / 4), true);
    return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = reinterpret_cast<hipStream_t>(malloc(8)); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    merge(st().host, st().streams[s]);
    return hipSuccess;
}
hipError_t hipStreamGetDevice(hipStream_t, hipDevice_t *d) { *d = 0; return hipSuccess; }
int hipGetStreamDeviceId(hipStream_t) { return 0; }
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) {
    std::lock_guard<std::mutex> lk(g_mu);
    const char *drop = getenv("FAKEHIP_DROP_WAIT");
    if (++st().waits == (drop ? atoi(drop) : -1)) return hipSuccess;  // the detector's own test: this edge is lost
    merge(st().streams[s], st().events[e]);
    return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = reinterpret_cast<hipEvent_t>(malloc(8)); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) {
    std::lock_guard<std::mutex> lk(g_mu);
    st().events.erase(e);
    free(e);
    return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    VClock &v = st().streams[s];
    merge(v, st().host);
    st().events[e] = v;
    return hipSuccess;
}
hipError_t hipGetLastError(void) {
    const hipError_t e = t_lastError;
    t_lastError = hipSuccess;
    return e;
}
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "stand-in HIP error"; }
}
