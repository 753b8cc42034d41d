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
// With it the launch arithmetic of every kernel -- including the ones that have not been on hardware yet
// (AFX_CQT_FUSED, AFX_CQT_CHROMA_V2, AFX_GEMM_BF16) -- runs at the BASELINE sizes and far beyond them without a
// GPU.  It says nothing about what the kernels compute.  Test infrastructure, never linked into the product.
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>

namespace {
// constructed on first use: clang's registration hooks run from static constructors of the kernel translation units
struct State {
    std::mutex mu;
    std::map<const void *, std::string> names;
    std::map<const void *, int> maxDynLds;
    std::map<std::string, unsigned long long> launches;
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
    g_next += (bytes + 4095) & ~4095ull;
    g_allocated += bytes;
    return p;
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

hipError_t hipLaunchKernel(const void *f, dim3 g, dim3 b, void **, size_t lds, hipStream_t) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_names.find(f);
    const std::string name = it == g_names.end() ? std::string("<unregistered kernel>") : it->second;
    ++g_launches[name];
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
hipError_t hipFreeAsync(void *, hipStream_t) { return hipSuccess; }
hipError_t hipMemcpy(void *, const void *, size_t, hipMemcpyKind) { return hipSuccess; }
hipError_t hipMemcpyAsync(void *, const void *, size_t, hipMemcpyKind, hipStream_t) { return hipSuccess; }
hipError_t hipMemset(void *, int, size_t) { return hipSuccess; }
hipError_t hipMemsetAsync(void *, int, size_t, hipStream_t) { return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = reinterpret_cast<hipStream_t>(malloc(8)); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamGetDevice(hipStream_t, hipDevice_t *d) { *d = 0; return hipSuccess; }
int hipGetStreamDeviceId(hipStream_t) { return 0; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = reinterpret_cast<hipEvent_t>(malloc(8)); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipGetLastError(void) {
    const hipError_t e = t_lastError;
    t_lastError = hipSuccess;
    return e;
}
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "stand-in HIP error"; }
}
