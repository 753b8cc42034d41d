// afx_comm.hip -- the one exchange step of the path as a C-ABI export: feature
// slabs of the ranks (one process per GPU, clips sharded contiguously) gathered to a root over
// RCCL / xGMI with ncclGather (/opt/rocm/include/rccl/rccl.h:745).  The reference has no collective;
// the transforms need none (every clip is independent) -- this is the only cross-GPU traffic.
//
// RCCL is bound at RUN TIME (dlopen): the library keeps linking libamdhip64 only, single-GPU users
// never map librccl, and a process that already holds a copy (PyTorch bundles one) shares it --
// two RCCL instances in one process would each build their own transport state.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <mutex>

#include "afx_batch.h"
#include "afx_device.h"
#include "afx_hipcheck.h"

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*getUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*commInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*commDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*gather)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*errorString)(ncclResult_t) = nullptr;
    char why[256] = "";
};
Rccl g_rccl;
std::once_flag g_rcclOnce;

const Rccl *rccl() {
    std::call_once(g_rcclOnce, [] {
        Rccl &r = g_rccl;
        const char *names[] = {"librccl.so.1", "librccl.so"};
        // a copy that is already mapped (RTLD_NOLOAD) wins, then the loader's search path / our RUNPATH
        for (int pass = 0; pass < 2 && !r.handle; ++pass)
            for (const char *n : names) {
                r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
                if (r.handle) break;
            }
        if (!r.handle) {
            snprintf(r.why, sizeof(r.why), "librccl.so.1 not found (%s)", dlerror());
            return;
        }
        r.getUniqueId = reinterpret_cast<decltype(r.getUniqueId)>(dlsym(r.handle, "ncclGetUniqueId"));
        r.commInitRank = reinterpret_cast<decltype(r.commInitRank)>(dlsym(r.handle, "ncclCommInitRank"));
        r.commDestroy = reinterpret_cast<decltype(r.commDestroy)>(dlsym(r.handle, "ncclCommDestroy"));
        r.gather = reinterpret_cast<decltype(r.gather)>(dlsym(r.handle, "ncclGather"));
        r.errorString = reinterpret_cast<decltype(r.errorString)>(dlsym(r.handle, "ncclGetErrorString"));
        if (!r.getUniqueId || !r.commInitRank || !r.commDestroy || !r.gather || !r.errorString) {
            snprintf(r.why, sizeof(r.why), "librccl lacks ncclGather / ncclCommInitRank");
            r.handle = nullptr;
        }
    });
    if (!g_rccl.handle) {
        afxdev_set_error("afx_gather: RCCL unavailable: %s", g_rccl.why);
        return nullptr;
    }
    return &g_rccl;
}

#define AFX_NCCL(r, call)                                                                   \
    do {                                                                                    \
        ncclResult_t _e = (call);                                                           \
        if (_e != ncclSuccess) {                                                            \
            afxdev_set_error("%s failed: %s", #call, (r)->errorString(_e));                 \
            return AFX_ERR_HIP;                                                             \
        }                                                                                   \
    } while (0)

}  // namespace

struct AfxComm {
    ncclComm_t comm;
    int world, rank, device;
};

extern "C" int afx_comm_get_unique_id(void *id) {
    if (!id) return AFX_ERR_ARG;
    static_assert(AFX_COMM_ID_BYTES == sizeof(ncclUniqueId), "AFX_COMM_ID_BYTES");
    const Rccl *r = rccl();
    if (!r) return AFX_ERR_UNSUPPORTED;
    ncclUniqueId u;
    AFX_NCCL(r, r->getUniqueId(&u));
    memcpy(id, &u, sizeof(u));
    return AFX_OK;
}

// The communicator is bound to the CALLING THREAD'S CURRENT HIP device -- what hipSetDevice / torch.cuda.set_device /
// afx_set_device last selected on this thread -- not to the library's default: with one process per GPU every rank
// must initialise RCCL on its own device (all on the default device = duplicate-GPU error or a hang).  The thread's
// current device is left as it was.
extern "C" int afx_comm_create(AfxCommObj *comm, int worldSize, int rank, const void *id) {
    if (!comm) return AFX_ERR_ARG;
    *comm = nullptr;
    if (!id || worldSize < 1 || rank < 0 || rank >= worldSize) return AFX_ERR_ARG;
    int dev = -1;
    if (afxdev_device_count() <= 0 || hipGetDevice(&dev) != hipSuccess || dev < 0) {
        afxdev_set_error("afx_comm_create: no current HIP device");
        return AFX_ERR_NODEVICE;
    }
    const Rccl *r = rccl();
    if (!r) return AFX_ERR_UNSUPPORTED;
    AfxComm *c = static_cast<AfxComm *>(calloc(1, sizeof(AfxComm)));
    if (!c) return AFX_ERR_NOMEM;
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclResult_t e = r->commInitRank(&c->comm, worldSize, u, rank);
    if (e != ncclSuccess) {
        afxdev_set_error("ncclCommInitRank(world %d, rank %d, device %d) failed: %s", worldSize, rank, dev, r->errorString(e));
        free(c);
        return AFX_ERR_HIP;
    }
    c->world = worldSize;
    c->rank = rank;
    c->device = dev;
    *comm = c;
    return AFX_OK;
}

extern "C" int afx_comm_world_size(AfxCommObj c) { return c ? c->world : 0; }
extern "C" int afx_comm_rank(AfxCommObj c) { return c ? c->rank : -1; }

extern "C" int afx_gather(AfxCommObj c, const float *dSend, long long count, float *dRecv, int root, void *hipStream) {
    if (!c || !dSend || count < 0 || root < 0 || root >= c->world) return AFX_ERR_ARG;
    if (c->rank == root && !dRecv) return AFX_ERR_ARG;
    const Rccl *r = rccl();
    if (!r) return AFX_ERR_UNSUPPORTED;
    // the stream (and with it the slabs) must live on the communicator's device
    if (hipStream) {
        hipDevice_t sd = 0;
        if (hipStreamGetDevice((hipStream_t)hipStream, &sd) == hipSuccess && (int)sd != c->device) {
            afxdev_set_error("afx_gather: stream on device %d, communicator on device %d", (int)sd, c->device);
            return AFX_ERR_ARG;
        }
    }
    if (count == 0) return AFX_OK;
    const int prev = afxdev_current_device();
    if (prev != c->device) AFX_HIP(hipSetDevice(c->device));
    const ncclResult_t e = r->gather(dSend, dRecv, (size_t)count, ncclFloat32, root, c->comm, (hipStream_t)hipStream);
    if (prev >= 0 && prev != c->device) (void)hipSetDevice(prev);  // the caller's device stays current
    if (e != ncclSuccess) {
        afxdev_set_error("ncclGather failed: %s", r->errorString(e));
        return AFX_ERR_HIP;
    }
    return AFX_OK;
}

extern "C" void afx_comm_free(AfxCommObj c) {
    if (!c) return;
    const Rccl *r = rccl();
    if (r && c->comm) (void)r->commDestroy(c->comm);
    free(c);
}
