// Multi-GPU result exchange behind the C ABI: RCCL (AMD's NCCL) over xGMI.
//
// The path shards over scan positions (one process per GPU); what the ranks exchange is the
// per-partition nav-grid result: an all-gather of equal row blocks for 'disjoint' nav buffers
// (ApplyMasksUDF, SumSigUDF, CoM), an all-reduce(sum) for sig buffers (SumUDF) -- the reference moves
// the same data as pickled partition results over TCP (src/libertem/executor/dask.py:581-646) and
// merges them one by one on the main process (src/libertem/udf/base.py:2340-2358).
//
// librccl is opened lazily (dlopen): a process that never creates a communicator does not need it.  A
// process that already runs torch.distributed has an RCCL mapped -- torch bundles its own copy as
// torch/lib/librccl.so -- and must not get a SECOND one (two RCCLs in one process: two sets of IPC
// handles and proxy threads on the same GPUs): the loaded objects are searched first
// (dl_iterate_phdr) and the copy that is there is bound with RTLD_NOLOAD; only a process without any
// RCCL opens one by name (LTMI_RCCL_LIB, then the loader path, then /opt/rocm/lib).
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include "ltmi_common.h"
#include <dlfcn.h>
#include <link.h>
#include <string.h>
#include <new>
#include <string>

namespace {

typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
enum { ncclSum = 0 };
// ncclDataType_t values (rccl.h): int8 0, uint8 1, int32 2, uint32 3, int64 4, uint64 5, float16 6,
// float32 7, float64 8
static int nccl_dtype(int dt, int64_t *scale) {
    *scale = 1;
    switch (dt) {
        case LTMI_I8: return 0;
        case LTMI_BOOL: case LTMI_U8: return 1;
        case LTMI_I32: return 2;
        case LTMI_U32: return 3;
        case LTMI_I64: return 4;
        case LTMI_U64: return 5;
        case LTMI_F32: return 7;
        case LTMI_F64: return 8;
        case LTMI_C64: *scale = 2; return 7;
        case LTMI_C128: *scale = 2; return 8;
    }
    return -1;          // 16-bit integers have no RCCL sum
}

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    std::string path;            // the file the functions come from
    bool was_loaded = false;     // bound to a copy that was already in the process
};

static int find_loaded_rccl(struct dl_phdr_info *info, size_t, void *out) {
    const char *name = info->dlpi_name;
    if (name && strstr(name, "librccl")) {
        *(std::string *)out = name;
        return 1;
    }
    return 0;
}

static Rccl *rccl() {
    static Rccl r;
    static bool tried = false;
    if (tried) return r.lib ? &r : nullptr;
    tried = true;
    std::string loaded;
    dl_iterate_phdr(find_loaded_rccl, &loaded);
    if (!loaded.empty()) {
        r.lib = dlopen(loaded.c_str(), RTLD_NOW | RTLD_NOLOAD);
        r.was_loaded = r.lib != nullptr;
    }
    if (!r.lib) {
        const char *env = getenv("LTMI_RCCL_LIB");
        const char *names[] = {env ? env : "librccl.so.1", "librccl.so.1", "librccl.so",
                               "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
    }
    if (!r.lib) return nullptr;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))dlsym(r.lib, "ncclAllGather");
    r.AllReduce = (decltype(r.AllReduce))dlsym(r.lib, "ncclAllReduce");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
    r.GetVersion = (decltype(r.GetVersion))dlsym(r.lib, "ncclGetVersion");
    Dl_info di;
    if (r.GetUniqueId && dladdr((void *)r.GetUniqueId, &di) && di.dli_fname) r.path = di.dli_fname;
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.AllReduce) {
        dlclose(r.lib);
        r.lib = nullptr;
        return nullptr;
    }
    return &r;
}

}  // namespace

struct ltmi_comm {
    int device = 0, rank = 0, world = 1;
    ncclComm_t comm = nullptr;
};

#define LTMI_RCCL(R, expr, what)                                                              \
    do {                                                                                      \
        ncclResult_t _r = (expr);                                                             \
        if (_r != 0) {                                                                        \
            ltmi::set_error("%s failed: %s", what,                                            \
                            (R)->GetErrorString ? (R)->GetErrorString(_r) : "RCCL error");    \
            return LTMI_E_RCCL_BASE + (int)_r;                                                \
        }                                                                                     \
    } while (0)

extern "C" int ltmi_comm_library_info(char *path_out, int64_t path_cap, int *version, int *was_loaded) {
    Rccl *r = rccl();
    if (!r) LTMI_FAIL(LTMI_E_INVALID, "librccl could not be loaded: %s", dlerror());
    if (path_out && path_cap > 0) {
        strncpy(path_out, r->path.c_str(), (size_t)path_cap - 1);
        path_out[path_cap - 1] = 0;
    }
    if (version) {
        *version = 0;
        if (r->GetVersion) (void)r->GetVersion(version);
    }
    if (was_loaded) *was_loaded = r->was_loaded ? 1 : 0;
    return LTMI_OK;
}

extern "C" int ltmi_comm_unique_id(void *id_out) {
    if (!id_out) LTMI_FAIL(LTMI_E_INVALID, "ltmi_comm_unique_id: null argument");
    Rccl *r = rccl();
    if (!r) LTMI_FAIL(LTMI_E_INVALID, "librccl could not be loaded: %s", dlerror());
    ncclUniqueId id;
    LTMI_RCCL(r, r->GetUniqueId(&id), "ncclGetUniqueId");
    memcpy(id_out, &id, sizeof(id));
    return LTMI_OK;
}

extern "C" int ltmi_comm_create(int device, int rank, int world, const void *id, ltmi_comm **out) {
    if (!id || !out || world < 1 || rank < 0 || rank >= world)
        LTMI_FAIL(LTMI_E_INVALID, "ltmi_comm_create: bad arguments (rank %d of %d)", rank, world);
    Rccl *r = rccl();
    if (!r) LTMI_FAIL(LTMI_E_INVALID, "librccl could not be loaded: %s", dlerror());
    LTMI_HIP(hipSetDevice(device));
    ltmi_comm *c = new (std::nothrow) ltmi_comm();
    if (!c) LTMI_FAIL(LTMI_E_NOMEM, "out of host memory");
    c->device = device;
    c->rank = rank;
    c->world = world;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclResult_t rc = r->CommInitRank(&c->comm, world, uid, rank);
    if (rc != 0) {
        delete c;
        ltmi::set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, world,
                        r->GetErrorString ? r->GetErrorString(rc) : "RCCL error");
        return LTMI_E_RCCL_BASE + (int)rc;
    }
    *out = c;
    return LTMI_OK;
}

extern "C" int ltmi_comm_destroy(ltmi_comm *c) {
    if (!c) return LTMI_OK;
    Rccl *r = rccl();
    if (r && c->comm) (void)r->CommDestroy(c->comm);
    delete c;
    return LTMI_OK;
}

extern "C" int ltmi_comm_all_gather(ltmi_comm *c, const void *send, void *recv,
                                    int64_t bytes_per_rank, void *stream) {
    if (!c || !send || !recv || bytes_per_rank < 0)
        LTMI_FAIL(LTMI_E_INVALID, "ltmi_comm_all_gather: bad arguments");
    if (bytes_per_rank == 0) return LTMI_OK;
    Rccl *r = rccl();
    if (!r) LTMI_FAIL(LTMI_E_INVALID, "librccl could not be loaded");
    LTMI_HIP(hipSetDevice(c->device));
    LTMI_RCCL(r, r->AllGather(send, recv, (size_t)bytes_per_rank, /*ncclUint8*/ 1, c->comm,
                              (hipStream_t)stream), "ncclAllGather");
    return LTMI_OK;
}

extern "C" int ltmi_comm_all_reduce_sum(ltmi_comm *c, void *buf, int dtype, int64_t n,
                                        void *stream) {
    if (!c || !buf || n < 0) LTMI_FAIL(LTMI_E_INVALID, "ltmi_comm_all_reduce_sum: bad arguments");
    if (n == 0) return LTMI_OK;
    int64_t scale = 1;
    const int nt = nccl_dtype(dtype, &scale);
    if (nt < 0)
        LTMI_FAIL(LTMI_E_DTYPE, "ltmi_comm_all_reduce_sum: no RCCL sum for dtype %s",
                  ltmi::dtype_name(dtype));
    Rccl *r = rccl();
    if (!r) LTMI_FAIL(LTMI_E_INVALID, "librccl could not be loaded");
    LTMI_HIP(hipSetDevice(c->device));
    LTMI_RCCL(r, r->AllReduce(buf, buf, (size_t)(n * scale), nt, ncclSum, c->comm,
                              (hipStream_t)stream), "ncclAllReduce");
    return LTMI_OK;
}
