// RCCL over xGMI for the two exchange steps of the image-sharded job: broadcast of the text conditioning from
// rank 0 and gather of the finished latents / images.  The library owns the data-path collectives (C entry points, plain
// pointers); the launcher only has to hand every rank the 128-byte unique id (any side channel: torch.distributed, MPI, a
// file).  librccl is opened lazily with dlopen so that single-GPU users and the CPU-only test container do not need it.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <string.h>
#include "common.hpp"
#include "mlsd_kernels.h"

namespace {

typedef struct { char internal[128]; } UniqueId;           // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef int (*fn_uid)(UniqueId*);
typedef int (*fn_init)(void**, int, UniqueId, int);
typedef int (*fn_destroy)(void*);
typedef int (*fn_bcast)(const void*, void*, size_t, int, int, void*, hipStream_t);      // ncclBroadcast(send, recv, count, dtype, root, comm, stream)
typedef int (*fn_allgather)(const void*, void*, size_t, int, void*, hipStream_t);      // ncclAllGather(send, recv, sendcount, dtype, comm, stream)
typedef const char* (*fn_errstr)(int);

struct Rccl { void* h; fn_uid uid; fn_init init; fn_destroy destroy; fn_bcast bcast; fn_allgather allgather; fn_errstr errstr; } g = {};

int rccl_load()
{
    if (g.h) return 0;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) { g.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (g.h) break; }
    if (!g.h) return mlsd_set_error(-1, "RCCL not available: %s", dlerror());
    g.uid = (fn_uid)dlsym(g.h, "ncclGetUniqueId"); g.init = (fn_init)dlsym(g.h, "ncclCommInitRank");
    g.destroy = (fn_destroy)dlsym(g.h, "ncclCommDestroy"); g.bcast = (fn_bcast)dlsym(g.h, "ncclBroadcast");
    g.allgather = (fn_allgather)dlsym(g.h, "ncclAllGather"); g.errstr = (fn_errstr)dlsym(g.h, "ncclGetErrorString");
    if (!g.uid || !g.init || !g.destroy || !g.bcast || !g.allgather) { g.h = nullptr; return mlsd_set_error(-1, "RCCL: missing symbols"); }
    return 0;
}

int chk(int r, const char* what) { return r == 0 ? 0 : mlsd_set_error(-1, "%s failed: %s", what, g.errstr ? g.errstr(r) : "RCCL error"); }

}  // namespace

extern "C" {

MLSD_API int mlsd_rccl_unique_id(void* out128)
{
    if (rccl_load()) return -1;
    UniqueId id;
    if (chk(g.uid(&id), "ncclGetUniqueId")) return -1;
    memcpy(out128, &id, 128);
    return 0;
}

MLSD_API int mlsd_rccl_init(void** comm, int world, int rank, const void* id128)
{
    if (rccl_load()) return -1;
    UniqueId id; memcpy(&id, id128, 128);
    return chk(g.init(comm, world, id, rank), "ncclCommInitRank");
}

MLSD_API int mlsd_rccl_destroy(void* comm) { return (comm && g.destroy) ? chk(g.destroy(comm), "ncclCommDestroy") : 0; }

MLSD_API int mlsd_rccl_bcast(void* comm, void* buf, size_t nbytes, int root, void* stream)
{
    if (!comm) return mlsd_set_error(-1, "mlsd_rccl_bcast: no communicator");
    return chk(g.bcast(buf, buf, nbytes, 0 /* ncclInt8 */, root, comm, (hipStream_t)stream), "ncclBroadcast");
}

MLSD_API int mlsd_rccl_all_gather(void* comm, const void* send, void* recv, size_t nbytes_per_rank, void* stream)
{
    if (!comm) return mlsd_set_error(-1, "mlsd_rccl_all_gather: no communicator");
    return chk(g.allgather(send, recv, nbytes_per_rank, 0 /* ncclInt8 */, comm, (hipStream_t)stream), "ncclAllGather");
}

}  // extern "C"
