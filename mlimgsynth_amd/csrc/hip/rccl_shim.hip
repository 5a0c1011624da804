// RCCL over xGMI for the two exchange steps of the image-sharded job: broadcast of the text conditioning from
// rank 0 and gather of the finished latents / images.  The library owns the data-path collectives (C entry points, plain
// pointers); the launcher only has to hand every rank the 128-byte unique id (any side channel: torch.distributed, MPI, a
// file).  librccl is opened lazily with dlopen so that single-GPU users and the CPU-only test container do not need it.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <string.h>
#include <stdlib.h>
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
typedef int (*fn_count)(const void*, int*);                                           // ncclCommCount(comm, &count)

struct Rccl { void* h; fn_uid uid; fn_init init; fn_destroy destroy; fn_bcast bcast; fn_allgather allgather; fn_errstr errstr; fn_count count; } g = {};

int rccl_load()
{
    if (g.h) return 0;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) { g.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (g.h) break; }
    if (!g.h) return mlsd_set_error(-1, "RCCL not available: %s", dlerror());
    g.uid = (fn_uid)dlsym(g.h, "ncclGetUniqueId"); g.init = (fn_init)dlsym(g.h, "ncclCommInitRank");
    g.destroy = (fn_destroy)dlsym(g.h, "ncclCommDestroy"); g.bcast = (fn_bcast)dlsym(g.h, "ncclBroadcast");
    g.allgather = (fn_allgather)dlsym(g.h, "ncclAllGather"); g.errstr = (fn_errstr)dlsym(g.h, "ncclGetErrorString");
    g.count = (fn_count)dlsym(g.h, "ncclCommCount");
    if (!g.uid || !g.init || !g.destroy || !g.bcast || !g.allgather) { g.h = nullptr; return mlsd_set_error(-1, "RCCL: missing symbols"); }
    return 0;
}

int chk(int r, const char* what) { return r == 0 ? 0 : mlsd_set_error(-1, "%s failed: %s", what, g.errstr ? g.errstr(r) : "RCCL error"); }

// A communicator is either RCCL's (device buffers, asynchronous on the stream) or a HOST transport supplied by the launcher
// (two callbacks that move host bytes: gloo, MPI, ...).  The host transport serves the CPU-only tests -- the engine's
// mlis_amd_bcast_cond / mlis_amd_gather_results run unchanged on two gloo ranks in the dry runtime -- and machines without RCCL
// (device buffers are staged through host memory there).
struct Comm {
    int kind;                 // 0 = RCCL, 1 = host transport
    void* nccl;
    int world, rank;
    mlsd_host_bcast_fn hb; mlsd_host_allgather_fn hg; void* user;
};

int host_stage(void* dev, void* host, size_t n, int to_host, hipStream_t st)
{
    if (mlsd_runtime_is_dry()) { if (dev != host) memcpy(to_host ? host : dev, to_host ? dev : host, n); return 0; }
    MLSD_HIP_TRY(hipMemcpyAsync(to_host ? host : dev, to_host ? dev : host, n, to_host ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice, st));
    MLSD_HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

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
    void* nc = nullptr;
    if (chk(g.init(&nc, world, id, rank), "ncclCommInitRank")) return -1;
    Comm* c = (Comm*)calloc(1, sizeof(Comm));
    c->kind = 0; c->nccl = nc; c->world = world; c->rank = rank;
    *comm = c;
    return 0;
}

MLSD_API int mlsd_comm_host(void** comm, int world, int rank, mlsd_host_bcast_fn bcast, mlsd_host_allgather_fn all_gather, void* user)
{
    if (!comm || !bcast || !all_gather || world < 1 || rank < 0 || rank >= world) return mlsd_set_error(-1, "mlsd_comm_host: bad arguments");
    Comm* c = (Comm*)calloc(1, sizeof(Comm));
    c->kind = 1; c->world = world; c->rank = rank; c->hb = bcast; c->hg = all_gather; c->user = user;
    *comm = c;
    return 0;
}

MLSD_API int mlsd_rccl_destroy(void* comm)
{
    Comm* c = (Comm*)comm;
    if (!c) return 0;
    int r = (c->kind == 0 && c->nccl && g.destroy) ? chk(g.destroy(c->nccl), "ncclCommDestroy") : 0;
    free(c);
    return r;
}

/* ranks of the communicator AS THE TRANSPORT REPORTS THEM: ncclCommCount for RCCL (what the launcher prints as `rccl_ranks`), the stated world of a host transport */
MLSD_API int mlsd_comm_count(void* comm, int* count, int* kind)
{
    Comm* c = (Comm*)comm;
    if (!c || !count) return mlsd_set_error(-1, "mlsd_comm_count: bad arguments");
    if (kind) *kind = c->kind;
    if (c->kind == 0) {
        if (!g.count) return mlsd_set_error(-1, "RCCL: ncclCommCount missing");
        return chk(g.count(c->nccl, count), "ncclCommCount");
    }
    *count = c->world;
    return 0;
}

MLSD_API int mlsd_rccl_bcast(void* comm, void* buf, size_t nbytes, int root, void* stream)
{
    Comm* c = (Comm*)comm;
    if (!c) return mlsd_set_error(-1, "mlsd_rccl_bcast: no communicator");
    if (c->kind == 0) return chk(g.bcast(buf, buf, nbytes, 0 /* ncclInt8 */, root, c->nccl, (hipStream_t)stream), "ncclBroadcast");
    void* h = mlsd_runtime_is_dry() ? buf : malloc(nbytes);
    if (!h) return mlsd_set_error(-1, "mlsd_rccl_bcast: out of host memory");
    int r = (c->rank == root) ? host_stage(buf, h, nbytes, 1, (hipStream_t)stream) : (mlsd_runtime_is_dry() ? 0 : (int)(hipStreamSynchronize((hipStream_t)stream) != hipSuccess));
    if (!r && c->hb(c->user, h, nbytes, root)) r = mlsd_set_error(-1, "host transport: broadcast failed");
    if (!r && c->rank != root) r = host_stage(buf, h, nbytes, 0, (hipStream_t)stream);
    if (h != buf) free(h);
    return r;
}

MLSD_API int mlsd_rccl_all_gather(void* comm, const void* send, void* recv, size_t nbytes_per_rank, void* stream)
{
    Comm* c = (Comm*)comm;
    if (!c) return mlsd_set_error(-1, "mlsd_rccl_all_gather: no communicator");
    if (c->kind == 0) return chk(g.allgather(send, recv, nbytes_per_rank, 0 /* ncclInt8 */, c->nccl, (hipStream_t)stream), "ncclAllGather");
    const bool dry = mlsd_runtime_is_dry();
    void* hs = dry ? (void*)send : malloc(nbytes_per_rank);
    void* hr = dry ? recv : malloc(nbytes_per_rank * c->world);
    if (!hs || !hr) { if (!dry) { free(hs); free(hr); } return mlsd_set_error(-1, "mlsd_rccl_all_gather: out of host memory"); }
    int r = host_stage((void*)send, hs, nbytes_per_rank, 1, (hipStream_t)stream);
    if (!r && c->hg(c->user, hs, hr, nbytes_per_rank)) r = mlsd_set_error(-1, "host transport: all-gather failed");
    if (!r) r = host_stage(recv, hr, nbytes_per_rank * c->world, 0, (hipStream_t)stream);
    if (!dry) { free(hs); free(hr); }
    return r;
}

}  // extern "C"
