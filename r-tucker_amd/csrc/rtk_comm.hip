// The exchange step of the entity-sharded path through the C ABI: RCCL all-gather of the per-shard score blocks
// over xGMI (BASELINE.json north_star), so that a host in any language can drive the
// sharded path -- the Python mirror (sharded.py) can use torch.distributed for the same collective.
//
// RCCL is bound at RUN time (dlopen of the copy already in the process -- torch ships its own librccl.so -- else
// librccl.so from the ROCm install): the library still loads on a box without RCCL, and a process never holds two
// copies with clashing ncclXxx symbols.  Four entry points: a unique id for rank 0 to hand to its peers by its
// own means (file, socket, MPI, torch.distributed's store), communicator init / destroy, and the in-place
// all-gather of equally sized blocks.
#include "rtk_common.h"
#include <dlfcn.h>
#include <mutex>
#include <string.h>

namespace {

constexpr int kIdBytes = 128;                     // NCCL_UNIQUE_ID_BYTES (rccl.h:40)
struct UniqueId { char internal[kIdBytes]; };     // ncclUniqueId is passed BY VALUE to ncclCommInitRank
typedef int (*fn_get_id)(UniqueId *);
typedef int (*fn_init_rank)(void **, int, UniqueId, int);
typedef int (*fn_destroy)(void *);
typedef int (*fn_allgather)(const void *, void *, size_t, int, void *, hipStream_t);
typedef const char *(*fn_errstr)(int);

struct Rccl {
    void *handle = nullptr;
    fn_get_id get_id = nullptr;
    fn_init_rank init_rank = nullptr;
    fn_destroy destroy = nullptr;
    fn_allgather allgather = nullptr;
    fn_errstr errstr = nullptr;
    bool ok = false;
};

Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char *n : names)                                   // a copy that is already mapped wins
            if ((r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL))) break;
        if (!r.handle)
            for (const char *n : names)
                if ((r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!r.handle) return;
        r.get_id = (fn_get_id)dlsym(r.handle, "ncclGetUniqueId");
        r.init_rank = (fn_init_rank)dlsym(r.handle, "ncclCommInitRank");
        r.destroy = (fn_destroy)dlsym(r.handle, "ncclCommDestroy");
        r.allgather = (fn_allgather)dlsym(r.handle, "ncclAllGather");
        r.errstr = (fn_errstr)dlsym(r.handle, "ncclGetErrorString");
        r.ok = r.get_id && r.init_rank && r.destroy && r.allgather;
    });
    return r;
}

int fail(const char *what, int rc) {
    Rccl &r = rccl();
    rtk_set_error("%s: RCCL error %d (%s)", what, rc, r.errstr ? r.errstr(rc) : "?");
    return RTK_ERR_LAUNCH;
}

struct Comm {
    void *nccl;
    int rank, world;
};

}  // namespace

#define RTK_NEED_RCCL(what)                                                                           \
    do {                                                                                              \
        if (!rccl().ok) {                                                                             \
            rtk_set_error("%s: librccl.so not found (or incomplete) in this process / ROCm install", what); \
            return RTK_ERR_UNSUPPORTED;                                                               \
        }                                                                                             \
    } while (0)

extern "C" int rtk_comm_unique_id(void *id_out) {
    RTK_REQUIRE(id_out, RTK_ERR_BAD_ARG, "rtk_comm_unique_id: null output");
    RTK_NEED_RCCL("rtk_comm_unique_id");
    UniqueId id;
    const int rc = rccl().get_id(&id);
    if (rc != 0) return fail("rtk_comm_unique_id", rc);
    memcpy(id_out, id.internal, kIdBytes);
    return RTK_OK;
}

extern "C" int rtk_comm_init(int rank, int world, const void *unique_id, void **comm_out) {
    RTK_REQUIRE(comm_out, RTK_ERR_BAD_ARG, "rtk_comm_init: null comm_out");
    *comm_out = nullptr;
    RTK_REQUIRE(unique_id, RTK_ERR_BAD_ARG, "rtk_comm_init: null unique_id");
    RTK_REQUIRE(world >= 1 && rank >= 0 && rank < world, RTK_ERR_BAD_ARG, "rtk_comm_init: rank %d not in [0, world = %d)", rank, world);
    RTK_NEED_RCCL("rtk_comm_init");
    UniqueId id;
    memcpy(id.internal, unique_id, kIdBytes);
    void *nc = nullptr;
    const int rc = rccl().init_rank(&nc, world, id, rank);          // uses the calling thread's current device
    if (rc != 0) return fail("rtk_comm_init", rc);
    *comm_out = new Comm{nc, rank, world};
    return RTK_OK;
}

extern "C" int rtk_allgather_scores(void *comm, void *buf, size_t bytes_per_rank, void *stream) {
    RTK_REQUIRE(comm, RTK_ERR_BAD_ARG, "rtk_allgather_scores: null communicator");
    RTK_REQUIRE(buf || bytes_per_rank == 0, RTK_ERR_BAD_ARG, "rtk_allgather_scores: null buffer");
    RTK_NEED_RCCL("rtk_allgather_scores");
    Comm *c = (Comm *)comm;
    if (bytes_per_rank == 0) return RTK_OK;
    // in place: rank p's block is slot p of the (world, bytes_per_rank) buffer (ncclAllGather's in-place form)
    const char *mine = (const char *)buf + (size_t)c->rank * bytes_per_rank;
    const int rc = rccl().allgather(mine, buf, bytes_per_rank, /* ncclInt8 */ 0, c->nccl, (hipStream_t)stream);
    if (rc != 0) return fail("rtk_allgather_scores", rc);
    return RTK_OK;
}

extern "C" int rtk_comm_destroy(void *comm) {
    if (!comm) return RTK_OK;
    Comm *c = (Comm *)comm;
    int rc = 0;
    if (rccl().ok) rc = rccl().destroy(c->nccl);
    delete c;
    if (rc != 0) return fail("rtk_comm_destroy", rc);
    return RTK_OK;
}
