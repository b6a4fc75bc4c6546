// gsf_comm.hip -- the ONE collective of the path: collecting the fused poses of every rank with RCCL over xGMI.
// RCCL is resolved at run time (dlopen/dlsym) so that libgsf.so has no link-time dependency on a particular librccl.so: in a
// PyTorch process the already-mapped copy (torch bundles its own) is reused, a native caller gets /opt/rocm's.
//
// The library can own its communicator (gsf_comm_unique_id / gsf_comm_init_rank / gsf_comm_destroy: one process per GPU, the
// 128-byte id travels by whatever side channel the host has -- torch.distributed's store in bench.py) or take a caller-provided
// ncclComm_t.  Two exchange patterns:
//   mode 0  one ncclAllGather (RCCL picks ring / tree);
//   mode 1  direct exchange: grouped ncclSend/ncclRecv with every peer, chunked.  An MI355X node is fully connected (7 xGMI links
//           per GPU, ~153 GB/s each, point to point): a ring is bounded by ONE link, the direct pattern drives all seven at once.
#include <dlfcn.h>
#include <string.h>

#include <mutex>

#include "gsf_internal.hpp"

using namespace gsf;

namespace {

typedef int (*allgather_fn)(const void*, void*, size_t, int, void*, hipStream_t);           // ncclAllGather
typedef int (*sendrecv_fn)(void*, size_t, int, int, void*, hipStream_t);                    // ncclSend / ncclRecv
typedef int (*group_fn)(void);
typedef int (*rank_fn)(void*, int*);
typedef const char* (*errstr_fn)(int);
struct UniqueId { char internal[128]; };                                                    // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef int (*uid_fn)(UniqueId*);
typedef int (*init_fn)(void**, int, UniqueId, int);                                         // ncclCommInitRank (id by value)
typedef int (*destroy_fn)(void*);
typedef int (*version_fn)(int*);                                                            // ncclGetVersion

struct Rccl {
    void* h = nullptr;
    allgather_fn allgather = nullptr; sendrecv_fn send = nullptr; sendrecv_fn recv = nullptr;
    group_fn gstart = nullptr, gend = nullptr; rank_fn crank = nullptr, csize = nullptr; errstr_fn errstr = nullptr;
    uid_fn uid = nullptr; init_fn init = nullptr; destroy_fn destroy = nullptr; version_fn version = nullptr;
    int ver = 0;                                                                             // e.g. 22105 for 2.21.5; 0 = unknown
    char why[256] = "";
    bool ok() const { return h && allgather && send && recv && gstart && gend && crank && csize && uid && init && destroy; }
};

Rccl* rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = { "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so" };
        (void)dlerror();
        for (const char* n : names) { r.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (r.h) break; }      // already mapped (torch)?
        if (!r.h) for (const char* n : names) { r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (r.h) break; }
        if (!r.h) { const char* e = dlerror(); snprintf(r.why, sizeof r.why, "%s", e ? e : "dlopen failed"); return; }
        r.allgather = (allgather_fn)dlsym(r.h, "ncclAllGather");
        r.send = (sendrecv_fn)dlsym(r.h, "ncclSend"); r.recv = (sendrecv_fn)dlsym(r.h, "ncclRecv");
        r.gstart = (group_fn)dlsym(r.h, "ncclGroupStart"); r.gend = (group_fn)dlsym(r.h, "ncclGroupEnd");
        r.crank = (rank_fn)dlsym(r.h, "ncclCommUserRank"); r.csize = (rank_fn)dlsym(r.h, "ncclCommCount");
        r.errstr = (errstr_fn)dlsym(r.h, "ncclGetErrorString");
        r.uid = (uid_fn)dlsym(r.h, "ncclGetUniqueId"); r.init = (init_fn)dlsym(r.h, "ncclCommInitRank");
        r.destroy = (destroy_fn)dlsym(r.h, "ncclCommDestroy");
        r.version = (version_fn)dlsym(r.h, "ncclGetVersion");
        if (r.version && r.version(&r.ver) != 0) r.ver = 0;
        if (!r.ok()) snprintf(r.why, sizeof r.why, "a required nccl* symbol is missing from librccl.so");
    });
    return &r;
}

constexpr int NCCL_FLOAT64 = 8;   // ncclDouble / ncclFloat64 in rccl.h

int need_rccl(Rccl*& R, const char* who)
{
    R = rccl();
    if (!R->ok()) { set_error("%s: librccl.so could not be resolved (%s)", who, R->why); return GSF_ERR_UNSUPPORTED; }
    return GSF_OK;
}

}  // namespace

#define GSF_NCCL(call)                                                                                     \
    do {                                                                                                   \
        int e__ = (call);                                                                                  \
        if (e__ != 0) { set_error("RCCL (version %d) error %d (%s) in %s", R->ver, e__, R->errstr ? R->errstr(e__) : "?", #call); return GSF_ERR_HIP; } \
    } while (0)

extern "C" {

// The resolved library's ncclGetVersion (MAJOR * 10000 + MINOR * 100 + PATCH): logged by the callers, so that a mismatch between the
// signatures assumed above (NCCL 2.x: 128-byte id by value) and the librccl.so of the process shows up as a number, not as a stall.
int gsf_comm_rccl_version(int32_t* version)
{
    GSF_REQUIRE(version, "version is NULL");
    Rccl* R; int rc = need_rccl(R, "gsf_comm_rccl_version");
    if (rc) return rc;
    *version = R->ver;
    return GSF_OK;
}

int gsf_comm_unique_id(uint8_t* id128)
{
    GSF_REQUIRE(id128, "id128 is NULL");
    Rccl* R; int rc = need_rccl(R, "gsf_comm_unique_id");
    if (rc) return rc;
    UniqueId u;
    GSF_NCCL(R->uid(&u));
    memcpy(id128, u.internal, sizeof u.internal);
    return GSF_OK;
}

int gsf_comm_init_rank(gsf_ctx* ctx, const uint8_t* id128, int32_t world, int32_t rank, void** comm)
{
    GSF_REQUIRE(ctx && id128 && comm, "NULL argument");
    GSF_REQUIRE(world >= 1 && rank >= 0 && rank < world, "rank/world out of range");
    *comm = nullptr;
    Rccl* R; int rc = need_rccl(R, "gsf_comm_init_rank");
    if (rc) return rc;
    GSF_REQUIRE(R->ver == 0 || (R->ver >= 20000 && R->ver < 30000), "the resolved librccl.so is not an NCCL 2.x API (ncclGetVersion): refusing to call it with 2.x signatures");
    GSF_HIP(hipSetDevice(ctx->device));
    UniqueId u;
    memcpy(u.internal, id128, sizeof u.internal);
    GSF_NCCL(R->init(comm, world, u, rank));
    return GSF_OK;
}

int gsf_comm_destroy(void* comm)
{
    if (!comm) return GSF_OK;
    Rccl* R; int rc = need_rccl(R, "gsf_comm_destroy");
    if (rc) return rc;
    GSF_NCCL(R->destroy(comm));
    return GSF_OK;
}

// All-gather `count` doubles per rank into recv[world][count] on the context's stream (see the modes at the top of the file).
int gsf_allgather_poses(gsf_ctx* ctx, void* nccl_comm, const double* send, double* recv, int64_t count, int32_t mode, int64_t chunk_count)
{
    GSF_REQUIRE(ctx && nccl_comm && send && recv && count >= 0, "bad arguments");
    GSF_REQUIRE(mode == 0 || mode == 1, "mode must be 0 (ncclAllGather) or 1 (direct send/recv)");
    Rccl* R; int rc = need_rccl(R, "gsf_allgather_poses");
    if (rc) return rc;
    if (count == 0) return GSF_OK;
    GSF_HIP(hipSetDevice(ctx->device));
    int rank = 0, world = 1;
    GSF_NCCL(R->crank(nccl_comm, &rank));
    GSF_NCCL(R->csize(nccl_comm, &world));
    if (mode == 0) {
        GSF_NCCL(R->allgather(send, recv, (size_t)count, NCCL_FLOAT64, nccl_comm, ctx->stream));
        return GSF_OK;
    }
    const int64_t step = chunk_count > 0 ? chunk_count : count;
    for (int64_t o = 0; o < count; o += step) {
        const int64_t n = (count - o < step) ? (count - o) : step;
        // this rank's own block: a device copy, overlapping the exchange
        GSF_HIP(hipMemcpyAsync(recv + (int64_t)rank * count + o, send + o, (size_t)n * 8, hipMemcpyDeviceToDevice, ctx->stream));
        GSF_NCCL(R->gstart());
        int err = 0; const char* what = "";
        // peers in rotated order (rank+1, rank+2, ...): at any instant every link has one sender and one receiver
        for (int d = 1; d < world && !err; ++d) {
            const int to = (rank + d) % world, from = (rank - d + world) % world;
            err = R->send((void*)(send + o), (size_t)n, NCCL_FLOAT64, to, nccl_comm, ctx->stream); what = "ncclSend";
            if (!err) { err = R->recv((void*)(recv + (int64_t)from * count + o), (size_t)n, NCCL_FLOAT64, from, nccl_comm, ctx->stream); what = "ncclRecv"; }
        }
        const int eend = R->gend();                                       // always close the group, also after a failed send/recv
        if (err || eend) {
            const int e = err ? err : eend;
            set_error("RCCL error %d (%s) in %s (direct exchange)", e, R->errstr ? R->errstr(e) : "?", err ? what : "ncclGroupEnd");
            return GSF_ERR_HIP;
        }
    }
    return GSF_OK;
}

}  // extern "C"
