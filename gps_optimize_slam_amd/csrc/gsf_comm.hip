// gsf_comm.hip -- the ONE collective of the path: collecting the fused poses of every rank with RCCL over xGMI.
// RCCL is resolved at run time (dlopen/dlsym) so that libgsf.so has no link-time dependency on a particular librccl.so: in a
// PyTorch process the already-mapped copy (torch bundles its own) is reused, a native caller gets /opt/rocm's.
#include <dlfcn.h>

#include "gsf_internal.hpp"

using namespace gsf;

namespace {

typedef int (*allgather_fn)(const void*, void*, size_t, int, void*, hipStream_t);           // ncclAllGather
typedef int (*sendrecv_fn)(void*, size_t, int, int, void*, hipStream_t);                    // ncclSend / ncclRecv
typedef int (*group_fn)(void);
typedef int (*rank_fn)(void*, int*);
typedef const char* (*errstr_fn)(int);

struct Rccl {
    void* h = nullptr;
    allgather_fn allgather = nullptr; sendrecv_fn send = nullptr; sendrecv_fn recv = nullptr;
    group_fn gstart = nullptr, gend = nullptr; rank_fn crank = nullptr, csize = nullptr; errstr_fn errstr = nullptr;
};

Rccl* rccl()
{
    static Rccl r;
    static bool tried = false;
    if (!tried) {
        tried = true;
        const char* names[] = { "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so" };
        for (const char* n : names) { r.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (r.h) break; }      // already mapped (torch)?
        if (!r.h) for (const char* n : names) { r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (r.h) break; }
        if (r.h) {
            r.allgather = (allgather_fn)dlsym(r.h, "ncclAllGather");
            r.send = (sendrecv_fn)dlsym(r.h, "ncclSend"); r.recv = (sendrecv_fn)dlsym(r.h, "ncclRecv");
            r.gstart = (group_fn)dlsym(r.h, "ncclGroupStart"); r.gend = (group_fn)dlsym(r.h, "ncclGroupEnd");
            r.crank = (rank_fn)dlsym(r.h, "ncclCommUserRank"); r.csize = (rank_fn)dlsym(r.h, "ncclCommCount");
            r.errstr = (errstr_fn)dlsym(r.h, "ncclGetErrorString");
        }
    }
    return &r;
}

constexpr int NCCL_FLOAT64 = 8;   // ncclDouble / ncclFloat64 in rccl.h

}  // namespace

#define GSF_NCCL(call)                                                                                     \
    do {                                                                                                   \
        int e__ = (call);                                                                                  \
        if (e__ != 0) { set_error("RCCL error %d (%s) in %s", e__, R->errstr ? R->errstr(e__) : "?", #call); return GSF_ERR_HIP; } \
    } while (0)

// All-gather `count` doubles per rank into recv[world][count] on the context's stream.
//   mode 0: one ncclAllGather (ring/tree as RCCL picks);
//   mode 1: direct exchange -- grouped ncclSend/ncclRecv with every peer in chunks of `chunk_count` doubles, so the seven
//           point-to-point xGMI links of a GPU carry traffic concurrently and the in-flight size stays bounded.
extern "C" int gsf_allgather_poses(gsf_ctx* ctx, void* nccl_comm, const double* send, double* recv, int64_t count, int32_t mode,
                                   int64_t chunk_count)
{
    GSF_REQUIRE(ctx && nccl_comm && send && recv && count >= 0, "bad arguments");
    GSF_REQUIRE(mode == 0 || mode == 1, "mode must be 0 (ncclAllGather) or 1 (direct send/recv)");
    Rccl* R = rccl();
    if (!R->h || !R->allgather || !R->send || !R->recv || !R->gstart || !R->gend || !R->crank || !R->csize) {
        set_error("gsf_allgather_poses: librccl.so could not be resolved (%s)", dlerror());
        return GSF_ERR_UNSUPPORTED;
    }
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
        GSF_NCCL(R->gstart());
        for (int p = 0; p < world; ++p) {
            if (p == rank) continue;
            GSF_NCCL(R->send((void*)(send + o), (size_t)n, NCCL_FLOAT64, p, nccl_comm, ctx->stream));
            GSF_NCCL(R->recv((void*)(recv + (int64_t)p * count + o), (size_t)n, NCCL_FLOAT64, p, nccl_comm, ctx->stream));
        }
        GSF_NCCL(R->gend());
        GSF_HIP(hipMemcpyAsync(recv + (int64_t)rank * count + o, send + o, (size_t)n * 8, hipMemcpyDeviceToDevice, ctx->stream));
    }
    return GSF_OK;
}
