// Data-parallel gradient reduction over RCCL / xGMI behind the C ABI (include/yat_hip.h, "communication" section).
// Replaces what Accelerate's DDP wrap does for the reference (common/trainer.py:31-37 Accelerator + DDP kwargs, :253
// prepare -> parameter broadcast, :344 backward -> bucketed all-reduce(avg) of the gradients).
//
// Thin by design: the library owns ONE communicator and one completion event per bucket, nothing else.  The all-reduce is
// enqueued on the caller's communication stream behind an event recorded on the stream that produced the bucket, so it
// overlaps whatever the producer stream enqueues next; yat_comm_wait makes a consumer stream wait for a bucket (or all).
// Which bucket is ready when is the caller's knowledge (the backward schedule lives in yat_amd/sana.py), so bucket
// *scheduling* stays there; this file is the transport.
//
// RCCL is bound at run time (dlopen): a process that already holds an RCCL (torch.distributed's) gets THAT copy, so there is
// one RCCL and one HIP runtime per process; a plain C caller gets the system librccl.so.1.  libyat_hip.so therefore loads
// on hosts without RCCL and yat_comm_init is the only call that can fail for its absence.
#include "common.hpp"
#include "../../include/yat_hip.h"
#include <dlfcn.h>
#include <link.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <string.h>

namespace {

struct Api {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclReduceScatter) ReduceScatter = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

constexpr int MAX_BUCKETS = 256;

struct Comm {
    Api api;
    ncclComm_t comm = nullptr;
    int rank = -1, world = 0;
    hipEvent_t ready[MAX_BUCKETS] = {};      // producer stream -> communication stream
    hipEvent_t done[MAX_BUCKETS] = {};       // communication stream -> consumers
    bool pending[MAX_BUCKETS] = {};
    char last_error[256] = "";
};

Comm g;   // the one piece of global state of the library: the communicator and its events

int find_loaded_rccl(struct dl_phdr_info* info, size_t, void* out) {
    if (info->dlpi_name && strstr(info->dlpi_name, "librccl.so")) {
        strncpy((char*)out, info->dlpi_name, 1023);
        return 1;
    }
    return 0;
}

int bind_rccl(Api& a) {
    if (a.handle) return YAT_OK;
    char path[1024] = "";
    dl_iterate_phdr(find_loaded_rccl, path);
    void* h = path[0] ? dlopen(path, RTLD_NOW | RTLD_LOCAL) : nullptr;
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        snprintf(g.last_error, sizeof g.last_error, "cannot load librccl: %s", dlerror());
        return YAT_ENOCOMM;
    }
#define YAT_BIND(field, sym)                                                            \
    a.field = (decltype(a.field))dlsym(h, sym);                                         \
    if (!a.field) {                                                                     \
        snprintf(g.last_error, sizeof g.last_error, "librccl lacks %s", sym);           \
        dlclose(h);                                                                     \
        return YAT_ENOCOMM;                                                             \
    }
    YAT_BIND(GetUniqueId, "ncclGetUniqueId")
    YAT_BIND(CommInitRank, "ncclCommInitRank")
    YAT_BIND(CommDestroy, "ncclCommDestroy")
    YAT_BIND(AllReduce, "ncclAllReduce")
    YAT_BIND(Broadcast, "ncclBroadcast")
    YAT_BIND(ReduceScatter, "ncclReduceScatter")
    YAT_BIND(AllGather, "ncclAllGather")
    YAT_BIND(GetErrorString, "ncclGetErrorString")
#undef YAT_BIND
    a.handle = h;
    return YAT_OK;
}

int nccl_rc(ncclResult_t r, const char* what) {
    if (r == ncclSuccess) return YAT_OK;
    snprintf(g.last_error, sizeof g.last_error, "%s: %s", what, g.api.GetErrorString ? g.api.GetErrorString(r) : "?");
    return YAT_ECOMM_BASE + (int)r;
}

int hip_rc(hipError_t e, const char* what) {
    if (e == hipSuccess) return YAT_OK;
    snprintf(g.last_error, sizeof g.last_error, "%s: %s", what, hipGetErrorString(e));
    return (int)e;
}

}  // namespace

extern "C" {

const char* yat_comm_last_error(void) { return g.last_error; }

int yat_comm_available(void) { return bind_rccl(g.api); }

int yat_comm_unique_id(void* id_out) {
    if (!id_out) return YAT_EINVAL;
    if (int rc = bind_rccl(g.api)) return rc;
    ncclUniqueId id;
    if (int rc = nccl_rc(g.api.GetUniqueId(&id), "ncclGetUniqueId")) return rc;
    static_assert(sizeof(id) == YAT_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(id_out, &id, sizeof id);
    return YAT_OK;
}

int yat_comm_init(int rank, int world, const void* unique_id) {
    if (world < 1 || rank < 0 || rank >= world || !unique_id) return YAT_EINVAL;
    if (g.comm) return YAT_EINVAL;                              // one communicator per process (one process per GPU)
    if (int rc = bind_rccl(g.api)) return rc;
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof id);
    ncclComm_t c = nullptr;
    if (int rc = nccl_rc(g.api.CommInitRank(&c, world, id, rank), "ncclCommInitRank")) return rc;
    g.comm = c;
    g.rank = rank;
    g.world = world;
    return YAT_OK;
}

int yat_comm_world(void) { return g.comm ? g.world : 0; }
int yat_comm_rank(void) { return g.comm ? g.rank : -1; }

int yat_comm_broadcast(void* ptr, uint64_t nbytes, int root, yat_stream_t stream) {
    if (!g.comm) return YAT_ENOCOMM;
    if (!ptr || !nbytes || root < 0 || root >= g.world) return YAT_EINVAL;
    return nccl_rc(g.api.Broadcast(ptr, ptr, nbytes, ncclUint8, root, g.comm, (hipStream_t)stream), "ncclBroadcast");
}

int yat_comm_allreduce(void* ptr, uint64_t count, int dtype, int op, yat_stream_t stream) {
    if (!g.comm) return YAT_ENOCOMM;
    if (!ptr || !count || dtype < 0 || dtype > 1 || op < 0 || op > 1) return YAT_EINVAL;
    return nccl_rc(g.api.AllReduce(ptr, ptr, count, dtype == 0 ? ncclBfloat16 : ncclFloat32, op == 0 ? ncclAvg : ncclSum, g.comm,
                                   (hipStream_t)stream), "ncclAllReduce");
}

int yat_bucket_allreduce_async(void* ptr, uint64_t nbytes, int bucket_id, yat_stream_t producer_stream,
                               yat_stream_t comm_stream) {
    if (!g.comm) return YAT_ENOCOMM;
    if (!ptr || !nbytes || (nbytes & 1) || bucket_id < 0 || bucket_id >= MAX_BUCKETS) return YAT_EINVAL;
    hipStream_t prod = (hipStream_t)producer_stream, cs = (hipStream_t)comm_stream;
    if (!g.done[bucket_id]) {
        if (int rc = hip_rc(hipEventCreateWithFlags(&g.ready[bucket_id], hipEventDisableTiming), "hipEventCreate")) return rc;
        if (int rc = hip_rc(hipEventCreateWithFlags(&g.done[bucket_id], hipEventDisableTiming), "hipEventCreate")) return rc;
    }
    if (prod != cs) {
        if (int rc = hip_rc(hipEventRecord(g.ready[bucket_id], prod), "hipEventRecord")) return rc;
        if (int rc = hip_rc(hipStreamWaitEvent(cs, g.ready[bucket_id], 0), "hipStreamWaitEvent")) return rc;
    }
    // gradients are bf16 (the parameters' dtype, train_sana.py:21-22); the mean over ranks is DDP's reduction
    if (int rc = nccl_rc(g.api.AllReduce(ptr, ptr, nbytes / 2, ncclBfloat16, ncclAvg, g.comm, cs), "ncclAllReduce")) return rc;
    if (int rc = hip_rc(hipEventRecord(g.done[bucket_id], cs), "hipEventRecord")) return rc;
    g.pending[bucket_id] = true;
    return YAT_OK;
}

// Sharded optimizer step (yat_amd/ddp.py ``shard_optimizer``): the bucket's gradients are REDUCE-SCATTERED instead of
// all-reduced -- rank r ends up with the mean of slice r of the bucket, in place (RCCL's in-place form: the receive buffer
// is the rank's own slice of the send buffer; the other slices keep this rank's local values) -- and the updated parameters
// come back by an in-place all-gather.  Same bytes on the wire as the all-reduce the pair replaces.
int yat_bucket_reduce_scatter_async(void* ptr, uint64_t nbytes, int bucket_id, yat_stream_t producer_stream,
                                    yat_stream_t comm_stream) {
    if (!g.comm) return YAT_ENOCOMM;
    if (!ptr || !nbytes || bucket_id < 0 || bucket_id >= MAX_BUCKETS || nbytes % ((uint64_t)16 * g.world)) return YAT_EINVAL;
    hipStream_t prod = (hipStream_t)producer_stream, cs = (hipStream_t)comm_stream;
    if (!g.done[bucket_id]) {
        if (int rc = hip_rc(hipEventCreateWithFlags(&g.ready[bucket_id], hipEventDisableTiming), "hipEventCreate")) return rc;
        if (int rc = hip_rc(hipEventCreateWithFlags(&g.done[bucket_id], hipEventDisableTiming), "hipEventCreate")) return rc;
    }
    if (prod != cs) {
        if (int rc = hip_rc(hipEventRecord(g.ready[bucket_id], prod), "hipEventRecord")) return rc;
        if (int rc = hip_rc(hipStreamWaitEvent(cs, g.ready[bucket_id], 0), "hipStreamWaitEvent")) return rc;
    }
    const uint64_t slice = nbytes / g.world;
    if (int rc = nccl_rc(g.api.ReduceScatter(ptr, (char*)ptr + slice * g.rank, slice / 2, ncclBfloat16, ncclAvg, g.comm, cs),
                         "ncclReduceScatter")) return rc;
    if (int rc = hip_rc(hipEventRecord(g.done[bucket_id], cs), "hipEventRecord")) return rc;
    g.pending[bucket_id] = true;
    return YAT_OK;
}

int yat_comm_allgather(void* ptr, uint64_t nbytes, yat_stream_t stream) {
    if (!g.comm) return YAT_ENOCOMM;
    if (!ptr || !nbytes || nbytes % ((uint64_t)16 * g.world)) return YAT_EINVAL;
    const uint64_t slice = nbytes / g.world;
    return nccl_rc(g.api.AllGather((char*)ptr + slice * g.rank, ptr, slice, ncclUint8, g.comm, (hipStream_t)stream), "ncclAllGather");
}

int yat_comm_wait(int bucket_id, yat_stream_t compute_stream) {
    if (!g.comm) return YAT_ENOCOMM;
    if (bucket_id >= MAX_BUCKETS) return YAT_EINVAL;
    const int lo = bucket_id < 0 ? 0 : bucket_id, hi = bucket_id < 0 ? MAX_BUCKETS : bucket_id + 1;
    for (int b = lo; b < hi; ++b) {
        if (!g.pending[b]) continue;
        if (int rc = hip_rc(hipStreamWaitEvent((hipStream_t)compute_stream, g.done[b], 0), "hipStreamWaitEvent")) return rc;
        g.pending[b] = false;
    }
    return YAT_OK;
}

int yat_comm_destroy(void) {
    if (!g.comm) return YAT_OK;
    for (int b = 0; b < MAX_BUCKETS; ++b) {
        if (g.done[b]) { (void)hipEventDestroy(g.ready[b]); (void)hipEventDestroy(g.done[b]); }
        g.ready[b] = g.done[b] = nullptr;
        g.pending[b] = false;
    }
    const int rc = nccl_rc(g.api.CommDestroy(g.comm), "ncclCommDestroy");
    g.comm = nullptr;
    g.rank = -1;
    g.world = 0;
    return rc;
}

}  // extern "C"
