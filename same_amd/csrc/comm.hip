// comm.hip -- the path's only exchange step: an RCCL all-gather of the fixed-width pruned
// candidate lists (int32 idx[rows][k], double cost[rows][k]) across the ranks that own
// aligned-row blocks.  One process per GPU; the unique id travels over whatever
// host channel the launcher has (same_amd/rendezvous.py: loopback TCP, plain Python).  The sharded sweeps add
// an all-gather of per-triangle flags and a small all-reduce of their counters.
#include <rccl/rccl.h>

#include "common.h"

static int nccl_fail(same_ctx *ctx, const char *what, ncclResult_t r) {
    if (ctx) ctx->err = std::string(what) + ": " + ncclGetErrorString(r);
    return SAME_EIO;
}
#define NCCL_TRY(ctx, call)                                   \
    do {                                                      \
        ncclResult_t r_ = (call);                             \
        if (r_ != ncclSuccess) return nccl_fail((ctx), #call, r_); \
    } while (0)

extern "C" {

int same_comm_unique_id(char out_id[SAME_UNIQUE_ID_BYTES]) {
    static_assert(sizeof(ncclUniqueId) == SAME_UNIQUE_ID_BYTES, "ncclUniqueId size");
    if (!out_id) return SAME_EINVAL;
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return SAME_EIO;
    memcpy(out_id, &id, sizeof id);
    return SAME_OK;
}

int same_comm_init(same_ctx *ctx, int nranks, int rank, const char id[SAME_UNIQUE_ID_BYTES]) {
    REQUIRE(ctx, ctx && id && nranks >= 1 && rank >= 0 && rank < nranks && ctx->comm == nullptr);
    SAME_TRY(same_use(ctx));
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    ncclComm_t comm = nullptr;
    NCCL_TRY(ctx, ncclCommInitRank(&comm, nranks, uid, rank));
    ctx->comm = comm;
    ctx->nranks = nranks;
    ctx->rank = rank;
    return SAME_OK;
}

int same_comm_destroy(same_ctx *ctx) {
    REQUIRE(ctx, ctx != nullptr);
    if (ctx->comm) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamSynchronize(ctx->comm_stream);
        ncclCommDestroy(ctx->comm);
        ctx->comm = nullptr;
        ctx->nranks = 1;
        ctx->rank = 0;
    }
    return SAME_OK;
}

int same_allgather_dev(same_ctx *ctx, const void *dsend, void *drecv, size_t send_bytes) {
    REQUIRE(ctx, ctx && ctx->comm && (send_bytes == 0 || (dsend && drecv)));
    SAME_TRY(same_use(ctx));
    if (send_bytes == 0) return SAME_OK;
    // outside an RCCL group the gather is stamped like the overlapped form (same_comm_gather_time reads it); inside a group
    // nothing is enqueued before same_comm_group_end, so stamps recorded here would bracket nothing
    const bool stamp = !ctx->in_group;
    if (stamp && !ctx->gather_open) {
        HIP_TRY(ctx, hipEventRecord(ctx->ev_gather0, ctx->stream));
        ctx->gather_open = true;
        ctx->gather_bytes = 0;
    }
    NCCL_TRY(ctx, ncclAllGather(dsend, drecv, send_bytes, ncclInt8, ctx->comm, ctx->stream));
    if (stamp) {
        HIP_TRY(ctx, hipEventRecord(ctx->ev_gathered, ctx->stream));
        ctx->gather_bytes += send_bytes;
        ctx->gather_stamped = true;
    }
    return SAME_OK;
}

// Overlapped form.  The gather is enqueued on the context's communication stream behind everything
// queued so far on the compute stream (so the send buffers are complete), and returns at once: compute
// queued afterwards (the next dense build) runs concurrently with it.  Before anything overwrites the
// send buffers or reads the gathered ones, call same_comm_wait: it makes the compute stream wait for
// every gather issued so far (a stream-side wait, the host does not block).
int same_allgather_dev_async(same_ctx *ctx, const void *dsend, void *drecv, size_t send_bytes) {
    REQUIRE(ctx, ctx && ctx->comm && (send_bytes == 0 || (dsend && drecv)));
    SAME_TRY(same_use(ctx));
    if (send_bytes == 0) return SAME_OK;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_ready, ctx->stream));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->comm_stream, ctx->ev_ready, 0));
    if (!ctx->gather_open) {   // first gather since the last same_comm_wait: the start stamp of same_comm_gather_time
        HIP_TRY(ctx, hipEventRecord(ctx->ev_gather0, ctx->comm_stream));
        ctx->gather_open = true;
        ctx->gather_bytes = 0;
    }
    NCCL_TRY(ctx, ncclAllGather(dsend, drecv, send_bytes, ncclInt8, ctx->comm, ctx->comm_stream));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_gathered, ctx->comm_stream));
    ctx->gather_bytes += send_bytes;
    ctx->gather_stamped = true;
    return SAME_OK;
}

// In-place all-reduce on the compute stream: sweep counters (u64 sum), point flags (u8 max = OR), timing (f64 max).
int same_allreduce_dev(same_ctx *ctx, void *dbuf, size_t count, int dtype, int op) {
    REQUIRE(ctx, ctx && ctx->comm && (count == 0 || dbuf));
    ncclDataType_t dt;
    switch (dtype) {
        case SAME_DT_U8: dt = ncclUint8; break;
        case SAME_DT_I32: dt = ncclInt32; break;
        case SAME_DT_U64: dt = ncclUint64; break;
        case SAME_DT_F64: dt = ncclFloat64; break;
        default: ctx->err = "unknown SAME_DT_* code"; return SAME_EINVAL;
    }
    ncclRedOp_t ro;
    switch (op) {
        case SAME_OP_SUM: ro = ncclSum; break;
        case SAME_OP_MAX: ro = ncclMax; break;
        case SAME_OP_MIN: ro = ncclMin; break;
        default: ctx->err = "unknown SAME_OP_* code"; return SAME_EINVAL;
    }
    SAME_TRY(same_use(ctx));
    if (count == 0) return SAME_OK;
    NCCL_TRY(ctx, ncclAllReduce(dbuf, dbuf, count, dt, ro, ctx->comm, ctx->stream));
    return SAME_OK;
}

// Several collectives issued between these two calls are handed to RCCL as one group (one fused launch instead of one
// launch per array): the seven per-triangle arrays of the sharded sweeps travel this way.
int same_comm_group_start(same_ctx *ctx) {
    REQUIRE(ctx, ctx && ctx->comm);
    NCCL_TRY(ctx, ncclGroupStart());
    ctx->in_group = true;
    return SAME_OK;
}

int same_comm_group_end(same_ctx *ctx) {
    REQUIRE(ctx, ctx && ctx->comm);
    ctx->in_group = false;
    NCCL_TRY(ctx, ncclGroupEnd());
    return SAME_OK;
}

// What the COMMUNICATOR says about itself (ncclCommCount / ncclCommUserRank), not what same_comm_init was told: a
// launcher that hands every rank world = 1, or ranks that each built a communicator of their own, show up here.
int same_comm_info(same_ctx *ctx, int *out_nranks, int *out_rank, int *out_rccl_version) {
    REQUIRE(ctx, ctx != nullptr);
    int n = 0, r = 0;
    if (ctx->comm) {
        NCCL_TRY(ctx, ncclCommCount(ctx->comm, &n));
        NCCL_TRY(ctx, ncclCommUserRank(ctx->comm, &r));
    }
    if (out_nranks) *out_nranks = n;
    if (out_rank) *out_rank = r;
    if (out_rccl_version) {
        int v = 0;
        if (ncclGetVersion(&v) != ncclSuccess) v = 0;
        *out_rccl_version = v;
    }
    return SAME_OK;
}

int same_comm_wait(same_ctx *ctx) {
    REQUIRE(ctx, ctx != nullptr);
    SAME_TRY(same_use(ctx));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_gathered, 0));
    ctx->gather_open = false;
    return SAME_OK;
}

// Device time of the gathers issued since the last same_comm_wait (or the last call of this function): from the moment the communication stream
// was released by the producer (its first gather could start) to the end of its last gather, HIP events on that stream.
// Blocks the host until that last gather has finished.
int same_comm_gather_time(same_ctx *ctx, float *out_ms, int64_t *out_send_bytes) {
    REQUIRE(ctx, ctx && out_ms);
    SAME_TRY(same_use(ctx));
    *out_ms = 0.0f;
    if (out_send_bytes) *out_send_bytes = (int64_t)ctx->gather_bytes;
    if (!ctx->gather_stamped) return SAME_OK;   // no gather issued yet
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev_gathered));
    HIP_TRY(ctx, hipEventElapsedTime(out_ms, ctx->ev_gather0, ctx->ev_gathered));
    ctx->gather_open = false;   // a read closes the batch too: callers of the in-stream form alone never call same_comm_wait, and
                                // their next gather must not be timed from the first one ever issued
    return SAME_OK;
}

// Device the communicator was created on (ncclCommCuDevice), -1 without a communicator.
int same_comm_device(same_ctx *ctx, int *out_device) {
    REQUIRE(ctx, ctx && out_device);
    *out_device = -1;
    if (ctx->comm) NCCL_TRY(ctx, ncclCommCuDevice(ctx->comm, out_device));
    return SAME_OK;
}

}  // extern "C"
