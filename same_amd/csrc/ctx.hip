// ctx.hip -- context lifetime, device memory, timing and error strings of libsame_hip.
#include "common.h"

extern "C" {

int same_abi_version(void) { return SAME_ABI_VERSION; }

int same_device_count(int *out_count) {
    if (!out_count) return SAME_EINVAL;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *out_count = 0; return SAME_ENODEV; }
    *out_count = n;
    return SAME_OK;
}

int same_ctx_create(int device, same_ctx **out) {
    if (!out) return SAME_EINVAL;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return SAME_ENODEV;
    if (device < 0 || device >= n) return SAME_EINVAL;
    same_ctx *ctx = new (std::nothrow) same_ctx();
    if (!ctx) return SAME_ENOMEM;
    ctx->device = device;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    // the gather overlaps a dense build that keeps every CU busy: give its stream the highest queue priority so the
    // collective's few workgroups are dispatched as CUs free up instead of queueing behind ~77k dense blocks
    int least = 0, greatest = 0;
    if (e == hipSuccess) e = hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&ctx->comm_stream, hipStreamNonBlocking, greatest);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ev_ready, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev_gathered);   // timing-enabled: the end stamp of same_comm_gather_time
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev_gather0);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev0);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev1);
    if (e == hipSuccess) { ctx->pinned_bytes = 1 << 16; e = hipHostMalloc(&ctx->pinned, ctx->pinned_bytes, hipHostMallocDefault); }
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) { same_ctx_destroy(ctx); return SAME_EIO; }
    ctx->cu_count = prop.multiProcessorCount;
    *out = ctx;
    return SAME_OK;
}

void same_ctx_destroy(same_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->comm) same_comm_destroy(ctx);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->comm_stream) (void)hipStreamSynchronize(ctx->comm_stream);
    for (int s = 0; s < SL_COUNT; ++s)
        if (ctx->slot[s]) (void)hipFree(ctx->slot[s]);
    for (auto &a : ctx->spread) same_spread_release(a);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->ev_ready) (void)hipEventDestroy(ctx->ev_ready);
    if (ctx->ev_gathered) (void)hipEventDestroy(ctx->ev_gathered);
    if (ctx->ev_gather0) (void)hipEventDestroy(ctx->ev_gather0);
    if (ctx->comm_stream) (void)hipStreamDestroy(ctx->comm_stream);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int same_ctx_sync(same_ctx *ctx) {
    REQUIRE(ctx, ctx != nullptr);
    SAME_TRY(same_use(ctx));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->comm_stream));
    return SAME_OK;
}

const char *same_strerror(int code) {
    switch (code) {
        case SAME_OK: return "ok";
        case SAME_EINVAL: return "invalid argument (shape, NULL pointer or unsupported size)";
        case SAME_ENOMEM: return "out of memory";
        case SAME_EIO: return "HIP/RCCL runtime failure";
        case SAME_ENODEV: return "no usable GPU device";
        case SAME_ERANGE: return "index out of range";
        case SAME_EUNSURE: return "points too close to degenerate to answer for Qhull (not an error: ask Qhull)";
        default: return "unknown error";
    }
}

const char *same_last_error(same_ctx *ctx) { return ctx ? ctx->err.c_str() : ""; }

int same_ctx_info(same_ctx *ctx, char *name, size_t name_len, int *cu_count, int64_t *hbm_bytes) {
    REQUIRE(ctx, ctx != nullptr);
    hipDeviceProp_t prop;
    HIP_TRY(ctx, hipGetDeviceProperties(&prop, ctx->device));
    if (name && name_len) { strncpy(name, prop.gcnArchName, name_len - 1); name[name_len - 1] = 0; }
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    return SAME_OK;
}

int same_ctx_pci_bus_id(same_ctx *ctx, char *out, size_t out_len) {
    REQUIRE(ctx, ctx && out && out_len >= 16 && out_len <= 4096);
    HIP_TRY(ctx, hipDeviceGetPCIBusId(out, (int)out_len, ctx->device));
    return SAME_OK;
}

int same_ctx_stat(same_ctx *ctx, int which, int64_t *out) {
    REQUIRE(ctx, ctx && out && which >= 0 && which < SAME_STAT_COUNT);
    *out = ctx->stats[which];
    return SAME_OK;
}

int same_dev_alloc(same_ctx *ctx, size_t bytes, void **out_dptr) {
    REQUIRE(ctx, ctx && out_dptr);
    SAME_TRY(same_use(ctx));
    *out_dptr = nullptr;
    HIP_TRY(ctx, hipMalloc(out_dptr, bytes ? bytes : 1));
    return SAME_OK;
}

int same_dev_free(same_ctx *ctx, void *dptr) {
    REQUIRE(ctx, ctx != nullptr);
    SAME_TRY(same_use(ctx));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < ctx->spread.size(); ++i)
        if (ctx->spread[i].va == dptr) {   // a same_dev_alloc_spread buffer
            same_spread_release(ctx->spread[i]);
            ctx->spread.erase(ctx->spread.begin() + i);
            return SAME_OK;
        }
    if (dptr) HIP_TRY(ctx, hipFree(dptr));
    return SAME_OK;
}

int same_h2d(same_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes) {
    REQUIRE(ctx, ctx && (bytes == 0 || (dst_dev && src_host)));
    SAME_TRY(same_use(ctx));
    if (bytes) HIP_TRY(ctx, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SAME_OK;
}

int same_d2h(same_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes) {
    REQUIRE(ctx, ctx && (bytes == 0 || (dst_host && src_dev)));
    SAME_TRY(same_use(ctx));
    if (bytes) HIP_TRY(ctx, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SAME_OK;
}

int same_dev_mem_info(same_ctx *ctx, int64_t *out_free, int64_t *out_total) {
    REQUIRE(ctx, ctx != nullptr);
    SAME_TRY(same_use(ctx));
    size_t f = 0, t = 0;
    HIP_TRY(ctx, hipMemGetInfo(&f, &t));
    if (out_free) *out_free = (int64_t)f;
    if (out_total) *out_total = (int64_t)t;
    return SAME_OK;
}

int same_dev_memset(same_ctx *ctx, void *dst_dev, int value, size_t bytes) {
    REQUIRE(ctx, ctx && (bytes == 0 || dst_dev));
    SAME_TRY(same_use(ctx));
    if (bytes) HIP_TRY(ctx, hipMemsetAsync(dst_dev, value, bytes, ctx->stream));
    return SAME_OK;
}

// device-to-device copy on the context's stream (the measured copy bandwidth bench.py reports beside the HBM spec)
int same_d2d(same_ctx *ctx, void *dst_dev, const void *src_dev, size_t bytes) {
    REQUIRE(ctx, ctx && (bytes == 0 || (dst_dev && src_dev)));
    SAME_TRY(same_use(ctx));
    if (bytes) HIP_TRY(ctx, hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return SAME_OK;
}

int same_ctx_release_scratch(same_ctx *ctx) {
    REQUIRE(ctx, ctx != nullptr);
    SAME_TRY(same_use(ctx));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->comm_stream));
    for (int s = 0; s < SL_COUNT; ++s)
        if (ctx->slot[s]) {
            HIP_TRY(ctx, hipFree(ctx->slot[s]));
            ctx->slot[s] = nullptr;
            ctx->slot_bytes[s] = 0;
        }
    return SAME_OK;
}

int same_timer_start(same_ctx *ctx) {
    REQUIRE(ctx, ctx != nullptr);
    SAME_TRY(same_use(ctx));
    HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    return SAME_OK;
}

int same_timer_stop(same_ctx *ctx, float *out_ms) {
    REQUIRE(ctx, ctx && out_ms);
    SAME_TRY(same_use(ctx));
    HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev1));
    HIP_TRY(ctx, hipEventElapsedTime(out_ms, ctx->ev0, ctx->ev1));
    return SAME_OK;
}

// split form of same_timer_stop: mark the end on the stream now, read the elapsed time later (the host can keep
// enqueueing work -- on this or another context -- while the timed kernel runs)
int same_timer_mark(same_ctx *ctx) {
    REQUIRE(ctx, ctx != nullptr);
    SAME_TRY(same_use(ctx));
    HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    return SAME_OK;
}

int same_timer_read(same_ctx *ctx, float *out_ms) {
    REQUIRE(ctx, ctx && out_ms);
    SAME_TRY(same_use(ctx));
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev1));
    HIP_TRY(ctx, hipEventElapsedTime(out_ms, ctx->ev0, ctx->ev1));
    return SAME_OK;
}

}  // extern "C"

int same_use(same_ctx *ctx) {
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return SAME_OK;
}

int same_slot(same_ctx *ctx, Slot s, size_t bytes, void **out) {
    if (bytes == 0) bytes = 16;
    if (ctx->slot_bytes[s] < bytes) {
        // the old block may still be in use by queued work
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->slot[s]) { HIP_TRY(ctx, hipFree(ctx->slot[s])); ctx->slot[s] = nullptr; ctx->slot_bytes[s] = 0; }
        size_t want = bytes + bytes / 4;  // headroom so slowly growing inputs do not realloc every call
        want = (want + 255) & ~size_t(255);
        HIP_TRY(ctx, hipMalloc(&ctx->slot[s], want));
        ctx->slot_bytes[s] = want;
    }
    *out = ctx->slot[s];
    return SAME_OK;
}

int same_up(same_ctx *ctx, Slot s, const void *host, size_t bytes, void **out) {
    SAME_TRY(same_slot(ctx, s, bytes, out));
    if (bytes) HIP_TRY(ctx, hipMemcpyAsync(*out, host, bytes, hipMemcpyHostToDevice, ctx->stream));
    return SAME_OK;
}

int same_down(same_ctx *ctx, void *host, const void *dev, size_t bytes) {
    if (bytes) HIP_TRY(ctx, hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return SAME_OK;
}

int check_index_range(same_ctx *ctx, const int32_t *idx, int64_t n, int64_t lo, int64_t hi, const char *what) {
    for (int64_t q = 0; q < n; ++q) {
        if (idx[q] < lo || idx[q] >= hi) {
            char buf[256];
            snprintf(buf, sizeof buf, "%s[%lld] = %d outside [%lld, %lld)", what, (long long)q, idx[q], (long long)lo, (long long)hi);
            ctx->err = buf;
            return SAME_ERANGE;
        }
    }
    return SAME_OK;
}
