// common.h -- context, error plumbing and scratch arena shared by the libsame_hip translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "same_hip_diag.h"

struct ncclComm;

// Named device scratch slots: each grows on demand and is reused across calls, so the
// host-buffer entry points do not hipMalloc per call.
enum Slot {
    SL_A = 0, SL_R, SL_AXY, SL_RXY, SL_PAIRS, SL_OUT0, SL_OUT1, SL_OUT2, SL_TRIS, SL_MATCH,
    SL_SIGN, SL_SIZE, SL_TYPE, SL_FLAG0, SL_FLAG1, SL_FLAG2, SL_COUNTS, SL_X, SL_MASK,
    // uniform-grid index of the reference cells built per call by the un-indexed prune entry points (knn.hip)
    SL_K_HIST, SL_K_RANK, SL_K_SXY, SL_K_SIDX, SL_K_BBOX, SL_K_START,
    SL_COUNT
};

// A buffer built by same_dev_alloc_spread (spread.hip): 1 GiB physical chunks mapped into one address range.
struct same_spread_alloc {
    char *va = nullptr;
    size_t bytes = 0;
    std::vector<void *> handles;   // hipMemGenericAllocationHandle_t of each GiB, in address order
};
void same_spread_release(same_spread_alloc &a);

struct same_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    hipStream_t comm_stream = nullptr;   // all-gather runs here so it can overlap compute on `stream`
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_ready = nullptr, ev_gathered = nullptr;  // producer-done / gather-done hand-offs between the two streams
    hipEvent_t ev_gather0 = nullptr;     // start stamp of the gathers issued since the last same_comm_wait (same_comm_gather_time)
    bool gather_open = false, gather_stamped = false, in_group = false;
    size_t gather_bytes = 0;
    void *slot[SL_COUNT] = {};
    size_t slot_bytes[SL_COUNT] = {};
    void *pinned = nullptr;  // small pinned staging block for scalar results
    size_t pinned_bytes = 0;
    std::string err;
    int cu_count = 0;
    ncclComm *comm = nullptr;
    int nranks = 1, rank = 0;
    std::vector<same_spread_alloc> spread;
    int64_t stats[SAME_STAT_COUNT] = {};   // what the library itself asked of the runtime on this context (same_ctx_stat)
};

// every kernel launch / fill / copy / wait of the window path and of the cores it shares goes through these, so that the
// per-window counts a profile shows can also be read from the library (same_ctx_stat) and held by a test
#define SAME_LAUNCH(ctx, kernel, grid, block, lds, ...)                                  \
    do {                                                                                 \
        ++(ctx)->stats[SAME_STAT_LAUNCHES];                                              \
        hipLaunchKernelGGL(kernel, grid, block, lds, (ctx)->stream, __VA_ARGS__);        \
    } while (0)
#define SAME_FILL(ctx, ptr, value, bytes)                                                \
    do {                                                                                 \
        ++(ctx)->stats[SAME_STAT_FILLS];                                                 \
        HIP_TRY((ctx), hipMemsetAsync((ptr), (value), (bytes), (ctx)->stream));          \
    } while (0)
#define SAME_COPY(ctx, dst, src, bytes, kind)                                            \
    do {                                                                                 \
        ++(ctx)->stats[SAME_STAT_COPIES];                                                \
        HIP_TRY((ctx), hipMemcpyAsync((dst), (src), (bytes), (kind), (ctx)->stream));    \
    } while (0)
#define SAME_WAIT(ctx)                                                                   \
    do {                                                                                 \
        ++(ctx)->stats[SAME_STAT_WAITS];                                                 \
        HIP_TRY((ctx), hipStreamSynchronize((ctx)->stream));                             \
    } while (0)

// Resident state of the lazy-constraint orientation sweep (same_sweep_bind): its own device blocks, so
// several sweeps can live on one context and none can be run against another's shapes.
struct same_sweep {
    same_ctx *ctx = nullptr;
    int64_t Tr = 0, n_r = 0, n_m = 0, P = 0;
    int32_t *tris = nullptr;      // [Tr][3]
    int8_t *sign = nullptr;       // [Tr]
    double *rxy = nullptr;        // [n_r][2]
    int32_t *pairs = nullptr;     // [P][2]
    int32_t *match = nullptr, *pidx = nullptr;  // [n_m]
    uint8_t *flag = nullptr;      // [Tr rounded up to 256]
    // one block: scan words | cnt[2] = {checked, flipped} | viol[Tr] -- the words and counters are zeroed by one fill, the counters and
    // the head of the ascending flipped list come back in ONE device-to-host copy
    unsigned long long *scan_base = nullptr;
    size_t scan_zero_bytes = 0;
    unsigned long long *cnt = nullptr;
    int32_t *viol = nullptr;      // = (int32_t *)(cnt + 2)
    double *x = nullptr;          // [P]
};

inline int same_fail(same_ctx *ctx, int code, const char *what, hipError_t e) {
    (void)hipGetLastError();  // HIP keeps the failure as the thread's "last error": clear it, or a later
                              // hipGetLastError() after a kernel launch would report this stale one
    if (ctx) {
        char buf[512];
        snprintf(buf, sizeof buf, "%s: %s (%d)", what, hipGetErrorString(e), (int)e);
        ctx->err = buf;
    }
    return code;
}

#define HIP_TRY(ctx, call)                                                    \
    do {                                                                      \
        hipError_t e_ = (call);                                               \
        if (e_ != hipSuccess) return same_fail((ctx), e_ == hipErrorOutOfMemory ? SAME_ENOMEM : SAME_EIO, #call, e_); \
    } while (0)

#define SAME_TRY(call)            \
    do {                          \
        int rc_ = (call);         \
        if (rc_ != SAME_OK) return rc_; \
    } while (0)

#define REQUIRE(ctx, cond)                                           \
    do {                                                             \
        if (!(cond)) {                                               \
            if (ctx) (ctx)->err = "invalid argument: " #cond;        \
            return SAME_EINVAL;                                      \
        }                                                            \
    } while (0)

int same_use(same_ctx *ctx);                                  // hipSetDevice(ctx->device)
int same_slot(same_ctx *ctx, Slot s, size_t bytes, void **out);  // grow-on-demand scratch
int same_up(same_ctx *ctx, Slot s, const void *host, size_t bytes, void **out);  // slot + async H2D
int same_down(same_ctx *ctx, void *host, const void *dev, size_t bytes);          // async D2H

template <typename T>
inline int slot_as(same_ctx *ctx, Slot s, size_t n, T **out) {
    void *p = nullptr;
    int rc = same_slot(ctx, s, n * sizeof(T), &p);
    *out = static_cast<T *>(p);
    return rc;
}
template <typename T>
inline int up_as(same_ctx *ctx, Slot s, const T *host, size_t n, T **out) {
    void *p = nullptr;
    int rc = same_up(ctx, s, host, n * sizeof(T), &p);
    *out = static_cast<T *>(p);
    return rc;
}

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Host-side index validation: a bad index must come back as SAME_ERANGE, never as a GPU fault.
int check_index_range(same_ctx *ctx, const int32_t *idx, int64_t n, int64_t lo, int64_t hi, const char *what);

// Device cores shared between the host-buffer entry points and the window pipeline (window.hip): every pointer is a device
// pointer, the calls enqueue on ctx->stream (same_greedy_core reads one counter back per round).
int same_pair_rowmin_core(same_ctx *ctx, const int32_t *dpairs, const double *dcosts, int64_t P, int64_t n_m, double *dout);
int same_greedy_core(same_ctx *ctx, const int32_t *dpairs, const double *dcosts, int64_t P, int64_t n_m, int64_t n_r,
                     const uint8_t *dprefer, int32_t *dmatch_pair, int *out_rounds);

// State of the greedy rule between rounds (match.hip): all of it zero before round 0 except `alive`.
constexpr int SAME_GREEDY_BATCH_MAX = 64;
struct same_greedy_state {
    uint8_t *alive = nullptr;                       // [P] pair still in play
    uint8_t *used = nullptr;                        // [n_m + n_r] end point taken: rows, then columns
    unsigned long long *key[2] = {nullptr, nullptr};   // [n_m + n_r] each: inverted minimum cost key per end point, alternating sets
    unsigned *idx[2] = {nullptr, nullptr};          // [n_m + n_r] each: inverted pair index among equal minima
    unsigned long long *sel = nullptr;              // [rounds of one batch] 1 = the round selected a pair
    bool float_costs = false;                       // every cost is exactly a float: (cost, pair index) packs into one word, 2 launches a round
};
int same_greedy_rounds_core(same_ctx *ctx, const int32_t *dpairs, const double *dcosts, int64_t P, const unsigned long long *dP,
                            int64_t n_m, int64_t n_r, const same_greedy_state &st, int32_t *dmatch_pair, int first, int count);

// Several independent problems per launch (the windows of a batch call): blockIdx.y = problem, the per-problem arguments travel by
// value in the kernarg segment -- hence the small bound.
constexpr int SAME_LAUNCH_WINDOWS = 8;
struct same_greedy_job {
    const int32_t *pairs = nullptr;
    const double *costs = nullptr;
    int64_t P = 0, n_m = 0, n_r = 0;
    same_greedy_state st;
    int32_t *match_pair = nullptr;
};
int same_greedy_rounds_batch_core(same_ctx *ctx, const same_greedy_job *jobs, int n_jobs, int first, int count);

// ascending sort of n_pad (a power of two >= 2048) 64-bit keys in place (merge.hip)
int same_sort_u64_core(same_ctx *ctx, unsigned long long *dkey, int64_t n_pad);
// de-duplication of (aligned code, ref code) pairs on device arrays, enqueue only (merge.hip)
int same_merge_dedup_core(same_ctx *ctx, const uint8_t *dviol, unsigned viol_mask, const int32_t *dwin, const int32_t *da, const int32_t *dr,
                          int64_t n, int32_t *dout, unsigned long long *dtotal);

// the window path's prune and candidate-list costs on row lists of the sections (knn.hip, cost.hip), for the windows of a batch in one
// launch each (<= SAME_LAUNCH_WINDOWS jobs)
struct same_knn_index;
struct same_knn_window_job {
    const int32_t *rows_m = nullptr;            // aligned rows of the window: rows_m[0, *dn_m) into the moving section
    const unsigned long long *dn_m = nullptr;
    int64_t cap_m = 0;                          // the bound on *dn_m the launch is sized by
    const int32_t *rows_r = nullptr;            // reference rows of the window (brute form)
    const unsigned long long *dn_r = nullptr;
    double box[4] = {0, 0, 0, 0};
    int32_t *idx = nullptr, *cnt = nullptr;     // [cap_m][k], [cap_m]
};
int same_knn_window_batch_core(same_ctx *ctx, const same_knn_index *ix, const double *dmov_xy, const same_knn_window_job *jobs, int n_jobs, int k);
struct same_cost_window_job {
    const int32_t *rows = nullptr;
    const unsigned long long *dn = nullptr;
    int64_t cap = 0;
    const int32_t *idx = nullptr;
    void *out = nullptr;                        // [cap][k] in the cost type
};
int same_padded_cost_window_batch_core(same_ctx *ctx, int cost_f32, const void *dA, const void *dR, int T, const void *daxy_c, const void *drxy_c,
                                       const same_cost_window_job *jobs, int n_jobs, int k, double w);
