// window_merge.hip -- the window merge (src/helpers.py:692-815) where the windows' matches are: on the device.
//
// The reference concatenates every window's match table (src/same.py:565-590: the rows of the window's central region), sorts the
// lot, drops duplicate (aligned, ref) pairs and runs one maximum matching.  In a tiled run nearly every pair stands alone -- both of
// its cells are named by no other row -- and is in the merged table as it is; what needs the graph algorithm are the cells the
// windows' overlaps disagree about.  So:
//
//   same_window_collect       (enqueue only, after same_window_filter_finish) appends the matched cells of each window's central region
//                             to the pass's ACCUMULATOR, in window order, cells ascending: (aligned section row, matched reference
//                             section row, flag byte, window id, plan position, the cell's index among the window's kept cells).
//   same_merge_acc_resolve    on the accumulated rows of the pass (the accumulators of the worker threads' contexts laid end to end =
//                             plan order): id codes of both cells, the de-duplication (merge.hip's kernels on the device arrays), the
//                             degree of every cell, and per surviving row: alone AND not near another rank's windows -> final, its
//                             place in the result is its aligned code (a direct table); anything else -> the REST list, which goes to
//                             the host (same_amd/merge.py: components, Hopcroft-Karp, the exchange of the seam rows between ranks).
//   same_merge_acc_finish     the host's winners join the table; the table's rows in code order are the merged table's rows
//                             (src/helpers.py:799-808: aligned ids ascending) -- their keys come back, and the host gathers the
//                             columns of exactly these rows.
//
// All integer / comparison work, exact by construction; a pass of 10^6 rows is ~25 MB of traffic: latency-bound, no roofline to speak of.
#include "window_internal.h"

struct same_merge_acc {
    same_ctx *ctx = nullptr;
    // the rows (struct of arrays, capacity `cap`); `dcount[0]` rows so far (device), then the collect call's scratch words
    int64_t cap = 0;
    int32_t *a_row = nullptr, *r_row = nullptr, *wid = nullptr, *pos = nullptr, *cidx = nullptr;
    uint8_t *flags = nullptr;
    unsigned long long *dcount = nullptr;     // [0] rows, [1] pad, [2 .. 2 + SAME_LAUNCH_WINDOWS) per-window counts of the group in flight
    int64_t bound = 0;                        // host: the rows collected since begin cannot exceed this (sum of the windows' kept cells)
    // the seams of this rank's share of the plan (same_merge_acc_begin)
    win::DevBuf near_start, near_boxes, scan_words;   // scan_words: the collect call's look-back words, one run per window of a group
    int n_pos = 0, all_seam = 0;
    double reach = 0.0;
    // resolve / finish
    win::DevBuf work, out;
    int32_t *ac = nullptr, *rc = nullptr, *row_of = nullptr;
    unsigned long long *counters = nullptr;   // in `work`: [0] survivors of the de-duplication, [1] rest rows, [2] final rows
    int64_t n_codes_a = 0;
    int64_t n_rows = 0, n_kept = 0, n_rest = 0, n_final = 0;
    int resolved = 0;
    int final_on_host = 0;                    // the final records have been copied to `host` (same_merge_acc_fetch does it on demand)
    int loaded = 0;                           // the rows came from the host (same_merge_acc_load): their "section rows" ARE the codes
    int64_t loaded_codes_a = 0, loaded_codes_r = 0;
    std::vector<char> host;                   // what the last resolve / finish brought back (fetched by same_merge_acc_fetch)
};

// what comes back: one record per row (include/same_hip.h)
struct RestRec {
    int32_t row, ac, rc, wid, pos, cidx;
    uint32_t flags;               // bit 0 XY-order flag, bit 1 area-flip flag, bit 2 the row is at a seam
};
struct FinalRec {
    int32_t a_row, r_row, cidx, wid, pos;
    uint32_t flags;
};
static_assert(sizeof(RestRec) == SAME_MERGE_REST_BYTES && sizeof(FinalRec) == SAME_MERGE_FINAL_BYTES, "record layouts are part of the ABI");

namespace {

using namespace devmath;
using namespace win;
using scan::Pair;

// ---- collect ---------------------------------------------------------------------------------------------------------------
struct CollectArgs {
    const int32_t *match_row;     // [n] section row of the matched reference cell, -1 = none
    const uint8_t *pflag;         // [n] flag byte (bit 0 XY-order sweep, bit 1 vertex of an area-flipped triangle)
    const int32_t *rows_ua;       // [n] section rows of the kept aligned cells
    const double *axy;            // [n][2] their XY
    int64_t n;
    double tx0, tx1, ty0, ty1;    // the central region (src/same.py:565-582), half open
    int32_t wid, pos;
};

__device__ __forceinline__ bool central(const CollectArgs &w, int64_t c) {
    if (c >= w.n || w.match_row[c] < 0) return false;
    const double2_t p = ld2(w.axy, c);
    return p.x >= w.tx0 && p.x < w.tx1 && p.y >= w.ty0 && p.y < w.ty1;     // the comparisons of src/same.py:575-580
}

// blocks of 256 cells (blockIdx.x) of up to eight windows (blockIdx.y): how many of a window's kept cells are matched and central;
// every block also clears its word of the window's scan (collect_write_kernel's look-back runs on them next)
__global__ __launch_bounds__(scan::NT) void collect_count_kernel(Batch<CollectArgs> b, unsigned long long *__restrict__ wcount,
                                                                  unsigned long long *__restrict__ status, int64_t status_stride) {
    const CollectArgs &w = b.w[blockIdx.y];
    if ((int64_t)blockIdx.x * scan::NT >= w.n) return;
    __shared__ unsigned part[scan::NT / 64];
    const unsigned long long bal = __ballot(central(w, (int64_t)blockIdx.x * scan::NT + threadIdx.x));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = (unsigned)__builtin_popcountll(bal);
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned n = 0;
        for (int q = 0; q < scan::NT / 64; ++q) n += part[q];
        if (n) atomicAdd(&wcount[blockIdx.y], (unsigned long long)n);
        status[blockIdx.y * status_stride + blockIdx.x] = 0ull;
    }
}

struct AccRows {
    int32_t *a_row, *r_row, *wid, *pos, *cidx;
    uint8_t *flags;
};

// the same blocks: a window's rows after those of the group's earlier windows, cells ascending (look-back scan over the window's blocks)
__global__ __launch_bounds__(scan::NT) void collect_write_kernel(Batch<CollectArgs> b, const unsigned long long *__restrict__ dcount,
                                                                  unsigned long long *__restrict__ status, int64_t status_stride, AccRows acc,
                                                                  int64_t cap) {
    __shared__ scan::Shared sh;
    const CollectArgs &w = b.w[blockIdx.y];
    if ((int64_t)blockIdx.x * scan::NT >= w.n) return;
    auto val = [&](int64_t c) { return Pair{central(w, c) ? 1u : 0u, 0u}; };
    Pair through;
    const Pair off = scan::exclusive(status + blockIdx.y * status_stride, (int)blockIdx.x, val, sh, &through);   // (a test tag in bit 0 of `status` moves along)
    const int64_t c = (int64_t)blockIdx.x * scan::NT + threadIdx.x;
    if (!central(w, c)) return;
    unsigned long long at = dcount[0] + off.a;
    for (unsigned q = 0; q < blockIdx.y; ++q) at += dcount[2 + q];
    if ((int64_t)at >= cap) return;           // the host sized the arrays by the windows' kept cells: never past them
    acc.a_row[at] = w.rows_ua[c];
    acc.r_row[at] = w.match_row[c];
    acc.flags[at] = w.pflag[c];
    acc.wid[at] = w.wid;
    acc.pos[at] = w.pos;
    acc.cidx[at] = (int32_t)c;
}

// the group's rows are in: count them in, clear the per-window counts for the next group
__global__ void collect_bump_kernel(unsigned long long *dcount, int n_w) {
    unsigned long long add = 0;
    for (int q = 0; q < n_w; ++q) {
        add += dcount[2 + q];
        dcount[2 + q] = 0ull;
    }
    dcount[0] += add;
}

// ---- resolve ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void codes_kernel(const int32_t *__restrict__ a_row, const int32_t *__restrict__ r_row, int64_t n,
                                                     const int32_t *__restrict__ a_codes, const int32_t *__restrict__ r_codes,
                                                     int32_t *__restrict__ ac, int32_t *__restrict__ rc) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ac[i] = a_codes ? a_codes[a_row[i]] : a_row[i];
    rc[i] = r_codes ? r_codes[r_row[i]] : r_row[i];
}

__global__ __launch_bounds__(256) void degree_kernel(const int32_t *__restrict__ kept, const unsigned long long *__restrict__ dm, int64_t bound,
                                                      const int32_t *__restrict__ ac, const int32_t *__restrict__ rc,
                                                      unsigned *__restrict__ deg_a, unsigned *__restrict__ deg_r) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= bound || i >= (int64_t)*dm) return;
    const int32_t row = kept[i];
    atomicAdd(&deg_a[ac[row]], 1u);
    atomicAdd(&deg_r[rc[row]], 1u);
}

struct SeamArgs {
    const int32_t *near_start;    // [n_pos + 1], or null: no seams (one rank)
    const double *near_boxes;     // [..][4] central regions of the other ranks' windows near a window of this rank
    const double *mov_xy, *ref_xy;
    int n_pos, all_seam;
    double reach;
};

// may another rank's table name one of the row's two cells?  (same_amd/merge.py::seam_rows: the same two box tests)
__device__ __forceinline__ bool seam_of(const SeamArgs &s, int32_t pos, int32_t a_row, int32_t r_row) {
    if (s.all_seam) return true;
    if (!s.near_start) return false;
    if (pos < 0 || pos >= s.n_pos) return true;
    const int32_t b = s.near_start[pos], e = s.near_start[pos + 1];
    if (b == e) return false;
    const double2_t a = ld2(s.mov_xy, a_row), r = ld2(s.ref_xy, r_row);
    for (int32_t q = b; q < e; ++q) {
        const double x0 = s.near_boxes[4 * (int64_t)q], x1 = s.near_boxes[4 * (int64_t)q + 1], y0 = s.near_boxes[4 * (int64_t)q + 2],
                     y1 = s.near_boxes[4 * (int64_t)q + 3];
        if (a.x >= x0 && a.x < x1 && a.y >= y0 && a.y < y1) return true;
        if (r.x >= x0 - s.reach && r.x <= x1 + s.reach && r.y >= y0 - s.reach && r.y <= y1 + s.reach) return true;
    }
    return false;
}

// class of every surviving row: 0 = alone (both cells of degree one) and not at a seam: final, row_of[aligned code] = row;
// 1 = contested, 3 = at a seam (alone or not): the host's
__global__ __launch_bounds__(256) void classify_kernel(const int32_t *__restrict__ kept, const unsigned long long *__restrict__ dm, int64_t bound,
                                                        const int32_t *__restrict__ ac, const int32_t *__restrict__ rc,
                                                        const unsigned *__restrict__ deg_a, const unsigned *__restrict__ deg_r,
                                                        const int32_t *__restrict__ a_row, const int32_t *__restrict__ r_row,
                                                        const int32_t *__restrict__ pos, SeamArgs seams, uint8_t *__restrict__ cls,
                                                        int32_t *__restrict__ row_of) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= bound || i >= (int64_t)*dm) return;
    const int32_t row = kept[i];
    const bool lone = deg_a[ac[row]] == 1u && deg_r[rc[row]] == 1u;
    const bool seam = seam_of(seams, pos[row], a_row[row], r_row[row]);
    cls[i] = seam ? 3 : (lone ? 0 : 1);
    if (lone && !seam) row_of[ac[row]] = row;
}

__global__ __launch_bounds__(scan::NT) void rest_kernel(const int32_t *__restrict__ kept, const unsigned long long *__restrict__ dm,
                                                         const uint8_t *__restrict__ cls, const int32_t *__restrict__ ac,
                                                         const int32_t *__restrict__ rc, AccRows acc, unsigned long long *__restrict__ status,
                                                         RestRec *__restrict__ out, unsigned long long *__restrict__ out_total) {
    __shared__ scan::Shared sh;
    const int64_t m = (int64_t)*dm;
    auto val = [&](int64_t i) { return Pair{i < m && cls[i] != 0 ? 1u : 0u, 0u}; };
    Pair through;
    const Pair off = scan::exclusive(status, (int)blockIdx.x, val, sh, &through);
    const int64_t i = (int64_t)blockIdx.x * scan::NT + threadIdx.x;
    if (i < m && cls[i] != 0) {
        const int32_t row = kept[i];
        out[off.a] = RestRec{row, ac[row], rc[row], acc.wid[row], acc.pos[row], acc.cidx[row], (acc.flags[row] & 3u) | (cls[i] == 3 ? 4u : 0u)};
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *out_total = through.a;
}

// ---- finish ----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void winners_kernel(const int32_t *__restrict__ rows, int64_t n, const int32_t *__restrict__ ac,
                                                       int32_t *__restrict__ row_of) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) row_of[ac[rows[i]]] = rows[i];
}

__global__ __launch_bounds__(scan::NT) void final_kernel(const int32_t *__restrict__ row_of, int64_t n_codes, AccRows acc,
                                                          unsigned long long *__restrict__ status, FinalRec *__restrict__ out,
                                                          unsigned long long *__restrict__ out_total) {
    __shared__ scan::Shared sh;
    auto val = [&](int64_t c) { return Pair{c < n_codes && row_of[c] >= 0 ? 1u : 0u, 0u}; };
    Pair through;
    const Pair off = scan::exclusive(status, (int)blockIdx.x, val, sh, &through);
    const int64_t c = (int64_t)blockIdx.x * scan::NT + threadIdx.x;
    if (c < n_codes && row_of[c] >= 0) {
        const int32_t row = row_of[c];
        out[off.a] = FinalRec{acc.a_row[row], acc.r_row[row], acc.cidx[row], acc.wid[row], acc.pos[row], acc.flags[row] & 3u};
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *out_total = through.a;
}

// The merged table's columns, column by column into host memory the device can write.  One thread per row; a column's writes are
// consecutive.  8-byte columns first: T type columns of the moving section, its X and Y, the reference section's X and Y, then the
// caller's extra 8-byte columns of the moving rows and of the reference rows (ids, sizes: copied as bit patterns, whatever their
// type), the cell's index in its window, the window id and its plan position as int64; then two byte columns: the area-flip flag and the
// XY-order flag.
constexpr int MAX_EXTRA_COLUMNS = 4;
struct ExtraColumns {
    const unsigned long long *mov[MAX_EXTRA_COLUMNS], *ref[MAX_EXTRA_COLUMNS];
    int n_mov, n_ref;
};
__global__ __launch_bounds__(256) void table_columns_kernel(const FinalRec *__restrict__ rows, int64_t n, const double *__restrict__ mov_types, int T,
                                                            const double *__restrict__ mov_xy, const double *__restrict__ ref_xy, ExtraColumns ex,
                                                            unsigned long long *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const FinalRec rec = rows[i];
    const int64_t a = rec.a_row, r = rec.r_row;
    const unsigned long long *__restrict__ t = reinterpret_cast<const unsigned long long *>(mov_types) + a * T;
    int64_t col = 0;
    for (int q = 0; q < T; ++q) out[(col++) * n + i] = t[q];
    const unsigned long long *pa = reinterpret_cast<const unsigned long long *>(mov_xy) + 2 * a,
                             *pr = reinterpret_cast<const unsigned long long *>(ref_xy) + 2 * r;
    out[(col++) * n + i] = pa[0];
    out[(col++) * n + i] = pa[1];
    out[(col++) * n + i] = pr[0];
    out[(col++) * n + i] = pr[1];
    for (int q = 0; q < ex.n_mov; ++q) out[(col++) * n + i] = ex.mov[q][a];
    for (int q = 0; q < ex.n_ref; ++q) out[(col++) * n + i] = ex.ref[q][r];
    out[(col++) * n + i] = (unsigned long long)(long long)rec.cidx;
    out[(col++) * n + i] = (unsigned long long)(long long)rec.wid;
    out[(col++) * n + i] = (unsigned long long)(long long)rec.pos;
    uint8_t *bytes = reinterpret_cast<uint8_t *>(out + col * n);
    bytes[i] = (uint8_t)((rec.flags >> 1) & 1u);
    bytes[n + i] = (uint8_t)(rec.flags & 1u);
}

AccRows rows_of(const same_merge_acc *a) { return AccRows{a->a_row, a->r_row, a->wid, a->pos, a->cidx, a->flags}; }

void free_rows(same_merge_acc *a) {
    for (void *p : {(void *)a->a_row, (void *)a->r_row, (void *)a->wid, (void *)a->pos, (void *)a->cidx, (void *)a->flags})
        if (p) (void)hipFree(p);
    a->a_row = a->r_row = a->wid = a->pos = a->cidx = nullptr;
    a->flags = nullptr;
    a->cap = 0;
}

// room for `need` rows; rows already there (the first `have`) move along.  Waits for the stream when it has to grow.
int reserve_rows(same_merge_acc *a, int64_t need, int64_t have) {
    if (need <= a->cap) return SAME_OK;
    same_ctx *ctx = a->ctx;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const int64_t cap = std::max<int64_t>(need + need / 4, 4096);
    int32_t **fields[5] = {&a->a_row, &a->r_row, &a->wid, &a->pos, &a->cidx};
    int32_t *olds[5], *news[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    uint8_t *old_flags = a->flags, *new_flags = nullptr;
    for (int q = 0; q < 5; ++q) olds[q] = *fields[q];
    hipError_t e = hipSuccess;
    for (int q = 0; q < 5 && e == hipSuccess; ++q) {
        e = hipMalloc(reinterpret_cast<void **>(&news[q]), (size_t)cap * sizeof(int32_t));
        if (e == hipSuccess && have) e = hipMemcpyAsync(news[q], olds[q], (size_t)have * sizeof(int32_t), hipMemcpyDeviceToDevice, ctx->stream);
    }
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&new_flags), (size_t)cap);
    if (e == hipSuccess && have) e = hipMemcpyAsync(new_flags, old_flags, (size_t)have, hipMemcpyDeviceToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {                  // the new arrays (partly made) go, the old ones stay
        for (int q = 0; q < 5; ++q)
            if (news[q]) (void)hipFree(news[q]);
        if (new_flags) (void)hipFree(new_flags);
        return same_fail(ctx, e == hipErrorOutOfMemory ? SAME_ENOMEM : SAME_EIO, "merge accumulator rows", e);
    }
    for (int q = 0; q < 5; ++q) {
        if (olds[q]) (void)hipFree(olds[q]);
        *fields[q] = news[q];
    }
    if (old_flags) (void)hipFree(old_flags);
    a->flags = new_flags;
    a->cap = cap;
    return SAME_OK;
}

// The accumulators of a pass laid end to end in accs[0] (the other contexts' collect calls only enqueued: their streams are waited for;
// the worker threads walked consecutive runs of the plan, so the result is in plan order).  -> the number of rows.
int lay_end_to_end(same_merge_acc *const *accs, int n_accs, int64_t *out_n) {
    same_merge_acc *a = accs[0];
    same_ctx *ctx = a->ctx;
    std::vector<int64_t> have((size_t)n_accs, 0);
    int64_t n = 0;
    for (int q = 0; q < n_accs; ++q) {
        unsigned long long c = 0;
        HIP_TRY(ctx, hipStreamSynchronize(accs[q]->ctx->stream));
        HIP_TRY(ctx, hipMemcpy(&c, accs[q]->dcount, sizeof c, hipMemcpyDeviceToHost));
        REQUIRE(ctx, (int64_t)c <= accs[q]->bound && (int64_t)c <= accs[q]->cap);
        have[(size_t)q] = (int64_t)c;
        n += (int64_t)c;
    }
    REQUIRE(ctx, n < ((int64_t)1 << 30));
    if (n_accs > 1) {
        SAME_TRY(reserve_rows(a, n, have[0]));
        int64_t at = have[0];
        for (int q = 1; q < n_accs; ++q) {
            const same_merge_acc *o = accs[q];
            const size_t m = (size_t)have[(size_t)q];
            if (!m) continue;
            int32_t *const dst[5] = {a->a_row, a->r_row, a->wid, a->pos, a->cidx};
            const int32_t *const src[5] = {o->a_row, o->r_row, o->wid, o->pos, o->cidx};
            for (int f = 0; f < 5; ++f) SAME_COPY(ctx, dst[f] + at, src[f], m * sizeof(int32_t), hipMemcpyDeviceToDevice);
            SAME_COPY(ctx, a->flags + at, o->flags, m, hipMemcpyDeviceToDevice);
            at += (int64_t)m;
        }
    }
    *out_n = n;
    return SAME_OK;
}

// every accumulated row as a final record, in the order it came (same_merge_acc_plain)
__global__ __launch_bounds__(256) void plain_kernel(AccRows acc, int64_t n, FinalRec *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = FinalRec{acc.a_row[i], acc.r_row[i], acc.cidx[i], acc.wid[i], acc.pos[i], acc.flags[i] & 3u};
}

}  // namespace

extern "C" {

int same_section_set_codes(same_section *s, const int32_t *codes, int64_t n_codes) {
    if (!s) return SAME_EINVAL;
    same_ctx *ctx = s->ctx;
    REQUIRE(ctx, !codes || (n_codes >= 0 && n_codes <= std::max<int64_t>(s->n, 0)));      // without codes n_codes means nothing
    SAME_TRY(same_use(ctx));
    if (s->id_codes) {
        HIP_TRY(ctx, hipDeviceSynchronize());
        (void)hipFree(s->id_codes);
        s->id_codes = nullptr;
    }
    s->n_codes = codes ? n_codes : s->n;
    if (!codes || s->n == 0) return SAME_OK;
    SAME_TRY(check_index_range(ctx, codes, s->n, 0, std::max<int64_t>(n_codes, 1), "id codes"));
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&s->id_codes), (size_t)s->n * sizeof(int32_t)));
    HIP_TRY(ctx, hipMemcpyAsync(s->id_codes, codes, (size_t)s->n * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SAME_OK;
}

int same_merge_acc_create(same_ctx *ctx, same_merge_acc **out) {
    REQUIRE(ctx, ctx && out);
    *out = nullptr;
    SAME_TRY(same_use(ctx));
    same_merge_acc *a = new (std::nothrow) same_merge_acc();
    if (!a) return SAME_ENOMEM;
    a->ctx = ctx;
    *out = a;                                     // freed by the caller's destroy on any failure below
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&a->dcount), (2 + SAME_LAUNCH_WINDOWS) * sizeof(unsigned long long)));
    HIP_TRY(ctx, hipMemsetAsync(a->dcount, 0, (2 + SAME_LAUNCH_WINDOWS) * sizeof(unsigned long long), ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SAME_OK;
}

void same_merge_acc_destroy(same_merge_acc *a) {
    if (!a) return;
    (void)hipSetDevice(a->ctx->device);
    (void)hipStreamSynchronize(a->ctx->stream);
    free_rows(a);
    if (a->dcount) (void)hipFree(a->dcount);
    for (DevBuf *b : {&a->near_start, &a->near_boxes, &a->scan_words, &a->work, &a->out}) release(*b);
    delete a;
}

int same_merge_acc_begin(same_merge_acc *a, int64_t expected_rows, int n_pos, const int32_t *near_start, const double *near_boxes, double reach,
                         int all_seam) {
    if (!a) return SAME_EINVAL;
    same_ctx *ctx = a->ctx;
    REQUIRE(ctx, expected_rows >= 0 && expected_rows < ((int64_t)1 << 30) && n_pos >= 0 && reach >= 0.0 && reach - reach == 0.0);
    REQUIRE(ctx, !near_start || n_pos == 0 || near_boxes || near_start[n_pos] == 0);
    SAME_TRY(same_use(ctx));
    a->resolved = 0;
    a->loaded = 0;
    a->bound = 0;
    a->n_rows = a->n_kept = a->n_rest = a->n_final = 0;
    a->n_pos = near_start ? n_pos : 0;
    a->all_seam = all_seam ? 1 : 0;
    a->reach = reach;
    SAME_FILL(ctx, a->dcount, 0, (2 + SAME_LAUNCH_WINDOWS) * sizeof(unsigned long long));
    SAME_TRY(reserve_rows(a, expected_rows, 0));
    if (near_start && n_pos > 0) {
        REQUIRE(ctx, near_start[0] == 0);
        for (int q = 0; q < n_pos; ++q) REQUIRE(ctx, near_start[q + 1] >= near_start[q]);
        const int64_t n_boxes = near_start[n_pos];
        SAME_TRY(ensure(ctx, a->near_start, (size_t)(n_pos + 1) * sizeof(int32_t)));
        SAME_TRY(ensure(ctx, a->near_boxes, (size_t)std::max<int64_t>(n_boxes, 1) * 4 * sizeof(double)));
        SAME_COPY(ctx, a->near_start.p, near_start, (size_t)(n_pos + 1) * sizeof(int32_t), hipMemcpyHostToDevice);
        if (n_boxes) SAME_COPY(ctx, a->near_boxes.p, near_boxes, (size_t)n_boxes * 4 * sizeof(double), hipMemcpyHostToDevice);
        SAME_WAIT(ctx);                            // the host arrays are the caller's again
    }
    return SAME_OK;
}

int same_merge_acc_load(same_merge_acc *a, const int32_t *a_code, const int32_t *r_code, const uint8_t *flags, const int32_t *window_ids,
                        const int32_t *pos, const int32_t *cidx, int64_t n, int64_t n_codes_a, int64_t n_codes_r) {
    if (!a) return SAME_EINVAL;
    same_ctx *ctx = a->ctx;
    REQUIRE(ctx, n >= 0 && n < ((int64_t)1 << 30) && n_codes_a >= 0 && n_codes_r >= 0 && n_codes_a < ((int64_t)1 << 31) && n_codes_r < ((int64_t)1 << 31));
    REQUIRE(ctx, n == 0 || (a_code && r_code && flags && window_ids && pos && cidx));
    SAME_TRY(check_index_range(ctx, a_code, n, 0, std::max<int64_t>(n_codes_a, 1), "aligned codes"));
    SAME_TRY(check_index_range(ctx, r_code, n, 0, std::max<int64_t>(n_codes_r, 1), "reference codes"));
    SAME_TRY(check_index_range(ctx, window_ids, n, 0, (int64_t)1 << 31, "window ids"));
    SAME_TRY(same_use(ctx));
    a->resolved = 0;
    a->n_rows = a->n_kept = a->n_rest = a->n_final = 0;
    a->n_pos = 0;
    a->all_seam = 0;
    a->reach = 0.0;
    SAME_TRY(reserve_rows(a, n, 0));
    a->bound = n;
    a->loaded = 1;
    a->loaded_codes_a = n_codes_a;
    a->loaded_codes_r = n_codes_r;
    const unsigned long long cnt[2] = {(unsigned long long)n, 0ull};
    SAME_COPY(ctx, a->dcount, cnt, sizeof cnt, hipMemcpyHostToDevice);
    if (n) {
        int32_t *const dst[5] = {a->a_row, a->r_row, a->wid, a->pos, a->cidx};
        const int32_t *const src[5] = {a_code, r_code, window_ids, pos, cidx};
        for (int f = 0; f < 5; ++f) SAME_COPY(ctx, dst[f], src[f], (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice);
        SAME_COPY(ctx, a->flags, flags, (size_t)n, hipMemcpyHostToDevice);
    }
    SAME_WAIT(ctx);                               // the host arrays are the caller's again
    return SAME_OK;
}

int same_window_collect(same_window *const *windows, int n_windows, same_merge_acc *a, const double *trims, const int32_t *window_ids,
                        const int32_t *plan_pos) {
    same_ctx *ctx = nullptr;
    SAME_TRY(check_batch(windows, n_windows, &ctx));
    REQUIRE(ctx, a && a->ctx == ctx && trims && window_ids && plan_pos && !a->resolved);
    int64_t add = 0;
    for (int i = 0; i < n_windows; ++i) {
        REQUIRE(ctx, windows[i]->finished && windows[i]->staged == 2 && window_ids[i] >= 0);
        add += windows[i]->n_ua;
    }
    REQUIRE(ctx, a->bound + add < ((int64_t)1 << 30));
    SAME_TRY(same_use(ctx));
    if (a->bound + add > a->cap) {                 // rare (begin sized the arrays for the pass): the rows so far move to larger arrays
        unsigned long long have = 0;
        HIP_TRY(ctx, hipMemcpyAsync(&have, a->dcount, sizeof have, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        SAME_TRY(reserve_rows(a, a->bound + add, (int64_t)have));
    }
    a->bound += add;
    int64_t most = 0;
    for (int i = 0; i < n_windows; ++i) most = std::max(most, windows[i]->n_ua);
    const int64_t stride = (int64_t)(scan::status_bytes(most) / 8);          // look-back words of one window
    SAME_TRY(ensure(ctx, a->scan_words, (size_t)stride * 8 * SAME_LAUNCH_WINDOWS));
    unsigned long long *words = static_cast<unsigned long long *>(a->scan_words.p);
    for (int g = 0; g < n_windows; g += SAME_LAUNCH_WINDOWS) {
        Batch<CollectArgs> b{};
        int n_g = 0;
        int64_t most_g = 0;
        for (int i = g; i < n_windows && i < g + SAME_LAUNCH_WINDOWS; ++i) {
            const same_window *w = windows[i];
            if (w->n_ua == 0) continue;
            const double *t = trims + 4 * i;
            b.w[n_g++] = CollectArgs{w->match_row, w->pflag, w->rows_ua, w->axy_c, w->n_ua, t[0], t[1], t[2], t[3], window_ids[i], plan_pos[i]};
            most_g = std::max(most_g, w->n_ua);
        }
        if (!n_g) continue;
        const dim3 grid(scan::blocks_for(most_g), (unsigned)n_g);
        SAME_LAUNCH(ctx, collect_count_kernel, grid, dim3(scan::NT), 0, b, a->dcount + 2, words, stride);
        SAME_LAUNCH(ctx, collect_write_kernel, grid, dim3(scan::NT), 0, b, a->dcount, scan::arg(words), stride, rows_of(a), a->cap);
        SAME_LAUNCH(ctx, collect_bump_kernel, dim3(1), dim3(1), 0, a->dcount, n_g);
    }
    HIP_TRY(ctx, hipGetLastError());
    return SAME_OK;
}

int same_merge_acc_resolve(same_merge_acc *const *accs, int n_accs, const same_section *mov, const same_section *ref, int64_t *out_counts) {
    if (!accs || n_accs < 1 || !accs[0]) return SAME_EINVAL;
    same_merge_acc *a = accs[0];
    same_ctx *ctx = a->ctx;
    REQUIRE(ctx, n_accs <= 64 && out_counts);
    const bool loaded = a->loaded != 0;           // rows that came from the host carry their codes: no sections, one accumulator
    REQUIRE(ctx, loaded ? n_accs == 1 : (mov && ref && mov->ctx->device == ctx->device && ref->ctx->device == ctx->device));
    for (int q = 0; q < n_accs; ++q) {
        REQUIRE(ctx, accs[q] && accs[q]->ctx->device == ctx->device && !accs[q]->resolved);
        for (int p = 0; p < q; ++p) REQUIRE(ctx, accs[p] != accs[q]);
    }
    for (int q = 0; q < 4; ++q) out_counts[q] = 0;
    SAME_TRY(same_use(ctx));
    int64_t n = 0;
    SAME_TRY(lay_end_to_end(accs, n_accs, &n));
    for (int q = 0; q < n_accs; ++q) accs[q]->resolved = 1;
    a->n_rows = n;
    a->n_codes_a = loaded ? a->loaded_codes_a : (mov->id_codes ? mov->n_codes : mov->n);
    const int64_t n_codes_a = std::max<int64_t>(a->n_codes_a, 1),
                  n_codes_r = std::max<int64_t>(loaded ? a->loaded_codes_r : (ref->id_codes ? ref->n_codes : ref->n), 1);
    const int32_t *codes_a = loaded ? nullptr : mov->id_codes, *codes_r = loaded ? nullptr : ref->id_codes;
    // layout of the work buffer: [scan words of the rest list | counters] zeroed; codes, survivors, classes, degree tables, the result table
    const int64_t nn = std::max<int64_t>(n, 1);
    Carver cv;
    const size_t o_status = cv.take(scan::status_bytes(nn)), o_counters = cv.take(64), o_deg_a = cv.take((size_t)n_codes_a * 4),
                 o_deg_r = cv.take((size_t)n_codes_r * 4);
    const size_t zero_bytes = cv.off;
    const size_t o_row_of = cv.take((size_t)n_codes_a * 4), o_ac = cv.take((size_t)nn * 4), o_rc = cv.take((size_t)nn * 4),
                 o_kept = cv.take((size_t)nn * 4), o_cls = cv.take((size_t)nn);
    SAME_TRY(ensure(ctx, a->work, cv.off));
    char *base = static_cast<char *>(a->work.p);
    unsigned long long *status = reinterpret_cast<unsigned long long *>(base + o_status), *counters = reinterpret_cast<unsigned long long *>(base + o_counters);
    unsigned *deg_a = reinterpret_cast<unsigned *>(base + o_deg_a), *deg_r = reinterpret_cast<unsigned *>(base + o_deg_r);
    a->row_of = reinterpret_cast<int32_t *>(base + o_row_of);
    a->ac = reinterpret_cast<int32_t *>(base + o_ac);
    a->rc = reinterpret_cast<int32_t *>(base + o_rc);
    int32_t *kept = reinterpret_cast<int32_t *>(base + o_kept);
    uint8_t *cls = reinterpret_cast<uint8_t *>(base + o_cls);
    SAME_FILL(ctx, base, 0, zero_bytes);
    SAME_FILL(ctx, a->row_of, 0xFF, (size_t)n_codes_a * 4);
    a->counters = counters;
    SAME_TRY(ensure(ctx, a->out, (size_t)nn * sizeof(RestRec)));          // the rest list: at most n records
    RestRec *rest = static_cast<RestRec *>(a->out.p);
    if (n) {
        SAME_LAUNCH(ctx, codes_kernel, dim3(grid_for(n)), dim3(256), 0, a->a_row, a->r_row, n, codes_a, codes_r, a->ac, a->rc);
        SAME_TRY(same_merge_dedup_core(ctx, a->flags, 1u, a->wid, a->ac, a->rc, n, kept, counters));          // counters[0] = survivors
        SAME_LAUNCH(ctx, degree_kernel, dim3(grid_for(n)), dim3(256), 0, kept, counters, n, a->ac, a->rc, deg_a, deg_r);
        const SeamArgs seams{a->n_pos ? static_cast<const int32_t *>(a->near_start.p) : nullptr, static_cast<const double *>(a->near_boxes.p),
                             loaded ? nullptr : mov->xy, loaded ? nullptr : ref->xy, a->n_pos, a->all_seam, a->reach};
        SAME_LAUNCH(ctx, classify_kernel, dim3(grid_for(n)), dim3(256), 0, kept, counters, n, a->ac, a->rc, deg_a, deg_r, a->a_row, a->r_row, a->pos,
                    seams, cls, a->row_of);
        SAME_LAUNCH(ctx, rest_kernel, dim3(scan::blocks_for(n)), dim3(scan::NT), 0, kept, counters, cls, a->ac, a->rc, rows_of(a), scan::arg(status), rest,
                    counters + 1);
        HIP_TRY(ctx, hipGetLastError());
        unsigned long long hc[2] = {0, 0};
        SAME_COPY(ctx, hc, counters, sizeof hc, hipMemcpyDeviceToHost);
        SAME_WAIT(ctx);
        REQUIRE(ctx, (int64_t)hc[0] <= n && (int64_t)hc[1] <= (int64_t)hc[0]);
        a->n_kept = (int64_t)hc[0];
        a->n_rest = (int64_t)hc[1];
    }
    a->host.resize((size_t)a->n_rest * sizeof(RestRec) + 64);
    if (a->n_rest) {
        SAME_COPY(ctx, a->host.data(), rest, (size_t)a->n_rest * sizeof(RestRec), hipMemcpyDeviceToHost);
        SAME_WAIT(ctx);
    }
    out_counts[0] = a->n_rows;
    out_counts[1] = a->n_kept;
    out_counts[2] = a->n_rest;
    out_counts[3] = a->n_kept - a->n_rest;
    return SAME_OK;
}

int same_merge_acc_plain(same_merge_acc *const *accs, int n_accs, int64_t *out_n_rows) {
    if (!accs || n_accs < 1 || !accs[0]) return SAME_EINVAL;
    same_merge_acc *a = accs[0];
    same_ctx *ctx = a->ctx;
    REQUIRE(ctx, n_accs <= 64 && out_n_rows && !a->loaded);
    for (int q = 0; q < n_accs; ++q) {
        REQUIRE(ctx, accs[q] && accs[q]->ctx->device == ctx->device && !accs[q]->resolved);
        for (int p = 0; p < q; ++p) REQUIRE(ctx, accs[p] != accs[q]);
    }
    *out_n_rows = 0;
    SAME_TRY(same_use(ctx));
    int64_t n = 0;
    SAME_TRY(lay_end_to_end(accs, n_accs, &n));
    for (int q = 0; q < n_accs; ++q) accs[q]->resolved = 2;
    a->n_rows = a->n_kept = a->n_final = n;
    a->n_rest = 0;
    a->final_on_host = 0;
    SAME_TRY(ensure(ctx, a->out, (size_t)std::max<int64_t>(n, 1) * sizeof(FinalRec)));
    if (n) SAME_LAUNCH(ctx, plain_kernel, dim3(grid_for(n)), dim3(256), 0, rows_of(a), n, static_cast<FinalRec *>(a->out.p));
    HIP_TRY(ctx, hipGetLastError());
    *out_n_rows = n;
    return SAME_OK;
}

int same_merge_acc_finish(same_merge_acc *a, const int32_t *winner_rows, int64_t n_winners, int64_t *out_n_final) {
    if (!a) return SAME_EINVAL;
    same_ctx *ctx = a->ctx;
    REQUIRE(ctx, a->resolved && a->row_of && out_n_final && n_winners >= 0 && n_winners <= a->n_rest && (n_winners == 0 || winner_rows));
    *out_n_final = 0;
    SAME_TRY(same_use(ctx));
    SAME_TRY(check_index_range(ctx, winner_rows, n_winners, 0, std::max<int64_t>(a->n_rows, 1), "winner rows"));
    const int64_t n_codes = std::max<int64_t>(a->n_codes_a, 1);
    unsigned long long *counters = a->counters, *status2 = nullptr;       // the second scan's words: a scratch slot of the context
    SAME_TRY(slot_as(ctx, SL_MASK, scan::status_bytes(n_codes) / 8, &status2));
    SAME_FILL(ctx, status2, 0, scan::status_bytes(n_codes));
    if (n_winners) {
        int32_t *dw = nullptr;
        SAME_TRY(up_as(ctx, SL_MATCH, winner_rows, (size_t)n_winners, &dw));
        SAME_LAUNCH(ctx, winners_kernel, dim3(grid_for(n_winners)), dim3(256), 0, dw, n_winners, a->ac, a->row_of);
    }
    SAME_TRY(ensure(ctx, a->out, (size_t)std::max<int64_t>(a->n_kept, 1) * sizeof(FinalRec)));      // the rest list has been fetched by now
    FinalRec *fin = static_cast<FinalRec *>(a->out.p);
    SAME_LAUNCH(ctx, final_kernel, dim3(scan::blocks_for(n_codes)), dim3(scan::NT), 0, a->row_of, a->n_codes_a, rows_of(a), scan::arg(status2), fin,
                counters + 2);
    HIP_TRY(ctx, hipGetLastError());
    unsigned long long total = 0;
    SAME_COPY(ctx, &total, counters + 2, sizeof total, hipMemcpyDeviceToHost);
    SAME_WAIT(ctx);
    REQUIRE(ctx, (int64_t)total <= a->n_kept);
    a->n_final = (int64_t)total;          // the records stay on the device until somebody asks for them (same_merge_acc_fetch, ..._columns)
    a->final_on_host = 0;
    a->resolved = 2;
    *out_n_final = a->n_final;
    return SAME_OK;
}

int same_merge_acc_columns(same_merge_acc *a, const same_section *mov, const same_section *ref, const void *const *extra_mov, int n_extra_mov,
                           const void *const *extra_ref, int n_extra_ref, void *out_host, int64_t n_final) {
    if (!a) return SAME_EINVAL;
    same_ctx *ctx = a->ctx;
    REQUIRE(ctx, a->resolved == 2 && !a->loaded && mov && ref && n_final == a->n_final && (n_final == 0 || out_host));
    REQUIRE(ctx, mov->ctx->device == ctx->device && ref->ctx->device == ctx->device && (mov->T == 0 || mov->types64));
    REQUIRE(ctx, n_extra_mov >= 0 && n_extra_mov <= MAX_EXTRA_COLUMNS && n_extra_ref >= 0 && n_extra_ref <= MAX_EXTRA_COLUMNS);
    REQUIRE(ctx, (n_extra_mov == 0 || extra_mov) && (n_extra_ref == 0 || extra_ref));
    if (n_final == 0) return SAME_OK;
    SAME_TRY(same_use(ctx));
    void *dev = nullptr;                          // the block as the device addresses it (it must come from same_host_alloc)
    if (hipHostGetDevicePointer(&dev, out_host, 0) != hipSuccess || !dev) {
        (void)hipGetLastError();
        ctx->err = "same_merge_acc_columns: the output block is not host memory the device can write (same_host_alloc)";
        return SAME_EINVAL;
    }
    ExtraColumns ex{};
    ex.n_mov = n_extra_mov;
    ex.n_ref = n_extra_ref;
    for (int q = 0; q < n_extra_mov; ++q) {
        REQUIRE(ctx, extra_mov[q]);
        ex.mov[q] = static_cast<const unsigned long long *>(extra_mov[q]);
    }
    for (int q = 0; q < n_extra_ref; ++q) {
        REQUIRE(ctx, extra_ref[q]);
        ex.ref[q] = static_cast<const unsigned long long *>(extra_ref[q]);
    }
    SAME_LAUNCH(ctx, table_columns_kernel, dim3(grid_for(n_final)), dim3(256), 0, static_cast<const FinalRec *>(a->out.p), n_final, mov->types64, mov->T,
                mov->xy, ref->xy, ex, static_cast<unsigned long long *>(dev));
    HIP_TRY(ctx, hipGetLastError());
    return SAME_OK;
}

int same_host_alloc(same_ctx *ctx, size_t bytes, void **out_ptr) {
    REQUIRE(ctx, ctx && out_ptr && bytes > 0);
    *out_ptr = nullptr;
    SAME_TRY(same_use(ctx));
    HIP_TRY(ctx, hipHostMalloc(out_ptr, bytes, hipHostMallocDefault));
    return SAME_OK;
}

int same_host_free(same_ctx *ctx, void *ptr) {
    REQUIRE(ctx, ctx != nullptr);
    if (!ptr) return SAME_OK;
    SAME_TRY(same_use(ctx));
    HIP_TRY(ctx, hipHostFree(ptr));
    return SAME_OK;
}

int same_merge_acc_fetch(same_merge_acc *a, int what, void *out, int64_t bytes) {
    if (!a) return SAME_EINVAL;
    same_ctx *ctx = a->ctx;
    REQUIRE(ctx, bytes >= 0 && (bytes == 0 || out));
    int64_t want = 0;
    if (what == SAME_MERGE_REST) {
        REQUIRE(ctx, a->resolved == 1);
        want = a->n_rest * (int64_t)sizeof(RestRec);
    } else if (what == SAME_MERGE_FINAL) {
        REQUIRE(ctx, a->resolved == 2);
        want = a->n_final * (int64_t)sizeof(FinalRec);
        if (bytes == want && want && !a->final_on_host) {
            SAME_TRY(same_use(ctx));
            a->host.resize((size_t)want + 64);
            SAME_COPY(ctx, a->host.data(), a->out.p, (size_t)want, hipMemcpyDeviceToHost);
            SAME_WAIT(ctx);
            a->final_on_host = 1;
        }
    } else {
        REQUIRE(ctx, !"unknown same_merge_acc_fetch selector");
    }
    REQUIRE(ctx, bytes == want);
    if (want) memcpy(out, a->host.data(), (size_t)want);
    return SAME_OK;
}

}  // extern "C"
