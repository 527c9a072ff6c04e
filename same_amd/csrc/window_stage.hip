// window_stage.hip -- same_window_stage: per window the rows of both sections inside the box, ascending (= np.flatnonzero of
// src/same.py:293-295), built from the few cells the box covers: a row's place in the list is the number of smaller rows in those
// cells (one binary search per cell), so the call reads O(window) rows whatever the section's size; radius / k prune against the
// reference SECTION's grid index (src/utils.py:709-728; candidates outside the box are not candidates), costs of the candidate lists
// in the cost type (src/same.py:1180-1189), compaction of the aligned side and of the pair list (src/utils.py:734-742).  Back to the
// host: four counts, the kept aligned rows and their XY -- the input of the host's Delaunay call (src/same.py:1023).
//
// Reference cells keep their SECTION rows through prune, costs and sweeps; the window's own numbering (position in the ascending
// list of reference rows in the box -- what the reference's frames would index before src/utils.py:740-742 drops the unreferenced)
// is only looked up for the pair list handed out and for the greedy rule's per-column state.  A window's numbers are those of
// the column pipeline bit for bit (tests/test_gpu_run_same.py::test_device_windows_*).
#include "window_internal.h"

namespace {

using namespace devmath;
using namespace win;
using scan::Pair;

__global__ __launch_bounds__(256) void zero_kernel(Batch<ZeroArgs> b) {
    const ZeroArgs &w = b.w[blockIdx.y];
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const size_t n16 = w.bytes[r] >> 4;
        u4 *dst = static_cast<u4 *>(w.p[r]);
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = u4{0u, 0u, 0u, 0u};
        if (blockIdx.x == 0 && threadIdx.x < (w.bytes[r] & 15)) static_cast<char *>(w.p[r])[(n16 << 4) + threadIdx.x] = 0;
    }
}

__global__ __launch_bounds__(256) void copy_back_kernel(Batch<CopyArgs> b) {
    const CopyArgs &w = b.w[blockIdx.y];
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const size_t n16 = w.bytes[r] >> 4;
        const u4 *src = static_cast<const u4 *>(w.src[r]);
        u4 *dst = static_cast<u4 *>(w.dst[r]);
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
        if (blockIdx.x == 0 && threadIdx.x < (w.bytes[r] & 15))
            static_cast<char *>(w.dst[r])[(n16 << 4) + threadIdx.x] = static_cast<const char *>(w.src[r])[(n16 << 4) + threadIdx.x];
    }
}

// ---- rows of a section inside a box, from the cells the box covers ---------------------------------------------------------
struct RunDesc {               // one section's share of a window
    const int32_t *order;      // the section's rows by cell
    const unsigned *starts;    // cell offsets into `order`
    const double *xy;          // section XY (box test of the candidates)
    int nx, cx0, ncx, cy0, ncy;   // covered cells: [cx0, cx0 + ncx) x [cy0, cy0 + ncy), ncx * ncy <= MAX_RUN_CELLS
    int n_cand;                // rows in those cells (the host knows the cell offsets)
    int aligned;               // the box is a union of cells: every candidate is inside
    uint32_t *merged;          // out: the candidates ascending by row (| OUTSIDE where the box test fails)
    unsigned long long *count; // out, aligned only: n_cand
};

// One thread per candidate: its place in the ascending list = the number of candidates with a smaller row = the sum over the
// covered cells of a lower bound in that cell's (ascending) run.  No sort, no scan; rows in different cells are distinct.
struct RowsArgs {
    RunDesc dm, dr;
    unsigned blocks_m, blocks;     // blocks [0, blocks_m) walk the moving section's candidates, [blocks_m, blocks) the reference's
    double bx0, bx1, by0, by1;
};
__global__ __launch_bounds__(256) void window_rows_kernel(Batch<RowsArgs> b) {
    __shared__ unsigned lo[MAX_RUN_CELLS], hi[MAX_RUN_CELLS], pref[MAX_RUN_CELLS + 1];
    const RowsArgs &wa = b.w[blockIdx.y];
    if (blockIdx.x >= wa.blocks) return;
    const unsigned blocks_m = wa.blocks_m;
    const double bx0 = wa.bx0, bx1 = wa.bx1, by0 = wa.by0, by1 = wa.by1;
    const bool second = blockIdx.x >= blocks_m;
    const RunDesc &d = second ? wa.dr : wa.dm;
    const int nc = d.ncx * d.ncy;
    if (threadIdx.x < 64) {     // one wave: the cells' runs and the prefix of their lengths
        const int c = threadIdx.x;
        unsigned a = 0, b = 0;
        if (c < nc) {
            const int64_t cell = (int64_t)(d.cy0 + c / d.ncx) * d.nx + (d.cx0 + c % d.ncx);
            a = d.starts[cell];
            b = d.starts[cell + 1];
        }
        unsigned incl = b - a;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned v = __shfl_up(incl, off, 64);
            if (c >= off) incl += v;
        }
        lo[c] = a;
        hi[c] = b;
        pref[c + 1] = incl;
        if (c == 0) pref[0] = 0;
    }
    __syncthreads();
    const int64_t q = (int64_t)(blockIdx.x - (second ? blocks_m : 0)) * blockDim.x + threadIdx.x;
    if (q == 0 && d.aligned) *d.count = (unsigned long long)d.n_cand;
    if (q >= d.n_cand) return;
    int c0 = 0, c1 = nc;        // the cell of candidate q: pref[c] <= q < pref[c + 1]
    while (c1 - c0 > 1) {
        const int mid = (c0 + c1) >> 1;
        if (pref[mid] <= (unsigned)q) c0 = mid; else c1 = mid;
    }
    const unsigned at = lo[c0] + ((unsigned)q - pref[c0]);
    const int32_t row = d.order[at];
    unsigned rank = at - lo[c0];
    // lower bounds in the other cells' runs, eight runs in lock step: the eight loads of a step are independent, so a step costs one
    // memory latency instead of eight (the searches are latency-bound: ~10 dependent loads each)
    constexpr int G = 8;
    for (int cb = 0; cb < nc; cb += G) {
        unsigned a[G], n[G];
        bool any = false;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int c = cb + g;
            const bool live = c < nc && c != c0;
            a[g] = live ? lo[c] : 0u;
            n[g] = live ? hi[c] - lo[c] : 0u;
            any = any || n[g] != 0u;
        }
        while (any) {
            int32_t v[G];
#pragma unroll
            for (int g = 0; g < G; ++g) v[g] = n[g] ? d.order[a[g] + (n[g] >> 1)] : 0;
            any = false;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (n[g]) {
                    const unsigned half = n[g] >> 1;
                    if (v[g] < row) { a[g] += half + 1; n[g] -= half + 1; } else n[g] = half;
                    any = any || n[g] != 0u;
                }
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int c = cb + g;
            if (c < nc && c != c0) rank += a[g] - lo[c];
        }
    }
    uint32_t out = (uint32_t)row;
    if (!d.aligned) {
        const double2_t p = ld2(d.xy, row);
        if (!(p.x >= bx0 && p.x < bx1 && p.y >= by0 && p.y < by1)) out |= OUTSIDE;   // src/same.py:293-295
    }
    d.merged[rank] = out;
}

// the candidates that passed the box test, in order (only for boxes that cut through cells); blocks [0, blocks_m) = moving
struct RowsCompact {
    const uint32_t *merged;
    int n_cand;
    int32_t *rows;
    unsigned long long *status, *count;
};
struct CompactArgs {
    RowsCompact cm, cr;
    unsigned blocks_m, blocks;
};
__global__ __launch_bounds__(scan::NT) void rows_compact_kernel(Batch<CompactArgs> bt) {
    __shared__ scan::Shared sh;
    const CompactArgs &wa = bt.w[blockIdx.y];
    if (blockIdx.x >= wa.blocks) return;
    const unsigned blocks_m = wa.blocks_m;
    const bool second = blockIdx.x >= blocks_m;
    const RowsCompact &c = second ? wa.cr : wa.cm;
    const int b = (int)(blockIdx.x - (second ? blocks_m : 0));
    const int nb = (int)(second ? wa.blocks - blocks_m : blocks_m);
    auto val = [&](int64_t i) { return Pair{i < c.n_cand && !(c.merged[i] & OUTSIDE) ? 1u : 0u, 0u}; };
    Pair through;
    const Pair off = scan::exclusive(c.status, b, val, sh, &through);
    const int64_t i = (int64_t)b * scan::NT + threadIdx.x;
    if (i < c.n_cand && !(c.merged[i] & OUTSIDE)) c.rows[off.a] = (int32_t)c.merged[i];
    if (b == nb - 1 && threadIdx.x == 0) *c.count = through.a;
}

// ---- the rows of a section inside a box without the grid (a box over more than MAX_RUN_CELLS cells) ------------------------
__global__ __launch_bounds__(256) void box_mask_kernel(const double *__restrict__ xy, int64_t n, double x0, double x1, double y0,
                                                        double y1, unsigned long long *__restrict__ mask) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool in = false;
    if (i < n) {
        const double2_t p = ld2(xy, i);
        in = p.x >= x0 && p.x < x1 && p.y >= y0 && p.y < y1;
    }
    const unsigned long long bal = __ballot(in);
    if ((threadIdx.x & 63) == 0) mask[i >> 6] = bal;
}
__global__ __launch_bounds__(scan::NT) void mask_compact_kernel(const unsigned long long *__restrict__ mask, int64_t n_words,
                                                                 unsigned long long *__restrict__ status, int32_t *__restrict__ rows,
                                                                 unsigned long long *__restrict__ count) {
    __shared__ scan::Shared sh;
    auto val = [&](int64_t w) { return Pair{w < n_words ? (unsigned)__builtin_popcountll(mask[w]) : 0u, 0u}; };
    Pair through;
    const Pair off = scan::exclusive(status, (int)blockIdx.x, val, sh, &through);
    const int64_t w = (int64_t)blockIdx.x * scan::NT + threadIdx.x;
    if (w < n_words) {
        unsigned long long bits = mask[w];
        unsigned pos = off.a;
        while (bits) {
            const int b = __builtin_ctzll(bits);
            bits &= bits - 1;
            rows[pos++] = (int32_t)(w * 64 + b);
        }
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *count = through.a;
}

// ---- compaction of the aligned side and of the pair list (src/utils.py:734-742): scan + scatter in one launch ---------------
// counts: [0] aligned rows in the box, [1] reference rows in the box, [2] aligned rows kept, [3] pairs
struct ScatterArgs {
    const int32_t *idx;        // [cap][k] reference SECTION rows, -1 padded
    const void *cost;          // [cap][k] in the cost type
    const int32_t *cnt;        // [cap]
    const int32_t *rows_m, *rows_r;
    const double *mov_xy, *mov_size;
    const int32_t *mov_type;   // or null
    unsigned long long *status, *counts;
    int32_t *ua, *rows_ua, *type_c, *prow, *pairs, *jsec;
    double *axy_c, *size_c, *cost64;
    int k;
    unsigned blocks;           // scan blocks of this window (its bound on the aligned rows)
};
template <typename F>
__global__ __launch_bounds__(scan::NT) void window_scatter_kernel(Batch<ScatterArgs> bt) {
    __shared__ scan::Shared sh;
    const ScatterArgs &s = bt.w[blockIdx.y];
    if (blockIdx.x >= s.blocks) return;
    const int64_t n_m = (int64_t)s.counts[0], n_r = (int64_t)s.counts[1];
    auto val = [&](int64_t i) {
        const int c = i < n_m ? s.cnt[i] : 0;
        return Pair{c > 0 ? 1u : 0u, c > 0 ? (unsigned)c : 0u};
    };
    Pair through;
    const Pair off = scan::exclusive(s.status, (int)blockIdx.x, val, sh, &through);
    const int64_t i = (int64_t)blockIdx.x * scan::NT + threadIdx.x;
    const int c = i < n_m ? s.cnt[i] : 0;
    if (c > 0) {
        const int32_t a = (int32_t)off.a, row = s.rows_m[i];
        s.ua[a] = (int32_t)i;
        s.rows_ua[a] = row;
        const double2_t p = ld2(s.mov_xy, row);
        s.axy_c[2 * (int64_t)a] = p.x;
        s.axy_c[2 * (int64_t)a + 1] = p.y;
        s.size_c[a] = s.mov_size[row];
        s.type_c[a] = s.mov_type ? s.mov_type[row] : 0;
        s.prow[a] = (int32_t)off.p;
        int64_t pp = off.p;
        const int64_t p_end = pp + c;                // the scan sized the list by cnt: never write past this row's share
        const F *cost = static_cast<const F *>(s.cost);
        // a reference cell's number in the window = its place in the ascending row list: lower bounds for eight candidates in lock
        // step (independent loads per step), then the pairs in list order
        constexpr int G = 8;
        for (int qb = 0; qb < s.k && pp < p_end; qb += G) {
            int32_t j[G];
            unsigned lb[G], n[G];
            bool any = false;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                j[g] = qb + g < s.k ? s.idx[i * s.k + qb + g] : -1;
                lb[g] = 0u;
                n[g] = j[g] >= 0 ? (unsigned)n_r : 0u;
                any = any || n[g] != 0u;
            }
            while (any) {
                int32_t v[G];
#pragma unroll
                for (int g = 0; g < G; ++g) v[g] = n[g] ? s.rows_r[lb[g] + (n[g] >> 1)] : 0;
                any = false;
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    if (n[g]) {
                        const unsigned half = n[g] >> 1;
                        if (v[g] < j[g]) { lb[g] += half + 1; n[g] -= half + 1; } else n[g] = half;
                        any = any || n[g] != 0u;
                    }
                }
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (j[g] >= 0 && pp < p_end) {
                    s.pairs[2 * pp] = a;
                    s.pairs[2 * pp + 1] = (int32_t)lb[g];
                    s.jsec[pp] = j[g];
                    s.cost64[pp] = (double)cost[i * s.k + qb + g];
                    ++pp;
                }
            }
        }
    }
    if (blockIdx.x == s.blocks - 1 && threadIdx.x == 0) {
        s.counts[2] = through.a;
        s.counts[3] = through.p;
        s.prow[through.a] = (int32_t)through.p;
    }
}

int ensure_host(same_window *w, size_t bytes) {
    if (bytes <= w->host_bytes) return SAME_OK;
    same_ctx *ctx = w->ctx;
    if (w->host) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipHostFree(w->host));
        w->host = nullptr;
        w->host_dev = nullptr;
        w->host_bytes = 0;
    }
    const size_t want = bytes + bytes / 4 + 4096;
    HIP_TRY(ctx, hipHostMalloc(&w->host, want, hipHostMallocDefault));
    w->host_bytes = want;
    void *dev = nullptr;
    w->host_dev = hipHostGetDevicePointer(&dev, w->host, 0) == hipSuccess ? static_cast<char *>(dev) : nullptr;
    if (!w->host_dev) (void)hipGetLastError();
    return SAME_OK;
}

// rows of `sec` inside the box through a mask over ALL its rows (boxes that cover more cells than the run path takes): ascending
// into dst (sized for the section), their number into *out_n.  A wait of its own: this is not the window loop's path.
int subset_rows_full(same_window *w, const same_section *sec, const double *box, DevBuf &dst, int64_t *out_n) {
    same_ctx *ctx = w->ctx;
    *out_n = 0;
    if (sec->n == 0) return SAME_OK;
    const int64_t n_words = (int64_t)grid_for(sec->n) * 4;
    Carver cv;
    const size_t o_status = cv.take(scan::status_bytes(n_words)), o_count = cv.take(16);
    const size_t zero_bytes = cv.off;
    const size_t o_mask = cv.take((size_t)n_words * 8);
    SAME_TRY(ensure(ctx, w->big_mask, cv.off));
    char *base = static_cast<char *>(w->big_mask.p);
    unsigned long long *status = reinterpret_cast<unsigned long long *>(base + o_status), *dcount = reinterpret_cast<unsigned long long *>(base + o_count),
                       *mask = reinterpret_cast<unsigned long long *>(base + o_mask);
    SAME_TRY(ensure(ctx, dst, (size_t)sec->n * sizeof(int32_t)));
    SAME_FILL(ctx, base, 0, zero_bytes);
    SAME_LAUNCH(ctx, box_mask_kernel, dim3(grid_for(sec->n)), dim3(256), 0, sec->xy, sec->n, box[0], box[1], box[2], box[3], mask);
    SAME_LAUNCH(ctx, mask_compact_kernel, dim3(scan::blocks_for(n_words)), dim3(scan::NT), 0, mask, n_words, scan::arg(status),
                static_cast<int32_t *>(dst.p), dcount);
    HIP_TRY(ctx, hipGetLastError());
    unsigned long long *h = static_cast<unsigned long long *>(w->host);
    SAME_COPY(ctx, h, dcount, sizeof(unsigned long long), hipMemcpyDeviceToHost);
    SAME_WAIT(ctx);
    *out_n = (int64_t)h[0];
    return SAME_OK;
}

__global__ void set_count_kernel(unsigned long long *p, unsigned long long v) { *p = v; }

}  // namespace

extern "C" {

int same_window_create(same_ctx *ctx, same_window **out) {
    REQUIRE(ctx, ctx && out);
    *out = nullptr;
    SAME_TRY(same_use(ctx));
    same_window *w = new (std::nothrow) same_window();
    if (!w) return SAME_ENOMEM;
    w->ctx = ctx;
    *out = w;
    SAME_TRY(ensure_host(w, 1 << 16));
    return SAME_OK;
}

void same_window_destroy(same_window *w) {
    if (!w) return;
    (void)hipSetDevice(w->ctx->device);
    (void)hipStreamSynchronize(w->ctx->stream);
    for (DevBuf *b : {&w->stage, &w->filter, &w->finish, &w->tris, &w->big_mask, &w->full_m, &w->full_r}) release(*b);
    if (w->host) (void)hipHostFree(w->host);
    delete w;
}

}  // extern "C"

namespace win {

// what a group of windows hands back, written straight into the windows' pinned host blocks (device-accessible: hipHostMalloc) in one
// launch: up to two 16-byte aligned regions per window.  The stream's wait makes them the host's.
int launch_copy_back(same_ctx *ctx, const CopyArgs *regions, int n_w) {
    Batch<CopyArgs> cb{};
    size_t most = 0;
    for (int q = 0; q < n_w; ++q) {
        cb.w[q] = regions[q];
        most = std::max(most, std::max(regions[q].bytes[0], regions[q].bytes[1]));
    }
    if (most == 0) return SAME_OK;
    const unsigned blocks = (unsigned)std::min<size_t>(64, (most + 4095) / 4096);
    SAME_LAUNCH(ctx, copy_back_kernel, dim3(blocks, (unsigned)n_w), dim3(256), 0, cb);
    return SAME_OK;
}

// zero_kernel over a group's regions (grid-stride: at most 128 blocks per window)
int launch_zero(same_ctx *ctx, const ZeroArgs *regions, int n_w) {
    Batch<ZeroArgs> zb{};
    size_t most = 0;
    for (int q = 0; q < n_w; ++q) {
        zb.w[q] = regions[q];
        most = std::max(most, std::max(regions[q].bytes[0], regions[q].bytes[1]));
    }
    if (most == 0) return SAME_OK;
    const unsigned blocks = (unsigned)std::min<size_t>(128, (most + 4095) / 4096);
    SAME_LAUNCH(ctx, zero_kernel, dim3(blocks, (unsigned)n_w), dim3(256), 0, zb);
    return SAME_OK;
}

// a batch call's windows: one context, no window twice
int check_batch(same_window *const *windows, int n_windows, same_ctx **out_ctx) {
    if (!windows || n_windows < 1 || !windows[0]) return SAME_EINVAL;
    same_ctx *ctx = windows[0]->ctx;
    REQUIRE(ctx, n_windows <= SAME_WINDOW_BATCH_MAX);
    for (int i = 0; i < n_windows; ++i) {
        REQUIRE(ctx, windows[i] && windows[i]->ctx == ctx);
        for (int j = 0; j < i; ++j) REQUIRE(ctx, windows[j] != windows[i]);
    }
    *out_ctx = ctx;
    return SAME_OK;
}

}  // namespace win

namespace {

struct StagePlan {
    ZeroArgs zero{};            // the head of the stage buffer
    int64_t full_m = 0, full_r = 0;   // whole-section path: the lists' lengths, copied in after the zeroing
    size_t back_bytes = 0, slots = 0;
    RowsArgs rows{};            // window_rows_kernel's share (blocks == 0: nothing to walk)
    CompactArgs compact{};      // rows_compact_kernel's share (blocks == 0: the box is a union of cells, or empty)
    same_knn_window_job knn;
    same_cost_window_job cost;
    ScatterArgs scatter{};
};

// One window's stage buffer laid out and zeroed (the whole-section path's lists copied in); no launch -- those come per GROUP of windows
// (launch_stage).  The caller holds the sections' grid locks.
int prepare_stage(same_window *w, const same_section *mov, const same_section *ref, const double *box, int k, StagePlan *sp) {
    same_ctx *ctx = w->ctx;
    w->staged = w->finished = w->filtered = 0;
    w->mov = mov;
    w->ref = ref;
    w->has_type = mov->type_id != nullptr;
    w->cost_f32 = mov->cost_f32;
    w->k = k;
    w->n_m = w->n_r = w->n_ua = w->P = w->Tr = 0;
    // the candidates: rows of the cells the box covers (their number is known here), or a mask over the whole section
    Cover cm = cover_of(mov, box), cr = cover_of(ref, box);
    int64_t cap_m = cm.n_cand, cap_r = cr.n_cand;
    if (!cm.use_runs) SAME_TRY(subset_rows_full(w, mov, box, w->full_m, &cap_m));     // rare: a box over more than MAX_RUN_CELLS cells
    if (!cr.use_runs) SAME_TRY(subset_rows_full(w, ref, box, w->full_r, &cap_r));
    REQUIRE(ctx, cap_m * (int64_t)k < ((int64_t)1 << 31) - 1 && cap_r < ((int64_t)1 << 31) - 1);   // offsets and scan totals are 31-bit
    w->cap_m = cap_m;
    w->cap_r = cap_r;
    const size_t cs = w->cost_f32 ? sizeof(float) : sizeof(double);
    const size_t slots = (size_t)cap_m * k, cm1 = (size_t)cap_m + 1;
    const bool compact_m = cm.use_runs && !cm.aligned && cap_m > 0, compact_r = cr.use_runs && !cr.aligned && cap_r > 0;
    // layout: [scan words | counts | kept XY | kept rows] (zeroed up to the counts; copied back from the counts on) | the rest
    Carver cv;
    const size_t st_scatter = scan::status_bytes(cap_m), st_cm = scan::status_bytes(cap_m), st_cr = scan::status_bytes(cap_r);
    const size_t o_st_scatter = cv.take(st_scatter), o_st_cm = cv.take(st_cm), o_st_cr = cv.take(st_cr);
    const size_t o_counts = cv.off;
    cv.off += 64;
    const size_t o_axy_c = cv.off;
    cv.off += (size_t)cap_m * 2 * sizeof(double);
    const size_t o_rows_ua = cv.off;
    cv.off += (size_t)cap_m * sizeof(int32_t);
    const size_t back_bytes = cv.off - o_counts;
    cv.off = (cv.off + 255) & ~size_t(255);
    const size_t o_merged_m = cv.take((size_t)cap_m * 4), o_merged_r = cv.take((size_t)cap_r * 4);
    const size_t o_rows_m = cv.take((size_t)cap_m * 4), o_rows_r = cv.take((size_t)cap_r * 4);
    const size_t o_idx = cv.take(slots * 4), o_cnt = cv.take((size_t)cap_m * 4), o_cost = cv.take(slots * cs);
    const size_t o_ua = cv.take((size_t)cap_m * 4), o_type_c = cv.take((size_t)cap_m * 4), o_size_c = cv.take((size_t)cap_m * 8);
    const size_t o_prow = cv.take(cm1 * 4), o_pairs = cv.take(slots * 8), o_jsec = cv.take(slots * 4), o_cost64 = cv.take(slots * 8);
    SAME_TRY(ensure(ctx, w->stage, cv.off));
    // everything the three calls of this window copy back fits the pinned block from now on (it must not move between them)
    w->host_finish_off = (back_bytes + 255) & ~size_t(255);
    w->host_filter_off = w->host_finish_off + ((SAME_GREEDY_BATCH_MAX * 8 + 128 + (size_t)cap_m * 5 + 64 + 255) & ~size_t(255));
    SAME_TRY(ensure_host(w, w->host_filter_off + 256));
    char *base = static_cast<char *>(w->stage.p);
    auto at = [&](size_t off) { return base + off; };
    w->counts = reinterpret_cast<unsigned long long *>(at(o_counts));
    w->axy_c = reinterpret_cast<double *>(at(o_axy_c));
    w->rows_ua = reinterpret_cast<int32_t *>(at(o_rows_ua));
    uint32_t *merged_m = reinterpret_cast<uint32_t *>(at(o_merged_m)), *merged_r = reinterpret_cast<uint32_t *>(at(o_merged_r));
    // aligned boxes: the merged list IS the row list; whole-section path: its own list is copied in
    w->rows_m = reinterpret_cast<int32_t *>(compact_m || !cm.use_runs ? at(o_rows_m) : at(o_merged_m));
    w->rows_r = reinterpret_cast<int32_t *>(compact_r || !cr.use_runs ? at(o_rows_r) : at(o_merged_r));
    w->idx = reinterpret_cast<int32_t *>(at(o_idx));
    w->cnt = reinterpret_cast<int32_t *>(at(o_cnt));
    void *cost = at(o_cost);
    w->ua = reinterpret_cast<int32_t *>(at(o_ua));
    w->type_c = reinterpret_cast<int32_t *>(at(o_type_c));
    w->size_c = reinterpret_cast<double *>(at(o_size_c));
    w->prow = reinterpret_cast<int32_t *>(at(o_prow));
    w->pairs = reinterpret_cast<int32_t *>(at(o_pairs));
    w->jsec = reinterpret_cast<int32_t *>(at(o_jsec));
    w->cost64 = reinterpret_cast<double *>(at(o_cost64));
    unsigned long long *dc = w->counts;

    sp->zero = ZeroArgs{{base, nullptr}, {o_counts + 64, 0}};       // scan words + counts, zeroed with the group's (launch_stage)
    sp->full_m = !cm.use_runs ? cap_m : 0;        // the whole-section path's list and count take their places after that
    sp->full_r = !cr.use_runs ? cap_r : 0;
    RowsArgs &ra = sp->rows;
    ra = RowsArgs{};
    ra.bx0 = box[0]; ra.bx1 = box[1]; ra.by0 = box[2]; ra.by1 = box[3];
    unsigned bm = 0, br = 0;
    if (cm.use_runs && cap_m) {
        ra.dm = RunDesc{mov->order, mov->starts, mov->xy, mov->grid.nx, cm.cx0, cm.ncx, cm.cy0, cm.ncy, (int)cap_m, cm.aligned ? 1 : 0, merged_m, dc};
        bm = grid_for(cap_m);
    }
    if (cr.use_runs && cap_r) {
        ra.dr = RunDesc{ref->order, ref->starts, ref->xy, ref->grid.nx, cr.cx0, cr.ncx, cr.cy0, cr.ncy, (int)cap_r, cr.aligned ? 1 : 0, merged_r, dc + 1};
        br = grid_for(cap_r);
    }
    ra.blocks_m = bm;
    ra.blocks = bm + br;
    CompactArgs &ca = sp->compact;
    ca = CompactArgs{};
    if (compact_m || compact_r) {
        ca.cm = RowsCompact{merged_m, compact_m ? (int)cap_m : 0, w->rows_m, scan::arg(reinterpret_cast<unsigned long long *>(at(o_st_cm))), dc};
        ca.cr = RowsCompact{merged_r, compact_r ? (int)cap_r : 0, w->rows_r, scan::arg(reinterpret_cast<unsigned long long *>(at(o_st_cr))), dc + 1};
        ca.blocks_m = compact_m ? scan::blocks_for(cap_m) : 0;
        ca.blocks = ca.blocks_m + (compact_r ? scan::blocks_for(cap_r) : 0);
    }
    sp->knn = same_knn_window_job{};
    sp->cost = same_cost_window_job{};
    sp->scatter = ScatterArgs{};
    if (cap_m) {
        sp->knn.rows_m = w->rows_m; sp->knn.dn_m = dc; sp->knn.cap_m = cap_m; sp->knn.rows_r = w->rows_r; sp->knn.dn_r = dc + 1;
        for (int q = 0; q < 4; ++q) sp->knn.box[q] = box[q];
        sp->knn.idx = w->idx; sp->knn.cnt = w->cnt;
        sp->cost.rows = w->rows_m; sp->cost.dn = dc; sp->cost.cap = cap_m; sp->cost.idx = w->idx; sp->cost.out = cost;
        sp->scatter = ScatterArgs{w->idx, cost, w->cnt, w->rows_m, w->rows_r, mov->xy, mov->size, mov->type_id,
                                  scan::arg(reinterpret_cast<unsigned long long *>(at(o_st_scatter))), dc, w->ua, w->rows_ua, w->type_c, w->prow, w->pairs,
                                  w->jsec, w->axy_c, w->size_c, w->cost64, k, scan::blocks_for(cap_m)};
    } else {
        sp->knn.dn_m = dc; sp->knn.dn_r = dc + 1;     // a window without candidates still names its (zero) counts: its blocks read them and leave
        sp->cost.dn = dc;
    }
    sp->back_bytes = back_bytes;
    sp->slots = slots;
    return SAME_OK;
}

// the stage kernels of a group of prepared windows (<= SAME_LAUNCH_WINDOWS, one pair of sections): row lists, prune, candidate costs,
// compaction -- one launch each for the whole group
int launch_stage(same_ctx *ctx, same_window *const *ws, StagePlan *const *sps, int n_w, const same_section *mov, const same_section *ref,
                 const same_knn_index *ix, int k, double dist_ct_coeff) {
    Batch<RowsArgs> rb{};
    Batch<CompactArgs> cb{};
    Batch<ScatterArgs> sb{};
    same_knn_window_job kj[SAME_LAUNCH_WINDOWS];
    same_cost_window_job cj[SAME_LAUNCH_WINDOWS];
    unsigned max_rows = 0, max_compact = 0, max_scatter = 0;
    for (int q = 0; q < n_w; ++q) {
        rb.w[q] = sps[q]->rows;
        cb.w[q] = sps[q]->compact;
        sb.w[q] = sps[q]->scatter;
        kj[q] = sps[q]->knn;
        cj[q] = sps[q]->cost;
        max_rows = std::max(max_rows, sps[q]->rows.blocks);
        max_compact = std::max(max_compact, sps[q]->compact.blocks);
        max_scatter = std::max(max_scatter, sps[q]->scatter.blocks);
    }
    const unsigned nw = (unsigned)n_w;
    ZeroArgs zr[SAME_LAUNCH_WINDOWS];
    for (int q = 0; q < n_w; ++q) zr[q] = sps[q]->zero;
    SAME_TRY(launch_zero(ctx, zr, n_w));
    for (int q = 0; q < n_w; ++q) {
        same_window *w = ws[q];
        if (sps[q]->full_m) {
            SAME_COPY(ctx, w->rows_m, w->full_m.p, (size_t)sps[q]->full_m * 4, hipMemcpyDeviceToDevice);
            SAME_LAUNCH(ctx, set_count_kernel, dim3(1), dim3(1), 0, w->counts, (unsigned long long)sps[q]->full_m);
        }
        if (sps[q]->full_r) {
            SAME_COPY(ctx, w->rows_r, w->full_r.p, (size_t)sps[q]->full_r * 4, hipMemcpyDeviceToDevice);
            SAME_LAUNCH(ctx, set_count_kernel, dim3(1), dim3(1), 0, w->counts + 1, (unsigned long long)sps[q]->full_r);
        }
    }
    if (max_rows) SAME_LAUNCH(ctx, window_rows_kernel, dim3(max_rows, nw), dim3(256), 0, rb);
    if (max_compact) SAME_LAUNCH(ctx, rows_compact_kernel, dim3(max_compact, nw), dim3(scan::NT), 0, cb);
    if (max_scatter) {
        SAME_TRY(same_knn_window_batch_core(ctx, ix, mov->xy, kj, n_w, k));
        SAME_TRY(same_padded_cost_window_batch_core(ctx, ws[0]->cost_f32, mov->types_c, ref->types_c, mov->T, mov->xy_c, ref->xy_c, cj, n_w, k, dist_ct_coeff));
        if (ws[0]->cost_f32)
            SAME_LAUNCH(ctx, window_scatter_kernel<float>, dim3(max_scatter, nw), dim3(scan::NT), 0, sb);
        else
            SAME_LAUNCH(ctx, window_scatter_kernel<double>, dim3(max_scatter, nw), dim3(scan::NT), 0, sb);
    }
    HIP_TRY(ctx, hipGetLastError());
    return SAME_OK;
}

// after the wait: the counts the copy brought back
int collect_stage(same_window *w, const StagePlan &sp, int64_t *out_counts) {
    same_ctx *ctx = w->ctx;
    const unsigned long long *tot = reinterpret_cast<const unsigned long long *>(w->host);
    w->n_m = (int64_t)tot[0];
    w->n_r = (int64_t)tot[1];
    w->n_ua = (int64_t)tot[2];
    w->P = (int64_t)tot[3];
    REQUIRE(ctx, w->n_m <= w->cap_m && w->n_r <= w->cap_r && w->n_ua <= w->n_m && w->P <= (int64_t)sp.slots);
    out_counts[0] = w->n_m;
    out_counts[1] = w->n_r;
    out_counts[2] = w->n_ua;
    out_counts[3] = w->P;
    w->staged = (w->n_m && w->n_r) ? 2 : 1;      // no pairs possible: the caller raises what run_same raises (src/same.py:1003)
    return SAME_OK;
}

}  // namespace

extern "C" {

int same_window_stage(same_window *const *windows, int n_windows, const same_section *mov, const same_section *ref, const double *boxes,
                      double radius, int k, double dist_ct_coeff, int64_t *out_counts) {
    same_ctx *ctx = nullptr;
    SAME_TRY(check_batch(windows, n_windows, &ctx));
    REQUIRE(ctx, mov && ref && boxes && out_counts && mov->ctx->device == ctx->device && ref->ctx->device == ctx->device);
    REQUIRE(ctx, mov->T == ref->T && mov->cost_f32 == ref->cost_f32 && k >= 1 && k <= SAME_MAX_KNN && radius >= 0.0);
    SAME_TRY(same_use(ctx));
    for (int i = 0; i < 4 * n_windows; ++i) out_counts[i] = 0;
    for (int i = 0; i < n_windows; ++i) windows[i]->staged = windows[i]->finished = windows[i]->filtered = 0;
    std::shared_ptr<same_knn_index> ix;           // held until this call's kernels have finished (both returns below wait first)
    SAME_TRY(knn_index_for(ctx, ref, radius, &ix));
    // both sections' grids stay as they are until this call's kernels are enqueued (same_section_bin waits for this, then for the device)
    std::shared_lock<std::shared_mutex> grid_m(const_cast<same_section *>(mov)->grid_lock), grid_r;
    if (ref != mov) grid_r = std::shared_lock<std::shared_mutex>(const_cast<same_section *>(ref)->grid_lock);
    // every window's buffer is laid out and zeroed, then the kernels run per group of SAME_LAUNCH_WINDOWS windows (one launch each for the
    // whole group), then every window's copy back; ONE wait for the batch
    std::vector<StagePlan> plans((size_t)n_windows);
    int rc = SAME_OK;
    for (int i = 0; i < n_windows && rc == SAME_OK; ++i) rc = prepare_stage(windows[i], mov, ref, boxes + 4 * i, k, &plans[(size_t)i]);
    for (int g = 0; g < n_windows && rc == SAME_OK; g += SAME_LAUNCH_WINDOWS) {
        StagePlan *sps[SAME_LAUNCH_WINDOWS];
        const int n_g = std::min(SAME_LAUNCH_WINDOWS, n_windows - g);
        for (int q = 0; q < n_g; ++q) sps[q] = &plans[(size_t)(g + q)];
        rc = launch_stage(ctx, windows + g, sps, n_g, mov, ref, ix.get(), k, dist_ct_coeff);
    }
    // what comes back per window -- the four counts, then the kept aligned rows' XY and section rows at the capacity cap_m -- in one launch
    // per group (straight into the pinned blocks), or one copy per window where a block is not device-addressable
    for (int g = 0; g < n_windows && rc == SAME_OK; g += SAME_LAUNCH_WINDOWS) {
        CopyArgs ca[SAME_LAUNCH_WINDOWS];
        const int n_g = std::min(SAME_LAUNCH_WINDOWS, n_windows - g);
        for (int q = 0; q < n_g && rc == SAME_OK; ++q) {
            same_window *w = windows[g + q];
            ca[q] = CopyArgs{};
            if (w->host_dev) {
                ca[q] = CopyArgs{{w->counts, nullptr}, {w->host_dev, nullptr}, {plans[(size_t)(g + q)].back_bytes, 0}};
                continue;
            }
            hipError_t e = hipMemcpyAsync(w->host, w->counts, plans[(size_t)(g + q)].back_bytes, hipMemcpyDeviceToHost, ctx->stream);
            ++ctx->stats[SAME_STAT_COPIES];
            if (e != hipSuccess) rc = same_fail(ctx, SAME_EIO, "stage copy back", e);
        }
        if (rc == SAME_OK) rc = launch_copy_back(ctx, ca, n_g);
    }
    if (rc != SAME_OK) {                          // nothing of a failed batch counts; what was enqueued is waited for before returning
        (void)hipStreamSynchronize(ctx->stream);
        for (int i = 0; i < n_windows; ++i) windows[i]->staged = 0;
        return rc;
    }
    SAME_WAIT(ctx);
    for (int i = 0; i < n_windows; ++i) SAME_TRY(collect_stage(windows[i], plans[(size_t)i], out_counts + 4 * i));
    return SAME_OK;
}

int same_window_fetch(same_window *w, int what, void *out, int64_t bytes) {
    if (!w) return SAME_EINVAL;
    same_ctx *ctx = w->ctx;
    REQUIRE(ctx, w->staged >= 1 && bytes >= 0 && (bytes == 0 || out));
    SAME_TRY(same_use(ctx));
    const char *h = static_cast<const char *>(w->host);
    const void *dev = nullptr;
    const void *host = nullptr;
    int64_t want = 0;
    const int64_t n_m = w->n_m, n_r = w->n_r, n_ua = w->n_ua, P = w->P, Tr = w->Tr;
    const bool full = w->staged == 2;
    switch (what) {
    case SAME_WINDOW_ALIGNED_XY: want = n_ua * 16; host = h + 64; REQUIRE(ctx, full || n_ua == 0); break;
    case SAME_WINDOW_ALIGNED_ROWS: want = n_ua * 4; host = h + 64 + (size_t)w->cap_m * 16; REQUIRE(ctx, full || n_ua == 0); break;
    case SAME_WINDOW_ROWS_M: want = n_m * 4; dev = w->rows_m; break;
    case SAME_WINDOW_ROWS_R: want = n_r * 4; dev = w->rows_r; break;
    case SAME_WINDOW_PAIRS: want = P * 8; dev = w->pairs; break;
    case SAME_WINDOW_COSTS: want = P * 8; dev = w->cost64; break;
    case SAME_WINDOW_KEPT: want = n_ua * 4; dev = w->ua; break;
    case SAME_WINDOW_SIGNS: want = Tr; dev = w->sign; REQUIRE(ctx, w->finished); break;
    case SAME_WINDOW_WEIGHTS: want = Tr * 8; dev = w->weight; REQUIRE(ctx, w->finished); break;
    case SAME_WINDOW_MATCH: want = n_ua * 4; dev = w->match_loc; REQUIRE(ctx, w->finished); break;
    case SAME_WINDOW_TRIANGLES: want = Tr * 12; dev = w->tris.p; REQUIRE(ctx, w->finished || w->filtered); break;
    default: REQUIRE(ctx, !"unknown same_window_fetch selector");
    }
    REQUIRE(ctx, bytes == want);
    if (want == 0) return SAME_OK;
    if (host) {                                   // already on the host since the stage call's own copy
        memcpy(out, host, (size_t)want);
        return SAME_OK;
    }
    HIP_TRY(ctx, hipMemcpyAsync(out, dev, (size_t)want, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SAME_OK;
}

}  // extern "C"
