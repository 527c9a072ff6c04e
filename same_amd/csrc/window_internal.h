// window_internal.h -- what the translation units of the device-resident window path share (section.hip, window_stage.hip,
// window_finish.hip, window_merge.hip): the resident objects behind the opaque handles of include/same_hip.h part 4, the
// per-launch argument blocks, the small host helpers.
//
// The window path (src/same.py:507-593 with both sections RESIDENT on the device).  A section's columns are uploaded once
// (same_section) and its rows are binned once into a grid of cells (same_section_bin: rows sorted by cell, ascending inside a cell)
// -- SURVEY a13's "one-pass bin".  A BATCH of windows is then two calls (window_stage.hip, window_finish.hip), and a third that
// enqueues only (window_merge.hip: the rows of the windows' central regions appended to the pass's merge accumulator).
//
// Every kernel takes up to SAME_LAUNCH_WINDOWS (8) windows per launch: blockIdx.y = window, the per-window arguments -- pointers into
// the window's OWN buffers, its counts -- travel by value in the kernarg segment (Batch<Args>), the grid is sized by the group's largest
// window and blocks beyond a window's share leave at once.  Nothing a call computes is sized by a number the host has to wait for:
// lists are allocated for the candidates of the covered cells (known from the host's copy of the cell offsets), their true lengths
// stay in a counter block on the device and every kernel reads them there.  same_ctx_stat counts the runtime calls.
//
// WHO MAY TOUCH WHAT -- the threading contract of this path.  A context (same_ctx) is one stream and is used by one thread at a time
// (the Python binding holds Context.lock around every call).  Windows and merge accumulators belong to ONE context.  Sections are
// shared by the worker threads' contexts; two locks guard them:
//
//   same_section::grid_lock  (std::shared_mutex)   guards  grid, order, starts, h_starts, n_binned.
//       shared:     every same_window_stage call, from cover_of() until its last kernel is ENQUEUED (the kernels read `order` /
//                   `starts` by the pointers they were launched with).
//       exclusive:  same_section_bin's swap of the five fields.  It first builds the new index WITHOUT the lock, then takes the lock
//                   (no stage call is between cover_of() and its last enqueue), waits for the DEVICE (hipDeviceSynchronize: kernels
//                   enqueued earlier by any context may still read the old arrays), frees the old arrays and stores the new ones.
//   same_section::lock       (std::mutex)          guards  the table `knn` (radius -> prune index, most recently used first, <= 16).
//       held only for look-ups, insertions and evictions of table entries -- NEVER across device work: an index that is not in the
//       table is built by the calling thread on its own context with the lock released, and inserted afterwards (a thread that lost the
//       race to another builder of the same radius drops its build and takes the winner's).  Entries are shared_ptr: a stage call keeps
//       the index it prunes with until its kernels have finished, so an entry evicted meanwhile is freed by its last user, after
//       the lock is let go (hipFree waits for the device).
//
//   Nothing else of a section changes after same_section_create (xy, types, sizes, type codes, id codes are written once, before the
//   handle is handed out; same_section_set_codes must precede the first collect call that reads the codes).
//   same_section_destroy expects that no call is using the section (it waits for the device, not for threads).
#pragma once
#include <algorithm>
#include <cmath>
#include <memory>
#include <mutex>
#include <new>
#include <shared_mutex>
#include <utility>
#include <vector>

#include "common.h"
#include "devmath.h"
#include "scan.h"

struct same_knn_index;
extern "C" int same_knn_index_build(same_ctx *ctx, const double *drxy, int64_t n_r, double radius, same_knn_index **out);
extern "C" void same_knn_index_destroy(same_knn_index *ix);

namespace win {

constexpr int MAX_RUN_CELLS = 64;        // cells of a section's grid one window may cover on the cell-run path
constexpr int64_t MAX_GRID_CELLS = (int64_t)1 << 22;
constexpr unsigned OUTSIDE = 0x80000000u;   // flag on a candidate row that fails the box test

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
};

inline int ensure(same_ctx *ctx, DevBuf &b, size_t bytes) {
    if (bytes <= b.bytes && b.p) return SAME_OK;
    const size_t want = std::max<size_t>(bytes + bytes / 4, 4096);
    if (b.p) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipFree(b.p));
        b.p = nullptr;
        b.bytes = 0;
    }
    HIP_TRY(ctx, hipMalloc(&b.p, want));
    b.bytes = want;
    return SAME_OK;
}
inline void release(DevBuf &b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
}

// carve a buffer: offsets are multiples of 256 bytes
struct Carver {
    size_t off = 0;
    size_t take(size_t bytes) {
        const size_t at = off;
        off += (bytes + 255) & ~size_t(255);
        return at;
    }
};

inline unsigned grid_for(int64_t n) { return (unsigned)ceil_div(n > 0 ? n : 1, 256); }

// ---- the section's grid of cells -------------------------------------------------------------------------------------------
// cell (cx, cy) = [x0 + cx*cw, x0 + (cx+1)*cw) x [y0 + cy*ch, y0 + (cy+1)*ch): a row belongs to the cell whose edges -- these
// very doubles -- bracket it under the comparisons of src/same.py:293-295, so a box whose edges are cell edges needs no test.
struct BinGrid {
    double x0 = 0.0, y0 = 0.0, cw = 1.0, ch = 1.0;
    int nx = 1, ny = 1;
};
__host__ __device__ inline double cell_edge(double origin, double width, int c) { return origin + (double)c * width; }

// Every kernel of the window calls takes the windows of a batch in ONE launch: blockIdx.y = window, the per-window arguments travel by
// value in the kernarg segment (Batch<A>, at most SAME_LAUNCH_WINDOWS windows); the grid is sized by the largest window and blocks
// beyond a window's own share leave at once.
template <typename A>
struct Batch {
    A w[SAME_LAUNCH_WINDOWS];
};

// the heads of the windows' buffers zeroed in one launch (scan words, counters, marks): up to two 16-byte aligned regions per window
struct ZeroArgs {
    void *p[2];
    size_t bytes[2];
};

struct CopyArgs {
    const void *src[2];
    void *dst[2];
    size_t bytes[2];
};

// covered cells of a box in a section's grid, and whether the box is exactly their union (section.hip)
struct Cover {
    int cx0 = 0, ncx = 0, cy0 = 0, ncy = 0;
    int64_t n_cand = 0;
    bool aligned = false, use_runs = false;
};

// counters of the filter: [0] kept (class 0), [1] added back, [2] cosines within `tol` of the threshold, [3] triangles left
enum { FC_KEEP = 0, FC_ADD = 1, FC_NEAR = 2, FC_TR = 3, FC_TIES = 4, FC_COPIED = 5 };
// counters of the finish call: [0] orientation checked, [1] flipped, [2] XY comparisons, [3] XY violations, [4] triangles with
// one, [5] area flips, [6] (host) greedy rounds, [7] matched aligned cells, [8] pairs the greedy rule could still take
enum { SC_CHECKED = 0, SC_FLIPPED = 1, SC_CMP = 2, SC_VIOL = 3, SC_TVIOL = 4, SC_AFLIP = 5, SC_ROUNDS = 6, SC_MATCHED = 7, SC_REMAINING = 8, SC_TIES = 9, SC_COUNT = 16 };

}  // namespace win

struct same_section {
    same_ctx *ctx = nullptr;
    int64_t n = 0;
    int T = 0;
    int cost_f32 = 0;
    double *xy = nullptr;     // [n][2]
    void *xy_c = nullptr;     // [n][2] in the cost type (== xy for fp64 costs)
    void *types_c = nullptr;  // [n][T] in the cost type
    double *types64 = nullptr; // [n][T] as the caller gave them (== types_c for fp64 costs): what the result table's type columns are read from
    double *size = nullptr;   // [n]
    int32_t *type_id = nullptr;  // [n] codes of the cell type (equal type <=> equal code), or none
    int32_t *id_codes = nullptr; // [n] rank of the row's cell id among the frame's ids (same_section_set_codes), or none: a row's code is its number
    int64_t n_codes = 0;
    // the grid of cells (same_section_bin)
    win::BinGrid grid;
    int32_t *order = nullptr;        // [n_binned] rows by cell, ascending inside a cell
    unsigned *starts = nullptr;      // [cells + 1] on the device ...
    std::vector<unsigned> h_starts;  // ... and on the host: a window's candidate count is known without asking the device
    int64_t n_binned = 0;
    // prune indices of this section as the REFERENCE side, one per radius used (built on first use, under the lock: sections are
    // shared by the worker threads' contexts)
    std::mutex lock;
    // most recently used first; at most MAX_KNN_INDICES (the oldest is dropped).  Shared: a stage call holds the index it prunes with
    // until its kernels have finished, so an index dropped from the table meanwhile is freed by its last user
    std::vector<std::pair<double, std::shared_ptr<same_knn_index>>> knn;
    // grid, order, starts, h_starts: read (shared) by every stage call from cover_of() until its kernels are enqueued, replaced
    // (exclusive, after a device-wide wait: kernels enqueued earlier may still be reading the old arrays) by same_section_bin
    std::shared_mutex grid_lock;
};

struct same_window {
    same_ctx *ctx = nullptr;
    const same_section *mov = nullptr, *ref = nullptr;
    int cost_f32 = 0, k = 0, staged = 0, finished = 0, has_type = 0, filtered = 0;
    int64_t cap_m = 0, cap_r = 0;                   // candidates of the covered cells: what the lists are sized for
    int64_t n_m = 0, n_r = 0, n_ua = 0, P = 0, Tr = 0;
    win::DevBuf stage, filter, finish, tris, big_mask, full_m, full_r;
    // stage block
    unsigned long long *counts = nullptr;           // [8], first words of the block the stage call copies back
    int32_t *rows_m = nullptr, *rows_r = nullptr, *idx = nullptr, *cnt = nullptr, *ua = nullptr, *rows_ua = nullptr, *type_c = nullptr,
            *prow = nullptr, *pairs = nullptr, *jsec = nullptr;
    double *axy_c = nullptr, *size_c = nullptr, *cost64 = nullptr;
    // finish block
    int8_t *sign = nullptr;
    double *weight = nullptr;
    int32_t *match_loc = nullptr;
    int32_t *match_row = nullptr;   // [n_ua] section row of the matched reference cell (-1 = none), flag byte per kept cell: what
    uint8_t *pflag = nullptr;       // same_window_collect reads after the finish call
    void *host = nullptr;     // pinned staging for everything that comes back
    char *host_dev = nullptr; // the same block as the device addresses it (null: not addressable -- copies go through the copy engine)
    size_t host_bytes = 0;
    size_t host_finish_off = 0;   // the pinned block: [stage call's copy | finish call's copy | the filter's counters]
    size_t host_filter_off = 0;
};

namespace win {

// section.hip
int knn_index_for(same_ctx *ctx, const same_section *ref, double radius, std::shared_ptr<same_knn_index> *out);
Cover cover_of(const same_section *s, const double *box);
// window_stage.hip
int launch_copy_back(same_ctx *ctx, const CopyArgs *regions, int n_w);
int launch_zero(same_ctx *ctx, const ZeroArgs *regions, int n_w);
int check_batch(same_window *const *windows, int n_windows, same_ctx **out_ctx);

}  // namespace win
