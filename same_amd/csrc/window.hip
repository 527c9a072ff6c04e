// window.hip -- the windows of the sliding-window path (src/same.py:507-593) with both sections RESIDENT on the device.
//
// A section's columns are uploaded once (same_section) and its rows are binned once into a grid of cells
// (same_section_bin: rows sorted by cell, ascending inside a cell) -- SURVEY a13's "one-pass bin".  A BATCH of windows is then two calls:
//
//   same_window_stage          per window the rows of both sections inside the box, ascending (= np.flatnonzero of src/same.py:293-295),
//                              built from the few cells the box covers: a row's place in the list is the number of smaller rows in those
//                              cells (one binary search per cell), so the call reads O(window) rows whatever the section's size; radius / k
//                              prune against the reference SECTION's grid index (src/utils.py:709-728; candidates outside the box are
//                              not candidates), costs of the candidate lists in the cost type (src/same.py:1180-1189), compaction of the
//                              aligned side and of the pair list (src/utils.py:734-742).  Back to the host: four counts, the kept
//                              aligned rows and their XY -- the input of the host's Delaunay call (src/same.py:1023).
//   same_window_filter_finish  per window the Delaunay simplices in; triangle classes (src/helpers.py:300-330), the keep list and the
//                              same-type triangles added back so that every node keeps one (src/helpers.py:331-340, :365-389) -- all on
//                              the device, in the reference's order (a cosine within 8 ulp of the angle threshold is left to the host,
//                              which re-decides it with the reference's literal arccos and calls again with prefiltered = 1: the kept
//                              triangles in, no filter); source signs / weights (src/same.py:1128-1146), per-row minimum and the greedy
//                              MIP start (src/init_helpers.py:104-133), the lazy-constraint body under that incumbent
//                              (src/same.py:645-669), XY-order sweep (src/violationhelper.py:53-117), signed-area flips
//                              (src/same.py:1362-1402).  Back to the host: the matched reference row per kept aligned cell, the
//                              per-cell violation flags and eight counters.
//
// Every kernel takes up to SAME_LAUNCH_WINDOWS (8) windows per launch: blockIdx.y = window, the per-window arguments -- pointers into
// the window's OWN buffers, its counts -- travel by value in the kernarg segment (Batch<Args>), the grid is sized by the group's largest
// window and blocks beyond a window's share leave at once.  A call lays every window's buffers out (prepare_*), then per group of
// eight: one launch that zeroes the heads of their buffers (scan words, counters, marks), the kernels, one launch that writes what
// comes back straight into the windows' pinned host blocks; ONE wait per call.  Nothing a call computes is sized by a number the host
// has to wait for: lists are allocated for the candidates of the covered cells (known from the host's copy of the cell offsets), their
// true lengths stay in a counter block on the device and every kernel reads them there.  same_ctx_stat counts the runtime calls,
// tests/test_gpu_run_same.py holds the per-window totals.  Ordered compactions are single launches over many blocks (scan.h).
//
// Reference cells keep their SECTION rows through prune, costs and sweeps; the window's own numbering (position in the ascending
// list of reference rows in the box -- what the reference's frames would index before src/utils.py:740-742 drops the unreferenced)
// is only looked up for the pair list handed out and for the greedy rule's per-column state.  A window's numbers are those of
// the column pipeline bit for bit (tests/test_gpu_run_same.py::test_device_windows_*).
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

namespace {

using namespace devmath;
using scan::Pair;

constexpr int MAX_RUN_CELLS = 64;        // cells of a section's grid one window may cover on the cell-run path
constexpr int64_t MAX_GRID_CELLS = (int64_t)1 << 22;
constexpr unsigned OUTSIDE = 0x80000000u;   // flag on a candidate row that fails the box test

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
};

int ensure(same_ctx *ctx, DevBuf &b, size_t bytes) {
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
void release(DevBuf &b) {
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

__device__ __forceinline__ int cell_of(double v, double origin, double width, int n) {
    const double f = __builtin_floor((v - origin) / width);
    int c = f < 0.0 ? 0 : (f >= (double)n ? n - 1 : (int)f);
    while (c > 0 && v < cell_edge(origin, width, c)) --c;             // the quotient may round across an edge: the edges decide
    while (c < n - 1 && v >= cell_edge(origin, width, c + 1)) ++c;
    return c;
}

// sort key of a row: cell << 32 | row (rows ascending inside a cell); rows with a NaN / infinite coordinate sort behind every cell
__global__ __launch_bounds__(256) void bin_key_kernel(const double *__restrict__ xy, int64_t n, int64_t n_pad, BinGrid g,
                                                       unsigned long long *__restrict__ key) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pad) return;
    unsigned long long k = ~0ull;
    if (i < n) {
        const double2_t p = ld2(xy, i);
        if (p.x - p.x == 0.0 && p.y - p.y == 0.0) {
            const int cx = cell_of(p.x, g.x0, g.cw, g.nx), cy = cell_of(p.y, g.y0, g.ch, g.ny);
            k = ((unsigned long long)((unsigned)cy * (unsigned)g.nx + (unsigned)cx) << 32) | (unsigned long long)i;
        } else {
            k = 0xFFFFFFFF00000000ull | (unsigned long long)i;
        }
    }
    key[i] = k;
}
__global__ __launch_bounds__(256) void bin_order_kernel(const unsigned long long *__restrict__ key, int64_t n, int32_t *__restrict__ order) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n) order[q] = (int32_t)(uint32_t)key[q];
}
// starts[c] = first sorted position of a row of cell >= c, c = 0 .. cells (starts[cells] = rows with finite coordinates)
__global__ __launch_bounds__(256) void bin_starts_kernel(const unsigned long long *__restrict__ key, int64_t n, int64_t cells,
                                                          unsigned *__restrict__ starts) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c > cells) return;
    const unsigned long long want = (unsigned long long)c << 32;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (key[mid] < want) lo = mid + 1; else hi = mid;
    }
    starts[c] = (unsigned)lo;
}
// bounding box of the finite points: keys of {min x, min y} by atomicMin, {max x, max y} by atomicMax
__global__ __launch_bounds__(256) void bin_bbox_kernel(const double *__restrict__ xy, int64_t n, unsigned long long *__restrict__ bbox) {
    unsigned long long kx0 = ~0ull, ky0 = ~0ull, kx1 = 0ull, ky1 = 0ull;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double2_t p = ld2(xy, i);
        if (p.x - p.x == 0.0 && p.y - p.y == 0.0) {
            const unsigned long long kx = order_key(p.x), ky = order_key(p.y);
            kx0 = kx < kx0 ? kx : kx0; ky0 = ky < ky0 ? ky : ky0; kx1 = kx > kx1 ? kx : kx1; ky1 = ky > ky1 ? ky : ky1;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long a = __shfl_xor(kx0, off, 64), b = __shfl_xor(ky0, off, 64), c = __shfl_xor(kx1, off, 64),
                                 d = __shfl_xor(ky1, off, 64);
        kx0 = a < kx0 ? a : kx0; ky0 = b < ky0 ? b : ky0; kx1 = c > kx1 ? c : kx1; ky1 = d > ky1 ? d : ky1;
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&bbox[0], kx0); atomicMin(&bbox[1], ky0); atomicMax(&bbox[2], kx1); atomicMax(&bbox[3], ky1);
    }
}
double host_key_to_double(unsigned long long k) {
    const unsigned long long u = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    double d;
    memcpy(&d, &u, sizeof d);
    return d;
}

__global__ __launch_bounds__(256) void to_float_kernel(const double *__restrict__ src, int64_t n, float *__restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (float)src[i];          // round to nearest even, as numpy's astype(float32)
}

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

struct CopyArgs {
    const void *src[2];
    void *dst[2];
    size_t bytes[2];
};
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

// ---- triangle filter (src/helpers.py:233-395) on the device ------------------------------------------------------------------
// counters of the filter: [0] kept (class 0), [1] added back, [2] cosines within `tol` of the threshold, [3] triangles left
enum { FC_KEEP = 0, FC_ADD = 1, FC_NEAR = 2, FC_TR = 3 };

// The filter's kernels take the windows of a batch in one launch each (Batch<FilterArgs>, blockIdx.y = window).  The filter settings
// are the call's: the same for every window.
struct FilterArgs {
    const double *xy;                  // kept aligned cells' XY
    const int32_t *raw;                // the triangulation's simplices
    int64_t Tr, n;                     // triangles, kept aligned cells
    const int32_t *type_id;            // or null
    uint8_t *cls;
    double *perim;
    uint8_t *has_kept, *any_valid;
    unsigned long long *best_p;        // or null (no re-adding)
    unsigned *best_t, *first_v;
    unsigned long long *st_keep, *st_own, *counters;
    int32_t *klist, *nlist, *out;
};

// classes; vertices with a kept triangle, vertices with any valid (kept or same-type) triangle; knife-edge cosines
__global__ __launch_bounds__(256) void filter_classify_kernel(Batch<FilterArgs> b, double radius, int angle_enabled, double cos_thr, int near_enabled,
                                                               double tol) {
    const FilterArgs &w = b.w[blockIdx.y];
    if ((int64_t)blockIdx.x * blockDim.x >= w.Tr) return;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int32_t *__restrict__ tris = w.raw, *__restrict__ type_id = w.type_id;
    bool near = false;
    if (t < w.Tr) {
        const int32_t a = tris[3 * t], bb = tris[3 * t + 1], d = tris[3 * t + 2];
        const TriClass r = classify_triangle(ld2(w.xy, a), ld2(w.xy, bb), ld2(w.xy, d), radius, angle_enabled, cos_thr,
                                             type_id && type_id[a] == type_id[bb] && type_id[bb] == type_id[d]);
        w.cls[t] = r.cls;
        w.perim[t] = r.perim;
        near = near_enabled && r.cls != 1 && fabs(r.maxcos - cos_thr) <= tol;
        if (r.cls == 0 || r.cls == 3) {
            w.any_valid[a] = 1; w.any_valid[bb] = 1; w.any_valid[d] = 1;
            if (r.cls == 0) { w.has_kept[a] = 1; w.has_kept[bb] = 1; w.has_kept[d] = 1; }
        }
        // best same-type triangle of every vertex, first pass (src/helpers.py:334-340): the smallest perimeter, as an atomic maximum
        // over the INVERTED bit pattern (perimeters are >= 0: the order of the bits is theirs; zero = none yet, one fill prepares it)
        if (w.best_p && r.cls == 3) {
            const unsigned long long key = ~(unsigned long long)__double_as_longlong(r.perim);
            atomicMax(&w.best_p[a], key); atomicMax(&w.best_p[bb], key); atomicMax(&w.best_p[d], key);
        }
    }
    const unsigned long long nb = __ballot(near);
    if ((threadIdx.x & 63) == 0 && nb) atomicAdd(&w.counters[FC_NEAR], (unsigned long long)__builtin_popcountll(nb));
}
// the kept (class 0) triangles in order; and, second pass of the best same-type triangle: the first triangle in input order among
// a vertex's equal smallest perimeters (an atomic maximum of the inverted triangle index)
__global__ __launch_bounds__(scan::NT) void filter_keep_kernel(Batch<FilterArgs> b) {
    __shared__ scan::Shared sh;
    const FilterArgs &w = b.w[blockIdx.y];
    const int nb = (int)scan::blocks_for(w.Tr);
    if ((int)blockIdx.x >= nb || w.Tr == 0) return;
    const int64_t Tr = w.Tr;
    const uint8_t *__restrict__ cls = w.cls;
    auto val = [&](int64_t t) { return Pair{t < Tr && cls[t] == 0 ? 1u : 0u, 0u}; };
    Pair through;
    const Pair off = scan::exclusive(w.st_keep, (int)blockIdx.x, val, sh, &through);
    const int64_t t = (int64_t)blockIdx.x * scan::NT + threadIdx.x;
    if (t < Tr) {
        const uint8_t c = cls[t];
        if (c == 0) w.klist[off.a] = (int32_t)t;
        if (w.best_p && c == 3) {
            const unsigned long long key = ~(unsigned long long)__double_as_longlong(w.perim[t]);
            for (int q = 0; q < 3; ++q) {
                const int32_t v = w.raw[3 * t + q];
                if (w.best_p[v] == key) atomicMax(&w.best_t[v], ~(unsigned)t);
            }
        }
    }
    if ((int)blockIdx.x == nb - 1 && threadIdx.x == 0) w.counters[FC_KEEP] = through.a;
}
// nodes without a kept triangle but with a valid one are walked in ascending order and bring their best triangle along unless an
// earlier node already did (src/helpers.py:365-389): first_v[t] = the first node that asks for t (inverted, zero = nobody) ...
__device__ __forceinline__ bool asks(const uint8_t *has_kept, const uint8_t *any_valid, const unsigned *best_t, int64_t v) {
    return !has_kept[v] && any_valid[v] && best_t[v] != 0u;
}
__global__ __launch_bounds__(256) void filter_first_node_kernel(Batch<FilterArgs> b) {
    const FilterArgs &w = b.w[blockIdx.y];
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= w.n || !asks(w.has_kept, w.any_valid, w.best_t, v)) return;
    atomicMax(&w.first_v[~w.best_t[v]], ~(unsigned)v);
}
// ... and the nodes that are the first to ask, compacted in order
__global__ __launch_bounds__(scan::NT) void filter_owner_kernel(Batch<FilterArgs> b) {
    __shared__ scan::Shared sh;
    const FilterArgs &w = b.w[blockIdx.y];
    const int nb = (int)scan::blocks_for(w.n);
    if ((int)blockIdx.x >= nb || w.n == 0) return;
    const int64_t n = w.n;
    auto own = [&](int64_t v) { return v < n && asks(w.has_kept, w.any_valid, w.best_t, v) && w.first_v[~w.best_t[v]] == ~(unsigned)v; };
    auto val = [&](int64_t v) { return Pair{own(v) ? 1u : 0u, 0u}; };
    Pair through;
    const Pair off = scan::exclusive(w.st_own, (int)blockIdx.x, val, sh, &through);
    const int64_t v = (int64_t)blockIdx.x * scan::NT + threadIdx.x;
    if (own(v)) w.nlist[off.a] = (int32_t)v;
    if ((int)blockIdx.x == nb - 1 && threadIdx.x == 0) w.counters[FC_ADD] = through.a;
}
// the kept triangles in the reference's order: class-0 triangles ascending, then the added-back ones in walk order
__global__ __launch_bounds__(256) void filter_emit_kernel(Batch<FilterArgs> b) {
    const FilterArgs &w = b.w[blockIdx.y];
    if ((int64_t)blockIdx.x * blockDim.x >= w.Tr) return;
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n_keep = (int64_t)w.counters[FC_KEEP], n_add = (int64_t)w.counters[FC_ADD];
    if (q == 0) w.counters[FC_TR] = (unsigned long long)(n_keep + n_add);
    if (q >= n_keep + n_add) return;
    const int64_t t = q < n_keep ? w.klist[q] : (int64_t)~w.best_t[w.nlist[q - n_keep]];
    w.out[3 * q] = w.raw[3 * t];
    w.out[3 * q + 1] = w.raw[3 * t + 1];
    w.out[3 * q + 2] = w.raw[3 * t + 2];
}

// ---- incumbent and sweeps ---------------------------------------------------------------------------------------------------
// counters of the finish call: [0] orientation checked, [1] flipped, [2] XY comparisons, [3] XY violations, [4] triangles with
// one, [5] area flips, [6] (host) greedy rounds, [7] matched aligned cells, [8] pairs the greedy rule could still take
enum { SC_CHECKED = 0, SC_FLIPPED = 1, SC_CMP = 2, SC_VIOL = 3, SC_TVIOL = 4, SC_AFLIP = 5, SC_ROUNDS = 6, SC_MATCHED = 7, SC_REMAINING = 8, SC_COUNT = 16 };
constexpr int WINDOW_GREEDY_ROUNDS = 3;   // rounds enqueued before the first look (cfg 5: 1-3 productive rounds per window)

// The finish call's kernels take the windows of a batch in one launch each too (Batch<A>, blockIdx.y = window).
// per kept aligned row: minimum pair cost (src/init_helpers.py:118-122; its pairs are a contiguous run of the pair list), whether
// it beats the no-match penalty, the row's pairs enter the greedy rule or not, no match yet
struct PreferArgs {
    const int32_t *prow;
    const double *cost64, *size_c;
    const unsigned long long *dn;
    uint8_t *alive;
    int32_t *match_pair;
    int64_t cap;               // the host's count of kept rows (sizes the launch; *dn is the same number on the device)
};
__global__ __launch_bounds__(256) void row_prefer_kernel(Batch<PreferArgs> b, double penalty) {
    const PreferArgs &w = b.w[blockIdx.y];
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= w.cap || a >= (int64_t)*w.dn) return;
    const int32_t lo = w.prow[a], hi = w.prow[a + 1];
    double best = __builtin_inf();
    for (int32_t p = lo; p < hi; ++p) {
        const double c = w.cost64[p];
        if (c < best) best = c;
    }
    const uint8_t prefer = best < penalty * w.size_c[a];
    for (int32_t p = lo; p < hi; ++p) w.alive[p] = prefer;
    w.match_pair[a] = -1;
}
// pair per row -> matched reference cell: its number in the window (handed out), its section row (the sweeps and the caller); and
// whether the greedy rule is finished: a pair still alive whose end points are both free would be taken by a further round
struct MatchRowsArgs {
    const int32_t *match_pair, *pairs, *jsec, *prow;
    const uint8_t *alive, *used;
    int64_t n_rows;
    const unsigned long long *dn;
    int32_t *match_loc, *match_row;
    uint8_t *pflag;
    unsigned long long *counters;
};
__global__ __launch_bounds__(256) void match_rows_kernel(Batch<MatchRowsArgs> b) {
    const MatchRowsArgs &w = b.w[blockIdx.y];
    if ((int64_t)blockIdx.x * blockDim.x >= w.n_rows) return;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool m = false;
    int open = 0;
    if (i < (int64_t)*w.dn) {
        const int32_t p = w.match_pair[i];
        w.match_loc[i] = p >= 0 ? w.pairs[2 * (int64_t)p + 1] : -1;
        w.match_row[i] = p >= 0 ? w.jsec[p] : -1;
        w.pflag[i] = 0;
        m = p >= 0;
        if (!m && !w.used[i])
            for (int32_t q = w.prow[i]; q < w.prow[i + 1]; ++q) open += w.alive[q] && !w.used[w.n_rows + w.pairs[2 * (int64_t)q + 1]];
    }
    const unsigned long long bal = __ballot(m), ob = __ballot(open != 0);
    if ((threadIdx.x & 63) == 0) {
        if (bal) atomicAdd(&w.counters[SC_MATCHED], (unsigned long long)__builtin_popcountll(bal));
        if (ob) atomicAdd(&w.counters[SC_REMAINING], (unsigned long long)__builtin_popcountll(ob));
    }
}
// per-cell flag byte: bit 0 = the XY-order sweep flags the cell (src/violationhelper.py:100-104), bit 1 = the cell is a vertex of a
// triangle whose signed area flips (src/same.py:1464-1469).  Two kinds of writers share a byte, so they OR into its 32-bit word
// (the array is word-aligned and padded to whole words by its carver; flagged cells are the exception, not the rule).
__device__ __forceinline__ void cell_flag_or(uint8_t *flags, int32_t i, unsigned bit) {
    atomicOr(reinterpret_cast<unsigned *>(flags) + (i >> 2), bit << (8 * (i & 3)));
}
// one pass over the kept triangles: source sign and weight (src/same.py:1128-1146), the lazy-constraint body under the incumbent
// (:645-669), the XY-order sweep (src/violationhelper.py:53-117), the signed-area flip (src/same.py:1362-1402; helpers.py:73-77)
struct SweepArgs {
    const int32_t *tris;
    int64_t Tr;                        // the number of triangles, or (dTr != null) the bound the launch is sized by
    const unsigned long long *dTr;
    const double *axy, *size_c, *ref_xy;
    const int32_t *match_row;
    int8_t *sign;
    double *weight;
    uint8_t *pflag;
    unsigned long long *counters;
};
__global__ __launch_bounds__(256) void window_sweeps_kernel(Batch<SweepArgs> b) {
    const SweepArgs &w = b.w[blockIdx.y];
    if ((int64_t)blockIdx.x * blockDim.x >= w.Tr) return;
    const int32_t *__restrict__ tris = w.tris;
    const double *__restrict__ axy = w.axy, *__restrict__ size_c = w.size_c, *__restrict__ ref_xy = w.ref_xy;
    const int32_t *__restrict__ match_row = w.match_row;
    int8_t *__restrict__ sign = w.sign;
    double *__restrict__ weight = w.weight;
    uint8_t *__restrict__ pflag = w.pflag;
    unsigned long long *__restrict__ counters = w.counters;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t Tr = w.dTr ? (int64_t)*w.dTr : w.Tr;
    int checked = 0, flipped = 0, ncmp = 0, nviol = 0, tv = 0, aflip = 0;
    if (t < Tr) {
        const int32_t v[3] = {tris[3 * t], tris[3 * t + 1], tris[3 * t + 2]};
        const int32_t m[3] = {match_row[v[0]], match_row[v[1]], match_row[v[2]]};
        double2_t a[3], r[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            a[q] = ld2(axy, v[q]);
            r[q] = m[q] >= 0 ? ld2(ref_xy, m[q]) : double2_t{0.0, 0.0};
        }
        const int8_t ss = orient_sign(a[0], a[1], a[2]);
        sign[t] = ss;
        weight[t] = size_c[v[0]] + size_c[v[1]] + size_c[v[2]];            // src/same.py:1131-1133
        const bool all3 = m[0] >= 0 && m[1] >= 0 && m[2] >= 0;
        const uint8_t f = orient_flag(ss, all3, r[0], r[1], r[2]);
        checked = f != 0;
        flipped = f == 2;
        const int E[3][2] = {{0, 1}, {0, 2}, {1, 2}};
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            const int p = E[e][0], q = E[e][1];
            if (m[p] >= 0 && m[q] >= 0) {  // both matched (implies >= 2 matched vertices, violationhelper.py:58-60)
                ++ncmp;
                const uint8_t e2 = xyorder_edge(a[p], a[q], r[p], r[q]);
                nviol += ((e2 >> 1) & 1) + ((e2 >> 2) & 1);
                if (e2) { tv = 1; cell_flag_or(pflag, v[p], 1u); cell_flag_or(pflag, v[q], 1u); }
            }
        }
        if (all3) {
            const double bf = signed_area(a[0], a[1], a[2]), af = signed_area(r[0], r[1], r[2]);
            aflip = bf * af < 0.0;                                         // src/same.py:1401
            if (aflip)
                for (int q = 0; q < 3; ++q) cell_flag_or(pflag, v[q], 2u);
        }
    }
    int vals[6] = {checked, flipped, ncmp, nviol, tv, aflip};
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) vals[q] += __shfl_down(vals[q], off, 64);
    __shared__ int part[4][6];
    if ((threadIdx.x & 63) == 0)
        for (int q = 0; q < 6; ++q) part[threadIdx.x >> 6][q] = vals[q];
    __syncthreads();
    if (threadIdx.x < 6) {      // one atomic per block per counter (integer sums: order-independent)
        const int s = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
        if (s) atomicAdd(&counters[threadIdx.x], (unsigned long long)s);
    }
}

}  // namespace

struct same_section {
    same_ctx *ctx = nullptr;
    int64_t n = 0;
    int T = 0;
    int cost_f32 = 0;
    double *xy = nullptr;     // [n][2]
    void *xy_c = nullptr;     // [n][2] in the cost type (== xy for fp64 costs)
    void *types_c = nullptr;  // [n][T] in the cost type
    double *size = nullptr;   // [n]
    int32_t *type_id = nullptr;  // [n] codes of the cell type (equal type <=> equal code), or none
    // the grid of cells (same_section_bin)
    BinGrid grid;
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
    DevBuf stage, filter, finish, tris, big_mask, full_m, full_r;
    // stage block
    unsigned long long *counts = nullptr;           // [8], first words of the block the stage call copies back
    int32_t *rows_m = nullptr, *rows_r = nullptr, *idx = nullptr, *cnt = nullptr, *ua = nullptr, *rows_ua = nullptr, *type_c = nullptr,
            *prow = nullptr, *pairs = nullptr, *jsec = nullptr;
    double *axy_c = nullptr, *size_c = nullptr, *cost64 = nullptr;
    // finish block
    int8_t *sign = nullptr;
    double *weight = nullptr;
    int32_t *match_loc = nullptr;
    void *host = nullptr;     // pinned staging for everything that comes back
    char *host_dev = nullptr; // the same block as the device addresses it (null: not addressable -- copies go through the copy engine)
    size_t host_bytes = 0;
    size_t host_finish_off = 0;   // the pinned block: [stage call's copy | finish call's copy | the filter's counters]
    size_t host_filter_off = 0;
};

namespace {

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

// the prune index of `ref` for this radius (built once, with the caller's context); the caller keeps *out until its kernels are done
int knn_index_for(same_ctx *ctx, const same_section *ref, double radius, std::shared_ptr<same_knn_index> *out) {
    same_section *s = const_cast<same_section *>(ref);
    std::shared_ptr<same_knn_index> dropped;      // freed after the lock is let go (hipFree waits for the device)
    {
        std::lock_guard<std::mutex> hold(s->lock);
        for (size_t q = 0; q < s->knn.size(); ++q)
            if (s->knn[q].first == radius) {
                if (q) std::rotate(s->knn.begin(), s->knn.begin() + q, s->knn.begin() + q + 1);     // most recently used first
                *out = s->knn.front().second;
                return SAME_OK;
            }
        // one index per radius a section is pruned with: a handful in a run, a stream of them in a parameter search over one long-lived
        // section -- the least recently used one goes when the table is full (each holds a sorted copy of the section's XY and rows).
        // Calls of other threads that are pruning with it right now hold it too: it is freed when the last of them has waited.
        constexpr size_t MAX_KNN_INDICES = 16;
        if (s->knn.size() >= MAX_KNN_INDICES) {
            dropped = std::move(s->knn.back().second);
            s->knn.pop_back();
        }
        same_knn_index *ix = nullptr;
        SAME_TRY(same_knn_index_build(ctx, s->xy, s->n, radius, &ix));
        s->knn.insert(s->knn.begin(), std::make_pair(radius, std::shared_ptr<same_knn_index>(ix, same_knn_index_destroy)));
        *out = s->knn.front().second;
    }
    return SAME_OK;
}

// covered cells of `box` in the section's grid, and whether the box is exactly their union
struct Cover {
    int cx0 = 0, ncx = 0, cy0 = 0, ncy = 0;
    int64_t n_cand = 0;
    bool aligned = false, use_runs = false;
};
int host_cell(double v, double origin, double width, int n) {      // cell_of on the host (the same expressions)
    const double f = std::floor((v - origin) / width);
    int c = f < 0.0 ? 0 : (f >= (double)n ? n - 1 : (int)f);
    while (c > 0 && v < cell_edge(origin, width, c)) --c;
    while (c < n - 1 && v >= cell_edge(origin, width, c + 1)) ++c;
    return c;
}
Cover cover_of(const same_section *s, const double *box) {
    Cover c;
    c.use_runs = true;
    if (s->n_binned == 0) return c;                                                       // no row with finite coordinates: nothing is inside any box
    const BinGrid &g = s->grid;
    const double x0 = box[0], x1 = box[1], y0 = box[2], y1 = box[3];
    if (!(x0 < x1) || !(y0 < y1)) return c;                                               // empty (or NaN) box
    const double gx1 = cell_edge(g.x0, g.cw, g.nx), gy1 = cell_edge(g.y0, g.ch, g.ny);
    if (!(x1 > g.x0) || !(x0 < gx1) || !(y1 > g.y0) || !(y0 < gy1)) return c;             // beside the grid
    // first cell whose upper edge is above the box's lower edge; last cell whose lower edge is below the box's upper edge
    const int cx0 = x0 <= g.x0 ? 0 : host_cell(x0, g.x0, g.cw, g.nx), cy0 = y0 <= g.y0 ? 0 : host_cell(y0, g.y0, g.ch, g.ny);
    int cx1 = x1 >= gx1 ? g.nx - 1 : host_cell(x1, g.x0, g.cw, g.nx), cy1 = y1 >= gy1 ? g.ny - 1 : host_cell(y1, g.y0, g.ch, g.ny);
    if (cx1 > cx0 && cell_edge(g.x0, g.cw, cx1) >= x1) --cx1;      // x1 is exclusive: a cell that starts at x1 holds nothing of the box
    if (cy1 > cy0 && cell_edge(g.y0, g.ch, cy1) >= y1) --cy1;
    c.cx0 = cx0; c.ncx = cx1 - cx0 + 1; c.cy0 = cy0; c.ncy = cy1 - cy0 + 1;
    c.use_runs = (int64_t)c.ncx * c.ncy <= MAX_RUN_CELLS;
    // every row of the covered cells is inside the box iff the box reaches (at least) the cells' outer edges
    c.aligned = x0 <= cell_edge(g.x0, g.cw, cx0) && x1 >= cell_edge(g.x0, g.cw, cx1 + 1) && y0 <= cell_edge(g.y0, g.ch, cy0) &&
                y1 >= cell_edge(g.y0, g.ch, cy1 + 1);
    for (int cy = cy0; cy <= cy1; ++cy)
        c.n_cand += (int64_t)s->h_starts[(size_t)cy * g.nx + cx1 + 1] - (int64_t)s->h_starts[(size_t)cy * g.nx + cx0];
    return c;
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

int bin_section(same_ctx *ctx, same_section *s, double x0, double y0, double cw, double ch, bool default_grid) {
    // a binned section is replaced as a whole: the old index goes when the new one stands
    const int64_t n = s->n;
    BinGrid g;
    std::vector<unsigned> h_starts;
    int32_t *order = nullptr;
    unsigned *starts = nullptr;
    int64_t n_binned = 0;
    if (n > 0) {
        // bounding box of the rows with finite coordinates
        unsigned long long *dbbox = nullptr, hb[4] = {~0ull, ~0ull, 0ull, 0ull};
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&dbbox), sizeof hb));
        hipError_t e = hipMemcpyAsync(dbbox, hb, sizeof hb, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(bin_bbox_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(n, 256), 256)), dim3(256), 0, ctx->stream, s->xy, n, dbbox);
            e = hipMemcpyAsync(hb, dbbox, sizeof hb, hipMemcpyDeviceToHost, ctx->stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        (void)hipFree(dbbox);
        if (e != hipSuccess) return same_fail(ctx, SAME_EIO, "section bounding box", e);
        const bool any = hb[0] != ~0ull;
        if (any) {
            const double mx0 = host_key_to_double(hb[0]), my0 = host_key_to_double(hb[1]), mx1 = host_key_to_double(hb[2]),
                         my1 = host_key_to_double(hb[3]);
            if (default_grid) {      // about 512 rows a cell, at most 128 x 128 cells, anchored at the lower corner
                const double side = std::sqrt((double)n / 512.0);
                const int want = (int)std::min(128.0, std::max(1.0, std::floor(side)));
                x0 = mx0; y0 = my0;
                cw = (mx1 - mx0) / want; ch = (my1 - my0) / want;
                if (!(cw > 0.0) || !std::isfinite(cw)) cw = 1.0;
                if (!(ch > 0.0) || !std::isfinite(ch)) ch = 1.0;
            }
            // whole cells from the caller's origin down to the lowest row and up past the highest
            auto fit = [](double origin, double width, double lo, double hi, double *o_out, int *n_out) -> bool {
                double shift = origin > lo ? std::ceil((origin - lo) / width) : 0.0;
                if (!(shift < 1e9)) return false;
                double o = origin - shift * width;
                while (o > lo) { shift += 1.0; o = origin - shift * width; }
                double cells = std::floor((hi - o) / width) + 1.0;
                if (!(cells < 1e9)) return false;
                int nn = (int)std::max(1.0, cells);
                while (!(cell_edge(o, width, nn) > hi)) {
                    if (nn >= (1 << 30)) return false;
                    ++nn;
                }
                *o_out = o;
                *n_out = nn;
                return true;
            };
            REQUIRE(ctx, fit(x0, cw, mx0, mx1, &g.x0, &g.nx) && fit(y0, ch, my0, my1, &g.y0, &g.ny));
            g.cw = cw;
            g.ch = ch;
            REQUIRE(ctx, (int64_t)g.nx * g.ny <= MAX_GRID_CELLS);
        }
        const int64_t cells = (int64_t)g.nx * g.ny;
        int64_t n_pad = 2048;
        while (n_pad < n) n_pad <<= 1;
        unsigned long long *key = nullptr;
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&key), (size_t)n_pad * 8));
        int rc = SAME_OK;
        e = hipMalloc(reinterpret_cast<void **>(&order), (size_t)n * sizeof(int32_t));
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&starts), (size_t)(cells + 1) * sizeof(unsigned));
        if (e == hipSuccess) {
            hipLaunchKernelGGL(bin_key_kernel, dim3(grid_for(n_pad)), dim3(256), 0, ctx->stream, s->xy, n, n_pad, g, key);
            rc = same_sort_u64_core(ctx, key, n_pad);
            if (rc == SAME_OK) {
                hipLaunchKernelGGL(bin_order_kernel, dim3(grid_for(n)), dim3(256), 0, ctx->stream, key, n, order);
                hipLaunchKernelGGL(bin_starts_kernel, dim3(grid_for(cells + 1)), dim3(256), 0, ctx->stream, key, n, cells, starts);
                h_starts.resize((size_t)cells + 1);
                e = hipGetLastError();
                if (e == hipSuccess) e = hipMemcpyAsync(h_starts.data(), starts, (size_t)(cells + 1) * sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            }
        }
        (void)hipFree(key);
        if (e != hipSuccess || rc != SAME_OK) {
            if (order) (void)hipFree(order);
            if (starts) (void)hipFree(starts);
            return rc != SAME_OK ? rc : same_fail(ctx, e == hipErrorOutOfMemory ? SAME_ENOMEM : SAME_EIO, "section grid", e);
        }
        n_binned = (int64_t)h_starts[(size_t)cells];
    } else {
        h_starts.assign(2, 0u);
    }
    // the swap: no stage call is between cover_of() and its last enqueue (exclusive lock), and what was enqueued before has finished
    std::unique_lock<std::shared_mutex> swap_hold(s->grid_lock);
    if (s->order || s->starts) (void)hipDeviceSynchronize();
    if (s->order) (void)hipFree(s->order);
    if (s->starts) (void)hipFree(s->starts);
    s->grid = g;
    s->order = order;
    s->starts = starts;
    s->h_starts.swap(h_starts);
    s->n_binned = n_binned;
    return SAME_OK;
}

}  // namespace

extern "C" {

int same_section_create(same_ctx *ctx, const double *xy, const double *types, int T, const double *size, const int32_t *type_id,
                        int64_t n, int cost_f32, same_section **out) {
    REQUIRE(ctx, ctx && out);
    *out = nullptr;
    REQUIRE(ctx, n >= 0 && n < ((int64_t)1 << 31) - 256 && T >= 0 && T <= SAME_MAX_TYPES);
    REQUIRE(ctx, n == 0 || (xy && size && (T == 0 || types)));
    SAME_TRY(same_use(ctx));
    same_section *s = new (std::nothrow) same_section();
    if (!s) return SAME_ENOMEM;
    s->ctx = ctx; s->n = n; s->T = T; s->cost_f32 = cost_f32 ? 1 : 0;
    *out = s;                                     // freed by the caller's destroy on any failure below
    const size_t nn = (size_t)std::max<int64_t>(n, 1), tt = (size_t)std::max(T, 1);
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&s->xy), nn * 2 * sizeof(double)));
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&s->size), nn * sizeof(double)));
    if (type_id) HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&s->type_id), nn * sizeof(int32_t)));
    if (n) {
        HIP_TRY(ctx, hipMemcpyAsync(s->xy, xy, (size_t)n * 2 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(s->size, size, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        if (type_id) HIP_TRY(ctx, hipMemcpyAsync(s->type_id, type_id, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    }
    if (!s->cost_f32) {
        s->xy_c = s->xy;
        HIP_TRY(ctx, hipMalloc(&s->types_c, nn * tt * sizeof(double)));
        if (n && T) HIP_TRY(ctx, hipMemcpyAsync(s->types_c, types, (size_t)n * T * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    } else {                                      // the float copies are made here, once: (float) of every operand, as astype(float32)
        HIP_TRY(ctx, hipMalloc(&s->xy_c, nn * 2 * sizeof(float)));
        HIP_TRY(ctx, hipMalloc(&s->types_c, nn * tt * sizeof(float)));
        if (n) {
            hipLaunchKernelGGL(to_float_kernel, dim3(grid_for(n * 2)), dim3(256), 0, ctx->stream, s->xy, n * 2, static_cast<float *>(s->xy_c));
            if (T) {
                double *tmp = nullptr;
                HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&tmp), (size_t)n * T * sizeof(double)));
                hipError_t e = hipMemcpyAsync(tmp, types, (size_t)n * T * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
                if (e == hipSuccess) {
                    hipLaunchKernelGGL(to_float_kernel, dim3(grid_for(n * T)), dim3(256), 0, ctx->stream, tmp, n * T,
                                       static_cast<float *>(s->types_c));
                    e = hipStreamSynchronize(ctx->stream);
                }
                (void)hipFree(tmp);
                if (e != hipSuccess) return same_fail(ctx, SAME_EIO, "section upload", e);
            }
            HIP_TRY(ctx, hipGetLastError());
        }
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return bin_section(ctx, s, 0.0, 0.0, 0.0, 0.0, true);     // a grid of its own until the caller names the windows' (same_section_bin)
}

int same_section_bin(same_section *s, double x0, double y0, double cell_w, double cell_h) {
    if (!s) return SAME_EINVAL;
    same_ctx *ctx = s->ctx;
    REQUIRE(ctx, std::isfinite(x0) && std::isfinite(y0) && cell_w > 0.0 && cell_h > 0.0 && std::isfinite(cell_w) && std::isfinite(cell_h));
    SAME_TRY(same_use(ctx));
    return bin_section(ctx, s, x0, y0, cell_w, cell_h, false);
}

void same_section_destroy(same_section *s) {
    if (!s) return;
    (void)hipSetDevice(s->ctx->device);
    (void)hipDeviceSynchronize();                 // windows of other contexts may still be reading the section
    s->knn.clear();
    if (s->xy_c && s->xy_c != s->xy) (void)hipFree(s->xy_c);
    if (s->xy) (void)hipFree(s->xy);
    if (s->types_c) (void)hipFree(s->types_c);
    if (s->size) (void)hipFree(s->size);
    if (s->type_id) (void)hipFree(s->type_id);
    if (s->order) (void)hipFree(s->order);
    if (s->starts) (void)hipFree(s->starts);
    delete s;
}

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

namespace {

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

namespace {

// ---- the filter and the finish as enqueue-only halves + their read-backs, so that the two calls can also run as one ----------
struct FilterPlan {
    FilterArgs args{};
    void *zero = nullptr;                     // head of the filter buffer, zeroed with the group's
    size_t zero_bytes = 0;
    unsigned long long *counters = nullptr;   // [4] FC_*
    bool readd = false;
};

// A window's filter buffer laid out (d_simplices: its triangulation, on the device already); no launch, no fill -- those come per GROUP
// of windows (launch_filter, and the zeroing with the finish buffers')
int prepare_filter(same_window *w, const int32_t *d_simplices, int64_t Tr, int ignore_same_type, int ensure_min_triangle_per_node, FilterPlan *plan) {
    same_ctx *ctx = w->ctx;
    const int64_t n = w->n_ua;
    const bool use_type = ignore_same_type && w->has_type;
    plan->readd = use_type && ensure_min_triangle_per_node;
    Carver cv;   // zeroed head: scan words, counters, vertex marks, inverted minima
    const size_t st_keep = scan::status_bytes(Tr), st_own = scan::status_bytes(n);
    const size_t o_st_keep = cv.take(st_keep), o_st_own = cv.take(st_own), o_counters = cv.take(64), o_has_kept = cv.take((size_t)n),
                 o_any_valid = cv.take((size_t)n), o_best_p = cv.take((size_t)n * 8), o_best_t = cv.take((size_t)n * 4),
                 o_first_v = cv.take((size_t)Tr * 4);
    const size_t zero_bytes = cv.off;
    const size_t o_cls = cv.take((size_t)Tr), o_perim = cv.take((size_t)Tr * 8), o_klist = cv.take((size_t)Tr * 4), o_nlist = cv.take((size_t)n * 4);
    SAME_TRY(ensure(ctx, w->filter, cv.off));
    SAME_TRY(ensure(ctx, w->tris, (size_t)std::max<int64_t>(Tr, 1) * 12));
    char *base = static_cast<char *>(w->filter.p);
    auto at = [&](size_t off) { return base + off; };
    unsigned long long *dc = reinterpret_cast<unsigned long long *>(at(o_counters));
    plan->counters = dc;
    FilterArgs &a = plan->args;
    a.xy = w->axy_c;
    a.raw = d_simplices;
    a.Tr = Tr;
    a.n = n;
    a.type_id = use_type ? w->type_c : nullptr;
    a.cls = reinterpret_cast<uint8_t *>(at(o_cls));
    a.perim = reinterpret_cast<double *>(at(o_perim));
    a.has_kept = reinterpret_cast<uint8_t *>(at(o_has_kept));
    a.any_valid = reinterpret_cast<uint8_t *>(at(o_any_valid));
    a.best_p = plan->readd ? reinterpret_cast<unsigned long long *>(at(o_best_p)) : nullptr;
    a.best_t = reinterpret_cast<unsigned *>(at(o_best_t));
    a.first_v = reinterpret_cast<unsigned *>(at(o_first_v));
    a.st_keep = scan::arg(reinterpret_cast<unsigned long long *>(at(o_st_keep)));
    a.st_own = scan::arg(reinterpret_cast<unsigned long long *>(at(o_st_own)));
    a.counters = dc;
    a.klist = reinterpret_cast<int32_t *>(at(o_klist));
    a.nlist = reinterpret_cast<int32_t *>(at(o_nlist));
    a.out = static_cast<int32_t *>(w->tris.p);
    plan->zero = base;
    plan->zero_bytes = zero_bytes;
    return SAME_OK;
}

// the filter of a group of prepared windows (<= SAME_LAUNCH_WINDOWS; the settings are the call's, `readd` follows from them and from
// the moving section, so it is the group's): three launches, five when same-type triangles come back
int launch_filter(same_ctx *ctx, FilterPlan *const *plans, int n_w, double radius, int angle_enabled, double cos_thr, double near_tol) {
    Batch<FilterArgs> b{};
    int64_t max_tr = 0, max_n = 0;
    for (int q = 0; q < n_w; ++q) {
        b.w[q] = plans[q]->args;
        max_tr = std::max(max_tr, plans[q]->args.Tr);
        max_n = std::max(max_n, plans[q]->args.n);
    }
    const unsigned nw = (unsigned)n_w;
    const int near_enabled = angle_enabled && cos_thr == cos_thr && cos_thr - cos_thr == 0.0;       // a finite threshold
    SAME_LAUNCH(ctx, filter_classify_kernel, dim3(grid_for(max_tr), nw), dim3(256), 0, b, radius, angle_enabled, cos_thr, near_enabled, near_tol);
    SAME_LAUNCH(ctx, filter_keep_kernel, dim3(scan::blocks_for(max_tr), nw), dim3(scan::NT), 0, b);
    if (plans[0]->readd) {
        SAME_LAUNCH(ctx, filter_first_node_kernel, dim3(grid_for(max_n), nw), dim3(256), 0, b);
        SAME_LAUNCH(ctx, filter_owner_kernel, dim3(scan::blocks_for(max_n), nw), dim3(scan::NT), 0, b);
    }
    SAME_LAUNCH(ctx, filter_emit_kernel, dim3(grid_for(max_tr), nw), dim3(256), 0, b);
    HIP_TRY(ctx, hipGetLastError());
    return SAME_OK;
}

struct FinishPlan {
    unsigned long long *zero = nullptr;       // head of the finish buffer: [sel | counters | point flags (padded) | matched rows]
    void *filter_zero = nullptr;              // head of the window's filter buffer when this call filters it (zeroed in the same launch)
    size_t filter_zero_bytes = 0;
    size_t zero_bytes = 0, back_off = 0, back_bytes = 0, o_counters = 0, o_pflag = 0, o_match_row = 0;
    same_greedy_state gs;
    int32_t *match_pair = nullptr, *match_row = nullptr;
    uint8_t *pflag = nullptr;
    unsigned long long *counters = nullptr;
    int64_t cap_tr = 0;
    const unsigned long long *dTr = nullptr;
};

// match rows + the one pass over the triangles, for a group of windows (<= SAME_LAUNCH_WINDOWS) in one launch each
int enqueue_tail(same_ctx *ctx, same_window *const *ws, FinishPlan *const *ps, int n_w) {
    Batch<MatchRowsArgs> mb{};
    Batch<SweepArgs> sb{};
    int64_t max_n = 0, max_tr = 0;
    for (int q = 0; q < n_w; ++q) {
        same_window *w = ws[q];
        FinishPlan *p = ps[q];
        const int64_t n = w->n_ua;
        mb.w[q] = MatchRowsArgs{p->match_pair, w->pairs, w->jsec, w->prow, p->gs.alive, p->gs.used, n, w->counts + 2, w->match_loc, p->match_row, p->pflag,
                                p->counters};
        sb.w[q] = SweepArgs{static_cast<const int32_t *>(w->tris.p), p->cap_tr, p->dTr, w->axy_c, w->size_c, w->ref->xy, p->match_row, w->sign, w->weight,
                            p->pflag, p->counters};
        max_n = std::max(max_n, n);
        max_tr = std::max(max_tr, p->cap_tr);
    }
    if (max_n) SAME_LAUNCH(ctx, match_rows_kernel, dim3(grid_for(max_n), (unsigned)n_w), dim3(256), 0, mb);
    if (max_tr) SAME_LAUNCH(ctx, window_sweeps_kernel, dim3(grid_for(max_tr), (unsigned)n_w), dim3(256), 0, sb);
    HIP_TRY(ctx, hipGetLastError());
    return SAME_OK;
}

// A window's finish buffer laid out, its triangles uploaded (prefiltered form); no launch, no fill -- those come per GROUP of windows
// (launch_finish; p->zero / p->zero_bytes name the head to zero).  cap_tr: the number of triangles, or (dTr != null) the bound the launch is sized by with the number itself on the device
int prepare_finish(same_window *w, const int32_t *host_tris, int64_t cap_tr, const unsigned long long *dTr, FinishPlan *p) {
    same_ctx *ctx = w->ctx;
    const int64_t n = w->n_ua, P = w->P, n_ends = n + w->n_r;
    Carver cv;
    const size_t o_used = cv.take((size_t)n_ends), o_key = cv.take((size_t)n_ends * 16), o_idx = cv.take((size_t)n_ends * 8);
    const size_t o_sel = cv.take(SAME_GREEDY_BATCH_MAX * 8);
    const size_t o_counters = cv.off;
    cv.off += SC_COUNT * 8;
    const size_t o_pflag = cv.off;
    cv.off += ((size_t)n + 7) & ~size_t(7);
    const size_t zero_bytes = (cv.off + 15) & ~size_t(15);
    cv.off = zero_bytes;
    const size_t o_match_row = cv.off;
    cv.off += (size_t)n * 4;
    const size_t back_end = cv.off;
    cv.off = (cv.off + 255) & ~size_t(255);
    const size_t tt = (size_t)std::max<int64_t>(cap_tr, 1);
    const size_t o_alive = cv.take((size_t)std::max<int64_t>(P, 1)), o_match_pair = cv.take((size_t)n * 4), o_match_loc = cv.take((size_t)n * 4),
                 o_sign = cv.take(tt), o_weight = cv.take(tt * 8);
    SAME_TRY(ensure(ctx, w->finish, cv.off));
    SAME_TRY(ensure(ctx, w->tris, tt * 12));
    char *base = static_cast<char *>(w->finish.p);
    auto at = [&](size_t off) { return base + off; };
    p->zero = reinterpret_cast<unsigned long long *>(base);
    p->zero_bytes = zero_bytes;
    p->back_off = o_sel;
    p->back_bytes = back_end - o_sel;
    p->o_counters = o_counters - o_sel;
    p->o_pflag = o_pflag - o_sel;
    p->o_match_row = o_match_row - o_sel;
    p->gs.alive = reinterpret_cast<uint8_t *>(at(o_alive));
    p->gs.used = reinterpret_cast<uint8_t *>(at(o_used));
    p->gs.key[0] = reinterpret_cast<unsigned long long *>(at(o_key));
    p->gs.key[1] = p->gs.key[0] + n_ends;
    p->gs.idx[0] = reinterpret_cast<unsigned *>(at(o_idx));
    p->gs.idx[1] = p->gs.idx[0] + n_ends;
    p->gs.sel = reinterpret_cast<unsigned long long *>(at(o_sel));
    p->gs.float_costs = w->cost_f32 != 0;          // cost64 = (double)float there: the two-launch rounds
    p->counters = reinterpret_cast<unsigned long long *>(at(o_counters));
    p->pflag = reinterpret_cast<uint8_t *>(at(o_pflag));
    p->match_row = reinterpret_cast<int32_t *>(at(o_match_row));
    p->match_pair = reinterpret_cast<int32_t *>(at(o_match_pair));
    w->match_loc = reinterpret_cast<int32_t *>(at(o_match_loc));
    w->sign = reinterpret_cast<int8_t *>(at(o_sign));
    w->weight = reinterpret_cast<double *>(at(o_weight));
    p->cap_tr = cap_tr;
    p->dTr = dTr;
    REQUIRE(ctx, w->host_finish_off + p->back_bytes <= w->host_filter_off);   // sized by the stage call
    if (host_tris && cap_tr) SAME_COPY(ctx, w->tris.p, host_tris, (size_t)cap_tr * 12, hipMemcpyHostToDevice);
    return SAME_OK;
}

// greedy MIP start of a group of prepared windows -- per-row minimum, rows that beat their penalty, the scan's matching (one pair per
// aligned row) -- and the tail: one launch per kernel for the whole group (windows of one batch call share the cost type)
int launch_finish(same_ctx *ctx, same_window *const *ws, FinishPlan *const *ps, int n_w, double no_match_penalty) {
    Batch<PreferArgs> pb{};
    same_greedy_job jobs[SAME_LAUNCH_WINDOWS];
    int64_t max_n = 0;
    for (int q = 0; q < n_w; ++q) {
        same_window *w = ws[q];
        FinishPlan *p = ps[q];
        pb.w[q] = PreferArgs{w->prow, w->cost64, w->size_c, w->counts + 2, p->gs.alive, p->match_pair, w->n_ua};
        jobs[q].pairs = w->pairs;
        jobs[q].costs = w->cost64;
        jobs[q].P = w->P;
        jobs[q].n_m = w->n_ua;
        jobs[q].n_r = w->n_r;
        jobs[q].st = p->gs;
        jobs[q].match_pair = p->match_pair;
        max_n = std::max(max_n, w->n_ua);
    }
    if (max_n) SAME_LAUNCH(ctx, row_prefer_kernel, dim3(grid_for(max_n), (unsigned)n_w), dim3(256), 0, pb, no_match_penalty);
    SAME_TRY(same_greedy_rounds_batch_core(ctx, jobs, n_w, 0, WINDOW_GREEDY_ROUNDS));
    return enqueue_tail(ctx, ws, ps, n_w);
}

// the finish call's answers: one copy (enqueue_finish_copy), a wait the CALLER makes (one for a whole batch of windows), then
// read_finish: more greedy rounds (and the tail again) when the rounds enqueued up front did not settle the matching
int enqueue_finish_copy(same_window *w, FinishPlan *p) {
    same_ctx *ctx = w->ctx;
    SAME_COPY(ctx, static_cast<char *>(w->host) + w->host_finish_off, reinterpret_cast<const char *>(p->gs.sel), p->back_bytes, hipMemcpyDeviceToHost);
    return SAME_OK;
}

int read_finish(same_window *w, FinishPlan *p, int32_t *out_match_row, uint8_t *out_point_flag, int64_t *out_stats) {
    same_ctx *ctx = w->ctx;
    const int64_t n = w->n_ua, P = w->P;
    char *h = static_cast<char *>(w->host) + w->host_finish_off;
    const char *dsel = reinterpret_cast<const char *>(p->gs.sel);
    const unsigned long long *sel = reinterpret_cast<const unsigned long long *>(h);
    const unsigned long long *cnt = reinterpret_cast<const unsigned long long *>(h + p->o_counters);
    int rounds = 0;
    if (P) {
        int q = 0;
        while (q < WINDOW_GREEDY_ROUNDS && sel[q] != 0) ++q;
        rounds = q;
        // every enqueued round took something AND a pair could still be taken (counted by match_rows_kernel): a long chain of pre-empting
        // pairs -- keep going in growing batches (one read per batch), then redo the tail
        if (q == WINDOW_GREEDY_ROUNDS && cnt[SC_REMAINING] != 0) {
            int batch = 4;
            for (;;) {
                REQUIRE(ctx, rounds <= P + 1);
                SAME_FILL(ctx, p->gs.sel, 0, (size_t)batch * 8);
                SAME_TRY(same_greedy_rounds_core(ctx, w->pairs, w->cost64, P, nullptr, n, w->n_r, p->gs, p->match_pair, rounds, batch));
                SAME_COPY(ctx, h, dsel, (size_t)batch * 8, hipMemcpyDeviceToHost);
                SAME_WAIT(ctx);
                ++ctx->stats[SAME_STAT_GREEDY_READBACKS];
                q = 0;
                while (q < batch && sel[q] != 0) ++q;
                rounds += q;
                if (q < batch) break;
                if (batch < SAME_GREEDY_BATCH_MAX) batch *= 2;
            }
            SAME_FILL(ctx, p->counters, 0, SC_COUNT * 8);
            SAME_TRY(enqueue_tail(ctx, &w, &p, 1));
            SAME_COPY(ctx, h, dsel, p->back_bytes, hipMemcpyDeviceToHost);
            SAME_WAIT(ctx);
        }
    }
    for (int q = 0; q < 8; ++q) out_stats[q] = (int64_t)cnt[q];
    out_stats[SC_ROUNDS] = rounds;
    memcpy(out_match_row, h + p->o_match_row, (size_t)n * sizeof(int32_t));
    memcpy(out_point_flag, h + p->o_pflag, (size_t)n);
    return SAME_OK;
}

}  // namespace

extern "C" {

int same_window_filter_finish(same_window *const *windows, int n_windows, const int32_t *simplices, const int64_t *simplex_offsets, int prefiltered,
                              double radius, int angle_enabled, double cos_thr, double near_tol, int ignore_same_type,
                              int ensure_min_triangle_per_node, double no_match_penalty, int32_t *out_match_row, uint8_t *out_point_flag,
                              int64_t *out_stats, int64_t *out_counts) {
    same_ctx *ctx = nullptr;
    SAME_TRY(check_batch(windows, n_windows, &ctx));
    REQUIRE(ctx, simplex_offsets && out_counts && out_stats && simplex_offsets[0] == 0);
    int64_t n_cells = 0;
    for (int i = 0; i < n_windows; ++i) {
        const int64_t Tr = simplex_offsets[i + 1] - simplex_offsets[i];
        REQUIRE(ctx, windows[i]->staged == 2 && Tr >= 0 && Tr < ((int64_t)1 << 31) - 512);
        n_cells += windows[i]->n_ua;
    }
    REQUIRE(ctx, (simplex_offsets[n_windows] == 0 || simplices) && (n_cells == 0 || (out_match_row && out_point_flag)));
    for (int q = 0; q < 3 * n_windows; ++q) out_counts[q] = 0;
    for (int q = 0; q < 8 * n_windows; ++q) out_stats[q] = 0;
    SAME_TRY(same_use(ctx));
    for (int i = 0; i < n_windows; ++i)
        SAME_TRY(check_index_range(ctx, simplices + 3 * simplex_offsets[i], (simplex_offsets[i + 1] - simplex_offsets[i]) * 3, 0, windows[i]->n_ua,
                                   "triangles"));
    struct Item {
        FilterPlan fplan;
        FinishPlan plan;
        bool filtered = false, enqueued = false;
    };
    std::vector<Item> items((size_t)n_windows);
    // ONE wait for the batch.  The call's simplices go up in ONE copy (a scratch slot of the context: they are read by this call's filter
    // only); per window its filter and finish buffers are laid out; then per GROUP of SAME_LAUNCH_WINDOWS windows the zeroing of the
    // buffers' heads, the filter's and the finish's kernels (one launch each for the whole group); then every window's copies back
    const int32_t *d_simplices = nullptr;
    if (!prefiltered && simplex_offsets[n_windows] > 0) {
        int32_t *d = nullptr;
        SAME_TRY(slot_as(ctx, SL_TRIS, (size_t)simplex_offsets[n_windows] * 3, &d));
        SAME_COPY(ctx, d, simplices, (size_t)simplex_offsets[n_windows] * 12, hipMemcpyHostToDevice);
        d_simplices = d;
    }
    int rc = SAME_OK;
    std::vector<same_window *> live;
    std::vector<FinishPlan *> plans;
    std::vector<FilterPlan *> fplans;
    for (int i = 0; i < n_windows && rc == SAME_OK; ++i) {
        same_window *w = windows[i];
        Item &it = items[(size_t)i];
        const int32_t *tri = simplices + 3 * simplex_offsets[i];
        const int64_t Tr = simplex_offsets[i + 1] - simplex_offsets[i];
        w->filtered = w->finished = 0;
        w->Tr = 0;
        if (w->n_ua == 0) continue;
        if (Tr && !prefiltered) {
            rc = prepare_filter(w, d_simplices + 3 * simplex_offsets[i], Tr, ignore_same_type, ensure_min_triangle_per_node, &it.fplan);
            if (rc == SAME_OK) rc = prepare_finish(w, nullptr, Tr, it.fplan.counters + FC_TR, &it.plan);
            if (rc == SAME_OK) {
                fplans.push_back(&it.fplan);
                it.plan.filter_zero = it.fplan.zero;
                it.plan.filter_zero_bytes = it.fplan.zero_bytes;
            }
            it.filtered = true;
        } else {
            rc = prepare_finish(w, Tr ? tri : nullptr, Tr, nullptr, &it.plan);     // the caller's kept triangles (or none)
        }
        if (rc == SAME_OK) {
            live.push_back(w);
            plans.push_back(&it.plan);
        }
    }
    for (size_t g = 0; g < live.size() && rc == SAME_OK; g += SAME_LAUNCH_WINDOWS) {      // the heads of both buffers of every window: one launch per group
        ZeroArgs zr[SAME_LAUNCH_WINDOWS];
        const int n_g = (int)std::min<size_t>(SAME_LAUNCH_WINDOWS, live.size() - g);
        for (int q = 0; q < n_g; ++q) {
            const FinishPlan *fp = plans[g + (size_t)q];
            zr[q] = ZeroArgs{{fp->zero, fp->filter_zero}, {fp->zero_bytes, fp->filter_zero_bytes}};
        }
        rc = launch_zero(ctx, zr, n_g);
    }
    // groups of at most SAME_LAUNCH_WINDOWS consecutive windows that agree on what a launch fixes for all of them (windows of one call
    // usually come from one pair of sections: whether same-type triangles come back, the cost type)
    for (size_t g = 0, e; g < fplans.size() && rc == SAME_OK; g = e) {
        for (e = g + 1; e < fplans.size() && e - g < SAME_LAUNCH_WINDOWS && fplans[e]->readd == fplans[g]->readd; ++e) {}
        rc = launch_filter(ctx, fplans.data() + g, (int)(e - g), radius, angle_enabled, cos_thr, near_tol);
    }
    for (size_t g = 0, e; g < live.size() && rc == SAME_OK; g = e) {
        for (e = g + 1; e < live.size() && e - g < SAME_LAUNCH_WINDOWS && live[e]->cost_f32 == live[g]->cost_f32; ++e) {}
        rc = launch_finish(ctx, live.data() + g, plans.data() + g, (int)(e - g), no_match_penalty);
    }
    // the finish block of every window and, beside it, the filter's counters: one launch per group straight into the pinned blocks (or
    // one copy from each buffer where a block is not device-addressable)
    {
        CopyArgs ca[SAME_LAUNCH_WINDOWS];
        int n_g = 0;
        for (int i = 0; i < n_windows && rc == SAME_OK; ++i) {
            same_window *w = windows[i];
            Item &it = items[(size_t)i];
            if (w->n_ua != 0) {
                if (w->host_dev) {
                    ca[n_g++] = CopyArgs{{it.plan.gs.sel, it.filtered ? it.fplan.counters : nullptr},
                                         {w->host_dev + w->host_finish_off, w->host_dev + w->host_filter_off},
                                         {it.plan.back_bytes, it.filtered ? 4 * sizeof(unsigned long long) : 0}};
                } else {
                    if (it.filtered) {
                        unsigned long long *hf = reinterpret_cast<unsigned long long *>(static_cast<char *>(w->host) + w->host_filter_off);
                        hipError_t e = hipMemcpyAsync(hf, it.fplan.counters, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream);
                        ++ctx->stats[SAME_STAT_COPIES];
                        if (e != hipSuccess) rc = same_fail(ctx, SAME_EIO, "filter counters", e);
                    }
                    if (rc == SAME_OK) rc = enqueue_finish_copy(w, &it.plan);
                }
                it.enqueued = rc == SAME_OK;
            }
            if (rc == SAME_OK && n_g && (n_g == SAME_LAUNCH_WINDOWS || i == n_windows - 1)) {
                rc = launch_copy_back(ctx, ca, n_g);
                n_g = 0;
            }
        }
    }
    if (rc != SAME_OK) {
        (void)hipStreamSynchronize(ctx->stream);
        return rc;
    }
    SAME_WAIT(ctx);
    int64_t cell0 = 0;
    for (int i = 0; i < n_windows; ++i) {
        same_window *w = windows[i];
        Item &it = items[(size_t)i];
        const int64_t Tr = simplex_offsets[i + 1] - simplex_offsets[i];
        int64_t *counts = out_counts + 3 * i, *stats = out_stats + 8 * i;
        if (!it.enqueued) {                 // no kept aligned cell: nothing to match, nothing to sweep
            w->filtered = w->finished = 1;
            continue;
        }
        SAME_TRY(read_finish(w, &it.plan, out_match_row + cell0, out_point_flag + cell0, stats));
        cell0 += w->n_ua;
        if (it.filtered) {
            const unsigned long long *hf = reinterpret_cast<const unsigned long long *>(static_cast<const char *>(w->host) + w->host_filter_off);
            const int64_t n_keep = (int64_t)hf[FC_KEEP], n_near = (int64_t)hf[FC_NEAR], n_add = it.fplan.readd ? (int64_t)hf[FC_ADD] : 0;
            counts[0] = n_keep;
            counts[1] = n_add;
            counts[2] = n_near;
            if (n_near) {             // the caller filters this window on the host and calls again with prefiltered = 1: nothing here counts
                for (int q = 0; q < 8; ++q) stats[q] = 0;
                continue;
            }
            w->Tr = n_keep + n_add;
        } else {
            counts[0] = Tr;
            w->Tr = Tr;
        }
        w->filtered = w->finished = 1;
    }
    return SAME_OK;
}

}  // extern "C"
