// section.hip -- a tissue section resident on the device (same_section): its columns, the grid of cells its rows are binned into,
// the prune indices it answers with as the reference side.  SURVEY a13's "one-pass bin": src/same.py:293-295 masks the whole frame
// once per window; here the rows are sorted by cell once and a window reads the cells its box covers.
// Locks: window_internal.h ("who may touch what").
#include "window_internal.h"

namespace {

using namespace devmath;
using namespace win;

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

namespace win {

// The prune index of `ref` for this radius; the caller keeps *out until its kernels are done.  The table's lock is held for look-ups
// and insertions only, never across the build (device work and waits): a radius that is not in the table is built by the calling
// thread on its own context with the lock released -- other threads keep staging windows against the radii that are there -- and
// inserted afterwards; of two threads that build the same radius at once the second to arrive drops its build and takes the first's.
int knn_index_for(same_ctx *ctx, const same_section *ref, double radius, std::shared_ptr<same_knn_index> *out) {
    same_section *s = const_cast<same_section *>(ref);
    auto find = [&]() -> bool {                   // under the lock: the entry for `radius`, moved to the front (most recently used first)
        for (size_t q = 0; q < s->knn.size(); ++q)
            if (s->knn[q].first == radius) {
                if (q) std::rotate(s->knn.begin(), s->knn.begin() + q, s->knn.begin() + q + 1);
                *out = s->knn.front().second;
                return true;
            }
        return false;
    };
    {
        std::lock_guard<std::mutex> hold(s->lock);
        if (find()) return SAME_OK;
    }
    same_knn_index *ix = nullptr;
    SAME_TRY(same_knn_index_build(ctx, s->xy, s->n, radius, &ix));          // no lock: s->xy and s->n never change
    std::shared_ptr<same_knn_index> built(ix, same_knn_index_destroy), dropped;   // both freed after the lock is let go (hipFree waits for the device)
    {
        std::lock_guard<std::mutex> hold(s->lock);
        if (find()) return SAME_OK;               // another thread's build of this radius got there first: `built` goes
        // one index per radius a section is pruned with: a handful in a run, a stream of them in a parameter search over one long-lived
        // section -- the least recently used one goes when the table is full (each holds a sorted copy of the section's XY and rows).
        // Calls of other threads that are pruning with it right now hold it too: it is freed when the last of them has waited.
        constexpr size_t MAX_KNN_INDICES = 16;
        if (s->knn.size() >= MAX_KNN_INDICES) {
            dropped = std::move(s->knn.back().second);
            s->knn.pop_back();
        }
        s->knn.insert(s->knn.begin(), std::make_pair(radius, built));
        *out = std::move(built);
    }
    return SAME_OK;
}

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

}  // namespace win

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
        s->types64 = static_cast<double *>(s->types_c);
        if (n && T) HIP_TRY(ctx, hipMemcpyAsync(s->types_c, types, (size_t)n * T * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    } else {                                      // the float copies are made here, once: (float) of every operand, as astype(float32)
        HIP_TRY(ctx, hipMalloc(&s->xy_c, nn * 2 * sizeof(float)));
        HIP_TRY(ctx, hipMalloc(&s->types_c, nn * tt * sizeof(float)));
        if (n) {
            hipLaunchKernelGGL(to_float_kernel, dim3(grid_for(n * 2)), dim3(256), 0, ctx->stream, s->xy, n * 2, static_cast<float *>(s->xy_c));
            if (T) {      // the doubles stay (64 MB for a million cells x 8 types): the merged table's type columns are gathered from them on the device
                HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&s->types64), (size_t)n * T * sizeof(double)));
                HIP_TRY(ctx, hipMemcpyAsync(s->types64, types, (size_t)n * T * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
                hipLaunchKernelGGL(to_float_kernel, dim3(grid_for(n * T)), dim3(256), 0, ctx->stream, s->types64, n * T, static_cast<float *>(s->types_c));
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
    if (s->types64 && s->types64 != s->types_c) (void)hipFree(s->types64);
    if (s->types_c) (void)hipFree(s->types_c);
    if (s->size) (void)hipFree(s->size);
    if (s->type_id) (void)hipFree(s->type_id);
    if (s->id_codes) (void)hipFree(s->id_codes);
    if (s->order) (void)hipFree(s->order);
    if (s->starts) (void)hipFree(s->starts);
    delete s;
}

}  // extern "C"
