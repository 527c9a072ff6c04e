// knn.hip -- radius-limited top-k prune of reference cells per aligned cell.
//
// Reference: utils.find_knn_within_radius, src/utils.py:720-728: cKDTree.query_ball_point
// (dx*dx+dy*dy <= r*r, inclusive) then rank by np.linalg.norm and keep the first k.  The
// build's tie rule is (squared distance, ref index) ascending -- an order the reference's
// unstable argsort may itself produce, made deterministic.
//
// Kernel shape (wave64): one wave owns RW aligned rows, whose XY live in SGPRs; the 64
// lanes sweep the reference cells 64 at a time with one coalesced 16 B load per lane.  Per
// row the in-radius predicate is balloted; hit lanes append (d2, j) to the row's candidate
// list in LDS at base + popcount(lower lanes).  When a list would overflow it is pruned in
// place to its best k by rank counting; the same rank counting writes the final, sorted
// rows.  No atomics, no inter-wave communication, results independent of sweep order.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "common.h"
#include "scan.h"

namespace {

constexpr int KNN_WAVES = 4;   // waves per block
// Candidate slots per row (CAP) and aligned rows per wave (RW) come in two sizes: CAP needs k + 64 <= CAP.
//   k <= 64  : CAP 128, 8 rows per wave  (the reference default is k = 8)
//   k <= 448 : CAP 512, 2 rows per wave  (same algorithm; LDS per block 48 KB)
constexpr int KNN_CAP_SMALL = 128, KNN_RW_SMALL = 8, KNN_CAP_LARGE = 512, KNN_RW_LARGE = 2;

template <int CAP>
struct RowList {
    double d2[CAP];
    int32_t j[CAP];
};

// rank of candidate (d, j) among the m candidates of `L` under (d2, j) ascending
template <int CAP>
__device__ __forceinline__ int rank_of(const RowList<CAP> &L, int m, double d, int32_t j) {
    int rank = 0;
    for (int q = 0; q < m; ++q) {
        const double dq = L.d2[q];   // uniform address: LDS broadcast
        const int32_t jq = L.j[q];
        rank += (dq < d) || (dq == d && jq < j);
    }
    return rank;
}

// keep the best min(m, k) candidates, sorted, in slots [0, min(m,k))
template <int CAP>
__device__ __forceinline__ int prune_in_place(RowList<CAP> &L, int m, int k, int lane) {
    double d[CAP / 64];
    int32_t j[CAP / 64];
    int rk[CAP / 64];
#pragma unroll
    for (int p = 0; p < CAP / 64; ++p) {
        const int c = lane + 64 * p;
        rk[p] = CAP;
        if (c < m) {
            d[p] = L.d2[c];
            j[p] = L.j[c];
            rk[p] = rank_of(L, m, d[p], j[p]);
        }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): every read above has landed before the overwrite
#pragma unroll
    for (int p = 0; p < CAP / 64; ++p)
        if (rk[p] < k) { L.d2[rk[p]] = d[p]; L.j[rk[p]] = j[p]; }
    return m < k ? m : k;
}

// Window form (WIN, csrc/window.hip): the aligned rows and the reference cells are ROW LISTS into their sections' arrays
// (win.arows / win.rrows, ascending section rows), their lengths live on the device (win.n_a / win.n_r: the launch is sized by
// an upper bound), and the candidate written is the reference cell's SECTION row -- ranking by (d2, section row) is the ranking
// by (d2, index in the window), the lists being ascending.
struct WinRows {
    const int32_t *arows, *rrows;
    const unsigned long long *n_a, *n_r;
    double bx0, bx1, by0, by1;   // the window's box (grid form: a reference cell outside it is not a candidate)
};
// the windows of one launch (blockIdx.y = window; by value in the kernarg segment): their row lists and where their lists go
struct WinBatch {
    WinRows w[SAME_LAUNCH_WINDOWS];
    int32_t *out_idx[SAME_LAUNCH_WINDOWS], *out_cnt[SAME_LAUNCH_WINDOWS];
};

template <int KNN_CAP, int KNN_RW, bool WIN = false>
__global__ __launch_bounds__(64 * KNN_WAVES) void knn_prune_kernel(
    const double *__restrict__ axy, const double *__restrict__ rxy, int64_t n_r, int64_t row_begin,
    int64_t row_end, double r2, int k, int32_t *__restrict__ out_idx, double *__restrict__ out_d2,
    int32_t *__restrict__ out_cnt, WinBatch wb) {
    __shared__ RowList<KNN_CAP> lists[KNN_WAVES * KNN_RW];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const WinRows &win = wb.w[WIN ? blockIdx.y : 0];
    if constexpr (WIN) {
        row_end = (int64_t)*win.n_a;
        n_r = (int64_t)*win.n_r;
        out_idx = wb.out_idx[blockIdx.y];
        out_cnt = wb.out_cnt[blockIdx.y];
    }
    const int64_t row0 = row_begin + ((int64_t)blockIdx.x * KNN_WAVES + wave) * KNN_RW;
    if (row0 >= row_end) return;
    RowList<KNN_CAP> *L = lists + wave * KNN_RW;

    double ax[KNN_RW], ay[KNN_RW];
    int cnt[KNN_RW];
#pragma unroll
    for (int r = 0; r < KNN_RW; ++r) {
        int64_t i = (row0 + r < row_end) ? row0 + r : row_end - 1;  // tail rows repeat the last one
        if constexpr (WIN) i = win.arows[i];
        ax[r] = axy[2 * i];
        ay[r] = axy[2 * i + 1];
        cnt[r] = 0;
    }

    typedef double double2_t __attribute__((ext_vector_type(2)));
    for (int64_t base = 0; base < n_r; base += 64) {
        int64_t j = base + lane;
        const bool valid = j < n_r;
        if constexpr (WIN) j = win.rrows[valid ? j : n_r - 1];
        const double2_t p = *reinterpret_cast<const double2_t *>(rxy + 2 * (valid ? j : n_r - 1));
#pragma unroll
        for (int r = 0; r < KNN_RW; ++r) {
            const double dx = p.x - ax[r], dy = p.y - ay[r];
            const double d2 = dx * dx + dy * dy;
            const bool in = valid && d2 <= r2;
            const unsigned long long mask = __ballot(in);
            if (mask) {
                const int n = __builtin_popcountll(mask);
                if (cnt[r] + n > KNN_CAP) cnt[r] = prune_in_place(L[r], cnt[r], k, lane);
                const int pos = cnt[r] + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                if (in) { L[r].d2[pos] = d2; L[r].j[pos] = (int32_t)j; }
                cnt[r] += n;
            }
        }
    }

#pragma unroll
    for (int r = 0; r < KNN_RW; ++r) {
        const int64_t i = row0 + r;
        if (i >= row_end) break;
        const int m = cnt[r];
        const int keep = m < k ? m : k;
        const int64_t o = (i - row_begin) * k;
#pragma unroll
        for (int p = 0; p < KNN_CAP / 64; ++p) {
            const int c = lane + 64 * p;
            if (c < m) {
                const double d = L[r].d2[c];
                const int32_t jj = L[r].j[c];
                const int rk = rank_of(L[r], m, d, jj);
                if (rk < k) {
                    out_idx[o + rk] = jj;
                    if (out_d2) out_d2[o + rk] = d;
                }
            }
        }
        for (int q = keep + lane; q < k; q += 64) {
            out_idx[o + q] = -1;
            if (out_d2) out_d2[o + q] = __builtin_inf();
        }
        if (lane == 0) out_cnt[i - row_begin] = keep;
    }
}

// ---------------------------------------------------------------------------------------------
// Grid-accelerated prune.  The reference cells are counting-sorted into a uniform grid whose cell
// edge is a hair above the radius, so every in-radius neighbour of an aligned cell lies in the
// 3 x 3 cells around it; in row-major cell order those are three contiguous runs of the sorted
// array.  One wave per aligned row sweeps the three runs as one virtual index space (about one
// 64-lane step at the reference's typical density), with the same exact fp64 inclusion test and
// the same (d2, original ref index) ranking as the brute-force kernel -- the grid only prunes
// candidates that cannot pass, so outputs are bit-identical.
struct GridDesc {
    double x0, y0, inv_cell;
    int gx, gy;
};

__device__ __forceinline__ unsigned long long f64_key(double v) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

// bbox[0..3] = keys of {min x, min y} (atomicMin) and {max x, max y} (atomicMax)
__global__ __launch_bounds__(256) void bbox_kernel(const double *__restrict__ xy, int64_t n, unsigned long long *__restrict__ bbox) {
    typedef double double2_t __attribute__((ext_vector_type(2)));
    unsigned long long kx0 = ~0ull, ky0 = ~0ull, kx1 = 0ull, ky1 = 0ull;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {  // grid-stride
        const double2_t p = *reinterpret_cast<const double2_t *>(xy + 2 * i);
        const unsigned long long kx = f64_key(p.x), ky = f64_key(p.y);
        kx0 = kx < kx0 ? kx : kx0; ky0 = ky < ky0 ? ky : ky0; kx1 = kx > kx1 ? kx : kx1; ky1 = ky > ky1 ? ky : ky1;
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

__device__ __forceinline__ int cell_coord(double v, double v0, double inv_cell, int g) {
    const double c = __builtin_floor((v - v0) * inv_cell);
    return c < 0.0 ? 0 : (c >= (double)g ? g - 1 : (int)c);
}

__global__ __launch_bounds__(256) void grid_count_kernel(const double *__restrict__ rxy, int64_t n_r, GridDesc g,
                                                          unsigned *__restrict__ hist, unsigned *__restrict__ rank) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_r) return;
    const int cx = cell_coord(rxy[2 * j], g.x0, g.inv_cell, g.gx), cy = cell_coord(rxy[2 * j + 1], g.y0, g.inv_cell, g.gy);
    rank[j] = atomicAdd(&hist[(int64_t)cy * g.gx + cx], 1u);
}

// start[0..n] = exclusive scan of the cell counts (start[n] = their total): one launch over many blocks (scan.h)
__global__ __launch_bounds__(scan::NT) void grid_scan_kernel(const unsigned *__restrict__ count, int64_t n, unsigned long long *__restrict__ status,
                                                              unsigned *__restrict__ start) {
    __shared__ scan::Shared sh;
    auto val = [&](int64_t q) { return scan::Pair{q < n ? count[q] : 0u, 0u}; };
    scan::Pair through;
    const scan::Pair off = scan::exclusive(status, (int)blockIdx.x, val, sh, &through);
    const int64_t q = (int64_t)blockIdx.x * scan::NT + threadIdx.x;
    if (q < n) start[q] = off.a;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) start[n] = through.a;
}

__global__ __launch_bounds__(256) void grid_scatter_kernel(const double *__restrict__ rxy, int64_t n_r, GridDesc g,
                                                            const unsigned *__restrict__ start, const unsigned *__restrict__ rank,
                                                            double *__restrict__ sxy, int32_t *__restrict__ sidx) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_r) return;
    const double x = rxy[2 * j], y = rxy[2 * j + 1];
    const int cx = cell_coord(x, g.x0, g.inv_cell, g.gx), cy = cell_coord(y, g.y0, g.inv_cell, g.gy);
    const unsigned pos = start[(int64_t)cy * g.gx + cx] + rank[j];
    sxy[2 * (int64_t)pos] = x;
    sxy[2 * (int64_t)pos + 1] = y;
    sidx[pos] = (int32_t)j;
}

template <int KNN_CAP, bool WIN = false>
__global__ __launch_bounds__(64 * KNN_WAVES) void knn_grid_kernel(
    const double *__restrict__ axy, const double *__restrict__ sxy, const int32_t *__restrict__ sidx,
    const unsigned *__restrict__ start, GridDesc g, int64_t row_begin, int64_t row_end, double r2, int k,
    int32_t *__restrict__ out_idx, double *__restrict__ out_d2, int32_t *__restrict__ out_cnt, WinBatch wb) {
    __shared__ RowList<KNN_CAP> lists[KNN_WAVES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const WinRows &win = wb.w[WIN ? blockIdx.y : 0];
    if constexpr (WIN) {
        row_end = (int64_t)*win.n_a;
        out_idx = wb.out_idx[blockIdx.y];
        out_cnt = wb.out_cnt[blockIdx.y];
    }
    const int64_t i = row_begin + (int64_t)blockIdx.x * KNN_WAVES + wave;
    if (i >= row_end) return;
    RowList<KNN_CAP> &L = lists[wave];
    const int64_t arow = WIN ? (int64_t)win.arows[i] : i;
    const double ax = axy[2 * arow], ay = axy[2 * arow + 1];
    // unclamped cell of the aligned point; neighbours clipped to the grid
    const double fcx = __builtin_floor((ax - g.x0) * g.inv_cell), fcy = __builtin_floor((ay - g.y0) * g.inv_cell);
    int cnt = 0;
    // a point farther than one cell outside the grid has no neighbour within the radius
    if (fcx >= -1.0 && fcx <= (double)g.gx && fcy >= -1.0 && fcy <= (double)g.gy) {
        const int cx = (int)fcx, cy = (int)fcy;
        const int xlo = cx - 1 < 0 ? 0 : cx - 1, xhi = cx + 1 > g.gx - 1 ? g.gx - 1 : cx + 1;
        unsigned s[3], n[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int yy = cy - 1 + q;
            s[q] = 0; n[q] = 0;
            if (yy >= 0 && yy < g.gy && xlo <= xhi) {
                s[q] = start[(int64_t)yy * g.gx + xlo];
                n[q] = start[(int64_t)yy * g.gx + xhi + 1] - s[q];
            }
        }
        const unsigned total = n[0] + n[1] + n[2];
        typedef double double2_t __attribute__((ext_vector_type(2)));
        for (unsigned base = 0; base < total; base += 64) {
            const unsigned v = base + lane;
            const bool valid = v < total;
            unsigned pos = v < n[0] ? s[0] + v : (v < n[0] + n[1] ? s[1] + (v - n[0]) : s[2] + (v - n[0] - n[1]));
            if (!valid) pos = s[0];
            const double2_t p = *reinterpret_cast<const double2_t *>(sxy + 2 * (int64_t)pos);
            const int32_t j = sidx[pos];
            const double dx = p.x - ax, dy = p.y - ay;
            const double d2 = dx * dx + dy * dy;
            bool in = valid && d2 <= r2;
            if constexpr (WIN) in = in && p.x >= win.bx0 && p.x < win.bx1 && p.y >= win.by0 && p.y < win.by1;   // src/same.py:293-295
            const unsigned long long mask = __ballot(in);
            if (mask) {
                const int m = __builtin_popcountll(mask);
                if (cnt + m > KNN_CAP) cnt = prune_in_place(L, cnt, k, lane);
                const int slot = cnt + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                if (in) { L.d2[slot] = d2; L.j[slot] = j; }
                cnt += m;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    const int keep = cnt < k ? cnt : k;
    const int64_t o = (i - row_begin) * k;
#pragma unroll
    for (int p = 0; p < KNN_CAP / 64; ++p) {
        const int c = lane + 64 * p;
        if (c < cnt) {
            const double d = L.d2[c];
            const int32_t jj = L.j[c];
            const int rk = rank_of(L, cnt, d, jj);
            if (rk < k) {
                out_idx[o + rk] = jj;
                if (out_d2) out_d2[o + rk] = d;
            }
        }
    }
    for (int q = keep + lane; q < k; q += 64) {
        out_idx[o + q] = -1;
        if (out_d2) out_d2[o + q] = __builtin_inf();
    }
    if (lane == 0) out_cnt[i - row_begin] = keep;
}

static double key_to_f64(unsigned long long k) {
    const unsigned long long u = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    double d;
    memcpy(&d, &u, sizeof d);
    return d;
}

int launch_knn_brute(same_ctx *ctx, const double *daxy, const double *drxy, int64_t n_r, int64_t rb, int64_t re,
                     double radius, int k, int32_t *didx, double *dd2, int32_t *dcnt);

// Grid geometry for a reference set: bounding box by a device reduction and one 32-byte read-back (the only host
// synchronisation of the grid path -- which is why an unchanged reference set should keep its index, same_knn_index_build).
// *usable = false when the grid would not help or cannot be built (NaN/inf coordinates, radius spanning everything).
int grid_geometry(same_ctx *ctx, const double *drxy, int64_t n_r, double radius, GridDesc *out, bool *usable) {
    *usable = false;
    unsigned long long *dbbox;
    SAME_TRY(slot_as(ctx, SL_K_BBOX, (size_t)4, &dbbox));
    unsigned long long *h = static_cast<unsigned long long *>(ctx->pinned);
    h[0] = h[1] = ~0ull; h[2] = h[3] = 0ull;
    HIP_TRY(ctx, hipMemcpyAsync(dbbox, h, 4 * sizeof(unsigned long long), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(bbox_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(n_r, 256), 64)), dim3(256), 0, ctx->stream, drxy, n_r, dbbox);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(h + 8, dbbox, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const double x0 = key_to_f64(h[8]), y0 = key_to_f64(h[9]), x1 = key_to_f64(h[10]), y1 = key_to_f64(h[11]);
    if (!(x1 >= x0) || !(y1 >= y0) || !std::isfinite(x1 - x0) || !std::isfinite(y1 - y0)) return SAME_OK;  // NaN/inf coordinates
    // cell edge: a hair above the radius (so |dx| <= r can never skip a cell, rounding included), and
    // large enough to keep the grid below ~1M cells and ~1 ref per cell on sparse inputs
    const double ext = std::max(x1 - x0, y1 - y0);
    double cell = radius * (1.0 + 1e-9);
    cell = std::max(cell, ext / 1024.0);
    cell = std::max(cell, std::sqrt((x1 - x0) * (y1 - y0) / (double)std::max<int64_t>(n_r, 1)) * 0.5);
    if (!(cell > 0.0)) cell = 1.0;  // all references coincide and radius == 0
    GridDesc g;
    g.x0 = x0; g.y0 = y0; g.inv_cell = 1.0 / cell;
    g.gx = (int)std::min(1025.0, std::floor((x1 - x0) / cell) + 1.0);
    g.gy = (int)std::min(1025.0, std::floor((y1 - y0) / cell) + 1.0);
    *out = g;
    // <= 9 cells: the radius spans the whole reference set, the 3 x 3 neighbourhood is everything -> brute force, 8 rows per wave
    *usable = (int64_t)g.gx * g.gy > 9;
    return SAME_OK;
}

// counting sort of the references by cell: start[cells+1] (exclusive scan of the counts), sorted XY and original indices
int grid_fill(same_ctx *ctx, const double *drxy, int64_t n_r, const GridDesc &g, unsigned *dstart, unsigned *drank, double *dsxy,
              int32_t *dsidx) {
    const int64_t cells = (int64_t)g.gx * g.gy;
    // scratch: [scan words | counts per cell], zeroed by one fill
    const size_t st_bytes = scan::status_bytes(cells);
    unsigned char *scratch;
    SAME_TRY(slot_as(ctx, SL_K_HIST, st_bytes + (size_t)cells * sizeof(unsigned), &scratch));
    unsigned long long *dstatus = reinterpret_cast<unsigned long long *>(scratch);
    unsigned *dcount = reinterpret_cast<unsigned *>(scratch + st_bytes);
    HIP_TRY(ctx, hipMemsetAsync(scratch, 0, st_bytes + (size_t)cells * sizeof(unsigned), ctx->stream));
    hipLaunchKernelGGL(grid_count_kernel, dim3((unsigned)ceil_div(n_r, 256)), dim3(256), 0, ctx->stream, drxy, n_r, g, dcount, drank);
    hipLaunchKernelGGL(grid_scan_kernel, dim3(scan::blocks_for(cells)), dim3(scan::NT), 0, ctx->stream, dcount, cells, scan::arg(dstatus), dstart);
    hipLaunchKernelGGL(grid_scatter_kernel, dim3((unsigned)ceil_div(n_r, 256)), dim3(256), 0, ctx->stream, drxy, n_r, g, dstart, drank,
                       dsxy, dsidx);
    HIP_TRY(ctx, hipGetLastError());
    return SAME_OK;
}

int grid_query(same_ctx *ctx, const double *daxy, const GridDesc &g, const unsigned *dhist, const double *dsxy, const int32_t *dsidx,
               int64_t rb, int64_t re, double radius, int k, int32_t *didx, double *dd2, int32_t *dcnt) {
    const int64_t rows = re - rb;
    if (rows <= 0) return SAME_OK;
    REQUIRE(ctx, ceil_div(rows, KNN_WAVES) < (int64_t)1 << 31);
    if (k <= KNN_CAP_SMALL - 64)
        hipLaunchKernelGGL(knn_grid_kernel<KNN_CAP_SMALL>, dim3((unsigned)ceil_div(rows, KNN_WAVES)), dim3(64 * KNN_WAVES), 0, ctx->stream,
                           daxy, dsxy, dsidx, dhist, g, rb, re, radius * radius, k, didx, dd2, dcnt, WinBatch{});
    else
        hipLaunchKernelGGL(knn_grid_kernel<KNN_CAP_LARGE>, dim3((unsigned)ceil_div(rows, KNN_WAVES)), dim3(64 * KNN_WAVES), 0, ctx->stream,
                           daxy, dsxy, dsidx, dhist, g, rb, re, radius * radius, k, didx, dd2, dcnt, WinBatch{});
    HIP_TRY(ctx, hipGetLastError());
    return SAME_OK;
}

int launch_knn_grid(same_ctx *ctx, const double *daxy, const double *drxy, int64_t n_r, int64_t rb, int64_t re,
                    double radius, int k, int32_t *didx, double *dd2, int32_t *dcnt) {
    GridDesc g;
    bool usable = false;
    SAME_TRY(grid_geometry(ctx, drxy, n_r, radius, &g, &usable));
    if (!usable) return launch_knn_brute(ctx, daxy, drxy, n_r, rb, re, radius, k, didx, dd2, dcnt);
    const int64_t cells = (int64_t)g.gx * g.gy;
    unsigned *dhist, *drank;
    double *dsxy;
    int32_t *dsidx;
    SAME_TRY(slot_as(ctx, SL_K_START, (size_t)cells + 1, &dhist));
    SAME_TRY(slot_as(ctx, SL_K_RANK, (size_t)n_r, &drank));
    SAME_TRY(slot_as(ctx, SL_K_SXY, (size_t)n_r * 2, &dsxy));
    SAME_TRY(slot_as(ctx, SL_K_SIDX, (size_t)n_r, &dsidx));
    SAME_TRY(grid_fill(ctx, drxy, n_r, g, dhist, drank, dsxy, dsidx));
    return grid_query(ctx, daxy, g, dhist, dsxy, dsidx, rb, re, radius, k, didx, dd2, dcnt);
}

int launch_knn_brute(same_ctx *ctx, const double *daxy, const double *drxy, int64_t n_r, int64_t rb, int64_t re,
                     double radius, int k, int32_t *didx, double *dd2, int32_t *dcnt) {
    const int64_t rows = re - rb;
    if (rows == 0) return SAME_OK;
    const bool small = k <= KNN_CAP_SMALL - 64;
    const int64_t blocks = ceil_div(rows, KNN_WAVES * (small ? KNN_RW_SMALL : KNN_RW_LARGE));
    REQUIRE(ctx, blocks < (int64_t)1 << 31);
    if (small)
        hipLaunchKernelGGL((knn_prune_kernel<KNN_CAP_SMALL, KNN_RW_SMALL>), dim3((unsigned)blocks), dim3(64 * KNN_WAVES), 0, ctx->stream,
                           daxy, drxy, n_r, rb, re, radius * radius, k, didx, dd2, dcnt, WinBatch{});
    else
        hipLaunchKernelGGL((knn_prune_kernel<KNN_CAP_LARGE, KNN_RW_LARGE>), dim3((unsigned)blocks), dim3(64 * KNN_WAVES), 0, ctx->stream,
                           daxy, drxy, n_r, rb, re, radius * radius, k, didx, dd2, dcnt, WinBatch{});
    HIP_TRY(ctx, hipGetLastError());
    return SAME_OK;
}

// Small problems stay on the single brute-force launch (the grid build costs four small launches and a
// 32-byte read-back); measured crossover is below 3 000 x 3 000 (0.047 vs 0.052 ms; 0.128 vs 3.9 ms at
// 100k x 100k, tools/knn_threshold_probe.py).  SAME_KNN_MODE=brute|grid overrides for testing.
int launch_knn(same_ctx *ctx, const double *daxy, const double *drxy, int64_t n_r, int64_t rb, int64_t re, double radius,
               int k, int32_t *didx, double *dd2, int32_t *dcnt) {
    const int64_t rows = re - rb;
    if (rows == 0) return SAME_OK;
    const char *mode = getenv("SAME_KNN_MODE");
    bool grid = n_r >= 2048 && (double)n_r * (double)rows >= 4.0e6 && std::isfinite(radius);
    if (mode && mode[0] == 'b') grid = false;
    if (mode && mode[0] == 'g') grid = n_r > 0;
    return grid ? launch_knn_grid(ctx, daxy, drxy, n_r, rb, re, radius, k, didx, dd2, dcnt)
                : launch_knn_brute(ctx, daxy, drxy, n_r, rb, re, radius, k, didx, dd2, dcnt);
}

}  // namespace

// Caller-held index of one reference set for one radius: the grid (or the decision that brute force is the
// better plan) is made once; prunes against it neither rebuild anything nor synchronise with the host.
struct same_knn_index {
    same_ctx *ctx = nullptr;       // the context that built it (not used after the build: an index may outlive it)
    int device = -1;
    const double *drxy = nullptr;  // the caller's device array (not owned; must stay valid and unchanged)
    int64_t n_r = 0;
    double radius = 0.0;
    bool grid = false;
    GridDesc g{};
    unsigned *hist = nullptr;
    double *sxy = nullptr;
    int32_t *sidx = nullptr;
};

// The window path's prune (csrc/window.hip; declared in common.h) for the windows of a batch in ONE launch: `ix` indexes the reference
// SECTION (all of its rows, built once per radius); a job's aligned rows are rows_m[0, *dn_m) of the moving section's XY, its
// candidates the reference rows inside its box (grid form: box test on the swept cells; brute form: the window's own ascending list
// rows_r[0, *dn_r)).  idx[cap_m][k] gets section rows (-1 padded), cnt[cap_m] the list lengths; rows past *dn_m are not written.
// Enqueue only.
int same_knn_window_batch_core(same_ctx *ctx, const same_knn_index *ix, const double *dmov_xy, const same_knn_window_job *jobs, int n_jobs, int k) {
    REQUIRE(ctx, ix && ix->device == ctx->device && k >= 1 && k <= SAME_MAX_KNN && n_jobs >= 1 && n_jobs <= SAME_LAUNCH_WINDOWS);
    WinBatch wb{};
    int64_t cap = 0;
    for (int q = 0; q < n_jobs; ++q) {
        const same_knn_window_job &j = jobs[q];
        REQUIRE(ctx, j.cap_m >= 0);
        wb.w[q] = WinRows{j.rows_m, j.rows_r, j.dn_m, j.dn_r, j.box[0], j.box[1], j.box[2], j.box[3]};
        wb.out_idx[q] = j.idx;
        wb.out_cnt[q] = j.cnt;
        cap = std::max(cap, j.cap_m);
    }
    if (cap == 0) return SAME_OK;
    const bool small = k <= KNN_CAP_SMALL - 64;
    const double r2 = ix->radius * ix->radius;
    if (ix->grid) {
        const dim3 blocks((unsigned)ceil_div(cap, KNN_WAVES), (unsigned)n_jobs);
        if (small)
            SAME_LAUNCH(ctx, (knn_grid_kernel<KNN_CAP_SMALL, true>), blocks, dim3(64 * KNN_WAVES), 0, dmov_xy, ix->sxy, ix->sidx, ix->hist, ix->g,
                        (int64_t)0, cap, r2, k, (int32_t *)nullptr, (double *)nullptr, (int32_t *)nullptr, wb);
        else
            SAME_LAUNCH(ctx, (knn_grid_kernel<KNN_CAP_LARGE, true>), blocks, dim3(64 * KNN_WAVES), 0, dmov_xy, ix->sxy, ix->sidx, ix->hist, ix->g,
                        (int64_t)0, cap, r2, k, (int32_t *)nullptr, (double *)nullptr, (int32_t *)nullptr, wb);
    } else {
        const dim3 blocks((unsigned)ceil_div(cap, KNN_WAVES * (small ? KNN_RW_SMALL : KNN_RW_LARGE)), (unsigned)n_jobs);
        if (small)
            SAME_LAUNCH(ctx, (knn_prune_kernel<KNN_CAP_SMALL, KNN_RW_SMALL, true>), blocks, dim3(64 * KNN_WAVES), 0, dmov_xy, ix->drxy, (int64_t)0,
                        (int64_t)0, cap, r2, k, (int32_t *)nullptr, (double *)nullptr, (int32_t *)nullptr, wb);
        else
            SAME_LAUNCH(ctx, (knn_prune_kernel<KNN_CAP_LARGE, KNN_RW_LARGE, true>), blocks, dim3(64 * KNN_WAVES), 0, dmov_xy, ix->drxy, (int64_t)0,
                        (int64_t)0, cap, r2, k, (int32_t *)nullptr, (double *)nullptr, (int32_t *)nullptr, wb);
    }
    HIP_TRY(ctx, hipGetLastError());
    return SAME_OK;
}

extern "C" {

int same_knn_index_build(same_ctx *ctx, const double *drxy, int64_t n_r, double radius, same_knn_index **out) {
    REQUIRE(ctx, ctx && out);
    *out = nullptr;
    REQUIRE(ctx, n_r >= 0 && (n_r == 0 || drxy) && radius >= 0.0);  // also rejects NaN
    SAME_TRY(same_use(ctx));
    same_knn_index *ix = new (std::nothrow) same_knn_index();
    if (!ix) return SAME_ENOMEM;
    ix->ctx = ctx; ix->device = ctx->device; ix->drxy = drxy; ix->n_r = n_r; ix->radius = radius;
    const char *mode = getenv("SAME_KNN_MODE");
    bool want_grid = n_r >= 2048 && std::isfinite(radius);
    if (mode && mode[0] == 'b') want_grid = false;
    if (mode && mode[0] == 'g') want_grid = n_r > 0;
    int rc = SAME_OK;
    if (want_grid) {
        bool usable = false;
        rc = grid_geometry(ctx, drxy, n_r, radius, &ix->g, &usable);
        if (rc == SAME_OK && usable) {
            const int64_t cells = (int64_t)ix->g.gx * ix->g.gy;
            void *rank = nullptr;
            hipError_t e = hipMalloc(reinterpret_cast<void **>(&ix->hist), (size_t)(cells + 1) * sizeof(unsigned));
            if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&ix->sxy), (size_t)n_r * 2 * sizeof(double));
            if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&ix->sidx), (size_t)n_r * sizeof(int32_t));
            if (e == hipSuccess) e = hipMalloc(&rank, (size_t)n_r * sizeof(unsigned));
            if (e != hipSuccess) rc = same_fail(ctx, e == hipErrorOutOfMemory ? SAME_ENOMEM : SAME_EIO, "hipMalloc(knn index)", e);
            if (rc == SAME_OK) rc = grid_fill(ctx, drxy, n_r, ix->g, ix->hist, static_cast<unsigned *>(rank), ix->sxy, ix->sidx);
            if (rc == SAME_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = SAME_EIO;
            if (rank) (void)hipFree(rank);
            ix->grid = rc == SAME_OK;
        }
    }
    if (rc != SAME_OK) { same_knn_index_destroy(ix); return rc; }
    *out = ix;
    return SAME_OK;
}

void same_knn_index_destroy(same_knn_index *ix) {
    if (!ix) return;
    (void)hipSetDevice(ix->device);        // hipFree waits for the device's outstanding work itself
    if (ix->hist) (void)hipFree(ix->hist);
    if (ix->sxy) (void)hipFree(ix->sxy);
    if (ix->sidx) (void)hipFree(ix->sidx);
    delete ix;
}

int same_knn_prune_indexed_dev(same_ctx *ctx, const same_knn_index *ix, const double *daxy, int64_t row_begin,
                               int64_t row_end, int k, int32_t *dout_idx, double *dout_d2, int32_t *dout_cnt) {
    REQUIRE(ctx, ctx && ix && ix->ctx == ctx && daxy && dout_idx && dout_cnt);   // the public form: the index of this very context
    REQUIRE(ctx, row_begin >= 0 && row_end >= row_begin && k >= 1 && k <= SAME_MAX_KNN);
    SAME_TRY(same_use(ctx));
    if (row_end == row_begin) return SAME_OK;
    if (ix->grid)
        return grid_query(ctx, daxy, ix->g, ix->hist, ix->sxy, ix->sidx, row_begin, row_end, ix->radius, k, dout_idx, dout_d2, dout_cnt);
    return launch_knn_brute(ctx, daxy, ix->drxy, ix->n_r, row_begin, row_end, ix->radius, k, dout_idx, dout_d2, dout_cnt);
}

int same_knn_prune_dev(same_ctx *ctx, const double *daxy, const double *drxy, int64_t n_r, int64_t row_begin,
                       int64_t row_end, double radius, int k, int32_t *dout_idx, double *dout_d2,
                       int32_t *dout_cnt) {
    REQUIRE(ctx, ctx && daxy && dout_idx && dout_cnt && (n_r == 0 || drxy));
    REQUIRE(ctx, n_r >= 0 && row_begin >= 0 && row_end >= row_begin && k >= 1 && k <= SAME_MAX_KNN);
    REQUIRE(ctx, radius >= 0.0);  // also rejects NaN
    SAME_TRY(same_use(ctx));
    return launch_knn(ctx, daxy, drxy, n_r, row_begin, row_end, radius, k, dout_idx, dout_d2, dout_cnt);
}

int same_knn_prune(same_ctx *ctx, const double *axy, int64_t n_m, const double *rxy, int64_t n_r, int64_t row_begin,
                   int64_t row_end, double radius, int k, int32_t *out_idx, double *out_d2, int32_t *out_cnt) {
    REQUIRE(ctx, ctx != nullptr);
    REQUIRE(ctx, n_m >= 0 && n_r >= 0 && row_begin >= 0 && row_end >= row_begin && row_end <= n_m);
    REQUIRE(ctx, k >= 1 && k <= SAME_MAX_KNN && radius >= 0.0);
    const int64_t rows = row_end - row_begin;
    if (rows == 0) return SAME_OK;
    REQUIRE(ctx, axy && out_idx && out_cnt && (n_r == 0 || rxy));
    SAME_TRY(same_use(ctx));
    double *dax, *drx, *dd2 = nullptr;
    int32_t *didx, *dcnt;
    SAME_TRY(up_as(ctx, SL_AXY, axy, (size_t)n_m * 2, &dax));
    SAME_TRY(up_as(ctx, SL_RXY, rxy, (size_t)n_r * 2, &drx));
    SAME_TRY(slot_as(ctx, SL_OUT0, (size_t)rows * k, &didx));
    SAME_TRY(slot_as(ctx, SL_OUT2, (size_t)rows, &dcnt));
    if (out_d2) SAME_TRY(slot_as(ctx, SL_OUT1, (size_t)rows * k, &dd2));
    SAME_TRY(launch_knn(ctx, dax, drx, n_r, row_begin, row_end, radius, k, didx, dd2, dcnt));
    SAME_TRY(same_down(ctx, out_idx, didx, (size_t)rows * k * sizeof(int32_t)));
    SAME_TRY(same_down(ctx, out_cnt, dcnt, (size_t)rows * sizeof(int32_t)));
    if (out_d2) SAME_TRY(same_down(ctx, out_d2, dd2, (size_t)rows * k * sizeof(double)));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SAME_OK;
}

}  // extern "C"
