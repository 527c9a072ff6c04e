// window.hip -- one window of the sliding-window path (src/same.py:507-593) with both sections RESIDENT on the device.
//
// The column pipeline (same_amd/windows.py::iter_window_arrays) subsets, prunes, compacts and gathers on the host and hands
// every kernel its operands through host buffers: at a million cells per section the window's ~10^4 rows are cache-missing
// gathers from 64 MB columns, and that host work -- not the kernels -- bounds BASELINE cfg 5.  Here a section's columns are
// uploaded once (same_section) and a window is three calls:
//
//   same_window_stage   box test over the sections' rows + ordered compaction (= np.flatnonzero of src/same.py:293-295, rows
//                       ascending), row gathers, radius / k prune (src/utils.py:709-728), costs of the candidate lists in the
//                       cost type (src/same.py:1180-1189), compaction of the aligned side and of the pair list
//                       (src/utils.py:734-742).  Back to the host: four counts, and (same_window_fetch) the kept aligned rows
//                       and their XY -- the input of the host's Delaunay call (src/same.py:1023).
//   same_window_filter  the Delaunay simplices in; triangle classes (src/helpers.py:300-330), the keep list and the same-type
//                       triangles added back so that every node keeps one (src/helpers.py:331-340, :365-389) -- all on the
//                       device, in the reference's order.  (A cosine within 8 ulp of the angle threshold is left to the host,
//                       which re-decides it with the reference's literal arccos: the call then only reports it.)
//   same_window_finish  kept triangles in (or the ones same_window_filter left on the device); source signs / weights
//                       (src/same.py:1128-1146), per-row minimum and the greedy MIP start (src/init_helpers.py:104-133), the
//                       lazy-constraint body under that incumbent (src/same.py:645-669), XY-order sweep
//                       (src/violationhelper.py:53-117), signed-area flips (src/same.py:1362-1402).  Back to the host: the
//                       matched reference row per kept aligned cell, the per-cell violation flag and eight counters.
//
// Reference cells are NOT renumbered (the reference drops unreferenced ones, src/utils.py:740-742): costs, the greedy rule and
// the sweeps read coordinates and pair order only, which a monotone renumbering does not change; the match comes back as
// section rows.  Everything reuses the kernels of the other translation units through their _dev entry points / cores, so
// a window's numbers are those of the column pipeline bit for bit (tests/test_gpu_run_same.py::test_device_windows_*).
#include <algorithm>
#include <new>

#include "common.h"

namespace {

typedef double double2_t __attribute__((ext_vector_type(2)));

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
};

int ensure(same_ctx *ctx, DevBuf &b, size_t bytes) {
    if (bytes <= b.bytes && b.p) return SAME_OK;
    const size_t want = std::max<size_t>(bytes + bytes / 4, 256);
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
template <typename T>
inline T *as(const DevBuf &b) { return static_cast<T *>(b.p); }

inline unsigned grid_for(int64_t n) { return (unsigned)ceil_div(n > 0 ? n : 1, 256); }

// rows inside the half-open box, one bit per row (src/same.py:293-295; NaN coordinates fail every comparison)
__global__ __launch_bounds__(256) void box_mask_kernel(const double *__restrict__ xy, int64_t n, double x0, double x1, double y0,
                                                        double y1, unsigned long long *__restrict__ mask) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool in = false;
    if (i < n) {
        const double2_t p = *reinterpret_cast<const double2_t *>(xy + 2 * i);
        in = p.x >= x0 && p.x < x1 && p.y >= y0 && p.y < y1;
    }
    const unsigned long long bal = __ballot(in);
    if ((threadIdx.x & 63) == 0) mask[i >> 6] = bal;
}

// dst[q][0..words) = src[pos[q]][0..words): rows of 4-byte words (XY pairs, type rows, sizes)
__global__ __launch_bounds__(256) void gather_rows_kernel(const uint32_t *__restrict__ src, int words, const int32_t *__restrict__ pos,
                                                           int64_t n, uint32_t *__restrict__ dst) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * words) return;
    const int64_t q = e / words;
    const int w = (int)(e - q * words);
    dst[e] = src[(int64_t)pos[q] * words + w];
}

__global__ __launch_bounds__(256) void to_float_kernel(const double *__restrict__ src, int64_t n, float *__restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (float)src[i];          // round to nearest even, as numpy's astype(float32)
}

// one block: exclusive scans of (cnt > 0) and cnt over the window's aligned rows; totals -> counts[2] = kept rows, counts[3] = pairs
__global__ __launch_bounds__(1024) void window_scan_kernel(const int32_t *__restrict__ cnt, int64_t n, int32_t *__restrict__ a_off,
                                                            int32_t *__restrict__ p_off, unsigned long long *__restrict__ counts) {
    __shared__ int wave_a[16], wave_p[16];
    __shared__ long long carry_a, carry_p;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) { carry_a = 0; carry_p = 0; }
    __syncthreads();
    for (int64_t base = 0; base < n; base += 1024) {
        const int64_t i = base + tid;
        const int c = i < n ? cnt[i] : 0;
        const int u = c > 0;
        int ia = u, ip = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int va = __shfl_up(ia, off, 64), vp = __shfl_up(ip, off, 64);
            if (lane >= off) { ia += va; ip += vp; }
        }
        if (lane == 63) { wave_a[wave] = ia; wave_p[wave] = ip; }
        __syncthreads();
        int oa = 0, op = 0;
        for (int q = 0; q < wave; ++q) { oa += wave_a[q]; op += wave_p[q]; }
        if (i < n) {
            a_off[i] = (int32_t)(carry_a + oa + ia - u);
            p_off[i] = (int32_t)(carry_p + op + ip - c);
        }
        __syncthreads();
        if (tid == 1023) { carry_a += oa + ia; carry_p += op + ip; }
        __syncthreads();
    }
    if (tid == 0) { counts[2] = (unsigned long long)carry_a; counts[3] = (unsigned long long)carry_p; }
}

// compaction of the aligned side and of the pair list (src/utils.py:734-742): one lane per aligned row with candidates
template <typename F>
__global__ __launch_bounds__(256) void window_scatter_kernel(
    const int32_t *__restrict__ idx, const F *__restrict__ cost, const int32_t *__restrict__ cnt, int64_t n_m, int k,
    const int32_t *__restrict__ a_off, const int32_t *__restrict__ p_off, const int32_t *__restrict__ rows_m,
    const double *__restrict__ axy_w, const double *__restrict__ size_w, const int32_t *__restrict__ type_w, int32_t *__restrict__ ua,
    int32_t *__restrict__ rows_ua, double *__restrict__ axy_c, double *__restrict__ size_c, int32_t *__restrict__ type_c,
    int32_t *__restrict__ pairs, double *__restrict__ cost64) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_m || cnt[i] <= 0) return;
    const int32_t a = a_off[i];
    ua[a] = (int32_t)i;
    rows_ua[a] = rows_m[i];
    axy_c[2 * (int64_t)a] = axy_w[2 * i];
    axy_c[2 * (int64_t)a + 1] = axy_w[2 * i + 1];
    size_c[a] = size_w[i];
    if (type_w) type_c[a] = type_w[i];
    int64_t p = p_off[i];
    const int64_t p_end = p + cnt[i];              // the scan sized the list by cnt: never write past this row's share
    for (int q = 0; q < k && p < p_end; ++q) {
        const int32_t j = idx[i * k + q];
        if (j >= 0) {
            pairs[2 * p] = a;
            pairs[2 * p + 1] = j;
            cost64[p] = (double)cost[i * k + q];
            ++p;
        }
    }
}

// rows whose best pair beats their no-match penalty (src/init_helpers.py:118-122)
__global__ __launch_bounds__(256) void prefer_kernel(const double *__restrict__ rowmin, const double *__restrict__ size, int64_t n,
                                                      double penalty, uint8_t *__restrict__ prefer) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) prefer[i] = rowmin[i] < penalty * size[i];
}

// pair per row -> matched reference (window numbering, for the sweeps) and its section row (for the caller)
__global__ __launch_bounds__(256) void match_rows_kernel(const int32_t *__restrict__ pair_of_row, const int32_t *__restrict__ pairs,
                                                          const int32_t *__restrict__ rows_r, int64_t n, int32_t *__restrict__ match,
                                                          int32_t *__restrict__ match_row, unsigned long long *__restrict__ stats) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool m = false;
    if (i < n) {
        const int32_t p = pair_of_row[i];
        const int32_t j = p >= 0 ? pairs[2 * (int64_t)p + 1] : -1;
        match[i] = j;
        match_row[i] = j >= 0 ? rows_r[j] : -1;
        m = j >= 0;
    }
    const unsigned long long bal = __ballot(m);
    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(&stats[7], (unsigned long long)__builtin_popcountll(bal));
}

__global__ __launch_bounds__(256) void count_flags_kernel(const uint8_t *__restrict__ flag, int64_t n, unsigned long long *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long bal = __ballot(i < n && flag[i] != 0);
    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(out, (unsigned long long)__builtin_popcountll(bal));
}

// ---- triangle filter (src/helpers.py:233-395) on the device ------------------------------------------------------------------
// classes come from same_tri_classify_dev; this marks what the re-add pass needs: vertices with a kept triangle, vertices with
// any valid (kept or same-type) triangle, the keep mask, and how many cosines sit within `tol` of the threshold
__global__ __launch_bounds__(256) void filter_mark_kernel(const uint8_t *__restrict__ cls, const double *__restrict__ maxcos,
                                                           const int32_t *__restrict__ tris, int64_t Tr, int near_enabled, double thr,
                                                           double tol, uint8_t *__restrict__ has_kept, uint8_t *__restrict__ any_valid,
                                                           unsigned long long *__restrict__ keep_mask,
                                                           unsigned long long *__restrict__ counters) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool keep = false, near = false;
    if (t < Tr) {
        const uint8_t c = cls[t];
        keep = c == 0;
        near = near_enabled && c != 1 && fabs(maxcos[t] - thr) <= tol;
        if (c == 0 || c == 3) {
            const int32_t a = tris[3 * t], b = tris[3 * t + 1], d = tris[3 * t + 2];
            any_valid[a] = 1; any_valid[b] = 1; any_valid[d] = 1;
            if (c == 0) { has_kept[a] = 1; has_kept[b] = 1; has_kept[d] = 1; }
        }
    }
    const unsigned long long kb = __ballot(keep), nb = __ballot(near);
    if ((threadIdx.x & 63) == 0) {
        keep_mask[t >> 6] = kb;
        if (nb) atomicAdd(&counters[2], (unsigned long long)__builtin_popcountll(nb));
    }
}

// best same-type triangle of every vertex = smallest perimeter, first in input order on ties (src/helpers.py:334-340):
// two passes of atomic minima, first over the perimeter's bit pattern (perimeters are >= 0: the order of the bits is theirs)
__global__ __launch_bounds__(256) void filter_best_perim_kernel(const uint8_t *__restrict__ cls, const double *__restrict__ perim,
                                                                 const int32_t *__restrict__ tris, int64_t Tr,
                                                                 unsigned long long *__restrict__ best_p) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= Tr || cls[t] != 3) return;
    const unsigned long long key = (unsigned long long)__double_as_longlong(perim[t]);
    for (int q = 0; q < 3; ++q) atomicMin(&best_p[tris[3 * t + q]], key);
}
__global__ __launch_bounds__(256) void filter_best_tri_kernel(const uint8_t *__restrict__ cls, const double *__restrict__ perim,
                                                               const int32_t *__restrict__ tris, int64_t Tr,
                                                               const unsigned long long *__restrict__ best_p, unsigned *__restrict__ best_t) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= Tr || cls[t] != 3) return;
    const unsigned long long key = (unsigned long long)__double_as_longlong(perim[t]);
    for (int q = 0; q < 3; ++q) {
        const int32_t v = tris[3 * t + q];
        if (best_p[v] == key) atomicMin(&best_t[v], (unsigned)t);
    }
}
// nodes without a kept triangle but with a valid one are walked in ascending order and bring their best triangle along unless an
// earlier node already did (src/helpers.py:365-389): first_v[t] = the first node that asks for t ...
__global__ __launch_bounds__(256) void filter_first_node_kernel(const uint8_t *__restrict__ has_kept, const uint8_t *__restrict__ any_valid,
                                                                 const unsigned *__restrict__ best_t, int64_t n, unsigned *__restrict__ first_v) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n || has_kept[v] || !any_valid[v] || best_t[v] == ~0u) return;
    atomicMin(&first_v[best_t[v]], (unsigned)v);
}
// ... and the nodes that are the first to ask, as a mask over the nodes (compacted in order afterwards)
__global__ __launch_bounds__(256) void filter_owner_mask_kernel(const uint8_t *__restrict__ has_kept, const uint8_t *__restrict__ any_valid,
                                                                 const unsigned *__restrict__ best_t, const unsigned *__restrict__ first_v,
                                                                 int64_t n, unsigned long long *__restrict__ mask) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool own = false;
    if (v < n && !has_kept[v] && any_valid[v] && best_t[v] != ~0u) own = first_v[best_t[v]] == (unsigned)v;
    const unsigned long long bal = __ballot(own);
    if ((threadIdx.x & 63) == 0) mask[v >> 6] = bal;
}
// the kept triangles in the reference's order: class-0 triangles ascending, then the added-back ones in walk order
__global__ __launch_bounds__(256) void filter_emit_kernel(const int32_t *__restrict__ raw, const int32_t *__restrict__ keep_list, int64_t n_keep,
                                                           const int32_t *__restrict__ owner_list, int64_t n_add,
                                                           const unsigned *__restrict__ best_t, int32_t *__restrict__ out) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_keep + n_add) return;
    const int64_t t = q < n_keep ? keep_list[q] : (int64_t)best_t[owner_list[q - n_keep]];
    out[3 * q] = raw[3 * t];
    out[3 * q + 1] = raw[3 * t + 1];
    out[3 * q + 2] = raw[3 * t + 2];
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
};

struct same_window {
    same_ctx *ctx = nullptr;
    int cost_f32 = 0, k = 0, staged = 0, finished = 0, has_type = 0, filtered = 0;
    int64_t n_m = 0, n_r = 0, n_ua = 0, P = 0, Tr = 0;
    DevBuf mask, counts, rows_m, rows_r, axy_w, rxy_w, axyc_w, rxyc_w, A_w, R_w, size_w, idx, cnt, cost, a_off, p_off, ua, rows_ua,
        axy_c, size_c, pairs, cost64;
    DevBuf type_w, type_c, raw, cls, perim, maxcos, kmask, has_kept, any_valid, best_p, best_t, first_v, nmask, nlist, klist;
    DevBuf tris, sign, weight, rowmin, prefer, pair_of_row, match, match_row, oflag, omask, edge, tflag, pflag, before, after, m3, flipped;
    void *host = nullptr;     // pinned staging for everything that comes back
    size_t host_bytes = 0;
};

namespace {

int ensure_host(same_window *w, size_t bytes) {
    if (bytes <= w->host_bytes) return SAME_OK;
    same_ctx *ctx = w->ctx;
    if (w->host) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipHostFree(w->host));
        w->host = nullptr;
        w->host_bytes = 0;
    }
    const size_t want = bytes + bytes / 4 + 4096;
    HIP_TRY(ctx, hipHostMalloc(&w->host, want, hipHostMallocDefault));
    w->host_bytes = want;
    return SAME_OK;
}

// Layout of the pinned staging block: [0, 64) scalars; the stage call's results (kept aligned XY, then their section rows, at the
// capacity n_m) from byte 64; the finish call's results behind them.  Sized once per window by the stage call.
inline size_t finish_off(int64_t n_m) { return (64 + (size_t)n_m * (sizeof(int32_t) + 2 * sizeof(double)) + 127) & ~size_t(63); }
inline size_t host_need(int64_t n_m) { return finish_off(n_m) + 128 + (size_t)n_m * 5 + 64; }

// rows of `sec` inside the box, ascending, into dst (sized for them); their number into *out_n.  One read-back.
int subset_rows(same_window *w, const same_section *sec, const double *box, DevBuf &dst, int64_t *out_n) {
    same_ctx *ctx = w->ctx;
    *out_n = 0;
    if (sec->n == 0) return ensure(ctx, dst, 4);
    const int64_t n_words = (int64_t)grid_for(sec->n) * 4;
    SAME_TRY(ensure(ctx, w->mask, (size_t)n_words * sizeof(unsigned long long)));
    int32_t *scratch;
    SAME_TRY(slot_as(ctx, SL_OUT0, (size_t)sec->n, &scratch));
    unsigned long long *dc = as<unsigned long long>(w->counts);
    hipLaunchKernelGGL(box_mask_kernel, dim3(grid_for(sec->n)), dim3(256), 0, ctx->stream, sec->xy, sec->n, box[0], box[1], box[2], box[3],
                       as<unsigned long long>(w->mask));
    HIP_TRY(ctx, hipGetLastError());
    SAME_TRY(same_compact_mask_core(ctx, as<unsigned long long>(w->mask), n_words, sec->n, scratch, dc));
    unsigned long long *h = static_cast<unsigned long long *>(w->host);
    HIP_TRY(ctx, hipMemcpyAsync(h, dc + 1, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const int64_t n = (int64_t)h[0];
    SAME_TRY(ensure(ctx, dst, (size_t)std::max<int64_t>(n, 1) * sizeof(int32_t)));
    if (n) HIP_TRY(ctx, hipMemcpyAsync(dst.p, scratch, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToDevice, ctx->stream));
    *out_n = n;
    return SAME_OK;
}

int gather(same_ctx *ctx, const void *src, size_t row_bytes, const int32_t *pos, int64_t n, DevBuf &dst) {
    SAME_TRY(ensure(ctx, dst, std::max<size_t>((size_t)n * row_bytes, 16)));
    if (n == 0 || row_bytes == 0) return SAME_OK;
    const int words = (int)(row_bytes / 4);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(n * words)), dim3(256), 0, ctx->stream, static_cast<const uint32_t *>(src), words, pos,
                       n, as<uint32_t>(dst));
    HIP_TRY(ctx, hipGetLastError());
    return SAME_OK;
}

void release(DevBuf &b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
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
    return SAME_OK;
}

void same_section_destroy(same_section *s) {
    if (!s) return;
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->ctx->stream);
    if (s->xy_c && s->xy_c != s->xy) (void)hipFree(s->xy_c);
    if (s->xy) (void)hipFree(s->xy);
    if (s->types_c) (void)hipFree(s->types_c);
    if (s->size) (void)hipFree(s->size);
    if (s->type_id) (void)hipFree(s->type_id);
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
    SAME_TRY(ensure(ctx, w->counts, 16 * sizeof(unsigned long long)));
    SAME_TRY(ensure_host(w, 1 << 16));
    return SAME_OK;
}

void same_window_destroy(same_window *w) {
    if (!w) return;
    (void)hipSetDevice(w->ctx->device);
    (void)hipStreamSynchronize(w->ctx->stream);
    DevBuf *all[] = {&w->mask, &w->counts, &w->rows_m, &w->rows_r, &w->axy_w, &w->rxy_w, &w->axyc_w, &w->rxyc_w, &w->A_w, &w->R_w, &w->size_w,
                     &w->idx, &w->cnt, &w->cost, &w->a_off, &w->p_off, &w->ua, &w->rows_ua, &w->axy_c, &w->size_c, &w->pairs, &w->cost64,
                     &w->type_w, &w->type_c, &w->raw, &w->cls, &w->perim, &w->maxcos, &w->kmask, &w->has_kept, &w->any_valid, &w->best_p,
                     &w->best_t, &w->first_v, &w->nmask, &w->nlist, &w->klist, &w->tris, &w->sign, &w->weight, &w->rowmin, &w->prefer,
                     &w->pair_of_row, &w->match, &w->match_row, &w->oflag, &w->omask, &w->edge, &w->tflag, &w->pflag, &w->before, &w->after,
                     &w->m3, &w->flipped};
    for (DevBuf *b : all) release(*b);
    if (w->host) (void)hipHostFree(w->host);
    delete w;
}

int same_window_stage(same_window *w, const same_section *mov, const same_section *ref, const double *box, double radius, int k,
                      double dist_ct_coeff, int64_t *out_counts) {
    if (!w) return SAME_EINVAL;
    same_ctx *ctx = w->ctx;
    REQUIRE(ctx, mov && ref && box && out_counts && mov->ctx->device == ctx->device && ref->ctx->device == ctx->device);
    REQUIRE(ctx, mov->T == ref->T && mov->cost_f32 == ref->cost_f32 && k >= 1 && k <= SAME_MAX_KNN && radius >= 0.0);
    SAME_TRY(same_use(ctx));
    w->staged = w->finished = w->filtered = 0;
    w->has_type = mov->type_id != nullptr;
    w->cost_f32 = mov->cost_f32;
    w->k = k;
    w->n_ua = w->P = w->Tr = 0;
    for (int q = 0; q < 4; ++q) out_counts[q] = 0;
    SAME_TRY(subset_rows(w, mov, box, w->rows_m, &w->n_m));
    SAME_TRY(subset_rows(w, ref, box, w->rows_r, &w->n_r));
    const int64_t n_m = w->n_m, n_r = w->n_r;
    out_counts[0] = n_m;
    out_counts[1] = n_r;
    w->staged = 1;
    if (n_m == 0 || n_r == 0) return SAME_OK;      // no pairs: the caller raises what run_same raises (src/same.py:1003)
    REQUIRE(ctx, n_m * (int64_t)k < ((int64_t)1 << 31) - 1);   // pair offsets are 32-bit
    const int T = mov->T;
    const size_t cs = w->cost_f32 ? sizeof(float) : sizeof(double);
    const int32_t *rm = as<int32_t>(w->rows_m), *rr = as<int32_t>(w->rows_r);
    SAME_TRY(gather(ctx, mov->xy, 2 * sizeof(double), rm, n_m, w->axy_w));
    SAME_TRY(gather(ctx, ref->xy, 2 * sizeof(double), rr, n_r, w->rxy_w));
    SAME_TRY(gather(ctx, mov->size, sizeof(double), rm, n_m, w->size_w));
    if (w->has_type) SAME_TRY(gather(ctx, mov->type_id, sizeof(int32_t), rm, n_m, w->type_w));
    SAME_TRY(ensure(ctx, w->type_c, (size_t)n_m * sizeof(int32_t)));
    SAME_TRY(gather(ctx, mov->types_c, (size_t)T * cs, rm, n_m, w->A_w));
    SAME_TRY(gather(ctx, ref->types_c, (size_t)T * cs, rr, n_r, w->R_w));
    if (w->cost_f32) {
        SAME_TRY(gather(ctx, mov->xy_c, 2 * cs, rm, n_m, w->axyc_w));
        SAME_TRY(gather(ctx, ref->xy_c, 2 * cs, rr, n_r, w->rxyc_w));
    }
    const size_t slots = (size_t)n_m * k;
    SAME_TRY(ensure(ctx, w->idx, slots * sizeof(int32_t)));
    SAME_TRY(ensure(ctx, w->cnt, (size_t)n_m * sizeof(int32_t)));
    SAME_TRY(ensure(ctx, w->cost, slots * cs));
    SAME_TRY(same_knn_prune_dev(ctx, as<double>(w->axy_w), as<double>(w->rxy_w), n_r, 0, n_m, radius, k, as<int32_t>(w->idx), nullptr,
                                as<int32_t>(w->cnt)));
    if (w->cost_f32)
        SAME_TRY(same_padded_cost_f32_dev(ctx, as<float>(w->A_w), as<float>(w->R_w), T, as<float>(w->axyc_w), as<float>(w->rxyc_w), 0, n_m, k,
                                          as<int32_t>(w->idx), (float)dist_ct_coeff, as<float>(w->cost)));
    else
        SAME_TRY(same_padded_cost_f64_dev(ctx, as<double>(w->A_w), as<double>(w->R_w), T, as<double>(w->axy_w), as<double>(w->rxy_w), 0, n_m, k,
                                          as<int32_t>(w->idx), dist_ct_coeff, as<double>(w->cost)));
    SAME_TRY(ensure(ctx, w->a_off, (size_t)n_m * sizeof(int32_t)));
    SAME_TRY(ensure(ctx, w->p_off, (size_t)n_m * sizeof(int32_t)));
    SAME_TRY(ensure(ctx, w->ua, (size_t)n_m * sizeof(int32_t)));
    SAME_TRY(ensure(ctx, w->rows_ua, (size_t)n_m * sizeof(int32_t)));
    SAME_TRY(ensure(ctx, w->axy_c, (size_t)n_m * 2 * sizeof(double)));
    SAME_TRY(ensure(ctx, w->size_c, (size_t)n_m * sizeof(double)));
    SAME_TRY(ensure(ctx, w->pairs, slots * 2 * sizeof(int32_t)));
    SAME_TRY(ensure(ctx, w->cost64, slots * sizeof(double)));
    unsigned long long *dc = as<unsigned long long>(w->counts);
    hipLaunchKernelGGL(window_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, as<int32_t>(w->cnt), n_m, as<int32_t>(w->a_off),
                       as<int32_t>(w->p_off), dc);
    if (w->cost_f32)
        hipLaunchKernelGGL(window_scatter_kernel<float>, dim3(grid_for(n_m)), dim3(256), 0, ctx->stream, as<int32_t>(w->idx), as<float>(w->cost),
                           as<int32_t>(w->cnt), n_m, k, as<int32_t>(w->a_off), as<int32_t>(w->p_off), rm, as<double>(w->axy_w),
                           as<double>(w->size_w), w->has_type ? as<int32_t>(w->type_w) : nullptr, as<int32_t>(w->ua), as<int32_t>(w->rows_ua),
                           as<double>(w->axy_c), as<double>(w->size_c), as<int32_t>(w->type_c), as<int32_t>(w->pairs), as<double>(w->cost64));
    else
        hipLaunchKernelGGL(window_scatter_kernel<double>, dim3(grid_for(n_m)), dim3(256), 0, ctx->stream, as<int32_t>(w->idx), as<double>(w->cost),
                           as<int32_t>(w->cnt), n_m, k, as<int32_t>(w->a_off), as<int32_t>(w->p_off), rm, as<double>(w->axy_w),
                           as<double>(w->size_w), w->has_type ? as<int32_t>(w->type_w) : nullptr, as<int32_t>(w->ua), as<int32_t>(w->rows_ua),
                           as<double>(w->axy_c), as<double>(w->size_c), as<int32_t>(w->type_c), as<int32_t>(w->pairs), as<double>(w->cost64));
    HIP_TRY(ctx, hipGetLastError());
    // one read-back: the two totals, then the kept aligned rows and their XY at the capacity n_m (n_ua <= n_m is not known yet)
    const size_t head = 64;
    SAME_TRY(ensure_host(w, host_need(n_m)));
    char *h = static_cast<char *>(w->host);
    HIP_TRY(ctx, hipMemcpyAsync(h, dc + 2, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(h + head, w->axy_c.p, (size_t)n_m * 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(h + head + (size_t)n_m * 2 * sizeof(double), w->rows_ua.p, (size_t)n_m * sizeof(int32_t), hipMemcpyDeviceToHost,
                                ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const unsigned long long *tot = reinterpret_cast<const unsigned long long *>(h);
    w->n_ua = (int64_t)tot[0];
    w->P = (int64_t)tot[1];
    out_counts[2] = w->n_ua;
    out_counts[3] = w->P;
    w->staged = 2;
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
    case SAME_WINDOW_ALIGNED_ROWS: want = n_ua * 4; host = h + 64 + (size_t)n_m * 16; REQUIRE(ctx, full || n_ua == 0); break;
    case SAME_WINDOW_ROWS_M: want = n_m * 4; dev = w->rows_m.p; break;
    case SAME_WINDOW_ROWS_R: want = n_r * 4; dev = w->rows_r.p; break;
    case SAME_WINDOW_PAIRS: want = P * 8; dev = w->pairs.p; break;
    case SAME_WINDOW_COSTS: want = P * 8; dev = w->cost64.p; break;
    case SAME_WINDOW_KEPT: want = n_ua * 4; dev = w->ua.p; break;
    case SAME_WINDOW_SIGNS: want = Tr; dev = w->sign.p; REQUIRE(ctx, w->finished); break;
    case SAME_WINDOW_WEIGHTS: want = Tr * 8; dev = w->weight.p; REQUIRE(ctx, w->finished); break;
    case SAME_WINDOW_MATCH: want = n_ua * 4; dev = w->match.p; REQUIRE(ctx, w->finished); break;
    case SAME_WINDOW_TRIANGLES: want = Tr * 12; dev = w->tris.p; REQUIRE(ctx, w->finished || w->filtered); break;
    default: REQUIRE(ctx, !"unknown same_window_fetch selector");
    }
    REQUIRE(ctx, bytes == want);
    if (want == 0) return SAME_OK;
    if (host) {                                   // already on the host since the stage call's own read-back
        memcpy(out, host, (size_t)want);
        return SAME_OK;
    }
    HIP_TRY(ctx, hipMemcpyAsync(out, dev, (size_t)want, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SAME_OK;
}

int same_window_filter(same_window *w, const int32_t *simplices, int64_t n_simplices, double radius, int angle_enabled, double cos_thr,
                       double near_tol, int ignore_same_type, int ensure_min_triangle_per_node, int64_t *out_counts) {
    if (!w) return SAME_EINVAL;
    same_ctx *ctx = w->ctx;
    REQUIRE(ctx, w->staged == 2 && out_counts && n_simplices >= 0 && n_simplices < ((int64_t)1 << 31) - 512 && (n_simplices == 0 || simplices));
    const int64_t n = w->n_ua, Tr = n_simplices;
    for (int q = 0; q < 3; ++q) out_counts[q] = 0;
    SAME_TRY(same_use(ctx));
    SAME_TRY(check_index_range(ctx, simplices, Tr * 3, 0, n, "triangles"));
    w->filtered = w->finished = 0;
    w->Tr = 0;
    if (Tr == 0 || n == 0) { w->filtered = 1; return SAME_OK; }
    const bool use_type = ignore_same_type && w->has_type;
    const int64_t t_words = (int64_t)grid_for(Tr) * 4, n_words = (int64_t)grid_for(n) * 4;
    SAME_TRY(ensure(ctx, w->raw, (size_t)Tr * 3 * sizeof(int32_t)));
    SAME_TRY(ensure(ctx, w->tris, (size_t)Tr * 3 * sizeof(int32_t)));
    SAME_TRY(ensure(ctx, w->cls, (size_t)Tr));
    SAME_TRY(ensure(ctx, w->perim, (size_t)Tr * sizeof(double)));
    SAME_TRY(ensure(ctx, w->maxcos, (size_t)Tr * sizeof(double)));
    SAME_TRY(ensure(ctx, w->kmask, (size_t)t_words * sizeof(unsigned long long)));
    SAME_TRY(ensure(ctx, w->klist, (size_t)Tr * sizeof(int32_t)));
    SAME_TRY(ensure(ctx, w->first_v, (size_t)Tr * sizeof(unsigned)));
    SAME_TRY(ensure(ctx, w->has_kept, (size_t)n));
    SAME_TRY(ensure(ctx, w->any_valid, (size_t)n));
    SAME_TRY(ensure(ctx, w->best_p, (size_t)n * sizeof(unsigned long long)));
    SAME_TRY(ensure(ctx, w->best_t, (size_t)n * sizeof(unsigned)));
    SAME_TRY(ensure(ctx, w->nmask, (size_t)n_words * sizeof(unsigned long long)));
    SAME_TRY(ensure(ctx, w->nlist, (size_t)n * sizeof(int32_t)));
    unsigned long long *dc = as<unsigned long long>(w->counts);   // [1] kept (class 0), [2] near the threshold, [5] added back
    HIP_TRY(ctx, hipMemsetAsync(dc, 0, 8 * sizeof(unsigned long long), ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(w->raw.p, simplices, (size_t)Tr * 3 * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    SAME_TRY(same_tri_classify_dev(ctx, as<double>(w->axy_c), as<int32_t>(w->raw), Tr, radius, angle_enabled, cos_thr,
                                   use_type ? as<int32_t>(w->type_c) : nullptr, as<uint8_t>(w->cls), as<double>(w->perim), as<double>(w->maxcos)));
    HIP_TRY(ctx, hipMemsetAsync(w->has_kept.p, 0, (size_t)n, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(w->any_valid.p, 0, (size_t)n, ctx->stream));
    const int near_enabled = angle_enabled && cos_thr == cos_thr && cos_thr - cos_thr == 0.0;       // a finite threshold
    hipLaunchKernelGGL(filter_mark_kernel, dim3(grid_for(Tr)), dim3(256), 0, ctx->stream, as<uint8_t>(w->cls), as<double>(w->maxcos),
                       as<int32_t>(w->raw), Tr, near_enabled, cos_thr, near_tol, as<uint8_t>(w->has_kept), as<uint8_t>(w->any_valid),
                       as<unsigned long long>(w->kmask), dc);
    HIP_TRY(ctx, hipGetLastError());
    SAME_TRY(same_compact_mask_core(ctx, as<unsigned long long>(w->kmask), t_words, Tr, as<int32_t>(w->klist), dc));          // count -> dc[1]
    const bool readd = use_type && ensure_min_triangle_per_node;
    if (readd) {
        HIP_TRY(ctx, hipMemsetAsync(w->best_p.p, 0xFF, (size_t)n * sizeof(unsigned long long), ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(w->best_t.p, 0xFF, (size_t)n * sizeof(unsigned), ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(w->first_v.p, 0xFF, (size_t)Tr * sizeof(unsigned), ctx->stream));
        hipLaunchKernelGGL(filter_best_perim_kernel, dim3(grid_for(Tr)), dim3(256), 0, ctx->stream, as<uint8_t>(w->cls), as<double>(w->perim),
                           as<int32_t>(w->raw), Tr, as<unsigned long long>(w->best_p));
        hipLaunchKernelGGL(filter_best_tri_kernel, dim3(grid_for(Tr)), dim3(256), 0, ctx->stream, as<uint8_t>(w->cls), as<double>(w->perim),
                           as<int32_t>(w->raw), Tr, as<unsigned long long>(w->best_p), as<unsigned>(w->best_t));
        hipLaunchKernelGGL(filter_first_node_kernel, dim3(grid_for(n)), dim3(256), 0, ctx->stream, as<uint8_t>(w->has_kept),
                           as<uint8_t>(w->any_valid), as<unsigned>(w->best_t), n, as<unsigned>(w->first_v));
        hipLaunchKernelGGL(filter_owner_mask_kernel, dim3(grid_for(n)), dim3(256), 0, ctx->stream, as<uint8_t>(w->has_kept),
                           as<uint8_t>(w->any_valid), as<unsigned>(w->best_t), as<unsigned>(w->first_v), n, as<unsigned long long>(w->nmask));
        HIP_TRY(ctx, hipGetLastError());
        SAME_TRY(same_compact_mask_core(ctx, as<unsigned long long>(w->nmask), n_words, n, as<int32_t>(w->nlist), dc + 4));   // count -> dc[5]
    }
    unsigned long long *h = static_cast<unsigned long long *>(w->host);   // bytes [0, 64) of the staging block are scalars
    HIP_TRY(ctx, hipMemcpyAsync(h, dc, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const int64_t n_keep = (int64_t)h[1], n_near = (int64_t)h[2], n_add = readd ? (int64_t)h[5] : 0;
    out_counts[0] = n_keep;
    out_counts[1] = n_add;
    out_counts[2] = n_near;
    if (n_near) return SAME_OK;                      // knife-edge cosines: the caller decides them as the reference does and passes the triangles in
    if (n_keep + n_add) {
        hipLaunchKernelGGL(filter_emit_kernel, dim3(grid_for(n_keep + n_add)), dim3(256), 0, ctx->stream, as<int32_t>(w->raw),
                           as<int32_t>(w->klist), n_keep, as<int32_t>(w->nlist), n_add, as<unsigned>(w->best_t), as<int32_t>(w->tris));
        HIP_TRY(ctx, hipGetLastError());
    }
    w->Tr = n_keep + n_add;
    w->filtered = 1;
    return SAME_OK;
}

int same_window_finish(same_window *w, const int32_t *tris, int64_t Tr, double no_match_penalty, int32_t *out_match_row,
                       uint8_t *out_point_flag, int64_t *out_stats) {
    if (!w) return SAME_EINVAL;
    same_ctx *ctx = w->ctx;
    const bool resident = tris == nullptr && Tr < 0;      // the triangles same_window_filter left on the device
    REQUIRE(ctx, w->staged == 2 && out_stats);
    if (resident) {
        REQUIRE(ctx, w->filtered);
        Tr = w->Tr;
    }
    REQUIRE(ctx, Tr >= 0 && Tr < ((int64_t)1 << 31) - 512 && (Tr == 0 || tris || resident));
    const int64_t n = w->n_ua, P = w->P;
    REQUIRE(ctx, n == 0 || (out_match_row && out_point_flag));
    for (int q = 0; q < 8; ++q) out_stats[q] = 0;
    SAME_TRY(same_use(ctx));
    if (!resident) SAME_TRY(check_index_range(ctx, tris, Tr * 3, 0, n, "triangles"));
    w->Tr = Tr;
    w->finished = 0;
    if (n == 0) { w->finished = 1; return SAME_OK; }
    const size_t tt = (size_t)std::max<int64_t>(Tr, 1), padded = (size_t)grid_for(Tr) * 256;
    SAME_TRY(ensure(ctx, w->tris, tt * 3 * sizeof(int32_t)));
    SAME_TRY(ensure(ctx, w->sign, tt));
    SAME_TRY(ensure(ctx, w->weight, tt * sizeof(double)));
    SAME_TRY(ensure(ctx, w->rowmin, (size_t)n * sizeof(double)));
    SAME_TRY(ensure(ctx, w->prefer, (size_t)n));
    SAME_TRY(ensure(ctx, w->pair_of_row, (size_t)n * sizeof(int32_t)));
    SAME_TRY(ensure(ctx, w->match, (size_t)n * sizeof(int32_t)));
    SAME_TRY(ensure(ctx, w->match_row, (size_t)n * sizeof(int32_t)));
    SAME_TRY(ensure(ctx, w->oflag, padded + 256));
    SAME_TRY(ensure(ctx, w->omask, ((size_t)grid_for(Tr) * 4 + 4) * sizeof(unsigned long long)));
    SAME_TRY(ensure(ctx, w->edge, tt * 3));
    SAME_TRY(ensure(ctx, w->tflag, tt));
    SAME_TRY(ensure(ctx, w->pflag, (size_t)n));
    SAME_TRY(ensure(ctx, w->before, tt * sizeof(double)));
    SAME_TRY(ensure(ctx, w->after, tt * sizeof(double)));
    SAME_TRY(ensure(ctx, w->m3, tt * 3));
    SAME_TRY(ensure(ctx, w->flipped, tt));
    unsigned long long *dc = as<unsigned long long>(w->counts);      // [0..1] orientation, [4..6] XY-order, [8..15] the stats block
    unsigned long long *dstats = dc + 8;
    HIP_TRY(ctx, hipMemsetAsync(dc, 0, 16 * sizeof(unsigned long long), ctx->stream));
    if (Tr) {
        if (!resident) HIP_TRY(ctx, hipMemcpyAsync(w->tris.p, tris, (size_t)Tr * 3 * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
        SAME_TRY(same_tri_sign_weight_dev(ctx, as<double>(w->axy_c), as<double>(w->size_c), as<int32_t>(w->tris), Tr, as<int8_t>(w->sign),
                                          as<double>(w->weight)));
    }
    // greedy MIP start: per-row minimum, rows that beat their penalty, the scan's matching (one pair per aligned row)
    SAME_TRY(same_pair_rowmin_core(ctx, as<int32_t>(w->pairs), as<double>(w->cost64), P, n, as<double>(w->rowmin)));
    hipLaunchKernelGGL(prefer_kernel, dim3(grid_for(n)), dim3(256), 0, ctx->stream, as<double>(w->rowmin), as<double>(w->size_c), n,
                       no_match_penalty, as<uint8_t>(w->prefer));
    HIP_TRY(ctx, hipGetLastError());
    int rounds = 0;
    SAME_TRY(same_greedy_core(ctx, as<int32_t>(w->pairs), as<double>(w->cost64), P, n, w->n_r, as<uint8_t>(w->prefer),
                              as<int32_t>(w->pair_of_row), &rounds));
    hipLaunchKernelGGL(match_rows_kernel, dim3(grid_for(n)), dim3(256), 0, ctx->stream, as<int32_t>(w->pair_of_row), as<int32_t>(w->pairs),
                       as<int32_t>(w->rows_r), n, as<int32_t>(w->match), as<int32_t>(w->match_row), dstats);
    HIP_TRY(ctx, hipGetLastError());
    // the three sweeps under that incumbent
    SAME_TRY(same_orient_counts_core(ctx, as<int32_t>(w->tris), Tr, as<int8_t>(w->sign), as<double>(w->rxy_w), as<int32_t>(w->match),
                                     as<uint8_t>(w->oflag), as<unsigned long long>(w->omask), dc));
    SAME_TRY(same_xyorder_sweep_dev(ctx, as<double>(w->axy_c), n, as<double>(w->rxy_w), as<int32_t>(w->tris), Tr, as<int32_t>(w->match),
                                    as<uint8_t>(w->edge), as<uint8_t>(w->tflag), as<uint8_t>(w->pflag), reinterpret_cast<uint64_t *>(dc + 4)));
    if (Tr) {
        SAME_TRY(same_area_flip_dev(ctx, as<double>(w->axy_c), as<double>(w->rxy_w), as<int32_t>(w->tris), Tr, as<int32_t>(w->match),
                                    as<double>(w->before), as<double>(w->after), as<uint8_t>(w->m3), as<uint8_t>(w->flipped)));
        hipLaunchKernelGGL(count_flags_kernel, dim3(grid_for(Tr)), dim3(256), 0, ctx->stream, as<uint8_t>(w->flipped), Tr, dstats + 5);
        HIP_TRY(ctx, hipGetLastError());
    }
    const size_t head = 128;
    char *h = static_cast<char *>(w->host) + finish_off(w->n_m);     // behind the stage call's results, which stay valid
    HIP_TRY(ctx, hipMemcpyAsync(h, dc, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(h + head, w->match_row.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(h + head + (size_t)n * sizeof(int32_t), w->pflag.p, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const unsigned long long *c = reinterpret_cast<const unsigned long long *>(h);
    out_stats[0] = (int64_t)c[0];        // orientation: triangles checked
    out_stats[1] = (int64_t)c[1];        //              flipped
    out_stats[2] = (int64_t)c[4];        // XY-order: comparisons
    out_stats[3] = (int64_t)c[5];        //           violations
    out_stats[4] = (int64_t)c[6];        //           triangles with a violation
    out_stats[5] = (int64_t)c[8 + 5];    // area flips
    out_stats[6] = rounds;               // rounds of the greedy rule
    out_stats[7] = (int64_t)c[8 + 7];    // matched aligned cells
    memcpy(out_match_row, h + head, (size_t)n * sizeof(int32_t));
    memcpy(out_point_flag, h + head + (size_t)n * sizeof(int32_t), (size_t)n);
    w->finished = 1;
    return SAME_OK;
}

}  // extern "C"
