// window_finish.hip -- same_window_filter_finish: per window the Delaunay simplices in; triangle classes (src/helpers.py:300-330),
// the keep list and the same-type triangles added back so that every node keeps one (src/helpers.py:331-340, :365-389) -- all on the
// device, in the reference's order (a cosine within 8 ulp of the angle threshold is left to the host, which re-decides it with the
// reference's literal arccos and calls again with prefiltered = 1: the kept triangles in, no filter); source signs / weights
// (src/same.py:1128-1146), per-row minimum and the greedy MIP start (src/init_helpers.py:104-133), the lazy-constraint body under
// that incumbent (src/same.py:645-669), XY-order sweep (src/violationhelper.py:53-117), signed-area flips (src/same.py:1362-1402).
// Back to the host: the matched reference row per kept aligned cell, the per-cell violation flags and eight counters.
#include "window_internal.h"

namespace {

using namespace devmath;
using namespace win;
using scan::Pair;

// ---- triangle filter (src/helpers.py:233-395) on the device ------------------------------------------------------------------
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
    // ORDER TIES (FC_TIES, SC_TIES): places where the reference's answer depends on the ORDER Qhull lists triangles or their corners
    // in, not only on which triangles there are -- counted so that a caller whose triangulation is not Qhull's (same_delaunay2d) knows
    // when to ask Qhull after all.  Here: a node that asks for its best same-type triangle and has a second one whose perimeter is
    // within the rounding of a three-term sum (the sum's order is the corners' order; equal perimeters go to the first triangle)
    if (w.best_p && q < w.Tr && w.cls[q] == 3) {
        bool tie = false;
        for (int c = 0; c < 3; ++c) {
            const int32_t v = w.raw[3 * q + c];
            if (!asks(w.has_kept, w.any_valid, w.best_t, v)) continue;
            const int64_t best = (int64_t)~w.best_t[v];
            tie |= best != q && fabs(w.perim[q] - w.perim[best]) <= 1.8e-15 * w.perim[best];
        }
        if (tie) atomicAdd(&w.counters[FC_TIES], 1ull);
    }
    if (q >= n_keep + n_add) return;
    const int64_t t = q < n_keep ? w.klist[q] : (int64_t)~w.best_t[w.nlist[q - n_keep]];
    w.out[3 * q] = w.raw[3 * t];
    w.out[3 * q + 1] = w.raw[3 * t + 1];
    w.out[3 * q + 2] = w.raw[3 * t + 2];
}

// ---- incumbent and sweeps ---------------------------------------------------------------------------------------------------

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
    int checked = 0, flipped = 0, ncmp = 0, nviol = 0, tv = 0, aflip = 0, ties = 0;
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
                // equal coordinates at an edge's ends: `<` one way round is not `<` the other way round (order tie, see filter_emit_kernel)
                ties |= a[p].x == a[q].x || a[p].y == a[q].y || r[p].x == r[q].x || r[p].y == r[q].y;
                nviol += ((e2 >> 1) & 1) + ((e2 >> 2) & 1);
                if (e2) { tv = 1; cell_flag_or(pflag, v[p], 1u); cell_flag_or(pflag, v[q], 1u); }
            }
        }
        if (all3) {
            const double bf = signed_area(a[0], a[1], a[2]), af = signed_area(r[0], r[1], r[2]);
            aflip = bf * af < 0.0;                                         // src/same.py:1401
            // a sign within rounding of zero may come out the other way with the corners in another order (order tie)
            ties |= sign_in_doubt(a[0], a[1], a[2]) || sign_in_doubt(r[0], r[1], r[2]);
            if (aflip)
                for (int q = 0; q < 3; ++q) cell_flag_or(pflag, v[q], 2u);
        }
    }
    int vals[7] = {checked, flipped, ncmp, nviol, tv, aflip, ties};
#pragma unroll
    for (int q = 0; q < 7; ++q)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) vals[q] += __shfl_down(vals[q], off, 64);
    __shared__ int part[4][7];
    if ((threadIdx.x & 63) == 0)
        for (int q = 0; q < 7; ++q) part[threadIdx.x >> 6][q] = vals[q];
    __syncthreads();
    if (threadIdx.x < 7) {      // one atomic per block per counter (integer sums: order-independent); SC_CHECKED .. SC_AFLIP, then SC_TIES
        const int s = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
        if (s) atomicAdd(&counters[threadIdx.x < 6 ? threadIdx.x : SC_TIES], (unsigned long long)s);
    }
}

// ---- the filter and the finish as enqueue-only halves + their read-backs, so that the two calls can also run as one ----------
struct FilterPlan {
    FilterArgs args{};
    void *zero = nullptr;                     // head of the filter buffer, zeroed with the group's
    size_t zero_bytes = 0;
    unsigned long long *counters = nullptr;   // FC_*
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
    w->match_row = p->match_row;
    w->pflag = p->pflag;
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

int read_finish(same_window *w, FinishPlan *p, int32_t *out_match_row, uint8_t *out_point_flag, int64_t *out_stats, int64_t *out_ties) {
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
    *out_ties += (int64_t)cnt[SC_TIES];
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
    for (int q = 0; q < 4 * n_windows; ++q) out_counts[q] = 0;
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
                                         {it.plan.back_bytes, it.filtered ? FC_COPIED * sizeof(unsigned long long) : 0}};
                } else {
                    if (it.filtered) {
                        unsigned long long *hf = reinterpret_cast<unsigned long long *>(static_cast<char *>(w->host) + w->host_filter_off);
                        hipError_t e = hipMemcpyAsync(hf, it.fplan.counters, FC_COPIED * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream);
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
        int64_t *counts = out_counts + 4 * i, *stats = out_stats + 8 * i;
        if (!it.enqueued) {                 // no kept aligned cell: nothing to match, nothing to sweep
            w->filtered = w->finished = 1;
            continue;
        }
        int32_t *const match_out = out_match_row + cell0;
        uint8_t *const flag_out = out_point_flag + cell0;
        cell0 += w->n_ua;
        if (it.filtered) {
            // the filter's counters first (they are on the host since the batch's one wait): a window with a cosine at the threshold is
            // redone by the caller with prefiltered = 1, so nothing of it is read back here -- in particular no further greedy rounds
            // (fills, launches and a wait per batch of rounds) are spent on a matching that is thrown away
            const unsigned long long *hf = reinterpret_cast<const unsigned long long *>(static_cast<const char *>(w->host) + w->host_filter_off);
            const int64_t n_keep = (int64_t)hf[FC_KEEP], n_near = (int64_t)hf[FC_NEAR], n_add = it.fplan.readd ? (int64_t)hf[FC_ADD] : 0;
            counts[0] = n_keep;
            counts[1] = n_add;
            counts[2] = n_near;
            counts[3] = (int64_t)hf[FC_TIES];
            if (n_near) {
                for (int q = 0; q < 8; ++q) stats[q] = 0;
                continue;
            }
            w->Tr = n_keep + n_add;
        } else {
            counts[0] = Tr;
            w->Tr = Tr;
        }
        SAME_TRY(read_finish(w, &it.plan, match_out, flag_out, stats, counts + 3));
        w->filtered = w->finished = 1;
    }
    return SAME_OK;
}

}  // extern "C"
