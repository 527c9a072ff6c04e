// match.hip -- SURVEY 8(f1): the greedy MIP start of src/init_helpers.py:109-133 resolved on the
// device, and the node-local flip statistics of eval_utils.check_triangle_violations
// (src/eval_utils.py:66-223).
//
// Greedy.  The reference sorts all candidate pairs by cost (stable) and scans them once, taking
// (i, j) when both ends are still free.  That sequential scan is equivalent to repeatedly taking
// every pair that is the minimum -- under the same total order (cost, pair index) -- at BOTH of
// its endpoints among the pairs still alive: such pairs cannot be pre-empted by anything the
// scan would visit earlier, and the globally smallest alive pair always qualifies, so the rounds
// terminate with exactly the scan's matching.  No sort is needed: each round is three maps over
// the pairs with 64-bit atomicMin on a monotone key of the cost, then 32-bit atomicMin on the
// pair index to break exact cost ties the way the stable sort does.
#include "common.h"

#include <algorithm>

namespace {

__device__ __forceinline__ unsigned long long cost_key(double v) {  // monotone u64 key of a double
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

__global__ __launch_bounds__(256) void greedy_init_kernel(const int32_t *__restrict__ pairs, int64_t P,
                                                           const uint8_t *__restrict__ prefer, uint8_t *__restrict__ alive) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < P) alive[p] = prefer[pairs[2 * p]];  // rows that prefer "unmatched" never enter (init_helpers.py:122,128)
}

// One round = three maps over the pairs.  The per-endpoint minima are kept INVERTED (atomicMax of ~key, ~pair index) so that an
// all-zero word means "nothing yet": the state starts from one memset, and the select map of round r clears the set round r + 1
// uses (two sets, alternating), so a round is three launches and no fill.  `key` / `idx` hold rows [0, n_m) then columns
// [n_m, n_m + n_r); dP (if not null) is the pair count on the device, P then only the bound the launch was sized by.
__device__ __forceinline__ unsigned long long inv_key(double c) {
    const unsigned long long k = ~cost_key(c);
    return k ? k : 1ull;   // ~key == 0 only for one NaN payload: keep 0 for "nothing yet"
}

// The three (two) maps of a round as device functions of the pair index p: the single-problem kernels and the batched ones (several
// independent problems per launch: the windows of a batch call, blockIdx.y = problem) are both thin wrappers.
__device__ __forceinline__ void min_key_body(const int32_t *__restrict__ pairs, const double *__restrict__ costs, int64_t P,
                                             uint8_t *__restrict__ alive, const uint8_t *__restrict__ used, int64_t n_m,
                                             unsigned long long *__restrict__ key, int64_t p) {
    if (p >= P || !alive[p]) return;
    const int32_t i = pairs[2 * p], j = pairs[2 * p + 1];
    if (used[i] || used[n_m + j]) { alive[p] = 0; return; }
    const unsigned long long k = inv_key(costs[p]);
    atomicMax(&key[i], k);
    atomicMax(&key[n_m + j], k);
}
__device__ __forceinline__ void min_idx_body(const int32_t *__restrict__ pairs, const double *__restrict__ costs, int64_t P,
                                             const uint8_t *__restrict__ alive, int64_t n_m, const unsigned long long *__restrict__ key,
                                             unsigned *__restrict__ idx, int64_t p) {
    if (p >= P || !alive[p]) return;
    const int32_t i = pairs[2 * p], j = pairs[2 * p + 1];
    const unsigned long long k = inv_key(costs[p]);
    if (k == key[i]) atomicMax(&idx[i], ~(unsigned)p);
    if (k == key[n_m + j]) atomicMax(&idx[n_m + j], ~(unsigned)p);
}
// only "did this round take anything" is ever asked: one plain store per block that took something (a same-address atomicAdd per wave
// serialised ~1 800 atomics at one L2 line and was most of the select kernel's time).  Every thread of the block calls this.
__device__ __forceinline__ void publish_selected(bool sel, unsigned long long *__restrict__ n_selected) {
    __shared__ int any_sel;
    if (threadIdx.x == 0) any_sel = 0;
    __syncthreads();
    if (__ballot(sel) && (threadIdx.x & 63) == 0) any_sel = 1;
    __syncthreads();
    if (threadIdx.x == 0 && any_sel) *n_selected = 1ull;
}
__device__ __forceinline__ void select_body(const int32_t *__restrict__ pairs, int64_t P, uint8_t *__restrict__ alive, int64_t n_m, int64_t n_ends,
                                            const unsigned *__restrict__ idx, uint8_t *__restrict__ used, int32_t *__restrict__ match_pair,
                                            unsigned long long *__restrict__ n_selected, unsigned long long *__restrict__ next_key,
                                            unsigned *__restrict__ next_idx, int64_t p) {
    bool sel = false;
    if (p < P && alive[p]) {
        const int32_t i = pairs[2 * p], j = pairs[2 * p + 1];
        if (idx[i] == ~(unsigned)p && idx[n_m + j] == ~(unsigned)p) {
            sel = true;
            alive[p] = 0;
            used[i] = 1;
            used[n_m + j] = 1;
            match_pair[i] = (int32_t)p;
        }
    }
    if (p < n_ends) { next_key[p] = 0ull; next_idx[p] = 0u; }   // the other set: nobody reads it in this round
    publish_selected(sel, n_selected);
}

__global__ __launch_bounds__(256) void greedy_min_key_kernel(
    const int32_t *__restrict__ pairs, const double *__restrict__ costs, int64_t P, const unsigned long long *__restrict__ dP,
    uint8_t *__restrict__ alive, const uint8_t *__restrict__ used, int64_t n_m, unsigned long long *__restrict__ key) {
    if (dP) P = (int64_t)*dP;
    min_key_body(pairs, costs, P, alive, used, n_m, key, (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
}

__global__ __launch_bounds__(256) void greedy_min_idx_kernel(
    const int32_t *__restrict__ pairs, const double *__restrict__ costs, int64_t P, const unsigned long long *__restrict__ dP,
    const uint8_t *__restrict__ alive, int64_t n_m, const unsigned long long *__restrict__ key, unsigned *__restrict__ idx) {
    if (dP) P = (int64_t)*dP;
    min_idx_body(pairs, costs, P, alive, n_m, key, idx, (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
}

__global__ __launch_bounds__(256) void greedy_select_kernel(
    const int32_t *__restrict__ pairs, int64_t P, const unsigned long long *__restrict__ dP, uint8_t *__restrict__ alive,
    int64_t n_m, int64_t n_ends, const unsigned *__restrict__ idx, uint8_t *__restrict__ used, int32_t *__restrict__ match_pair,
    unsigned long long *__restrict__ n_selected, unsigned long long *__restrict__ next_key, unsigned *__restrict__ next_idx) {
    if (dP) P = (int64_t)*dP;
    select_body(pairs, P, alive, n_m, n_ends, idx, used, match_pair, n_selected, next_key, next_idx, (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
}

// Rounds for costs that are exactly floats (the fp32-cost windows, cost64 = (double)float): (cost, pair index) fits ONE 64-bit word
// -- float key in the high half, inverted pair index in the low half -- so a single atomicMax per end point finds the minimum cost
// AND the first pair among equal costs: a round is two launches instead of three.  Same order, same matching as the generic rounds.
__device__ __forceinline__ unsigned long long packed_key(double c, int64_t p) {
    const unsigned u = __float_as_uint((float)c);
    const unsigned k = (u >> 31) ? ~u : (u | 0x80000000u);      // monotone key of the float
    return ((unsigned long long)(~k) << 32) | (unsigned long long)(~(unsigned)p);   // larger word = smaller (cost, index); never 0: p < 2^32 - 1
}
__device__ __forceinline__ void min_packed_body(const int32_t *__restrict__ pairs, const double *__restrict__ costs, int64_t P,
                                                uint8_t *__restrict__ alive, const uint8_t *__restrict__ used, int64_t n_m,
                                                unsigned long long *__restrict__ key, int64_t p) {
    if (p >= P || !alive[p]) return;
    const int32_t i = pairs[2 * p], j = pairs[2 * p + 1];
    if (used[i] || used[n_m + j]) { alive[p] = 0; return; }
    const unsigned long long k = packed_key(costs[p], p);
    atomicMax(&key[i], k);
    atomicMax(&key[n_m + j], k);
}
__device__ __forceinline__ void select_packed_body(const int32_t *__restrict__ pairs, const double *__restrict__ costs, int64_t P,
                                                   uint8_t *__restrict__ alive, int64_t n_m, int64_t n_ends,
                                                   const unsigned long long *__restrict__ key, uint8_t *__restrict__ used,
                                                   int32_t *__restrict__ match_pair, unsigned long long *__restrict__ n_selected,
                                                   unsigned long long *__restrict__ next_key, int64_t p) {
    bool sel = false;
    if (p < P && alive[p]) {
        const int32_t i = pairs[2 * p], j = pairs[2 * p + 1];
        const unsigned long long k = packed_key(costs[p], p);
        if (key[i] == k && key[n_m + j] == k) {
            sel = true;
            alive[p] = 0;
            used[i] = 1;
            used[n_m + j] = 1;
            match_pair[i] = (int32_t)p;
        }
    }
    if (p < n_ends) next_key[p] = 0ull;
    publish_selected(sel, n_selected);
}
__global__ __launch_bounds__(256) void greedy_min_packed_kernel(
    const int32_t *__restrict__ pairs, const double *__restrict__ costs, int64_t P, const unsigned long long *__restrict__ dP,
    uint8_t *__restrict__ alive, const uint8_t *__restrict__ used, int64_t n_m, unsigned long long *__restrict__ key) {
    if (dP) P = (int64_t)*dP;
    min_packed_body(pairs, costs, P, alive, used, n_m, key, (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
}
__global__ __launch_bounds__(256) void greedy_select_packed_kernel(
    const int32_t *__restrict__ pairs, const double *__restrict__ costs, int64_t P, const unsigned long long *__restrict__ dP,
    uint8_t *__restrict__ alive, int64_t n_m, int64_t n_ends, const unsigned long long *__restrict__ key, uint8_t *__restrict__ used,
    int32_t *__restrict__ match_pair, unsigned long long *__restrict__ n_selected, unsigned long long *__restrict__ next_key) {
    if (dP) P = (int64_t)*dP;
    select_packed_body(pairs, costs, P, alive, n_m, n_ends, key, used, match_pair, n_selected, next_key, (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
}

// ---- the rounds of several independent problems in ONE launch per map (blockIdx.y = problem; the arguments travel by value) ----
struct GreedyArgs {
    const int32_t *pairs;
    const double *costs;
    int64_t P, n_m, n_ends;
    uint8_t *alive, *used;
    unsigned long long *key[2];
    unsigned *idx[2];
    unsigned long long *sel;
    int32_t *match_pair;
};
struct GreedyBatch {
    GreedyArgs w[SAME_LAUNCH_WINDOWS];
};
// kind: 0 min key, 1 min idx, 2 select, 3 min packed, 4 select packed; s = the key / idx set of this round, r = the round's slot in sel
template <int KIND>
__global__ __launch_bounds__(256) void greedy_batch_kernel(GreedyBatch b, int s, int r) {
    const GreedyArgs &a = b.w[blockIdx.y];
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t span = (KIND == 2 || KIND == 4) ? (a.P > a.n_ends ? a.P : a.n_ends) : a.P;
    if ((int64_t)blockIdx.x * blockDim.x >= span) return;      // whole block beyond this problem (uniform): the grid is sized by the largest
    if (KIND == 0) min_key_body(a.pairs, a.costs, a.P, a.alive, a.used, a.n_m, a.key[s], p);
    if (KIND == 1) min_idx_body(a.pairs, a.costs, a.P, a.alive, a.n_m, a.key[s], a.idx[s], p);
    if (KIND == 2) select_body(a.pairs, a.P, a.alive, a.n_m, a.n_ends, a.idx[s], a.used, a.match_pair, a.sel + r, a.key[s ^ 1], a.idx[s ^ 1], p);
    if (KIND == 3) min_packed_body(a.pairs, a.costs, a.P, a.alive, a.used, a.n_m, a.key[s], p);
    if (KIND == 4) select_packed_body(a.pairs, a.costs, a.P, a.alive, a.n_m, a.n_ends, a.key[s], a.used, a.match_pair, a.sel + r, a.key[s ^ 1], p);
}

// ---- node-local flip statistics (src/eval_utils.py:66-223) -------------------------------------
typedef double double2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2_t ld2(const double *xy, int64_t i) { return *reinterpret_cast<const double2_t *>(xy + 2 * i); }
__device__ __forceinline__ double area3(double2_t p1, double2_t p2, double2_t p3) {  // _signed_area, eval_utils.py:116-121
    return 0.5 * (p1.x * (p2.y - p3.y) + p2.x * (p3.y - p1.y) + p3.x * (p1.y - p2.y));
}

// tri_flag: bit0 all three matched, bit1 same type (only when type_id given), bit2 flipped
__global__ __launch_bounds__(256) void tri_flip_stats_kernel(
    const double *__restrict__ axy, const double *__restrict__ mxy, const uint8_t *__restrict__ matched,
    const int32_t *__restrict__ type_id, const int32_t *__restrict__ tris, int64_t Tr, uint8_t *__restrict__ tri_flag,
    unsigned *__restrict__ node_tri, unsigned *__restrict__ node_flip) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= Tr) return;
    const int32_t a = tris[3 * t], b = tris[3 * t + 1], c = tris[3 * t + 2];
    uint8_t f = 0;
    if (matched[a] && matched[b] && matched[c]) {
        f = 1;
        const bool same = type_id && type_id[a] == type_id[b] && type_id[b] == type_id[c];
        if (same) f |= 2;
        const double sb = area3(ld2(axy, a), ld2(axy, b), ld2(axy, c));
        const double sa = area3(ld2(mxy, a), ld2(mxy, b), ld2(mxy, c));
        // np.sign semantics incl. NaN (a NaN sign compares unequal to everything, itself included): eval_utils.py:160-166
        const int s0 = sb != sb ? 2 : (sb > 0.0) - (sb < 0.0), s1 = sa != sa ? 3 : (sa > 0.0) - (sa < 0.0);
        const bool flipped = s0 != s1 && s0 != 0 && s1 != 0;
        if (flipped) f |= 4;
        if (!same) {  // integer counters: order-independent
            atomicAdd(&node_tri[a], 1u); atomicAdd(&node_tri[b], 1u); atomicAdd(&node_tri[c], 1u);
            if (flipped) { atomicAdd(&node_flip[a], 1u); atomicAdd(&node_flip[b], 1u); atomicAdd(&node_flip[c], 1u); }
        }
    }
    tri_flag[t] = f;
}


// ---- metacell collapse (metacell_utils.greedy_triangle_collapse) ----------------
// Per triangle of the current metacell triangulation (src/metacell_utils.py:233-260, :393-431):
//   valid     : no edge > r_max (strict, unlike a7) and no corner angle < min_angle_deg
//   candidate : valid, all three cell types equal, size[a]+size[b]+size[c] <= max_size
//   perimeter : |ab| + |bc| + |ca| with 1-D np.linalg.norm (= sqrt(ddot) = sqrt(fma(y,y,x*x)))
// The angle rule is a cosine threshold as in tri.hip.  compute_angle here has no zero-length guard
// (:236-238): a zero side gives cos = 0/0 = NaN; Python's min(angle1, angle2, angle3) keeps a NaN
// that comes first and skips later ones, which is reproduced literally.
__device__ __forceinline__ double nrm2(double x, double y) { return __builtin_sqrt(__builtin_fma(y, y, x * x)); }
__device__ __forceinline__ double corner_cos_raw(double2_t p1, double2_t p2, double2_t p3) {  // corner at p2
    const double v1x = p1.x - p2.x, v1y = p1.y - p2.y, v2x = p3.x - p2.x, v2y = p3.y - p2.y;
    double c = __builtin_fma(v1y, v2y, v1x * v2x) / (nrm2(v1x, v1y) * nrm2(v2x, v2y));
    c = c < -1.0 ? -1.0 : c;  // np.clip: NaN stays NaN, +-inf clip to +-1
    c = c > 1.0 ? 1.0 : c;
    return c;
}

__global__ __launch_bounds__(256) void collapse_candidates_kernel(
    const double *__restrict__ xy, const int32_t *__restrict__ tris, int64_t Tr, int rmax_enabled, double r_max,
    int angle_enabled, double cos_thr, const int32_t *__restrict__ type_id, const double *__restrict__ size, double max_size,
    uint8_t *__restrict__ out_flag, double *__restrict__ out_perim, double *__restrict__ out_total) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= Tr) return;
    const int32_t a = tris[3 * t], b = tris[3 * t + 1], c = tris[3 * t + 2];
    const double2_t p1 = ld2(xy, a), p2 = ld2(xy, b), p3 = ld2(xy, c);
    bool valid = true;
    if (rmax_enabled) {  // is_triangle_valid, :247-252
        const double e1 = nrm2(p2.x - p1.x, p2.y - p1.y), e2 = nrm2(p3.x - p2.x, p3.y - p2.y), e3 = nrm2(p1.x - p3.x, p1.y - p3.y);
        double mx = e1 > e2 ? e1 : e2;
        mx = mx > e3 ? mx : e3;
        if (mx > r_max) valid = false;
    }
    if (valid && angle_enabled) {  // :255-260
        const double c1 = corner_cos_raw(p2, p1, p3), c2 = corner_cos_raw(p1, p2, p3), c3 = corner_cos_raw(p1, p3, p2);
        if (c1 == c1) {  // Python min(): a leading NaN wins (test passes); later NaNs are skipped
            double mc = c1;
            if (c2 == c2 && c2 > mc) mc = c2;
            if (c3 == c3 && c3 > mc) mc = c3;
            if (mc >= cos_thr) valid = false;
        }
    }
    uint8_t f = valid ? 1 : 0;
    // priority, :420-424: (|a-b| + |b-c|) + |c-a|
    const double per = nrm2(p1.x - p2.x, p1.y - p2.y) + nrm2(p2.x - p3.x, p2.y - p3.y) + nrm2(p3.x - p1.x, p3.y - p1.y);
    const double tot = size[a] + size[b] + size[c];  // :408-410
    if (valid && type_id[a] == type_id[b] && type_id[b] == type_id[c] && !(tot > max_size)) f |= 2;
    out_flag[t] = f;
    out_perim[t] = per;
    out_total[t] = tot;
}

// Greedy vertex-disjoint selection in (key, item index) order (src/metacell_utils.py:438-449): the same
// local-minimum rule as the pair matching, with three endpoints per item.
__global__ __launch_bounds__(256) void disjoint_min_key_kernel(const int32_t *__restrict__ items, const double *__restrict__ keys,
                                                                int64_t M, uint8_t *__restrict__ alive,
                                                                const uint8_t *__restrict__ used,
                                                                unsigned long long *__restrict__ vkey) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M || !alive[m]) return;
    const int32_t a = items[3 * m], b = items[3 * m + 1], c = items[3 * m + 2];
    if (used[a] || used[b] || used[c]) { alive[m] = 0; return; }
    const unsigned long long k = cost_key(keys[m]);
    atomicMin(&vkey[a], k); atomicMin(&vkey[b], k); atomicMin(&vkey[c], k);
}
__global__ __launch_bounds__(256) void disjoint_min_idx_kernel(const int32_t *__restrict__ items, const double *__restrict__ keys,
                                                                int64_t M, const uint8_t *__restrict__ alive,
                                                                const unsigned long long *__restrict__ vkey,
                                                                unsigned *__restrict__ vidx) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M || !alive[m]) return;
    const unsigned long long k = cost_key(keys[m]);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int32_t v = items[3 * m + q];
        if (k == vkey[v]) atomicMin(&vidx[v], (unsigned)m);
    }
}
__global__ __launch_bounds__(256) void disjoint_select_kernel(const int32_t *__restrict__ items, int64_t M, uint8_t *__restrict__ alive,
                                                               const unsigned *__restrict__ vidx, uint8_t *__restrict__ used,
                                                               uint8_t *__restrict__ selected,
                                                               unsigned long long *__restrict__ n_selected) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool sel = false;
    if (m < M && alive[m]) {
        const int32_t a = items[3 * m], b = items[3 * m + 1], c = items[3 * m + 2];
        if (vidx[a] == (unsigned)m && vidx[b] == (unsigned)m && vidx[c] == (unsigned)m) {
            sel = true;
            alive[m] = 0;
            selected[m] = 1;
            used[a] = 1; used[b] = 1; used[c] = 1;
        }
    }
    const unsigned long long bal = __ballot(sel);
    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(n_selected, (unsigned long long)__builtin_popcountll(bal));
}
__global__ __launch_bounds__(256) void disjoint_reset_kernel(unsigned long long *__restrict__ vkey, unsigned *__restrict__ vidx, int64_t n) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n) { vkey[q] = ~0ull; vidx[q] = ~0u; }
}

inline unsigned grid_for(int64_t n) { return (unsigned)ceil_div(n > 0 ? n : 1, 256); }

// ---- the small assignments of unpack_metacell_matches(strategy='nearest') ------------
// src/metacell_utils.py:711-761 solves, per metacell match, linear_sum_assignment on
// cdist(aligned members, ref members) with the ref columns tiled ceil(n_a/n_r) times when there
// are more aligned than ref members.  Problems are tiny (members <= max_metacell_size^iterations)
// and independent, so one lane solves one problem start to finish with its work arrays in a
// private slice of a scratch buffer; the algorithm is scipy's shortest-augmenting-path solver
// (Crouse 2016) step for step -- same scan order of the remaining columns, same "prefer a free
// column among equal minima" rule, same dual updates in the same fp64 operation order -- so
// ties resolve exactly as in the reference.  rows <= cols always holds here.
__global__ __launch_bounds__(256) void batched_assign_kernel(
    int64_t n_prob, const int64_t *__restrict__ a_off, const int64_t *__restrict__ r_off, const double *__restrict__ axy,
    const double *__restrict__ rxy, const int64_t *__restrict__ w_off, double *__restrict__ work, int32_t *__restrict__ out_ref,
    unsigned *__restrict__ infeasible) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_prob) return;
    const int64_t a0 = a_off[p], r0 = r_off[p];
    const int na = (int)(a_off[p + 1] - a0), nref = (int)(r_off[p + 1] - r0);
    if (na == 0) return;
    const int nc = na <= nref ? nref : nref * ((na + nref - 1) / nref);
    double *cost = work + w_off[p], *u = cost + (int64_t)na * nc, *v = u + na, *spc = v + nc;
    int *path = reinterpret_cast<int *>(spc + nc), *row4col = path + nc, *remaining = row4col + nc, *col4row = remaining + nc;
    uint8_t *SR = reinterpret_cast<uint8_t *>(col4row + na), *SC = SR + na;
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    for (int i = 0; i < na; ++i) {
        const double ax = axy[2 * (a0 + i)], ay = axy[2 * (a0 + i) + 1];
        for (int j = 0; j < nref; ++j) {
            const double dx = ax - rxy[2 * (r0 + j)], dy = ay - rxy[2 * (r0 + j) + 1];
            const double d = sqrt(dx * dx + dy * dy);  // cdist euclidean: s = dx*dx; s += dy*dy; sqrt (contraction is off)
            for (int jj = j; jj < nc; jj += nref) cost[(int64_t)i * nc + jj] = d;
        }
        u[i] = 0.0;
        col4row[i] = -1;
    }
    for (int j = 0; j < nc; ++j) { v[j] = 0.0; row4col[j] = -1; path[j] = -1; }
    for (int cur = 0; cur < na; ++cur) {
        double min_val = 0.0;
        int num_remaining = nc, sink = -1, i = cur;
        for (int it = 0; it < nc; ++it) { remaining[it] = nc - it - 1; SC[it] = 0; spc[it] = inf; }
        for (int r = 0; r < na; ++r) SR[r] = 0;
        while (sink == -1) {
            int index = -1;
            double lowest = inf;
            SR[i] = 1;
            const double ui = u[i];
            for (int it = 0; it < num_remaining; ++it) {
                const int j = remaining[it];
                const double r = min_val + cost[(int64_t)i * nc + j] - ui - v[j];
                double s = spc[j];
                if (r < s) { path[j] = i; spc[j] = s = r; }
                if (s < lowest || (s == lowest && row4col[j] == -1)) { lowest = s; index = it; }
            }
            min_val = lowest;
            if (index < 0 || min_val == inf) {  // NaN / inf coordinates: no finite augmenting path (scipy raises)
                atomicOr(infeasible, 1u);
                return;
            }
            const int j = remaining[index];
            if (row4col[j] == -1) sink = j; else i = row4col[j];
            SC[j] = 1;
            remaining[index] = remaining[--num_remaining];
        }
        u[cur] += min_val;
        for (int r = 0; r < na; ++r)
            if (SR[r] && r != cur) u[r] += min_val - spc[col4row[r]];
        for (int j = 0; j < nc; ++j)
            if (SC[j]) v[j] -= min_val - spc[j];
        int j = sink;
        for (;;) {
            const int r = path[j];
            row4col[j] = r;
            const int t = col4row[r];
            col4row[r] = j;
            j = t;
            if (r == cur) break;
        }
    }
    for (int i = 0; i < na; ++i) out_ref[a0 + i] = col4row[i] % nref;
}

}  // namespace

// Rounds [first, first + count) of the greedy rule on device-resident state (declared in common.h; shared with window.hip).
// st.used / st.key[] / st.idx[] / st.sel zero before round 0 (one memset); st.alive = the rows' "prefers a match" flag per
// pair; dmatch_pair = -1.  Round r sets st.sel[r - first] to 1 if it selected a pair (it stays 0 otherwise).  Enqueue only: 3 launches a
// round, 2 when st.float_costs says every cost is exactly a float (then P < 2^32 - 1 and st.idx is not used).
int same_greedy_rounds_core(same_ctx *ctx, const int32_t *dp, const double *dc, int64_t P, const unsigned long long *dP, int64_t n_m,
                            int64_t n_r, const same_greedy_state &st, int32_t *dmatch_pair, int first, int count) {
    const int64_t n_ends = n_m + n_r;
    const unsigned gp = grid_for(P), gs = grid_for(P > n_ends ? P : n_ends);
    for (int r = first; r < first + count; ++r) {
        const int s = r & 1;
        if (st.float_costs) {      // (cost, index) in one word: two launches a round
            SAME_LAUNCH(ctx, greedy_min_packed_kernel, dim3(gp), dim3(256), 0, dp, dc, P, dP, st.alive, st.used, n_m, st.key[s]);
            SAME_LAUNCH(ctx, greedy_select_packed_kernel, dim3(gs), dim3(256), 0, dp, dc, P, dP, st.alive, n_m, n_ends, st.key[s], st.used,
                        dmatch_pair, st.sel + (r - first), st.key[s ^ 1]);
            continue;
        }
        SAME_LAUNCH(ctx, greedy_min_key_kernel, dim3(gp), dim3(256), 0, dp, dc, P, dP, st.alive, st.used, n_m, st.key[s]);
        SAME_LAUNCH(ctx, greedy_min_idx_kernel, dim3(gp), dim3(256), 0, dp, dc, P, dP, st.alive, n_m, st.key[s], st.idx[s]);
        SAME_LAUNCH(ctx, greedy_select_kernel, dim3(gs), dim3(256), 0, dp, P, dP, st.alive, n_m, n_ends, st.idx[s], st.used, dmatch_pair,
                    st.sel + (r - first), st.key[s ^ 1], st.idx[s ^ 1]);
    }
    HIP_TRY(ctx, hipGetLastError());
    return SAME_OK;
}

// The same rounds for n_jobs <= SAME_LAUNCH_WINDOWS independent problems (jobs with P == 0 are skipped by their blocks): one launch per
// map for all of them.  All jobs share float_costs.  Enqueue only.
int same_greedy_rounds_batch_core(same_ctx *ctx, const same_greedy_job *jobs, int n_jobs, int first, int count) {
    REQUIRE(ctx, n_jobs >= 1 && n_jobs <= SAME_LAUNCH_WINDOWS);
    GreedyBatch b{};
    int64_t max_p = 0, max_s = 0;
    const bool packed = jobs[0].st.float_costs;
    for (int q = 0; q < n_jobs; ++q) {
        const same_greedy_job &j = jobs[q];
        REQUIRE(ctx, j.st.float_costs == packed);
        const int64_t n_ends = j.n_m + j.n_r;
        b.w[q] = GreedyArgs{j.pairs, j.costs, j.P, j.n_m, j.P ? n_ends : 0, j.st.alive, j.st.used, {j.st.key[0], j.st.key[1]}, {j.st.idx[0], j.st.idx[1]},
                            j.st.sel, j.match_pair};
        max_p = std::max(max_p, j.P);
        max_s = std::max(max_s, j.P ? std::max(j.P, n_ends) : 0);
    }
    if (max_p == 0) return SAME_OK;
    const dim3 gp(grid_for(max_p), (unsigned)n_jobs), gs(grid_for(max_s), (unsigned)n_jobs);
    for (int r = first; r < first + count; ++r) {
        const int s = r & 1;
        if (packed) {
            SAME_LAUNCH(ctx, greedy_batch_kernel<3>, gp, dim3(256), 0, b, s, r - first);
            SAME_LAUNCH(ctx, greedy_batch_kernel<4>, gs, dim3(256), 0, b, s, r - first);
            continue;
        }
        SAME_LAUNCH(ctx, greedy_batch_kernel<0>, gp, dim3(256), 0, b, s, r - first);
        SAME_LAUNCH(ctx, greedy_batch_kernel<1>, gp, dim3(256), 0, b, s, r - first);
        SAME_LAUNCH(ctx, greedy_batch_kernel<2>, gs, dim3(256), 0, b, s, r - first);
    }
    HIP_TRY(ctx, hipGetLastError());
    return SAME_OK;
}

// The greedy MIP start on device-resident pairs / costs / prefer flags: dmatch[n_m] = pair index per aligned row or -1.
// Scratch: SL_FLAG1, SL_FLAG2, SL_OUT0, SL_OUT1, SL_COUNTS and the pinned block.  A round in which nothing is alive is a no-op, so
// rounds are enqueued in batches (4, 4, 8, 16, ... 64) and the per-round selection counts of a batch come back in ONE read: the
// host waits for the device once per batch, not once per round (a monotone cost chain needs a round per pair).
int same_greedy_core(same_ctx *ctx, const int32_t *dp, const double *dc, int64_t P, int64_t n_m, int64_t n_r, const uint8_t *dprefer,
                     int32_t *dmatch, int *out_rounds) {
    if (out_rounds) *out_rounds = 0;
    if (n_m <= 0) return SAME_OK;
    HIP_TRY(ctx, hipMemsetAsync(dmatch, 0xFF, (size_t)n_m * sizeof(int32_t), ctx->stream));
    if (P <= 0) return SAME_OK;
    const int64_t n_ends = n_m + n_r;
    same_greedy_state st;
    unsigned long long *keys;
    unsigned *idxs;
    SAME_TRY(slot_as(ctx, SL_FLAG1, (size_t)P, &st.alive));
    SAME_TRY(slot_as(ctx, SL_FLAG2, (size_t)n_ends, &st.used));
    SAME_TRY(slot_as(ctx, SL_OUT0, (size_t)2 * n_ends, &keys));
    SAME_TRY(slot_as(ctx, SL_OUT1, (size_t)2 * n_ends, &idxs));
    SAME_TRY(slot_as(ctx, SL_COUNTS, (size_t)SAME_GREEDY_BATCH_MAX, &st.sel));
    st.key[0] = keys; st.key[1] = keys + n_ends;
    st.idx[0] = idxs; st.idx[1] = idxs + n_ends;
    HIP_TRY(ctx, hipMemsetAsync(st.used, 0, (size_t)n_ends, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(keys, 0, (size_t)2 * n_ends * sizeof(unsigned long long), ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(idxs, 0, (size_t)2 * n_ends * sizeof(unsigned), ctx->stream));
    SAME_LAUNCH(ctx, greedy_init_kernel, dim3(grid_for(P)), dim3(256), 0, dp, P, dprefer, st.alive);
    unsigned long long *h = static_cast<unsigned long long *>(ctx->pinned);
    int rounds = 0, batch = 4;
    for (int n_batch = 0;; ++n_batch) {
        REQUIRE(ctx, rounds <= P + 1);  // each productive round removes at least one pair
        HIP_TRY(ctx, hipMemsetAsync(st.sel, 0, (size_t)batch * sizeof(unsigned long long), ctx->stream));
        SAME_TRY(same_greedy_rounds_core(ctx, dp, dc, P, nullptr, n_m, n_r, st, dmatch, rounds, batch));
        SAME_TRY(same_down(ctx, h, st.sel, (size_t)batch * sizeof(unsigned long long)));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        ++ctx->stats[SAME_STAT_GREEDY_READBACKS];
        int q = 0;
        while (q < batch && h[q] != 0) ++q;   // the first round that selected nothing: nothing was alive any more
        rounds += q;
        if (q < batch) break;
        if (n_batch >= 1 && batch < SAME_GREEDY_BATCH_MAX) batch *= 2;
    }
    if (out_rounds) *out_rounds = rounds;
    return SAME_OK;
}

extern "C" {

int same_greedy_match(same_ctx *ctx, const int32_t *pairs, const double *costs, int64_t P, int64_t n_m, int64_t n_r,
                      const uint8_t *prefer, int32_t *out_match_pair, int *out_rounds) {
    REQUIRE(ctx, ctx != nullptr);
    REQUIRE(ctx, P >= 0 && n_m >= 0 && n_r >= 0 && P < ((int64_t)1 << 32) - 1);
    if (out_rounds) *out_rounds = 0;
    REQUIRE(ctx, n_m == 0 || out_match_pair);
    for (int64_t i = 0; i < n_m; ++i) out_match_pair[i] = -1;
    if (P == 0 || n_m == 0) return SAME_OK;
    REQUIRE(ctx, pairs && costs && prefer);
    SAME_TRY(same_use(ctx));
    for (int64_t p = 0; p < P; ++p)
        if (pairs[2 * p] < 0 || pairs[2 * p] >= n_m || pairs[2 * p + 1] < 0 || pairs[2 * p + 1] >= n_r) {
            ctx->err = "pair index out of range";
            return SAME_ERANGE;
        }
    int32_t *dp, *dmatch;
    double *dc;
    uint8_t *dprefer;
    SAME_TRY(up_as(ctx, SL_PAIRS, pairs, (size_t)P * 2, &dp));
    SAME_TRY(up_as(ctx, SL_X, costs, (size_t)P, &dc));
    SAME_TRY(up_as(ctx, SL_FLAG0, prefer, (size_t)n_m, &dprefer));
    SAME_TRY(slot_as(ctx, SL_MATCH, (size_t)n_m, &dmatch));
    int rounds = 0;
    SAME_TRY(same_greedy_core(ctx, dp, dc, P, n_m, n_r, dprefer, dmatch, &rounds));
    if (out_rounds) *out_rounds = rounds;
    SAME_TRY(same_down(ctx, out_match_pair, dmatch, (size_t)n_m * sizeof(int32_t)));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SAME_OK;
}

int same_tri_flip_stats(same_ctx *ctx, const double *axy, const double *mapped_xy, const uint8_t *matched, int64_t n,
                        const int32_t *type_id, const int32_t *tris, int64_t Tr, uint8_t *out_tri_flag,
                        uint32_t *out_node_tri, uint32_t *out_node_flip) {
    REQUIRE(ctx, ctx != nullptr);
    REQUIRE(ctx, n >= 0 && Tr >= 0 && (n == 0 || (out_node_tri && out_node_flip)));
    for (int64_t i = 0; i < n; ++i) { out_node_tri[i] = 0; out_node_flip[i] = 0; }
    if (Tr == 0) return SAME_OK;
    REQUIRE(ctx, axy && mapped_xy && matched && tris && out_tri_flag && out_node_tri && out_node_flip);
    SAME_TRY(same_use(ctx));
    SAME_TRY(check_index_range(ctx, tris, Tr * 3, 0, n, "triangles"));
    double *dax, *dmx;
    uint8_t *dmatched, *dflag;
    int32_t *dtype = nullptr, *dtris;
    unsigned *dnt, *dnf;
    SAME_TRY(up_as(ctx, SL_AXY, axy, (size_t)n * 2, &dax));
    SAME_TRY(up_as(ctx, SL_RXY, mapped_xy, (size_t)n * 2, &dmx));
    SAME_TRY(up_as(ctx, SL_FLAG0, matched, (size_t)n, &dmatched));
    if (type_id) SAME_TRY(up_as(ctx, SL_TYPE, type_id, (size_t)n, &dtype));
    SAME_TRY(up_as(ctx, SL_TRIS, tris, (size_t)Tr * 3, &dtris));
    SAME_TRY(slot_as(ctx, SL_FLAG1, (size_t)Tr, &dflag));
    SAME_TRY(slot_as(ctx, SL_OUT0, (size_t)n * 2, &dnt));
    dnf = dnt + n;
    HIP_TRY(ctx, hipMemsetAsync(dnt, 0, (size_t)n * 2 * sizeof(unsigned), ctx->stream));
    hipLaunchKernelGGL(tri_flip_stats_kernel, dim3(grid_for(Tr)), dim3(256), 0, ctx->stream, dax, dmx, dmatched, dtype, dtris, Tr,
                       dflag, dnt, dnf);
    HIP_TRY(ctx, hipGetLastError());
    SAME_TRY(same_down(ctx, out_tri_flag, dflag, (size_t)Tr));
    SAME_TRY(same_down(ctx, out_node_tri, dnt, (size_t)n * sizeof(unsigned)));
    SAME_TRY(same_down(ctx, out_node_flip, dnf, (size_t)n * sizeof(unsigned)));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SAME_OK;
}

int same_collapse_candidates(same_ctx *ctx, const double *xy, int64_t n, const int32_t *tris, int64_t Tr, int rmax_enabled,
                             double r_max, int angle_enabled, double cos_thr, const int32_t *type_id, const double *size,
                             double max_size, uint8_t *out_flag, double *out_perim, double *out_total) {
    REQUIRE(ctx, ctx != nullptr);
    REQUIRE(ctx, n >= 0 && Tr >= 0);
    if (Tr == 0) return SAME_OK;
    REQUIRE(ctx, xy && tris && type_id && size && out_flag && out_perim && out_total);
    SAME_TRY(same_use(ctx));
    SAME_TRY(check_index_range(ctx, tris, Tr * 3, 0, n, "triangles"));
    double *dxy, *dsize, *dperim, *dtot;
    int32_t *dtris, *dtype;
    uint8_t *dflag;
    SAME_TRY(up_as(ctx, SL_AXY, xy, (size_t)n * 2, &dxy));
    SAME_TRY(up_as(ctx, SL_SIZE, size, (size_t)n, &dsize));
    SAME_TRY(up_as(ctx, SL_TYPE, type_id, (size_t)n, &dtype));
    SAME_TRY(up_as(ctx, SL_TRIS, tris, (size_t)Tr * 3, &dtris));
    SAME_TRY(slot_as(ctx, SL_FLAG0, (size_t)Tr, &dflag));
    SAME_TRY(slot_as(ctx, SL_OUT0, (size_t)Tr, &dperim));
    SAME_TRY(slot_as(ctx, SL_OUT1, (size_t)Tr, &dtot));
    hipLaunchKernelGGL(collapse_candidates_kernel, dim3(grid_for(Tr)), dim3(256), 0, ctx->stream, dxy, dtris, Tr, rmax_enabled, r_max,
                       angle_enabled, cos_thr, dtype, dsize, max_size, dflag, dperim, dtot);
    HIP_TRY(ctx, hipGetLastError());
    SAME_TRY(same_down(ctx, out_flag, dflag, (size_t)Tr));
    SAME_TRY(same_down(ctx, out_perim, dperim, (size_t)Tr * sizeof(double)));
    SAME_TRY(same_down(ctx, out_total, dtot, (size_t)Tr * sizeof(double)));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SAME_OK;
}

int same_greedy_disjoint(same_ctx *ctx, const int32_t *items, const double *keys, int64_t M, int64_t n_nodes,
                         uint8_t *out_selected, int *out_rounds) {
    REQUIRE(ctx, ctx != nullptr);
    REQUIRE(ctx, M >= 0 && n_nodes >= 0 && M < ((int64_t)1 << 32) - 1);
    if (out_rounds) *out_rounds = 0;
    if (M == 0) return SAME_OK;
    REQUIRE(ctx, items && keys && out_selected);
    SAME_TRY(same_use(ctx));
    SAME_TRY(check_index_range(ctx, items, M * 3, 0, n_nodes, "items"));
    int32_t *ditems;
    double *dkeys;
    uint8_t *dalive, *dused, *dsel;
    unsigned long long *dvkey, *dcount;
    unsigned *dvidx;
    SAME_TRY(up_as(ctx, SL_TRIS, items, (size_t)M * 3, &ditems));
    SAME_TRY(up_as(ctx, SL_X, keys, (size_t)M, &dkeys));
    SAME_TRY(slot_as(ctx, SL_FLAG0, (size_t)M, &dalive));
    SAME_TRY(slot_as(ctx, SL_FLAG1, (size_t)M, &dsel));
    SAME_TRY(slot_as(ctx, SL_FLAG2, (size_t)n_nodes, &dused));
    SAME_TRY(slot_as(ctx, SL_OUT0, (size_t)n_nodes, &dvkey));
    SAME_TRY(slot_as(ctx, SL_OUT1, (size_t)n_nodes, &dvidx));
    SAME_TRY(slot_as(ctx, SL_COUNTS, (size_t)4, &dcount));
    HIP_TRY(ctx, hipMemsetAsync(dalive, 1, (size_t)M, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(dsel, 0, (size_t)M, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(dused, 0, (size_t)n_nodes, ctx->stream));
    // rounds in batches (4, 4, 8 ... 64) with one read of the per-round selection counts per batch, as same_greedy_core does: a
    // round in which nothing is alive is a no-op, so overshooting costs launches, never results
    unsigned long long *h = static_cast<unsigned long long *>(ctx->pinned);
    int rounds = 0, batch = 4;
    SAME_TRY(slot_as(ctx, SL_COUNTS, (size_t)SAME_GREEDY_BATCH_MAX, &dcount));
    for (int n_batch = 0;; ++n_batch) {
        REQUIRE(ctx, rounds <= M + 1);
        HIP_TRY(ctx, hipMemsetAsync(dcount, 0, (size_t)batch * sizeof(unsigned long long), ctx->stream));
        for (int q = 0; q < batch; ++q) {
            hipLaunchKernelGGL(disjoint_reset_kernel, dim3(grid_for(n_nodes)), dim3(256), 0, ctx->stream, dvkey, dvidx, n_nodes);
            hipLaunchKernelGGL(disjoint_min_key_kernel, dim3(grid_for(M)), dim3(256), 0, ctx->stream, ditems, dkeys, M, dalive, dused, dvkey);
            hipLaunchKernelGGL(disjoint_min_idx_kernel, dim3(grid_for(M)), dim3(256), 0, ctx->stream, ditems, dkeys, M, dalive, dvkey, dvidx);
            hipLaunchKernelGGL(disjoint_select_kernel, dim3(grid_for(M)), dim3(256), 0, ctx->stream, ditems, M, dalive, dvidx, dused, dsel, dcount + q);
        }
        HIP_TRY(ctx, hipGetLastError());
        SAME_TRY(same_down(ctx, h, dcount, (size_t)batch * sizeof(unsigned long long)));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        ++ctx->stats[SAME_STAT_GREEDY_READBACKS];
        int q = 0;
        while (q < batch && h[q] != 0) ++q;
        rounds += q;
        if (q < batch) break;
        if (n_batch >= 1 && batch < SAME_GREEDY_BATCH_MAX) batch *= 2;
    }
    if (out_rounds) *out_rounds = rounds;
    SAME_TRY(same_down(ctx, out_selected, dsel, (size_t)M));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SAME_OK;
}

int same_batched_assign(same_ctx *ctx, int64_t n_prob, const int64_t *a_off, const int64_t *r_off, const double *axy,
                        const double *rxy, int32_t *out_ref) {
    REQUIRE(ctx, ctx != nullptr);
    REQUIRE(ctx, n_prob >= 0);
    if (n_prob == 0) return SAME_OK;
    REQUIRE(ctx, a_off && r_off);
    REQUIRE(ctx, a_off[0] == 0 && r_off[0] == 0);
    // private work slice per problem, in 8-byte words: cost[na*nc] u[na] v[nc] spc[nc] | path,row4col,remaining[nc] col4row[na] | SR[na] SC[nc]
    std::vector<int64_t> w_off((size_t)n_prob + 1);
    int64_t words = 0;
    for (int64_t p = 0; p < n_prob; ++p) {
        const int64_t na = a_off[p + 1] - a_off[p], nref = r_off[p + 1] - r_off[p];
        REQUIRE(ctx, na >= 0 && nref >= 0 && na <= SAME_ASSIGN_MAX_MEMBERS && nref <= SAME_ASSIGN_MAX_MEMBERS);
        REQUIRE(ctx, na == 0 || nref > 0);
        w_off[(size_t)p] = words;
        if (na == 0) continue;
        const int64_t nc = na <= nref ? nref : nref * ((na + nref - 1) / nref);
        words += na * nc + na + 2 * nc + (3 * nc + na + 1) / 2 + (na + nc + 7) / 8;
    }
    w_off[(size_t)n_prob] = words;
    const int64_t n_a = a_off[n_prob], n_r = r_off[n_prob];
    if (n_a == 0) return SAME_OK;
    REQUIRE(ctx, axy && rxy && out_ref);
    SAME_TRY(same_use(ctx));
    int64_t *da_off, *dr_off, *dw_off;
    double *daxy, *drxy, *dwork;
    int32_t *dout;
    unsigned *dflag;
    SAME_TRY(up_as(ctx, SL_A, a_off, (size_t)n_prob + 1, &da_off));
    SAME_TRY(up_as(ctx, SL_R, r_off, (size_t)n_prob + 1, &dr_off));
    SAME_TRY(up_as(ctx, SL_X, w_off.data(), (size_t)n_prob + 1, &dw_off));
    SAME_TRY(up_as(ctx, SL_AXY, axy, (size_t)n_a * 2, &daxy));
    SAME_TRY(up_as(ctx, SL_RXY, rxy, (size_t)n_r * 2, &drxy));
    SAME_TRY(slot_as(ctx, SL_OUT0, (size_t)std::max<int64_t>(words, 1), &dwork));
    SAME_TRY(slot_as(ctx, SL_OUT1, (size_t)n_a, &dout));
    SAME_TRY(slot_as(ctx, SL_COUNTS, (size_t)4, &dflag));
    HIP_TRY(ctx, hipMemsetAsync(dflag, 0, sizeof(unsigned), ctx->stream));
    hipLaunchKernelGGL(batched_assign_kernel, dim3(grid_for(n_prob)), dim3(256), 0, ctx->stream, n_prob, da_off, dr_off, daxy, drxy,
                       dw_off, dwork, dout, dflag);
    HIP_TRY(ctx, hipGetLastError());
    unsigned *h = static_cast<unsigned *>(ctx->pinned);
    SAME_TRY(same_down(ctx, h, dflag, sizeof(unsigned)));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (h[0]) {
        ctx->err = "same_batched_assign: a cost matrix is infeasible (non-finite coordinates)";
        return SAME_ERANGE;
    }
    SAME_TRY(same_down(ctx, out_ref, dout, (size_t)n_a * sizeof(int32_t)));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SAME_OK;
}

}  // extern "C"
