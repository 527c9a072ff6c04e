// sweep.hip -- violation sweeps under a candidate matching: the lazy-constraint orientation
// sweep (a10, src/same.py:631-669), the XY-order report sweep (a11,
// src/violationhelper.py:53-117), plus the small integer/index kernels around them (matching
// from the solver's x vector, per-row minimum cost and dense assignment matrix for the MIP
// start (a5), window membership (a13)).
//
// The orientation sweep runs once per solver incumbent, so it is latency-bound: triangles,
// source signs, reference XY and the pair list stay resident (same_sweep_bind); a call uploads
// only x (or the match vector), runs 2-3 small kernels and downloads two counters plus the
// ascending list of flipped triangles.  Ordered compaction: kernel 1 ballots the flipped
// predicate into one 64-bit word per wave; kernel 2 (one block) scans the popcounts and writes
// the indices, so the list is ascending without a sort and bit-reproducible.
#include <algorithm>
#include <utility>

#include "common.h"
#include "devmath.h"
#include "scan.h"

namespace {

using namespace devmath;   // ld2, orient_sign, orient_flag, xyorder_edge, order_key: one definition (devmath.h)

// ---- matching from x: last pair (highest pair index) with x > 0.5 wins per aligned row -----
__global__ __launch_bounds__(256) void match_init_kernel(int32_t *__restrict__ pidx, int64_t n_m) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_m) pidx[i] = -1;
}
__global__ __launch_bounds__(256) void match_scan_kernel(const double *__restrict__ x, const int32_t *__restrict__ pairs,
                                                          int64_t P, int32_t *__restrict__ pidx) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < P && x[p] > 0.5) atomicMax(&pidx[pairs[2 * p]], (int32_t)p);  // src/same.py:636-639
}
__global__ __launch_bounds__(256) void match_resolve_kernel(const int32_t *__restrict__ pairs, const int32_t *__restrict__ pidx,
                                                             int64_t n_m, int32_t *__restrict__ match) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_m) match[i] = pidx[i] >= 0 ? pairs[2 * (int64_t)pidx[i] + 1] : -1;
}

// ---- orientation sweep -------------------------------------------------------------------
// flags of a block of triangles (the sharded sweep's per-rank share): 0 not checked, 1 checked, 2 checked and flipped
__global__ __launch_bounds__(256) void orient_flag_kernel(
    const int32_t *__restrict__ tris, int64_t Tr, const int8_t *__restrict__ src_sign, const double *__restrict__ rxy,
    const int32_t *__restrict__ match, uint8_t *__restrict__ flag) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= Tr) return;
    const int32_t a = tris[3 * t], b = tris[3 * t + 1], c = tris[3 * t + 2];
    const int32_t ja = match[a], jb = match[b], jc = match[c];
    const bool all3 = ja >= 0 && jb >= 0 && jc >= 0;           // src/same.py:649-650
    const double2_t z = {0.0, 0.0};
    flag[t] = orient_flag(src_sign[t], all3, all3 ? ld2(rxy, ja) : z, all3 ? ld2(rxy, jb) : z, all3 ? ld2(rxy, jc) : z);   // :658-669
}

// checked counter and the ascending list of flipped triangles from a COMPLETE flag array (the all-gathered flags of a sharded
// sweep), one launch: multi-block look-back scan (scan.h).  counters[0] = checked, counters[1] = flipped.
__global__ __launch_bounds__(scan::NT) void flags_compact_kernel(const uint8_t *__restrict__ flag, int64_t Tr, unsigned long long *__restrict__ status,
                                                                  int32_t *__restrict__ viol, unsigned long long *__restrict__ counters) {
    __shared__ scan::Shared sh;
    __shared__ int wave_checked[scan::NT / 64];
    auto val = [&](int64_t i) { return scan::Pair{i < Tr && flag[i] == 2 ? 1u : 0u, 0u}; };
    scan::Pair through;
    const scan::Pair off = scan::exclusive(status, (int)blockIdx.x, val, sh, &through);
    const int64_t t = (int64_t)blockIdx.x * scan::NT + threadIdx.x;
    const uint8_t f = t < Tr ? flag[t] : 0;
    if (f == 2) viol[off.a] = (int32_t)t;
    const unsigned long long checked = __ballot(f != 0);
    if ((threadIdx.x & 63) == 0) wave_checked[threadIdx.x >> 6] = __builtin_popcountll(checked);
    __syncthreads();
    if (threadIdx.x == 0) {
        int c = 0;
        for (int q = 0; q < scan::NT / 64; ++q) c += wave_checked[q];
        if (c) atomicAdd(&counters[0], (unsigned long long)c);
        if (blockIdx.x == gridDim.x - 1) counters[1] = through.a;
    }
}

// nearest candidate of every row of a padded candidate list: match[i] = idx[i][0] (-1 = no candidate)
__global__ __launch_bounds__(256) void first_candidate_kernel(const int32_t *__restrict__ idx, int64_t rows, int k,
                                                               int32_t *__restrict__ match) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows) match[i] = idx[i * k];
}

// The per-incumbent sweep in ONE launch: flag per triangle, the checked counter, and the flipped triangles as an ascending list
// through the multi-block look-back scan of scan.h (no second launch for the compaction, no single-block scan).
// counters[0] = checked, counters[1] = flipped.
__global__ __launch_bounds__(scan::NT) void orient_sweep_kernel(
    const int32_t *__restrict__ tris, int64_t Tr, const int8_t *__restrict__ src_sign, const double *__restrict__ rxy,
    const int32_t *__restrict__ match, uint8_t *__restrict__ flag, unsigned long long *__restrict__ status, int32_t *__restrict__ viol,
    unsigned long long *__restrict__ counters) {
    __shared__ scan::Shared sh;
    __shared__ int wave_checked[scan::NT / 64];
    auto flag_of = [&](int64_t t) -> uint8_t {
        if (t >= Tr) return 0;
        const int32_t a = tris[3 * t], b = tris[3 * t + 1], c = tris[3 * t + 2];
        const int32_t ja = match[a], jb = match[b], jc = match[c];
        const bool all3 = ja >= 0 && jb >= 0 && jc >= 0;           // src/same.py:649-650
        const double2_t z = {0.0, 0.0};
        return orient_flag(src_sign[t], all3, all3 ? ld2(rxy, ja) : z, all3 ? ld2(rxy, jb) : z, all3 ? ld2(rxy, jc) : z);   // :658-669
    };
    const int64_t t = (int64_t)blockIdx.x * scan::NT + threadIdx.x;
    const uint8_t f = flag_of(t);
    if (t < Tr) flag[t] = f;
    // the scan's element function must be able to produce ANY block's counters (a predecessor that has not published is recomputed,
    // never waited for): the own element comes from the register, everything else from the inputs
    auto val = [&](int64_t i) { return scan::Pair{(i == t ? f : flag_of(i)) == 2 ? 1u : 0u, 0u}; };
    scan::Pair through;
    const scan::Pair off = scan::exclusive(status, (int)blockIdx.x, val, sh, &through);
    if (f == 2) viol[off.a] = (int32_t)t;
    const unsigned long long checked = __ballot(f != 0);
    if ((threadIdx.x & 63) == 0) wave_checked[threadIdx.x >> 6] = __builtin_popcountll(checked);
    __syncthreads();
    if (threadIdx.x == 0) {  // one atomic per block (integer sum: order-independent)
        int c = 0;
        for (int q = 0; q < scan::NT / 64; ++q) c += wave_checked[q];
        if (c) atomicAdd(&counters[0], (unsigned long long)c);
        if (blockIdx.x == gridDim.x - 1) counters[1] = through.a;
    }
}

// ---- XY-order sweep ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void xyorder_kernel(
    const double *__restrict__ axy, const double *__restrict__ rxy, const int32_t *__restrict__ tris, int64_t Tr,
    const int32_t *__restrict__ match, uint8_t *__restrict__ edge_flags, uint8_t *__restrict__ tri_flag,
    uint8_t *__restrict__ point_flag, unsigned long long *__restrict__ counts) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int ncmp = 0, nviol = 0, tv = 0;
    if (t < Tr) {
        int32_t v[3] = {tris[3 * t], tris[3 * t + 1], tris[3 * t + 2]};
        int32_t m[3] = {match[v[0]], match[v[1]], match[v[2]]};
        double2_t a[3], r[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            a[q] = ld2(axy, v[q]);
            r[q] = m[q] >= 0 ? ld2(rxy, m[q]) : double2_t{0.0, 0.0};
        }
        const int E[3][2] = {{0, 1}, {0, 2}, {1, 2}};
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            const int p = E[e][0], q = E[e][1];
            uint8_t f = 0;
            if (m[p] >= 0 && m[q] >= 0) {  // both matched (implies >= 2 matched vertices, violationhelper.py:58-60)
                f = 1;
                ++ncmp;
                const uint8_t e2 = xyorder_edge(a[p], a[q], r[p], r[q]);   // violationhelper.py:68-75
                f |= e2;
                nviol += ((e2 >> 1) & 1) + ((e2 >> 2) & 1);
                if (f & 6) { tv = 1; point_flag[v[p]] = 1; point_flag[v[q]] = 1; }  // benign: every writer stores 1
            }
            edge_flags[3 * t + e] = f;
        }
        tri_flag[t] = (uint8_t)tv;
    }
    // wave reduction, then block reduction through LDS -> one atomic per block per counter
    // (integer sums: order-independent, bit-reproducible)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        ncmp += __shfl_down(ncmp, off, 64);
        nviol += __shfl_down(nviol, off, 64);
        tv += __shfl_down(tv, off, 64);
    }
    __shared__ int part[4][3];
    if ((threadIdx.x & 63) == 0) {
        part[threadIdx.x >> 6][0] = ncmp;
        part[threadIdx.x >> 6][1] = nviol;
        part[threadIdx.x >> 6][2] = tv;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int v = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
        if (v) atomicAdd(&counts[threadIdx.x], (unsigned long long)v);
    }
}

// ---- a5 helpers -----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fill_u64_kernel(unsigned long long *__restrict__ p, int64_t n, unsigned long long v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ __launch_bounds__(256) void rowmin_kernel(const int32_t *__restrict__ pairs, const double *__restrict__ costs,
                                                      int64_t P, unsigned long long *__restrict__ keys) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < P) atomicMin(&keys[pairs[2 * p]], order_key(costs[p]));  // src/init_helpers.py:118-122
}
__global__ __launch_bounds__(256) void rowmin_decode_kernel(const unsigned long long *__restrict__ keys, int64_t n,
                                                             double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = key_to_double(keys[i]);
}
__global__ __launch_bounds__(256) void fill_f64_kernel(double *__restrict__ p, int64_t n, double v) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i + 1 < n) {
        typedef double d2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<d2 *>(p + i) = d2{v, v};
    } else if (i < n) {
        p[i] = v;
    }
}
// later duplicates win (src/init_helpers.py:152-153): keep the highest pair index per cell
__global__ __launch_bounds__(256) void scatter_pairs_kernel(const int32_t *__restrict__ pairs, const double *__restrict__ costs,
                                                             int64_t P, int64_t ld, double *__restrict__ out,
                                                             const int32_t *__restrict__ winner) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < P && winner[p] == (int32_t)p) out[(int64_t)pairs[2 * p] * ld + pairs[2 * p + 1]] = costs[p];
}
__global__ __launch_bounds__(256) void diag_kernel(const double *__restrict__ unmatched, int64_t n_m, int64_t n_r, int64_t ld,
                                                    double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_m) out[i * ld + n_r + i] = unmatched[i];  // src/init_helpers.py:154-155
}

// ---- a13: window membership ----------------------------------------------------------------
__global__ __launch_bounds__(256) void window_count_kernel(const double *__restrict__ xy, int64_t n, const double *__restrict__ boxes,
                                                            int64_t n_boxes, unsigned long long *__restrict__ counts,
                                                            uint8_t *__restrict__ mask) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < n;
    const double2_t p = ld2(xy, valid ? i : 0);
    for (int64_t b = 0; b < n_boxes; ++b) {
        const double x0 = boxes[4 * b], x1 = boxes[4 * b + 1], y0 = boxes[4 * b + 2], y1 = boxes[4 * b + 3];
        const bool in = valid && p.x >= x0 && p.x < x1 && p.y >= y0 && p.y < y1;  // src/same.py:293-295
        if (mask && valid) mask[b * n + i] = in;
        const unsigned long long bal = __ballot(in);
        if ((threadIdx.x & 63) == 0 && bal) atomicAdd(&counts[b], (unsigned long long)__builtin_popcountll(bal));
    }
}

inline unsigned grid_for(int64_t n) { return (unsigned)ceil_div(n > 0 ? n : 1, 256); }

// flags of triangles [t_begin, t_end) of a bound sweep, written at their absolute positions in dflag
int launch_orient_flags(same_sweep *s, const int32_t *dmatch, int64_t t_begin, int64_t t_end, uint8_t *dflag) {
    same_ctx *ctx = s->ctx;
    const int64_t n = t_end - t_begin;
    if (n <= 0) return SAME_OK;
    hipLaunchKernelGGL(orient_flag_kernel, dim3(grid_for(n)), dim3(256), 0, ctx->stream, s->tris + 3 * t_begin, n,
                       s->sign + t_begin, s->rxy, dmatch, dflag + t_begin);
    HIP_TRY(ctx, hipGetLastError());
    return SAME_OK;
}

// ascending list of flipped triangles + counters from a complete flag array (the single-GPU sweep's own flags,
// or the all-gathered flags of a triangle-block sharded sweep)
// How many entries of the flipped list travel with the counters in the first read-back (through the context's pinned block)
constexpr int64_t VIOL_HEAD = 4096;
inline size_t head_bytes(const same_sweep *s) { return 2 * sizeof(unsigned long long) + (size_t)std::min<int64_t>(s->Tr, VIOL_HEAD) * sizeof(int32_t); }

// after the compaction kernel: counters + head of the list in one copy, the rest of a long list in a second one
int read_back(same_sweep *s, int64_t *out_checked, int32_t *out_viol_idx, int64_t *out_nviol) {
    same_ctx *ctx = s->ctx;
    unsigned char *h = static_cast<unsigned char *>(ctx->pinned);
    HIP_TRY(ctx, hipMemcpyAsync(h, s->cnt, head_bytes(s), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    unsigned long long c[2];
    memcpy(c, h, sizeof c);
    *out_checked = (int64_t)c[0];
    *out_nviol = (int64_t)c[1];
    if (c[1] && out_viol_idx) {
        const int64_t head = std::min<int64_t>((int64_t)c[1], VIOL_HEAD);
        memcpy(out_viol_idx, h + sizeof c, (size_t)head * sizeof(int32_t));
        if ((int64_t)c[1] > head) {
            SAME_TRY(same_down(ctx, out_viol_idx + head, s->viol + head, (size_t)((int64_t)c[1] - head) * sizeof(int32_t)));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        }
    }
    return SAME_OK;
}

// ascending list of flipped triangles + counters from a complete flag array (the all-gathered flags of a triangle-block sharded sweep)
int compact_from_flags(same_sweep *s, const uint8_t *dflag, int64_t *out_checked, int32_t *out_viol_idx, int64_t *out_nviol) {
    same_ctx *ctx = s->ctx;
    HIP_TRY(ctx, hipMemsetAsync(s->scan_base, 0, s->scan_zero_bytes, ctx->stream));
    hipLaunchKernelGGL(flags_compact_kernel, dim3(scan::blocks_for(s->Tr)), dim3(scan::NT), 0, ctx->stream, dflag, s->Tr, scan::arg(s->scan_base), s->viol, s->cnt);
    HIP_TRY(ctx, hipGetLastError());
    return read_back(s, out_checked, out_viol_idx, out_nviol);
}

// The per-incumbent sweep from the handle's own match block is the launch-bound inner loop of the path (the solver calls it
// for every incumbent): three stream operations -- one fill (scan words + counters), ONE kernel (flags, counters, ordered list),
// one read-back.  Replaying the round-2 form (four operations) as one captured hipGraph was measured SLOWER on this ROCm (69 us
// against 61 us per call at 95k triangles, profiles/archive/r02_sweep_latency.log), so plain launches are the only form.
int run_orient(same_sweep *s, const int32_t *dmatch, int64_t *out_checked, int32_t *out_viol_idx, int64_t *out_nviol,
               uint8_t *out_flag) {
    same_ctx *ctx = s->ctx;
    *out_checked = 0;
    *out_nviol = 0;
    if (s->Tr == 0) return SAME_OK;
    HIP_TRY(ctx, hipMemsetAsync(s->scan_base, 0, s->scan_zero_bytes, ctx->stream));     // the scan's words + the two counters: one fill
    hipLaunchKernelGGL(orient_sweep_kernel, dim3(scan::blocks_for(s->Tr)), dim3(scan::NT), 0, ctx->stream, s->tris, s->Tr, s->sign, s->rxy, dmatch,
                       s->flag, scan::arg(s->scan_base), s->viol, s->cnt);
    HIP_TRY(ctx, hipGetLastError());
    if (out_flag) SAME_TRY(same_down(ctx, out_flag, s->flag, (size_t)s->Tr));
    return read_back(s, out_checked, out_viol_idx, out_nviol);
}

template <typename T>
int sweep_block(same_ctx *ctx, T **out, size_t n, const T *host) {
    void *p = nullptr;
    HIP_TRY(ctx, hipMalloc(&p, (n ? n : 1) * sizeof(T)));
    *out = static_cast<T *>(p);
    if (host && n) HIP_TRY(ctx, hipMemcpyAsync(p, host, n * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    return SAME_OK;
}

int sweep_fill(same_sweep *s, const int32_t *tris, const int8_t *src_sign, const double *rxy, const int32_t *pairs) {
    same_ctx *ctx = s->ctx;
    SAME_TRY(sweep_block(ctx, &s->tris, (size_t)s->Tr * 3, tris));
    SAME_TRY(sweep_block(ctx, &s->sign, (size_t)s->Tr, src_sign));
    SAME_TRY(sweep_block(ctx, &s->rxy, (size_t)s->n_r * 2, rxy));
    SAME_TRY(sweep_block(ctx, &s->pairs, (size_t)s->P * 2, pairs));
    SAME_TRY(sweep_block<int32_t>(ctx, &s->match, (size_t)s->n_m, nullptr));
    SAME_TRY(sweep_block<int32_t>(ctx, &s->pidx, (size_t)s->n_m, nullptr));
    SAME_TRY(sweep_block<uint8_t>(ctx, &s->flag, (size_t)ceil_div(s->Tr, 256) * 256 + 256, nullptr));
    // one block: [scan words of orient_sweep_kernel | cnt[2] | the flipped list] -- the words and the counters are zeroed by one fill,
    // the counters and the head of the list come back in one copy
    const size_t words = scan::status_bytes(s->Tr) / 8;
    SAME_TRY(sweep_block<unsigned long long>(ctx, &s->scan_base, words + 2 + (size_t)(s->Tr + 1) / 2 + 2, nullptr));
    s->scan_zero_bytes = (words + 2) * sizeof(unsigned long long);
    s->cnt = s->scan_base + words;
    s->viol = reinterpret_cast<int32_t *>(s->cnt + 2);
    SAME_TRY(sweep_block<double>(ctx, &s->x, (size_t)s->P, nullptr));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SAME_OK;
}

}  // namespace

// ---- device cores shared with the window pipeline (window.hip); declared in common.h ----------------------------------
// per-row minimum pair cost from device-resident pairs / costs into dout[n_m] (enqueue only; scratch: SL_MASK)
int same_pair_rowmin_core(same_ctx *ctx, const int32_t *dp, const double *dc, int64_t P, int64_t n_m, double *dout) {
    if (n_m <= 0) return SAME_OK;
    unsigned long long *dk;
    SAME_TRY(slot_as(ctx, SL_MASK, (size_t)n_m, &dk));
    hipLaunchKernelGGL(fill_u64_kernel, dim3(grid_for(n_m)), dim3(256), 0, ctx->stream, dk, n_m, 0xFFF0000000000000ull /* key(+inf) */);
    if (P) hipLaunchKernelGGL(rowmin_kernel, dim3(grid_for(P)), dim3(256), 0, ctx->stream, dp, dc, P, dk);
    hipLaunchKernelGGL(rowmin_decode_kernel, dim3(grid_for(n_m)), dim3(256), 0, ctx->stream, dk, n_m, dout);
    HIP_TRY(ctx, hipGetLastError());
    return SAME_OK;
}

extern "C" {

int same_sweep_bind(same_ctx *ctx, const int32_t *tris, int64_t Tr, const int8_t *src_sign, const double *rxy,
                    int64_t n_r, int64_t n_m, const int32_t *pairs, int64_t P, same_sweep **out) {
    REQUIRE(ctx, ctx && out);
    *out = nullptr;
    REQUIRE(ctx, Tr >= 0 && n_r >= 0 && n_m >= 0 && P >= 0 && Tr < ((int64_t)1 << 31));
    REQUIRE(ctx, (Tr == 0 || (tris && src_sign)) && (n_r == 0 || rxy) && (P == 0 || pairs));
    SAME_TRY(same_use(ctx));
    SAME_TRY(check_index_range(ctx, tris, Tr * 3, 0, n_m, "triangles"));
    for (int64_t p = 0; p < P; ++p)
        if (pairs[2 * p] < 0 || pairs[2 * p] >= n_m || pairs[2 * p + 1] < 0 || pairs[2 * p + 1] >= n_r) {
            ctx->err = "pair index out of range";
            return SAME_ERANGE;
        }
    same_sweep *s = new (std::nothrow) same_sweep();
    if (!s) return SAME_ENOMEM;
    s->ctx = ctx; s->Tr = Tr; s->n_r = n_r; s->n_m = n_m; s->P = P;
    const int rc = sweep_fill(s, tris, src_sign, rxy, pairs);
    if (rc != SAME_OK) { same_sweep_unbind(s); return rc; }
    *out = s;
    return SAME_OK;
}

void same_sweep_unbind(same_sweep *s) {
    if (!s) return;
    same_ctx *ctx = s->ctx;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    void *blocks[] = {s->tris, s->sign, s->rxy, s->pairs, s->match, s->pidx, s->flag, s->scan_base, s->x};   // viol lives in cnt's block
    for (void *b : blocks)
        if (b) (void)hipFree(b);
    delete s;
}

int same_orient_sweep(same_sweep *s, const int32_t *match, int64_t n_m, int64_t *out_checked, int32_t *out_viol_idx,
                      int64_t *out_nviol, uint8_t *out_flag) {
    if (!s) return SAME_EINVAL;
    same_ctx *ctx = s->ctx;
    REQUIRE(ctx, out_checked && out_nviol && n_m == s->n_m && (n_m == 0 || match));
    SAME_TRY(same_use(ctx));
    SAME_TRY(check_index_range(ctx, match, n_m, -1, s->n_r, "match"));
    if (n_m) HIP_TRY(ctx, hipMemcpyAsync(s->match, match, (size_t)n_m * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    return run_orient(s, s->match, out_checked, out_viol_idx, out_nviol, out_flag);
}

int same_orient_sweep_x(same_sweep *s, const double *x_vals, int64_t P, int64_t *out_checked, int32_t *out_viol_idx,
                        int64_t *out_nviol, uint8_t *out_flag, int32_t *out_match, int32_t *out_pair_idx) {
    if (!s) return SAME_EINVAL;
    same_ctx *ctx = s->ctx;
    REQUIRE(ctx, out_checked && out_nviol && P == s->P && (P == 0 || x_vals));
    SAME_TRY(same_use(ctx));
    const int64_t n_m = s->n_m;
    if (P) HIP_TRY(ctx, hipMemcpyAsync(s->x, x_vals, (size_t)P * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if (n_m) {
        hipLaunchKernelGGL(match_init_kernel, dim3(grid_for(n_m)), dim3(256), 0, ctx->stream, s->pidx, n_m);
        if (P) hipLaunchKernelGGL(match_scan_kernel, dim3(grid_for(P)), dim3(256), 0, ctx->stream, s->x, s->pairs, P, s->pidx);
        hipLaunchKernelGGL(match_resolve_kernel, dim3(grid_for(n_m)), dim3(256), 0, ctx->stream, s->pairs, s->pidx, n_m, s->match);
        HIP_TRY(ctx, hipGetLastError());
        if (out_match) SAME_TRY(same_down(ctx, out_match, s->match, (size_t)n_m * sizeof(int32_t)));
        if (out_pair_idx) SAME_TRY(same_down(ctx, out_pair_idx, s->pidx, (size_t)n_m * sizeof(int32_t)));
    }
    return run_orient(s, s->match, out_checked, out_viol_idx, out_nviol, out_flag);
}

int same_xyorder_sweep(same_ctx *ctx, const double *axy, int64_t n_m, const double *rxy, int64_t n_r,
                       const int32_t *tris, int64_t Tr, const int32_t *match, uint8_t *edge_flags, uint8_t *tri_flag,
                       uint8_t *point_flag, int64_t counts[3]) {
    REQUIRE(ctx, ctx && counts);
    REQUIRE(ctx, n_m >= 0 && n_r >= 0 && Tr >= 0 && (n_m == 0 || point_flag));
    counts[0] = counts[1] = counts[2] = 0;
    if (n_m) memset(point_flag, 0, (size_t)n_m);
    if (Tr == 0) return SAME_OK;
    REQUIRE(ctx, axy && tris && match && edge_flags && tri_flag && (n_r == 0 || rxy));
    SAME_TRY(same_use(ctx));
    SAME_TRY(check_index_range(ctx, tris, Tr * 3, 0, n_m, "triangles"));
    SAME_TRY(check_index_range(ctx, match, n_m, -1, n_r, "match"));
    double *dax, *drx;
    int32_t *dtris, *dmatch;
    uint8_t *dedge, *dtf, *dpf;
    unsigned long long *dcnt;
    SAME_TRY(up_as(ctx, SL_AXY, axy, (size_t)n_m * 2, &dax));
    SAME_TRY(up_as(ctx, SL_RXY, rxy, (size_t)n_r * 2, &drx));
    SAME_TRY(up_as(ctx, SL_TRIS, tris, (size_t)Tr * 3, &dtris));
    SAME_TRY(up_as(ctx, SL_MATCH, match, (size_t)n_m, &dmatch));
    SAME_TRY(slot_as(ctx, SL_FLAG0, (size_t)Tr * 3, &dedge));
    SAME_TRY(slot_as(ctx, SL_FLAG1, (size_t)Tr, &dtf));
    SAME_TRY(slot_as(ctx, SL_FLAG2, (size_t)n_m, &dpf));
    SAME_TRY(slot_as(ctx, SL_COUNTS, (size_t)4, &dcnt));
    HIP_TRY(ctx, hipMemsetAsync(dpf, 0, (size_t)n_m, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(dcnt, 0, 4 * sizeof(unsigned long long), ctx->stream));
    hipLaunchKernelGGL(xyorder_kernel, dim3(grid_for(Tr)), dim3(256), 0, ctx->stream, dax, drx, dtris, Tr, dmatch, dedge, dtf,
                       dpf, dcnt);
    HIP_TRY(ctx, hipGetLastError());
    unsigned long long *h = static_cast<unsigned long long *>(ctx->pinned);
    SAME_TRY(same_down(ctx, edge_flags, dedge, (size_t)Tr * 3));
    SAME_TRY(same_down(ctx, tri_flag, dtf, (size_t)Tr));
    SAME_TRY(same_down(ctx, point_flag, dpf, (size_t)n_m));
    SAME_TRY(same_down(ctx, h, dcnt, 3 * sizeof(unsigned long long)));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int q = 0; q < 3; ++q) counts[q] = (int64_t)h[q];
    return SAME_OK;
}

int same_pair_rowmin(same_ctx *ctx, const int32_t *pairs, const double *costs, int64_t P, int64_t n_m,
                     double *out_min) {
    REQUIRE(ctx, ctx != nullptr);
    REQUIRE(ctx, P >= 0 && n_m >= 0);
    if (n_m == 0) return SAME_OK;
    REQUIRE(ctx, out_min && (P == 0 || (pairs && costs)));
    SAME_TRY(same_use(ctx));
    for (int64_t p = 0; p < P; ++p)
        if (pairs[2 * p] < 0 || pairs[2 * p] >= n_m) { ctx->err = "pair row out of range"; return SAME_ERANGE; }
    int32_t *dp;
    double *dc, *dout;
    SAME_TRY(up_as(ctx, SL_PAIRS, pairs, (size_t)P * 2, &dp));
    SAME_TRY(up_as(ctx, SL_X, costs, (size_t)P, &dc));
    SAME_TRY(slot_as(ctx, SL_OUT0, (size_t)n_m, &dout));
    SAME_TRY(same_pair_rowmin_core(ctx, dp, dc, P, n_m, dout));
    SAME_TRY(same_down(ctx, out_min, dout, (size_t)n_m * sizeof(double)));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SAME_OK;
}

int same_assign_matrix(same_ctx *ctx, const int32_t *pairs, const double *costs, int64_t P, const double *unmatched,
                       int64_t n_m, int64_t n_r, double big_m, double *out) {
    REQUIRE(ctx, ctx != nullptr);
    REQUIRE(ctx, P >= 0 && n_m >= 0 && n_r >= 0);
    if (n_m == 0) return SAME_OK;
    REQUIRE(ctx, out && unmatched && (P == 0 || (pairs && costs)));
    SAME_TRY(same_use(ctx));
    const int64_t ld = n_r + n_m, total = n_m * ld;
    // duplicate (i, j) entries: the reference's scatter keeps the last one; resolve on the host index-wise
    std::vector<int32_t> winner((size_t)P);
    {
        // pairs are nearly always unique; detect duplicates with a sort-free pass over a hash of (i, j)
        std::vector<int64_t> key((size_t)P);
        for (int64_t p = 0; p < P; ++p) {
            if (pairs[2 * p] < 0 || pairs[2 * p] >= n_m || pairs[2 * p + 1] < 0 || pairs[2 * p + 1] >= n_r) {
                ctx->err = "pair index out of range";
                return SAME_ERANGE;
            }
            key[p] = (int64_t)pairs[2 * p] * n_r + pairs[2 * p + 1];
            winner[p] = (int32_t)p;
        }
        std::vector<int64_t> order((size_t)P);
        for (int64_t p = 0; p < P; ++p) order[p] = p;
        bool sorted = true;
        for (int64_t p = 1; p < P && sorted; ++p) sorted = key[p - 1] < key[p];
        if (!sorted) {
            std::vector<std::pair<int64_t, int64_t>> kv((size_t)P);
            for (int64_t p = 0; p < P; ++p) kv[p] = {key[p], p};
            std::sort(kv.begin(), kv.end());
            for (int64_t p = 0; p + 1 < P; ++p)
                if (kv[p].first == kv[p + 1].first) winner[kv[p].second] = -1;  // a later duplicate exists
        }
    }
    int32_t *dp, *dwin;
    double *dc, *dun, *dout;
    SAME_TRY(up_as(ctx, SL_PAIRS, pairs, (size_t)P * 2, &dp));
    SAME_TRY(up_as(ctx, SL_MATCH, winner.data(), (size_t)P, &dwin));
    SAME_TRY(up_as(ctx, SL_X, costs, (size_t)P, &dc));
    SAME_TRY(up_as(ctx, SL_SIZE, unmatched, (size_t)n_m, &dun));
    SAME_TRY(slot_as(ctx, SL_OUT0, (size_t)total, &dout));
    hipLaunchKernelGGL(fill_f64_kernel, dim3(grid_for(ceil_div(total, 2))), dim3(256), 0, ctx->stream, dout, total, big_m);
    if (P) hipLaunchKernelGGL(scatter_pairs_kernel, dim3(grid_for(P)), dim3(256), 0, ctx->stream, dp, dc, P, ld, dout, dwin);
    hipLaunchKernelGGL(diag_kernel, dim3(grid_for(n_m)), dim3(256), 0, ctx->stream, dun, n_m, n_r, ld, dout);
    HIP_TRY(ctx, hipGetLastError());
    SAME_TRY(same_down(ctx, out, dout, (size_t)total * sizeof(double)));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SAME_OK;
}

int same_window_count(same_ctx *ctx, const double *xy, int64_t n, const double *boxes, int64_t n_boxes,
                      int64_t *out_count, uint8_t *out_mask) {
    REQUIRE(ctx, ctx != nullptr);
    REQUIRE(ctx, n >= 0 && n_boxes >= 0);
    if (n_boxes == 0) return SAME_OK;
    REQUIRE(ctx, boxes && out_count && (n == 0 || xy));
    for (int64_t b = 0; b < n_boxes; ++b) out_count[b] = 0;
    if (n == 0) return SAME_OK;
    SAME_TRY(same_use(ctx));
    double *dxy, *dbox;
    unsigned long long *dcnt;
    uint8_t *dmask = nullptr;
    SAME_TRY(up_as(ctx, SL_AXY, xy, (size_t)n * 2, &dxy));
    SAME_TRY(up_as(ctx, SL_X, boxes, (size_t)n_boxes * 4, &dbox));
    SAME_TRY(slot_as(ctx, SL_MASK, (size_t)n_boxes, &dcnt));
    if (out_mask) SAME_TRY(slot_as(ctx, SL_FLAG0, (size_t)n_boxes * n, &dmask));
    HIP_TRY(ctx, hipMemsetAsync(dcnt, 0, (size_t)n_boxes * sizeof(unsigned long long), ctx->stream));
    hipLaunchKernelGGL(window_count_kernel, dim3(grid_for(n)), dim3(256), 0, ctx->stream, dxy, n, dbox, n_boxes, dcnt, dmask);
    HIP_TRY(ctx, hipGetLastError());
    SAME_TRY(same_down(ctx, out_count, dcnt, (size_t)n_boxes * sizeof(unsigned long long)));
    if (out_mask) SAME_TRY(same_down(ctx, out_mask, dmask, (size_t)n_boxes * n));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SAME_OK;
}

// ---- device-resident forms -------------------------------------------------------------------
// dcounts: 3 x uint64 in HBM {comparisons, violations, violated triangles}; dpoint_flag is zeroed here.
int same_xyorder_sweep_dev(same_ctx *ctx, const double *daxy, int64_t n_m, const double *drxy, const int32_t *dtris,
                           int64_t Tr, const int32_t *dmatch, uint8_t *dedge_flags, uint8_t *dtri_flag,
                           uint8_t *dpoint_flag, uint64_t *dcounts) {
    REQUIRE(ctx, ctx && Tr >= 0 && n_m >= 0 && dcounts && (n_m == 0 || dpoint_flag));
    SAME_TRY(same_use(ctx));
    if (n_m) HIP_TRY(ctx, hipMemsetAsync(dpoint_flag, 0, (size_t)n_m, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(dcounts, 0, 3 * sizeof(uint64_t), ctx->stream));
    if (Tr == 0) return SAME_OK;
    REQUIRE(ctx, daxy && drxy && dtris && dmatch && dedge_flags && dtri_flag);
    hipLaunchKernelGGL(xyorder_kernel, dim3(grid_for(Tr)), dim3(256), 0, ctx->stream, daxy, drxy, dtris, Tr, dmatch, dedge_flags,
                       dtri_flag, dpoint_flag, reinterpret_cast<unsigned long long *>(dcounts));
    HIP_TRY(ctx, hipGetLastError());
    return SAME_OK;
}

// orientation sweep on the bound state with a match vector that is already on the device
int same_orient_sweep_dev(same_sweep *s, const int32_t *dmatch, int64_t *out_checked, int32_t *out_viol_idx,
                          int64_t *out_nviol) {
    if (!s) return SAME_EINVAL;
    REQUIRE(s->ctx, out_checked && out_nviol && (s->n_m == 0 || dmatch));
    SAME_TRY(same_use(s->ctx));
    return run_orient(s, dmatch, out_checked, out_viol_idx, out_nviol, nullptr);
}

// triangle-block form for the sharded sweep: flags of [t_begin, t_end) only, enqueued
int same_orient_flags_dev(same_sweep *s, const int32_t *dmatch, int64_t t_begin, int64_t t_end, uint8_t *dflag) {
    if (!s) return SAME_EINVAL;
    same_ctx *ctx = s->ctx;
    REQUIRE(ctx, t_begin >= 0 && t_begin <= t_end && t_end <= s->Tr && (t_begin == t_end || (dmatch && dflag)));
    SAME_TRY(same_use(ctx));
    return launch_orient_flags(s, dmatch, t_begin, t_end, dflag);
}

int same_orient_from_flags_dev(same_sweep *s, const uint8_t *dflag, int64_t *out_checked, int32_t *out_viol_idx,
                               int64_t *out_nviol) {
    if (!s) return SAME_EINVAL;
    REQUIRE(s->ctx, out_checked && out_nviol && (s->Tr == 0 || dflag));
    SAME_TRY(same_use(s->ctx));
    *out_checked = 0;
    *out_nviol = 0;
    if (s->Tr == 0) return SAME_OK;
    return compact_from_flags(s, dflag, out_checked, out_viol_idx, out_nviol);
}

int same_first_candidate_dev(same_ctx *ctx, const int32_t *didx, int64_t rows, int k, int32_t *dmatch) {
    REQUIRE(ctx, ctx && rows >= 0 && k >= 1 && (rows == 0 || (didx && dmatch)));
    SAME_TRY(same_use(ctx));
    if (rows == 0) return SAME_OK;
    hipLaunchKernelGGL(first_candidate_kernel, dim3(grid_for(rows)), dim3(256), 0, ctx->stream, didx, rows, k, dmatch);
    HIP_TRY(ctx, hipGetLastError());
    return SAME_OK;
}

}  // extern "C"
