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
#include "common.h"

namespace {

constexpr int KNN_RW = 8;      // aligned rows per wave
constexpr int KNN_WAVES = 4;   // waves per block
constexpr int KNN_CAP = 128;   // candidate slots per row; needs k + 64 <= KNN_CAP

struct RowList {
    double d2[KNN_CAP];
    int32_t j[KNN_CAP];
};

// rank of candidate (d, j) among the m candidates of `L` under (d2, j) ascending
__device__ __forceinline__ int rank_of(const RowList &L, int m, double d, int32_t j) {
    int rank = 0;
    for (int q = 0; q < m; ++q) {
        const double dq = L.d2[q];   // uniform address: LDS broadcast
        const int32_t jq = L.j[q];
        rank += (dq < d) || (dq == d && jq < j);
    }
    return rank;
}

// keep the best min(m, k) candidates, sorted, in slots [0, min(m,k))
__device__ __forceinline__ int prune_in_place(RowList &L, int m, int k, int lane) {
    double d[KNN_CAP / 64];
    int32_t j[KNN_CAP / 64];
    int rk[KNN_CAP / 64];
#pragma unroll
    for (int p = 0; p < KNN_CAP / 64; ++p) {
        const int c = lane + 64 * p;
        rk[p] = KNN_CAP;
        if (c < m) {
            d[p] = L.d2[c];
            j[p] = L.j[c];
            rk[p] = rank_of(L, m, d[p], j[p]);
        }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): every read above has landed before the overwrite
#pragma unroll
    for (int p = 0; p < KNN_CAP / 64; ++p)
        if (rk[p] < k) { L.d2[rk[p]] = d[p]; L.j[rk[p]] = j[p]; }
    return m < k ? m : k;
}

__global__ __launch_bounds__(64 * KNN_WAVES) void knn_prune_kernel(
    const double *__restrict__ axy, const double *__restrict__ rxy, int64_t n_r, int64_t row_begin,
    int64_t row_end, double r2, int k, int32_t *__restrict__ out_idx, double *__restrict__ out_d2,
    int32_t *__restrict__ out_cnt) {
    __shared__ RowList lists[KNN_WAVES * KNN_RW];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t row0 = row_begin + ((int64_t)blockIdx.x * KNN_WAVES + wave) * KNN_RW;
    if (row0 >= row_end) return;
    RowList *L = lists + wave * KNN_RW;

    double ax[KNN_RW], ay[KNN_RW];
    int cnt[KNN_RW];
#pragma unroll
    for (int r = 0; r < KNN_RW; ++r) {
        const int64_t i = (row0 + r < row_end) ? row0 + r : row_end - 1;  // tail rows repeat the last one
        ax[r] = axy[2 * i];
        ay[r] = axy[2 * i + 1];
        cnt[r] = 0;
    }

    typedef double double2_t __attribute__((ext_vector_type(2)));
    for (int64_t base = 0; base < n_r; base += 64) {
        const int64_t j = base + lane;
        const bool valid = j < n_r;
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

int launch_knn(same_ctx *ctx, const double *daxy, const double *drxy, int64_t n_r, int64_t rb, int64_t re,
               double radius, int k, int32_t *didx, double *dd2, int32_t *dcnt) {
    const int64_t rows = re - rb;
    if (rows == 0) return SAME_OK;
    const int64_t blocks = ceil_div(rows, KNN_WAVES * KNN_RW);
    REQUIRE(ctx, blocks < (int64_t)1 << 31);
    hipLaunchKernelGGL(knn_prune_kernel, dim3((unsigned)blocks), dim3(64 * KNN_WAVES), 0, ctx->stream, daxy, drxy, n_r, rb, re,
                       radius * radius, k, didx, dd2, dcnt);
    HIP_TRY(ctx, hipGetLastError());
    return SAME_OK;
}

}  // namespace

extern "C" {

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
