// merge.hip -- the de-duplication half of merge_window_matches_unique_ref.
//
// Reference: src/helpers.py:745-753 -- the per-window match tables, concatenated, are stably sorted by
// (filtered_violation, window_id) and only the first row of every (aligned id, ref id) pair is kept: of the
// proposals several overlapping windows make for one pair, the non-violating one wins, then the smaller window id,
// then the earlier row.  What follows in the reference (the Hopcroft-Karp matching on the kept rows, :755-815) is a
// sequential graph algorithm on a table that is by then a fraction of the size and stays on the host (same_amd/merge.py).
//
// Device form (all integer work, a few hundred KB of traffic -- latency-bound, no roofline to speak of):
//   1. key[i] = viol << 63 | window_id << 32 | i         one 64-bit key per row; the row index in the low half makes
//                                                         the order total, so ANY correct sort of the keys IS the
//                                                         reference's stable mergesort
//   2. bitonic sort of the keys (padded to a power of two with ~0): strides >= 2048 as one global compare-exchange
//      pass each, all smaller strides of a stage inside LDS (2048 keys per workgroup)
//   3. first[pair] = min over sorted positions s of rows with that pair: open-addressing table keyed by
//      aligned_code << 32 | ref_code, 64-bit CAS to claim a slot, atomicMin on the position
//   4. keep[s] = (s == first[pair of row at s]); ordered compaction of the surviving ROW INDICES -> out_rows (one launch:
//      multi-block look-back scan, scan.h), in the reference's post-drop_duplicates order.
#include <algorithm>

#include "common.h"
#include "scan.h"

namespace {

constexpr int SORT_BLOCK = 2048;   // keys one workgroup sorts in LDS (16 KB)

__global__ __launch_bounds__(256) void merge_key_kernel(const uint8_t *__restrict__ viol, unsigned viol_mask, const int32_t *__restrict__ window, int64_t n,
                                                          int64_t n_pad, unsigned long long *__restrict__ key) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pad) return;
    key[i] = i < n ? ((unsigned long long)((viol[i] & viol_mask) ? 1 : 0) << 63) | ((unsigned long long)(uint32_t)window[i] << 32) | (unsigned long long)i
                   : ~0ull;        // padding sorts last (a real key never has all bits set: i < 2^31)
}

// one compare-exchange pass of the bitonic network at stride j of stage k (both powers of two), in global memory
__global__ __launch_bounds__(256) void bitonic_global_kernel(unsigned long long *__restrict__ key, int64_t n_pad, int64_t j, int64_t k) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per pair
    if (t >= n_pad / 2) return;
    const int64_t lo = ((t / j) * 2 * j) + (t % j), hi = lo + j;
    const bool up = (lo & k) == 0;
    const unsigned long long a = key[lo], b = key[hi];
    if ((a > b) == up) { key[lo] = b; key[hi] = a; }
}

// every pass of stage k whose stride fits one workgroup's 2048 keys (j <= j_top <= 1024), in LDS
__global__ __launch_bounds__(1024) void bitonic_lds_kernel(unsigned long long *__restrict__ key, int64_t k, int j_top) {
    __shared__ unsigned long long s[SORT_BLOCK];
    const int64_t base = (int64_t)blockIdx.x * SORT_BLOCK;
    s[threadIdx.x] = key[base + threadIdx.x];
    s[threadIdx.x + 1024] = key[base + threadIdx.x + 1024];
    __syncthreads();
    for (int j = j_top; j >= 1; j >>= 1) {
        const int t = threadIdx.x;
        const int lo = ((t / j) * 2 * j) + (t % j), hi = lo + j;
        const bool up = ((base + lo) & k) == 0;
        const unsigned long long a = s[lo], b = s[hi];
        if ((a > b) == up) { s[lo] = b; s[hi] = a; }
        __syncthreads();
    }
    key[base + threadIdx.x] = s[threadIdx.x];
    key[base + threadIdx.x + 1024] = s[threadIdx.x + 1024];
}

// the first log2(2048) stages entirely in LDS: sorts every run of 2048 keys, alternating direction as the network needs
__global__ __launch_bounds__(1024) void bitonic_lds_head_kernel(unsigned long long *__restrict__ key) {
    __shared__ unsigned long long s[SORT_BLOCK];
    const int64_t base = (int64_t)blockIdx.x * SORT_BLOCK;
    s[threadIdx.x] = key[base + threadIdx.x];
    s[threadIdx.x + 1024] = key[base + threadIdx.x + 1024];
    __syncthreads();
    for (int k = 2; k <= SORT_BLOCK; k <<= 1)
        for (int j = k >> 1; j >= 1; j >>= 1) {
            const int t = threadIdx.x;
            const int lo = ((t / j) * 2 * j) + (t % j), hi = lo + j;
            const bool up = ((base + lo) & k) == 0;
            const unsigned long long a = s[lo], b = s[hi];
            if ((a > b) == up) { s[lo] = b; s[hi] = a; }
            __syncthreads();
        }
    key[base + threadIdx.x] = s[threadIdx.x];
    key[base + threadIdx.x + 1024] = s[threadIdx.x + 1024];
}

__device__ __forceinline__ uint64_t mix64(uint64_t x) {   // splitmix64 finaliser: pair codes are dense small integers
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27; x *= 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}

constexpr unsigned long long EMPTY = ~0ull;

// slot of `pair` in the table (claiming one if it is new); the table has at least twice as many slots as rows
__device__ __forceinline__ int64_t pair_slot(unsigned long long *__restrict__ tkey, int64_t mask, unsigned long long pair) {
    int64_t h = (int64_t)(mix64(pair) & (uint64_t)mask);
    for (;;) {
        const unsigned long long seen = atomicCAS(&tkey[h], EMPTY, pair);
        if (seen == EMPTY || seen == pair) return h;
        h = (h + 1) & mask;
    }
}

__global__ __launch_bounds__(256) void merge_first_kernel(const unsigned long long *__restrict__ key, int64_t n, const int32_t *__restrict__ a_code,
                                                            const int32_t *__restrict__ r_code, unsigned long long *__restrict__ tkey,
                                                            unsigned int *__restrict__ tfirst, int64_t mask) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // sorted position
    if (s >= n) return;
    const uint32_t row = (uint32_t)key[s];
    const unsigned long long pair = ((unsigned long long)(uint32_t)a_code[row] << 32) | (uint32_t)r_code[row];
    atomicMin(&tfirst[pair_slot(tkey, mask, pair)], (unsigned int)s);
}

// keep[s] = (s is the first sorted position of its pair); the surviving ROW INDICES in sorted order, their number: one launch over
// many blocks (look-back scan, scan.h; a block that has to recompute a predecessor's count probes the finished table again)
__global__ __launch_bounds__(scan::NT) void merge_compact_kernel(const unsigned long long *__restrict__ key, int64_t n, const int32_t *__restrict__ a_code,
                                                                  const int32_t *__restrict__ r_code, unsigned long long *__restrict__ tkey,
                                                                  const unsigned int *__restrict__ tfirst, int64_t mask,
                                                                  unsigned long long *__restrict__ status, int32_t *__restrict__ out_rows,
                                                                  unsigned long long *__restrict__ out_total) {
    __shared__ scan::Shared sh;
    auto keep_of = [&](int64_t q) -> bool {
        if (q >= n) return false;
        const uint32_t row = (uint32_t)key[q];
        const unsigned long long pair = ((unsigned long long)(uint32_t)a_code[row] << 32) | (uint32_t)r_code[row];
        return tfirst[pair_slot(tkey, mask, pair)] == (unsigned int)q;
    };
    const int64_t s0 = (int64_t)blockIdx.x * scan::NT + threadIdx.x;
    const bool keep = keep_of(s0);
    auto val = [&](int64_t q) { return scan::Pair{(q == s0 ? keep : keep_of(q)) ? 1u : 0u, 0u}; };
    scan::Pair through;
    const scan::Pair off = scan::exclusive(status, (int)blockIdx.x, val, sh, &through);
    if (keep) out_rows[off.a] = (int32_t)(uint32_t)key[s0];
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *out_total = through.a;
}

}  // namespace

// Ascending sort of n_pad 64-bit keys on the device (n_pad a power of two >= 2048; pad with ~0).  Enqueue only.  Shared with the
// section index of the window path (window.hip), declared in common.h.
int same_sort_u64_core(same_ctx *ctx, unsigned long long *dkey, int64_t n_pad) {
    REQUIRE(ctx, n_pad >= SORT_BLOCK && (n_pad & (n_pad - 1)) == 0);
    hipLaunchKernelGGL(bitonic_lds_head_kernel, dim3((unsigned)(n_pad / SORT_BLOCK)), dim3(1024), 0, ctx->stream, dkey);
    for (int64_t k = 2 * SORT_BLOCK; k <= n_pad; k <<= 1) {
        for (int64_t j = k >> 1; j >= SORT_BLOCK; j >>= 1)
            hipLaunchKernelGGL(bitonic_global_kernel, dim3((unsigned)ceil_div(n_pad / 2, 256)), dim3(256), 0, ctx->stream, dkey, n_pad, j, k);
        hipLaunchKernelGGL(bitonic_lds_kernel, dim3((unsigned)(n_pad / SORT_BLOCK)), dim3(1024), 0, ctx->stream, dkey, k, SORT_BLOCK / 2);
    }
    HIP_TRY(ctx, hipGetLastError());
    return SAME_OK;
}

// The de-duplication on DEVICE arrays, enqueue only (the host-buffer entry point below; the merge accumulator of the window path,
// window_merge.hip): rows whose (aligned code, ref code) pair is the first of its kind in the order (viol & viol_mask != 0, window id,
// row) -> dout (row indices, in that order), their number -> *dtotal.  Scratch: the context's slots SL_X, SL_OUT0, SL_OUT1, SL_MASK.
int same_merge_dedup_core(same_ctx *ctx, const uint8_t *dviol, unsigned viol_mask, const int32_t *dwin, const int32_t *da, const int32_t *dr,
                          int64_t n, int32_t *dout, unsigned long long *dtotal) {
    REQUIRE(ctx, n > 0 && n < ((int64_t)1 << 30));
    int64_t n_pad = SORT_BLOCK;
    while (n_pad < n) n_pad <<= 1;
    int64_t slots = 2;
    while (slots < 2 * n) slots <<= 1;
    unsigned long long *dkey, *dtkey, *dstatus;
    unsigned int *dtfirst;
    const size_t st_words = scan::status_bytes(n) / 8;
    SAME_TRY(slot_as(ctx, SL_X, (size_t)n_pad, &dkey));
    SAME_TRY(slot_as(ctx, SL_OUT0, (size_t)slots, &dtkey));
    SAME_TRY(slot_as(ctx, SL_OUT1, (size_t)slots, &dtfirst));
    SAME_TRY(slot_as(ctx, SL_MASK, st_words, &dstatus));
    SAME_FILL(ctx, dtkey, 0xFF, (size_t)slots * 8);
    SAME_FILL(ctx, dtfirst, 0xFF, (size_t)slots * 4);
    SAME_FILL(ctx, dstatus, 0, st_words * 8);
    SAME_LAUNCH(ctx, merge_key_kernel, dim3((unsigned)ceil_div(n_pad, 256)), dim3(256), 0, dviol, viol_mask, dwin, n, n_pad, dkey);
    SAME_TRY(same_sort_u64_core(ctx, dkey, n_pad));
    const unsigned grid = (unsigned)ceil_div(n, 256);
    SAME_LAUNCH(ctx, merge_first_kernel, dim3(grid), dim3(256), 0, dkey, n, da, dr, dtkey, dtfirst, slots - 1);
    SAME_LAUNCH(ctx, merge_compact_kernel, dim3(scan::blocks_for(n)), dim3(scan::NT), 0, dkey, n, da, dr, dtkey, dtfirst, slots - 1,
                scan::arg(dstatus), dout, dtotal);
    HIP_TRY(ctx, hipGetLastError());
    return SAME_OK;
}

extern "C" int same_merge_dedup(same_ctx *ctx, const uint8_t *viol, const int32_t *window_id, const int32_t *aligned_code,
                                const int32_t *ref_code, int64_t n, int32_t *out_rows, int64_t *out_n) {
    REQUIRE(ctx, ctx && out_n && n >= 0 && n < ((int64_t)1 << 30));   // row indices and sorted positions are 32-bit, scan totals 31-bit; 2^30 rows need ~45 GB of scratch
    *out_n = 0;
    if (n == 0) return SAME_OK;
    REQUIRE(ctx, viol && window_id && aligned_code && ref_code && out_rows);
    for (int64_t i = 0; i < n; ++i)
        if (window_id[i] < 0 || aligned_code[i] < 0 || ref_code[i] < 0) {
            ctx->err = "same_merge_dedup: window ids and id codes must be non-negative";
            return SAME_ERANGE;
        }
    SAME_TRY(same_use(ctx));
    uint8_t *dviol;
    int32_t *dwin, *da, *dr, *dout;
    unsigned long long *dtotal;
    SAME_TRY(up_as(ctx, SL_FLAG0, viol, (size_t)n, &dviol));
    SAME_TRY(up_as(ctx, SL_PAIRS, window_id, (size_t)n, &dwin));
    SAME_TRY(up_as(ctx, SL_MATCH, aligned_code, (size_t)n, &da));
    SAME_TRY(up_as(ctx, SL_TRIS, ref_code, (size_t)n, &dr));
    SAME_TRY(slot_as(ctx, SL_OUT2, (size_t)n + 4, &dout));        // the rows, then (8-byte aligned) their number
    dtotal = reinterpret_cast<unsigned long long *>(dout + (((size_t)n + 1) & ~size_t(1)));
    SAME_TRY(same_merge_dedup_core(ctx, dviol, 0xFFu, dwin, da, dr, n, dout, dtotal));
    unsigned long long total = 0;
    SAME_TRY(same_down(ctx, &total, dtotal, sizeof total));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    REQUIRE(ctx, (int64_t)total <= n);
    SAME_TRY(same_down(ctx, out_rows, dout, (size_t)total * sizeof(int32_t)));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *out_n = (int64_t)total;
    return SAME_OK;
}
