// cost.hip -- L1 cell-type + coordinate cost kernels (SURVEY 8a4 / dense generalisation).
//
//   c(i,j) = w * sum_t |A[i,t] - R[j,t]|  +  (w*0.001) * (|ax-rx| + |ay-ry|)
//
// Reference: src/same.py:1180-1189 (per pair), src/init_helpers.py:151-155 (the dense matrix
// these costs are scattered into).  Parity contract: the type sum is a left-to-right fp64
// accumulation and nothing is fused (-ffp-contract=off), so results are bit-identical to
// the reference's object-dtype pandas row arithmetic.
//
// Dense kernel design (MI355X): L1 distance is not a contraction, so MFMA does not apply;
// the kernel is a store stream (8 B out per 2T+6 fp64 VALU ops) that sits near the fp64
// VALU / HBM ridge at T=20.  Each lane OWNS CPL adjacent ref columns and keeps their T type
// values + XY in VGPRs for the whole row chunk; the aligned row (T+2 values) is wave-uniform
// and is fetched with scalar loads into SGPRs, so the inner loop is exactly
//   v_add_f64 d, s[a_t], -v[r_t]  ;  v_add_f64 acc, acc, |d|
// with no LDS or vector-memory instruction per element.  Every lane stores 16 B per row
// (CPL=2 doubles / 4 floats): one wave-instruction writes 1 KiB of one output row, a
// workgroup 4 KiB.  Bound (MI355X, T=20): the board power cap -- the VALU is 92-94 % busy with fp64
// (80 % with fp32) at the 1.76-1.9 GHz the board holds under its 1400 W cap, and idling the VALU
// only raises the clock (profiles/r03_dense_occupancy.log); traffic is 1.004x the algorithmic bytes.
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace {

template <typename F> struct vec_of;
template <> struct vec_of<double> { static constexpr int cpl = 2; };
template <> struct vec_of<float> { static constexpr int cpl = 4; };

template <typename F> __device__ __forceinline__ F absf(F x);
template <> __device__ __forceinline__ double absf<double>(double x) { return __builtin_fabs(x); }
template <> __device__ __forceinline__ float absf<float>(float x) { return __builtin_fabsf(x); }

// Block = 256 threads = 4 waves side by side over 256*CPL ref columns; it sweeps `rows_per_block`
// aligned rows.  What keeps the fp64 VALU and the store stream overlapped:
//
// (1) Scalar-load pipelining.  SMEM returns out of order, so the only usable wait is lgkmcnt(0);
//     the row is consumed in two halves and each half is fetched one phase ahead, always in the
//     order  wait(current half) -> issue(next half) -> compute.  `touch` is an empty asm that
//     merely *uses* one SGPR of the current half, which makes the compiler place its s_waitcnt
//     there; the sched_barriers keep the next half's s_loads from moving above it; the loop body
//     is a single basic block so nothing is sunk out of place.  Waits stay compiler-generated.
// (2) The store address is a scalar row pointer plus a fixed per-lane byte offset, so no address VGPR
//     is rewritten per row and no VALU instruction is spent on addressing.
// (3) Block -> tile map (map_mode).  2: blocks that share an XCD (blockIdx % 8) walk adjacent column
//     tiles of the same row chunk -- every XCD streams whole output rows, the best store rate (store-bound
//     shapes).  4: XCD x owns a contiguous range of column tiles for the whole launch, so its share of R
//     stays in its own L2 (shapes bound by fp64 issue, where the store pattern no longer matters and the
//     R re-fetch does: FETCH_SIZE 3.46 GB -> 0.15 GB per launch).  0: column tile fastest, the plain map
//     of the scalar-store fallback.
// Variants that were measured and did not ship (two column groups per lane, DEPTH > 1 result sets, an LDS pad to cap
// the occupancy, maps 1 / 3 / 5, non-nt store policies) are in the history of this file: tools/probes/README.md.
__device__ __forceinline__ void touch(double v) { asm volatile("" ::"s"(v)); }
__device__ __forceinline__ void touch(float v) { asm volatile("" ::"s"(v)); }

// 16-byte nontemporal store, address = scalar row pointer + fixed per-lane byte offset (hipcc otherwise keeps a
// 64-bit VGPR pointer and bumps it with a v_lshl_add_u64 per row).  The trailing `s_nop 1` gives the 2 wait states
// gfx940+ requires between a store of more than 64 bits and a VALU write of its data VGPRs; the compiler only
// inserts them for its own instructions.
template <typename V16>
__device__ __forceinline__ void store16_nt_saddr(char *row_uniform, unsigned lane_byte_off, V16 v) {
    typedef int i4 __attribute__((ext_vector_type(4)));
    static_assert(sizeof(V16) == 16, "16-byte vector expected");
    const i4 bits = __builtin_bit_cast(i4, v);
    asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" ::"v"(lane_byte_off), "v"(bits), "s"(row_uniform) : "memory");
}

enum : int { MAP_TILE_FASTEST = 0, MAP_XCD_ROWS = 2, MAP_XCD_TILES = 4 };

template <typename F, int T, int CPL, bool VEC_STORE, bool W1>
__global__ __launch_bounds__(256) void dense_cost_kernel(
    const F *__restrict__ A, const F *__restrict__ R, const F *__restrict__ axy,
    const F *__restrict__ rxy, int64_t n_r, int64_t row_begin, int64_t row_end, F w, F dcoef,
    F *__restrict__ out, int64_t ld, int col_tiles, int rows_per_block, int64_t n_store, int map_mode,
    int row_chunks) {
    // VEC_STORE: every lane with j0 < n_store stores all CPL columns with one 16 B store (n_store is a
    // multiple of CPL; columns in [n_r, n_store) are caller-owned padding).  Host guarantees
    // rows_per_block <= row_end - row_begin; the last chunk is shifted back to overlap its neighbour
    // instead of being ragged (the overlap recomputes identical values).
    static_assert(!VEC_STORE || sizeof(F) * CPL == 16, "the vector store is the 16-byte scalar-base one");
    constexpr int H = (T + 1) / 2;  // first half: a[0..H); second half: a[H..T) + XY
    constexpr int TT = T > 0 ? T : 1;
    typedef F vecF __attribute__((ext_vector_type(CPL)));
    int tile, chunk;
    if (map_mode == MAP_TILE_FASTEST) {
        tile = blockIdx.x % col_tiles;
        chunk = blockIdx.x / col_tiles;
    } else if (map_mode == MAP_XCD_TILES) {  // XCD x (= blockIdx % 8, the hardware's round-robin) owns the tiles [x*ct/8, (x+1)*ct/8)
        const unsigned b = blockIdx.x, xcd = b & 7u, k = b >> 3;
        const unsigned lo = xcd * (unsigned)col_tiles / 8u, hi = (xcd + 1u) * (unsigned)col_tiles / 8u;
        const unsigned widest = ((unsigned)col_tiles + 7u) >> 3;      // the grid is sized for the widest owner
        const unsigned m = k % widest;
        chunk = k / widest;
        if (m >= hi - lo || chunk >= row_chunks) return;
        tile = lo + m;
    } else {  // MAP_XCD_ROWS: blocks that share an XCD take adjacent column tiles
        const unsigned b = blockIdx.x, xcd = b & 7u, k = b >> 3;
        const unsigned per = (gridDim.x + 7u) >> 3;  // blocks per XCD group
        const unsigned lin = xcd * per + k;          // may exceed the tile count: such blocks exit
        tile = lin % col_tiles;
        chunk = lin / col_tiles;
        if (chunk >= row_chunks) return;
    }
    const int64_t j0 = (((int64_t)tile * 4 + (threadIdx.x >> 6)) * 64 + (threadIdx.x & 63)) * CPL;
    int64_t i0 = row_begin + (int64_t)chunk * rows_per_block;
    if (i0 + rows_per_block > row_end) i0 = row_end - rows_per_block;

    F r[CPL][TT];
    F rx[CPL], ry[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        int64_t j = j0 + c;
        if (j >= n_r) j = n_r - 1;  // clamp: lanes past the edge compute a valid column
        const F *rp = R + j * T;
#pragma unroll
        for (int t = 0; t < T; ++t) r[c][t] = rp[t];
        rx[c] = rxy[2 * j];
        ry[c] = rxy[2 * j + 1];
    }
    if (j0 >= (VEC_STORE ? n_store : n_r)) return;

    F h0[H > 0 ? H : 1];
    const F *__restrict__ arow = A + i0 * T;  // wave-uniform -> scalar loads
    const F *__restrict__ axyrow = axy + 2 * i0;
#pragma unroll
    for (int t = 0; t < H; ++t) h0[t] = arow[t];
    char *orow = reinterpret_cast<char *>(out + (i0 - row_begin) * ld);  // wave-uniform row pointer
    const unsigned lane_off = (unsigned)(j0 * sizeof(F));              // fixed per-lane byte offset (< 4 GiB rows)
    const int64_t row_pitch = ld * (int64_t)sizeof(F);
    for (int q = 0; q < rows_per_block; ++q) {
        // ---- phase 0: wait(h0) -> issue(second half of this row) -> compute t in [0,H)
        if (H > 0) touch(h0[0]);
        __builtin_amdgcn_sched_barrier(0);
        F h1[T - H > 0 ? T - H : 1];
#pragma unroll
        for (int t = H; t < T; ++t) h1[t - H] = arow[t];
        const F ax = axyrow[0], ay = axyrow[1];
        __builtin_amdgcn_sched_barrier(0);
        // columns interleaved per type: each dependent add sits CPL instructions after its subtract
        F s[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) s[c] = F(0);
#pragma unroll
        for (int t = 0; t < H; ++t) {
            F dd[CPL];
#pragma unroll
            for (int c = 0; c < CPL; ++c) dd[c] = h0[t] - r[c][t];
#pragma unroll
            for (int c = 0; c < CPL; ++c) s[c] = s[c] + absf<F>(dd[c]);
        }
        // ---- phase 1: wait(h1, xy) -> issue(first half of the next row) -> compute t in [H,T), XY, store
        touch(ax);
        __builtin_amdgcn_sched_barrier(0);
        {
            const bool last = (q + 1 >= rows_per_block);
            const F *__restrict__ an = last ? arow : arow + T;  // last row re-reads itself (harmless)
#pragma unroll
            for (int t = 0; t < H; ++t) h0[t] = an[t];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = H; t < T; ++t) {
            F dd[CPL];
#pragma unroll
            for (int c = 0; c < CPL; ++c) dd[c] = h1[t - H] - r[c][t];
#pragma unroll
            for (int c = 0; c < CPL; ++c) s[c] = s[c] + absf<F>(dd[c]);
        }
        F v[CPL];
        {
            F dx[CPL], dy[CPL], dc[CPL], ws[CPL];
#pragma unroll
            for (int c = 0; c < CPL; ++c) dx[c] = ax - rx[c];
#pragma unroll
            for (int c = 0; c < CPL; ++c) dy[c] = ay - ry[c];
#pragma unroll
            for (int c = 0; c < CPL; ++c) dc[c] = absf<F>(dx[c]) + absf<F>(dy[c]);
#pragma unroll
            for (int c = 0; c < CPL; ++c) ws[c] = W1 ? s[c] : w * s[c];  // 1.0*s == s exactly: the multiply is skipped, not approximated
#pragma unroll
            for (int c = 0; c < CPL; ++c) dc[c] = dcoef * dc[c];
#pragma unroll
            for (int c = 0; c < CPL; ++c) v[c] = ws[c] + dc[c];
        }
        if constexpr (VEC_STORE) {
            vecF res;
#pragma unroll
            for (int c = 0; c < CPL; ++c) res[c] = v[c];
            store16_nt_saddr(orow, lane_off, res);
        } else {
            F *dst = reinterpret_cast<F *>(orow + lane_off);
#pragma unroll
            for (int c = 0; c < CPL; ++c)
                if (j0 + c < n_r) dst[c] = v[c];
        }
        orow += row_pitch;
        arow += T;
        axyrow += 2;
    }
}

// Any T (used above the unrolled range, T > 48, where a lane can no longer keep its column's type values in registers).
// The roles are turned round: a lane owns ONE reference column and carries RB = 32 running sums, one per aligned row of
// the block's row chunk; the type axis is walked in pieces of 8 -- the lane loads its column's 8 values once per piece
// (one 64-byte run) and every row of the chunk consumes them, the row's own 8 values arriving as one wave-uniform scalar
// load.  Per element that is still the two fp64 adds of the reference's left-to-right sum (the running sum is carried
// across pieces, so the order of additions is unchanged) plus 1/32 of a vector load; stores are 8 B per lane, contiguous
// across the wave.  Measured at 50k x 50k fp64 (profiles/archive/r02_dense_generic_kernel.log): T=49 17.7 ms, T=64 22.8 ms,
// T=128 44.0 ms, i.e. ~14.5 T fp64 lane-instructions/s -- half the column-resident kernel's rate (one 64-byte scalar
// load per 16 VALU instructions), against 497 / 616 / 1317 ms for the one-column-per-lane, load-per-element form it
// replaces.  It also beats the column-resident kernel at fp64 T=48 (17.6 vs 20.6 ms, where that kernel is down to one
// column per lane), not below (T=44: 16.2 vs 11.7 ms; fp32 T=48: 9.1 vs 5.8 ms).
template <typename F, int RB>
__global__ __launch_bounds__(256) void dense_cost_rowblock_kernel(
    const F *__restrict__ A, const F *__restrict__ R, const F *__restrict__ axy,
    const F *__restrict__ rxy, int T, int64_t n_r, int64_t row_begin, int64_t row_end, F w, F dcoef,
    F *__restrict__ out, int64_t ld, int col_tiles) {
    constexpr int TC = 8;
    const int tile = blockIdx.x % col_tiles;
    const int64_t chunk = blockIdx.x / col_tiles;
    const int64_t j = (int64_t)tile * 256 + threadIdx.x;
    const int64_t jc = j < n_r ? j : n_r - 1;          // lanes past the edge compute a valid column and do not store
    const int64_t i0 = row_begin + chunk * RB;
    const F *__restrict__ rp = R + jc * T;
    F acc[RB];
#pragma unroll
    for (int q = 0; q < RB; ++q) acc[q] = F(0);
    int t0 = 0;
    for (; t0 + TC <= T; t0 += TC) {
        F r[TC];
#pragma unroll
        for (int e = 0; e < TC; ++e) r[e] = rp[t0 + e];
#pragma unroll
        for (int q = 0; q < RB; ++q) {
            int64_t i = i0 + q;
            if (i >= row_end) i = row_end - 1;           // rows past the end recompute the last row (never stored)
            const F *__restrict__ a = A + i * T + t0;    // wave-uniform: scalar loads
#pragma unroll
            for (int e = 0; e < TC; ++e) acc[q] = acc[q] + absf<F>(a[e] - r[e]);
        }
    }
    for (; t0 < T; ++t0) {                                // T % 8 trailing types
        const F r = rp[t0];
#pragma unroll
        for (int q = 0; q < RB; ++q) {
            int64_t i = i0 + q;
            if (i >= row_end) i = row_end - 1;
            acc[q] = acc[q] + absf<F>(A[i * T + t0] - r);
        }
    }
    const F rx = rxy[2 * jc], ry = rxy[2 * jc + 1];
    if (j >= n_r) return;
#pragma unroll
    for (int q = 0; q < RB; ++q) {
        const int64_t i = i0 + q;
        if (i < row_end) {
            const F dc = absf<F>(axy[2 * i] - rx) + absf<F>(axy[2 * i + 1] - ry);
            out[(i - row_begin) * ld + j] = w * acc[q] + dcoef * dc;
        }
    }
}

// ---- opt-in fixed-point build ------------------------------------------------------------------------------------
// At T=20 the fp64 build is bounded by the energy of its 40 fp64 adds per output under the board power cap (DESIGN.md
// 5.1), not by HBM.  The type sum sum_t |a_t - r_t| is a sum of absolute differences, and for that CDNA has an integer
// instruction: v_sad_u32 d = |s0 - s1| + s2 -- one full-rate 32-bit op per element where the fp64 form needs two
// 64-bit ones.  With the type values on a common fixed-point grid q(v) = rint((v - offset) * 2^s) the sum is EXACT in
// integers (no rounding in the accumulation at all); the only error is the grid itself: |S_q 2^-s - S| <= T 2^-s, which
// the caller sizes from the data range (ops.quantize_types picks the largest s whose sums fit 32 bits: s = 24 for
// probability rows on the reference's 0-100 scale, i.e. <= 1.2e-6 ABSOLUTE at T = 20 on costs of order 100, tighter than
// the fp32 variant of config 5 by four orders of magnitude).  Output stays fp64:
//   c = w * (double(S_q) * 2^-s) + (w * 0.001) * (|ax - rx| + |ay - ry|)        (XY part in fp64 as before)
// A sum too small for the grid to carry a RELATIVE tolerance (fewer than T / rel_tol grid steps: near-identical cells) is
// recomputed on the spot with the reference's fp64 expression, so every output is within rel_tol of the reference's
// value (rel_tol = 1e-6 is BASELINE.json's own tolerance for fp64 costs) -- by construction, not by sampling.
// It is NOT the reference's arithmetic, so it is never the default: same_dense_cost_q32_dev is a separate entry point,
// with its own twin in the test oracle (bit-equal) and a stated bound against the fp64 costs.  Its natural
// customer is the dense matrix of the Hungarian MIP-start heuristic (src/init_helpers.py:151-155), where a 1e-6
// perturbation of a start value is immaterial.
__global__ __launch_bounds__(256) void quantize_u32_kernel(const double *__restrict__ src, int64_t n, double offset, double scale,
                                                            uint32_t *__restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double q = __builtin_rint((src[i] - offset) * scale);      // round half to even, as the oracle's rint()
    q = q < 0.0 ? 0.0 : (q > 4294967295.0 ? 4294967295.0 : q);
    dst[i] = (uint32_t)q;
}

// the rare exact path of the fixed-point build (a type sum too small for the grid): kept out of line so that the row loop
// stays small -- it runs for about one output per aligned row at most
__device__ __attribute__((noinline)) double exact_type_sum(const double *__restrict__ ap, const double *__restrict__ rp, int T) {
    double s = 0.0;
    for (int t = 0; t < T; ++t) s = s + __builtin_fabs(ap[t] - rp[t]);
    return s;
}

__device__ __forceinline__ uint32_t sad_u32(uint32_t a, uint32_t r, uint32_t acc) {
    return (a > r ? a - r : r - a) + acc;   // selected as v_sad_u32 (checked in the ISA)
}

template <int T>
__global__ __launch_bounds__(256) void dense_cost_q32_kernel(
    const uint32_t *__restrict__ Aq, const uint32_t *__restrict__ Rq, const double *__restrict__ A, const double *__restrict__ R,
    const double *__restrict__ axy, const double *__restrict__ rxy, int64_t n_r, int64_t row_begin, int64_t row_end, double w,
    double wq, double dcoef, uint32_t guard, double *__restrict__ out, int64_t ld, int col_tiles, int rows_per_block, int64_t n_store,
    int row_chunks) {
    constexpr int CPL = 2;   // 16 B per lane per row: a wave writes 1 KiB of one output row (4 columns per lane, two stores 32 B
                             // apart, measured 42 ms against 11.7 ms: half-line writes)
    constexpr int TT = T > 0 ? T : 1;
    typedef double d2 __attribute__((ext_vector_type(2)));
    // store stream: blocks that share an XCD (blockIdx % 8) take adjacent column tiles of one row chunk (map 2 of the fp64 kernel)
    const unsigned b = blockIdx.x, xcd = b & 7u, kk = b >> 3;
    const unsigned per = (gridDim.x + 7u) >> 3;
    const unsigned lin = xcd * per + kk;
    const int tile = lin % col_tiles, chunk = lin / col_tiles;
    if (chunk >= row_chunks) return;
    const int64_t j0 = (((int64_t)tile * 4 + (threadIdx.x >> 6)) * 64 + (threadIdx.x & 63)) * CPL;
    int64_t i0 = row_begin + (int64_t)chunk * rows_per_block;
    if (i0 + rows_per_block > row_end) i0 = row_end - rows_per_block;   // last chunk overlaps its neighbour (identical values)
    uint32_t r[CPL][TT];
    double rx[CPL], ry[CPL];
    int64_t jj[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        int64_t j = j0 + c;
        if (j >= n_r) j = n_r - 1;
        jj[c] = j;
        const uint32_t *rp = Rq + j * T;
#pragma unroll
        for (int t = 0; t < T; ++t) r[c][t] = rp[t];
        rx[c] = rxy[2 * j];
        ry[c] = rxy[2 * j + 1];
    }
    if (j0 >= n_store) return;
    const uint32_t *__restrict__ arow = Aq + i0 * T;       // wave-uniform -> scalar loads
    const double *__restrict__ axyrow = axy + 2 * i0;
    char *orow = reinterpret_cast<char *>(out + (i0 - row_begin) * ld);
    const unsigned lane_off = (unsigned)(j0 * sizeof(double));
    const int64_t row_pitch = ld * (int64_t)sizeof(double);
    // one output row: T integer SADs per column, the fp64 remainder, the guard, one 16-byte store
    auto row = [&](const uint32_t (&a)[TT], double ax, double ay, int64_t i, char *dst) {
        uint32_t acc[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) acc[c] = 0u;
#pragma unroll
        for (int t = 0; t < T; ++t) {
#pragma unroll
            for (int c = 0; c < CPL; ++c) acc[c] = sad_u32(a[t], r[c][t], acc[c]);
        }
        double v[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const double dc = __builtin_fabs(ax - rx[c]) + __builtin_fabs(ay - ry[c]);
            v[c] = (double)acc[c] * wq + dcoef * dc;
            if (acc[c] < guard) {
                // a type sum this small cannot carry the relative tolerance on a grid (T grid steps of error against fewer than
                // T / rel_tol steps of value): the reference's own fp64 expression instead -- near-identical cells, about one
                // column per row when the sections are jittered copies, none for unrelated ones
                v[c] = w * exact_type_sum(A + i * (int64_t)T, R + jj[c] * (int64_t)T, T) + dcoef * dc;
            }
        }
        store16_nt_saddr(dst, lane_off, d2{v[0], v[1]});
    };
    // Rows go in pairs through two scalar register sets: while one row is computed from set A the next row's scalars are
    // on their way into set B, and the other way round -- no copies between the sets (the one-set form spent 3.5e9 scalar
    // moves per launch on them).
    uint32_t a[TT], b2[TT];
#pragma unroll
    for (int t = 0; t < T; ++t) a[t] = arow[t];
    double ax = axyrow[0], ay = axyrow[1];
    int q = 0;
    for (; q + 1 < rows_per_block; q += 2) {
#pragma unroll
        for (int t = 0; t < T; ++t) b2[t] = arow[T + t];
        const double bx = axyrow[2], by = axyrow[3];
        row(a, ax, ay, i0 + q, orow);
        const bool more = q + 2 < rows_per_block;
        const uint32_t *__restrict__ an_p = more ? arow + 2 * T : arow;
        const double *__restrict__ axn_p = more ? axyrow + 4 : axyrow;
#pragma unroll
        for (int t = 0; t < T; ++t) a[t] = an_p[t];
        ax = axn_p[0];
        ay = axn_p[1];
        row(b2, bx, by, i0 + q + 1, orow + row_pitch);
        orow += 2 * row_pitch;
        arow += 2 * T;
        axyrow += 4;
    }
    if (q < rows_per_block) row(a, ax, ay, i0 + q, orow);
}

template <typename F> __device__ __forceinline__ F inf_of();
template <> __device__ __forceinline__ double inf_of<double>() { return __builtin_inf(); }
template <> __device__ __forceinline__ float inf_of<float>() { return __builtin_inff(); }

// One lane per pair (a4).  Consecutive pairs share the aligned row (L1 broadcast); ref rows
// are gathered.  Gather-bound, P*(2*(T+2)+1)*sizeof(F) B of touched data.  F = float is the config-5 variant:
// the same expression evaluated in float (element (i, j) of the fp32 dense build).
template <typename F>
__global__ __launch_bounds__(256) void pair_cost_kernel(
    const F *__restrict__ A, const F *__restrict__ R, int T, const F *__restrict__ axy,
    const F *__restrict__ rxy, const int32_t *__restrict__ pairs, int64_t P, F w, F dcoef,
    F *__restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const int64_t i = pairs[2 * p], j = pairs[2 * p + 1];
    const F *a = A + i * T, *r = R + j * T;
    F s = F(0);
    for (int t = 0; t < T; ++t) s = s + absf<F>(a[t] - r[t]);
    const F dc = absf<F>(axy[2 * i] - rxy[2 * j]) + absf<F>(axy[2 * i + 1] - rxy[2 * j + 1]);
    out[p] = w * s + dcoef * dc;
}

// Window form (csrc/window.hip): list row q / k is row arows[q / k] of A / axy (the moving SECTION's arrays), the number of list
// rows lives on the device and the launch is sized by an upper bound.
struct WinCost {
    const int32_t *arows;
    const unsigned long long *n_a;
};
// the windows of one launch (blockIdx.y = window; by value in the kernarg segment): their row lists, candidate lists and outputs
struct WinCostBatch {
    WinCost w[SAME_LAUNCH_WINDOWS];
    const int32_t *idx[SAME_LAUNCH_WINDOWS];
    void *out[SAME_LAUNCH_WINDOWS];
};

// Costs of padded candidate lists idx[(i-row_begin)*k + q] (-1 = empty -> +inf).
template <typename F, bool WIN = false>
__global__ __launch_bounds__(256) void padded_cost_kernel(
    const F *__restrict__ A, const F *__restrict__ R, int T, const F *__restrict__ axy,
    const F *__restrict__ rxy, int64_t row_begin, int64_t n_slots, int k, const int32_t *__restrict__ idx,
    F w, F dcoef, F *__restrict__ out, WinCostBatch wb) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const WinCost &win = wb.w[WIN ? blockIdx.y : 0];
    if constexpr (WIN) {
        n_slots = (int64_t)*win.n_a * k;
        idx = wb.idx[blockIdx.y];
        out = static_cast<F *>(wb.out[blockIdx.y]);
    }
    if (q >= n_slots) return;
    const int64_t j = idx[q];
    if (j < 0) { out[q] = inf_of<F>(); return; }
    const int64_t i = WIN ? (int64_t)win.arows[q / k] : row_begin + q / k;
    const F *a = A + i * T, *r = R + j * T;
    F s = F(0);
    for (int t = 0; t < T; ++t) s = s + absf<F>(a[t] - r[t]);
    const F dc = absf<F>(axy[2 * i] - rxy[2 * j]) + absf<F>(axy[2 * i + 1] - rxy[2 * j + 1]);
    out[q] = w * s + dcoef * dc;
}

// Same result, staged: a wave owns 64 consecutive slots.  One lane per slot gathering its own reference row makes every
// load instruction touch 64 different cache lines; here the wave first copies its 64 rows into LDS with lanes running
// along the rows (each load instruction covers ~3 rows = a handful of lines), then every lane walks its row in LDS in the
// reference's left-to-right order.  Row pitch in LDS is T|1 elements (odd: 2-way bank aliasing at worst).
template <typename F, bool WIN = false>
__global__ void padded_cost_lds_kernel(
    const F *__restrict__ A, const F *__restrict__ R, int T, const F *__restrict__ axy,
    const F *__restrict__ rxy, int64_t row_begin, int64_t n_slots, int k, const int32_t *__restrict__ idx,
    F w, F dcoef, F *__restrict__ out, WinCostBatch wb) {
    extern __shared__ double lds_raw[];
    const WinCost &win = wb.w[WIN ? blockIdx.y : 0];
    if constexpr (WIN) {
        n_slots = (int64_t)*win.n_a * k;
        idx = wb.idx[blockIdx.y];
        out = static_cast<F *>(wb.out[blockIdx.y]);
    }
    F *lds = reinterpret_cast<F *>(lds_raw);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const int P = T | 1;
    F *rows = lds + (size_t)wave * 64 * P;
    int *jl = reinterpret_cast<int *>(lds + (size_t)waves * 64 * P) + wave * 64;
    const int64_t q0 = ((int64_t)blockIdx.x * waves + wave) * 64;
    if (q0 >= n_slots) return;                       // whole wave out of range (wave-uniform)
    const int64_t q = q0 + lane;
    const int j = q < n_slots ? idx[q] : -1;
    jl[lane] = j;
    __builtin_amdgcn_wave_barrier();                 // jl is written and read by this wave only
    {   // element e = slot * T + t of the wave's 64 x T block, e = lane, lane + 64, ...: (slot, t) advanced without dividing
        const int ds = 64 / T, dt = 64 - ds * T;
        int sl = lane / T, t = lane - sl * T;
        for (int it = 0; it < T; ++it) {
            const int js = jl[sl];
            rows[sl * P + t] = js >= 0 ? R[(int64_t)js * T + t] : F(0);
            sl += ds;
            t += dt;
            if (t >= T) { t -= T; ++sl; }
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (q >= n_slots) return;
    if (j < 0) { out[q] = inf_of<F>(); return; }
    const int64_t i = WIN ? (int64_t)win.arows[q / k] : row_begin + q / k;
    const F *a = A + i * T;
    const F *r = rows + lane * P;
    F s = F(0);
    for (int t = 0; t < T; ++t) s = s + absf<F>(a[t] - r[t]);
    const F dc = absf<F>(axy[2 * i] - rxy[2 * (int64_t)j]) + absf<F>(axy[2 * i + 1] - rxy[2 * (int64_t)j + 1]);
    out[q] = w * s + dcoef * dc;
}

template <typename F, int T, int CPL, bool VEC>
int launch_one(same_ctx *ctx, const F *A, const F *R, const F *axy, const F *rxy, int64_t n_r, int64_t rb, int64_t re,
               F w, F *out_rb, int64_t ld, int64_t n_cols, int rows_per_block, int map_mode) {
    // out_rb points at the output row of `rb`
    const int64_t rows = re - rb;
    if (rows <= 0) return SAME_OK;
    const int col_tiles = (int)ceil_div(n_cols, 256 * CPL);
    const int64_t chunks = ceil_div(rows, rows_per_block);
    int64_t blocks = chunks * col_tiles;
    if (map_mode == MAP_XCD_ROWS) blocks = ceil_div(blocks, 8) * 8;
    if (map_mode == MAP_XCD_TILES) blocks = ceil_div(col_tiles, 8) * 8 * chunks;
    REQUIRE(ctx, blocks < (int64_t)1 << 31);
    if (w == F(1))
        hipLaunchKernelGGL((dense_cost_kernel<F, T, CPL, VEC, true>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, A, R, axy, rxy, n_r,
                           rb, re, w, w * F(0.001), out_rb, ld, col_tiles, rows_per_block, n_cols, map_mode, (int)chunks);
    else
        hipLaunchKernelGGL((dense_cost_kernel<F, T, CPL, VEC, false>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, A, R, axy, rxy, n_r,
                           rb, re, w, w * F(0.001), out_rb, ld, col_tiles, rows_per_block, n_cols, map_mode, (int)chunks);
    HIP_TRY(ctx, hipGetLastError());
    return SAME_OK;
}

// SAME_DENSE_MAP = 0 | 2 | 4 forces one of the three shipped block maps (a test hook: every map must give the same bits)
static int forced_map() {
    static const int m = [] {
        const char *v = getenv("SAME_DENSE_MAP");
        const int x = v ? atoi(v) : -1;
        return (x == MAP_TILE_FASTEST || x == MAP_XCD_ROWS || x == MAP_XCD_TILES) ? x : -1;
    }();
    return m;
}

template <typename F, int T>
int launch_dense_T(same_ctx *ctx, const F *A, const F *R, const F *axy, const F *rxy, int64_t n_r, int64_t rb,
                   int64_t re, F w, F *out, int64_t ld, int64_t n_store) {
    constexpr int CPLV = vec_of<F>::cpl;
    // T large: one column per lane keeps the register file within budget (and gives up the 16-byte store)
    constexpr int CPL = (T * CPLV * (int)(sizeof(F) / 4) <= 160) ? CPLV : 1;
    const int64_t rows = re - rb;
    // Store-bound shapes want every XCD to stream whole output rows (MAP_XCD_ROWS: 11.4 ms at 100k x 100k vs 14.1 ms for
    // MAP_XCD_TILES); once fp64 issue under the power cap is the limit (T >= 16, equal times) MAP_XCD_TILES keeps each XCD's
    // share of R in its own L2 (profiles/archive/r01_dense_map_fetch.md).
    const int map = forced_map() >= 0 ? forced_map() : ((sizeof(F) == 8 && T >= 16) ? MAP_XCD_TILES : MAP_XCD_ROWS);
    // vector stores need whole CPL-groups: n_store (a multiple of CPL, n_r <= n_store <= ld) says how many
    // columns may be written; without such padding the scalar-store variant is used
    const bool vec_ok = CPL > 1 && (ld % CPL == 0) && (n_store % CPL == 0) && n_store >= n_r && n_store <= ld &&
                        (reinterpret_cast<uintptr_t>(out) % (CPL * sizeof(F)) == 0) && ld * (int64_t)sizeof(F) < ((int64_t)1 << 32);
    if constexpr (CPL > 1) {
        if (vec_ok) {
            const int col_tiles = (int)ceil_div(n_store, 256 * CPL);
            // enough row chunks to fill the chip several times over, long enough to amortise the column prologue: 256 rows per
            // block measured best at 100k x 100k (store-bound T<=12: 6.4-6.9 TB/s vs 5.9 at 64 or 512; T=20: equal) --
            // profiles/archive/r01_dense_probe_rpb.log
            int rows_per_block = 256;
            while (rows_per_block > 32 && ceil_div(rows, rows_per_block) * col_tiles < 4096) rows_per_block /= 2;
            if (rows_per_block > rows) rows_per_block = (int)rows;   // fewer rows than one chunk: one chunk of exactly these rows
            return launch_one<F, T, CPL, true>(ctx, A, R, axy, rxy, n_r, rb, re, w, out, ld, n_store, rows_per_block, map);
        }
    }
    const int rpb = (int)std::min<int64_t>(rows, 64);
    return launch_one<F, T, CPL, false>(ctx, A, R, axy, rxy, n_r, rb, re, w, out, ld, n_r, rpb,
                                        forced_map() >= 0 ? forced_map() : MAP_TILE_FASTEST);
}

template <typename F>
int launch_dense(same_ctx *ctx, const F *A, const F *R, int T, const F *axy, const F *rxy, int64_t n_r, int64_t rb,
                 int64_t re, F w, F *out, int64_t ld, int64_t n_store) {
    REQUIRE(ctx, ctx && A && R && axy && rxy && out);
    REQUIRE(ctx, T >= 0 && T <= SAME_MAX_TYPES && n_r >= 0 && rb >= 0 && re >= rb && ld >= n_r);
    SAME_TRY(same_use(ctx));
    if (n_r == 0 || re == rb) return SAME_OK;
    // above this many type columns the row-blocked kernel takes over from the column-resident one (measured crossover:
    // profiles/archive/r02_dense_generic_kernel.log)
    constexpr int rowblock_min_T = sizeof(F) == 8 ? 48 : 49;
    if (T < rowblock_min_T)
    switch (T) {
#define CASE_T(n) case n: return launch_dense_T<F, n>(ctx, A, R, axy, rxy, n_r, rb, re, w, out, ld, n_store);
        CASE_T(0) CASE_T(1) CASE_T(2) CASE_T(3) CASE_T(4) CASE_T(5) CASE_T(6) CASE_T(7) CASE_T(8)
        CASE_T(9) CASE_T(10) CASE_T(11) CASE_T(12) CASE_T(13) CASE_T(14) CASE_T(15) CASE_T(16)
        CASE_T(17) CASE_T(18) CASE_T(19) CASE_T(20) CASE_T(21) CASE_T(22) CASE_T(23) CASE_T(24)
        CASE_T(25) CASE_T(26) CASE_T(27) CASE_T(28) CASE_T(29) CASE_T(30) CASE_T(31) CASE_T(32)
        CASE_T(33) CASE_T(34) CASE_T(35) CASE_T(36) CASE_T(37) CASE_T(38) CASE_T(39) CASE_T(40)
        CASE_T(41) CASE_T(42) CASE_T(43) CASE_T(44) CASE_T(45) CASE_T(46) CASE_T(47) CASE_T(48)
#undef CASE_T
        default: break;
    }
    constexpr int RB = 32;
    const int64_t rows = re - rb;
    const int col_tiles = (int)ceil_div(n_r, 256);
    const int64_t blocks = ceil_div(rows, RB) * col_tiles;
    REQUIRE(ctx, blocks < (int64_t)1 << 31);
    hipLaunchKernelGGL((dense_cost_rowblock_kernel<F, RB>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, A, R, axy, rxy,
                       T, n_r, rb, re, w, w * F(0.001), out, ld, col_tiles);
    HIP_TRY(ctx, hipGetLastError());
    return SAME_OK;
}

template <typename F>
int dense_host(same_ctx *ctx, const F *A, const F *R, int64_t n_m, int64_t n_r, int T, const F *axy, const F *rxy,
               int64_t rb, int64_t re, F w, F *out, int64_t ld) {
    REQUIRE(ctx, ctx && out && (n_m == 0 || axy) && (n_r == 0 || rxy));
    REQUIRE(ctx, n_m >= 0 && n_r >= 0 && T >= 0 && rb >= 0 && re >= rb && re <= n_m && ld >= n_r);
    REQUIRE(ctx, T == 0 || ((n_m == 0 || A) && (n_r == 0 || R)));
    SAME_TRY(same_use(ctx));
    if (re == rb || n_r == 0) return SAME_OK;
    F *dA, *dR, *dax, *drx, *dout;
    SAME_TRY(up_as(ctx, SL_A, A, (size_t)n_m * T, &dA));
    SAME_TRY(up_as(ctx, SL_R, R, (size_t)n_r * T, &dR));
    SAME_TRY(up_as(ctx, SL_AXY, axy, (size_t)n_m * 2, &dax));
    SAME_TRY(up_as(ctx, SL_RXY, rxy, (size_t)n_r * 2, &drx));
    const int64_t rows = re - rb;
    const int64_t dld = (n_r + 3) & ~int64_t(3);  // device tile is padded so the 16 B store path is taken
    SAME_TRY(slot_as(ctx, SL_OUT0, (size_t)rows * dld, &dout));
    SAME_TRY(launch_dense<F>(ctx, dA, dR, T, dax, drx, n_r, rb, re, w, dout, dld, dld));
    HIP_TRY(ctx, hipMemcpy2DAsync(out, (size_t)ld * sizeof(F), dout, (size_t)dld * sizeof(F), (size_t)n_r * sizeof(F),
                                  (size_t)rows, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SAME_OK;
}

template <typename F>
int pair_cost_host(same_ctx *ctx, const F *A, const F *R, int64_t n_m, int64_t n_r, int T, const F *axy, const F *rxy,
                   const int32_t *pairs, int64_t P, F w, F *out_c) {
    REQUIRE(ctx, ctx != nullptr);
    REQUIRE(ctx, n_m >= 0 && n_r >= 0 && T >= 0 && T <= SAME_MAX_TYPES && P >= 0);
    if (P == 0) return SAME_OK;
    REQUIRE(ctx, axy && rxy && pairs && out_c && (T == 0 || (A && R)));
    SAME_TRY(same_use(ctx));
    // validate on the host: a bad index must never become a device fault
    for (int64_t p = 0; p < P; ++p) {
        if (pairs[2 * p] < 0 || pairs[2 * p] >= n_m || pairs[2 * p + 1] < 0 || pairs[2 * p + 1] >= n_r) {
            ctx->err = "pair index out of range";
            return SAME_ERANGE;
        }
    }
    F *dA, *dR, *dax, *drx, *dout;
    int32_t *dp;
    SAME_TRY(up_as(ctx, SL_A, A, (size_t)n_m * T, &dA));
    SAME_TRY(up_as(ctx, SL_R, R, (size_t)n_r * T, &dR));
    SAME_TRY(up_as(ctx, SL_AXY, axy, (size_t)n_m * 2, &dax));
    SAME_TRY(up_as(ctx, SL_RXY, rxy, (size_t)n_r * 2, &drx));
    SAME_TRY(up_as(ctx, SL_PAIRS, pairs, (size_t)P * 2, &dp));
    SAME_TRY(slot_as(ctx, SL_OUT0, (size_t)P, &dout));
    hipLaunchKernelGGL(pair_cost_kernel<F>, dim3((unsigned)ceil_div(P, 256)), dim3(256), 0, ctx->stream, dA, dR, T, dax, drx,
                       dp, P, w, w * F(0.001), dout);
    HIP_TRY(ctx, hipGetLastError());
    SAME_TRY(same_down(ctx, out_c, dout, (size_t)P * sizeof(F)));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SAME_OK;
}

// n_win > 0: the window form -- `wb` carries n_win windows' lists and outputs (didx / dout_cost unused), row_end - row_begin is the
// largest window's bound on its rows
template <typename F, bool WIN = false>
int padded_cost_dev(same_ctx *ctx, const F *dA, const F *dR, int T, const F *daxy, const F *drxy, int64_t row_begin,
                    int64_t row_end, int k, const int32_t *didx, F w, F *dout_cost, const WinCostBatch &wb = WinCostBatch{}, int n_win = 1) {
    REQUIRE(ctx, ctx && daxy && drxy && (WIN || (didx && dout_cost)) && (T == 0 || (dA && dR)));
    REQUIRE(ctx, T >= 0 && T <= SAME_MAX_TYPES && k >= 1 && row_begin >= 0 && row_end >= row_begin);
    SAME_TRY(same_use(ctx));
    const int64_t n_slots = (row_end - row_begin) * k;   // WIN: the upper bound the launch is sized by
    if (n_slots == 0) return SAME_OK;
    size_t per_wave = (size_t)64 * (T | 1) * sizeof(F) + 64 * sizeof(int);
    per_wave = (per_wave + 7) & ~size_t(7);
    int waves = (int)std::min<size_t>(4, (size_t)65536 / per_wave);
    if (T >= 2 && waves >= 1 && n_slots >= 64 * 64) {   // LDS-staged rows; tiny inputs and very wide rows: one lane gathers its row
        // the per-wave index list sits after ALL waves' row blocks: keep it 4-byte aligned for float rows of odd pitch
        SAME_LAUNCH(ctx, (padded_cost_lds_kernel<F, WIN>), dim3((unsigned)ceil_div(n_slots, 64 * waves), (unsigned)n_win), dim3(64 * waves),
                    waves * per_wave, dA, dR, T, daxy, drxy, row_begin, n_slots, k, didx, w, w * F(0.001), dout_cost, wb);
    } else {
        SAME_LAUNCH(ctx, (padded_cost_kernel<F, WIN>), dim3((unsigned)ceil_div(n_slots, 256), (unsigned)n_win), dim3(256), 0, dA, dR, T,
                    daxy, drxy, row_begin, n_slots, k, didx, w, w * F(0.001), dout_cost, wb);
    }
    HIP_TRY(ctx, hipGetLastError());
    return SAME_OK;
}


template <int T>
int launch_q32_T(same_ctx *ctx, const uint32_t *Aq, const uint32_t *Rq, const double *A, const double *R, const double *axy,
                 const double *rxy, int64_t n_r, int64_t rb, int64_t re, double w, double wq, double dcoef, uint32_t guard, double *out,
                 int64_t ld) {
    const int64_t rows = re - rb;
    int rows_per_block = 256;
    const int col_tiles = (int)ceil_div(ld, 256 * 2);
    while (rows_per_block > 32 && ceil_div(rows, rows_per_block) * col_tiles < 4096) rows_per_block /= 2;
    if (rows_per_block > rows) rows_per_block = (int)rows;
    const int64_t chunks = ceil_div(rows, rows_per_block);
    const int64_t blocks = ceil_div(chunks * col_tiles, 8) * 8;
    REQUIRE(ctx, blocks < (int64_t)1 << 31);
    hipLaunchKernelGGL((dense_cost_q32_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, Aq, Rq, A, R, axy, rxy, n_r, rb, re, w,
                       wq, dcoef, guard, out, ld, col_tiles, rows_per_block, ld, (int)chunks);
    HIP_TRY(ctx, hipGetLastError());
    return SAME_OK;
}

}  // namespace

// The window path's candidate-list costs (csrc/window.hip; declared in common.h) for the windows of a batch in ONE launch: a job's
// rows are rows[0, *dn) of the moving section, its candidates reference SECTION rows idx[cap][k]; cap = the upper bound on *dn.
// Enqueue only.
int same_padded_cost_window_batch_core(same_ctx *ctx, int cost_f32, const void *dA, const void *dR, int T, const void *daxy_c, const void *drxy_c,
                                       const same_cost_window_job *jobs, int n_jobs, int k, double w) {
    REQUIRE(ctx, n_jobs >= 1 && n_jobs <= SAME_LAUNCH_WINDOWS);
    WinCostBatch wb{};
    int64_t cap = 0;
    for (int q = 0; q < n_jobs; ++q) {
        REQUIRE(ctx, jobs[q].cap >= 0 && (jobs[q].cap == 0 || (jobs[q].idx && jobs[q].out)));
        wb.w[q] = WinCost{jobs[q].rows, jobs[q].dn};
        wb.idx[q] = jobs[q].idx;
        wb.out[q] = jobs[q].out;
        cap = std::max(cap, jobs[q].cap);
    }
    if (cost_f32)
        return padded_cost_dev<float, true>(ctx, static_cast<const float *>(dA), static_cast<const float *>(dR), T, static_cast<const float *>(daxy_c),
                                            static_cast<const float *>(drxy_c), 0, cap, k, nullptr, (float)w, nullptr, wb, n_jobs);
    return padded_cost_dev<double, true>(ctx, static_cast<const double *>(dA), static_cast<const double *>(dR), T, static_cast<const double *>(daxy_c),
                                         static_cast<const double *>(drxy_c), 0, cap, k, nullptr, w, nullptr, wb, n_jobs);
}

extern "C" {

int same_dense_cost_f64_dev(same_ctx *ctx, const double *dA, const double *dR, int T, const double *daxy,
                            const double *drxy, int64_t n_r, int64_t row_begin, int64_t row_end, double w,
                            double *dout, int64_t ld) {
    return launch_dense<double>(ctx, dA, dR, T, daxy, drxy, n_r, row_begin, row_end, w, dout, ld, n_r);
}

int same_dense_cost_f32_dev(same_ctx *ctx, const float *dA, const float *dR, int T, const float *daxy,
                            const float *drxy, int64_t n_r, int64_t row_begin, int64_t row_end, float w,
                            float *dout, int64_t ld) {
    return launch_dense<float>(ctx, dA, dR, T, daxy, drxy, n_r, row_begin, row_end, w, dout, ld, n_r);
}

int same_dense_cost_f64(same_ctx *ctx, const double *A, const double *R, int64_t n_m, int64_t n_r, int T,
                        const double *axy, const double *rxy, int64_t row_begin, int64_t row_end, double w,
                        double *out, int64_t ld) {
    return dense_host<double>(ctx, A, R, n_m, n_r, T, axy, rxy, row_begin, row_end, w, out, ld);
}

int same_dense_cost_f32(same_ctx *ctx, const float *A, const float *R, int64_t n_m, int64_t n_r, int T,
                        const float *axy, const float *rxy, int64_t row_begin, int64_t row_end, float w, float *out,
                        int64_t ld) {
    return dense_host<float>(ctx, A, R, n_m, n_r, T, axy, rxy, row_begin, row_end, w, out, ld);
}

int same_pair_cost_f64(same_ctx *ctx, const double *A, const double *R, int64_t n_m, int64_t n_r, int T,
                       const double *axy, const double *rxy, const int32_t *pairs, int64_t P, double w,
                       double *out_c) {
    return pair_cost_host<double>(ctx, A, R, n_m, n_r, T, axy, rxy, pairs, P, w, out_c);
}

int same_pair_cost_f32(same_ctx *ctx, const float *A, const float *R, int64_t n_m, int64_t n_r, int T,
                       const float *axy, const float *rxy, const int32_t *pairs, int64_t P, float w,
                       float *out_c) {
    return pair_cost_host<float>(ctx, A, R, n_m, n_r, T, axy, rxy, pairs, P, w, out_c);
}

int same_padded_cost_f64_dev(same_ctx *ctx, const double *dA, const double *dR, int T, const double *daxy,
                             const double *drxy, int64_t row_begin, int64_t row_end, int k, const int32_t *didx,
                             double w, double *dout_cost) {
    return padded_cost_dev<double>(ctx, dA, dR, T, daxy, drxy, row_begin, row_end, k, didx, w, dout_cost);
}

int same_padded_cost_f32_dev(same_ctx *ctx, const float *dA, const float *dR, int T, const float *daxy,
                             const float *drxy, int64_t row_begin, int64_t row_end, int k, const int32_t *didx,
                             float w, float *dout_cost) {
    return padded_cost_dev<float>(ctx, dA, dR, T, daxy, drxy, row_begin, row_end, k, didx, w, dout_cost);
}

int same_quantize_u32_dev(same_ctx *ctx, const double *dsrc, int64_t n, double offset, double scale, uint32_t *ddst) {
    REQUIRE(ctx, ctx && n >= 0 && (n == 0 || (dsrc && ddst)) && scale > 0.0 && offset == offset);
    SAME_TRY(same_use(ctx));
    if (n == 0) return SAME_OK;
    hipLaunchKernelGGL(quantize_u32_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, ctx->stream, dsrc, n, offset, scale, ddst);
    HIP_TRY(ctx, hipGetLastError());
    return SAME_OK;
}

int same_dense_cost_q32_dev(same_ctx *ctx, const uint32_t *dAq, const uint32_t *dRq, const double *dA, const double *dR, int T,
                            const double *daxy, const double *drxy, int64_t n_r, int64_t row_begin, int64_t row_end, double w,
                            double inv_scale, double rel_tol, double *dout, int64_t ld) {
    REQUIRE(ctx, ctx && daxy && drxy && dout && (T == 0 || (dAq && dRq)));
    REQUIRE(ctx, T >= 0 && T <= SAME_Q32_MAX_TYPES && n_r >= 0 && row_begin >= 0 && row_end >= row_begin && inv_scale > 0.0);
    REQUIRE(ctx, rel_tol >= 0.0 && (rel_tol == 0.0 || T == 0 || (dA && dR)));
    // 16-byte stores of whole column pairs: the row pitch is the store width (columns [n_r, ld) are caller-owned padding)
    REQUIRE(ctx, ld >= n_r && ld % 2 == 0 && reinterpret_cast<uintptr_t>(dout) % 16 == 0 && ld * (int64_t)sizeof(double) < ((int64_t)1 << 32));
    SAME_TRY(same_use(ctx));
    if (n_r == 0 || row_end == row_begin) return SAME_OK;
    // sums below `guard` grid steps are recomputed in the reference's fp64 arithmetic: T steps of grid error are within
    // rel_tol of any sum of at least T / rel_tol steps (+ T for the sum's own displacement)
    uint32_t guard = 0;
    if (rel_tol > 0.0 && T > 0) {
        const double g = std::ceil((double)T / rel_tol) + (double)T;
        guard = g >= 4294967295.0 ? 4294967295u : (uint32_t)g;
    }
    const double wq = w * inv_scale, dcoef = w * 0.001;
    switch (T) {
#define CASE_Q(n) case n: return launch_q32_T<n>(ctx, dAq, dRq, dA, dR, daxy, drxy, n_r, row_begin, row_end, w, wq, dcoef, guard, dout, ld);
        CASE_Q(0) CASE_Q(1) CASE_Q(2) CASE_Q(3) CASE_Q(4) CASE_Q(5) CASE_Q(6) CASE_Q(7) CASE_Q(8)
        CASE_Q(9) CASE_Q(10) CASE_Q(11) CASE_Q(12) CASE_Q(13) CASE_Q(14) CASE_Q(15) CASE_Q(16)
        CASE_Q(17) CASE_Q(18) CASE_Q(19) CASE_Q(20) CASE_Q(21) CASE_Q(22) CASE_Q(23) CASE_Q(24)
        CASE_Q(25) CASE_Q(26) CASE_Q(27) CASE_Q(28) CASE_Q(29) CASE_Q(30) CASE_Q(31) CASE_Q(32)
#undef CASE_Q
        default: break;
    }
    return SAME_EINVAL;
}

}  // extern "C"
