// mfma_sub.hip -- can the idle FP64 MFMA pipe take part of the dense cost kernel's subtractions?  (round 5)
//
// The T = 20 fp64 dense build (same_amd/csrc/cost.hip, dense_cost_kernel<double,20,2>) is bound by fp64 VALU issue: per output
// 20 x (v_add_f64 d = a_t - r_t ; v_add_f64 acc += |d|).  For a 16 x 16 (aligned row, reference column) tile and one type t,
//     v_mfma_f64_16x16x4_f64  with  A row i = [a_it, 1, 0, 0],  B column j = [1, -r_jt, 0, 0]^T,  C = 0
// yields D_ij = fma(1, -r_jt, fma(a_it, 1, 0)) = RN(a_it - r_jt): the products are exact, so there is ONE rounding and the
// value is the VALU's -- provided the pipe rounds per k-step like an fma chain (checked here bit for bit).  `acc += |D|` stays a
// VALU op.  An MFMA costs 64 cycles of its pipe per 256 differences against 16 cycles of VALU for the same subtractions, so only
// a FRACTION of the output tiles can go that way before the MFMA pipe binds: with 40 % of the tiles the two pipes balance at
// 25.6 cycles per 256 elements against 32 (-20 %), if the clock holds under the board's power cap.
//
// The MFMA's result layout puts 4 rows x 1 column of a tile on a lane (col = lane & 15, row = (lane >> 4) + 4 reg), so a tile's
// running sums live in that layout too and ALL of its subtractions come from the MFMA; the aligned row is then no longer
// wave-uniform (no SGPR operands), which is why the two forms are mixed by TILE, not by type: `mfma` units (16 waves of 32
// columns x 256 rows each: two MFMAs per type give a lane the even and the odd column of a pair -> the same 16-byte stores) run
// beside `valu` units (the shipped kernel's loop) on the same SIMDs.
//
// Output per configuration: ms per launch (HIP events, mean of the timed launches), GB/s of algorithmic bytes, xor / sum
// checksums of the whole output (equal across configurations <=> bit-identical), and sampled rows against a host loop.
// tools/probes/mfma_sub.py runs it per configuration beside the sysfs telemetry (clock, watts).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form -o tools/probes/mfma_sub tools/probes/mfma_sub.hip
//        (without -amdgpu-mfma-vgpr-form the results land in AGPRs and every one is moved out by two v_accvgpr_read: +16 VALU per step)
// Run:   ./mfma_sub <mfma units per period> <period> [n=100000] [seconds=3] [LDS pad VALU kernel] [LDS pad MFMA kernel] [data 0|1]    (0 1 = the VALU form alone)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x)                                                                                      \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
    } while (0)

constexpr int T = 20;
constexpr int ROWS = 256;     // rows per unit
constexpr int UNIT_COLS = 512;

typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void touch(double v) { asm volatile("" ::"s"(v)); }

__device__ __forceinline__ void store16_nt_saddr(char *row_uniform, unsigned lane_byte_off, d2 v) {
    const i4 bits = __builtin_bit_cast(i4, v);
    asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" ::"v"(lane_byte_off), "v"(bits), "s"(row_uniform) : "memory");
}

__device__ __forceinline__ void store16_nt(double *p, d2 v) { __builtin_nontemporal_store(v, reinterpret_cast<d2 *>(p)); }

// ---- the shipped kernel's row loop (csrc/cost.hip), one block = 512 columns x ROWS rows --------------------------------------
__device__ __forceinline__ void valu_unit(const double *__restrict__ A, const double *__restrict__ R, const double *__restrict__ axy,
                                          const double *__restrict__ rxy, int64_t n_r, double dcoef, double *__restrict__ out, int64_t ld,
                                          int64_t i0, int64_t jbase) {
    constexpr int CPL = 2, H = T / 2;
    const int64_t j0 = jbase + (int64_t)threadIdx.x * CPL;
    double r[CPL][T], rx[CPL], ry[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        int64_t j = j0 + c;
        if (j >= n_r) j = n_r - 1;
#pragma unroll
        for (int t = 0; t < T; ++t) r[c][t] = R[j * T + t];
        rx[c] = rxy[2 * j];
        ry[c] = rxy[2 * j + 1];
    }
    if (j0 >= n_r) return;
    double h0[H];
    const double *__restrict__ arow = A + i0 * T;
    const double *__restrict__ axyrow = axy + 2 * i0;
#pragma unroll
    for (int t = 0; t < H; ++t) h0[t] = arow[t];
    char *orow = reinterpret_cast<char *>(out + i0 * ld);
    const unsigned lane_off = (unsigned)(j0 * sizeof(double));
    const int64_t row_pitch = ld * (int64_t)sizeof(double);
    for (int q = 0; q < ROWS; ++q) {
        touch(h0[0]);
        __builtin_amdgcn_sched_barrier(0);
        double h1[T - H];
#pragma unroll
        for (int t = H; t < T; ++t) h1[t - H] = arow[t];
        const double ax = axyrow[0], ay = axyrow[1];
        __builtin_amdgcn_sched_barrier(0);
        double s[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) s[c] = 0.0;
#pragma unroll
        for (int t = 0; t < H; ++t) {
            double dd[CPL];
#pragma unroll
            for (int c = 0; c < CPL; ++c) dd[c] = h0[t] - r[c][t];
#pragma unroll
            for (int c = 0; c < CPL; ++c) s[c] = s[c] + __builtin_fabs(dd[c]);
        }
        touch(ax);
        __builtin_amdgcn_sched_barrier(0);
        {
            const double *__restrict__ an = (q + 1 >= ROWS) ? arow : arow + T;
#pragma unroll
            for (int t = 0; t < H; ++t) h0[t] = an[t];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = H; t < T; ++t) {
            double dd[CPL];
#pragma unroll
            for (int c = 0; c < CPL; ++c) dd[c] = h1[t - H] - r[c][t];
#pragma unroll
            for (int c = 0; c < CPL; ++c) s[c] = s[c] + __builtin_fabs(dd[c]);
        }
        double v[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const double dc = __builtin_fabs(ax - rx[c]) + __builtin_fabs(ay - ry[c]);
            v[c] = s[c] + dcoef * dc;
        }
        store16_nt_saddr(orow, lane_off, d2{v[0], v[1]});
        orow += row_pitch;
        arow += T;
        axyrow += 2;
    }
}

// ---- the MFMA form: one WAVE = 32 columns x ROWS rows, in tiles of 16 rows -------------------------------------------------------
// lane l: c = l & 15, g = l >> 4.  Columns of the lane: jw + 2c (MFMA "E") and jw + 2c + 1 (MFMA "O"); rows of a tile: g + 4 reg.
// A operand of (tile, t): lanes g == 0 hold a[i0 + c][t] (k = 0), lanes g == 1 hold 1.0 (k = 1), the rest 0.
// B operand of (pair half, t): lanes g == 0 hold 1.0, lanes g == 1 hold -r[column][t], the rest 0.  XY ride along as types T, T+1.
// Both operands are read from LDS one step ahead (a register array indexed by an unrolled t made the compiler issue all 44 MFMAs of a
// tile first and spill their results): per wave  B[K][18] pairs (16 columns + a row of ones + a row of zeros), A[2][18][K] (two
// tiles: the next one is fetched while this one runs; rows 16 / 17 = ones / zeros).  The type loop is rolled in pairs of steps.
constexpr int KK = T + 2;
struct WaveLds {
    d2 b[KK][18];
    double a[2][18][KK];
};

__device__ __forceinline__ void mfma_unit(const double *__restrict__ AX /* [n_m][T+2]: types then X, Y */, const double *__restrict__ R,
                                          const double *__restrict__ rxy, int64_t n_r, double dcoef, double *__restrict__ out, int64_t ld,
                                          int64_t i0, int64_t jw, WaveLds &L) {
    constexpr int K = KK;
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    int64_t jE = jw + 2 * c, jO = jE + 1;
    const bool live = jE < n_r;          // n_r is even in this probe: a pair is in or out as a whole
    if (!live) { jE = n_r - 2; jO = n_r - 1; }
    // fill B: lanes 0..15 write their pair's column values, step by step; lanes 16 / 17 write the constant rows
    if (lane < 16) {
#pragma unroll
        for (int t = 0; t < K; ++t) {
            const double rE = t < T ? R[jE * T + t] : rxy[2 * jE + (t - T)];
            const double rO = t < T ? R[jO * T + t] : rxy[2 * jO + (t - T)];
            L.b[t][c] = d2{-rE, -rO};
        }
    } else if (lane < 18) {
        const double v = lane == 16 ? 1.0 : 0.0;
#pragma unroll
        for (int t = 0; t < K; ++t) L.b[t][lane] = d2{v, v};
        for (int t = 0; t < K; ++t) { L.a[0][lane][t] = v; L.a[1][lane][t] = v; }
    }
    // tile 0 of A: 16 rows x K doubles, contiguous in AX from row i0
    for (int e = lane; e < 16 * K; e += 64) L.a[0][e / K][e % K] = AX[i0 * K + e];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0);
    const int brow = g == 1 ? c : (g == 0 ? 16 : 17);     // which row of L.b this lane's B operand comes from
    const int arow = g == 0 ? c : (g == 1 ? 16 : 17);     // which row of L.a[.]
    const d4 zero = {0.0, 0.0, 0.0, 0.0};
    for (int tile = 0; tile < ROWS / 16; ++tile) {
        const int cur = tile & 1;
        // next tile's rows: fetched now, parked in registers, written to the other LDS buffer at the end of this tile
        double nxt[(16 * K + 63) / 64];
        const bool more = tile + 1 < ROWS / 16;
#pragma unroll
        for (int q = 0; q < (16 * K + 63) / 64; ++q) {
            const int e = lane + 64 * q;
            nxt[q] = (more && e < 16 * K) ? AX[(i0 + (int64_t)(tile + 1) * 16) * K + e] : 0.0;
        }
        const double *ar = &L.a[cur][arow][0];
        d4 sE = zero, sO = zero, cE = zero, cO = zero;
        // X and Y first (dc = |dx| + |dy|), then the types left to right
        {
            const double ax = ar[T], ay = ar[T + 1];
            const d2 bx = L.b[T][brow], by = L.b[T + 1][brow];
            const d4 xE = __builtin_amdgcn_mfma_f64_16x16x4f64(ax, bx.x, zero, 0, 0, 0);
            const d4 xO = __builtin_amdgcn_mfma_f64_16x16x4f64(ax, bx.y, zero, 0, 0, 0);
            const d4 yE = __builtin_amdgcn_mfma_f64_16x16x4f64(ay, by.x, zero, 0, 0, 0);
            const d4 yO = __builtin_amdgcn_mfma_f64_16x16x4f64(ay, by.y, zero, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                cE[e] = __builtin_fabs(xE[e]) + __builtin_fabs(yE[e]);
                cO[e] = __builtin_fabs(xO[e]) + __builtin_fabs(yO[e]);
            }
        }
        double a0 = ar[0];
        d2 b0 = L.b[0][brow];
        d4 dE = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0.x, zero, 0, 0, 0);
        d4 dO = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0.y, zero, 0, 0, 0);
        double a1 = ar[1];
        d2 b1 = L.b[1][brow];
#pragma unroll 1
        for (int t = 0; t < T; t += 2) {
            // step t + 1's MFMAs are issued, then the sums take step t's differences; then the same one step on
            const d4 eE = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1.x, zero, 0, 0, 0);
            const d4 eO = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1.y, zero, 0, 0, 0);
            const int t2 = t + 2 < T ? t + 2 : T - 1, t3 = t + 3 < T ? t + 3 : T - 1;     // past the end: a harmless repeat, never summed
            a0 = ar[t2];
            b0 = L.b[t2][brow];
#pragma unroll
            for (int e = 0; e < 4; ++e) sE[e] = sE[e] + __builtin_fabs(dE[e]);
#pragma unroll
            for (int e = 0; e < 4; ++e) sO[e] = sO[e] + __builtin_fabs(dO[e]);
            dE = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0.x, zero, 0, 0, 0);
            dO = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0.y, zero, 0, 0, 0);
            a1 = ar[t3];
            b1 = L.b[t3][brow];
#pragma unroll
            for (int e = 0; e < 4; ++e) sE[e] = sE[e] + __builtin_fabs(eE[e]);
#pragma unroll
            for (int e = 0; e < 4; ++e) sO[e] = sO[e] + __builtin_fabs(eO[e]);
        }
        if (live) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int64_t row = i0 + (int64_t)tile * 16 + g + 4 * e;
                store16_nt(out + row * ld + jE, d2{sE[e] + dcoef * cE[e], sO[e] + dcoef * cO[e]});
            }
        }
#pragma unroll
        for (int q = 0; q < (16 * K + 63) / 64; ++q) {
            const int e = lane + 64 * q;
            if (e < 16 * K) L.a[cur ^ 1][e / K][e % K] = nxt[q];
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// units: u -> (column tile u % col_tiles, row chunk u / col_tiles).  Of every `period` consecutive units the first `n_mfma` take the
// MFMA form, the others the VALU form.  The two forms are TWO KERNELS on two streams (one register allocation each: 126 VGPRs for
// the VALU loop, <= 256 for the MFMA one; in one kernel both would run at the larger footprint), co-resident on the CUs.
__device__ __forceinline__ bool unit_of(int64_t slot, int n_mine, int first, int period, int64_t n_units, int col_tiles, int64_t n_m, int *tile,
                                        int64_t *i0) {
    const int64_t u = slot / n_mine * period + first + slot % n_mine;
    if (u >= n_units) return false;
    *tile = (int)(u % col_tiles);
    int64_t r0 = u / col_tiles * ROWS;
    if (r0 + ROWS > n_m) r0 = n_m - ROWS;
    *i0 = r0;
    return true;
}

__global__ __launch_bounds__(256) void valu_kernel(const double *__restrict__ A, const double *__restrict__ R, const double *__restrict__ axy,
                                                    const double *__restrict__ rxy, int64_t n_r, int64_t n_m, double dcoef,
                                                    double *__restrict__ out, int64_t ld, int col_tiles, int64_t n_units, int n_mfma, int period) {
    int tile;
    int64_t i0;
    if (!unit_of(blockIdx.x, period - n_mfma, n_mfma, period, n_units, col_tiles, n_m, &tile, &i0)) return;
    valu_unit(A, R, axy, rxy, n_r, dcoef, out, ld, i0, (int64_t)tile * UNIT_COLS);
}

__global__ __launch_bounds__(256) void mfma_kernel(
    const double *__restrict__ AX, const double *__restrict__ R, const double *__restrict__ rxy, const double *__restrict__ consts, int64_t n_r,
    int64_t n_m, double dcoef, double *__restrict__ out, int64_t ld, int col_tiles, int64_t n_units, int n_mfma, int period) {
    int tile;
    int64_t i0;
    if (!unit_of(blockIdx.x >> 2, n_mfma, 0, period, n_units, col_tiles, n_m, &tile, &i0)) return;
    __shared__ WaveLds lds[4];
    mfma_unit(AX, R, rxy, n_r, dcoef, out, ld, i0, (int64_t)tile * UNIT_COLS + (blockIdx.x & 3) * 128 + (threadIdx.x >> 6) * 32, lds[threadIdx.x >> 6]);
}

__global__ void checksum_kernel(const unsigned long long *__restrict__ p, int64_t n_rows, int64_t n_cols, int64_t ld, unsigned long long *out) {
    unsigned long long x = 0, s = 0;
    const int64_t total = n_rows * n_cols;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const unsigned long long v = p[(e / n_cols) * ld + (e % n_cols)];
        x ^= v * (unsigned long long)(2 * (e % 1000003) + 1);
        s += v;
    }
    atomicXor(out, x);
    atomicAdd(out + 1, s);
}

int main(int argc, char **argv) {
    const int n_mfma = argc > 1 ? atoi(argv[1]) : 0, period = argc > 2 ? atoi(argv[2]) : 1;
    const int64_t n = argc > 3 ? atoll(argv[3]) : 100000;
    const double seconds = argc > 4 ? atof(argv[4]) : 3.0;
    const int valu_lds = argc > 5 ? atoi(argv[5]) : 0;       // unused dynamic LDS of the VALU kernel: caps its blocks per CU (room for an MFMA block)
    const int mfma_lds = argc > 6 ? atoi(argv[6]) : 0;       // unused dynamic LDS of the MFMA kernel (50.7 KB of its own): 32768 -> one block per CU
    const int data_mode = argc > 7 ? atoi(argv[7]) : 0;      // 0: uniform [0,100) with exact zeros; 1: Dirichlet(0.3) rows x 100 (bench.py's synthetic sections)
    if (n_mfma < 0 || period < 1 || n_mfma > period || n < ROWS || (n & 1)) { fprintf(stderr, "bad arguments\n"); return 2; }
    const int64_t ld = n;
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<double> uni(0.0, 100.0);
    std::vector<double> A((size_t)n * T), R((size_t)n * T), axy((size_t)n * 2), rxy((size_t)n * 2), AX((size_t)n * (T + 2));
    if (data_mode == 0) {
        for (auto &v : A) v = (rng() % 7 == 0) ? 0.0 : uni(rng);            // probability rows have exact zeros
        for (auto &v : R) v = (rng() % 7 == 0) ? 0.0 : uni(rng);
    } else {
        std::gamma_distribution<double> gam(0.3, 1.0);
        for (auto *M : {&A, &R})
            for (int64_t i = 0; i < n; ++i) {
                double row[T], sum = 0.0;
                for (int t = 0; t < T; ++t) { row[t] = gam(rng); sum += row[t]; }
                for (int t = 0; t < T; ++t) (*M)[(size_t)i * T + t] = sum > 0.0 ? row[t] / sum * 100.0 : 0.0;
            }
    }
    for (int64_t i = 0; i < n; i += 97)                                  // some cells identical in both sections: differences of exactly 0
        for (int t = 0; t < T; ++t) R[(size_t)i * T + t] = A[(size_t)i * T + t];
    for (auto &v : axy) v = uni(rng) * 30.0;
    for (auto &v : rxy) v = uni(rng) * 30.0;
    for (int64_t i = 0; i < n; ++i) {
        for (int t = 0; t < T; ++t) AX[(size_t)i * (T + 2) + t] = A[(size_t)i * T + t];
        AX[(size_t)i * (T + 2) + T] = axy[2 * i];
        AX[(size_t)i * (T + 2) + T + 1] = axy[2 * i + 1];
    }
    std::vector<double> consts(2 * (T + 2), 0.0);
    for (int t = 0; t < T + 2; ++t) consts[t] = 1.0;
    double *dA, *dAX, *dR, *daxy, *drxy, *dconsts, *dout;
    unsigned long long *dsum;
    auto up = [](double **d, const std::vector<double> &h) {
        CK(hipMalloc(reinterpret_cast<void **>(d), h.size() * 8));
        CK(hipMemcpy(*d, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    };
    up(&dA, A); up(&dAX, AX); up(&dR, R); up(&daxy, axy); up(&drxy, rxy); up(&dconsts, consts);
    CK(hipMalloc(reinterpret_cast<void **>(&dout), (size_t)n * ld * 8));
    CK(hipMalloc(reinterpret_cast<void **>(&dsum), 16));
    const int col_tiles = (int)((n + UNIT_COLS - 1) / UNIT_COLS);
    const int64_t chunks = (n + ROWS - 1) / ROWS, n_units = chunks * col_tiles;
    const int64_t periods = (n_units + period - 1) / period;
    const int64_t valu_blocks = periods * (period - n_mfma), mfma_blocks = periods * n_mfma * 4;
    const double dcoef = 0.001;
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    hipEvent_t fork, join;
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    auto launch = [&] {      // both kernels start together (s1 waits for s0's fork) and s0 ends after both
        CK(hipEventRecord(fork, s0));
        CK(hipStreamWaitEvent(s1, fork, 0));
        if (mfma_blocks)
            hipLaunchKernelGGL(mfma_kernel, dim3((unsigned)mfma_blocks), dim3(256), mfma_lds, s1, dAX, dR, drxy, dconsts, n, n, dcoef, dout, ld, col_tiles, n_units,
                               n_mfma, period);
        if (valu_blocks)
            hipLaunchKernelGGL(valu_kernel, dim3((unsigned)valu_blocks), dim3(256), valu_lds, s0, dA, dR, daxy, drxy, n, n, dcoef, dout, ld, col_tiles, n_units,
                               n_mfma, period);
        CK(hipEventRecord(join, s1));
        CK(hipStreamWaitEvent(s0, join, 0));
    };
    if (mfma_lds) CK(hipFuncSetAttribute(reinterpret_cast<const void *>(mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, mfma_lds));
    if (valu_lds) CK(hipFuncSetAttribute(reinterpret_cast<const void *>(valu_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, valu_lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    launch();
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    printf("TIMED_START\n");
    fflush(stdout);
    double total_ms = 0;
    int reps = 0;
    while (total_ms < seconds * 1e3 && reps < 2000) {
        CK(hipEventRecord(e0, s0));
        launch();
        CK(hipEventRecord(e1, s0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        total_ms += ms;
        ++reps;
    }
    const double ms = total_ms / reps;
    printf("TIMED_END\n");
    fflush(stdout);
    const double bytes = 8.0 * n * n + 8.0 * (T + 2) * 2 * n;
    CK(hipDeviceSynchronize());
    CK(hipMemset(dsum, 0, 16));
    hipLaunchKernelGGL(checksum_kernel, dim3(4096), dim3(256), 0, 0, reinterpret_cast<const unsigned long long *>(dout), n, n, ld, dsum);
    unsigned long long hs[2];
    CK(hipMemcpy(hs, dsum, 16, hipMemcpyDeviceToHost));
    // sampled rows against the host's left-to-right loop (the reference's expression, src/same.py:1183-1188)
    int64_t bad = 0, checked = 0;
    std::vector<double> row((size_t)n);
    for (int64_t i : {(int64_t)0, (int64_t)17, (int64_t)97, n / 2 + 3, n - 1}) {
        CK(hipMemcpy(row.data(), dout + i * ld, (size_t)n * 8, hipMemcpyDeviceToHost));
        for (int64_t j = 0; j < n; ++j) {
            volatile double s = 0.0;
            for (int t = 0; t < T; ++t) { volatile double d = A[(size_t)i * T + t] - R[(size_t)j * T + t]; s = s + std::fabs(d); }
            volatile double dx = axy[2 * i] - rxy[2 * j], dy = axy[2 * i + 1] - rxy[2 * j + 1];
            volatile double dc = std::fabs(dx) + std::fabs(dy);
            volatile double prod = dcoef * dc;
            const double want = s + prod;
            ++checked;
            if (!(want == row[(size_t)j]) ) ++bad;
        }
    }
    printf("{\"data\": %d, \"mfma_kernel_lds_pad\": %d, \"valu_kernel_lds_pad\": %d, \"mfma_units\": %d, \"period\": %d, \"mfma_fraction\": %.3f, \"n\": %lld, \"launches\": %d, \"ms\": %.4f, \"GBs\": %.1f, \"frac_of_8TBs\": %.4f, "
           "\"checksum_xor\": \"%016llx\", \"checksum_sum\": \"%016llx\", \"host_rows_checked\": %lld, \"host_mismatches\": %lld}\n",
           data_mode, mfma_lds, valu_lds, n_mfma, period, (double)n_mfma / period, (long long)n, reps, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000.0, hs[0], hs[1],
           (long long)checked, (long long)bad);
    return bad ? 1 : 0;
}
