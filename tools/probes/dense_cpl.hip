// dense_cpl.hip -- the fp64 dense cost loop with ONE column per lane instead of two: twice the waves per SIMD (round 5).
//
// dense_cost_kernel<double,20,2> (same_amd/csrc/cost.hip) keeps two columns of R per lane (80 VGPRs of r values, 117-126 in all: four waves
// per SIMD) and stores 16 bytes per lane and row.  With one column per lane the r values take 40 VGPRs, eight waves fit a SIMD (more
// latency hiding: VALU busy 0.92 -> ?), the store is 8 bytes per lane (a wave still writes whole 512-byte runs) and every wave loads the
// row's 22 scalars for 64 outputs instead of 128.  The instruction count per output is the same 44.  Under the 1 400 W cap the question is
// whether busy x clock moves (profiles/r03_dense_bound.md: fewer waves lowered it slightly).  Both forms as kernels of one program
// (the two-column one is the shipped loop, copied), 100k x 100k fp64 into a plain 80 GB block, T = 20; per form the mean launch time and
// xor / sum checksums of all outputs (equal <=> bit-identical).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -o tools/probes/dense_cpl tools/probes/dense_cpl.hip
// Run:   ./dense_cpl [n=100000] [seconds=3]
#include <hip/hip_runtime.h>

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

constexpr int T = 20, ROWS = 256;
typedef double d2 __attribute__((ext_vector_type(2)));
typedef int i4 __attribute__((ext_vector_type(4)));
typedef int i2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void touch(double v) { asm volatile("" ::"s"(v)); }
__device__ __forceinline__ void store16_nt_saddr(char *row_uniform, unsigned lane_byte_off, d2 v) {
    const i4 bits = __builtin_bit_cast(i4, v);
    asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" ::"v"(lane_byte_off), "v"(bits), "s"(row_uniform) : "memory");
}
__device__ __forceinline__ void store8_nt_saddr(char *row_uniform, unsigned lane_byte_off, double v) {
    const i2 bits = __builtin_bit_cast(i2, v);
    asm volatile("global_store_dwordx2 %0, %1, %2 nt" ::"v"(lane_byte_off), "v"(bits), "s"(row_uniform) : "memory");
}

template <int CPL, int WAVES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES))) void dense_f64_kernel(
    const double *__restrict__ A, const double *__restrict__ R, const double *__restrict__ axy, const double *__restrict__ rxy, int64_t n_r, int64_t n_m,
    double dcoef, double *__restrict__ out, int64_t ld, int col_tiles) {
    constexpr int H = T / 2;
    const int tile = blockIdx.x % col_tiles;
    int64_t i0 = (int64_t)(blockIdx.x / col_tiles) * ROWS;
    if (i0 + ROWS > n_m) i0 = n_m - ROWS;
    const int64_t j0 = ((int64_t)tile * 256 + threadIdx.x) * CPL;
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
    auto acc = [&](double a, int t, double (&s)[CPL]) {
        double dd[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) dd[c] = a - r[c][t];
#pragma unroll
        for (int c = 0; c < CPL; ++c) s[c] = s[c] + __builtin_fabs(dd[c]);
    };
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
        for (int t = 0; t < H; ++t) acc(h0[t], t, s);
        touch(ax);
        __builtin_amdgcn_sched_barrier(0);
        {
            const double *__restrict__ an = (q + 1 >= ROWS) ? arow : arow + T;
#pragma unroll
            for (int t = 0; t < H; ++t) h0[t] = an[t];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = H; t < T; ++t) acc(h1[t - H], t, s);
        double v[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const double dc = __builtin_fabs(ax - rx[c]) + __builtin_fabs(ay - ry[c]);
            v[c] = s[c] + dcoef * dc;
        }
        if constexpr (CPL == 2)
            store16_nt_saddr(orow, lane_off, d2{v[0], v[1]});
        else
            store8_nt_saddr(orow, lane_off, v[0]);
        orow += row_pitch;
        arow += T;
        axyrow += 2;
    }
}

__global__ void checksum_kernel(const unsigned long long *__restrict__ p, int64_t total, unsigned long long *out) {
    unsigned long long x = 0, s = 0;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const unsigned long long v = p[e];
        x ^= v * (unsigned long long)(2 * (e % 1000003) + 1);
        s += v;
    }
    atomicXor(out, x);
    atomicAdd(out + 1, s);
}

int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 100000;
    const double seconds = argc > 2 ? atof(argv[2]) : 3.0;
    if (n < ROWS || (n & 1)) { fprintf(stderr, "n must be even, at least %d\n", ROWS); return 2; }
    std::mt19937_64 rng(777);
    std::gamma_distribution<double> gam(0.3, 1.0);
    std::uniform_real_distribution<double> uni(0.0, 3000.0);
    std::vector<double> A((size_t)n * T), R((size_t)n * T), axy((size_t)n * 2), rxy((size_t)n * 2);
    for (auto *M : {&A, &R})
        for (int64_t i = 0; i < n; ++i) {                       // Dirichlet(0.3) rows x 100, as bench.py's sections
            double row[T], sum = 0.0;
            for (int t = 0; t < T; ++t) { row[t] = gam(rng); sum += row[t]; }
            for (int t = 0; t < T; ++t) (*M)[(size_t)i * T + t] = sum > 0.0 ? row[t] / sum * 100.0 : 0.0;
        }
    for (auto &v : axy) v = uni(rng);
    for (auto &v : rxy) v = uni(rng);
    double *dA, *dR, *daxy, *drxy, *dout;
    unsigned long long *dsum;
    auto up = [](double **d, const std::vector<double> &h) {
        CK(hipMalloc(reinterpret_cast<void **>(d), h.size() * 8));
        CK(hipMemcpy(*d, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    };
    up(&dA, A); up(&dR, R); up(&daxy, axy); up(&drxy, rxy);
    CK(hipMalloc(reinterpret_cast<void **>(&dout), (size_t)n * n * 8));
    CK(hipMalloc(reinterpret_cast<void **>(&dsum), 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const char *names[3] = {"two columns per lane, 4 waves (the shipped loop)", "one column per lane, 8 waves", "one column per lane, 4 waves"};
    unsigned long long sums[3][2];
    for (int form = 0; form < 3; ++form) {
        const int cpl = form == 0 ? 2 : 1;
        const int col_tiles = (int)((n + 256 * cpl - 1) / (256 * cpl));
        const int64_t blocks = (n + ROWS - 1) / ROWS * col_tiles;
        auto launch = [&] {
            if (form == 0)
                hipLaunchKernelGGL((dense_f64_kernel<2, 4>), dim3((unsigned)blocks), dim3(256), 0, 0, dA, dR, daxy, drxy, n, n, 0.001, dout, n, col_tiles);
            else if (form == 1)
                hipLaunchKernelGGL((dense_f64_kernel<1, 8>), dim3((unsigned)blocks), dim3(256), 0, 0, dA, dR, daxy, drxy, n, n, 0.001, dout, n, col_tiles);
            else
                hipLaunchKernelGGL((dense_f64_kernel<1, 4>), dim3((unsigned)blocks), dim3(256), 0, 0, dA, dR, daxy, drxy, n, n, 0.001, dout, n, col_tiles);
        };
        launch();
        CK(hipDeviceSynchronize());
        CK(hipGetLastError());
        double total_ms = 0;
        int reps = 0;
        while (total_ms < seconds * 1e3 && reps < 4000) {
            CK(hipEventRecord(e0, 0));
            launch();
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            total_ms += ms;
            ++reps;
        }
        CK(hipMemset(dsum, 0, 16));
        hipLaunchKernelGGL(checksum_kernel, dim3(4096), dim3(256), 0, 0, reinterpret_cast<const unsigned long long *>(dout), n * n, dsum);
        CK(hipMemcpy(sums[form], dsum, 16, hipMemcpyDeviceToHost));
        const double ms = total_ms / reps, bytes = 8.0 * n * n + 8.0 * (T + 2) * 2 * n;
        printf("%-50s %4d launches  %8.3f ms  %7.1f GB/s  %.3f of 8 TB/s  checksums %016llx %016llx\n", names[form], reps, ms, bytes / ms / 1e6,
               bytes / ms / 1e6 / 8000.0, sums[form][0], sums[form][1]);
    }
    const bool same = sums[0][0] == sums[1][0] && sums[0][1] == sums[1][1] && sums[0][0] == sums[2][0] && sums[0][1] == sums[2][1];
    printf("outputs %s\n", same ? "bit-identical" : "DIFFER");
    return same ? 0 : 1;
}
