// pk_sub_f32.hip -- the fp32 dense cost loop with the subtraction PACKED by hand (round 5).
//
// dense_cost_kernel<float,20,4> (same_amd/csrc/cost.hip) issues, per type and per lane's four columns, 4 v_sub_f32 (SGPR row value minus
// the column's value) and 4 v_add_f32 with the |d| input modifier.  The form asked for: the subtraction of a column PAIR as one
//     v_pk_add_f32 d[0:1], s[a], v[r0:r1]  op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]      (a broadcast to both halves, r negated)
// and the accumulate left as two plain `v_add_f32 acc, acc, |d|` (the packed encoding has no abs): 3 VALU instructions per two elements
// instead of 4, the same IEEE roundings.  Here: both forms as kernels of one program (the plain one is the shipped loop, copied), 100k x
// 100k fp32 into a plain 40 GB block, T = 20; per form the mean launch time and xor / sum checksums of all outputs (equal <=> bit-identical).
// The explicit float2 arithmetic makes hipcc emit the packed instruction with -fno-slp-vectorize in force (checked in the ISA: see the log).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -o tools/probes/pk_sub_f32 tools/probes/pk_sub_f32.hip
// Run:   ./pk_sub_f32 [n=100000] [seconds=3]
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

constexpr int T = 20, ROWS = 256, CPL = 4;
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void touch(float v) { asm volatile("" ::"s"(v)); }
__device__ __forceinline__ void store16_nt_saddr(char *row_uniform, unsigned lane_byte_off, f4 v) {
    const i4 bits = __builtin_bit_cast(i4, v);
    asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" ::"v"(lane_byte_off), "v"(bits), "s"(row_uniform) : "memory");
}

template <bool PK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void dense_f32_kernel(const float *__restrict__ A, const float *__restrict__ R, const float *__restrict__ axy,
                                                         const float *__restrict__ rxy, int64_t n_r, int64_t n_m, float dcoef,
                                                         float *__restrict__ out, int64_t ld, int col_tiles) {
    constexpr int H = T / 2;
    const int tile = blockIdx.x % col_tiles;
    int64_t i0 = (int64_t)(blockIdx.x / col_tiles) * ROWS;
    if (i0 + ROWS > n_m) i0 = n_m - ROWS;
    const int64_t j0 = ((int64_t)tile * 256 + threadIdx.x) * CPL;
    float r[CPL][T], rx[CPL], ry[CPL];
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
    float h0[H];
    const float *__restrict__ arow = A + i0 * T;
    const float *__restrict__ axyrow = axy + 2 * i0;
#pragma unroll
    for (int t = 0; t < H; ++t) h0[t] = arow[t];
    char *orow = reinterpret_cast<char *>(out + i0 * ld);
    const unsigned lane_off = (unsigned)(j0 * sizeof(float));
    const int64_t row_pitch = ld * (int64_t)sizeof(float);
    auto acc = [&](float a, int t, float (&s)[CPL]) {
        if constexpr (PK) {
            // the column pairs (0,1) and (2,3) subtracted by ONE instruction each; the accumulate stays plain (|d| as an input modifier)
            const f2 d01 = f2{a, a} - f2{r[0][t], r[1][t]}, d23 = f2{a, a} - f2{r[2][t], r[3][t]};
            s[0] = s[0] + __builtin_fabsf(d01.x);
            s[1] = s[1] + __builtin_fabsf(d01.y);
            s[2] = s[2] + __builtin_fabsf(d23.x);
            s[3] = s[3] + __builtin_fabsf(d23.y);
        } else {
            float dd[CPL];
#pragma unroll
            for (int c = 0; c < CPL; ++c) dd[c] = a - r[c][t];
#pragma unroll
            for (int c = 0; c < CPL; ++c) s[c] = s[c] + __builtin_fabsf(dd[c]);
        }
    };
    for (int q = 0; q < ROWS; ++q) {
        touch(h0[0]);
        __builtin_amdgcn_sched_barrier(0);
        float h1[T - H];
#pragma unroll
        for (int t = H; t < T; ++t) h1[t - H] = arow[t];
        const float ax = axyrow[0], ay = axyrow[1];
        __builtin_amdgcn_sched_barrier(0);
        float s[CPL] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < H; ++t) acc(h0[t], t, s);
        touch(ax);
        __builtin_amdgcn_sched_barrier(0);
        {
            const float *__restrict__ an = (q + 1 >= ROWS) ? arow : arow + T;
#pragma unroll
            for (int t = 0; t < H; ++t) h0[t] = an[t];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = H; t < T; ++t) acc(h1[t - H], t, s);
        f4 v;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const float dc = __builtin_fabsf(ax - rx[c]) + __builtin_fabsf(ay - ry[c]);
            v[c] = s[c] + dcoef * dc;
        }
        store16_nt_saddr(orow, lane_off, v);
        orow += row_pitch;
        arow += T;
        axyrow += 2;
    }
}

__global__ void checksum_kernel(const unsigned *__restrict__ p, int64_t total, unsigned long long *out) {
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
    if (n < ROWS || (n & 3)) { fprintf(stderr, "n must be a multiple of 4, at least %d\n", ROWS); return 2; }
    std::mt19937_64 rng(777);
    std::gamma_distribution<double> gam(0.3, 1.0);
    std::uniform_real_distribution<double> uni(0.0, 3000.0);
    std::vector<float> A((size_t)n * T), R((size_t)n * T), axy((size_t)n * 2), rxy((size_t)n * 2);
    for (auto *M : {&A, &R})
        for (int64_t i = 0; i < n; ++i) {                       // Dirichlet(0.3) rows x 100, as bench.py's sections
            double row[T], sum = 0.0;
            for (int t = 0; t < T; ++t) { row[t] = gam(rng); sum += row[t]; }
            for (int t = 0; t < T; ++t) (*M)[(size_t)i * T + t] = (float)(sum > 0.0 ? row[t] / sum * 100.0 : 0.0);
        }
    for (auto &v : axy) v = (float)uni(rng);
    for (auto &v : rxy) v = (float)uni(rng);
    float *dA, *dR, *daxy, *drxy, *dout;
    unsigned long long *dsum;
    auto up = [](float **d, const std::vector<float> &h) {
        CK(hipMalloc(reinterpret_cast<void **>(d), h.size() * 4));
        CK(hipMemcpy(*d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    };
    up(&dA, A); up(&dR, R); up(&daxy, axy); up(&drxy, rxy);
    CK(hipMalloc(reinterpret_cast<void **>(&dout), (size_t)n * n * 4));
    CK(hipMalloc(reinterpret_cast<void **>(&dsum), 16));
    const int col_tiles = (int)((n + 256 * CPL - 1) / (256 * CPL));
    const int64_t blocks = (n + ROWS - 1) / ROWS * col_tiles;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    unsigned long long sums[2][2];
    for (int form = 0; form < 2; ++form) {
        auto launch = [&] {
            if (form)
                hipLaunchKernelGGL(dense_f32_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, 0, dA, dR, daxy, drxy, n, n, 0.001f, dout, n, col_tiles);
            else
                hipLaunchKernelGGL(dense_f32_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, 0, dA, dR, daxy, drxy, n, n, 0.001f, dout, n, col_tiles);
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
        hipLaunchKernelGGL(checksum_kernel, dim3(4096), dim3(256), 0, 0, reinterpret_cast<const unsigned *>(dout), n * n, dsum);
        CK(hipMemcpy(sums[form], dsum, 16, hipMemcpyDeviceToHost));
        const double ms = total_ms / reps, bytes = 4.0 * n * n + 4.0 * (T + 2) * 2 * n;
        printf("%-28s %4d launches  %8.3f ms  %7.1f GB/s  %.3f of 8 TB/s  checksums %016llx %016llx\n",
               form ? "packed subtraction (PK)" : "plain (the shipped loop)", reps, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000.0, sums[form][0], sums[form][1]);
    }
    const bool same = sums[0][0] == sums[1][0] && sums[0][1] == sums[1][1];
    printf("outputs %s\n", same ? "bit-identical" : "DIFFER");
    return same ? 0 : 1;
}
