// Dev probe: can an fp64-VALU-only kernel and an HBM store stream overlap on MI355X, or do they add up?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void store_stream(double *out, size_t n2) {  // n2 = number of double2
    typedef double d2 __attribute__((ext_vector_type(2)));
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    d2 v = {1.0 * threadIdx.x, 2.0};
    for (; i < n2; i += stride) __builtin_nontemporal_store(v, reinterpret_cast<d2 *>(out) + i);
}

__global__ __launch_bounds__(256) void valu_only(double *sink, int iters, double a0) {
    double r[8], acc[2] = {0, 0};
    for (int q = 0; q < 8; ++q) r[q] = threadIdx.x * 0.37 + q;
    double a = a0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            double d0 = a - r[q], d1 = a - r[(q + 3) & 7];
            acc[0] = acc[0] + __builtin_fabs(d0);
            acc[1] = acc[1] + __builtin_fabs(d1);
        }
        a += 1e-9;
    }
    if (acc[0] + acc[1] == -1.0) sink[threadIdx.x] = acc[0];
}

int main(int argc, char **argv) {
    const size_t bytes = 80ull << 30;
    double *buf, *sink;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 4096));
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    hipEvent_t e0, e1, f0, f1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
    // 7.03e9 wave-instructions of fp64 adds in total = blocks*4 waves * iters*32
    const int vblocks = argc > 1 ? atoi(argv[1]) : 1024, iters = (int)(7.03e9 / (vblocks * 4.0 * 32.0));
    const int sblocks = argc > 2 ? atoi(argv[2]) : 2048;
    for (int rep = 0; rep < 3; ++rep) {
        float ta, tb, tc1, tc2;
        CK(hipEventRecord(e0, s1)); hipLaunchKernelGGL(store_stream, dim3(sblocks), dim3(256), 0, s1, buf, bytes / 16); CK(hipEventRecord(e1, s1));
        CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ta, e0, e1));
        CK(hipEventRecord(f0, s2)); hipLaunchKernelGGL(valu_only, dim3(vblocks), dim3(256), 0, s2, sink, iters, 3.0); CK(hipEventRecord(f1, s2));
        CK(hipEventSynchronize(f1)); CK(hipEventElapsedTime(&tb, f0, f1));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, s1)); CK(hipEventRecord(f0, s2));
        hipLaunchKernelGGL(store_stream, dim3(sblocks), dim3(256), 0, s1, buf, bytes / 16);
        hipLaunchKernelGGL(valu_only, dim3(vblocks), dim3(256), 0, s2, sink, iters, 3.0);
        CK(hipEventRecord(e1, s1)); CK(hipEventRecord(f1, s2));
        CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&tc1, e0, e1)); CK(hipEventElapsedTime(&tc2, f0, f1));
        printf("store alone %.2f ms (%.0f GB/s)  valu alone %.2f ms  together: store %.2f ms, valu %.2f ms\n", ta, bytes / ta / 1e6, tb, tc1, tc2);
    }
    return 0;
}
