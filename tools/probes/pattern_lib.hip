// Dev probe (not product): pattern_store.hip's kernel behind a C call, so a Python probe can run it on the very buffers the
// product kernels write (placement matters).  build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o tools/probes/libpattern.so tools/probes/pattern_lib.hip
#include <hip/hip_runtime.h>
#include <cstdint>
typedef double v2d __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void pattern_kernel(char *base, uint64_t pitch, int n_rows, int n_tiles, uint64_t row_bytes, int rpb,
                                                      int chunks, unsigned R, unsigned per) {
    const unsigned b = blockIdx.x, region = b % R, k = b / R;
    if (k >= per) return;
    const uint64_t lin = (uint64_t)region * per + k;
    const int tile = (int)(lin % n_tiles), chunk = (int)(lin / n_tiles);
    if (chunk >= chunks) return;
    int r0 = chunk * rpb; if (r0 + rpb > n_rows) r0 = n_rows - rpb;
    const uint64_t col = (uint64_t)tile * 4096 + threadIdx.x * 16;
    if (col + 16 > row_bytes) return;
    char *p = base + (uint64_t)r0 * pitch + col;
    const v2d val = {1.0, 2.0};
    for (int r = 0; r < rpb; ++r, p += pitch) __builtin_nontemporal_store(val, (v2d *)p);
}

extern "C" float pattern_time(void *buf, uint64_t pitch, int n_rows, uint64_t row_bytes, int rpb, unsigned R, int reps) {
    const int n_tiles = (int)((row_bytes + 4095) / 4096), chunks = (n_rows + rpb - 1) / rpb;
    const uint64_t total = (uint64_t)chunks * n_tiles;
    const unsigned per = (unsigned)((total + R - 1) / R);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(pattern_kernel, dim3(per * R), dim3(256), 0, 0, (char *)buf, pitch, n_rows, n_tiles, row_bytes, rpb, chunks, R, per);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (r && ms < best) best = ms;
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    return hipGetLastError() == hipSuccess ? best : -1.0f;
}
