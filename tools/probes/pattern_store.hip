// Dev probe (not product): the dense build's store pattern with the spread of concurrently active blocks as a knob.
// Matrix n_rows x (n_tiles * 4 KB) at row pitch `pitch`; block = one 4 KB column tile x `rpb` rows, top to bottom.  The
// linear (chunk, tile) space (tile fastest) is cut into R contiguous regions and block b works in region b % R, so R = 1
// is "all blocks on neighbouring rows", R = 8 is the product's map 2 (one region per XCD), larger R spreads wider.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/probes/pattern_store tools/probes/pattern_store.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
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

int main(int argc, char **argv) {
    const int n = 100000, n_buf = argc > 1 ? atoi(argv[1]) : 2;
    const uint64_t row_bytes = (uint64_t)n * 8, max_pitch = 1u << 20;
    const int n_tiles = (int)((row_bytes + 4095) / 4096);
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<char *> bufs(n_buf);
    for (auto &b : bufs) CK(hipMalloc(&b, (uint64_t)n * max_pitch));
    auto timed = [&](auto launch) {
        float best = 1e30f;
        for (int r = 0; r < 3; ++r) {
            CK(hipEventRecord(e0, st)); launch(); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r && ms < best) best = ms;
        }
        return best;
    };
    const unsigned Rs[] = {1, 8, 16, 32, 64, 128, 256, 512, 2048};
    for (char *b : bufs) {
        printf("buffer %p\n", (void *)b);
        for (uint64_t pitch : {(uint64_t)800000, (uint64_t)802816, (uint64_t)1048576})
            for (int rpb : {64, 256}) {
                const int chunks = (n + rpb - 1) / rpb;
                const uint64_t total = (uint64_t)chunks * n_tiles;
                printf("  pitch %7llu rpb %3d:", (unsigned long long)pitch, rpb);
                for (unsigned R : Rs) {
                    const unsigned per = (unsigned)((total + R - 1) / R);
                    float ms = timed([&] { hipLaunchKernelGGL(pattern_kernel, dim3(per * R), dim3(256), 0, st, b, pitch, n, n_tiles, row_bytes, rpb, chunks, R, per); });
                    printf("  R=%u %.2f", R, ms);
                }
                printf("  ms\n"); fflush(stdout);
            }
    }
    return 0;
}
