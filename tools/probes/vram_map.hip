// Dev probe (not product): is the streaming-store rate a property of WHERE in VRAM a buffer lies?  Allocates nearly all of
// the card in 8 GB blocks and times the dense build's store pattern (10000 rows x 800000 B, one region per XCD) into each.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/probes/vram_map tools/probes/vram_map.hip
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
    const int rows = argc > 1 ? atoi(argv[1]) : 10000;           // 10000 rows x 800000 B = 8 GB per block
    const uint64_t row_bytes = 800000, blk = (uint64_t)rows * row_bytes;
    size_t fr, tot; CK(hipMemGetInfo(&fr, &tot));
    printf("VRAM free %.1f GB of %.1f GB; block %.1f GB\n", fr * 1e-9, tot * 1e-9, blk * 1e-9);
    std::vector<char *> bufs;
    while (true) {
        CK(hipMemGetInfo(&fr, &tot));
        if (fr < blk + (4ull << 30)) break;
        char *p; if (hipMalloc(&p, blk) != hipSuccess) { (void)hipGetLastError(); break; }
        bufs.push_back(p);
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int n_tiles = (int)((row_bytes + 4095) / 4096), rpb = 64, chunks = (rows + rpb - 1) / rpb;
    const unsigned R = 8, per = (unsigned)(((uint64_t)chunks * n_tiles + R - 1) / R);
    for (int pass = 0; pass < 2; ++pass) {
        printf("pass %d, GB/s per block in allocation order:\n", pass);
        for (size_t i = 0; i < bufs.size(); ++i) {
            float best = 1e30f;
            for (int r = 0; r < 6; ++r) {
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(pattern_kernel, dim3(per * R), dim3(256), 0, 0, bufs[i], row_bytes, rows, n_tiles, row_bytes, rpb, chunks, R, per);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r && ms < best) best = ms;
            }
            printf(" %5.0f", blk / best * 1e-6); if (i % 12 == 11) printf("\n");
        }
        printf("\n");
    }
    printf("addresses:"); for (auto p : bufs) printf(" %p", (void *)p); printf("\n");
    return 0;
}
