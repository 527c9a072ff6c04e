// Dev probe (not product): store rate when one launch writes into TWO 8 GB blocks at once (4 GB in each), for every pair
// of blocks on the card.  If the rate is limited per physical region, pairs from different regions run faster than pairs
// from the same one, and the matrix shows the regions.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/probes/vram_pairs tools/probes/vram_pairs.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));

// regions 0..3 (XCDs 0..3) write rows [0, n_rows) of A, regions 4..7 rows [0, n_rows) of B
__global__ __launch_bounds__(256) void pair_kernel(char *A, char *B, uint64_t pitch, int n_rows, int n_tiles, uint64_t row_bytes, int rpb,
                                                   int chunks, unsigned per) {
    const unsigned b = blockIdx.x, region = b & 7u, k = b >> 3;
    if (k >= per) return;
    char *base = region < 4 ? A : B;
    const uint64_t lin = (uint64_t)(region & 3u) * per + k;
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
    const int rows = 10000, half = rows / 2;
    const uint64_t row_bytes = 800000, blk = (uint64_t)rows * row_bytes;
    size_t fr, tot;
    std::vector<char *> bufs;
    while (true) {
        CK(hipMemGetInfo(&fr, &tot));
        if (fr < blk + (4ull << 30)) break;
        char *p; if (hipMalloc(&p, blk) != hipSuccess) { (void)hipGetLastError(); break; }
        bufs.push_back(p);
    }
    const int nb = (int)bufs.size();
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int n_tiles = (int)((row_bytes + 4095) / 4096), rpb = 64, chunks = (half + rpb - 1) / rpb;
    const unsigned per = (unsigned)(((uint64_t)chunks * n_tiles + 3) / 4);
    auto run = [&](char *A, char *B) {
        float best = 1e30f;
        for (int r = 0; r < 4; ++r) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(pair_kernel, dim3(per * 8), dim3(256), 0, 0, A, B, row_bytes, half, n_tiles, row_bytes, rpb, chunks, per);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r && ms < best) best = ms;
        }
        return blk / best * 1e-6 / 100.0;   // in units of 100 GB/s
    };
    printf("%d blocks of 8 GB; entry (i, j) = rate in 100 GB/s writing the first 4 GB of block i and of block j at once; (i, i) = first and second half of block i\n   ", nb);
    for (int j = 0; j < nb; ++j) printf("%3d", j); printf("\n");
    for (int i = 0; i < nb; ++i) {
        printf("%3d", i);
        for (int j = 0; j < nb; ++j) {
            if (j < i) { printf("   "); continue; }
            printf("%3.0f", run(bufs[i], j == i ? bufs[i] + (uint64_t)half * row_bytes : bufs[j]));
        }
        printf("\n"); fflush(stdout);
    }
    return 0;
}
