// Dev probe (not product): does the streaming-store rate depend on WHICH XCD writes WHICH interleave unit of memory?
// Grid = 8 * S blocks; block b runs on XCD b % 8 (round-robin dispatch; checked with the XCC_ID register).  Memory is cut
// into units of G bytes, eight consecutive units form a group.  mode "own": block (xcd, slot) writes only unit (xcd + rot) % 8
// of the groups it owns.  mode "all": every block writes whole groups (uniform traffic).  Both write the same bytes overall.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/probes/xcd_store tools/probes/xcd_store.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef double v2d __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void own_kernel(char *base, uint64_t n_groups, int g_log2, int rot, int slots, uint32_t *xcc_out) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    if (threadIdx.x == 0 && blockIdx.x < 64) xcc_out[blockIdx.x] = __builtin_amdgcn_s_getreg(20 | (3 << 11));
    const uint64_t G = 1ull << g_log2;
    const int sel = (xcd + rot) & 7;
    const v2d val = {1.0, 2.0};
    if (G >= 4096) {
        // one unit = G/4096 block-iterations
        const uint64_t per_unit = G >> 12;
        for (uint64_t grp = slot; grp < n_groups; grp += slots) {
            char *u = base + ((grp << 3) + sel) * G + threadIdx.x * 16;
            for (uint64_t s = 0; s < per_unit; ++s) __builtin_nontemporal_store(val, (v2d *)(u + (s << 12)));
        }
    } else {
        const uint64_t units_per_iter = 4096 >> g_log2;                 // units covered by one block iteration
        const uint64_t k = (threadIdx.x * 16) >> g_log2, off = (threadIdx.x * 16) & (G - 1);
        for (uint64_t g0 = (uint64_t)slot * units_per_iter; g0 + units_per_iter <= n_groups; g0 += (uint64_t)slots * units_per_iter)
            __builtin_nontemporal_store(val, (v2d *)(base + (((g0 + k) << 3) + sel) * G + off));
    }
}

__global__ __launch_bounds__(256) void all_kernel(char *base, uint64_t bytes, int slots8) {
    const v2d val = {1.0, 2.0};
    for (uint64_t o = (uint64_t)blockIdx.x * 4096 + threadIdx.x * 16; o + 16 <= bytes; o += (uint64_t)slots8 * 4096)
        __builtin_nontemporal_store(val, (v2d *)(base + o));
}

int main(int argc, char **argv) {
    const uint64_t bytes = (argc > 1 ? strtoull(argv[1], 0, 10) : 80ull) << 30;
    const int n_buf = argc > 2 ? atoi(argv[2]) : 2;
    const int slots = argc > 3 ? atoi(argv[3]) : 1024;   // blocks per XCD
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    uint32_t *xcc; CK(hipMalloc(&xcc, 64 * 4));
    std::vector<char *> bufs(n_buf);
    for (auto &b : bufs) CK(hipMalloc(&b, bytes));
    auto timed = [&](auto launch) {
        float best = 1e30f;
        for (int r = 0; r < 4; ++r) {
            CK(hipEventRecord(e0, st)); launch(); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r && ms < best) best = ms;
        }
        return best;
    };
    for (char *b : bufs) {
        float all = timed([&] { hipLaunchKernelGGL(all_kernel, dim3(8 * slots), dim3(256), 0, st, b, bytes, 8 * slots); });
        printf("buffer %p  %.0f GiB  uniform: %.2f ms (%.0f GB/s)\n", (void *)b, bytes / 1073741824.0, all, bytes / all * 1e-6);
        for (int gl : {8, 9, 10, 11, 12, 13, 14, 16, 18, 20, 21}) {
            const uint64_t G = 1ull << gl, n_groups = bytes / (8 * G);
            printf("  unit %7llu B:", (unsigned long long)G);
            for (int rot = 0; rot < 8; ++rot) {
                float ms = timed([&] { hipLaunchKernelGGL(own_kernel, dim3(8 * slots), dim3(256), 0, st, b, n_groups, gl, rot, slots, xcc); });
                printf(" %5.0f", (bytes / 8) / ms * 1e-6);
            }
            printf("   GB/s per rot 0..7\n"); fflush(stdout);
        }
    }
    uint32_t h[64]; CK(hipMemcpy(h, xcc, sizeof h, hipMemcpyDeviceToHost));
    printf("XCC_ID of blocks 0..15:"); for (int i = 0; i < 16; ++i) printf(" %u", h[i] & 15); printf("\n");
    return 0;
}
