// Dev probe (not product): (1) split the card into 1 GiB physical chunks (hipMemCreate), find which chunks share a "slow
// together" region by timed pair stores, (2) store rate with the eight XCD regions on 1 / 2 / 3 such regions, (3) the dense
// build's store pattern into an 80 GB range mapped from one region vs mapped round-robin over the regions.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/probes/vmm_ranks tools/probes/vmm_ranks.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <string>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "line %d: %s: %s\n", __LINE__, #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));
struct Bases { char *p[8]; };

// XCD region r (= blockIdx % 8) writes n_rows x row_bytes at pitch row_bytes starting at b.p[r]
__global__ __launch_bounds__(256) void multi_kernel(Bases b, int n_rows, int n_tiles, uint64_t row_bytes, int rpb, int chunks) {
    const unsigned region = blockIdx.x & 7u, k = blockIdx.x >> 3;
    const int tile = (int)(k % n_tiles), chunk = (int)(k / n_tiles);
    if (chunk >= chunks) return;
    int r0 = chunk * rpb; if (r0 + rpb > n_rows) r0 = n_rows - rpb;
    const uint64_t col = (uint64_t)tile * 4096 + threadIdx.x * 16;
    if (col + 16 > row_bytes) return;
    char *p = b.p[region] + (uint64_t)r0 * row_bytes + col;
    const v2d val = {1.0, 2.0};
    for (int r = 0; r < rpb; ++r, p += row_bytes) __builtin_nontemporal_store(val, (v2d *)p);
}

// the same shape, reading: every lane sums its 16-byte loads (the sum is written once per block so the loads stay)
__global__ __launch_bounds__(256) void multi_read_kernel(Bases b, int n_rows, int n_tiles, uint64_t row_bytes, int rpb, int chunks, double *sink) {
    const unsigned region = blockIdx.x & 7u, k = blockIdx.x >> 3;
    const int tile = (int)(k % n_tiles), chunk = (int)(k / n_tiles);
    if (chunk >= chunks) return;
    int r0 = chunk * rpb; if (r0 + rpb > n_rows) r0 = n_rows - rpb;
    const uint64_t col = (uint64_t)tile * 4096 + threadIdx.x * 16;
    if (col + 16 > row_bytes) return;
    const char *p = b.p[region] + (uint64_t)r0 * row_bytes + col;
    v2d acc = {0.0, 0.0};
    for (int r = 0; r < rpb; ++r, p += row_bytes) acc += __builtin_nontemporal_load((const v2d *)p);
    if (acc.x + acc.y == 12345.678) sink[blockIdx.x & 1023] = acc.x;
}

// the dense build's pattern (map 2: one contiguous share of the (chunk, tile) space per XCD)
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

static hipEvent_t e0, e1;
template <typename L> static float best_ms(L launch, int reps = 4) {
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r && ms < best) best = ms;
    }
    CK(hipGetLastError());
    return best;
}

int main() {
    const uint64_t CH = 1ull << 30;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gmin = 0, grec = 0;
    CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
    CK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
    size_t fr, tot; CK(hipMemGetInfo(&fr, &tot));
    printf("granularity min %zu recommended %zu; free %.1f GB\n", gmin, grec, fr * 1e-9);
    std::vector<hipMemGenericAllocationHandle_t> h;
    while (true) {
        CK(hipMemGetInfo(&fr, &tot));
        if (fr < CH + (6ull << 30)) break;
        hipMemGenericAllocationHandle_t x;
        if (hipMemCreate(&x, CH, &prop, 0) != hipSuccess) { (void)hipGetLastError(); break; }
        h.push_back(x);
    }
    const int n = (int)h.size();
    char *va; CK(hipMemAddressReserve((void **)&va, n * CH, 0, nullptr, 0));
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    for (int i = 0; i < n; ++i) CK(hipMemMap(va + i * CH, CH, 0, h[i], 0));
    CK(hipMemSetAccess(va, n * CH, &acc, 1));
    printf("%d chunks of 1 GiB mapped at %p\n", n, (void *)va);
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

    // pair timing: XCD regions 0..3 share chunk a (a quarter each), 4..7 share chunk b
    const uint64_t row_bytes = 1 << 18;                       // 256 KB rows, 1024 rows per quarter-chunk
    const int q_rows = (int)(CH / 4 / row_bytes), n_tiles = (int)(row_bytes / 4096), rpb = 64, chunks = q_rows / rpb;
    auto pair_rate = [&](char *a, char *b) {
        Bases B; for (int r = 0; r < 8; ++r) B.p[r] = (r < 4 ? a : b) + (uint64_t)(r & 3) * (CH / 4);
        float ms = best_ms([&] { hipLaunchKernelGGL(multi_kernel, dim3(8 * n_tiles * chunks), dim3(256), 0, 0, B, q_rows, n_tiles, row_bytes, rpb, chunks); });
        return 2.0 * CH / ms * 1e-6;
    };
    // classify: label[i] in {0,1,2,...}; a chunk is "fast with" a reference when the pair rate is nearer the fast level
    std::vector<int> label(n, -1); std::vector<int> refs;
    std::vector<double> lo_hi;
    for (int i = 0; i < n; ++i) {
        int found = -1; std::string s;
        for (size_t r = 0; r < refs.size(); ++r) {
            double g = pair_rate(va + (uint64_t)refs[r] * CH, va + (uint64_t)i * CH);
            lo_hi.push_back(g);
            if (g < 6350.0 && found < 0) found = (int)r;       // threshold refined from the printed histogram if needed
        }
        if (found < 0 && refs.size() < 6) { refs.push_back(i); found = (int)refs.size() - 1; }
        label[i] = found;
    }
    printf("regions found: %zu; chunk labels in allocation order:\n", refs.size());
    for (int i = 0; i < n; ++i) printf("%c", label[i] < 0 ? '?' : 'A' + label[i]); printf("\n");
    // histogram of the pair rates seen (to check the threshold)
    int hist[16] = {0}; for (double g : lo_hi) { int b = (int)((g - 4000) / 250); if (b < 0) b = 0; if (b > 15) b = 15; hist[b]++; }
    printf("pair-rate histogram (250 GB/s bins from 4000):"); for (int b = 0; b < 16; ++b) printf(" %d", hist[b]); printf("\n");
    std::vector<std::vector<int>> by(refs.size());
    for (int i = 0; i < n; ++i) if (label[i] >= 0) by[label[i]].push_back(i);
    for (size_t r = 0; r < by.size(); ++r) printf("region %c: %zu chunks\n", (char)('A' + r), by[r].size());

    // (2) eight XCD regions, each writing one whole chunk, chunks drawn from 1, 2, 3 regions
    if (by.size() >= 3 && by[0].size() >= 8 && by[1].size() >= 8 && by[2].size() >= 8) {
        const int w_rows = (int)(CH / row_bytes), w_chunks = w_rows / rpb;
        auto spread = [&](std::vector<int> pick, const char *what) {
            Bases B; for (int r = 0; r < 8; ++r) B.p[r] = va + (uint64_t)pick[r] * CH;
            float ms = best_ms([&] { hipLaunchKernelGGL(multi_kernel, dim3(8 * n_tiles * w_chunks), dim3(256), 0, 0, B, w_rows, n_tiles, row_bytes, rpb, w_chunks); }, 6);
            printf("  8 GiB over %-28s %.3f ms  %.0f GB/s\n", what, ms, 8.0 * CH / ms * 1e-6);
        };
        double *sink; CK(hipMalloc(&sink, 1024 * 8));
        auto rspread = [&](std::vector<int> pick, const char *what) {
            Bases B; for (int r = 0; r < 8; ++r) B.p[r] = va + (uint64_t)pick[r] * CH;
            float ms = best_ms([&] { hipLaunchKernelGGL(multi_read_kernel, dim3(8 * n_tiles * w_chunks), dim3(256), 0, 0, B, w_rows, n_tiles, row_bytes, rpb, w_chunks, sink); }, 6);
            printf("  8 GiB READ over %-23s %.3f ms  %.0f GB/s\n", what, ms, 8.0 * CH / ms * 1e-6);
        };
        auto &A = by[0], &Bq = by[1], &C = by[2];
        rspread({A[0], A[1], A[2], A[3], A[4], A[5], A[6], A[7]}, "one region (A)");
        rspread({C[0], C[1], C[2], C[3], C[4], C[5], C[6], C[7]}, "one region (C)");
        rspread({A[0], Bq[0], A[1], Bq[1], A[2], Bq[2], A[3], Bq[3]}, "two regions (A,B) 4+4");
        rspread({A[0], Bq[0], C[0], A[1], Bq[1], C[1], A[2], Bq[2]}, "three regions 3+3+2");
        spread({A[0], A[1], A[2], A[3], A[4], A[5], A[6], A[7]}, "one region (A)");
        spread({Bq[0], Bq[1], Bq[2], Bq[3], Bq[4], Bq[5], Bq[6], Bq[7]}, "one region (B)");
        spread({C[0], C[1], C[2], C[3], C[4], C[5], C[6], C[7]}, "one region (C)");
        spread({A[0], Bq[0], A[1], Bq[1], A[2], Bq[2], A[3], Bq[3]}, "two regions (A,B) 4+4");
        spread({A[0], C[0], A[1], C[1], A[2], C[2], A[3], C[3]}, "two regions (A,C) 4+4");
        spread({A[0], Bq[0], C[0], A[1], Bq[1], C[1], A[2], Bq[2]}, "three regions 3+3+2");
        spread({A[0], A[1], A[2], A[3], A[4], A[5], Bq[0], Bq[1]}, "two regions (A,B) 6+2");
        spread({A[0], A[1], A[2], A[3], A[4], A[5], A[6], Bq[1]}, "two regions (A,B) 7+1");
    }

    // (3) the 100000 x 100000 fp64 matrix (80 GB) through a range mapped from one region vs round-robin over regions
    const uint64_t need = (100000ull * 800000ull + CH - 1) / CH;   // 75 chunks
    auto run80 = [&](std::vector<int> order, const char *what, size_t align = 0) {
        if (order.size() < need) { printf("  %s: not enough chunks (%zu)\n", what, order.size()); return; }
        char *v2; CK(hipMemAddressReserve((void **)&v2, need * CH, align, nullptr, 0));
        printf("  [range at %p] ", (void *)v2);
        for (uint64_t i = 0; i < need; ++i) { CK(hipMemUnmap(va + (uint64_t)order[i] * CH, CH)); CK(hipMemMap(v2 + i * CH, CH, 0, h[order[i]], 0)); }
        CK(hipMemSetAccess(v2, need * CH, &acc, 1));
        const int rows = 100000, rpb2 = 256, ch2 = (rows + rpb2 - 1) / rpb2, nt2 = (800000 + 4095) / 4096;
        const unsigned R = 8, per = (unsigned)(((uint64_t)ch2 * nt2 + R - 1) / R);
        float ms = best_ms([&] { hipLaunchKernelGGL(pattern_kernel, dim3(per * R), dim3(256), 0, 0, v2, 800000ull, rows, nt2, 800000ull, rpb2, ch2, R, per); }, 5);
        const unsigned per1 = (unsigned)((uint64_t)ch2 * nt2);
        float lin = best_ms([&] { hipLaunchKernelGGL(pattern_kernel, dim3(per1), dim3(256), 0, 0, v2, 800000ull, rows, nt2, 800000ull, rpb2, ch2, 1u, per1); }, 4);
        printf("  80 GB matrix, %-34s dense pattern %.2f ms (%.0f GB/s)   single front %.2f ms\n", what, ms, 80e9 / ms * 1e-6, lin);
        for (uint64_t i = 0; i < need; ++i) { CK(hipMemUnmap(v2 + i * CH, CH)); CK(hipMemMap(va + (uint64_t)order[i] * CH, CH, 0, h[order[i]], 0)); }
        CK(hipMemSetAccess(va, n * CH, &acc, 1));
        // the range is left reserved on purpose: a freed range is handed out again and the next mapping did not take effect there
    };
    {
        std::vector<int> nat; for (int i = 0; i < n; ++i) nat.push_back(i);
        run80(nat, "allocation order, align 0");
    }
    if (by.size() >= 3) {
        run80(by[0], "one region (A)");
        run80(by[1], "one region (B)");
        std::vector<int> rr; for (size_t i = 0; rr.size() < need && i < 200; ++i) for (size_t r = 0; r < 3; ++r) if (i < by[r].size()) rr.push_back(by[r][i]);
        run80(rr, "round-robin A,B,C per GiB");
        std::vector<int> rr2; for (size_t i = 0; rr2.size() < need && i < 200; ++i) for (size_t r = 0; r < 2; ++r) if (i < by[r].size()) rr2.push_back(by[r][i]);
        run80(rr2, "round-robin A,B per GiB");
        std::vector<int> blk; for (size_t r = 0; r < 3; ++r) for (size_t i = 0; i < 25 && i < by[r].size(); ++i) blk.push_back(by[r][i]);
        run80(blk, "25 GiB of A, then B, then C");
        std::vector<int> nat; for (int i = 0; i < n; ++i) nat.push_back(i);
        run80(nat, "allocation order (as hipMalloc would)");
    }
    return 0;
}
