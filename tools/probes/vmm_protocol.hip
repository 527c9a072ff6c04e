// Dev probe (not product): a mapping protocol that (a) gives surplus memory back and (b) never writes through a stale
// translation.  Label in a scratch range S, map the chosen chunks into a final range F (variant 0: after unmapping them
// from S; variant 1: while still mapped in S), release the others, hipMemAddressFree(S).  Then: is the surplus memory back, is
// F intact, and what happens to F when plain hipMalloc buffers (possibly at S's old addresses) are written?
// build: hipcc -O2 --offload-arch=gfx950 -o tools/probes/vmm_protocol tools/probes/vmm_protocol.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "line %d: %s: %s\n", __LINE__, #x, hipGetErrorString(e_)); return 1; } } while (0)
static double free_gib() { size_t f, t; (void)hipMemGetInfo(&f, &t); return f / 1073741824.0; }
__global__ void fill(uint32_t *p, uint32_t v, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v; }
__global__ void count_ne(const uint32_t *p, uint32_t v, size_t n, unsigned long long *bad) {
    unsigned long long c = 0; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += p[i] != v;
    if (c) atomicAdd(bad, c);
}
int main() {
    const size_t CH = size_t(1) << 30, NW = CH / 4; const int N = 16;
    hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    unsigned long long *bad; CK(hipMalloc(&bad, 8));
    auto wrong = [&](char *at, uint32_t want) -> long long {
        if (hipMemset(bad, 0, 8) != hipSuccess) return -1;
        hipLaunchKernelGGL(count_ne, dim3(4096), dim3(256), 0, 0, (const uint32_t *)at, want, NW, bad);
        unsigned long long h = 0; if (hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost) != hipSuccess) return -1; return (long long)h;
    };
    for (int variant = 0; variant < 2; ++variant) {
        printf("variant %d (%s): free %.1f GiB at start\n", variant, variant ? "chosen chunks mapped in F while still mapped in S" : "chosen chunks unmapped from S first", free_gib());
        char *S, *F; CK(hipMemAddressReserve((void **)&S, N * CH, 0, nullptr, 0)); CK(hipMemAddressReserve((void **)&F, N / 2 * CH, 0, nullptr, 0));
        std::vector<hipMemGenericAllocationHandle_t> h(N);
        for (int i = 0; i < N; ++i) { CK(hipMemCreate(&h[i], CH, &prop, 0)); CK(hipMemMap(S + i * CH, CH, 0, h[i], 0)); }
        CK(hipMemSetAccess(S, N * CH, &acc, 1));
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (uint32_t *)(S + i * CH), (uint32_t)(100 + i), NW);
        CK(hipDeviceSynchronize());
        printf("  16 chunks in S=%p: free %.1f GiB\n", (void *)S, free_gib());
        for (int k = 0; k < N / 2; ++k) {                      // chosen: the even chunks
            if (variant == 0) CK(hipMemUnmap(S + 2 * k * CH, CH));
            hipError_t e = hipMemMap(F + k * CH, CH, 0, h[2 * k], 0);
            if (e != hipSuccess) { printf("  second mapping refused: %s\n", hipGetErrorString(e)); return 1; }
        }
        CK(hipMemSetAccess(F, N / 2 * CH, &acc, 1));
        for (int i = 0; i < N; ++i) {
            if (i % 2 || variant == 1) CK(hipMemUnmap(S + i * CH, CH));
            if (i % 2) CK(hipMemRelease(h[i]));
        }
        CK(hipMemAddressFree(S, N * CH));
        CK(hipDeviceSynchronize());
        printf("  surplus released, S freed: free %.1f GiB (8 GiB should be back)\n", free_gib());
        long long w = 0; for (int k = 0; k < N / 2; ++k) w += wrong(F + k * CH, (uint32_t)(100 + 2 * k));
        printf("  F holds the chosen chunks' values: %s (%lld words wrong)\n", w ? "WRONG" : "ok", w);
        // plain allocations afterwards (may land on S's old addresses), written in full
        std::vector<char *> pl(8); int on_S = 0;
        for (auto &p : pl) { CK(hipMalloc((void **)&p, CH)); if (p >= S && p < S + N * CH) ++on_S; hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (uint32_t *)p, 7u, NW); }
        CK(hipDeviceSynchronize());
        w = 0; for (int k = 0; k < N / 2; ++k) w += wrong(F + k * CH, (uint32_t)(100 + 2 * k));
        long long wp = 0; for (auto p : pl) wp += wrong(p, 7u);
        printf("  8 plain hipMalloc GiB written (%d of them inside S's old range): F %s (%lld wrong), plain buffers %s (%lld wrong)\n", on_S, w ? "CORRUPTED" : "intact", w, wp ? "WRONG" : "ok", wp);
        // a new reservation: does it come back at S's old address?
        char *S2; CK(hipMemAddressReserve((void **)&S2, N * CH, 0, nullptr, 0));
        printf("  a new reservation of the same size: %p (%s)\n", (void *)S2, S2 == S ? "S's old address" : "a fresh address");
        // with an address hint well above everything seen so far
        char *S3, *hint = (char *)(((uintptr_t)(S > F ? S : F) + (size_t(1) << 40)) & ~((uintptr_t)(1 << 21) - 1));
        hipError_t e3 = hipMemAddressReserve((void **)&S3, N * CH, 0, hint, 0);
        printf("  reservation with hint %p: %s -> %p\n", (void *)hint, hipGetErrorString(e3), e3 == hipSuccess ? (void *)S3 : nullptr);
        if (e3 == hipSuccess) CK(hipMemAddressFree(S3, N * CH));
        CK(hipMemAddressFree(S2, N * CH));
        for (auto p : pl) CK(hipFree(p));
        for (int k = 0; k < N / 2; ++k) { CK(hipMemUnmap(F + k * CH, CH)); CK(hipMemRelease(h[2 * k])); }
        CK(hipMemAddressFree(F, N / 2 * CH));
        CK(hipDeviceSynchronize());
        printf("  F unmapped, released, freed: free %.1f GiB\n", free_gib());
    }
    return 0;
}
