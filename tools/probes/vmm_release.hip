// Dev probe (not product): does memory taken with hipMemCreate come back after hipMemUnmap + hipMemRelease?  Variants: mapped
// once; mapped at one address, unmapped, mapped at another (what spread.hip does); released while still mapped.
// build: hipcc -O2 --offload-arch=gfx950 -o tools/probes/vmm_release tools/probes/vmm_release.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "line %d: %s: %s\n", __LINE__, #x, hipGetErrorString(e_)); return 1; } } while (0)
static double free_gib() { size_t f, t; (void)hipMemGetInfo(&f, &t); return f / 1073741824.0; }
int main() {
    const size_t CH = size_t(1) << 30; const int N = 40;
    hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    printf("start: free %.1f GiB\n", free_gib());
    for (int variant = 0; variant < 5; ++variant) {
        std::vector<hipMemGenericAllocationHandle_t> h(N);
        char *va, *vb; CK(hipMemAddressReserve((void **)&va, N * CH, 0, nullptr, 0)); CK(hipMemAddressReserve((void **)&vb, N * CH, 0, nullptr, 0));
        for (int i = 0; i < N; ++i) { CK(hipMemCreate(&h[i], CH, &prop, 0)); CK(hipMemMap(va + i * CH, CH, 0, h[i], 0)); CK(hipMemSetAccess(va + i * CH, CH, &acc, 1)); }
        CK(hipMemset(va, 1, N * CH)); CK(hipDeviceSynchronize());
        printf("variant %d: %d GiB created and mapped: free %.1f GiB\n", variant, N, free_gib());
        if (variant == 0) { for (int i = 0; i < N; ++i) { CK(hipMemUnmap(va + i * CH, CH)); CK(hipMemRelease(h[i])); } }
        if (variant == 1) {
            for (int i = 0; i < N; ++i) { CK(hipMemUnmap(va + i * CH, CH)); CK(hipMemMap(vb + i * CH, CH, 0, h[i], 0)); }
            CK(hipMemSetAccess(vb, N * CH, &acc, 1)); CK(hipMemset(vb, 2, N * CH)); CK(hipDeviceSynchronize());
            for (int i = 0; i < N; ++i) { CK(hipMemUnmap(vb + i * CH, CH)); CK(hipMemRelease(h[i])); }
        }
        if (variant == 2) { for (int i = 0; i < N; ++i) { CK(hipMemRelease(h[i])); CK(hipMemUnmap(va + i * CH, CH)); } }
        if (variant == 3) { for (int i = 0; i < N; ++i) { CK(hipMemUnmap(va + i * CH, CH)); CK(hipMemRelease(h[i])); } CK(hipMemAddressFree(va, N * CH)); CK(hipMemAddressFree(vb, N * CH)); }
        if (variant == 4) { for (int i = 0; i < N; ++i) CK(hipMemUnmap(va + i * CH, CH)); CK(hipMemAddressFree(va, N * CH)); CK(hipMemAddressFree(vb, N * CH)); for (int i = 0; i < N; ++i) CK(hipMemRelease(h[i])); }
        CK(hipDeviceSynchronize());
        printf("           after unmap + release%s: free %.1f GiB\n", variant >= 3 ? " + address free" : "", free_gib());
    }
    void *p; hipError_t e = hipMalloc(&p, size_t(250) << 30);
    printf("hipMalloc(250 GiB) afterwards: %s; free %.1f GiB\n", hipGetErrorString(e), free_gib());
    return 0;
}
