// Dev probe (not product): is it safe to map a different physical chunk at an address that had another chunk before?
// X at p: fill 1.  Unmap; Y at p: fill 2.  Map X at q: must still read 1 (and Y at p 2).  Same again after freeing and
// re-reserving the range.  build: hipcc -O3 --offload-arch=gfx950 -o tools/probes/vmm_remap tools/probes/vmm_remap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "line %d: %s: %s\n", __LINE__, #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void fill(uint32_t *p, uint32_t v, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v; }
__global__ void count_ne(const uint32_t *p, uint32_t v, size_t n, unsigned long long *bad) {
    unsigned long long c = 0; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += p[i] != v;
    if (c) atomicAdd(bad, c);
}
int main() {
    const size_t CH = 1ull << 30, N = CH / 4;
    hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    hipMemGenericAllocationHandle_t X, Y; CK(hipMemCreate(&X, CH, &prop, 0)); CK(hipMemCreate(&Y, CH, &prop, 0));
    unsigned long long *bad; CK(hipMalloc(&bad, 8));
    auto check = [&](char *at, uint32_t want, const char *what) {
        CK(hipMemset(bad, 0, 8)); hipLaunchKernelGGL(count_ne, dim3(4096), dim3(256), 0, 0, (const uint32_t *)at, want, N, bad);
        unsigned long long h; CK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
        uint32_t w0, wm; CK(hipMemcpy(&w0, at, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&wm, at + CH / 2, 4, hipMemcpyDeviceToHost));
        printf("  %-40s %s (%llu words differ; first word %u, middle word %u)\n", what, h ? "WRONG" : "ok", h, w0, wm);
    };
    for (int variant = 0; variant < 4; ++variant) {
        const char *names[] = {"same reservation, unmap then map:", "free and re-reserve the range between the two mappings:",
                               "same reservation, access revoked (ProtNone) before the unmap:", "access revoked, unmap, free and re-reserve:"};
        printf("%s\n", names[variant]);
        hipMemAccessDesc none = acc; none.flags = hipMemAccessFlagsProtNone;
        char *p, *q; CK(hipMemAddressReserve((void **)&p, CH, 0, nullptr, 0)); CK(hipMemAddressReserve((void **)&q, CH, 0, nullptr, 0));
        CK(hipMemMap(p, CH, 0, X, 0)); CK(hipMemSetAccess(p, CH, &acc, 1));
        hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (uint32_t *)p, 1u, N); CK(hipDeviceSynchronize());
        if (variant >= 2) { hipError_t e = hipMemSetAccess(p, CH, &none, 1); printf("  hipMemSetAccess(ProtNone): %s\n", hipGetErrorString(e)); (void)hipGetLastError(); }
        CK(hipMemUnmap(p, CH));
        if (variant & 1) { CK(hipMemAddressFree(p, CH)); char *p2; CK(hipMemAddressReserve((void **)&p2, CH, 0, nullptr, 0)); printf("  range %p -> %p\n", (void *)p, (void *)p2); p = p2; }
        CK(hipMemMap(p, CH, 0, Y, 0)); CK(hipMemSetAccess(p, CH, &acc, 1));
        hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (uint32_t *)p, 2u, N); CK(hipDeviceSynchronize());
        CK(hipMemMap(q, CH, 0, X, 0)); CK(hipMemSetAccess(q, CH, &acc, 1));
        check(q, 1u, "X (mapped elsewhere) still holds 1");
        check(p, 2u, "Y at the reused address holds 2");
        CK(hipMemUnmap(p, CH)); CK(hipMemUnmap(q, CH)); CK(hipMemAddressFree(p, CH)); CK(hipMemAddressFree(q, CH));
    }
    return 0;
}
