// Dev probe (not product): after hipFree of large buffers, how long until the memory can be allocated again?
// build: hipcc -O2 --offload-arch=gfx950 -o tools/probes/free_latency tools/probes/free_latency.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "line %d: %s: %s\n", __LINE__, #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    const size_t S = size_t(80) << 30;
    size_t fr, tot;
    for (int round = 0; round < 3; ++round) {
        std::vector<void *> b(3);
        for (auto &p : b) CK(hipMalloc(&p, S));
        for (auto &p : b) CK(hipMemset(p, 1, S));
        CK(hipDeviceSynchronize());
        CK(hipMemGetInfo(&fr, &tot));
        printf("round %d: 3 x 80 GiB live, free %.1f GiB\n", round, fr / 1073741824.0);
        const auto t0 = std::chrono::steady_clock::now();
        for (auto &p : b) CK(hipFree(p));
        auto ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
        printf("  hipFree x3 returned after %.1f ms\n", ms());
        void *q = nullptr;
        for (int i = 0; i < 400; ++i) {
            CK(hipMemGetInfo(&fr, &tot));
            hipError_t e = hipMalloc(&q, 3 * S);          // all of it at once
            if (e == hipSuccess) { printf("  240 GiB allocatable again after %.1f ms (free reads %.1f GiB)\n", ms(), fr / 1073741824.0); break; }
            (void)hipGetLastError();
            if (i % 20 == 0) printf("  t=%.1f ms: free reads %.1f GiB, hipMalloc(240 GiB): %s\n", ms(), fr / 1073741824.0, hipGetErrorString(e));
            std::this_thread::sleep_for(std::chrono::milliseconds(10));
        }
        if (!q) { printf("  never came back within 4 s\n"); return 2; }
        CK(hipFree(q));
        std::this_thread::sleep_for(std::chrono::milliseconds(500));
    }
    return 0;
}
