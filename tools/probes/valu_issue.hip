// valu_issue.hip -- what the SIMD sustains on the dense cost kernel's inner-loop instruction forms (gfx950).
// The fp32 dense build (csrc/cost.hip, dense_cost_kernel<float,T,4>) runs at ~80 % of the 2-cycles-per-VALU floor
// at the clock it holds; this probe times each instruction FORM of that loop on its own, at 1 / 2 / 4 waves per SIMD:
//   vvv      v_sub_f32 v, v, v             (VOP2, registers only)
//   svv      v_sub_f32 v, s, v             (VOP2, one SGPR source: how the aligned row reaches the VALU)
//   abs      v_add_f32 v, v, |v|           (VOP3: the abs modifier forces the 64-bit encoding)
//   pair32   v_sub_f32 d, s, r ; v_add_f32 acc, acc, |d|      x4 columns (the real loop body)
//   pk       v_pk_add_f32
//   pair64   v_add_f64 d, s, -r ; v_add_f64 acc, acc, |d|     x2 columns (the fp64 loop body)
// Output: shader cycles (s_memtime) per instruction per SIMD, i.e. elapsed / (instructions issued by ONE wave * waves per SIMD).
// Build: hipcc --offload-arch=gfx950 -O2 -o valu_issue valu_issue.hip ; run: ./valu_issue
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
    } while (0)

constexpr int ITERS = 4096;   // loop trips; each trip issues 32 instructions of the form under test

template <int FORM>
__global__ __launch_bounds__(256) void probe(float *sink, unsigned long long *cycles, float sa, double sd) {
    float v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, r0 = 0.5f, r1 = 1.5f, r2 = 2.5f, r3 = 3.5f;
    float d0 = 0, d1 = 0, d2 = 0, d3 = 0;
    double a0 = threadIdx.x, a1 = a0 + 1, q0 = 0.5, q1 = 1.5, e0 = 0, e1 = 0;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {v0, v1}, p1 = {v2, v3}, p2 = {r0, r1}, p3 = {r2, r3};
    unsigned long long t0, t1;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < ITERS; ++it) {
        if constexpr (FORM == 0) {
            asm volatile(".rept 8\n\tv_sub_f32 %0, %4, %0\n\tv_sub_f32 %1, %5, %1\n\tv_sub_f32 %2, %6, %2\n\tv_sub_f32 %3, %7, %3\n\t.endr"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(r0), "v"(r1), "v"(r2), "v"(r3));
        } else if constexpr (FORM == 1) {
            asm volatile(".rept 8\n\tv_sub_f32 %0, %4, %0\n\tv_sub_f32 %1, %4, %1\n\tv_sub_f32 %2, %4, %2\n\tv_sub_f32 %3, %4, %3\n\t.endr"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "s"(sa));
        } else if constexpr (FORM == 2) {
            asm volatile(".rept 8\n\tv_add_f32 %0, %0, |%4|\n\tv_add_f32 %1, %1, |%5|\n\tv_add_f32 %2, %2, |%6|\n\tv_add_f32 %3, %3, |%7|\n\t.endr"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(r0), "v"(r1), "v"(r2), "v"(r3));
        } else if constexpr (FORM == 3) {
            asm volatile(".rept 4\n\t"
                         "v_sub_f32 %4, %12, %8\n\tv_sub_f32 %5, %12, %9\n\tv_sub_f32 %6, %12, %10\n\tv_sub_f32 %7, %12, %11\n\t"
                         "v_add_f32 %0, %0, |%4|\n\tv_add_f32 %1, %1, |%5|\n\tv_add_f32 %2, %2, |%6|\n\tv_add_f32 %3, %3, |%7|\n\t.endr"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3)
                         : "v"(r0), "v"(r1), "v"(r2), "v"(r3), "s"(sa));
        } else if constexpr (FORM == 4) {
            asm volatile(".rept 16\n\tv_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3\n\t.endr" : "+v"(p0), "+v"(p1) : "v"(p2), "v"(p3));
        } else if constexpr (FORM == 5) {
            asm volatile(".rept 8\n\t"
                         "v_add_f64 %2, %6, -%4\n\tv_add_f64 %3, %6, -%5\n\t"
                         "v_add_f64 %0, %0, |%2|\n\tv_add_f64 %1, %1, |%3|\n\t.endr"
                         : "+v"(a0), "+v"(a1), "+v"(e0), "+v"(e1) : "v"(q0), "v"(q1), "s"(sd));
        } else if constexpr (FORM == 6) {   // the fp32 pair with the subtrahend row in VGPRs instead of an SGPR
            asm volatile(".rept 4\n\t"
                         "v_sub_f32 %4, %12, %8\n\tv_sub_f32 %5, %12, %9\n\tv_sub_f32 %6, %12, %10\n\tv_sub_f32 %7, %12, %11\n\t"
                         "v_add_f32 %0, %0, |%4|\n\tv_add_f32 %1, %1, |%5|\n\tv_add_f32 %2, %2, |%6|\n\tv_add_f32 %3, %3, |%7|\n\t.endr"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3)
                         : "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(sa));
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    sink[blockIdx.x * 256 + threadIdx.x] = v0 + v1 + v2 + v3 + d0 + d1 + d2 + d3 + p0.x + p0.y + p1.x + p1.y + (float)(a0 + a1 + e0 + e1);
}

template <int FORM>
void run(const char *name, int instr_per_trip, float *sink, unsigned long long *cyc, int cus) {
    for (int wps : {1, 2, 4}) {            // waves per SIMD = blocks per CU (a block is 4 waves, one per SIMD)
        const int blocks = cus * wps;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(probe<FORM>, dim3(blocks), dim3(256), 0, 0, sink, cyc, 1.25f, 1.25);   // warm-up
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(probe<FORM>, dim3(blocks), dim3(256), 0, 0, sink, cyc, 1.25f, 1.25);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(blocks * 4);
        CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
        double mean = 0;
        for (auto c : h) mean += (double)c;
        mean /= h.size();
        const double n_instr = (double)ITERS * instr_per_trip;
        printf("%-7s waves/SIMD %d: %7.0f cycles/wave, %5.2f cycles per instruction per wave, %5.2f per instruction per SIMD  (%.3f ms)\n", name, wps,
               mean, mean / n_instr, mean / n_instr / wps, ms);
    }
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float *sink;
    unsigned long long *cyc;
    CK(hipMalloc(&sink, (size_t)cus * 4 * 256 * 4));
    CK(hipMalloc(&cyc, (size_t)cus * 4 * 4 * 8));
    printf("%s, %d CUs; s_memtime ticks per instruction (ticks: see the 100 MHz note below)\n", prop.gcnArchName, cus);
    run<0>("vvv", 32, sink, cyc, cus);
    run<1>("svv", 32, sink, cyc, cus);
    run<2>("abs", 32, sink, cyc, cus);
    run<3>("pair32", 32, sink, cyc, cus);
    run<6>("pair32v", 32, sink, cyc, cus);
    run<4>("pk", 32, sink, cyc, cus);
    run<5>("pair64", 32, sink, cyc, cus);
    printf("note: if the cycle figures look ~20x too small, s_memtime counts a fixed 100 MHz reference on this part; use the ms column "
           "(ms * sclk / instructions) instead\n");
    return 0;
}
