// scan.h -- a multi-block exclusive scan of two small counters per element in ONE launch (chained scan with look-back), used for
// every ordered compaction of the window path (window.hip): kept aligned rows + pair offsets, kept triangles, the nodes that
// bring a same-type triangle back, rows of a section inside a box.
//
// Every block scans its NT elements locally and publishes its totals in one 8-byte word (2-bit state | 31-bit count a | 31-bit
// count p; the data IS the flag: an agent-scope atomic store, written through to memory, read by agent-scope atomic loads --
// the only inter-workgroup traffic, cdna guide "Guideline 16", form R2).  Block b then looks back: wave 0 reads the words of
// the 64 blocks before it at once and sums them down to the nearest block whose INCLUSIVE prefix is already known.  A
// predecessor that has not published yet is NOT waited for: its total is recomputed from the input by this block (the counters
// are pure functions of arrays earlier launches wrote), so there is no spin, no dependence on dispatch order or placement, and
// every wave reaches the end of the kernel whatever the others do.  All sums are integers: the result does not depend on which
// path supplied a term.  The words must be zero when the launch starts (one hipMemsetAsync per call zeroes them with the rest
// of the call's counters).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

namespace scan {

typedef __attribute__((address_space(1))) unsigned long long gu64;

constexpr int NT = 256;   // threads per block of every kernel built on the primitive
constexpr unsigned long long ST_AGG = 1ull << 62, ST_PFX = 2ull << 62;
constexpr unsigned VMASK = 0x7FFFFFFFu;
enum { MODE_DONE = 0, MODE_MORE = 1, MODE_RECOMPUTE = 2 };

struct Pair {
    unsigned a, p;
};

struct Shared {
    unsigned wa[NT / 64], wp[NT / 64];
    unsigned acc_a, acc_p;
    int mode, consumed;
};

__device__ __forceinline__ unsigned long long pack(unsigned long long st, Pair v) {
    return st | ((unsigned long long)(v.a & VMASK) << 31) | (unsigned long long)(v.p & VMASK);
}
__device__ __forceinline__ void publish(unsigned long long *status, int b, unsigned long long w) {
    __hip_atomic_store(((gu64 *)status) + b, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long peek(unsigned long long *status, int b) {
    return __hip_atomic_load(((gu64 *)status) + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// totals of one block's values, the same in every thread (two barriers)
__device__ __forceinline__ Pair block_total(Pair v, Shared &s) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned a = v.a, p = v.p;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_xor(a, off, 64);
        p += __shfl_xor(p, off, 64);
    }
    if (lane == 0) { s.wa[wave] = a; s.wp[wave] = p; }
    __syncthreads();
    Pair t{0, 0};
#pragma unroll
    for (int q = 0; q < NT / 64; ++q) { t.a += s.wa[q]; t.p += s.wp[q]; }
    __syncthreads();
    return t;
}

// Exclusive prefix (over ALL blocks of one scan) of this thread's element, element index = b * NT + threadIdx.x, b = the block's
// number within the scan (blockIdx.x unless a launch carries two scans); `val(i)` is the element's pair of counters and must
// return {0, 0} past the end of the input.  *through = inclusive total through this block (the grand total in the last block).
// Launch with NT threads per block; one status word per block, zeroed before the launch.
template <class ValFn>
__device__ __forceinline__ Pair exclusive(unsigned long long *status, int b, ValFn val, Shared &s, Pair *through) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // test hook (scan::arg on the host, SAME_SCAN_FORCE_RECOMPUTE=1): bit 0 of the (8-byte aligned) pointer makes every look-back treat its
    // predecessors as "not published", so the recompute path -- rare and timing-dependent otherwise -- carries the whole scan
    const bool force_recompute = (reinterpret_cast<uintptr_t>(status) & 1u) != 0;
    status = reinterpret_cast<unsigned long long *>(reinterpret_cast<uintptr_t>(status) & ~uintptr_t(7));
    const Pair v = val((int64_t)b * NT + tid);
    unsigned ia = v.a, ip = v.p;   // inclusive within the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned ua = __shfl_up(ia, off, 64), up = __shfl_up(ip, off, 64);
        if (lane >= off) { ia += ua; ip += up; }
    }
    if (lane == 63) { s.wa[wave] = ia; s.wp[wave] = ip; }
    __syncthreads();
    Pair local{ia - v.a, ip - v.p}, total{0, 0};
#pragma unroll
    for (int q = 0; q < NT / 64; ++q) {
        if (q < wave) { local.a += s.wa[q]; local.p += s.wp[q]; }
        total.a += s.wa[q];
        total.p += s.wp[q];
    }
    if (tid == 0) publish(status, b, pack(b == 0 ? ST_PFX : ST_AGG, total));
    Pair base{0, 0};
    int hi = b - 1;   // nearest predecessor not yet accounted for (uniform over the block)
    while (hi >= 0) {
        if (wave == 0) {
            const int q = hi - lane;
            unsigned long long w = q >= 0 ? peek(status, q) : ST_PFX;   // in front of block 0: a prefix of nothing
            if (force_recompute && q >= 0) w = 0ull;
            const unsigned st = (unsigned)(w >> 62);
            const unsigned long long pfx = __ballot(st == 2), not_ready = __ballot(st == 0);
            const int fp = pfx ? __builtin_ctzll(pfx) : 64, fn = not_ready ? __builtin_ctzll(not_ready) : 64;
            const int stop = fp < fn ? fp : fn;   // lanes [0, stop) hold block totals; lane `stop` a prefix (done) or nothing yet
            const bool take = lane < stop || (lane == stop && fp < fn);
            unsigned xa = take ? (unsigned)(w >> 31) & VMASK : 0u, xp = take ? (unsigned)w & VMASK : 0u;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                xa += __shfl_xor(xa, off, 64);
                xp += __shfl_xor(xp, off, 64);
            }
            if (lane == 0) {
                s.acc_a = xa;
                s.acc_p = xp;
                s.mode = fp < fn ? MODE_DONE : (fn < 64 ? MODE_RECOMPUTE : MODE_MORE);
                s.consumed = stop;
            }
        }
        __syncthreads();
        base.a += s.acc_a;
        base.p += s.acc_p;
        const int mode = s.mode;
        hi -= s.consumed;
        __syncthreads();
        if (mode == MODE_DONE) break;
        if (mode == MODE_RECOMPUTE) {   // block `hi` has published nothing yet: its totals from the input, no waiting
            const Pair r = block_total(val((int64_t)hi * NT + tid), s);
            base.a += r.a;
            base.p += r.p;
            hi -= 1;
        }
    }
    through->a = base.a + total.a;
    through->p = base.p + total.p;
    if (tid == 0 && b != 0) publish(status, b, pack(ST_PFX, *through));
    return Pair{base.a + local.a, base.p + local.p};
}

// host side: the status pointer as a kernel argument (tagged when SAME_SCAN_FORCE_RECOMPUTE is set in the environment: a test switch)
inline unsigned long long *arg(unsigned long long *status) {
    static const bool force = [] { const char *e = getenv("SAME_SCAN_FORCE_RECOMPUTE"); return e && e[0] && e[0] != '0'; }();
    return force ? reinterpret_cast<unsigned long long *>(reinterpret_cast<uintptr_t>(status) | 1u) : status;
}
__host__ __device__ inline unsigned blocks_for(int64_t n) { return (unsigned)((n > 0 ? n + NT - 1 : NT) / NT); }
inline size_t status_bytes(int64_t n) { return ((size_t)blocks_for(n) * 8 + 15) & ~size_t(15); }

}  // namespace scan
