// spread.hip -- HBM-region-aware allocation for the large streaming outputs of libsame_hip (the dense cost matrix).
//
// Measured on MI355X (profiles/archive/r02_hbm_regions.md): the card's 288 GiB are three physical regions of 96 GiB, and a
// streaming store confined to ONE region runs at ~5.5-5.8 TB/s while the same store spread over two or three regions runs
// at ~7.0 TB/s.  hipMalloc hands an 80 GB buffer out of whatever regions its free lists hold, so the store time of the
// 100k x 100k build moved between 11.4 and 14.4 ms from one allocation to the next (what round 1 read as a per-box
// "store ceiling").  The regions are not visible in any address; they only show in timing.
//
// same_dev_alloc_spread therefore builds the buffer itself through the virtual-memory API: physical memory is taken in
// 1 GiB chunks (hipMemCreate); every chunk is labelled by timing a store into it TOGETHER with a store into a reference
// chunk of each region found so far (same region: slow level, different region: fast level, ~20 % apart); and the chunks
// are mapped into one contiguous address range round-robin over the regions, so ANY access pattern -- the dense kernel's
// eight XCD fronts, a single linear front, a copy -- is spread over the regions at all times.  Surplus chunks go back to
// the driver.
//
// Two behaviours of this ROCm's virtual-memory calls shape the code (tools/probes/vmm_{remap,release,protocol}.hip,
// profiles/archive/r02_hbm_vmm_*.log):
//  * memory taken with hipMemCreate only returns to the card when the ADDRESS RANGE it was mapped in is handed back with
//    hipMemAddressFree -- hipMemUnmap + hipMemRelease alone leave it charged;
//  * an address that carried chunk X keeps reaching X after hipMemUnmap + hipMemMap of another chunk there for as long as
//    X is alive somewhere, silently -- even across hipMemAddressFree and a new reservation that lands on the same address.
//    (A plain hipMalloc landing on such an address is fine: measured.)
// So: labelling happens in a scratch range whose slots are used once and which is freed when the call returns (the surplus
// chunks' memory goes back with it); the final range is freed when the buffer is; and every range is reserved at an address
// this process has never used for a mapping before -- a process-wide cursor walks up a 48 TiB stretch of the address space
// as the hint, and a reservation that comes back anywhere on used ground is parked, never mapped.
#include <algorithm>
#include <atomic>
#include <mutex>
#include <chrono>
#include <thread>
#include <cstdlib>

#include "common.h"
#include "spread_plan.h"

namespace {

constexpr size_t CHUNK = size_t(1) << 30;
constexpr int MAX_REGIONS = 3;
constexpr int MIXED = MAX_REGIONS;       // label of a chunk whose own two halves already run at the fast level (it straddles)
constexpr size_t MIN_CHUNKS = 6;         // below this a plain allocation: nothing to spread
constexpr double LEVEL_RATIO = 1.12;     // a rate above this multiple of the same-region level is the fast level (measured: same
                                         // region 0.95-1.08x of the level, other region 1.18-1.27x: profiles/archive/r02_spread_levels.log)
constexpr size_t MAX_SHARE_PERMILLE = 500;   // chosen chunks: no region above half (4+4 over two regions runs within 2 % of 3+3+2)
constexpr uintptr_t VA_FIRST = uintptr_t(0x100000000000);   // 16 TiB: far below where mmap / hipMalloc hand out addresses
constexpr uintptr_t VA_LAST = uintptr_t(0x400000000000);    // 64 TiB: 48 TiB of never-reused addresses for this process
constexpr size_t EXTRA_CHUNKS = 128;     // how far past the buffer's own chunks to look for balance: the driver hands chunks out in
                                         // runs of up to a whole region (96), and a chunk costs ~10 ms to take and label

typedef double v2d __attribute__((ext_vector_type(2)));

// XCD share r (= blockIdx % 8, the hardware's round-robin) writes a quarter of the `span` bytes at a (r < 4) or at b
// (r >= 4): 256 KB rows, 64 rows per block, the same 4 KB-per-block store the dense kernels issue.
constexpr uint64_t P_ROW = 1 << 18;
constexpr int P_TILES = (int)(P_ROW / 4096), P_RPB = 64;
__global__ __launch_bounds__(256) void spread_pair_kernel(char *a, char *b, uint64_t span, unsigned blocks_per_pass) {
    const unsigned share = blockIdx.x & 7u, k = (blockIdx.x >> 3) % blocks_per_pass;   // the grid holds several passes, one after the other
    const unsigned tile = k % P_TILES, chunk = k / P_TILES;              // blocks_per_pass = P_TILES * (span / 4 / P_ROW / P_RPB)
    char *p = (share < 4 ? a : b) + (uint64_t)(share & 3u) * (span / 4) + (uint64_t)chunk * P_RPB * P_ROW + (uint64_t)tile * 4096 + threadIdx.x * 16;
    const v2d val = {0.0, 0.0};
    for (int r = 0; r < P_RPB; ++r, p += P_ROW) __builtin_nontemporal_store(val, (v2d *)p);
}

// One store over the whole finished range, shaped like the stores the buffer was built for: at step p share r (= blockIdx % 8,
// one XCD) writes its own eighth of chunk (p + r) mod n, so at any moment the eight fronts lie in eight CONSECUTIVE chunks --
// with the chunks laid round-robin over the regions, in all of them at once -- every share has the same amount of work, and
// after n steps every byte has been written once.  Its rate is what same_dev_alloc_spread reports as verified.
__global__ __launch_bounds__(256) void spread_sweep_kernel(char *va, unsigned n_chunks) {
    constexpr unsigned PER_SLICE = (unsigned)(CHUNK / 8 / (P_RPB * 4096));   // blocks per eighth of a chunk: 64 tiles x 8 row groups
    const unsigned share = blockIdx.x & 7u, k = blockIdx.x >> 3;
    const unsigned step = k / PER_SLICE, within = k % PER_SLICE;
    const unsigned chunk = (step + share) % n_chunks;
    const unsigned tile = within % P_TILES, rows = within / P_TILES;
    char *p = va + (uint64_t)chunk * CHUNK + (uint64_t)share * (CHUNK / 8) + (uint64_t)rows * P_RPB * P_ROW + (uint64_t)tile * 4096 + threadIdx.x * 16;
    const v2d val = {0.0, 0.0};
    for (int r = 0; r < P_RPB; ++r, p += P_ROW) __builtin_nontemporal_store(val, (v2d *)p);
}

struct Timer {
    same_ctx *ctx;
    int rc = SAME_OK;
    // events of its own: the context's ev0 / ev1 belong to same_timer_start / stop, which a caller may have open around
    // an allocation (a caller may well allocate its output block inside a region it is timing)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    explicit Timer(same_ctx *c) : ctx(c) {
        if (hipEventCreate(&ev0) != hipSuccess || hipEventCreate(&ev1) != hipSuccess) rc = SAME_EIO;
    }
    ~Timer() {
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
    }
    Timer(const Timer &) = delete;
    Timer &operator=(const Timer &) = delete;
    // GB/s of one store over the whole finished range (spread_sweep_kernel); best of two after one untimed
    double sweep(char *va, size_t n_chunks) {
        const unsigned grid = 8u * (unsigned)n_chunks * (unsigned)(CHUNK / 8 / (P_RPB * 4096));
        const int passes = (int)std::max<size_t>(1, (24 + n_chunks - 1) / n_chunks);   // >= 24 GiB per reading: a short launch reads low
        float best = 1e30f;
        for (int r = 0; r < 3; ++r) {
            if (hipEventRecord(ev0, ctx->stream) != hipSuccess) { rc = SAME_EIO; return 0.0; }
            for (int q = 0; q < passes; ++q)
                hipLaunchKernelGGL(spread_sweep_kernel, dim3(grid), dim3(256), 0, ctx->stream, va, (unsigned)n_chunks);
            float ms = 0.f;
            if (hipEventRecord(ev1, ctx->stream) != hipSuccess || hipEventSynchronize(ev1) != hipSuccess ||
                hipEventElapsedTime(&ms, ev0, ev1) != hipSuccess) { rc = SAME_EIO; return 0.0; }
            if (r && ms < best) best = ms;
        }
        if (hipGetLastError() != hipSuccess) { rc = SAME_EIO; return 0.0; }
        return (double)passes * (double)n_chunks * (double)CHUNK / best * 1e-6;
    }
    // GB/s of writing `span` bytes at a and `span` bytes at b at once, 4 GiB in all per launch.  `warm`: one untimed launch
    // first (the first store into a freshly mapped chunk pays for its page tables); then `timed` launches, best of them
    // (the two levels are ~20 % apart, the readings within ~5 %; noise only ever lowers a rate, and is_fast() reads again
    // when a rate lands between the levels)
    double rate(char *a, char *b, uint64_t span, bool warm = true, int timed = 1) {
        const unsigned passes = (unsigned)(2 * CHUNK / span);
        const unsigned per_pass = (unsigned)P_TILES * (unsigned)(span / 4 / P_ROW / P_RPB);
        const unsigned grid = 8u * per_pass * passes;
        float best = 1e30f;
        for (int r = warm ? -1 : 0; r < timed; ++r) {
            if (hipEventRecord(ev0, ctx->stream) != hipSuccess) { rc = SAME_EIO; return 0.0; }
            hipLaunchKernelGGL(spread_pair_kernel, dim3(grid), dim3(256), 0, ctx->stream, a, b, span, per_pass);
            float ms = 0.f;
            if (hipEventRecord(ev1, ctx->stream) != hipSuccess || hipEventSynchronize(ev1) != hipSuccess ||
                hipEventElapsedTime(&ms, ev0, ev1) != hipSuccess) { rc = SAME_EIO; return 0.0; }
            if (r >= 0 && ms < best) best = ms;
        }
        if (hipGetLastError() != hipSuccess) { rc = SAME_EIO; return 0.0; }
        return 4.0 * (double)CHUNK / best * 1e-6;
    }
    double halves(char *a) { return rate(a, a + CHUNK / 2, CHUNK / 2, true, 1); }   // a new chunk against itself: the same-region level
    double pair(char *a, char *b) { return rate(a, b, CHUNK, false, 1); }           // both chunks have been written by their halves() already
};

struct Chunk {
    hipMemGenericAllocationHandle_t h;
    char *at;          // scratch address while mapped there, else nullptr
    double halves;     // first half against second half
    int label;         // -1 until labelled
};

}  // namespace

// A range at addresses this process has not mapped anything at before (header comment).  nullptr when there is none.
static char *reserve_fresh(size_t bytes) {
    static std::mutex mu;
    static uintptr_t cursor = VA_FIRST;
    static std::vector<std::pair<uintptr_t, uintptr_t>> used;      // every range ever handed out here, [begin, end)
    std::lock_guard<std::mutex> lock(mu);
    for (int attempt = 0; attempt < 4; ++attempt) {
        if (cursor + bytes + CHUNK > VA_LAST) return nullptr;
        void *hint = (void *)cursor, *p = nullptr;
        cursor += bytes + CHUNK;                                    // a gap of one chunk between ranges
        if (hipMemAddressReserve(&p, bytes, size_t(2) << 20, hint, 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        const uintptr_t b = (uintptr_t)p, e = b + bytes;
        bool fresh = true;
        for (auto &u : used) fresh = fresh && (e <= u.first || b >= u.second);
        if (fresh) { used.emplace_back(b, e); return (char *)p; }
        // the hint was not honoured and the range lies on used ground: keep it reserved (parked) so it is not offered again
    }
    return nullptr;
}

void same_spread_release(same_spread_alloc &a) {
    for (size_t i = 0; i < a.handles.size(); ++i) {
        (void)hipMemUnmap(a.va + i * CHUNK, CHUNK);
        (void)hipMemRelease((hipMemGenericAllocationHandle_t)a.handles[i]);
    }
    a.handles.clear();
    if (a.va) (void)hipMemAddressFree(a.va, a.bytes);   // this is what returns the memory to the card (header comment)
    a.va = nullptr;
}

static double env_seconds(const char *name, double dflt) {
    const char *v = getenv(name);
    if (!v || !*v) return dflt;
    char *end = nullptr;
    const double x = strtod(v, &end);
    return (end != v && x > 0.0) ? x : dflt;
}

// The virtual-memory path proper.  Leaves nothing behind on failure (every chunk given back).
static int spread_build(same_ctx *ctx, size_t n_need, size_t budget, void **out_dptr, int64_t *info) {
    const auto t0 = std::chrono::steady_clock::now();
    auto elapsed = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    // Wall-time bound (SAME_SPREAD_MAX_SECONDS, default 3): past it the search for better-balanced chunks stops and the
    // best choice so far is mapped; if even taking and labelling the buffer's own chunks runs past twice the bound the call
    // gives everything back and the caller gets the plain allocation (hipMemCreate itself is most of the time: ~25 ms / GiB).
    const double limit_s = env_seconds("SAME_SPREAD_MAX_SECONDS", 3.0);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));

    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = ctx->device;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;

    // most chunks this call can examine: the look-ahead (SAME_SPREAD_MAX_EXTRA_GIB overrides the 128) never takes more than
    // half of what the card has free beyond the buffer itself, so ranks sharing a card leave each other room while labelling
    size_t extra = EXTRA_CHUNKS;
    if (const char *v = getenv("SAME_SPREAD_MAX_EXTRA_GIB")) extra = (size_t)std::max(0L, atol(v));
    extra = std::min(extra, (budget - n_need) / 2);
    const size_t slots = n_need + extra;
    char *scratch = reserve_fresh(slots * CHUNK);
    if (!scratch) {
        ctx->err = "no unused address range left for a spread buffer (addresses are never reused for mappings); plain allocation";
        return SAME_ENOMEM;
    }
    std::vector<Chunk> ch;
    std::vector<int> refs;                                          // reference chunk of each region found so far
    std::vector<std::vector<int>> by_class(MAX_REGIONS + 1);
    std::vector<size_t> take(MAX_REGIONS + 1, 0);
    Timer tm(ctx);
    if (tm.rc != SAME_OK) {
        (void)hipMemAddressFree(scratch, slots * CHUNK);
        ctx->err = "hipEventCreate (labelling timer) failed; plain allocation";
        return SAME_EIO;
    }
    double slow = 0.0;                                              // same-region level (settle_level)
    auto give_back = [&]() {                                        // everything still held by the labelling, and its range
        for (auto &c : ch) {
            if (c.at) (void)hipMemUnmap(c.at, CHUNK);
            (void)hipMemRelease(c.h);
        }
        ch.clear();
        (void)hipMemAddressFree(scratch, slots * CHUNK);
    };
    double t_create = 0.0, t_time = 0.0;                            // seconds in hipMemCreate/Map/SetAccess and in timed stores (debug)
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    auto take_chunk = [&]() -> int {                                // 1 = got one, 0 = the driver has no more, < 0 = error
        Chunk c{};
        const double tc0 = now();
        hipError_t e = hipMemCreate(&c.h, CHUNK, &prop, 0);
        if (e != hipSuccess) { (void)hipGetLastError(); return 0; }
        c.at = scratch + ch.size() * CHUNK;                         // every scratch slot is used once
        e = hipMemMap(c.at, CHUNK, 0, c.h, 0);
        if (e == hipSuccess) e = hipMemSetAccess(c.at, CHUNK, &acc, 1);
        if (e != hipSuccess) { (void)hipMemRelease(c.h); return same_fail(ctx, SAME_EIO, "hipMemMap/hipMemSetAccess (labelling a chunk)", e); }
        const double tc1 = now();
        c.halves = tm.halves(c.at);
        t_create += tc1 - tc0;
        t_time += now() - tc1;
        c.label = -1;
        if (tm.rc != SAME_OK) { (void)hipMemUnmap(c.at, CHUNK); (void)hipMemRelease(c.h); ctx->err = "timing a labelling store failed"; return tm.rc; }
        ch.push_back(c);
        return 1;
    };
    auto settle_level = [&]() {                                     // lower quartile of the halves rates: inside the same-region cluster
        std::vector<double> v;
        for (auto &c : ch) v.push_back(c.halves);
        std::sort(v.begin(), v.end());
        slow = v[v.size() / 4];
    };
    // timing noise only ever lowers a rate, so a reading under the threshold but well above the level is taken again
    static const bool debug = getenv("SAME_SPREAD_DEBUG") != nullptr;   // every rate read while labelling, on stderr
    auto is_fast = [&](double g, char *a, char *b, uint64_t span) {
        const double fast_above = slow * LEVEL_RATIO;
        const double first = g;
        if (g < fast_above && g > slow * 1.04) g = std::max(g, tm.rate(a, b, span, false, 2));   // read again, best of two
        if (debug) fprintf(stderr, "[spread] %s rate %.0f%s vs level %.0f -> %s\n", span == CHUNK ? "pair" : "halves", first,
                           g != first ? (" (again: " + std::to_string((int)g) + ")").c_str() : "", slow, g >= fast_above ? "fast" : "slow");
        return g >= fast_above;
    };
    auto label_from = [&](size_t first) -> int {                    // label chunks [first, size) against the references
        for (size_t i = first; i < ch.size(); ++i) {
            Chunk &c = ch[i];
            if (is_fast(c.halves, c.at, c.at + CHUNK / 2, CHUNK / 2)) c.label = MIXED;
            else {
                c.label = -1;
                for (size_t r = 0; r < refs.size() && c.label < 0; ++r) {
                    const double tp0 = now();
                    const double g = tm.pair(ch[refs[r]].at, c.at);
                    t_time += now() - tp0;
                    if (tm.rc != SAME_OK) { ctx->err = "timing a labelling store failed"; return tm.rc; }
                    if (!is_fast(g, ch[refs[r]].at, c.at, CHUNK)) c.label = (int)r;
                }
                if (c.label < 0) {
                    if ((int)refs.size() < MAX_REGIONS) { refs.push_back((int)i); c.label = (int)refs.size() - 1; }
                    else c.label = MIXED;                           // fast with all three references: treat as a straddler
                }
            }
            by_class[c.label].push_back((int)i);
        }
        return SAME_OK;
    };
    auto choose = [&]() {                                           // the most balanced n_need chunks of those labelled
        std::vector<size_t> have;
        for (auto &v : by_class) have.push_back(v.size());
        take = spread_plan::water_fill(have, n_need);
        size_t got = 0;
        for (size_t t : take) got += t;
        return got;
    };
    auto lopsided = [&]() { return spread_plan::lopsided(take, MAX_REGIONS, n_need, MAX_SHARE_PERMILLE); };

    // 1. the chunks the buffer needs (the same-region level settles over these: most chunks lie inside one region);
    // 2. label them; 3. while the best choice is lop-sided, take and label more, within the look-ahead and the card's memory
    int got = 1;
    while (got == 1 && ch.size() < n_need && elapsed() < 2.0 * limit_s) got = take_chunk();
    if (got < 0) { give_back(); return got; }
    if (got == 1 && ch.size() < n_need) {
        give_back();
        ctx->err = "taking the chunks of a spread buffer ran past twice SAME_SPREAD_MAX_SECONDS; plain allocation";
        return SAME_EIO;
    }
    if (ch.size() < n_need) { give_back(); return same_fail(ctx, SAME_ENOMEM, "hipMemCreate (1 GiB chunks for a spread buffer)", hipErrorOutOfMemory); }
    settle_level();
    int rc = label_from(0);
    if (rc != SAME_OK) { give_back(); return rc; }
    bool timed_out = false;
    while (choose() == n_need && lopsided() && ch.size() < slots) {
        if (elapsed() > limit_s) { timed_out = true; break; }   // keep what there is: the verification below says what it is worth
        got = take_chunk();
        if (got < 0) { give_back(); return got; }
        if (got == 0) break;
        rc = label_from(ch.size() - 1);
        if (rc != SAME_OK) { give_back(); return rc; }
    }
    choose();

    // the final range, chunks laid round-robin over the regions
    std::vector<int> order = spread_plan::interleave(by_class, take);
    char *va = reserve_fresh(n_need * CHUNK);
    if (!va) { give_back(); ctx->err = "no unused address range left for a spread buffer; plain allocation"; return SAME_ENOMEM; }
    hipError_t e = hipSuccess;
    same_spread_alloc out;
    out.va = va;
    out.bytes = n_need * CHUNK;
    std::vector<char> used(ch.size(), 0);
    for (size_t i = 0; i < order.size() && e == hipSuccess; ++i) {
        Chunk &c = ch[order[i]];
        e = hipMemUnmap(c.at, CHUNK);
        c.at = nullptr;
        if (e == hipSuccess) e = hipMemMap(va + i * CHUNK, CHUNK, 0, c.h, 0);
        if (e == hipSuccess) { out.handles.push_back((void *)c.h); used[order[i]] = 1; }
    }
    if (e == hipSuccess) e = hipMemSetAccess(va, n_need * CHUNK, &acc, 1);
    const size_t examined = ch.size();
    std::vector<int> label_of(ch.size());
    for (size_t i = 0; i < ch.size(); ++i) label_of[i] = ch[i].label;
    for (size_t i = 0; i < ch.size(); ++i)
        if (!used[i]) {
            if (ch[i].at) (void)hipMemUnmap(ch[i].at, CHUNK);
            (void)hipMemRelease(ch[i].h);
        }
    ch.clear();
    (void)hipMemAddressFree(scratch, slots * CHUNK);   // the surplus chunks' memory goes back to the card with their range
    if (e != hipSuccess) {
        same_spread_release(out);
        return same_fail(ctx, SAME_EIO, "hipMemMap/hipMemSetAccess (spread buffer)", e);
    }
    // Check the finished mapping instead of trusting the plan: (a) one store over the whole range must run at the fast level;
    // (b) neighbouring chunks of the range, sampled, must time against each other as their labels say (different regions:
    // fast, same region: slow) -- an address that still reached an earlier chunk (header comment) would show here.
    const double final_rate = tm.sweep(va, n_need);
    int pairs_checked = 0, pairs_agree = 0;
    {
        const size_t n_pairs = order.size() > 1 ? std::min<size_t>(8, order.size() - 1) : 0;
        for (size_t q = 0; q < n_pairs && tm.rc == SAME_OK; ++q) {
            const size_t i = q * (order.size() - 1) / n_pairs;
            const int la = label_of[order[i]], lb = label_of[order[i + 1]];
            if (la == MIXED || lb == MIXED) continue;            // a straddler is fast against anything, itself included
            double g = tm.pair(va + i * CHUNK, va + (i + 1) * CHUNK);
            const bool want_fast = la != lb;
            if (want_fast && g < slow * LEVEL_RATIO) g = std::max(g, tm.pair(va + i * CHUNK, va + (i + 1) * CHUNK));   // noise only lowers a rate
            ++pairs_checked;
            pairs_agree += ((g >= slow * LEVEL_RATIO) == want_fast) ? 1 : 0;
        }
    }
    if (tm.rc != SAME_OK) {
        same_spread_release(out);
        ctx->err = "timing the finished spread buffer failed; plain allocation";
        return SAME_EIO;
    }
    const bool verified = final_rate >= slow * LEVEL_RATIO && pairs_agree == pairs_checked;
    if (debug) fprintf(stderr, "[spread] %zu chunks examined: %.2f s in hipMemCreate/Map/SetAccess, %.2f s in timed stores, %.2f s in all\n", examined,
                       t_create, t_time, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    ctx->spread.push_back(out);
    *out_dptr = va;
    info[0] = 1;
    info[1] = (int64_t)n_need;
    for (int c = 0; c <= MAX_REGIONS; ++c) info[2 + c] = (int64_t)take[c];
    info[6] = (int64_t)examined;
    info[7] = (int64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
    info[8] = (int64_t)(slow + 0.5);
    info[9] = verified ? 1 : 0;
    info[10] = (int64_t)(final_rate + 0.5);
    info[11] = pairs_checked;
    info[12] = pairs_agree;
    info[13] = timed_out ? 1 : 0;
    if (!verified) ctx->err = "spread buffer mapped, but a store over it did not reach the fast level (info[9] = 0): it is used as it is";
    return SAME_OK;
}

extern "C" int same_dev_alloc_spread(same_ctx *ctx, size_t bytes, void **out_dptr, int64_t *out_info) {
    REQUIRE(ctx, ctx && out_dptr);
    SAME_TRY(same_use(ctx));
    *out_dptr = nullptr;
    int64_t info[SAME_SPREAD_INFO_LEN] = {};
    const size_t n_need = (bytes + CHUNK - 1) / CHUNK;
    const char *env = getenv("SAME_SPREAD");
    size_t free_b = 0, total_b = 0;
    HIP_TRY(ctx, hipMemGetInfo(&free_b, &total_b));
    const size_t reserve = size_t(4) << 30;                        // left to the rest of the process while labelling
    const size_t budget = free_b > reserve ? (free_b - reserve) / CHUNK : 0;
    // small buffers, an explicit opt-out, or a card without room for the chunks: the plain allocation.  So does a failure of
    // the virtual-memory path itself (a driver without it, a sibling process taking the memory meanwhile): WHERE the buffer
    // lies is a matter of speed, never of results, and the reason stays readable through same_last_error()
    int rc = SAME_EIO;
    bool tried = false;
    if (!(env && env[0] == '0') && n_need >= MIN_CHUNKS && budget >= n_need) {
        tried = true;
        rc = spread_build(ctx, n_need, budget, out_dptr, info);
        if (rc != SAME_OK) {
            memset(info, 0, sizeof info);
            *out_dptr = nullptr;
        }
    }
    if (rc != SAME_OK) {
        // the reason the spread path gave up stays readable after the plain allocation succeeded -- but only a reason of THIS
        // call: when the spread path was not attempted at all there is nothing to report
        const std::string why = tried ? ctx->err : std::string();
        // a sibling rank on the same card may be holding look-ahead chunks of its own labelling for a second or two: an
        // out-of-memory answer here is retried a few times before it is believed
        int arc = same_dev_alloc(ctx, bytes, out_dptr);
        for (int attempt = 0; attempt < 4 && arc == SAME_ENOMEM && n_need >= MIN_CHUNKS && bytes <= total_b; ++attempt) {   // never for a request the card cannot hold at all
            std::this_thread::sleep_for(std::chrono::milliseconds(750));
            arc = same_dev_alloc(ctx, bytes, out_dptr);
        }
        SAME_TRY(arc);
        ctx->err = why;
    }
    if (out_info) memcpy(out_info, info, sizeof info);
    return SAME_OK;
}
