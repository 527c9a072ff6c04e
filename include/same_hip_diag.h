/*
 * same_hip_diag.h -- measurement hooks and opt-in controls of libsame_hip.so that are NOT part of the path's boundary
 * (include/same_hip.h): what bench.py, the profiling tools and the tests read -- runtime-call counters, HIP-event timers, the
 * card's memory and PCI id, the spread allocator for 80 GB streaming outputs, the all-gather's device time -- and the fixed-point
 * dense build that serves as the roofline control.  A reference maintainer binding the path needs none of these.
 */
#ifndef SAME_HIP_DIAG_H
#define SAME_HIP_DIAG_H

#include "same_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- what the library asked of the runtime ------------------------------------------------------------------------------- */
/* What the library itself has asked of the HIP runtime on this context since it was created, for the entry points that count
 * (the window path, the greedy start): kernel launches, hipMemsetAsync fills, hipMemcpyAsync copies, stream waits, and the
 * device-to-host reads the greedy rounds made.  The difference of two reads around a call is that call's cost in runtime calls --
 * the number a rocprof trace shows, available to a test (tests/test_gpu_run_same.py holds launches / fills / copies / waits per
 * window).  which = SAME_STAT_*. */
enum { SAME_STAT_LAUNCHES = 0, SAME_STAT_FILLS = 1, SAME_STAT_COPIES = 2, SAME_STAT_WAITS = 3, SAME_STAT_GREEDY_READBACKS = 4, SAME_STAT_COUNT = 5 };
int same_ctx_stat(same_ctx *ctx, int which, int64_t *out);


/* ---- timers -------------------------------------------------------------------------------------------------------------- */
/* HIP events recorded on the context's stream (where the kernels run). */
int same_timer_start(same_ctx *ctx);
int same_timer_stop(same_ctx *ctx, float *out_ms); /* records, synchronises, returns elapsed ms */
/* the same in two steps: mark the end now (no wait), read the elapsed time later */
int same_timer_mark(same_ctx *ctx);
int same_timer_read(same_ctx *ctx, float *out_ms);

/* ---- the card ------------------------------------------------------------------------------------------------------------ */
/* "domain:bus:device.function" of the context's GPU (names its sysfs directory: power / clock telemetry) */
int same_ctx_pci_bus_id(same_ctx *ctx, char *out, size_t out_len);
/* free and total bytes of the context's card right now (hipMemGetInfo); either pointer may be NULL */
int same_dev_mem_info(same_ctx *ctx, int64_t *out_free, int64_t *out_total);

/* ---- placement of large streaming outputs ---------------------------------------------------------------------------------- */
/* For LARGE STREAMING OUTPUTS (the dense cost matrix).  MI355X's HBM is three physical regions of 96 GiB and a streaming
 * store confined to one of them runs ~20 % below one spread over two or three (profiles/archive/r02_hbm_regions.md); hipMalloc
 * places a buffer wherever its free lists point.  This call takes the memory in 1 GiB chunks through the virtual-memory
 * API, finds each chunk's region by timed stores and maps the chunks round-robin over the regions into one contiguous
 * range.  The result is used and freed like any same_dev_alloc buffer.
 *  - Cost: 1-4 s once for 75 GiB (most of it the driver's own hipMemCreate); may hold up to 128 GiB more than `bytes`
 *    while it looks for chunks of a second region; all of that goes back to the card before the call returns.
 *  - Buffers under 6 GiB, SAME_SPREAD=0 in the environment, a card without that much free memory, or a failure of the
 *    virtual-memory calls themselves give a plain same_dev_alloc: placement is a matter of speed, never of results
 *    (out_info[0] says which it was; same_last_error() keeps the reason).
 *  - The memory goes back to the card on same_dev_free.  The ADDRESSES of a spread buffer are never used for another
 *    mapping (a ROCm quirk, see spread.hip): they come from a 48 TiB stretch of the process's address space, after which
 *    the plain allocation is used.
 *  - The finished range is checked, not trusted: one store over all of it is timed and sampled neighbouring chunks are
 *    timed against each other; out_info[9] says whether the store ran at the fast level and the pairs behaved as labelled.
 *    An unverified buffer is still returned (and is still correct memory): only its speed is in question.
 *  - Wall time is bounded: past SAME_SPREAD_MAX_SECONDS (default 3) the search for better-balanced chunks stops and the best
 *    choice so far is mapped; past twice that while still taking the buffer's own chunks, the plain allocation is used.
 * out_info (may be NULL), SAME_SPREAD_INFO_LEN int64: [0] 1 = spread, 0 = plain; [1] GiB chunks mapped; [2..4] chunks from
 * region 0/1/2; [5] chunks that straddle regions; [6] chunks examined; [7] microseconds spent; [8] same-region level, GB/s;
 * [9] 1 = verified; [10] GB/s of one store over the finished range; [11] neighbouring pairs timed, [12] of them as labelled;
 * [13] 1 = the search stopped at the time bound. */
#define SAME_SPREAD_INFO_LEN 14
int same_dev_alloc_spread(same_ctx *ctx, size_t bytes, void **out_dptr, int64_t *out_info);

/* ---- opt-in fixed-point dense build ---------------------------------------------------------
 * NOT the reference's arithmetic and never a default: the type values are put on a common 32-bit
 * fixed-point grid q(v) = rint((v - offset) * scale), the type sum becomes an exact integer sum of
 * absolute differences (one v_sad_u32 per element instead of two fp64 adds), the rest of the expression
 * is unchanged fp64:  out = w * (double(S_q) * inv_scale) + (w*0.001) * (|ax-rx| + |ay-ry|).
 * Against same_dense_cost_f64_dev: |S_q * inv_scale - S| <= T * inv_scale.  With rel_tol > 0 every type
 * sum of fewer than T / rel_tol + T grid steps (near-identical cells) is recomputed from the fp64
 * matrices dA / dR with the reference's own expression, so EVERY output is within rel_tol (relative) of
 * the fp64 build's -- rel_tol = 1e-6 is BASELINE.json's tolerance for fp64 costs; rel_tol = 0 keeps the
 * pure grid result (dA / dR may then be NULL).  The caller chooses offset / scale so that every
 * row-pair sum fits 32 bits (same_amd.ops.quantize_types); the row pitch ld must be even and columns
 * [n_r, ld) are written too (padding owned by the caller).  T <= SAME_Q32_MAX_TYPES.  Meant for the
 * dense matrix of the Hungarian MIP-start heuristic (src/init_helpers.py:151-155) and as the roofline
 * control of DESIGN.md 5.1 (the same 80 GB of stores without the fp64 adds). */
#define SAME_Q32_MAX_TYPES 32
int same_quantize_u32_dev(same_ctx *ctx, const double *dsrc, int64_t n, double offset, double scale,
                          uint32_t *ddst);
int same_dense_cost_q32_dev(same_ctx *ctx, const uint32_t *dAq, const uint32_t *dRq, const double *dA,
                            const double *dR, int T, const double *daxy, const double *drxy, int64_t n_r,
                            int64_t row_begin, int64_t row_end, double w, double inv_scale,
                            double rel_tol, double *dout, int64_t ld);

/* ---- the all-gather's own time -------------------------------------------------------------------------------------------- */
/* Device time (HIP events on the stream they ran on) of the all-gathers issued since the last same_comm_wait or the last
 * call of this function, whichever came later -- the overlapped ones, or the in-stream ones issued outside a group -- and the
 * bytes this rank sent in them (may be NULL); waits for the last of them.  0 ms if none.  Reading closes the batch: the next
 * gather starts a new one (a caller that only uses the in-stream form never needs same_comm_wait). */
int same_comm_gather_time(same_ctx *ctx, float *out_ms, int64_t *out_send_bytes);

#ifdef __cplusplus
}
#endif
#endif /* SAME_HIP_DIAG_H */
