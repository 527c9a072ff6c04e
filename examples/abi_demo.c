/* abi_demo.c -- the C ABI used from plain C (no Python, no C++): prune + pair costs + orientation sweep on a
 * tiny instance, the dense tile, the window-merge de-duplication and the device-resident window path, printing the results.  Build (from the repo root):
 *   gcc -std=c11 -Iinclude examples/abi_demo.c -o /tmp/abi_demo -Lsame_amd -lsame_hip -Wl,-rpath,$PWD/same_amd -lm
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "same_hip.h"
#include "same_hip_diag.h" /* same_ctx_stat, same_dev_mem_info, same_dev_alloc_spread: measurement hooks, not the path */

#define CHECK(call)                                                                            \
    do {                                                                                       \
        int rc_ = (call);                                                                      \
        if (rc_ != SAME_OK) {                                                                  \
            fprintf(stderr, "%s -> %d (%s) %s\n", #call, rc_, same_strerror(rc_), ctx ? same_last_error(ctx) : ""); \
            return 1;                                                                          \
        }                                                                                      \
    } while (0)

int main(void) {
    same_ctx *ctx = NULL;
    int n_dev = 0;
    same_device_count(&n_dev);
    if (n_dev < 1) { fprintf(stderr, "no GPU: %s\n", same_strerror(SAME_ENODEV)); return 2; }
    CHECK(same_ctx_create(0, &ctx));
    enum { NM = 6, NR = 7, T = 3, K = 3 };
    /* aligned cells on a line, reference cells jittered around them */
    double axy[NM * 2], rxy[NR * 2], A[NM * T], R[NR * T];
    for (int i = 0; i < NM; ++i) { axy[2 * i] = 10.0 * i; axy[2 * i + 1] = (i % 2) * 8.0; for (int t = 0; t < T; ++t) A[i * T + t] = (t == i % T) ? 90.0 : 5.0; }
    for (int j = 0; j < NR; ++j) { rxy[2 * j] = 10.0 * j + 1.5; rxy[2 * j + 1] = (j % 2) * 8.0 - 1.0; for (int t = 0; t < T; ++t) R[j * T + t] = (t == j % T) ? 80.0 : 10.0; }
    int32_t idx[NM * K], cnt[NM];
    double d2[NM * K];
    CHECK(same_knn_prune(ctx, axy, NM, rxy, NR, 0, NM, 12.0, K, idx, d2, cnt));
    int32_t pairs[NM * K * 2];
    int P = 0;
    for (int i = 0; i < NM; ++i)
        for (int q = 0; q < cnt[i]; ++q) { pairs[2 * P] = i; pairs[2 * P + 1] = idx[i * K + q]; ++P; }
    double cost[NM * K];
    CHECK(same_pair_cost_f64(ctx, A, R, NM, NR, T, axy, rxy, pairs, P, 1.0, cost));
    for (int p = 0; p < P; ++p) {
        const int i = pairs[2 * p], j = pairs[2 * p + 1];
        printf("pair (%d,%d) dist=%.3f cost=%.4f\n", i, j, hypot(rxy[2 * j] - axy[2 * i], rxy[2 * j + 1] - axy[2 * i + 1]), cost[p]);
    }
    /* triangles over the aligned cells, source signs, and the sweep under "nearest reference" matching */
    int32_t tris[4 * 3] = {0, 1, 2, 1, 2, 3, 2, 3, 4, 3, 4, 5};
    int8_t sign[4];
    CHECK(same_tri_sign_weight(ctx, axy, NULL, NM, tris, 4, sign, NULL));
    int32_t match[NM];
    for (int i = 0; i < NM; ++i) match[i] = cnt[i] ? idx[i * K] : -1;
    match[2] = cnt[4] ? idx[4 * K] : -1; /* swap two matches to fold a triangle */
    match[4] = cnt[2] ? idx[2 * K] : -1;
    same_sweep *sweep = NULL;
    CHECK(same_sweep_bind(ctx, tris, 4, sign, rxy, NR, NM, NULL, 0, &sweep));
    int64_t checked = 0, nviol = 0;
    int32_t viol[4];
    CHECK(same_orient_sweep(sweep, match, NM, &checked, viol, &nviol, NULL));
    same_sweep_unbind(sweep);
    printf("checked %lld triangles, %lld flipped:", (long long)checked, (long long)nviol);
    for (int q = 0; q < nviol; ++q) printf(" %d", viol[q]);
    printf("\n");
    /* resident operands: the dense tile of all NM x NR costs built on the device into a block from same_dev_alloc_spread
     * (large blocks are laid over the card's HBM regions; one this small is a plain allocation -- info[0] says which) */
    int64_t mem_free = 0, mem_total = 0, info[SAME_SPREAD_INFO_LEN];
    CHECK(same_dev_mem_info(ctx, &mem_free, &mem_total));
    void *dA = NULL, *dR = NULL, *dax = NULL, *drx = NULL, *dD = NULL;
    const int64_t ld = NR + (NR & 1);   /* even leading dimension: 16-byte rows */
    CHECK(same_dev_alloc(ctx, sizeof A, &dA));   CHECK(same_h2d(ctx, dA, A, sizeof A));
    CHECK(same_dev_alloc(ctx, sizeof R, &dR));   CHECK(same_h2d(ctx, dR, R, sizeof R));
    CHECK(same_dev_alloc(ctx, sizeof axy, &dax)); CHECK(same_h2d(ctx, dax, axy, sizeof axy));
    CHECK(same_dev_alloc(ctx, sizeof rxy, &drx)); CHECK(same_h2d(ctx, drx, rxy, sizeof rxy));
    CHECK(same_dev_alloc_spread(ctx, (size_t)NM * ld * sizeof(double), &dD, info));
    CHECK(same_dense_cost_f64_dev(ctx, dA, dR, T, dax, drx, NR, 0, NM, 1.0, dD, ld));
    double D[NM * (NR + 1)];
    CHECK(same_d2h(ctx, D, dD, (size_t)NM * ld * sizeof(double)));
    printf("card: %.1f of %.1f GiB free; dense tile (%s allocation): D[0][0]=%.4f D[%d][%d]=%.4f\n", mem_free / 1073741824.0,
           mem_total / 1073741824.0, info[0] ? "spread" : "plain", D[0], NM - 1, NR - 1, D[(NM - 1) * ld + NR - 1]);
    void *blocks[] = {dA, dR, dax, drx, dD};
    for (int q = 0; q < 5; ++q) CHECK(same_dev_free(ctx, blocks[q]));
    /* window merge, the de-duplication step (src/helpers.py:745-753): three windows propose matches; pair (aligned 4, ref 9) comes from
     * windows 2, 0 and 1 -- the non-violating proposals win, then the smaller window id: row 3 survives, rows 0 and 5 do not */
    const uint8_t mviol[6] = {0, 1, 0, 0, 1, 1};
    const int32_t mwin[6] = {2, 0, 1, 0, 2, 1}, ma[6] = {4, 7, 5, 4, 6, 4}, mr[6] = {9, 8, 3, 9, 1, 9};
    int32_t mrows[6];
    int64_t mkept = 0;
    CHECK(same_merge_dedup(ctx, mviol, mwin, ma, mr, 6, mrows, &mkept));
    printf("merge de-duplication keeps %lld of 6 rows:", (long long)mkept);
    for (int q = 0; q < mkept; ++q) printf(" %d", mrows[q]);
    printf("\n");
    if (mkept != 4 || mrows[0] != 3 || mrows[1] != 2 || mrows[2] != 1 || mrows[3] != 4) return 4;
    /* the same instance as ONE window with both sections resident on the device (src/same.py:507-593 per window): subsetting, prune,
     * costs and compaction in same_window_stage; the four triangles above play the Delaunay simplices of the kept aligned cells for
     * same_window_filter_finish (radius 30, no angle rule, no type rule; then the greedy MIP start and the three sweeps).  Both calls
     * take a BATCH of windows: here a batch of one */
    double size[NM > NR ? NM : NR];
    for (int i = 0; i < (NM > NR ? NM : NR); ++i) size[i] = 1.0;
    same_section *smov = NULL, *sref = NULL;
    same_window *win = NULL, *win_b = NULL;
    CHECK(same_section_create(ctx, axy, A, T, size, NULL, NM, 0, &smov));
    CHECK(same_section_create(ctx, rxy, R, T, size, NULL, NR, 0, &sref));
    CHECK(same_window_create(ctx, &win));
    CHECK(same_window_create(ctx, &win_b));
    const double box[4] = {-100.0, 100.0, -100.0, 100.0};
    const int64_t one_window[2] = {0, 4};          /* simplex offsets: window 0 owns triangles [0, 4) */
    int64_t wc[4], fc[4], st[8];
    CHECK(same_window_stage(&win, 1, smov, sref, box, 12.0, K, 1.0, wc));
    double wcost[NM * K];
    CHECK(same_window_fetch(win, SAME_WINDOW_COSTS, wcost, wc[3] * (int64_t)sizeof(double)));
    int same_costs = wc[3] == P;
    for (int p = 0; same_costs && p < P; ++p) same_costs = wcost[p] == cost[p];
    int32_t mrow[NM];
    uint8_t pflag[NM];
    CHECK(same_window_filter_finish(&win, 1, tris, one_window, 0, 30.0, 0, 0.0, 0.0, 0, 1, 100.0, mrow, pflag, st, fc));
    printf("window: %lld aligned, %lld ref, %lld kept, %lld pairs (costs %s the pair list's); %lld of 4 triangles kept; matched ref rows:",
           (long long)wc[0], (long long)wc[1], (long long)wc[2], (long long)wc[3], same_costs ? "equal" : "DIFFER FROM", (long long)fc[0]);
    for (int i = 0; i < wc[2]; ++i) printf(" %d", mrow[i]);
    printf("; orientation checked %lld flipped %lld\n", (long long)st[0], (long long)st[1]);
    /* a6 without scipy (ABI 8): the library's own triangulator, host code.  It answers only when its triangles are beyond doubt the
     * SET Qhull gives; the aligned cells above have three hull points on one line, which is exactly where it says "ask Qhull"
     * (SAME_EUNSURE, not an error); the same cells bent off the line are answered.  fc[3] counts the window's ORDER TIES -- places
     * where the sweeps' numbers hang on the order of the triangles or their corners, which only Qhull's own simplices may decide */
    double bent[NM * 2], margin = 0.0;
    for (int i = 0; i < NM; ++i) { bent[2 * i] = axy[2 * i]; bent[2 * i + 1] = axy[2 * i + 1] + 0.37 * i * i; }
    int32_t own[(2 * NM) * 3];
    int64_t n_own = 0, n_bent = 0;
    const int rc_line = same_delaunay2d(axy, NM, own, 2 * NM, &n_own, 16.0, NULL);
    const int rc_bent = same_delaunay2d(bent, NM, own, 2 * NM, &n_bent, 16.0, &margin);
    printf("own triangulator: the cells above %s; bent off the line %s, %lld triangles; order ties of the window: %lld\n",
           rc_line == SAME_EUNSURE ? "are left to Qhull" : "UNEXPECTED", rc_bent == SAME_OK && margin > 16.0 ? "answered" : "UNEXPECTED",
           (long long)n_bent, (long long)fc[3]);
    if (rc_line != SAME_EUNSURE || rc_bent != SAME_OK || n_bent != 5) return 6;
    /* the same window TWICE in one batch (two window states), with the sections re-binned on a 50-unit grid from (-100, -100) -- the box is
     * then a union of cells and no row is tested; the runtime calls the library issued for the batch are read from its counters: per window
     * the launches of before, and ONE wait per call for the two windows together */
    CHECK(same_section_bin(smov, -100.0, -100.0, 50.0, 50.0));
    CHECK(same_section_bin(sref, -100.0, -100.0, 50.0, 50.0));
    same_window *pair[2] = {win, win_b};
    const double boxes2[8] = {-100.0, 100.0, -100.0, 100.0, -100.0, 100.0, -100.0, 100.0};
    const int64_t two_windows[3] = {0, 4, 8};
    int32_t tris2[24];
    for (int q = 0; q < 24; ++q) tris2[q] = tris[q % 12];
    int64_t c0[4], c1[4], wc2[8], fc2[8], st2[16];
    int32_t mrow2[2 * NM];
    uint8_t pflag2[2 * NM];
    for (int q = 0; q < 4; ++q) CHECK(same_ctx_stat(ctx, q, &c0[q]));
    CHECK(same_window_stage(pair, 2, smov, sref, boxes2, 12.0, K, 1.0, wc2));
    CHECK(same_window_filter_finish(pair, 2, tris2, two_windows, 0, 30.0, 0, 0.0, 0.0, 0, 1, 100.0, mrow2, pflag2, st2, fc2));
    for (int q = 0; q < 4; ++q) CHECK(same_ctx_stat(ctx, q, &c1[q]));
    int same_window = 1;
    for (int b = 0; b < 2; ++b) {
        same_window = same_window && fc2[4 * b] == fc[0] && fc2[4 * b + 1] == fc[1];
        for (int q = 0; q < 4; ++q) same_window = same_window && wc2[4 * b + q] == wc[q];
        for (int q = 0; q < 8; ++q) same_window = same_window && st2[8 * b + q] == st[q];
        for (int i = 0; i < wc[2]; ++i) same_window = same_window && mrow2[b * wc[2] + i] == mrow[i] && pflag2[b * wc[2] + i] == pflag[i];
    }
    printf("the window twice in one batch (binned sections): %s; %lld launches, %lld fills, %lld copies, %lld waits for the two\n",
           same_window ? "the same answers" : "DIFFERENT ANSWERS", (long long)(c1[0] - c0[0]), (long long)(c1[1] - c0[1]),
           (long long)(c1[2] - c0[2]), (long long)(c1[3] - c0[3]));
    if (!same_window || c1[0] - c0[0] > 60 || c1[3] - c0[3] > 2) return 6;
    /* the window merge where the matches are (src/helpers.py:692-815): both windows' central rows go to an accumulator on the device
     * (enqueue only), the merge keeps one row per (aligned, ref) pair -- here every pair was proposed by both windows: window 0's rows win --,
     * settles the rows that stand alone, and writes the merged table's columns into host memory the device can write */
    same_merge_acc *acc = NULL;
    const int32_t wids[2] = {0, 1}, ppos[2] = {0, 1};
    int64_t mc[4], n_final = 0;
    CHECK(same_merge_acc_create(ctx, &acc));
    CHECK(same_merge_acc_begin(acc, 2 * NM, 0, NULL, NULL, 0.0, 0));
    CHECK(same_window_collect(pair, 2, acc, boxes2, wids, ppos));
    CHECK(same_merge_acc_resolve(&acc, 1, smov, sref, mc));
    CHECK(same_merge_acc_finish(acc, NULL, 0, &n_final));      /* no row was contested: nothing for a matching to decide */
    struct { int32_t a_row, r_row, cidx, wid, pos; uint32_t flags; } fin[NM];
    if (n_final > NM || sizeof fin[0] != SAME_MERGE_FINAL_BYTES) return 7;
    CHECK(same_merge_acc_fetch(acc, SAME_MERGE_FINAL, fin, n_final * SAME_MERGE_FINAL_BYTES));
    void *block = NULL;
    const size_t n8 = (size_t)T + 4 + 3;                       /* type columns, X, Y, ref_X, ref_Y, aligned_idx, window_id, plan position */
    CHECK(same_host_alloc(ctx, n8 * 8 * (size_t)n_final + 2 * (size_t)n_final + 8, &block));
    CHECK(same_merge_acc_columns(acc, smov, sref, NULL, 0, NULL, 0, block, n_final));
    CHECK(same_ctx_sync(ctx));
    const double *col = (const double *)block;
    int merged_ok = mc[0] == 2 * mc[1] && mc[2] == 0 && mc[3] == n_final && n_final == mc[1];
    for (int64_t i = 0; merged_ok && i < n_final; ++i)
        merged_ok = fin[i].wid == 0 && fin[i].r_row == mrow[fin[i].cidx] && col[(size_t)T * n_final + i] == axy[2 * fin[i].a_row]
                    && col[((size_t)T + 2) * n_final + i] == rxy[2 * fin[i].r_row] && (i == 0 || fin[i].a_row > fin[i - 1].a_row);
    printf("window merge on the device: %lld rows from two windows -> %lld after the de-duplication, %lld left to the host, %lld merged rows (%s)\n",
           (long long)mc[0], (long long)mc[1], (long long)mc[2], (long long)n_final, merged_ok ? "window 0's, aligned rows ascending" : "WRONG");
    CHECK(same_host_free(ctx, block));
    same_merge_acc_destroy(acc);
    if (!merged_ok) return 7;
    same_window_destroy(win_b);
    same_window_destroy(win);
    same_section_destroy(smov);
    same_section_destroy(sref);
    if (!same_costs || wc[0] != NM || wc[1] != NR) return 5;
    /* error convention: a bad index is reported, not dereferenced */
    int32_t bad[2] = {0, 99};
    int rc = same_pair_cost_f64(ctx, A, R, NM, NR, T, axy, rxy, bad, 1, 1.0, cost);
    printf("bad pair -> %d (%s)\n", rc, same_strerror(rc));
    same_ctx_destroy(ctx);
    return rc == SAME_ERANGE ? 0 : 3;
}
