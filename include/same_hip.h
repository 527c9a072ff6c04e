/*
 * same_hip.h -- C ABI of libsame_hip.so: the MI355X (gfx950) implementation of SAME's
 * pre-MIP data-parallel path.
 *
 * The reference (rohitsinghlab/SAME) is pure Python and has no FFI; its boundary for this
 * path is a set of Python functions and inline loops.  Each entry
 * point below names the reference code it replaces (file:line into the reference tree).
 * The Python host layer in same_amd/ keeps the reference's signatures and calls these
 * through ctypes; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *  - plain C: pointers + sizes only, no C++/torch types.
 *  - every function returns 0 on success or a negative SAME_E* code; on failure outputs are
 *    unspecified but never written out of bounds.  same_strerror() names the code and
 *    same_last_error() returns the HIP/RCCL detail recorded on the context.
 *  - "host" entry points take caller-owned, C-contiguous host buffers; the library copies
 *    in, runs on the context's own stream, copies out and is synchronous on return.  It
 *    never keeps a host pointer past the call.
 *  - "_dev" entry points take device pointers obtained from same_dev_alloc() and only
 *    enqueue work on the context's stream (call same_ctx_sync() to wait).  They exist so
 *    large operands (the dense cost matrix is 80 GB at 100k x 100k) stay resident in HBM.
 *  - indices are int32, sizes int64.  pairs are int32 [P][2] = (aligned i, ref j).
 *    triangles are int32 [Tr][3] of aligned-row indices.  xy arrays are double [n][2].
 *    type matrices are row-major double [n][T] (the commonCT columns, in commonCT order).
 *  - threading: a context is used by one caller at a time; any thread may make the call
 *    (every entry point selects the context's device itself).
 */
#ifndef SAME_HIP_H
#define SAME_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAME_OK 0
#define SAME_EINVAL (-22)   /* bad shape / NULL / unsupported size */
#define SAME_ENOMEM (-12)   /* host or device allocation failed */
#define SAME_EIO (-5)       /* HIP or RCCL call failed; see same_last_error() */
#define SAME_ENODEV (-19)   /* no usable GPU */
#define SAME_ERANGE (-34)   /* an index in pairs/triangles/match is out of range */
#define SAME_EUNSURE (-11)  /* same_delaunay2d only: not an error -- the points are too close to degenerate to answer for Qhull */

#define SAME_ABI_VERSION 8
#define SAME_MAX_KNN 448     /* largest k supported by the prune kernel (k <= 64 runs the 8-rows-per-wave form) */
#define SAME_MAX_TYPES 4096  /* largest T (type columns) */

typedef struct same_ctx same_ctx;
typedef struct same_sweep same_sweep; /* resident state of one lazy-constraint sweep (same_sweep_bind) */

/* ---- index of this header ------------------------------------------------------------------------------------------------------
 * part 0  context, device memory                                     same_ctx_*, same_dev_*, same_h2d / d2h / d2d
 * part 1  HOST-BUFFER entry points (caller's arrays in and out; one   same_pair_cost_*, same_dense_cost_f64 / f32, same_knn_prune, same_tri_*, same_sweep_* /
 *         call = upload, kernels, download, wait)                     same_orient_sweep*, same_xyorder_sweep, same_area_flip, same_pair_rowmin,
 *                                                                     same_assign_matrix, same_greedy_*, same_tri_flip_stats, same_collapse_candidates,
 *                                                                     same_batched_assign, same_eager_signs, same_window_count, same_merge_dedup
 * part 2  DEVICE-RESIDENT forms (operands already in HBM; enqueue     same_dense_cost_*_dev, same_knn_prune_dev, same_knn_index_*,
 *         only unless noted)                                          same_knn_prune_indexed_dev, same_padded_cost_*_dev, same_tri_*_dev,
 *                                                                     same_area_flip_dev, same_xyorder_sweep_dev, same_orient_*_dev, same_first_candidate_dev
 * part 3  WINDOW path, sections resident (BASELINE cfg 5)             same_section_*, same_window_*, same_merge_acc_*, same_delaunay2d (host)
 * part 4  COMM: RCCL collectives between the ranks' contexts          same_comm_*, same_allgather_dev*, same_allreduce_dev
 * Every declaration cites the reference lines it replaces (file:line into the reference tree).
 * Measurement hooks and opt-in controls that are NOT the path -- runtime-call counters, timers, the spread allocator, the fixed-point
 * dense build, the all-gather's device time -- are declared in same_hip_diag.h (same library). */

/* ======================================================================================================================
 * part 0 -- context, device memory
 * ====================================================================================================================== */
/* ---- context ------------------------------------------------------------------------- */
int same_abi_version(void);
int same_device_count(int *out_count);
int same_ctx_create(int device, same_ctx **out);
void same_ctx_destroy(same_ctx *ctx);
int same_ctx_sync(same_ctx *ctx);
const char *same_strerror(int code);
const char *same_last_error(same_ctx *ctx);
/* device name, CU count, HBM bytes (any pointer may be NULL) */
int same_ctx_info(same_ctx *ctx, char *name, size_t name_len, int *cu_count, int64_t *hbm_bytes);
/* ---- device memory (for resident operands) ------------------------------------------------ */
int same_dev_alloc(same_ctx *ctx, size_t bytes, void **out_dptr);
int same_dev_free(same_ctx *ctx, void *dptr);   /* either kind of buffer */
int same_h2d(same_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int same_d2h(same_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
int same_dev_memset(same_ctx *ctx, void *dst_dev, int value, size_t bytes);
/* device-to-device copy, enqueued on the context's stream (no wait) */
int same_d2d(same_ctx *ctx, void *dst_dev, const void *src_dev, size_t bytes);
/* The host-buffer entry points stage through per-context scratch blocks that grow on demand and are
 * reused across calls; this frees them all (same_sweep handles own their blocks and are not affected). */
int same_ctx_release_scratch(same_ctx *ctx);
/* ======================================================================================================================
 * part 1 -- HOST-BUFFER entry points: the caller's (NumPy) arrays in and out; synchronous
 * ====================================================================================================================== */
/* ---- a4: pair costs -------------------------------------------------------------------
 * Replaces the loop at src/same.py:1180-1189:
 *   c[p] = w * sum_t |A[i,t]-R[j,t]|  +  (w*0.001) * (|ax-rx| + |ay-ry|),  (i,j) = pairs[p]
 * with the type sum accumulated left to right in fp64 and no fused multiply-add. */
int same_pair_cost_f64(same_ctx *ctx, const double *A, const double *R, int64_t n_m, int64_t n_r,
                       int T, const double *axy, const double *rxy, const int32_t *pairs,
                       int64_t P, double w, double *out_c);
/* fp32 variant (BASELINE.json config 5: "fp32 cost"): the same expression evaluated in float, operands given as
 * float -- element (i, j) of same_dense_cost_f32.  The reference has no fp32 path; the check is bit-equality
 * with the test oracle's float twin plus a forward error bound against the fp64 costs. */
int same_pair_cost_f32(same_ctx *ctx, const float *A, const float *R, int64_t n_m, int64_t n_r,
                       int T, const float *axy, const float *rxy, const int32_t *pairs,
                       int64_t P, float w, float *out_c);

/* ---- dense cost tile builder ----------------------------------------------------------
 * The same expression for every (i, j), i in [row_begin,row_end), j in [0,n_r):
 *   out[(i-row_begin)*ld + j].  Generalises the only dense matrix of the reference
 * (src/init_helpers.py:151-155) to the metric of BASELINE.json (100k x 100k).
 * The device-resident forms (same_dense_cost_*_dev, part 2) are what large operands use: this form copies in, builds, copies out. */
int same_dense_cost_f64(same_ctx *ctx, const double *A, const double *R, int64_t n_m, int64_t n_r,
                        int T, const double *axy, const double *rxy, int64_t row_begin,
                        int64_t row_end, double w, double *out, int64_t ld);
int same_dense_cost_f32(same_ctx *ctx, const float *A, const float *R, int64_t n_m, int64_t n_r,
                        int T, const float *axy, const float *rxy, int64_t row_begin,
                        int64_t row_end, float w, float *out, int64_t ld);

/* ---- a2: KNN prune within a radius ----------------------------------------------------
 * Replaces the per-row body of utils.find_knn_within_radius (src/utils.py:720-728):
 * for aligned rows [row_begin,row_end): refs with dx*dx+dy*dy <= radius*radius (cKDTree
 * query_ball_point, p=2), ranked by (squared distance, ref index) ascending, first k kept.
 * out_idx[(i-row_begin)*k + q] (-1 padded), out_d2 likewise (+inf padded; may be NULL),
 * out_cnt[i-row_begin] = number kept.  The frame compaction of src/utils.py:734-742 stays
 * on the host (same_amd/knn.py). */
int same_knn_prune(same_ctx *ctx, const double *axy, int64_t n_m, const double *rxy, int64_t n_r,
                   int64_t row_begin, int64_t row_end, double radius, int k, int32_t *out_idx,
                   double *out_d2, int32_t *out_cnt);

/* ---- a7: triangle classes -------------------------------------------------------------
 * Replaces the per-triangle decisions of helpers.filter_triangles_by_radius
 * (src/helpers.py:303-341).  out_class: 0 kept, 1 max side >= radius, 2 min angle too
 * small, 3 passed both but all three type_id equal (type_id NULL = test off).
 * The angle rule degrees(arccos(c)) < min_angle_deg is passed as a cosine threshold
 * (angle_enabled, cos_thr): fails iff clipped cosine >= cos_thr; out_maxcos gets the
 * largest clipped corner cosine (2.0 if a side has zero length).  out_perim = s1+s2+s3.
 * The re-add pass (src/helpers.py:365-389) stays on the host (same_amd/triangles.py). */
int same_tri_classify(same_ctx *ctx, const double *xy, int64_t n_pts, const int32_t *tris,
                      int64_t Tr, double radius, int angle_enabled, double cos_thr,
                      const int32_t *type_id, uint8_t *out_class, double *out_perim,
                      double *out_maxcos);

/* ---- a8: triangle weights and source orientation signs --------------------------------
 * Replaces src/same.py:1128-1135 and :1139-1146.  size/out_weight may both be NULL. */
int same_tri_sign_weight(same_ctx *ctx, const double *xy, const double *size, int64_t n_pts,
                         const int32_t *tris, int64_t Tr, int8_t *out_sign, double *out_weight);

/* ---- a10: lazy-constraint orientation sweep -------------------------------------------
 * Replaces the body of _lazy_orientation_callback (src/same.py:631-669).
 * same_sweep_bind uploads triangles, source signs, reference XY and the pair list once (the model._*
 * state of src/same.py:1153-1158) into device blocks owned by the returned handle; pairs may be NULL
 * when only the match-vector form is used.  Several handles may live on one context (one per window /
 * per model); same_sweep_unbind frees one.  Every sweep call names the length of the array it passes and
 * fails with SAME_EINVAL if it is not the bound one, so a handle can never be run against another
 * model's shapes.  Calls on handles of one context must not overlap (the context has one stream), and
 * every handle must be unbound before its context is destroyed.
 * same_orient_sweep_x: x_vals[P] -> matching (last pair with x > 0.5 wins per aligned row,
 * src/same.py:634-639) -> sweep.  out_viol_idx needs room for Tr entries and comes out
 * ascending; out_flag (may be NULL): 0 skipped, 1 checked, 2 flipped;
 * out_match / out_pair_idx (may be NULL): the matching and its pair indices (for cbLazy). */
int same_sweep_bind(same_ctx *ctx, const int32_t *tris, int64_t Tr, const int8_t *src_sign,
                    const double *rxy, int64_t n_r, int64_t n_m, const int32_t *pairs, int64_t P,
                    same_sweep **out);
void same_sweep_unbind(same_sweep *sweep);
int same_orient_sweep(same_sweep *sweep, const int32_t *match, int64_t n_m, int64_t *out_checked,
                      int32_t *out_viol_idx, int64_t *out_nviol, uint8_t *out_flag);
int same_orient_sweep_x(same_sweep *sweep, const double *x_vals, int64_t P, int64_t *out_checked,
                        int32_t *out_viol_idx, int64_t *out_nviol, uint8_t *out_flag,
                        int32_t *out_match, int32_t *out_pair_idx);

/* ---- a11: XY-order preservation sweep -------------------------------------------------
 * Replaces the triangle loop of violationhelper.verify_spatial_preservation
 * (src/violationhelper.py:53-117).  edge_flags[t*3+e]: bit0 compared, bit1 X violated,
 * bit2 Y violated; e = (v0,v1),(v0,v2),(v1,v2).  counts = {comparisons, violations,
 * violated triangles}.  point_flag[n_m]. */
int same_xyorder_sweep(same_ctx *ctx, const double *axy, int64_t n_m, const double *rxy,
                       int64_t n_r, const int32_t *tris, int64_t Tr, const int32_t *match,
                       uint8_t *edge_flags, uint8_t *tri_flag, uint8_t *point_flag,
                       int64_t counts[3]);

/* ---- a12: signed areas before/after and flips -----------------------------------------
 * Replaces src/same.py:1362-1402 with helpers.calculate_signed_area (src/helpers.py:73-77).
 * after = NaN unless all three vertices are matched. */
int same_area_flip(same_ctx *ctx, const double *axy, int64_t n_m, const double *rxy, int64_t n_r,
                   const int32_t *tris, int64_t Tr, const int32_t *match, double *out_before,
                   double *out_after, uint8_t *out_matched3, uint8_t *out_flipped);

/* ---- a5: MIP-start helpers ------------------------------------------------------------
 * Per-row minimum pair cost (src/init_helpers.py:118-122; +inf for rows without pairs) and
 * the dense assignment matrix [n_m][n_r+n_m] (src/init_helpers.py:151-155). */
int same_pair_rowmin(same_ctx *ctx, const int32_t *pairs, const double *costs, int64_t P,
                     int64_t n_m, double *out_min);
int same_assign_matrix(same_ctx *ctx, const int32_t *pairs, const double *costs, int64_t P,
                       const double *unmatched, int64_t n_m, int64_t n_r, double big_m,
                       double *out);

/* ---- f1: greedy MIP start resolved on the device, node-local flip statistics ----------
 * same_greedy_match replaces the sort + sequential scan of src/init_helpers.py:109-133 with an
 * equivalent parallel rule (a pair is taken when it is the (cost, pair index)-minimum at both of
 * its endpoints among the pairs still alive).  prefer[i] = best_cost_i < unmatched_cost_i
 * (:118-122).  out_match_pair[i] = index of the pair chosen for aligned row i, or -1.
 * same_tri_flip_stats replaces the triangle loop of eval_utils.check_triangle_violations
 * (src/eval_utils.py:123-187) on node-indexed arrays: out_tri_flag bit0 all matched, bit1 same
 * type (type_id NULL = off), bit2 flipped; node counters over non-same-type triangles. */
int same_greedy_match(same_ctx *ctx, const int32_t *pairs, const double *costs, int64_t P,
                      int64_t n_m, int64_t n_r, const uint8_t *prefer, int32_t *out_match_pair,
                      int *out_rounds);
int same_tri_flip_stats(same_ctx *ctx, const double *axy, const double *mapped_xy,
                        const uint8_t *matched, int64_t n, const int32_t *type_id,
                        const int32_t *tris, int64_t Tr, uint8_t *out_tri_flag,
                        uint32_t *out_node_tri, uint32_t *out_node_flip);

/* ---- f2: metacell collapse (metacell_utils.greedy_triangle_collapse) -----------------------
 * same_collapse_candidates replaces the per-triangle work of one collapse iteration
 * (src/metacell_utils.py:233-260 validity, :393-431 candidate test and priority):
 * out_flag bit0 = valid (no edge > r_max, no corner angle < min_angle, as a cosine threshold),
 * bit1 = collapsible (valid, one cell type, size sum <= max_size); out_perim = priority;
 * out_total = size sum.  same_greedy_disjoint replaces the sort + vertex-disjoint scan of
 * :438-449 on items[M][3] with keys[M] (ties by item index), out_selected[M]. */
int same_collapse_candidates(same_ctx *ctx, const double *xy, int64_t n, const int32_t *tris,
                             int64_t Tr, int rmax_enabled, double r_max, int angle_enabled,
                             double cos_thr, const int32_t *type_id, const double *size,
                             double max_size, uint8_t *out_flag, double *out_perim,
                             double *out_total);
int same_greedy_disjoint(same_ctx *ctx, const int32_t *items, const double *keys, int64_t M,
                         int64_t n_nodes, uint8_t *out_selected, int *out_rounds);

/* ---- f4: unpack_metacell_matches(strategy='nearest') --------------------------------------
 * Replaces the per-match cdist + np.tile + scipy linear_sum_assignment of
 * src/metacell_utils.py:711-761 with one batched launch.  Problem p assigns the aligned members
 * axy[a_off[p]..a_off[p+1]) to the ref members rxy[r_off[p]..r_off[p+1]) (CSR offsets, n_prob+1
 * entries, first 0); when there are more aligned than ref members every ref member may be used
 * ceil(n_a/n_r) times (tiled columns, :744-748).  out_ref[a_off[p]+i] = index within problem p's
 * ref members given to aligned member i; ties resolve as scipy's solver resolves them.
 * SAME_ERANGE if a problem has non-finite coordinates (scipy raises ValueError there);
 * SAME_EINVAL if a problem has aligned members but no ref member, or more than
 * SAME_ASSIGN_MAX_MEMBERS on either side (one lane solves one problem in O(n^3): metacells are
 * a handful of cells; the bound keeps a malformed input from occupying the GPU for minutes). */
#define SAME_ASSIGN_MAX_MEMBERS 512
int same_batched_assign(same_ctx *ctx, int64_t n_prob, const int64_t *a_off, const int64_t *r_off,
                        const double *axy, const double *rxy, int32_t *out_ref);

/* ---- a14: eager reference-orientation signs -------------------------------------------
 * Replaces calc_ref_area over all candidate combinations (src/helpers.py:425-441,455-510):
 * out[((t*k+x)*k+y)*k+z] = sign(round(cross, 3)) for cand[tris[t][0]][x], ..., 2 if any is -1. */
int same_eager_signs(same_ctx *ctx, const double *rxy, int64_t n_r, const int32_t *tris,
                     int64_t Tr, const int32_t *cand, int64_t n_m, int k, int8_t *out);

/* ---- a13: window membership -----------------------------------------------------------
 * subset_data (src/same.py:293-295) for a batch of boxes: boxes[b] = {x0,x1,y0,y1},
 * half-open.  out_count[b] = rows inside; out_mask (may be NULL) [n_boxes][n]. */
int same_window_count(same_ctx *ctx, const double *xy, int64_t n, const double *boxes,
                      int64_t n_boxes, int64_t *out_count, uint8_t *out_mask);

/* ---- f3: window merge, the de-duplication step ------------------------------------------
 * Replaces src/helpers.py:745-753 (merged_df.sort_values(['filtered_violation', 'window_id'], kind='mergesort') then
 * drop_duplicates([aligned, ref], keep='first')): rows i = 0..n-1 of the concatenated per-window match tables carry a
 * violation flag, a window id and integer codes of their aligned / ref ids (equal ids <=> equal codes; any
 * non-negative int32).  out_rows receives the indices of the rows that survive, in the order the reference's frame has
 * after those two calls (stable by violation, then window id; first row of every pair); *out_n their number.
 * The maximum-cardinality matching that follows (:755-815) is sequential and stays with the caller. */
int same_merge_dedup(same_ctx *ctx, const uint8_t *viol, const int32_t *window_id, const int32_t *aligned_code,
                     const int32_t *ref_code, int64_t n, int32_t *out_rows, int64_t *out_n);

/* ======================================================================================================================
 * part 2 -- DEVICE-RESIDENT forms: every pointer is a device pointer of the context's GPU; calls enqueue on the context's stream
 * ====================================================================================================================== */
/* ---- dense cost tile builder, operands resident (the roofline kernel of the bench; expression and indexing as
 * same_dense_cost_f64 in part 1): all pointers are device pointers; dA / dR / daxy / drxy hold the full arrays. */
int same_dense_cost_f64_dev(same_ctx *ctx, const double *dA, const double *dR, int T,
                            const double *daxy, const double *drxy, int64_t n_r,
                            int64_t row_begin, int64_t row_end, double w, double *dout, int64_t ld);
int same_dense_cost_f32_dev(same_ctx *ctx, const float *dA, const float *dR, int T,
                            const float *daxy, const float *drxy, int64_t n_r, int64_t row_begin,
                            int64_t row_end, float w, float *dout, int64_t ld);

/* ---- a2 on resident operands: the prune of same_knn_prune (part 1) with device pointers, the caller-held index, the costs of
 * the padded candidate lists */
int same_knn_prune_dev(same_ctx *ctx, const double *daxy, const double *drxy, int64_t n_r,
                       int64_t row_begin, int64_t row_end, double radius, int k,
                       int32_t *dout_idx, double *dout_d2, int32_t *dout_cnt);
/* Caller-held index of one reference set for one radius.  same_knn_prune_dev rebuilds the uniform grid
 * of the references (a counting sort) and reads their bounding box back on every call; when the same
 * references are pruned against repeatedly -- windows that share a reference section, the row blocks of
 * a sharded build, a benchmark step -- build the index once: same_knn_prune_indexed_dev then only
 * enqueues the query kernel (no rebuild, no host synchronisation) and returns bit-identical lists.
 * drxy is the caller's device array: it is not copied for the brute-force plan (small or degenerate
 * sets), so it must stay valid and unchanged until same_knn_index_destroy. */
typedef struct same_knn_index same_knn_index;
int same_knn_index_build(same_ctx *ctx, const double *drxy, int64_t n_r, double radius,
                         same_knn_index **out);
void same_knn_index_destroy(same_knn_index *index);
int same_knn_prune_indexed_dev(same_ctx *ctx, const same_knn_index *index, const double *daxy,
                               int64_t row_begin, int64_t row_end, int k, int32_t *dout_idx,
                               double *dout_d2, int32_t *dout_cnt);
/* costs of the padded candidate lists (a2 + a4 fused for the sharded path):
 * out_cost[(i-row_begin)*k + q] = pair cost of (i, idx[..]) or +inf where idx == -1. */
int same_padded_cost_f64_dev(same_ctx *ctx, const double *dA, const double *dR, int T,
                             const double *daxy, const double *drxy, int64_t row_begin,
                             int64_t row_end, int k, const int32_t *didx, double w,
                             double *dout_cost);
int same_padded_cost_f32_dev(same_ctx *ctx, const float *dA, const float *dR, int T,
                             const float *daxy, const float *drxy, int64_t row_begin,
                             int64_t row_end, int k, const int32_t *didx, float w,
                             float *dout_cost);

/* ---- device-resident forms of the triangle kernels and sweeps -------------------------
 * Same semantics as the host-buffer entry points above; every pointer is a device pointer
 * and the call only enqueues on the context's stream.  Index ranges are the caller's
 * responsibility here (the host-buffer forms validate them).  dcounts = 3 x uint64. */
int same_tri_classify_dev(same_ctx *ctx, const double *dxy, const int32_t *dtris, int64_t Tr,
                          double radius, int angle_enabled, double cos_thr, const int32_t *dtype_id,
                          uint8_t *dout_class, double *dout_perim, double *dout_maxcos);
int same_tri_sign_weight_dev(same_ctx *ctx, const double *dxy, const double *dsize,
                             const int32_t *dtris, int64_t Tr, int8_t *dout_sign, double *dout_weight);
int same_area_flip_dev(same_ctx *ctx, const double *daxy, const double *drxy, const int32_t *dtris,
                       int64_t Tr, const int32_t *dmatch, double *dout_before, double *dout_after,
                       uint8_t *dout_matched3, uint8_t *dout_flipped);
int same_xyorder_sweep_dev(same_ctx *ctx, const double *daxy, int64_t n_m, const double *drxy,
                           const int32_t *dtris, int64_t Tr, const int32_t *dmatch,
                           uint8_t *dedge_flags, uint8_t *dtri_flag, uint8_t *dpoint_flag,
                           uint64_t *dcounts);
int same_orient_sweep_dev(same_sweep *sweep, const int32_t *dmatch, int64_t *out_checked,
                          int32_t *out_viol_idx, int64_t *out_nviol);
/* Triangle-block forms for the sweep sharded over GPUs (the loops being sharded are
 * src/same.py:645-669 and src/violationhelper.py:53-117).  same_orient_flags_dev writes the flags of
 * triangles [t_begin, t_end) at their absolute positions dflag[t] (any block boundaries) and only
 * enqueues; after the blocks of all ranks have been all-gathered into one flag array,
 * same_orient_from_flags_dev produces what same_orient_sweep produces: checked count and the ascending
 * list of flipped triangles (src/same.py:687-703 relies on that order).  The XY-order and area sweeps
 * shard by calling their _dev forms on offset pointers (dtris + 3*t_begin, outputs + t_begin). */
int same_orient_flags_dev(same_sweep *sweep, const int32_t *dmatch, int64_t t_begin, int64_t t_end,
                          uint8_t *dflag);
int same_orient_from_flags_dev(same_sweep *sweep, const uint8_t *dflag, int64_t *out_checked,
                               int32_t *out_viol_idx, int64_t *out_nviol);
/* dmatch[i] = didx[i*k]: the nearest candidate of every row of a padded candidate list (-1 = none). */
int same_first_candidate_dev(same_ctx *ctx, const int32_t *didx, int64_t rows, int k, int32_t *dmatch);

/* ======================================================================================================================
 * part 3 -- WINDOW path: both sections resident on the device, a batch of windows = two calls
 * ====================================================================================================================== */
/* ---- a13 + the per-window pre-MIP path with the sections resident on the device -------------
 * The reference's window loop (src/same.py:507-593) subsets both frames per window (:293-295: one boolean mask over the
 * whole frame per window) and runs the whole pre-MIP path on the subset.  Here:
 *   same_section      one section's columns uploaded once (XY, the commonCT type columns, cell sizes; cost_f32 != 0 keeps the
 *                     cost operands as float, BASELINE cfg 5), its rows binned ONCE into a grid of cells (rows sorted by cell,
 *                     ascending inside a cell).  same_section_bin(section, x0, y0, cell_w, cell_h) re-bins it on the caller's
 *                     grid: with the window grid's origin (int(x_min), int(y_min)) and cell = gcd(window step, window size)
 *                     every window box of src/same.py:481-488 / :527-542 is a union of cells and its rows need no test at all;
 *                     other boxes test the rows of the cells they cut.  A box covering more than 64 cells falls back to one
 *                     mask over the section.  It waits for stage calls in flight on the section and for the device before it replaces the
 *                     grid (a lock per section); still, re-bin between passes, not between the two calls of a window.  Sections
 *                     may be shared by the windows of several contexts of one device.
 *   same_window       the device state of one window in flight (buffers grow on demand and are reused).  It reads the two sections it was
 *                     staged on until its finish call returns: destroy windows (or stage them elsewhere) before their sections.
 *   same_window_stage: for each of n_windows windows (window i takes boxes[4 i .. 4 i + 3] = {x0,x1,y0,y1}, half-open): the rows of both
 *     sections inside the box (ascending row order), radius / k prune (src/utils.py:709-728), candidate costs (src/same.py:1180-1189),
 *     compaction of the aligned cells that have candidates and of the pair list (src/utils.py:734-742).  out_counts[4 i ..] = {aligned
 *     rows in the box, reference rows in the box, aligned rows kept, pairs}.  Reference cells are not renumbered: pair[1] / match index
 *     the window's reference rows.  Every kernel takes up to EIGHT windows per launch (blockIdx.y = window, their arguments by value):
 *     per group of eight one zeroing launch, at most five kernels and one launch that writes what comes back into the windows' pinned
 *     blocks, reading O(rows of the covered cells); ONE wait for the whole batch.  The windows of a batch belong to one context and
 *     are distinct; n_windows <= SAME_WINDOW_BATCH_MAX.  If any window is refused nothing of the batch counts.
 *   same_window_filter_finish: for each window of the batch, the Delaunay simplices of its kept aligned cells (window i:
 *     simplices[3 simplex_offsets[i] .. 3 simplex_offsets[i+1]), offsets[0] = 0) ->
 *     - filter_triangles_by_radius on the device (src/helpers.py:233-395: classes :300-330, the keep list, the same-type triangles
 *       added back so that every node keeps one :331-340 / :365-389, in the reference's order); the section's type_id codes play
 *       aligned_df["cell_type"].  Simplices must be distinct as vertex rows (Qhull's are): the re-add pass de-duplicates by triangle,
 *       the reference by vertex row.  prefiltered != 0 skips this step: the rows passed ARE the kept triangles, in the reference's
 *       order (a caller's own triangulation; the knife-edge fallback below), and the filter arguments are ignored;
 *     - source signs / weights (src/same.py:1128-1146), greedy MIP start with prefer = rowmin < no_match_penalty * size
 *       (src/init_helpers.py:104-133), lazy-constraint body (src/same.py:645-669), XY-order sweep (src/violationhelper.py:53-117),
 *       area flips (src/same.py:1362-1402).
 *     Outputs are laid end to end in window order: out_match_row / out_point_flag hold kept[0] + kept[1] + ... entries (kept = the
 *     stage call's out_counts[4 i + 2]); out_match_row = SECTION row of the matched reference cell or -1; out_point_flag = per-cell
 *     flag byte: bit 0 the XY-order sweep flags the cell (src/violationhelper.py:100-104), bit 1 the cell is a vertex of a triangle
 *     whose signed area flips (src/same.py:1464-1469); out_stats[8 i ..] = {orientation checked, flipped, XY comparisons, XY
 *     violations, triangles with a violation, area flips, greedy rounds, matched cells}; out_counts[4 i ..] = {kept, added back,
 *     cosines within near_tol of cos_thr, order ties} (prefiltered: {n, 0, 0, order ties}).  When a window's third count is not zero
 *     nothing of that window counts (its slices of the outputs mean nothing, no triangles are left on the device): the caller
 *     re-decides its triangles with the reference's literal arccos (same_amd/triangles.py) and calls again for that window with
 *     prefiltered = 1.  ORDER TIES (ABI 8) count the places where the reference's answer depends on the order in which the
 *     triangulation lists its triangles or their corners, beyond which triangles there are: an edge of the XY-order sweep whose
 *     ends share an x or a y (`<` is not symmetric, src/violationhelper.py:68-75), a signed area or orientation within rounding of
 *     zero (src/same.py:1401, :658), two same-type triangles of one node whose perimeters agree to the rounding of a three-term
 *     sum (src/helpers.py:334-340 keeps the first).  With Qhull's own simplices the count is of no consequence; a caller that
 *     triangulates otherwise (same_delaunay2d: same triangles, another order) asks Qhull for a window whose count is not zero.  The call's
 *     simplices go up in ONE copy; per group of eight windows one zeroing launch, 14 kernels (17 with fp64 costs) and one launch
 *     that writes the answers into the pinned blocks; ONE wait for the batch (more greedy rounds, in batches with a wait each, only
 *     for a window in which a pair could still be taken after the rounds enqueued up front).
 * same_window_fetch copies one array of the window's state to the host; bytes must be the array's exact size.
 * (Since ABI 6 same_window_stage / same_window_filter_finish take batches; same_window_filter and same_window_finish of ABI 5 are gone --
 * the former is the latter's first half, the latter is prefiltered = 1.) */
#define SAME_WINDOW_BATCH_MAX 64
typedef struct same_section same_section;
typedef struct same_window same_window;
enum {
    SAME_WINDOW_ALIGNED_XY = 0,   /* double[kept][2]: XY of the kept aligned cells (the Delaunay input)            */
    SAME_WINDOW_ALIGNED_ROWS = 1, /* int32[kept]: their section rows                                               */
    SAME_WINDOW_ROWS_M = 2,       /* int32[aligned rows in the box]                                                */
    SAME_WINDOW_ROWS_R = 3,       /* int32[reference rows in the box]                                              */
    SAME_WINDOW_PAIRS = 4,        /* int32[pairs][2]: (kept aligned index, reference index in the window)         */
    SAME_WINDOW_COSTS = 5,        /* double[pairs] (float costs widened)                                           */
    SAME_WINDOW_KEPT = 6,         /* int32[kept]: index of each kept aligned cell among the box's aligned rows     */
    SAME_WINDOW_SIGNS = 7,        /* int8[triangles]   (after same_window_filter_finish)                            */
    SAME_WINDOW_WEIGHTS = 8,      /* double[triangles] (after same_window_filter_finish)                            */
    SAME_WINDOW_MATCH = 9,        /* int32[kept]: matched reference index in the window or -1 (after finish)       */
    SAME_WINDOW_TRIANGLES = 10    /* int32[triangles][3]: the kept triangles (after ..._filter_finish)           */
};
int same_section_create(same_ctx *ctx, const double *xy, const double *types, int T, const double *size,
                        const int32_t *type_id /* may be NULL */, int64_t n, int cost_f32, same_section **out);
int same_section_bin(same_section *section, double x0, double y0, double cell_w, double cell_h);
void same_section_destroy(same_section *section);
int same_window_create(same_ctx *ctx, same_window **out);
void same_window_destroy(same_window *window);
int same_window_stage(same_window *const *windows, int n_windows, const same_section *moving, const same_section *ref,
                      const double *boxes, double radius, int k, double dist_ct_coeff, int64_t *out_counts);
int same_window_fetch(same_window *window, int what, void *out, int64_t bytes);
int same_window_filter_finish(same_window *const *windows, int n_windows, const int32_t *simplices, const int64_t *simplex_offsets,
                              int prefiltered, double radius, int angle_enabled, double cos_thr, double near_tol, int ignore_same_type,
                              int ensure_min_triangle_per_node, double no_match_penalty, int32_t *out_match_row,
                              uint8_t *out_point_flag, int64_t *out_stats, int64_t *out_counts);

/* ---- a6 on the window path without the library call --------------------------------------------------------------------------
 * The reference triangulates every window's kept aligned cells with scipy.spatial.Delaunay (Qhull; src/same.py:1023), on the host
 * -- three quarters of a cfg 5 pass.  same_delaunay2d is this library's own triangulator (HOST code, no device, no context, safe to
 * call from many threads at once): a sweep-hull construction with edge flips, every sign it relies on clear of its rounding bound by
 * three orders of magnitude.  It ANSWERS only when the answer is beyond doubt the set of triangles Qhull gives: after the
 * construction every interior edge and hull corner is measured the way Qhull sees it (distance of the fourth point from the lifted
 * triangle's plane in 'Qbb'-scaled paraboloid coordinates, against Qhull's round-off allowance for these coordinates), and when the
 * smallest of these ratios (*out_margin) is not above `guard`, or a sign was in doubt on the way (duplicate, collinear, cocircular
 * points), the return is SAME_EUNSURE and the caller asks Qhull as the reference does.  Measured against scipy 1.15.3 on 6 000 sets
 * (uniform, blobs, clusters, strips; offsets to 3e8): Qhull's triangles differ from the exact Delaunay triangulation only where the
 * margin is below 0.3 (tools/delaunay_margin.py, profiles/r06_delaunay_margin.md: 6 000 sets); same_amd uses guard = 16 (60 x that).
 * xy: n points (x, y) interleaved.  out_tris: room for `cap` triangles (2 n - 5 always suffices); *out_n_tris triangles are written,
 * counter-clockwise, in this function's own order (Qhull's order and corner order are its own: see the ORDER TIES of
 * same_window_filter_finish for what that touches).  SAME_OK | SAME_EUNSURE | SAME_EINVAL (NULL, n < 0 or above 3.5e8, cap too small) |
 * SAME_ENOMEM. */
int same_delaunay2d(const double *xy, int64_t n, int32_t *out_tris, int64_t cap, int64_t *out_n_tris, double guard,
                    double *out_margin /* may be NULL */);

/* ---- f3 on the window path: the window merge where the windows' matches are ------------------------------------------------
 * The reference trims every window's match table to the window's central region (src/same.py:565-582), concatenates the tables and
 * merges them (helpers.merge_window_matches_unique_ref, src/helpers.py:692-815: one row per (aligned, ref) pair -- not violating
 * first, then the smaller window id, then the earlier row --, then one maximum matching of the pairs that are left, rows in the order
 * of the aligned ids).  In a tiled run nearly every pair stands alone (no other row names either of its cells) and is in the merged
 * table as it is.  On the device, per pass over a plan (ABI 7):
 *   same_section_set_codes   codes[row] = rank of the row's cell id among the section's ids, 0 .. n_codes-1 (equal id <=> equal code; the
 *                            merge compares and orders by them); NULL: a row's code is its number (ids ascending by row).
 *   same_merge_acc           the rows of one pass (one per context; grows on demand, reused from pass to pass).
 *   same_merge_acc_begin     a new pass.  expected_rows sizes the arrays up front (an upper bound saves a re-allocation, nothing more).
 *                            Seams, for a plan dealt over ranks (NULL near_start: one rank, no seams): window (plan position) p of this
 *                            rank lies near the central regions near_boxes[4 q .. 4 q + 3] = {x0, x1, y0, y1}, q in [near_start[p],
 *                            near_start[p+1]), of OTHER ranks' windows; a row of p whose aligned cell lies in one of them, or whose
 *                            reference cell lies within `reach` (the prune's radius) of one, may be named by another rank's table too
 *                            and is left to the host.  all_seam != 0: every row is (cell ids that name several rows of a frame).
 *   same_window_collect      after same_window_filter_finish, ENQUEUE ONLY: the matched kept cells of each window whose XY lies in
 *                            trims[4 i .. 4 i + 3] = {x0, x1, y0, y1} (half open) are appended to the accumulator of the windows'
 *                            context, windows in call order, cells ascending: (aligned section row, matched reference section row, flag
 *                            byte, window_ids[i], plan_pos[i], index among the window's kept cells).  Three launches per eight windows.
 *   same_merge_acc_resolve   the pass is over: the accumulators (accs[0] first -- the call runs on its context; the others are waited
 *                            for and laid behind it, in order: plan order when the contexts walked consecutive runs of the plan) ->
 *                            codes, the de-duplication of src/helpers.py:745-753 (merge.hip's kernels, on the device arrays), degrees,
 *                            classes.  out_counts = {rows, rows after the de-duplication, REST rows, rows final already}.  A surviving row
 *                            whose two cells are named by no other surviving row and that is not at a seam is final.  The REST --
 *                            contested or at a seam -- is for the host (same_amd/merge.py: connected components, Hopcroft-Karp, the
 *                            exchange between ranks): same_merge_acc_fetch(accs[0], SAME_MERGE_REST) = one record per row, in the order
 *                            the de-duplication left: int32 {row in the accumulator, aligned code, ref code, window id, plan position,
 *                            index among the window's kept cells}, uint32 flags (bit 0 XY-order flag, bit 1 area-flip flag, bit 2 seam).
 *   same_merge_acc_finish    winner_rows = the REST rows (accumulator row numbers) the host's matching kept -> the merged table's rows
 *                            in the order of the aligned codes (src/helpers.py:799-808), their number in *out_n_final; they stay on the
 *                            device (same_merge_acc_columns reads them there) unless fetched: same_merge_acc_fetch(.., SAME_MERGE_FINAL) =
 *                            int32 {aligned section row, reference section row, index among the window's kept cells, window id, plan
 *                            position}, uint32 flags (bits 0, 1 as above) per row.  The host gathers the columns of exactly these rows.
 *   same_merge_acc_load      rows from the HOST instead of from windows (the seam rows of every rank after their exchange: the common step
 *                            of a merge dealt over ranks, same_amd/merge.py): the accumulator holds exactly these n rows, their "section
 *                            rows" ARE codes (a_code < n_codes_a, r_code < n_codes_r), ties of the de-duplication go to the earlier row;
 *                            same_merge_acc_resolve then takes this one accumulator with NULL sections. */
typedef struct same_merge_acc same_merge_acc;
#define SAME_MERGE_REST 0
#define SAME_MERGE_FINAL 1
#define SAME_MERGE_REST_BYTES 28
#define SAME_MERGE_FINAL_BYTES 24
int same_section_set_codes(same_section *section, const int32_t *codes /* may be NULL */, int64_t n_codes);
int same_merge_acc_create(same_ctx *ctx, same_merge_acc **out);
void same_merge_acc_destroy(same_merge_acc *acc);
int same_merge_acc_begin(same_merge_acc *acc, int64_t expected_rows, int n_pos, const int32_t *near_start /* may be NULL */,
                         const double *near_boxes, double reach, int all_seam);
int same_window_collect(same_window *const *windows, int n_windows, same_merge_acc *acc, const double *trims,
                        const int32_t *window_ids, const int32_t *plan_pos);
int same_merge_acc_load(same_merge_acc *acc, const int32_t *a_code, const int32_t *r_code, const uint8_t *flags, const int32_t *window_ids,
                        const int32_t *pos, const int32_t *cidx, int64_t n, int64_t n_codes_a, int64_t n_codes_r);
int same_merge_acc_resolve(same_merge_acc *const *accs, int n_accs, const same_section *moving /* NULL after ..._load */,
                           const same_section *ref /* NULL after ..._load */, int64_t *out_counts /* [4] */);
int same_merge_acc_finish(same_merge_acc *acc, const int32_t *winner_rows, int64_t n_winners, int64_t *out_n_final);
/* Without the merge: every accumulated row is a final row, in the order the windows were collected (src/same.py:583-590: the window
 * tables concatenated) -- for the same fetch / same_merge_acc_columns calls.  accs as for same_merge_acc_resolve. */
int same_merge_acc_plain(same_merge_acc *const *accs, int n_accs, int64_t *out_n_rows);
int same_merge_acc_fetch(same_merge_acc *acc, int what, void *out, int64_t bytes);
/* The merged table's columns, written by the DEVICE (after same_merge_acc_finish; enqueue only -- same_ctx_sync waits) into out_host,
 * column after column, n_final entries each, in the order of the final rows:
 *   8-byte columns: the moving section's T type columns (src/same.py:1264-1278 copies them from aligned_df), its X and Y, the reference
 *   section's X and Y (ref_X, ref_Y); then n_extra_mov columns gathered by moving row and n_extra_ref columns gathered by reference row
 *   from the caller's DEVICE arrays of 8-byte values (cell ids, sizes: copied as bit patterns, whatever their type; at most 4 each); then
 *   aligned_idx (the cell's index among its window's kept cells), window_id and the window's plan position as int64;
 *   byte columns: triangle_violation (the area-flip flag, src/same.py:1464-1469) and the XY-order flag, 0 / 1.
 * out_host holds (T + 4 + n_extra_mov + n_extra_ref + 3) * 8 * n_final + 2 * n_final bytes and must be host memory the device can write:
 * same_host_alloc (page-locked, hipHostMalloc; same_host_free returns it).  The float columns are the bytes the caller uploaded with
 * same_section_create. */
int same_merge_acc_columns(same_merge_acc *acc, const same_section *moving, const same_section *ref, const void *const *extra_moving,
                           int n_extra_moving, const void *const *extra_ref, int n_extra_ref, void *out_host, int64_t n_final);
int same_host_alloc(same_ctx *ctx, size_t bytes, void **out_ptr);
int same_host_free(same_ctx *ctx, void *ptr);

/* ======================================================================================================================
 * part 4 -- COMM: one communicator per context, RCCL over xGMI
 * ====================================================================================================================== */
/* ---- multi-GPU: RCCL all-gather of the pruned candidate lists -------------
 * One process per GPU.  Rank 0 calls same_comm_unique_id and hands the 128 bytes to the
 * other ranks by any host channel; all ranks then call same_comm_init.  same_allgather_dev
 * gathers equal-sized device blocks (send_bytes each) into recv (nranks*send_bytes) on the
 * context's stream. */
#define SAME_UNIQUE_ID_BYTES 128
int same_comm_unique_id(char out_id[SAME_UNIQUE_ID_BYTES]);
int same_comm_init(same_ctx *ctx, int nranks, int rank, const char id[SAME_UNIQUE_ID_BYTES]);
int same_comm_destroy(same_ctx *ctx);
int same_allgather_dev(same_ctx *ctx, const void *dsend, void *drecv, size_t send_bytes);
/* Overlapped form: the gather runs on a second stream of the context, ordered after all compute queued
 * so far, and does not block later compute (e.g. the next row block's dense build).  same_comm_wait
 * orders the compute stream after every gather issued so far; call it before overwriting the send
 * buffers or reading the gathered ones.  same_ctx_sync waits for both streams. */
int same_allgather_dev_async(same_ctx *ctx, const void *dsend, void *drecv, size_t send_bytes);
int same_comm_wait(same_ctx *ctx);
/* In-place all-reduce of count elements on the context's stream (sweep counters: U64 SUM; point flags:
 * U8 MAX = logical OR; timings: F64 MAX). */
#define SAME_DT_U8 0
#define SAME_DT_I32 1
#define SAME_DT_U64 2
#define SAME_DT_F64 3
#define SAME_OP_SUM 0
#define SAME_OP_MAX 1
#define SAME_OP_MIN 2
int same_allreduce_dev(same_ctx *ctx, void *dbuf, size_t count, int dtype, int op);
/* collectives issued between these two calls go to RCCL as one group (ncclGroupStart / ncclGroupEnd): one fused launch */
int same_comm_group_start(same_ctx *ctx);
int same_comm_group_end(same_ctx *ctx);
/* What the communicator itself reports -- ncclCommCount (0 = no communicator) and ncclCommUserRank, not the arguments
 * same_comm_init was given -- and the RCCL version the library is running against.  Any pointer may be NULL. */
int same_comm_info(same_ctx *ctx, int *out_nranks, int *out_rank, int *out_rccl_version);
/* ncclCommCuDevice of the communicator (-1 = none) */
int same_comm_device(same_ctx *ctx, int *out_device);
#ifdef __cplusplus
}
#endif
#endif /* SAME_HIP_H */
