"""Array-level wrappers over the C ABI (one function per libsame_hip entry point).

Inputs are NumPy arrays (converted to the ABI's dtypes / C order); outputs are fresh NumPy
arrays.  These are the only callers of the library besides bench.py; the modules that mirror
the reference's functions (knn.py, cost.py, triangles.py, sweeps.py, init_helpers.py) are
written on top of them.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import as_c, c_i64, default_context

F64, F32, I32, I8, U8, I64 = np.float64, np.float32, np.int32, np.int8, np.uint8, np.int64


def _ctx(ctx):
    return ctx if ctx is not None else default_context()


def _tris(triangles):
    t = np.asarray(triangles)
    return as_c(t.reshape(-1, 3) if t.size else np.zeros((0, 3)), I32)


def pair_cost(A, R, axy, rxy, pairs, w, dtype=F64, ctx=None):
    """Costs of a pair list.  dtype float32 = the config-5 variant (operands converted to float, float arithmetic)."""
    ctx = _ctx(ctx)
    dt = np.dtype(dtype)
    assert dt in (np.dtype(F64), np.dtype(F32))
    A, R = as_c(A, dt), as_c(R, dt)
    axy, rxy = as_c(axy, dt).reshape(-1, 2), as_c(rxy, dt).reshape(-1, 2)
    pairs = as_c(pairs, I32).reshape(-1, 2)
    T = A.shape[1] if A.ndim == 2 else 0
    out = np.empty(len(pairs), dt)
    fn = ctx.lib.same_pair_cost_f64 if dt == np.dtype(F64) else ctx.lib.same_pair_cost_f32
    with ctx.lock:
        ctx.check(fn(ctx.handle, A.ctypes.data, R.ctypes.data, len(axy), len(rxy), T, axy.ctypes.data,
                     rxy.ctypes.data, pairs.ctypes.data, len(pairs), float(w), out.ctypes.data), "same_pair_cost")
    return out


def dense_cost(A, R, axy, rxy, w, row_begin=0, row_end=None, dtype=F64, ctx=None):
    ctx = _ctx(ctx)
    dt = np.dtype(dtype)
    assert dt in (np.dtype(F64), np.dtype(F32))
    A, R = as_c(A, dt), as_c(R, dt)
    axy, rxy = as_c(axy, dt).reshape(-1, 2), as_c(rxy, dt).reshape(-1, 2)
    n_m, n_r = len(axy), len(rxy)
    row_end = n_m if row_end is None else int(row_end)
    T = A.shape[1] if A.ndim == 2 else 0
    out = np.empty((max(row_end - row_begin, 0), n_r), dt)
    fn = ctx.lib.same_dense_cost_f64 if dt == np.dtype(F64) else ctx.lib.same_dense_cost_f32
    with ctx.lock:
        ctx.check(fn(ctx.handle, A.ctypes.data, R.ctypes.data, n_m, n_r, T, axy.ctypes.data, rxy.ctypes.data, int(row_begin),
                     row_end, float(w), out.ctypes.data, n_r), "same_dense_cost")
    return out


def quantize_types(A, R):
    """Grid for the opt-in fixed-point dense build: -> (offset, log2_scale) such that every value of A and R is >= offset and
    every row-pair sum of absolute differences, on the grid, fits 32 bits.  sum_t |a_t - r_t| <= sum_t (a_t - offset) +
    sum_t (r_t - offset), so the largest row sum of each matrix bounds it; each grid value may round up by half a step."""
    A, R = np.asarray(A, dtype=F64), np.asarray(R, dtype=F64)
    if A.size == 0 or R.size == 0 or A.shape[1] == 0:
        return 0.0, 0
    if not (np.isfinite(A).all() and np.isfinite(R).all()):
        raise ValueError("the fixed-point build needs finite type values")
    offset = float(min(A.min(), R.min()))
    bound = float((A - offset).sum(axis=1).max() + (R - offset).sum(axis=1).max())
    T = A.shape[1]
    if bound <= 0.0:
        return offset, 0
    log2_scale = int(np.floor(np.log2((2.0 ** 32 - 1.0 - T) / bound)))
    while (bound * 2.0 ** log2_scale + T) >= 2.0 ** 32:   # guard the floor() against the last ulp
        log2_scale -= 1
    return offset, min(log2_scale, 1000)


def dense_cost_q32(A, R, axy, rxy, w, row_begin=0, row_end=None, grid=None, rel_tol=1e-6, ctx=None):
    """Opt-in fixed-point dense build (include/same_hip.h: same_dense_cost_q32_dev) -> (costs (rows, n_r) f64, T * 2^-s absolute
    error bound on the type sum).  Every output is also within `rel_tol` (relative) of `dense_cost`'s: sums too small for the
    grid are recomputed in fp64 in the kernel.  NOT the reference's arithmetic; `dense_cost` is."""
    ctx = _ctx(ctx)
    A, R = as_c(A, F64), as_c(R, F64)
    axy, rxy = as_c(axy, F64).reshape(-1, 2), as_c(rxy, F64).reshape(-1, 2)
    n_m, n_r = len(axy), len(rxy)
    row_end = n_m if row_end is None else int(row_end)
    rows = max(row_end - int(row_begin), 0)
    T = A.shape[1] if A.ndim == 2 else 0
    offset, log2_scale = quantize_types(A, R) if grid is None else grid
    scale = float(2.0 ** log2_scale)
    out = np.empty((rows, n_r), F64)
    if rows == 0 or n_r == 0:
        return out, T / scale
    ld = (n_r + 1) & ~1
    with ctx.lock:
        dA, dR, dax, drx = ctx.to_device(A), ctx.to_device(R), ctx.to_device(axy), ctx.to_device(rxy)
        # a plain block: the tile goes back over PCIe, which is what bounds this host-buffer form (HBM placement, same_dev_alloc_spread,
        # only matters for blocks that stay resident -- callers of the _dev entry points choose it themselves)
        dAq, dRq, dout = ctx.alloc(max(A.size, 1) * 4), ctx.alloc(max(R.size, 1) * 4), ctx.alloc(rows * ld * 8)
        ctx.check(ctx.lib.same_quantize_u32_dev(ctx.handle, dA.ptr, A.size, offset, scale, dAq.ptr), "same_quantize_u32_dev")
        ctx.check(ctx.lib.same_quantize_u32_dev(ctx.handle, dR.ptr, R.size, offset, scale, dRq.ptr), "same_quantize_u32_dev")
        ctx.check(ctx.lib.same_dense_cost_q32_dev(ctx.handle, dAq.ptr, dRq.ptr, dA.ptr, dR.ptr, T, dax.ptr, drx.ptr, n_r, int(row_begin),
                                                  row_end, float(w), 1.0 / scale, float(rel_tol), dout.ptr, ld), "same_dense_cost_q32_dev")
        full = dout.download((rows, ld), F64)
    out[:] = full[:, :n_r]
    return out, T / scale


def knn_prune(axy, rxy, radius, knn, row_begin=0, row_end=None, want_d2=True, ctx=None):
    """-> (idx (rows,k) int32 -1 padded, d2 (rows,k) f64 +inf padded or None, cnt (rows,) int32)."""
    ctx = _ctx(ctx)
    axy, rxy = as_c(axy, F64).reshape(-1, 2), as_c(rxy, F64).reshape(-1, 2)
    n_m, n_r = len(axy), len(rxy)
    row_end = n_m if row_end is None else int(row_end)
    rows = max(row_end - row_begin, 0)
    idx = np.empty((rows, int(knn)), I32)
    d2 = np.empty((rows, int(knn)), F64) if want_d2 else None
    cnt = np.empty(rows, I32)
    with ctx.lock:
        ctx.check(ctx.lib.same_knn_prune(ctx.handle, axy.ctypes.data, n_m, rxy.ctypes.data, n_r, int(row_begin), row_end,
                                         float(radius), int(knn), idx.ctypes.data, _lib._ptr(d2), cnt.ctypes.data),
                  "same_knn_prune")
    return idx, d2, cnt


def tri_classify(xy, triangles, radius, angle_enabled, cos_thr, type_id=None, ctx=None):
    ctx = _ctx(ctx)
    xy = as_c(xy, F64).reshape(-1, 2)
    tris = _tris(triangles)
    Tr = len(tris)
    tid = None if type_id is None else as_c(type_id, I32)
    cls, perim, maxcos = np.empty(Tr, U8), np.empty(Tr, F64), np.empty(Tr, F64)
    with ctx.lock:
        ctx.check(ctx.lib.same_tri_classify(ctx.handle, xy.ctypes.data, len(xy), tris.ctypes.data, Tr, float(radius),
                                            int(angle_enabled), float(cos_thr), _lib._ptr(tid), cls.ctypes.data,
                                            perim.ctypes.data, maxcos.ctypes.data), "same_tri_classify")
    return cls, perim, maxcos


def tri_sign_weight(xy, size, triangles, ctx=None):
    ctx = _ctx(ctx)
    xy = as_c(xy, F64).reshape(-1, 2)
    tris = _tris(triangles)
    Tr = len(tris)
    size = None if size is None else as_c(size, F64)
    sign = np.empty(Tr, I8)
    weight = None if size is None else np.empty(Tr, F64)
    with ctx.lock:
        ctx.check(ctx.lib.same_tri_sign_weight(ctx.handle, xy.ctypes.data, _lib._ptr(size), len(xy), tris.ctypes.data, Tr,
                                               sign.ctypes.data, _lib._ptr(weight)), "same_tri_sign_weight")
    return sign, weight


class BoundSweep:
    """Resident state of the lazy-constraint sweep (model._* of src/same.py:1153-1158): one same_sweep handle that owns
    its device blocks.  Any number of these may live on one context; calls are serialised on the context's lock (the
    solver may call back from a thread other than the one that built the model)."""

    def __init__(self, triangles, src_sign, rxy, n_aligned, pairs=None, ctx=None):
        self.ctx = c = _ctx(ctx)
        tris = _tris(triangles)
        src_sign = as_c(np.asarray(src_sign), I8)
        rxy = as_c(rxy, F64).reshape(-1, 2)
        self.n_m = int(n_aligned)
        pairs = None if pairs is None else as_c(pairs, I32).reshape(-1, 2)
        self.P = 0 if pairs is None else len(pairs)
        self.has_pairs = pairs is not None
        self.Tr, self.n_r = len(tris), len(rxy)
        assert len(src_sign) == self.Tr
        h = ctypes.c_void_p()
        with c.lock:
            c.check(c.lib.same_sweep_bind(c.handle, tris.ctypes.data, self.Tr, src_sign.ctypes.data, rxy.ctypes.data,
                                          self.n_r, self.n_m, _lib._ptr(pairs), self.P, ctypes.byref(h)), "same_sweep_bind")
        self.handle = h.value

    def sweep_match(self, match, want_flag=False):
        """match (n_aligned,) int32, -1 unmatched -> (checked, violating idx ascending[, flag])."""
        c = self.ctx
        match = as_c(match, I32)
        checked, nviol = c_i64(0), c_i64(0)
        flag = np.empty(self.Tr, U8) if want_flag else None
        viol = np.empty(max(self.Tr, 1), I32)   # per call: concurrent callers never share an output buffer
        with c.lock:
            c.check(c.lib.same_orient_sweep(self.handle, match.ctypes.data, len(match), ctypes.byref(checked), viol.ctypes.data,
                                            ctypes.byref(nviol), _lib._ptr(flag)), "same_orient_sweep")
        viol = viol[: nviol.value].copy()
        return (checked.value, viol, flag) if want_flag else (checked.value, viol)

    def sweep_x(self, x_vals):
        """x_vals (P,) -> (checked, violating idx ascending, match, pair_idx)."""
        c = self.ctx
        assert self.has_pairs
        x = as_c(x_vals, F64)
        checked, nviol = c_i64(0), c_i64(0)
        match, pidx = np.empty(self.n_m, I32), np.empty(self.n_m, I32)
        viol = np.empty(max(self.Tr, 1), I32)
        with c.lock:
            c.check(c.lib.same_orient_sweep_x(self.handle, x.ctypes.data, len(x), ctypes.byref(checked), viol.ctypes.data,
                                              ctypes.byref(nviol), None, match.ctypes.data, pidx.ctypes.data),
                    "same_orient_sweep_x")
        return checked.value, viol[: nviol.value].copy(), match, pidx

    def close(self):
        if getattr(self, "handle", None) and self.ctx.handle:
            with self.ctx.lock:
                self.ctx.lib.same_sweep_unbind(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def xyorder_sweep(axy, rxy, triangles, match, ctx=None):
    ctx = _ctx(ctx)
    axy, rxy = as_c(axy, F64).reshape(-1, 2), as_c(rxy, F64).reshape(-1, 2)
    tris = _tris(triangles)
    match = as_c(match, I32)
    Tr, n_m = len(tris), len(axy)
    assert len(match) == n_m
    edge, tflag, pflag = np.empty((Tr, 3), U8), np.empty(Tr, U8), np.empty(n_m, U8)
    counts = np.zeros(3, I64)
    with ctx.lock:
        ctx.check(ctx.lib.same_xyorder_sweep(ctx.handle, axy.ctypes.data, n_m, rxy.ctypes.data, len(rxy), tris.ctypes.data, Tr,
                                             match.ctypes.data, edge.ctypes.data, tflag.ctypes.data, pflag.ctypes.data,
                                             counts.ctypes.data), "same_xyorder_sweep")
    return edge, tflag, pflag, counts


def area_flip(axy, rxy, triangles, match, ctx=None):
    ctx = _ctx(ctx)
    axy, rxy = as_c(axy, F64).reshape(-1, 2), as_c(rxy, F64).reshape(-1, 2)
    tris = _tris(triangles)
    match = as_c(match, I32)
    Tr = len(tris)
    assert len(match) == len(axy)
    before, after = np.empty(Tr, F64), np.empty(Tr, F64)
    m3, fl = np.empty((Tr, 3), U8), np.empty(Tr, U8)
    with ctx.lock:
        ctx.check(ctx.lib.same_area_flip(ctx.handle, axy.ctypes.data, len(axy), rxy.ctypes.data, len(rxy), tris.ctypes.data, Tr,
                                         match.ctypes.data, before.ctypes.data, after.ctypes.data, m3.ctypes.data,
                                         fl.ctypes.data), "same_area_flip")
    return before, after, m3, fl


def pair_rowmin(pairs, costs, n_aligned, ctx=None):
    ctx = _ctx(ctx)
    pairs, costs = as_c(pairs, I32).reshape(-1, 2), as_c(costs, F64)
    out = np.empty(int(n_aligned), F64)
    with ctx.lock:
        ctx.check(ctx.lib.same_pair_rowmin(ctx.handle, pairs.ctypes.data, costs.ctypes.data, len(pairs), int(n_aligned),
                                           out.ctypes.data), "same_pair_rowmin")
    return out


def assign_matrix(pairs, costs, unmatched, n_aligned, n_ref, big_m, ctx=None):
    ctx = _ctx(ctx)
    pairs, costs, unmatched = as_c(pairs, I32).reshape(-1, 2), as_c(costs, F64), as_c(unmatched, F64)
    out = np.empty((int(n_aligned), int(n_ref) + int(n_aligned)), F64)
    with ctx.lock:
        ctx.check(ctx.lib.same_assign_matrix(ctx.handle, pairs.ctypes.data, costs.ctypes.data, len(pairs), unmatched.ctypes.data,
                                             int(n_aligned), int(n_ref), float(big_m), out.ctypes.data), "same_assign_matrix")
    return out


def eager_signs(rxy, triangles, cand, ctx=None):
    ctx = _ctx(ctx)
    rxy = as_c(rxy, F64).reshape(-1, 2)
    tris = _tris(triangles)
    cand = as_c(cand, I32)
    n_m, k = cand.shape
    out = np.empty((len(tris), k, k, k), I8)
    with ctx.lock:
        ctx.check(ctx.lib.same_eager_signs(ctx.handle, rxy.ctypes.data, len(rxy), tris.ctypes.data, len(tris), cand.ctypes.data,
                                           n_m, k, out.ctypes.data), "same_eager_signs")
    return out


def window_count(xy, boxes, want_mask=False, ctx=None):
    ctx = _ctx(ctx)
    xy = as_c(xy, F64).reshape(-1, 2)
    boxes = as_c(boxes, F64).reshape(-1, 4)
    counts = np.zeros(len(boxes), I64)
    mask = np.empty((len(boxes), len(xy)), U8) if want_mask else None
    with ctx.lock:
        ctx.check(ctx.lib.same_window_count(ctx.handle, xy.ctypes.data, len(xy), boxes.ctypes.data, len(boxes),
                                            counts.ctypes.data, _lib._ptr(mask)), "same_window_count")
    return (counts, mask.astype(bool)) if want_mask else counts


def greedy_match(pairs, costs, n_aligned, n_ref, prefer, ctx=None):
    """Device form of the greedy scan (src/init_helpers.py:109-133) -> (match_pair (n_aligned,) int32, rounds)."""
    ctx = _ctx(ctx)
    pairs, costs = as_c(pairs, I32).reshape(-1, 2), as_c(costs, F64)
    prefer = as_c(prefer, U8)
    assert len(prefer) == int(n_aligned) and len(costs) == len(pairs)
    out = np.empty(int(n_aligned), I32)
    rounds = ctypes.c_int(0)
    with ctx.lock:
        ctx.check(ctx.lib.same_greedy_match(ctx.handle, pairs.ctypes.data, costs.ctypes.data, len(pairs), int(n_aligned),
                                            int(n_ref), prefer.ctypes.data, out.ctypes.data, ctypes.byref(rounds)),
                  "same_greedy_match")
    return out, rounds.value


def tri_flip_stats(axy, mapped_xy, matched, triangles, type_id=None, ctx=None):
    """-> (tri_flag (Tr,) uint8 [bit0 matched, bit1 same type, bit2 flipped], node_tri, node_flip (n,) uint32)."""
    ctx = _ctx(ctx)
    axy, mxy = as_c(axy, F64).reshape(-1, 2), as_c(mapped_xy, F64).reshape(-1, 2)
    matched = as_c(matched, U8)
    tris = _tris(triangles)
    tid = None if type_id is None else as_c(type_id, I32)
    n, Tr = len(axy), len(tris)
    assert len(mxy) == n and len(matched) == n
    flag, nt, nf = np.empty(Tr, U8), np.empty(n, np.uint32), np.empty(n, np.uint32)
    with ctx.lock:
        ctx.check(ctx.lib.same_tri_flip_stats(ctx.handle, axy.ctypes.data, mxy.ctypes.data, matched.ctypes.data, n, _lib._ptr(tid),
                                              tris.ctypes.data, Tr, flag.ctypes.data, nt.ctypes.data, nf.ctypes.data),
                  "same_tri_flip_stats")
    return flag, nt, nf


def collapse_candidates(xy, triangles, r_max, angle_enabled, cos_thr, type_id, size, max_size, ctx=None):
    """One collapse iteration's per-triangle work -> (flag [bit0 valid, bit1 collapsible], perimeter, total size)."""
    ctx = _ctx(ctx)
    xy = as_c(xy, F64).reshape(-1, 2)
    tris = _tris(triangles)
    type_id, size = as_c(type_id, I32), as_c(size, F64)
    Tr = len(tris)
    flag, perim, total = np.empty(Tr, U8), np.empty(Tr, F64), np.empty(Tr, F64)
    with ctx.lock:
        ctx.check(ctx.lib.same_collapse_candidates(ctx.handle, xy.ctypes.data, len(xy), tris.ctypes.data, Tr,
                                                   0 if r_max is None else 1, 0.0 if r_max is None else float(r_max),
                                                   int(angle_enabled), float(cos_thr), type_id.ctypes.data, size.ctypes.data,
                                                   float(max_size), flag.ctypes.data, perim.ctypes.data, total.ctypes.data),
                  "same_collapse_candidates")
    return flag, perim, total


def greedy_disjoint(items, keys, n_nodes, ctx=None):
    """Vertex-disjoint greedy selection in (key, item index) order -> (selected (M,) bool, rounds)."""
    ctx = _ctx(ctx)
    items, keys = as_c(items, I32).reshape(-1, 3), as_c(keys, F64)
    sel = np.zeros(len(items), U8)
    rounds = ctypes.c_int(0)
    with ctx.lock:
        ctx.check(ctx.lib.same_greedy_disjoint(ctx.handle, items.ctypes.data, keys.ctypes.data, len(items), int(n_nodes),
                                               sel.ctypes.data, ctypes.byref(rounds)), "same_greedy_disjoint")
    return sel.astype(bool), rounds.value


def batched_assign(a_off, r_off, axy, rxy, ctx=None):
    """One small optimal assignment per problem (CSR member lists) -> int32 local ref index per aligned member."""
    ctx = _ctx(ctx)
    a_off, r_off = as_c(a_off, np.int64), as_c(r_off, np.int64)
    if len(a_off) != len(r_off) or len(a_off) < 1:
        raise ValueError("a_off and r_off must both hold n_prob + 1 offsets")
    axy, rxy = as_c(axy, F64).reshape(-1, 2), as_c(rxy, F64).reshape(-1, 2)
    if len(axy) != a_off[-1] or len(rxy) != r_off[-1]:
        raise ValueError("coordinate arrays do not match the last offsets")
    out = np.full(len(axy), -1, I32)
    with ctx.lock:
        ctx.check(ctx.lib.same_batched_assign(ctx.handle, len(a_off) - 1, a_off.ctypes.data, r_off.ctypes.data, axy.ctypes.data,
                                              rxy.ctypes.data, out.ctypes.data), "same_batched_assign")
    return out


def merge_dedup(viol, window_id, aligned_code, ref_code, ctx=None):
    """De-duplication step of the window merge (src/helpers.py:745-753) -> int32 indices of the surviving rows, in the order of
    the reference's frame after sort_values(['filtered_violation', 'window_id'], mergesort) + drop_duplicates(keep='first')."""
    ctx = _ctx(ctx)
    viol = as_c(np.asarray(viol).astype(bool), np.uint8)
    w, a, r = as_c(window_id, I32), as_c(aligned_code, I32), as_c(ref_code, I32)
    n = len(viol)
    if not (len(w) == len(a) == len(r) == n):
        raise ValueError("viol, window_id, aligned_code and ref_code must have one entry per row")
    out = np.empty(max(n, 1), I32)
    m = ctypes.c_int64(0)
    with ctx.lock:
        ctx.check(ctx.lib.same_merge_dedup(ctx.handle, viol.ctypes.data, w.ctypes.data, a.ctypes.data, r.ctypes.data, n, out.ctypes.data,
                                           ctypes.byref(m)), "same_merge_dedup")
    return out[: m.value].copy()
