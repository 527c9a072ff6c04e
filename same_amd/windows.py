"""Sliding-window tiler: the grid, merge-right / merge-down and central-trim rules of
sliding_window_matching (src/same.py:481-488, :509-582) as a plan of independent windows.

Windows are the shard unit for very large sections (BASELINE config 5).  Cell counts of every
box the merge logic can ask about (base, right-merged, down-merged, both) come from one
batched device pass (csrc/sweep.hip window_count_kernel) instead of one pandas boolean mask
per window; the sequential walk that decides merges is then pure host lookups."""
import os

import numpy as np

from . import ops


def window_grid(ref_xy, mov_xy, window_size, overlap):
    """src/same.py:481-488.  The reference takes the extent with pandas' min / max, which skip NaN (rows without coordinates are in no
    window; they do not stop the job): nanmin / nanmax here.  Infinite coordinates reach int() as they do there (OverflowError)."""
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)          # an all-NaN column: NaN, and int(NaN) raises as in the reference
        x_min = min(np.nanmin(ref_xy[:, 0]), np.nanmin(mov_xy[:, 0]))
        x_max = max(np.nanmax(ref_xy[:, 0]), np.nanmax(mov_xy[:, 0]))
        y_min = min(np.nanmin(ref_xy[:, 1]), np.nanmin(mov_xy[:, 1]))
        y_max = max(np.nanmax(ref_xy[:, 1]), np.nanmax(mov_xy[:, 1]))
    step = window_size - overlap
    xs = list(range(int(x_min), int(x_max), step))
    ys = list(range(int(y_min), int(y_max), step))
    return xs, ys, (x_min, x_max, y_min, y_max)


def _candidate_boxes(xs, ys, ws):
    """All boxes the walk may query: variant 0 base, 1 right-merged, 2 down-merged, 3 both."""
    boxes = np.empty((len(xs), len(ys), 4, 4), np.float64)
    for i, x in enumerate(xs):
        xr = xs[i + 1] + ws if i + 1 < len(xs) else x + ws
        for j, y in enumerate(ys):
            yd = ys[j + 1] + ws if j + 1 < len(ys) else y + ws
            boxes[i, j, 0] = (x, x + ws, y, y + ws)
            boxes[i, j, 1] = (x, xr, y, y + ws)
            boxes[i, j, 2] = (x, x + ws, y, yd)
            boxes[i, j, 3] = (x, xr, y, yd)
    return boxes


def window_plan(ref_xy, mov_xy, window_size, overlap, min_cells, skip_ids=(), ctx=None):
    """The windows the reference's double while-loop would run, in order.

    Each entry: i0/j0 (grid cell where the window starts), i/j (after merges), grid_id
    (= len(xs)*j0+i0, what the resume check compares), window_id (= len(xs)*j+i, what is stored),
    box (x0,x1,y0,y1 half-open), trim (central region kept), n_ref, n_mov."""
    ref_xy = np.ascontiguousarray(ref_xy, dtype=np.float64).reshape(-1, 2)
    mov_xy = np.ascontiguousarray(mov_xy, dtype=np.float64).reshape(-1, 2)
    xs, ys, (x_min, x_max, y_min, y_max) = window_grid(ref_xy, mov_xy, window_size, overlap)
    if not xs or not ys:
        return []
    boxes = _candidate_boxes(xs, ys, window_size)
    flat = boxes.reshape(-1, 4)
    n_ref = ops.window_count(ref_xy, flat, ctx=ctx).reshape(boxes.shape[:3])
    n_mov = ops.window_count(mov_xy, flat, ctx=ctx).reshape(boxes.shape[:3])
    skip_ids = set(skip_ids)

    plan = []
    i = 0
    while i < len(xs):
        j = 0
        while j < len(ys):
            if len(xs) * j + i in skip_ids:
                j += 1
                continue
            i0, j0 = i, j
            x, y = xs[i], ys[j]
            variant = 0
            nr, nm = n_ref[i0, j0, 0], n_mov[i0, j0, 0]
            if nr < min_cells or nm < min_cells:
                if i + 1 < len(xs):
                    variant = 1
                    nr, nm = n_ref[i0, j0, 1], n_mov[i0, j0, 1]
                    if nr >= min_cells and nm >= min_cells:
                        i += 1
                if (nr < min_cells or nm < min_cells) and j + 1 < len(ys):
                    variant |= 2
                    nr, nm = n_ref[i0, j0, variant], n_mov[i0, j0, variant]
                    if nr >= min_cells and nm >= min_cells:
                        j += 1
            if nr >= min_cells and nm >= min_cells:
                x0, x1, y0, y1 = (float(v) for v in boxes[i0, j0, variant])
                left, right = x == int(x_min), x1 >= int(x_max)
                top, bottom = y == int(y_min), y1 >= int(y_max)
                trim = (x0 if left else x0 + overlap / 2, x1 if right else x1 - overlap / 2,
                        y0 if top else y0 + overlap / 2, y1 if bottom else y1 - overlap / 2)
                plan.append({"i0": i0, "j0": j0, "i": i, "j": j, "grid_id": len(xs) * j0 + i0,
                             "window_id": len(xs) * j + i, "box": (x0, x1, y0, y1), "trim": trim,
                             "n_ref": int(nr), "n_mov": int(nm)})
            j += 1
        i += 1
    return plan


def assign_windows(plan, n_ranks):
    """Round-robin windows over ranks, heaviest first (windows are independent: no collective)."""
    order = sorted(range(len(plan)), key=lambda w: -(plan[w]["n_ref"] * plan[w]["n_mov"]))
    shards = [[] for _ in range(n_ranks)]
    for pos, w in enumerate(order):
        shards[pos % n_ranks].append(w)
    return [sorted(s) for s in shards]


def assign_window_blocks(plan, n_ranks):
    """Contiguous runs of the plan per rank, balanced by aligned cells (what a window costs: its triangulation and its pairs).
    The plan walks the window grid column by column (src/same.py:509-511), so a run is a strip of whole columns plus two partial ones:
    most overlaps between windows are then overlaps between windows of ONE rank, and the window merge (src/helpers.py:692-815) only
    has the strips' borders to settle between ranks (merge.seam_rows).  The ranks' tables laid end to end are in plan order."""
    weight = np.array([max(1, w["n_mov"]) for w in plan], dtype=np.float64)
    shards = [[] for _ in range(n_ranks)]
    if len(plan):
        middle = (np.cumsum(weight) - weight / 2) / weight.sum()              # where a window sits in the job's work, 0..1
        for w, q in enumerate(np.minimum((middle * n_ranks).astype(np.int64), n_ranks - 1).tolist()):
            shards[q].append(w)
    return shards


WINDOW_DEALS = ("block", "round_robin")


def deal_windows(plan, n_ranks, deal="block"):
    """-> owner[w] = the rank that runs window w of the plan: 'block' (assign_window_blocks) or 'round_robin' (assign_windows)."""
    if deal not in WINDOW_DEALS:
        raise ValueError(f"window deal must be one of {WINDOW_DEALS}, got {deal!r}")
    owner = np.zeros(len(plan), np.int32)
    for q, share in enumerate((assign_window_blocks if deal == "block" else assign_windows)(plan, n_ranks)):
        owner[share] = q
    return owner


class GridRows:
    """Row indices of the points inside half-open boxes [x0, x1) x [y0, y1), for many boxes of one point set: the points are
    binned once into a uniform grid (counting sort by cell), a box gathers the cells it touches (one contiguous run per grid
    row) and only those points are tested exactly.  Indices come back ascending -- `np.flatnonzero` of the reference's four
    comparisons (src/same.py:293-295); NaN / infinite coordinates fall outside every box either way."""

    GRID = 256

    def __init__(self, x, y):
        self.x, self.y = x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
        finite = np.isfinite(x) & np.isfinite(y)
        ok = None if finite.all() else np.flatnonzero(finite)
        xo, yo = (x, y) if ok is None else (x[ok], y[ok])
        self.nx = self.ny = 1
        self.x0 = self.y0 = 0.0
        self.inv = 1.0
        if len(xo):
            self.x0, self.y0 = float(xo.min()), float(yo.min())
            extent = max(float(xo.max()) - self.x0, float(yo.max()) - self.y0)
            if extent > 0.0:
                self.inv = self.GRID / extent * (1.0 - 1e-12)
                self.nx = self.ny = self.GRID
        ix = ((xo - self.x0) * self.inv).astype(np.int32)
        iy = ((yo - self.y0) * self.inv).astype(np.int32)
        np.minimum(ix, self.nx - 1, out=ix)
        np.minimum(iy, self.ny - 1, out=iy)
        key = (iy * self.nx + ix).astype(np.uint16)     # GRID^2 cells fit 16 bits: numpy's stable sort of uint16 is a radix sort
        order = np.argsort(key, kind="stable")
        self.order = order if ok is None else ok[order]
        self.starts = np.concatenate(([0], np.cumsum(np.bincount(key, minlength=self.nx * self.ny))))

    def _cell(self, v, v0, n):
        c = np.floor((v - v0) * self.inv)
        return int(min(max(c, 0), n - 1)) if c == c else 0

    def rows(self, x_min, x_max, y_min, y_max):
        ix0, ix1 = self._cell(x_min, self.x0, self.nx), self._cell(x_max, self.x0, self.nx)
        iy0, iy1 = self._cell(y_min, self.y0, self.ny), self._cell(y_max, self.y0, self.ny)
        st, od = self.starts, self.order
        runs = [od[st[iy * self.nx + ix0]: st[iy * self.nx + ix1 + 1]] for iy in range(iy0, iy1 + 1)]
        cand = np.concatenate(runs) if runs else od[:0]
        xx, yy = self.x[cand], self.y[cand]
        return np.sort(cand[(xx >= x_min) & (xx < x_max) & (yy >= y_min) & (yy < y_max)])

    def positions(self, x_min, x_max, y_min, y_max, xs, ys):
        """As rows(), for data kept in GRID order: `xs`, `ys` are the coordinates permuted by `self.order`.  -> (rows ascending,
        pos) with order[pos] == rows, so `column_in_grid_order[pos]` is the column's values for `rows`.  The candidates of a box
        are a few contiguous runs of the grid-ordered arrays, so everything a window reads of a million-cell section comes from
        ~1 MB of neighbouring memory instead of 10^4 cache-missing rows scattered over the whole array."""
        ix0, ix1 = self._cell(x_min, self.x0, self.nx), self._cell(x_max, self.x0, self.nx)
        iy0, iy1 = self._cell(y_min, self.y0, self.ny), self._cell(y_max, self.y0, self.ny)
        st = self.starts
        runs = [np.arange(st[iy * self.nx + ix0], st[iy * self.nx + ix1 + 1]) for iy in range(iy0, iy1 + 1)]
        pos = np.concatenate(runs) if runs else np.zeros(0, np.int64)
        xx, yy = xs[pos], ys[pos]
        pos = pos[(xx >= x_min) & (xx < x_max) & (yy >= y_min) & (yy < y_max)]
        rows = self.order[pos]
        o = np.argsort(rows, kind="stable")
        return rows[o], pos[o]


class Section:
    """One tissue section as columns (no DataFrame): what the window pipeline reads of it.
    xy (n, 2) float64; types (n, T) float64, the commonCT columns in commonCT order; type_id (n,) int32 codes of the cell
    type (equal type <=> equal code); size (n,) (integer dtype kept: it decides the dtype of the triangle weights).
    The host-side grid and the grid-ordered copies of the columns (`grid`, `g_*`) are what the COLUMN pipeline subsets with; they are
    built on first use, so a section that only feeds a DeviceSection never pays for them."""

    def __init__(self, xy, types, type_id=None, size=None):
        self.xy = np.ascontiguousarray(xy, dtype=np.float64).reshape(-1, 2)
        t = np.ascontiguousarray(types, dtype=np.float64)
        self.types = t if (t.ndim == 2 and len(t) == len(self.xy)) else t.reshape(len(self.xy), -1)      # (0, T) stays (0, T)
        self.type_id = None if type_id is None else np.ascontiguousarray(type_id, dtype=np.int32)
        self.size = np.ones(len(self.xy), np.int64) if size is None else np.asarray(size)
        self._host = None

    def _host_grid(self):
        if self._host is None:
            grid = GridRows(self.xy[:, 0], self.xy[:, 1])
            od = grid.order     # the same columns once more in GRID order (GridRows.positions): what the window loop gathers from
            g_xy = np.ascontiguousarray(self.xy[od])
            self._host = {"grid": grid, "g_xy": g_xy, "g_types": np.ascontiguousarray(self.types[od]), "g_size": self.size[od],
                          "g_x": np.ascontiguousarray(g_xy[:, 0]), "g_y": np.ascontiguousarray(g_xy[:, 1]),
                          "g_type_id": None if self.type_id is None else self.type_id[od]}
        return self._host

    grid = property(lambda self: self._host_grid()["grid"])
    g_xy = property(lambda self: self._host_grid()["g_xy"])
    g_types = property(lambda self: self._host_grid()["g_types"])
    g_size = property(lambda self: self._host_grid()["g_size"])
    g_x = property(lambda self: self._host_grid()["g_x"])
    g_y = property(lambda self: self._host_grid()["g_y"])
    g_type_id = property(lambda self: self._host_grid()["g_type_id"])

    def window(self, box):
        """-> (rows ascending, positions into the grid-ordered columns) of the section's points inside the half-open box."""
        return self.grid.positions(*box, self.g_x, self.g_y)

    @classmethod
    def from_frame(cls, df, commonCT):
        import pandas as pd

        type_id = pd.factorize(df["cell_type"].to_numpy(), use_na_sentinel=False)[0] if "cell_type" in df.columns else None
        return cls(df[["X", "Y"]].to_numpy(dtype=np.float64), df[list(commonCT)].to_numpy(dtype=np.float64), type_id,
                   df["size"].to_numpy() if "size" in df.columns else None)


class WindowArrays:
    """Pre-MIP artefacts of one window as flat arrays (what PreparedInputs holds, without the frames): `rows_m` / `rows_r` are
    the section rows of the window's compacted aligned / reference cells, `pairs` (P, 2) int64 index into them, `costs` (P,)
    float64 in pair order, `triangles` (Tr, 3) of compacted aligned rows in the reference's kept order, `weights`, `signs`."""

    __slots__ = ("window", "rows_m", "rows_r", "pairs", "costs", "triangles", "weights", "signs", "axy", "rxy", "size", "error")

    def __init__(self, window):
        self.window = window
        for name in self.__slots__[1:]:
            setattr(self, name, None)

    @property
    def n_aligned(self):
        return len(self.rows_m)

    @property
    def n_ref(self):
        return len(self.rows_r)


def iter_window_arrays(ref, moving, plan, radius=250, knn=8, dist_ct_coeff=1.0, min_angle_deg=15, ignore_same_type_triangles=True,
                       cost_dtype=np.float64, ctx=None):
    """The pre-MIP path of every window of `plan` on COLUMNS (two `Section`s), in plan order: yields one `WindowArrays` per
    window -- the same pairs, costs, kept triangles, weights and signs `api.iter_prepared_windows` computes from frames
    (tests/test_gpu_run_same.py::test_window_arrays_equal_prepared_windows), without building a DataFrame per window.  The
    frame pipeline spends ~90 % of a window in pandas `take`s that the reference's return types force at its boundary; a
    caller that feeds the artefacts straight to kernels (bench.py --workload cfg5) does not need them.  Windows n+1..n+k are
    subset, pruned and compacted ahead and triangulated by the Qhull helpers while window n runs, as in the frame pipeline.
    A window whose prune leaves no pairs yields a WindowArrays whose `.error` is the ValueError run_same would raise."""
    from . import qhull_pool
    from ._trace import stage as marked
    from .knn import pairs_from_padded
    from .triangles import filter_triangles_by_radius

    depth = qhull_pool.lookahead()
    qhull_pool.warm(min(depth, len(plan)))
    cost_dtype = np.dtype(cost_dtype)

    def stage(w):
        with marked("prune+compact"):
            return prune(w)

    def prune(w):
        out = WindowArrays(w)
        (rows_r, pos_r), (rows_m, pos_m) = ref.window(w["box"]), moving.window(w["box"])
        axy, rxy = moving.g_xy[pos_m], ref.g_xy[pos_r]
        # the argument checks cKDTree makes for the reference are moot here: GridRows only returns finite points
        r = float(radius)
        if r != r or int(knn) <= 0 or len(rxy) == 0 or len(axy) == 0:
            idx = np.full((len(axy), 1), -1, np.int32)
        else:
            idx, _, _ = ops.knn_prune(axy, rxy, abs(r), knn, want_d2=False, ctx=ctx)
        kp = pairs_from_padded(idx)
        if len(kp) == 0:
            out.error = ValueError("No valid_pairs after KNN filtering. Increase radius and/or knn.")
            return out, None
        used_a = np.zeros(len(axy), bool)
        used_a[kp[:, 0]] = True
        used_r = np.zeros(len(rxy), bool)
        used_r[kp[:, 1]] = True
        ua, ur = np.flatnonzero(used_a), np.flatnonzero(used_r)            # compaction (src/utils.py:734-742)
        out.pairs = np.column_stack(((np.cumsum(used_a) - 1)[kp[:, 0]], (np.cumsum(used_r) - 1)[kp[:, 1]])).astype(np.int64)
        out.rows_m, out.rows_r = rows_m[ua], rows_r[ur]
        out.axy, out.rxy = np.ascontiguousarray(axy[ua]), np.ascontiguousarray(rxy[ur])
        return out, (qhull_pool.pool().submit(out.axy), pos_m[ua], pos_r[ur])

    def finish(out, staged):
        ticket, pos_m, pos_r = staged
        with marked("triangulate (wait for helper)"):
            tris = ticket.result()
        with marked("triangle filter"):
            tid = moving.g_type_id[pos_m] if (ignore_same_type_triangles and moving.g_type_id is not None) else None
            out.triangles = filter_triangles_by_radius(out.axy, tris, radius, ignore_same_type_triangles=ignore_same_type_triangles,
                                                       min_angle_deg=min_angle_deg, verbose=False, ctx=ctx, _rows_as_array=True,
                                                       _type_id=tid)
        with marked("triangle weights + source signs"):
            out.size = moving.g_size[pos_m]
            sign, weight = ops.tri_sign_weight(out.axy, out.size.astype(np.float64), out.triangles, ctx=ctx)
            out.weights = weight.astype(np.int64) if np.issubdtype(out.size.dtype, np.integer) else weight
            out.signs = sign.astype(np.float64)
        with marked("pair costs"):
            out.costs = ops.pair_cost(moving.g_types[pos_m], ref.g_types[pos_r], out.axy, out.rxy, out.pairs, dist_ct_coeff,
                                      dtype=cost_dtype, ctx=ctx).astype(np.float64, copy=False)
        return out

    ahead = {}
    for q in range(len(plan)):
        for nxt in range(q, min(q + 1 + depth, len(plan))):
            if nxt not in ahead:
                ahead[nxt] = stage(plan[nxt])
        out, staged = ahead.pop(q)
        yield out if out.error is not None else finish(out, staged)


# ---- the window path with the sections resident on the device (csrc/window.hip) -------------------------------------------

_W_ALIGNED_XY, _W_ALIGNED_ROWS, _W_ROWS_M, _W_ROWS_R, _W_PAIRS, _W_COSTS, _W_KEPT, _W_SIGNS, _W_WEIGHTS, _W_MATCH, _W_TRIANGLES = range(11)


def window_cell_grid(plan_or_grid, window_size, overlap):
    """(x0, y0, cell) of the grid on which every box of a window plan is a union of cells: the plan's boxes start at the grid
    origins int(x_min) + i * step and end window_size later (merged ones at a later origin + window_size: src/same.py:481-488,
    :527-542), so with cell = gcd(step, window_size) all their edges are cell edges (window sizes whose gcd with the step is small get
    quarter-window cells instead: correct for any box, only no longer test-free).  `plan_or_grid` = (xs, ys) of window_grid."""
    import math

    xs, ys = plan_or_grid
    step = int(window_size) - int(overlap)
    cell = float(math.gcd(step, int(window_size)))
    # a window of more than ~5 x 5 such cells (8 x 8 once merged) would leave the cell-run path (<= 64 cells):
    if window_size / cell > 5:
        # quarter windows instead -- boxes then cut through cells and their candidates are tested, still O(window)
        cell = window_size / 4.0
    return float(xs[0]), float(ys[0]), cell


class DeviceSection:
    """A `Section`'s XY, type columns and sizes uploaded once (same_section_create) and binned into a grid of cells; every window
    reads them in place.  cost_dtype float32 keeps the cost operands as float (BASELINE cfg 5), float64 is the reference's
    arithmetic.  `bin(x0, y0, cell)` re-bins the rows on the window grid (window_cell_grid), on which a window's rows are
    whole cells: do it once, before the windows run."""

    def __init__(self, section, cost_dtype=np.float64, ctx=None):
        import ctypes

        self.ctx = ctx = ops._ctx(ctx)
        self.section = section
        self.cost_dtype = np.dtype(cost_dtype)
        if self.cost_dtype not in (np.dtype(np.float64), np.dtype(np.float32)):
            raise ValueError(f"cost_dtype must be float64 or float32, not {self.cost_dtype}")
        size = np.ascontiguousarray(section.size, dtype=np.float64)
        tid = None if section.type_id is None else np.ascontiguousarray(section.type_id, dtype=np.int32)
        h = ctypes.c_void_p()
        with ctx.lock:
            rc = ctx.lib.same_section_create(ctx.handle, section.xy.ctypes.data, section.types.ctypes.data, section.types.shape[1],
                                             size.ctypes.data, None if tid is None else tid.ctypes.data, len(section.xy),
                                             int(self.cost_dtype == np.dtype(np.float32)), ctypes.byref(h))
            if rc != 0 and h.value:
                ctx.lib.same_section_destroy(h)
            ctx.check(rc, "same_section_create")
        self.handle = h

    def bin(self, x0, y0, cell_w, cell_h=None):
        with self.ctx.lock:
            self.ctx.check(self.ctx.lib.same_section_bin(self.handle, float(x0), float(y0), float(cell_w),
                                                         float(cell_w if cell_h is None else cell_h)),
                           "same_section_bin")
        return self

    def set_codes(self, codes, n_codes):
        """codes[row] = rank of the row's cell id among the section's ids (what the window merge on the device compares and orders by);
        None: a row's code is its number."""
        c = None if codes is None else np.ascontiguousarray(codes, dtype=np.int32)
        with self.ctx.lock:
            self.ctx.check(self.ctx.lib.same_section_set_codes(self.handle, None if c is None else c.ctypes.data, int(n_codes)),
                           "same_section_set_codes")

    def close(self):
        if getattr(self, "handle", None) and self.ctx.handle:        # a context that is already gone took its device memory along
            with self.ctx.lock:
                self.ctx.lib.same_section_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceWindow:
    """Device state of one window in flight (same_window): `stage` then `filter_finish` (or `finish` with triangles filtered by the
    caller), arrays of the state through `fetch`.  `stage_windows` / `filter_finish_windows` run a BATCH of windows per library call."""

    def __init__(self, ctx=None):
        import ctypes

        self.ctx = ctx = ops._ctx(ctx)
        h = ctypes.c_void_p()
        with ctx.lock:
            rc = ctx.lib.same_window_create(ctx.handle, ctypes.byref(h))
            if rc != 0 and h.value:
                ctx.lib.same_window_destroy(h)
            ctx.check(rc, "same_window_create")
        self.handle = h
        self.counts = (0, 0, 0, 0)
        self.n_triangles = 0

    def stage(self, moving, ref, box, radius, knn, dist_ct_coeff):
        """-> (aligned rows in the box, reference rows in the box, aligned rows kept, pairs)"""
        return stage_windows([self], moving, ref, [box], radius, knn, dist_ct_coeff)[0]

    # what -> (dtype, which count gives the length: 0 aligned in box, 1 refs in box, 2 kept, 3 pairs, 4 triangles, trailing width)
    _FETCH = {_W_ALIGNED_XY: (np.float64, 2, 2), _W_ALIGNED_ROWS: (np.int32, 2, 0), _W_ROWS_M: (np.int32, 0, 0),
              _W_ROWS_R: (np.int32, 1, 0),
              _W_PAIRS: (np.int32, 3, 2), _W_COSTS: (np.float64, 3, 0), _W_KEPT: (np.int32, 2, 0), _W_SIGNS: (np.int8, 4, 0),
              _W_WEIGHTS: (np.float64, 4, 0), _W_MATCH: (np.int32, 2, 0), _W_TRIANGLES: (np.int32, 4, 3)}

    def fetch(self, what):
        dtype, which, width = self._FETCH[what]
        n = self.n_triangles if which == 4 else self.counts[which]
        out = np.empty((n, width) if width else (n,), dtype)
        with self.ctx.lock:
            self.ctx.check(self.ctx.lib.same_window_fetch(self.handle, int(what), out.ctypes.data, out.nbytes), "same_window_fetch")
        return out

    STAT_NAMES = ("checked", "flipped", "xy_comparisons", "xy_violations", "xy_triangles", "area_flips", "greedy_rounds", "matched")

    def filter_finish(self, simplices, radius, angle_enabled, cos_thr, near_tol, ignore_same_type, no_match_penalty,
                      ensure_min_triangle_per_node=True):
        """filter_triangles_by_radius of the kept aligned cells' Delaunay simplices, then signs, weights, the greedy incumbent and the three
        sweeps, in one call.  -> (kept, added back, near, match_row, flag, stats); with near != 0 (cosines within near_tol of the
        threshold) the last three are None and the caller filters on the host and calls finish() with its triangles."""
        return filter_finish_windows([self], [simplices], radius, angle_enabled, cos_thr, near_tol, ignore_same_type, no_match_penalty,
                                     ensure_min_triangle_per_node)[0]

    def finish(self, triangles, no_match_penalty):
        """The same with triangles the CALLER filtered (kept ones, in the reference's order).  -> (section row of the matched reference
        cell per kept aligned cell or -1, flag byte per kept cell: bit 0 = XY-order sweep, bit 1 = vertex of an area-flipped triangle;
        stats dict)."""
        return filter_finish_windows([self], [triangles], 0.0, 0, 0.0, 0.0, False, no_match_penalty, True, prefiltered=True)[0][3:]

    def close(self):
        if getattr(self, "handle", None) and self.ctx.handle:
            with self.ctx.lock:
                self.ctx.lib.same_window_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# SAME_MERGE_REST
REST_RECORD = np.dtype([("row", "<i4"), ("ac", "<i4"), ("rc", "<i4"), ("wid", "<i4"), ("pos", "<i4"), ("cidx", "<i4"), ("flags", "<u4")])
# SAME_MERGE_FINAL
FINAL_RECORD = np.dtype([("a_row", "<i4"), ("r_row", "<i4"), ("cidx", "<i4"), ("wid", "<i4"), ("pos", "<i4"), ("flags", "<u4")])


class _PinnedBlocks:
    """Page-locked host blocks the device writes the merged table's columns into (same_host_alloc), handed out as numpy arrays.
    A block goes back to the pool when the last array made of it is gone (a result table is usually dropped before the next pass: the
    pool then serves every pass from the same block, and no pass pays the pinning again); at most KEEP blocks wait in the pool, and while
    more than LIMIT are out with callers (tables kept alive) `take` declines -- the caller then gathers on the host as before.
    A block comes back from a finalizer, i.e. wherever the interpreter happens to drop the last reference (possibly inside `take` itself,
    during a garbage collection): that path only moves entries between lists under a re-entrant lock; memory the pool does not keep is
    handed back to the driver by the next `take` / `drop`, never from the finalizer."""

    KEEP, LIMIT = 2, 4

    def __init__(self):
        import threading

        self.free, self.surplus, self.out, self.lock = [], [], 0, threading.RLock()

    def take(self, ctx, nbytes):
        """-> (ctypes char array over a pinned block of >= nbytes, address) or None"""
        import ctypes
        import weakref

        self._release_surplus()
        with self.lock:
            if self.out >= self.LIMIT:
                return None
            fit = [q for q, (c, cap, _p) in enumerate(list(self.free)) if c is ctx and cap >= nbytes]
            got = self.free.pop(min(fit, key=lambda q: self.free[q][1])) if fit else None
            self.out += 1
        if got is not None:
            _c, cap, ptr = got
        else:
            cap, ptr = (int(nbytes * 1.1) + (1 << 20)) & ~((1 << 20) - 1), ctypes.c_void_p()
            with ctx.lock:
                rc = ctx.lib.same_host_alloc(ctx.handle, cap, ctypes.byref(ptr))
            if rc != 0:
                with self.lock:
                    self.out -= 1
                return None
            ptr = ptr.value
        buf = (ctypes.c_char * cap).from_address(ptr)
        weakref.finalize(buf, self._back, ctx, cap, ptr)          # when the last array over `buf` is gone
        return buf, ptr

    def _back(self, ctx, cap, ptr):
        with self.lock:
            self.out -= 1
            (self.free if len(self.free) < self.KEEP and ctx.handle else self.surplus).append((ctx, cap, ptr))

    def _release_surplus(self):
        with self.lock:
            gone, self.surplus = self.surplus, []
        for ctx, _cap, ptr in gone:
            self._release(ctx, ptr)

    @staticmethod
    def _release(ctx, ptr):
        try:
            if ctx.handle:
                with ctx.lock:
                    ctx.lib.same_host_free(ctx.handle, ptr)
        except Exception:
            pass

    def drop(self, ctx=None):
        """free the waiting blocks (of one context, or all)"""
        with self.lock:
            gone = [e for e in self.free if ctx is None or e[0] is ctx]
            self.free = [e for e in self.free if e not in gone]
        for c, _cap, ptr in gone:
            self._release(c, ptr)
        self._release_surplus()


PINNED_BLOCKS = _PinnedBlocks()


class MergeAccumulator:
    """The rows of one pass over a window plan on one context (same_merge_acc, csrc/window_merge.hip): `collect` appends the matched cells
    of the windows' central regions where they are -- on the device --, `resolve_accumulators` / `finish` run the window merge on them."""

    def __init__(self, ctx=None):
        import ctypes

        self.ctx = ctx = ops._ctx(ctx)
        h = ctypes.c_void_p()
        with ctx.lock:
            rc = ctx.lib.same_merge_acc_create(ctx.handle, ctypes.byref(h))
            if rc != 0 and h.value:
                ctx.lib.same_merge_acc_destroy(h)
            ctx.check(rc, "same_merge_acc_create")
        self.handle = h

    def begin(self, expected_rows, near=None, reach=0.0, all_seam=False):
        """A new pass.  near = (near_start int32[n_pos + 1], near_boxes float64[.., 4]): per plan position the central regions of OTHER
        ranks' windows close enough to share cells with it (merge.seam_tables); None: one rank."""
        ctx = self.ctx
        if near is None:
            n_pos, ns, nb = 0, None, None
        else:
            ns, nb = np.ascontiguousarray(near[0], dtype=np.int32), np.ascontiguousarray(near[1], dtype=np.float64).reshape(-1, 4)
            n_pos = len(ns) - 1
        with ctx.lock:
            ctx.check(ctx.lib.same_merge_acc_begin(self.handle, int(expected_rows), n_pos, None if ns is None else ns.ctypes.data,
                                                   None if nb is None or len(nb) == 0 else nb.ctypes.data, float(reach),
                                                   int(bool(all_seam))),
                      "same_merge_acc_begin")

    def collect(self, states, trims, window_ids, plan_pos):
        """Enqueue only: the matched cells of the finished windows `states` that lie in their central regions join the accumulator."""
        ctx, n = self.ctx, len(states)
        t = np.ascontiguousarray(trims, dtype=np.float64).reshape(n, 4)
        w, p = np.ascontiguousarray(window_ids, dtype=np.int32), np.ascontiguousarray(plan_pos, dtype=np.int32)
        with ctx.lock:
            ctx.check(ctx.lib.same_window_collect(_handles(states), n, self.handle, t.ctypes.data, w.ctypes.data, p.ctypes.data),
                      "same_window_collect")

    def load(self, a_code, r_code, flags, window_ids, pos, cidx, n_codes_a, n_codes_r):
        """Rows from the host instead of from windows (the ranks' seam rows after their exchange): the accumulator then holds exactly
        these rows, whose cells are named by codes; `resolve_accumulators([acc], None, None)` and `finish` follow."""
        ctx = self.ctx
        i32 = lambda v: np.ascontiguousarray(v, dtype=np.int32)
        a, r, w, p, c = i32(a_code), i32(r_code), i32(window_ids), i32(pos), i32(cidx)
        f = np.ascontiguousarray(flags, dtype=np.uint8)
        with ctx.lock:
            ctx.check(ctx.lib.same_merge_acc_load(self.handle, a.ctypes.data, r.ctypes.data, f.ctypes.data, w.ctypes.data, p.ctypes.data,
                                                  c.ctypes.data,
                                                  len(a), int(n_codes_a), int(n_codes_r)), "same_merge_acc_load")

    def finish(self, winner_rows, fetch=True):
        """The REST rows the host's matching kept (accumulator row numbers) -> the merged table's rows, aligned codes ascending: as
        FINAL_RECORDs, or (fetch=False) only their number -- they stay on the device for `columns`; `final_rows()` fetches them later."""
        import ctypes

        ctx = self.ctx
        w = np.ascontiguousarray(winner_rows, dtype=np.int32)
        n = ctypes.c_int64(0)
        with ctx.lock:
            ctx.check(ctx.lib.same_merge_acc_finish(self.handle, w.ctypes.data if len(w) else None, len(w), ctypes.byref(n)),
                      "same_merge_acc_finish")
        self.n_final = n.value
        return self.final_rows() if fetch else self.n_final

    def final_rows(self):
        out = np.empty(self.n_final, FINAL_RECORD)
        with self.ctx.lock:
            self.ctx.check(self.ctx.lib.same_merge_acc_fetch(self.handle, 1, out.ctypes.data, out.nbytes), "same_merge_acc_fetch")
        return out

    def columns(self, dmoving, dref, n_final, n_types, extra_moving=(), extra_ref=()):
        """After finish(): the merged table's columns written by the device straight into page-locked host memory (enqueue only:
        `ctx.sync()` before reading): the moving section's type columns, X, Y, the reference's X, Y, the caller's extra 8-byte device
        columns (DeviceBuffers of 8-byte values per moving / reference row: ids, sizes), aligned_idx, window_id and plan position as int64,
        and the two flag columns.  -> (uint64 array (n_types + 4 + extras + 3, n_final), uint8 array (2, n_final)) over a pooled block,
        or None when no block is to be had (the caller gathers on the host)."""
        import ctypes

        n8 = n_types + 4 + len(extra_moving) + len(extra_ref) + 3
        got = PINNED_BLOCKS.take(self.ctx, max(1, n8 * 8 * n_final + 2 * n_final)) if n_final else None
        if got is None:
            return None
        buf, ptr = got
        ctx = self.ctx
        em = (ctypes.c_void_p * max(1, len(extra_moving)))(*[b.ptr for b in extra_moving])
        er = (ctypes.c_void_p * max(1, len(extra_ref)))(*[b.ptr for b in extra_ref])
        with ctx.lock:
            ctx.check(ctx.lib.same_merge_acc_columns(self.handle, dmoving.handle, dref.handle, em, len(extra_moving), er, len(extra_ref),
                                                     ptr,
                                                     int(n_final)), "same_merge_acc_columns")
        wide = np.frombuffer(buf, dtype=np.uint64, count=n8 * n_final).reshape(n8, n_final)
        flags = np.frombuffer(buf, dtype=np.uint8, count=2 * n_final, offset=n8 * 8 * n_final).reshape(2, n_final)
        return wide, flags

    def close(self):
        if getattr(self, "handle", None) and self.ctx.handle:
            with self.ctx.lock:
                self.ctx.lib.same_merge_acc_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def plain_accumulators(accs):
    """The pass is over and its table is wanted WITHOUT the merge: every accumulated row is a row of the table, windows in the order they
    were collected.  -> accs[0], holding the rows (`.n_final`, `.final_rows()`, `.columns(...)`)."""
    import ctypes

    ctx = accs[0].ctx
    n = ctypes.c_int64(0)
    handles = (ctypes.c_void_p * len(accs))(*[a.handle.value for a in accs])
    with ctx.lock:
        ctx.check(ctx.lib.same_merge_acc_plain(handles, len(accs), ctypes.byref(n)), "same_merge_acc_plain")
    accs[0].n_final = n.value
    return accs[0]


def resolve_accumulators(accs, dmoving, dref):
    """The pass is over: the accumulators of the worker contexts (in the order their runs of the plan have) -> ((rows, rows after the
    de-duplication, rest rows, rows final already), the REST rows as REST_RECORDs): what the device could not decide alone -- cells some
    window disagrees about, rows at a seam between ranks -- for merge.py's graph step; `accs[0].finish(winners)` completes the merge."""
    import ctypes

    ctx = accs[0].ctx
    counts = np.zeros(4, np.int64)
    handles = (ctypes.c_void_p * len(accs))(*[a.handle.value for a in accs])
    with ctx.lock:
        ctx.check(ctx.lib.same_merge_acc_resolve(handles, len(accs), None if dmoving is None else dmoving.handle,
                                                 None if dref is None else dref.handle, counts.ctypes.data), "same_merge_acc_resolve")
        rest = np.empty(int(counts[2]), REST_RECORD)
        ctx.check(ctx.lib.same_merge_acc_fetch(accs[0].handle, 0, rest.ctypes.data, rest.nbytes), "same_merge_acc_fetch")
    return tuple(int(c) for c in counts), rest


WINDOW_BATCH_MAX = 64      # SAME_WINDOW_BATCH_MAX


def _handles(states):
    import ctypes

    return (ctypes.c_void_p * len(states))(*[s.handle.value for s in states])


def stage_windows(states, moving, ref, boxes, radius, knn, dist_ct_coeff):
    """same_window_stage for a batch of windows of one context (one wait for all of them).  -> [counts per window]"""
    ctx, n = states[0].ctx, len(states)
    boxes = np.ascontiguousarray(boxes, dtype=np.float64).reshape(n, 4)
    counts = np.zeros((n, 4), np.int64)
    with ctx.lock:
        ctx.check(ctx.lib.same_window_stage(_handles(states), n, moving.handle, ref.handle, boxes.ctypes.data, float(radius), int(knn),
                                            float(dist_ct_coeff), counts.ctypes.data), "same_window_stage")
    for s, c in zip(states, counts.tolist()):
        s.counts, s.n_triangles = tuple(c), 0
    return [s.counts for s in states]


def filter_finish_windows(states, simplices, radius, angle_enabled, cos_thr, near_tol, ignore_same_type, no_match_penalty,
                          ensure_min_triangle_per_node=True, prefiltered=False):
    """same_window_filter_finish for a batch (one wait for all of them): `simplices[i]` are window i's Delaunay simplices, or with
    `prefiltered` its kept triangles.  -> [(kept, added back, near, match_row, flag byte, stats dict) per window]; a window with near != 0
    has None for the last three.  Every state's `order_ties` is set to the call's count of places where the answer hangs on the ORDER
    of the triangles or of their corners (include/same_hip.h; of consequence only when the simplices are not Qhull's own)."""
    ctx, n = states[0].ctx, len(states)
    tris = [ops._tris(t) for t in simplices]
    offsets = np.zeros(n + 1, np.int64)
    np.cumsum([len(t) for t in tris], out=offsets[1:])
    flat = tris[0] if n == 1 else np.concatenate(tris)
    kept_cells = [s.counts[2] for s in states]
    cell_off = np.concatenate(([0], np.cumsum(kept_cells))).astype(np.int64)
    match_row, flag = np.empty(int(cell_off[-1]), np.int32), np.empty(int(cell_off[-1]), np.uint8)
    stats, counts = np.zeros((n, 8), np.int64), np.zeros((n, 4), np.int64)
    with ctx.lock:
        ctx.check(ctx.lib.same_window_filter_finish(_handles(states), n, flat.ctypes.data, offsets.ctypes.data, int(bool(prefiltered)),
                                                    float(radius),
                                                    int(angle_enabled), float(cos_thr), float(near_tol), int(bool(ignore_same_type)),
                                                    int(bool(ensure_min_triangle_per_node)), float(no_match_penalty), match_row.ctypes.data,
                                                    flag.ctypes.data, stats.ctypes.data, counts.ctypes.data), "same_window_filter_finish")
    out = []
    for i, s in enumerate(states):
        kept, added, near, s.order_ties = (int(c) for c in counts[i])
        s.n_triangles = 0 if near else kept + added
        if near:
            out.append((kept, added, near, None, None, None))
        else:
            a, b = int(cell_off[i]), int(cell_off[i + 1])
            out.append((kept, added, near, match_row[a:b], flag[a:b], dict(zip(DeviceWindow.STAT_NAMES, stats[i].tolist()))))
    return out


class DeviceWindowResult:
    """What one window of `iter_device_windows` leaves on the host: `rows_m` section rows of the kept aligned cells, `axy` their XY,
    `triangles` the kept Delaunay triangles over them (None unless asked for or filtered on the host; `n_triangles` always),
    `match_row` the section row of each cell's matched reference cell (-1 = none), `point_flag` the XY-order sweep's per-cell flag,
    `flip_flag` 1 for the vertices of triangles whose signed area flips, `stats` the sweeps' counters, `counts` (aligned in box, refs in
    box, kept, pairs); `state` is the live DeviceWindow until the generator is asked for the first window of the next batch (pairs,
    costs, signs ... through `state.fetch`)."""

    __slots__ = ("window", "error", "rows_m", "axy", "triangles", "n_triangles", "match_row", "point_flag", "flip_flag", "stats",
                 "counts", "state")

    def __init__(self, window):
        self.window = window
        for name in self.__slots__[1:]:
            setattr(self, name, None)


class TriangulationCache:
    """Delaunay simplices remembered per window (a DIAGNOSTIC: bench.py's "what would a pass cost if the triangulations were free").
    `submit(points, key)` hands back the simplices of `key` when it has seen the window before, else asks the helper pool and keeps
    the answer; the simplices are the pool's, i.e. scipy's, either way."""

    class _Ready:
        def __init__(self, value):
            self._value = value

        def result(self):
            return self._value

    class _Pending:
        def __init__(self, cache, key, ticket):
            self.cache, self.key, self.ticket = cache, key, ticket

        def result(self):
            v = self.cache.known[self.key] = self.ticket.result()
            return v

    def __init__(self):
        self.known = {}

    @staticmethod
    def _tag(points, key):
        """What an entry is remembered under: the window's id AND its points (count + a checksum of the coordinates' bits), so that a
        cache reused with another plan or section whose windows reuse ids asks the pool again instead of answering with stale simplices."""
        p = np.ascontiguousarray(points, dtype=np.float64)
        return key, len(p), int(np.bitwise_xor.reduce(p.view(np.uint64).ravel())) if len(p) else 0

    def submit(self, points, key=None):
        from . import qhull_pool

        if key is None:                                   # a window without an id cannot be remembered
            return qhull_pool.pool().submit(points)
        tag = self._tag(points, key)
        if tag in self.known:
            return self._Ready(self.known[tag])
        return self._Pending(self, tag, qhull_pool.pool().submit(points))


def iter_device_windows(ref, moving, dref, dmoving, plan, radius=250, knn=8, dist_ct_coeff=1.0, min_angle_deg=15,
                        ignore_same_type_triangles=True, no_match_penalty=100.0, ctx=None, fetch_triangles=False, triangulator=None,
                        triangulate=True, batch=None, collector=None):
    """The window path of `iter_window_arrays` + the greedy incumbent and the three sweeps, with both sections resident on the
    device (`dref`, `dmoving`: DeviceSections of `ref`, `moving`): per window the host only triangulates (Qhull helpers, windows
    ahead as before) and receives the match; the triangle filter runs on the device too, unless a cosine sits within 8 ulp of the
    angle threshold (then the host re-decides it with the reference's literal expression, as triangles.classify_triangles does).  Yields one
    DeviceWindowResult per window in plan order; the numbers are those of the column pipeline
    (tests/test_gpu_run_same.py::test_device_windows_equal_the_column_pipeline).  A window without pairs yields `.error`.
    Windows go to the library in BATCHES of `batch` (default $SAME_WINDOW_BATCH, else 8): one stage call, and later one filter + finish
    call, for up to that many windows -- one wait per call instead of one per window, and the device works on one window while the host
    enqueues the next.  The states of a batch stay live (`result.state`) until the generator is asked for the first window of the next.
    `triangulator` (default: the Qhull helper pool) is anything with `submit(points, key=...) -> ticket with .result()`; a ticket
    whose `.native` is true after `.result()` brought simplices that are not scipy's own and has `.qhull()` for those (delaunay.py).
    `triangulate=False` stops after the stage call (rows, prune, costs, compaction): the caller brings its own triangles
    (api.sliding_window_matching with a caller's triangulation) and reads pairs / costs through `state.fetch`.
    `collector(states, windows)` is called once per finished batch with its windows' live states (the window merge's accumulator:
    MergeAccumulator.collect)."""
    import os
    from collections import deque

    from . import qhull_pool
    from ._trace import stage as marked
    from .triangles import cos_threshold, filter_triangles_by_radius

    ctx = ops._ctx(ctx)
    angle_enabled, cos_thr = cos_threshold(min_angle_deg)
    near_tol = float(8 * np.spacing(abs(cos_thr))) if (angle_enabled and np.isfinite(cos_thr)) else 0.0
    own_threads = getattr(triangulator, "threads", None)        # a triangulator with threads of its own (delaunay.NativeTriangulator)
    depth = qhull_pool.lookahead() if own_threads is None else int(own_threads)
    if own_threads is None:
        qhull_pool.warm(min(depth, len(plan)))
    B = max(1, min(int(batch if batch is not None else os.environ.get("SAME_WINDOW_BATCH", "8")), WINDOW_BATCH_MAX, max(len(plan), 1)))
    if batch is None and depth > 0:
        # a rank with three helpers (eight ranks on a 16-CPU host) gains nothing from collecting eight tickets at once
        B = min(B, max(2, depth))
    # windows staged and not yet finished: one per helper BEYOND the batch being finished (a batch's tickets are collected together, and
    # nothing is handed over meanwhile: with only `depth` in flight, 12 helpers and batches of 8 ran 8 of 12 helpers -- 264 against 431
    # windows/s for one thread of a rank that shares its host with another)
    ahead_max = max(B, depth + int(os.environ.get("SAME_WINDOW_AHEAD", str(B))))
    # window states (their device buffers, grown once) are kept with the context from one call to the next: a pass over a plan
    # then makes no device allocation at all.  In use at once: the windows ahead + the batch being staged + the batch last yielded
    cache = ctx.__dict__.setdefault("_device_windows", [])
    want = min(ahead_max + 2 * B, max(len(plan), 1) + B)
    free = [cache.pop() for _ in range(min(want, len(cache)))]
    r = float(radius)
    prune_possible = r == r and int(knn) > 0

    def take_state():
        return free.pop() if free else DeviceWindow(ctx)

    def stage_batch(windows):
        """-> [(result, state or None, ticket or None)] for `windows`, staged by ONE library call"""
        outs = [DeviceWindowResult(w) for w in windows]
        if not prune_possible:
            for out in outs:
                out.error = ValueError("No valid_pairs after KNN filtering. Increase radius and/or knn.")
            return [(out, None, None) for out in outs]
        states = [take_state() for _ in windows]
        try:
            with marked("subset + prune + costs + compaction (device)"):
                counts = stage_windows(states, dmoving, dref, [w["box"] for w in windows], abs(r), knn, dist_ct_coeff)
        except BaseException:
            free.extend(states)                       # a refused batch (SAME_EINVAL ...) must not take its states out of the pool
            raise
        staged = []
        for q, (out, state) in enumerate(zip(outs, states)):
            out.counts = counts[q]
            if out.counts[3] == 0:
                free.append(state)
                out.error = ValueError("No valid_pairs after KNN filtering. Increase radius and/or knn.")
                staged.append((out, None, None))
                continue
            ticket = None
            try:
                out.rows_m, out.axy = state.fetch(_W_ALIGNED_ROWS), state.fetch(_W_ALIGNED_XY)
                if triangulate:
                    with marked("triangulate (hand-over; waits for a free helper)"):
                        ticket = (qhull_pool.pool().submit(out.axy) if triangulator is None
                                  else triangulator.submit(out.axy, key=out.window.get("window_id")))
            except BaseException:
                free.extend(st for st in states[q:])
                free.extend(st for _o, st, _t in staged if st is not None)
                raise
            staged.append((out, state, ticket))
        return staged

    def finish_batch(group):
        """filter + signs + incumbent + sweeps of the staged windows of `group`, by ONE library call (+ one per window whose filter met a
        cosine at the threshold)"""
        todo = [(out, state, ticket) for out, state, ticket in group if state is not None]
        for out, state, _t in todo:
            out.state, out.n_triangles = state, 0
        if not triangulate or not todo:
            return
        with marked("triangulate (wait for helper)"):
            tris = [ticket.result() for _o, _s, ticket in todo]
        with marked("filter + signs + incumbent + sweeps (device)"):
            res = filter_finish_windows([st for _o, st, _t in todo], tris, radius, angle_enabled, cos_thr, near_tol,
                                        ignore_same_type_triangles,
                                        no_match_penalty)
        # simplices that are not Qhull's own (delaunay.py: the same triangles in another order): where the window's numbers hang on that
        # order -- the device counted such places, or a cosine sits at the threshold and the host is about to re-decide the filter --
        # the window is finished again with scipy's
        for q, ((_o, state, ticket), r) in enumerate(zip(todo, res)):
            if getattr(ticket, "native", False) and (state.order_ties or r[2]):
                with marked("order ties: the window again with Qhull's simplices"):
                    tris[q] = ticket.qhull()
                    res[q] = filter_finish_windows([state], [tris[q]], radius, angle_enabled, cos_thr, near_tol,
                                                   ignore_same_type_triangles, no_match_penalty)[0]
        for (out, state, _t), simplices, (_kept, _added, near, match_row, cell_flags, stats) in zip(todo, tris, res):
            if near:
                with marked("triangle filter (host: a cosine at the threshold)"):
                    tid = moving.type_id[out.rows_m] if (ignore_same_type_triangles and moving.type_id is not None) else None
                    out.triangles = filter_triangles_by_radius(out.axy, simplices, radius,
                                                               ignore_same_type_triangles=ignore_same_type_triangles,
                                                               min_angle_deg=min_angle_deg, verbose=False, ctx=ctx, _rows_as_array=True,
                                                               _type_id=tid)
                with marked("signs + incumbent + sweeps (device)"):
                    match_row, cell_flags, stats = state.finish(out.triangles, no_match_penalty)
            out.match_row, out.stats = match_row, stats
            # the library packs both per-cell flags into one byte
            out.point_flag, out.flip_flag = cell_flags & 1, (cell_flags >> 1) & 1
            out.n_triangles = state.n_triangles
            if fetch_triangles and out.triangles is None:
                out.triangles = state.fetch(_W_TRIANGLES)
        if collector is not None:
            with marked("central rows to the merge accumulator (device, enqueue only)"):
                collector([st for _o, st, _t in todo], [o.window for o, _s, _t in todo])

    pending, live, nxt = deque(), [], 0
    try:
        while nxt < len(plan) or pending:
            while nxt < len(plan) and (not pending or len(pending) + min(B, len(plan) - nxt) <= ahead_max):
                windows = plan[nxt:nxt + B]
                nxt += len(windows)
                pending.extend(stage_batch(windows))
            free.extend(live)                     # the caller has moved on: the previous batch's states go back to the pool
            live = []
            group = [pending.popleft() for _ in range(min(B, len(pending)))]
            live = [state for _o, state, _t in group if state is not None]
            finish_batch(group)
            for out, _state, _ticket in group:
                yield out
    finally:
        free.extend(state for _o, state, _t in pending if state is not None)
        free.extend(live)
        cache.extend(free)
        for extra in cache[want:]:            # a context keeps what one pass needs, not every state it ever had
            extra.close()
        del cache[want:]
