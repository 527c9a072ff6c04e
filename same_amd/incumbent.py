"""The window loop without a solver: sliding_window_matching (src/same.py:297-595) with every window's solution taken to be the greedy
MIP start (src/init_helpers.py:104-133) -- the incumbent the reference itself hands Gurobi first -- swept by the lazy-constraint body
(src/same.py:645-669), the XY-order sweep (src/violationhelper.py:53-117) and the area flips (src/same.py:1362-1402), and trimmed
to each window's central region (src/same.py:565-582).  It is the product function for hosts without a Gurobi licence, the first stage
of a two-stage run (the table is what `merge_window_matches_unique_ref` takes) and what `bench.py --workload cfg5` times.

Same arguments as `sliding_window_matching`; the result has its columns wherever they are defined without a solver:

    aligned_idx [ref_idx] <commonCT...> X Y ref_X ref_Y size ref_size Ref_<cell_id_col> Aligned_<cell_id_col> time_limit_reached
    triangle_violation filtered_violation run_time window_id

* `triangle_violation`: the cell is a vertex of a triangle whose signed area flips under the matching (as src/same.py:1464-1469).
* `filtered_violation`: the reference intersects the XY-order sweep's points with the triangles the SOLVER penalised
  (src/same.py:1411-1432); without a solver there are no penalties, and the column carries the sweep's own per-cell flag -- the
  ranking key the window merge sorts by (src/helpers.py:745-753) keeps its meaning: unflagged proposals of a pair win.
* `ref_idx` (index in the window's compacted reference frame) needs the window's pair list on the host and is only made on request
  (`window_local_indices=True`); `aligned_idx` is free.  `run_time` is 0.0, `time_limit_reached` False.

Two routes produce the same table (tests/test_gpu_run_same.py::test_incumbent_table_routes_agree):
  device   both frames resident on the GPU, two library calls per window, the incumbent and the sweeps computed where the pairs are
           (csrc/window.hip); the host triangulates, receives (match, flags) per window and gathers the table's columns ONCE at the
           end.  Windows are walked by `workers` threads with a context each.
  general  every window becomes a `PreparedInputs` (either pipeline of same_amd.api) and the incumbent + sweeps run through the
           host-buffer entry points: caller-supplied triangulations (MetaCell inputs), the cell-type-priority filter, inputs the
           sections cannot hold.
"""
import os
import threading

import numpy as np
import pandas as pd

from . import ops
from ._trace import stage
from .api import _stage_prune, prepare_same_inputs
from .window_api import _WindowJob, _WindowSubsetter, _prepared_from_device, _staged_from_device, _window_error

STAT_KEYS = ("pairs", "triangles", "checked", "flipped", "xy_violations", "area_flips", "matched")


def _default_workers():
    from . import qhull_pool

    share = qhull_pool.cpu_budget() / qhull_pool.cpu_sharers()[0]
    return 2 if share >= 8 else 1       # a second Python thread only pays where there are CPUs to feed it


class _TableBuilder:
    """The device route's result table.  As the windows come a builder keeps, per window, the section rows of the matched cells inside the
    central trim (a few small index arrays); the columns are gathered from the caller's frames ONCE, when the pass is over, by `table()`:
    the final columns are allocated at their full length and filled slice by slice on `GATHER_THREADS` threads (numpy copies without the
    interpreter lock; the Qhull helpers are idle by then) -- no per-window frames, no concatenation.  Where the frame's own columns are
    float64 (the usual case) the type columns and the coordinates come from the sections' row-major copies: a slice's rows are fetched as
    whole rows (one cache line per row instead of one per column) and laid out as columns while they are in cache."""

    SLICE = 16384        # rows per task: a slice's row-major block of type columns stays in L2 between its gather and its split

    def __init__(self, job, sections, with_ref_idx):
        ref, mov = job.ref, job.moving
        self.job, self.with_ref_idx = job, with_ref_idx
        self.cts, self.cid = list(job.commonCT), job.optim_params["cell_id_col"]
        f64 = np.dtype(np.float64)
        # a frame's column may be a strided view of its block (a frame made from a 2-D array): np.take would copy such a source whole on
        # EVERY call, so the 1-D sources are made contiguous here, once per job (no copy where they already are)
        col = lambda df, c: np.ascontiguousarray(df[c].to_numpy())
        plain_types = all(mov[c].dtype == f64 for c in self.cts) and len(set(self.cts)) == len(self.cts)
        self.type_block = sections[1].types if plain_types else None
        self.type_cols = None if self.type_block is not None else [col(mov, c) for c in self.cts]
        both_xy = all(df[c].dtype == f64 for df in (ref, mov) for c in ("X", "Y"))
        self.mov_xy, self.ref_xy = (sections[1].xy, sections[0].xy) if both_xy else (None, None)
        self.xy_cols = None if both_xy else ([col(mov, c) for c in ("X", "Y")], [col(ref, c) for c in ("X", "Y")])
        self.mov_size = col(mov, "size") if "size" in mov.columns else None
        self.ref_size = col(ref, "size") if "size" in ref.columns else None
        self.ref_id, self.mov_id = col(ref, self.cid), col(mov, self.cid)
        self.parts = []

    def add(self, pos, w, dw, ref_idx=None):
        x, y = dw.axy[:, 0], dw.axy[:, 1]
        tx0, tx1, ty0, ty1 = w["trim"]                                  # central region (src/same.py:565-582), matched cells only
        c = np.flatnonzero((dw.match_row >= 0) & (x >= tx0) & (x < tx1) & (y >= ty0) & (y < ty1))
        self.parts.append((pos, w["window_id"], dw.rows_m[c], dw.match_row[c], c, None if ref_idx is None else ref_idx[c],
                           dw.point_flag[c], dw.flip_flag[c]))

    @staticmethod
    def keys(builders):
        """What the window merge reads of the rows, without their columns: (aligned section row, reference section row, XY-order flag,
        window id, plan position) per row, the builders' windows laid end to end."""
        parts = [p for b in builders for p in b.parts]
        lens = [len(p[2]) for p in parts]
        cat = lambda q, dt: (np.concatenate([p[q] for p in parts]) if parts else np.zeros(0)).astype(dt, copy=False)
        return (cat(2, np.int64), cat(3, np.int64), cat(6, bool), np.repeat(np.array([p[1] for p in parts], np.int64), lens),
                np.repeat(np.array([p[0] for p in parts], np.int64), lens))

    @staticmethod
    def table(builders, select=None, plan_pos=None):
        """The builders' windows laid end to end (each builder walked a contiguous run of the plan, so this is plan order); `select`:
        these rows of that table only, in this order (the rows the window merge keeps: the other rows' columns are never gathered)."""
        parts = [p for b in builders for p in b.parts]
        lens = [len(p[2]) for p in parts]
        if (int(sum(lens)) if select is None else len(select)) == 0:
            return pd.DataFrame()
        me = builders[0]                 # the sources are the job's: the same for every builder
        pick = (lambda v: v) if select is None else (lambda v: v[select])
        cat = lambda q, dt: pick(np.concatenate([p[q] for p in parts])).astype(dt, copy=False)
        spread = lambda values: pick(np.repeat(np.array(values, np.int64), lens))
        with_pos = me.job.mine is not None if plan_pos is None else plan_pos
        return me.gather(cat(2, np.int64), cat(3, np.int64), cat(4, np.int64), cat(5, np.int64) if me.with_ref_idx else None, cat(7, bool),
                         cat(6, bool), spread([p[1] for p in parts]), spread([p[0] for p in parts]) if with_pos else None)

    def gather(self, ra, rr, aligned_idx, ref_idx, triangle_violation, filtered_violation, window_id, plan_pos=None, only=None):
        """The result table of matched cells (moving section rows `ra` -> reference section rows `rr`): the columns of
        src/same.py:1264-1278, :1464-1470, gathered from the caller's frames slice by slice on the gather threads.
        only: gather just these table columns (-> dict of arrays): the ones the device-side gather of `table_from_device` does not cover."""
        from concurrent.futures import ThreadPoolExecutor

        from .merge import GATHER_THREADS

        me, n = self, len(ra)
        want = (lambda name: True) if only is None else (lambda name: name in only)
        out = {"aligned_idx": aligned_idx}
        if ref_idx is not None:
            out["ref_idx"] = ref_idx
        new = lambda like: np.empty(n, like.dtype)
        types_wanted = any(want(ct) for ct in me.cts)
        xy_wanted = any(want(k) for k in ("X", "Y", "ref_X", "ref_Y"))
        for ct, src in zip(me.cts, me.type_cols if me.type_block is None else [me.type_block] * len(me.cts)):
            out[ct] = new(src) if types_wanted else None
        if not xy_wanted:
            for k in ("X", "Y", "ref_X", "ref_Y"):
                out[k] = None
        elif me.mov_xy is not None:
            for k in ("X", "Y", "ref_X", "ref_Y"):
                out[k] = np.empty(n, np.float64)
        else:
            (mx, my), (rx, ry) = me.xy_cols
            out["X"], out["Y"], out["ref_X"], out["ref_Y"] = new(mx), new(my), new(rx), new(ry)
        cid_r, cid_a = f"Ref_{me.cid}", f"Aligned_{me.cid}"
        out["size"] = (new(me.mov_size) if me.mov_size is not None else np.ones(n, np.int64)) if want("size") else None
        out["ref_size"] = (new(me.ref_size) if me.ref_size is not None else np.ones(n, np.int64)) if want("ref_size") else None
        out[cid_r], out[cid_a] = new(me.ref_id) if want(cid_r) else None, new(me.mov_id) if want(cid_a) else None
        out["time_limit_reached"] = np.zeros(n, bool)
        out["triangle_violation"] = triangle_violation
        out["filtered_violation"] = filtered_violation
        out["run_time"] = np.zeros(n)
        out["window_id"] = window_id
        if plan_pos is not None:
            out["__plan_pos"] = plan_pos

        def fill(lo):
            hi = min(n, lo + _TableBuilder.SLICE)
            a, r = ra[lo:hi], rr[lo:hi]
            # np.take(src, rows, axis=0) copies whole rows: 3-6x the speed of src[rows] on the (n, 8) / (n, 2) blocks
            take = np.take
            if not types_wanted:
                pass
            elif me.type_block is not None:
                block = take(me.type_block, a, axis=0)       # (rows, T): the commonCT columns in commonCT order
                for q, ct in enumerate(me.cts):
                    out[ct][lo:hi] = block[:, q]
            else:
                for ct, col in zip(me.cts, me.type_cols):
                    take(col, a, out=out[ct][lo:hi], mode="clip")       # (rows are valid: "clip" only spares numpy its bounce buffer)
            if not xy_wanted:
                pass
            elif me.mov_xy is not None:
                axy, rxy = take(me.mov_xy, a, axis=0), take(me.ref_xy, r, axis=0)
                out["X"][lo:hi], out["Y"][lo:hi], out["ref_X"][lo:hi], out["ref_Y"][lo:hi] = axy[:, 0], axy[:, 1], rxy[:, 0], rxy[:, 1]
            else:
                (mx, my), (rx, ry) = me.xy_cols
                for k, col, rows in (("X", mx, a), ("Y", my, a), ("ref_X", rx, r), ("ref_Y", ry, r)):
                    take(col, rows, out=out[k][lo:hi], mode="clip")
            if me.mov_size is not None and want("size"):
                take(me.mov_size, a, out=out["size"][lo:hi], mode="clip")
            if me.ref_size is not None and want("ref_size"):
                take(me.ref_size, r, out=out["ref_size"][lo:hi], mode="clip")
            if want(cid_r):
                take(me.ref_id, r, out=out[cid_r][lo:hi], mode="clip")
            if want(cid_a):
                take(me.mov_id, a, out=out[cid_a][lo:hi], mode="clip")

        gathers_any = types_wanted or xy_wanted or any(want(k) and src is not None for k, src in
                                                        (("size", me.mov_size), ("ref_size", me.ref_size), (cid_r, me.ref_id),
                                                         (cid_a, me.mov_id)))
        starts = range(0, n, _TableBuilder.SLICE)
        if not gathers_any:
            pass
        elif len(starts) == 1:
            fill(0)
        else:
            with ThreadPoolExecutor(max_workers=GATHER_THREADS) as pool:
                list(pool.map(fill, starts))
        return pd.DataFrame(out, copy=False) if only is None else out

    def device_columns_possible(self):
        """the frame's type columns and coordinates are float64 and the type columns distinct: the sections hold exactly their values"""
        return self.type_block is not None and self.mov_xy is not None

    def table_from_device(self, frames, acc, with_plan_pos=False):
        """The merged table with its columns gathered where the sections are: the DEVICE writes the type columns, X, Y, ref_X, ref_Y, the
        8-byte id / size columns of the frames, aligned_idx, window_id and the two flag columns of the final rows straight into a pooled
        page-locked host block (MergeAccumulator.columns); the table's arrays are views of it.  Columns the device cannot hold (ids that
        are strings ...) are gathered by the host meanwhile.  None: no block to be had -- the caller gathers on the host."""
        n = acc.n_final
        with stage("table: page-locked block + the device's gather enqueued"):
            extra = frames.table_columns(self.cid)
            got = acc.columns(frames.dmov, frames.dref, n, len(self.cts), [b for _n, b, _d in extra["mov"]],
                              [b for _n, b, _d in extra["ref"]])
        if got is None:
            return None
        wide, flags = got
        names = (list(self.cts) + ["X", "Y", "ref_X", "ref_Y"] + [nm for nm, _b, _d in extra["mov"] + extra["ref"]]
                 + ["aligned_idx", "window_id", "__plan_pos"])
        dtypes = [np.float64] * (len(self.cts) + 4) + [d for _n, _b, d in extra["mov"] + extra["ref"]] + [np.int64, np.int64, np.int64]
        cid_r, cid_a = f"Ref_{self.cid}", f"Aligned_{self.cid}"
        missing = [k for k in ("size", "ref_size", cid_r, cid_a) if k not in names]
        host = {}
        if missing:            # beside the device's gather (which was only enqueued)
            final = acc.final_rows()
            host = self.gather(final["a_row"].astype(np.int64), final["r_row"].astype(np.int64), None, None, None, None, None,
                               only=set(missing))
        with stage("table: wait for the device's columns"):
            acc.ctx.sync()
        with stage("table: the frame over the block"):
            return self._frame_over(wide, flags, names, dtypes, host, n, with_plan_pos)

    def _frame_over(self, wide, flags, names, dtypes, host, n, with_plan_pos):
        cid_r, cid_a = f"Ref_{self.cid}", f"Aligned_{self.cid}"
        dev = {nm: wide[q].view(dt) for q, (nm, dt) in enumerate(zip(names, dtypes))}
        col = lambda k: dev[k] if k in dev else host[k]
        out = {"aligned_idx": dev["aligned_idx"]}
        for k in list(self.cts) + ["X", "Y", "ref_X", "ref_Y", "size", "ref_size", cid_r, cid_a]:
            out[k] = col(k)
        out["time_limit_reached"] = np.zeros(n, bool)
        out["triangle_violation"], out["filtered_violation"] = flags[0].view(bool), flags[1].view(bool)
        out["run_time"] = np.zeros(n)
        out["window_id"] = dev["window_id"]
        if with_plan_pos:
            out["__plan_pos"] = dev["__plan_pos"]
        return pd.DataFrame(out, copy=False)


def incumbent_of_prepared(prep, commonCT, with_ref_idx=True, ctx=None, use_device=True):
    """(match table of ONE window as run_same's post-solve builds it, stats) from its pre-MIP artefacts, through the host-buffer entry
    points: greedy start -> matching -> lazy-constraint body, XY-order sweep, area flips.  The general route of this module.
    use_device: a PreparedInputs made by the device-resident window path with its pair list untouched carries the incumbent and the
    sweeps already (computed where the pairs are, by same_window_filter_finish): take them instead of computing them again."""
    op = prep.optim_params
    dw = getattr(prep, "device", None)
    if use_device and dw is not None and dw.match_row is not None and isinstance(prep.valid_pairs, np.ndarray):
        return _table_of_device_window(prep, dw, commonCT, with_ref_idx)
    pairs = np.ascontiguousarray(np.asarray(prep.valid_pairs, dtype=np.int64).reshape(-1, 2), dtype=np.int32)
    costs, n_a, n_r = prep.costs_array, prep.n_aligned, prep.n_ref
    a_df, r_df, tris = prep.aligned_df, prep.ref_df, prep.triangles_array
    size = a_df["size"].to_numpy(dtype=np.float64)
    wants = ops.pair_rowmin(pairs, costs, n_a, ctx=ctx) < float(op["no_match_penalty"]) * size       # src/init_helpers.py:104,118-122
    pair_of_row, _rounds = ops.greedy_match(pairs, costs, n_a, n_r, wants, ctx=ctx)
    ai = np.flatnonzero(pair_of_row >= 0)
    ri = pairs[pair_of_row[ai], 1].astype(np.int64)
    match = np.full(n_a, -1, np.int32)
    match[ai] = ri
    axy, rxy = a_df[["X", "Y"]].to_numpy(dtype=np.float64), r_df[["X", "Y"]].to_numpy(dtype=np.float64)
    t32 = np.ascontiguousarray(tris, dtype=np.int32).reshape(-1, 3)
    sw = ops.BoundSweep(t32, prep.signs_array.astype(np.int8), rxy, n_a, ctx=ctx)
    try:
        checked, viol = sw.sweep_match(match)
    finally:
        sw.close()
    _edge, _tflag, pflag, counts = ops.xyorder_sweep(axy, rxy, t32, match, ctx=ctx)
    _before, _after, _m3, flipped = ops.area_flip(axy, rxy, t32, match, ctx=ctx)
    flip_node = np.zeros(n_a, bool)
    if len(t32):
        flip_node[t32[flipped.astype(bool)].reshape(-1)] = True
    stats = {"pairs": len(pairs), "triangles": len(t32), "checked": int(checked), "flipped": len(viol), "xy_violations": int(counts[1]),
             "area_flips": int(np.count_nonzero(flipped)), "matched": len(ai)}
    return _window_table(prep, commonCT, ai, ri, flip_node, pflag, with_ref_idx), stats


def _match_table(a_df, r_df, ra, rr, commonCT, cid, aligned_idx, ref_idx, triangle_violation, filtered_violation):
    """run_same's post-solve match table (src/same.py:1264-1278, :1464-1470): the columns of the matched cells read from `a_df` / `r_df`
    at rows `ra` / `rr` -- a window's own two frames at its own indices, or the CALLER's frames at section rows (the same values: a
    window's frames are rows of the caller's).  Plain indexing: a frame's column may be a strided view of its block, which np.take would
    first copy whole."""
    out = {"aligned_idx": aligned_idx.astype(np.int64)}
    if ref_idx is not None:
        out["ref_idx"] = ref_idx.astype(np.int64)
    for ct in list(commonCT) + ["X", "Y"]:
        out[ct] = a_df[ct].to_numpy()[ra]
    for ct in ("X", "Y"):
        out[f"ref_{ct}"] = r_df[ct].to_numpy()[rr]
    out["size"] = a_df["size"].to_numpy()[ra] if "size" in a_df.columns else np.ones(len(ra), np.int64)       # src/same.py:934-940
    out["ref_size"] = r_df["size"].to_numpy()[rr] if "size" in r_df.columns else np.ones(len(rr), np.int64)
    out[f"Ref_{cid}"] = r_df[cid].to_numpy()[rr]
    out[f"Aligned_{cid}"] = a_df[cid].to_numpy()[ra]
    out["time_limit_reached"] = np.zeros(len(ra), bool)
    out["triangle_violation"] = np.asarray(triangle_violation).astype(bool)
    out["filtered_violation"] = np.asarray(filtered_violation).astype(bool)
    out["run_time"] = np.zeros(len(ra))
    return pd.DataFrame(out, copy=False)      # the columns are this function's own arrays: no consolidating copy


def _window_table(prep, commonCT, ai, ri, flip_node, pflag, with_ref_idx):
    """the table of matched aligned rows `ai` -> reference rows `ri` of the window's own (compacted) frames"""
    return _match_table(prep.aligned_df, prep.ref_df, ai, ri, commonCT, prep.optim_params["cell_id_col"], ai, ri if with_ref_idx else None,
                        flip_node[ai], pflag[ai])


def _table_of_device_window(prep, dw, commonCT, with_ref_idx):
    """The window's table from what the device left.  The columns are read from the CALLER's frames by section row (prep.rows_m, the
    matched reference's section row), so the window's own two frames -- rows of the caller's, made on first access -- are never made."""
    ai = np.flatnonzero(dw.match_row >= 0)
    rj = dw.match_row[ai].astype(np.int64)                    # section rows of the matched reference cells
    stats = _device_stats(dw)
    # rows_r (ascending section rows of the compacted reference frame) -> the compacted index of every matched reference cell
    ri = np.searchsorted(prep.rows_r, rj) if (with_ref_idx or prep.sources is None) else None
    if prep.sources is None:
        return _window_table(prep, commonCT, ai, ri, dw.flip_flag, dw.point_flag, with_ref_idx), stats
    a_src, r_src = prep.sources
    return _match_table(a_src, r_src, np.asarray(prep.rows_m, dtype=np.int64)[ai], rj, commonCT, prep.optim_params["cell_id_col"], ai, ri,
                        dw.flip_flag[ai], dw.point_flag[ai]), stats


def _device_stats(dw):
    """a window's stats record (STAT_KEYS) from what the device counted"""
    st = dw.stats
    return {"pairs": dw.counts[3], "triangles": dw.n_triangles, "checked": st["checked"], "flipped": st["flipped"],
            "xy_violations": st["xy_violations"], "area_flips": st["area_flips"], "matched": st["matched"]}


def _device_ref_idx(dw):
    """index of every kept aligned cell's matched reference in the window's COMPACTED reference frame (src/utils.py:734-742), -1 = none"""
    from .windows import _W_MATCH, _W_PAIRS, _W_ROWS_R

    st = dw.state
    pairs, n_box = st.fetch(_W_PAIRS), len(st.fetch(_W_ROWS_R))
    used = np.zeros(n_box, bool)
    used[pairs[:, 1]] = True
    m = st.fetch(_W_MATCH)
    return np.where(m >= 0, (np.cumsum(used) - 1)[np.maximum(m, 0)], -1)


def sliding_window_incumbent(ref, moving, commonCT=None, outprefix=None, moving_delaunay=None, moving_delaunay_vertex_col=None,
                             optim_params=None, gurobi_params=None, ignore_precomputed_triangulation=False, *, workers=None,
                             window_local_indices=False, return_stats=False, triangulator=None, ctx=None, merge=False, batch=None,
                             _shard=None, _pipeline=None, _route=None, _merge_channel=None):
    """See the module text.  -> DataFrame (with return_stats: (DataFrame, [per-window stats dict in plan order])).
    workers: threads walking this process's windows on the device route (default: 2 where the process has >= 8 CPUs, else 1).
    A window whose prune leaves no pairs raises the ValueError run_same raises for it (src/same.py:1003), as the reference's loop does.
    `triangulator`, `batch` (windows per library call on the device route): see windows.iter_device_windows.  `_route` = 'device' |
    'general' (testing: forces a route).  optim_params["hip_delaunay"] = "native" (device route): the windows are triangulated by
    libsame_hip's own triangulator where that is provably the same as asking scipy (delaunay.py); the table is the same.
    merge=True: the table after `merge_window_matches_unique_ref` (src/helpers.py:692-815) -- one row per aligned and per reference cell,
    aligned ids ascending -- without the pre-merge table ever being laid out: the merge reads the rows' keys, and only the rows it keeps
    get their columns.  With `_shard` and a `_merge_channel` (dist.MergeChannel) the result is this rank's PART of the merged table
    (dist.sharded_merged_window_incumbent)."""
    job = _WindowJob(ref, moving, commonCT, outprefix, moving_delaunay, moving_delaunay_vertex_col, optim_params, gurobi_params,
                     ignore_precomputed_triangulation, _shard)
    frames, own = job.device_frames(_pipeline, ctx=ctx)
    fast = frames is not None and not job.caller_triangulation and not job.optim_params["ignore_knn_if_matched"]
    if _route is not None:
        if _route == "device" and not fast:
            raise ValueError("the device route does not apply to these inputs")
        fast = _route == "device"
    stats = {}
    try:
        if merge and job.all_matches:
            raise ValueError("merge=True does not resume from an outprefix that already holds windows")
        if fast:
            if triangulator is None:
                from . import delaunay

                if delaunay.mode(job.optim_params) == "native":       # optim_params["hip_delaunay"] / $SAME_DELAUNAY (delaunay.py)
                    triangulator = delaunay.shared()
                    triangulator.reset()
            table = _device_route(job, frames, workers, window_local_indices, triangulator, stats, merge, _merge_channel, batch)
        else:
            table = _general_route(job, frames, window_local_indices, stats, ctx)
            if merge:
                from .merge import merge_table_part

                # (reach: the prune's radius bounds the distance of a pair's two cells on every route -- no collective to measure it, so
                # ranks on different routes still make the same exchanges)
                from .window_api import codes_of_ids, frame_id_codes

                cid = job.optim_params["cell_id_col"]
                _codes, unique, (mov_ids, ref_ids) = frame_id_codes(job.moving, job.ref, cid)
                to_codes = lambda a, r: (codes_of_ids(mov_ids, a), codes_of_ids(ref_ids, r))       # what the device route exchanges too
                table = merge_table_part(table, job.plan, job.owner, _merge_channel, cid, reach=abs(float(job.optim_params["radius"])),
                                         ids_unique=unique, id_codes=to_codes)
    finally:
        if own:
            frames.close()
    if job.output_file and len(table):
        table.to_csv(job.output_file, index=False)
    return (table, [stats[pos] for pos in sorted(stats)]) if return_stats else table


def _merged_rows(job, frames, builders, channel):
    """The window merge on the device route's keys -> rows of the builders' table that the merged table keeps, in its order."""
    from . import merge as M

    a_row, r_row, viol, wid, pos = _TableBuilder.keys(builders)
    with stage("merge: cell ids of the rows"):
        (a_code, r_code), unique = frames.id_codes(job.optim_params["cell_id_col"])
        a_ids, r_ids = a_code[a_row], r_code[r_row]
    if channel is None or channel.world == 1:
        return M.merged_part_rows(a_ids, r_ids, viol, wid, None, None)
    with stage("merge: seam rows marked"):
        if unique:
            axy, rxy = frames.mov_sec.xy, frames.ref_sec.xy
            # rows whose window has no foreign window near are not looked at (seam_rows); the prune's radius bounds a pair's distance
            def coords(b, e):
                a, r = axy[a_row[b:e]], rxy[r_row[b:e]]
                return a[:, 0], a[:, 1], r[:, 0], r[:, 1]

            seam = M.seam_rows(pos, coords, job.plan, job.owner, channel.rank, abs(float(job.optim_params["radius"])))
        else:
            seam = np.ones(len(a_row), bool)
    return M.merged_part_rows(a_ids, r_ids, viol, wid, pos, seam, channel.rank, channel.tables)


def _device_route(job, frames, workers, with_ref_idx, triangulator, stats, merge=False, channel=None, batch=None):
    n_workers = max(1, int(workers if workers is not None else _default_workers()))
    n_workers = min(n_workers, max(1, len(job.todo)))
    contexts = frames.worker_contexts(n_workers)
    sections = (frames.ref_sec, frames.mov_sec)
    builders = [_TableBuilder(job, sections, with_ref_idx) for _ in range(n_workers)]
    lock = threading.Lock()
    # worker q walks a contiguous run of this process's windows
    cut = [len(job.todo) * q // n_workers for q in range(n_workers + 1)]
    # merge=True: the windows' matches stay where they are.  Every batch's central rows join the pass's accumulator on the device
    # (csrc/window_merge.hip); the merge runs there, and only what it could not decide alone comes to the host (`_merge_on_device`).
    # (window_local_indices needs every window's pair list on the host: the keys go through the builders then, `_merged_rows`.)
    accs = None
    device_table = builders[0].device_columns_possible() and os.environ.get("SAME_TABLE_COLUMNS", "device") != "host"
    if not with_ref_idx and (merge or (device_table and not job.all_matches)):
        accs = _begin_accumulators(job, frames, contexts, cut, channel if merge else None)
    pos_of = {id(w): pos for pos, w in job.todo}

    def walk(q):
        mine = job.todo[cut[q]:cut[q + 1]]
        collector = None
        if accs is not None:
            collector = lambda states, windows: accs[q].collect(states, [w["trim"] for w in windows], [w["window_id"] for w in windows],
                                                                [pos_of[id(w)] for w in windows])
        for (pos, w), dw in zip(mine, frames.windows([w for _p, w in mine], ctx=contexts[q], triangulator=triangulator,
                                                     collector=collector, batch=batch)):
            if dw.error is not None:
                raise dw.error
            with stage("table rows (central trim)"):
                if accs is None:
                    builders[q].add(pos, w, dw, _device_ref_idx(dw) if with_ref_idx else None)
                rec = _device_stats(dw)
            with lock:
                stats[pos] = rec

    if n_workers == 1:
        walk(0)
    else:
        errors = []

        def guarded(q):
            try:
                walk(q)
            except BaseException as e:   # noqa: BLE001 -- re-raised in the calling thread below
                errors.append(e)

        threads = [threading.Thread(target=guarded, args=(q,), name=f"same-windows-{q}") for q in range(n_workers)]
        [t.start() for t in threads]
        [t.join() for t in threads]
        if errors:
            raise errors[0]
    if accs is not None:
        calls0 = accs[0].ctx.stats()
        if merge:
            done = _merge_on_device(job, frames, accs, channel)
        else:               # the windows' tables end to end, as they were collected (src/same.py:583-590)
            from .windows import plain_accumulators

            with stage("table rows (accumulated on the device, in plan order)"):
                done = plain_accumulators(accs)
        with_pos = job.mine is not None and not merge
        with stage("table (columns of the rows that stay: by the device into page-locked memory, else on the gather threads)"):
            if not done.n_final:
                return pd.DataFrame()
            me = builders[0]
            if device_table:
                table = me.table_from_device(frames, done, with_pos)     # the columns gathered on the device, into page-locked memory
                if table is not None:
                    # (the merge counted its own calls; what is new here: the rows laid end to end, the columns' launch and wait)
                    _count_merge_calls(frames, accs[0].ctx, accs[0].ctx.stats() if merge else calls0, passes=0 if merge else 1)
                    return table
            final = done.final_rows()
            flags = final["flags"]
            return me.gather(final["a_row"].astype(np.int64), final["r_row"].astype(np.int64), final["cidx"].astype(np.int64), None,
                             (flags & 2) != 0, (flags & 1) != 0, final["wid"].astype(np.int64),
                             final["pos"].astype(np.int64) if with_pos else None)
    select = _merged_rows(job, frames, builders, channel) if merge else None
    with stage("table (columns gathered on the gather threads)"):
        table = _TableBuilder.table(builders, select, plan_pos=False if merge else None)
    if job.all_matches:                      # rows of windows finished by an earlier run (resume)
        table = pd.concat(job.all_matches + ([table] if len(table) else []), ignore_index=True)
    return table


def _begin_accumulators(job, frames, contexts, cut, channel):
    """A merge accumulator per worker context, begun for this pass: sized for the worker's windows, with the seams of this rank's share
    of the plan when the plan is dealt over ranks."""
    from . import merge as M

    accs, unique = frames.accumulators(contexts, job.optim_params["cell_id_col"])
    near, reach = None, 0.0
    if channel is not None and channel.world > 1 and unique:
        reach = abs(float(job.optim_params["radius"]))          # the prune's radius bounds the distance of a pair's two cells
        known = frames.__dict__.setdefault("_seam_tables", {})
        key = (channel.rank, channel.world, reach, job.owner.tobytes(), len(job.plan))
        if key not in known:
            known[key] = M.seam_tables(job.plan, job.owner, channel.rank, reach)
        near = known[key]
    calls0 = accs[0].ctx.stats()
    for q, acc in enumerate(accs):
        expected = sum(w["n_mov"] for _pos, w in job.todo[cut[q]:cut[q + 1]])
        acc.begin(expected, near, reach, all_seam=channel is not None and channel.world > 1 and not unique)
    _count_merge_calls(frames, accs[0].ctx, calls0, passes=0)
    return accs


def _count_merge_calls(frames, ctx0, calls0, passes=1):
    """what a pass asked of the runtime ONCE on the first worker's context -- the accumulator's begin, the merge (resolve, finish: a sort's
    worth of launches) or the rows laid end to end, the table's columns; the windows' calls are counted per window -- read by bench.py
    and by the launch-budget test"""
    spent = frames.__dict__.setdefault("merge_runtime_calls", {})
    for k, v in ctx0.stats().items():
        spent[k] = spent.get(k, 0) + v - calls0[k]
    spent["passes"] = spent.get("passes", 0) + passes


def _merge_on_device(job, frames, accs, channel):
    """The window merge of the accumulated rows (src/helpers.py:692-815): codes, de-duplication, degrees and the rows that stand alone
    on the device; components / Hopcroft-Karp of the contested cells, and the seam rows' exchange between ranks, here.
    -> the accumulator that holds the merged table's rows (aligned ids ascending; `.n_final`, `.final_rows()`)."""
    from . import merge as M
    from .windows import resolve_accumulators

    ctx0 = accs[0].ctx
    calls0 = ctx0.stats()
    with stage("merge: accumulated rows resolved (device)"):
        _counts, rest = resolve_accumulators(accs, frames.dmov, frames.dref)
    a, r, wid = rest["ac"].astype(np.int64), rest["rc"].astype(np.int64), rest["wid"].astype(np.int64)
    viol = (rest["flags"] & 1) != 0
    if channel is None or channel.world == 1:
        rows = M._resolve_rows(a, r, viol, wid, M.already_deduplicated) if len(rest) else np.zeros(0, np.int64)
    else:
        seam = (rest["flags"] & 4) != 0
        mine, sent = M.part_decided_here(a, r, viol, wid, rest["pos"].astype(np.int64), seam, channel.rank, M.already_deduplicated,
                                         seq=rest["cidx"])
        with stage("merge: seam rows exchanged"):
            parts = channel.tables(sent)
        with stage("merge: seam step (the same on every rank)"):
            rows = np.concatenate((mine, _seam_step_on_device(frames, ctx0, parts, channel.rank)))
    with stage("merge: winners to the device, final rows in order"):
        # the rows stay on the device: the table's columns are gathered from them there
        accs[0].finish(rest["row"][rows], fetch=False)
    _count_merge_calls(frames, ctx0, calls0)
    return accs[0]


def _seam_step_on_device(frames, ctx, parts, rank):
    """The common step of a merge dealt over ranks (merge.part_after_seam_step) with the gathered seam rows on the device: they are
    loaded into an accumulator of their own (their cells named by codes, in the single process's order), de-duplicated, counted and
    classed there like a pass's own rows; the host matches the few cells that are still contested.  -> this rank's winners (indices into
    the table it sent)."""
    from . import merge as M
    from .windows import MergeAccumulator, resolve_accumulators

    parts = [p for p in parts if len(p["row"])]
    if not parts:
        return np.zeros(0, np.int64)
    col = lambda c: np.concatenate([p[c] for p in parts])
    order_key = col("order")
    order = None if bool(np.all(order_key[1:] >= order_key[:-1])) else np.argsort(order_key, kind="stable")
    take = (lambda c: col(c)) if order is None else (lambda c: col(c)[order])
    a, r = take("a"), take("r")
    acc = frames.__dict__.get("_seam_acc")
    if acc is None or acc.ctx is not ctx:
        acc = frames.__dict__["_seam_acc"] = MergeAccumulator(ctx)
        frames.__dict__.setdefault("_accs", {})[("seam", id(ctx))] = acc           # closed with the frames
    # pos carries the sender's rank, cidx its row in the table it sent: what comes back names the winners by both
    acc.load(a, r, take("viol"), take("window"), take("rank"), take("row"), int(a.max()) + 1, int(r.max()) + 1)
    _counts, rest = resolve_accumulators([acc], None, None)
    won = M._resolve_rows(rest["ac"].astype(np.int64), rest["rc"].astype(np.int64), (rest["flags"] & 1) != 0, rest["wid"].astype(np.int64),
                          M.already_deduplicated, mark=False) if len(rest) else np.zeros(0, np.int64)
    final = acc.finish(rest["row"][won])
    return final["cidx"][final["pos"] == rank].astype(np.int64)


def _general_route(job, frames, with_ref_idx, stats, ctx):
    commonCT, op, gp = job.commonCT, job.optim_params, job.gurobi_params

    def prepared():
        if frames is not None:
            plan = [w for _pos, w in job.todo]
            for (pos, w), dw in zip(job.todo, frames.windows(plan, triangulate=not job.caller_triangulation, ctx=ctx,
                                                             fetch_triangles=True)):
                if dw.error is not None:
                    raise _window_error(dw, op)
                if job.caller_triangulation:
                    st = _staged_from_device(dw, frames, commonCT, op, gp, job.moving_delaunay, job.vertex_col, verbose=False)
                    yield pos, w, prepare_same_inputs(None, None, commonCT, verbose=False, ctx=ctx, _staged=st)
                else:
                    yield pos, w, _prepared_from_device(dw, frames, op, gp, verbose=False, vertex_col=job.vertex_col)
            return
        from . import qhull_pool

        ref_rows, moving_rows = _WindowSubsetter(job.ref), _WindowSubsetter(job.moving)
        depth = qhull_pool.lookahead()               # windows cut, pruned and handed to the Qhull helpers ahead of the one being finished
        qhull_pool.warm(min(depth, len(job.todo)))
        ahead = {}
        for q, (pos, w) in enumerate(job.todo):
            for nxt in range(q, min(q + 1 + depth, len(job.todo))):
                if nxt not in ahead:
                    box = job.todo[nxt][1]["box"]
                    ahead[nxt] = _stage_prune(ref_rows.subset(*box), moving_rows.subset(*box), commonCT, job.moving_delaunay,
                                              job.vertex_col, op, gp,
                                              job.ignore_pre, False, ctx, prefetch=True, fresh_frames=True)
            yield pos, w, prepare_same_inputs(None, None, commonCT, verbose=False, ctx=ctx, _staged=ahead.pop(q))

    keep_csv, job.outprefix = job.outprefix, None        # the table is written once, by the caller of this route
    try:
        for pos, w, prep in prepared():
            if len(prep.valid_pairs) == 0:               # every node unconstrained under the caller's triangulation: nothing to match
                continue
            with stage("incumbent + sweeps + table (general route)"):
                window_matches, stats[pos] = incumbent_of_prepared(prep, commonCT, with_ref_idx, ctx=ctx, use_device=False)
            job.collect(pos, w, window_matches)
    finally:
        job.outprefix = keep_csv
    return job.result()
