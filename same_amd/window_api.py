"""The reference's window API on top of the pre-MIP path: sliding_window_matching (src/same.py:297-595) and what stands behind it.

Split from same_amd/api.py (run_same and prepare_same_inputs live there).  Two pipelines stand behind the same signature; which one
runs, and on what, is said at the head of the code below and in DESIGN.md section 0."""
import os
from typing import Any, Dict, Optional

import numpy as np
import pandas as pd

from . import ops
from ._trace import stage
from .api import PreparedInputs, _Staged, _run_same, _say, _stage_prune, prepare_same_inputs
from .params import init_gurobi_params, init_optim_params


# ------------------------------------------------------------------------------------------ windows
# The reference's window loop (src/same.py:507-593) cuts both frames per window with four comparisons over the WHOLE frame
# (subset_data, :293-295) and runs the whole pre-MIP path of run_same on the subsets.  Two pipelines stand behind the same
# signature here:
#   "device" (default)  both frames are uploaded ONCE as sections binned on the window grid (windows.DeviceSection); per window the
#                       rows of the box, the prune, the pair costs, the compaction, the triangle filter, weights and signs come from
#                       two library calls on the resident sections (csrc/window.hip); the host only triangulates (Qhull helpers,
#                       windows ahead) and hands `PreparedInputs` -- whose frames are made from the section rows when somebody reads
#                       them -- to the unchanged run_same body.
#   "frames"            every window's frames are cut on the host and go through the host-buffer entry points (`_stage_prune`): the
#                       pipeline of rounds 1-4, kept as the fallback for inputs the sections cannot hold and as the comparison
#                       the tests and `bench.py --cfg5-pipeline frames` run.
# SAME_WINDOW_PIPELINE=frames forces the second one.
def window_pipeline(requested=None):
    v = (requested or os.environ.get("SAME_WINDOW_PIPELINE", "device")).lower()
    if v not in ("device", "frames"):
        raise ValueError(f"window pipeline must be 'device' or 'frames', got {v!r}")
    return v


def subset_data(df, x_min, x_max, y_min, y_max):
    """src/same.py:293-295."""
    return df[(df["X"] >= x_min) & (df["X"] < x_max) & (df["Y"] >= y_min) & (df["Y"] < y_max)]


class _WindowSubsetter:
    """subset_data for many boxes of one frame.  The reference evaluates four comparisons over the WHOLE frame for every
    window (O(N * windows), src/same.py:521-526); here the rows are binned once into a uniform grid (windows.GridRows), a
    window gathers the cells its box touches and only those rows are tested exactly.  The rows come back in frame order with
    their original labels, i.e. the frame `subset_data` returns (NaN / infinite coordinates fall outside every box either way)."""

    def __init__(self, df):
        from .windows import GridRows

        self.df = df
        self.grid = GridRows(df["X"].to_numpy(dtype=np.float64), df["Y"].to_numpy(dtype=np.float64))

    def subset(self, x_min, x_max, y_min, y_max):
        return self.df.iloc[self.grid.rows(x_min, x_max, y_min, y_max)]


def _window_frame(df, rows, vertex_col=None, aligned=False):
    """Rows `rows` of the caller's frame as the frame run_same holds after its prune: the helper columns of src/same.py:934-970
    (size default, __orig_idx = the caller's index labels, __tri_vid on the aligned side) and the renumbering of src/utils.py:739-740."""
    # iloc with a row list makes fresh data; the shallow copy only drops pandas' "copy of a slice" mark
    out = df.iloc[rows].copy(deep=False)
    labels = out.index.to_numpy()
    out.index = pd.RangeIndex(len(out))           # the renumbering, in place: reset_index(drop=True) would copy every block once more
    if "size" not in out.columns:
        out["size"] = 1
    if "__orig_idx" not in out.columns:
        out["__orig_idx"] = labels
    if aligned:
        if vertex_col is None:
            out["__tri_vid"] = labels
        else:
            if vertex_col not in out.columns:
                raise ValueError(f"aligned_delaunay_vertex_col='{vertex_col}' not in aligned_df")
            out["__tri_vid"] = out[vertex_col].to_numpy()
    return out


def frame_id_codes(moving, ref, cid):
    """((aligned code per moving row, reference code per ref row), whether every id names one row, (sorted distinct moving ids, sorted
    distinct ref ids)): the cell ids of both frames as int64 ranks in the ORDER of the ids (equal id <=> equal code) -- what the window
    merge compares, orders by and exchanges between ranks instead of the ids themselves (strings, floats ...)."""
    def codes(df):
        ids = df[cid].to_numpy()
        if ids.dtype.kind in "iu" and (len(ids) < 2 or bool(np.all(ids[1:] > ids[:-1]))):
            return np.arange(len(ids), dtype=np.int64), True, ids                 # ascending ids: a row's code is its number
        c, uniq = pd.factorize(ids, sort=True, use_na_sentinel=False)
        return c.astype(np.int64), len(uniq) == len(ids), np.asarray(uniq)

    (mc, mu, m_ids), (rc, ru, r_ids) = codes(moving), codes(ref)
    return (mc, rc), mu and ru, (m_ids, r_ids)


def codes_of_ids(sorted_distinct_ids, ids):
    """the codes `frame_id_codes` gave these ids"""
    return pd.Index(sorted_distinct_ids).get_indexer(np.asarray(ids)).astype(np.int64)


class _DeviceFrames:
    """The two frames of a window loop as sections resident on the device (windows.DeviceSection), binned on the grid on which every
    window box is a union of cells (windows.window_cell_grid).  `windows(plan)` runs the per-window device path over them."""

    def __init__(self, ref, moving, commonCT, optim_params, cell_grid=None, ctx=None):
        from .windows import DeviceSection, Section

        self.ref, self.moving, self.commonCT, self.op = ref, moving, list(commonCT), optim_params
        self.cost_dtype = np.dtype(optim_params.get("hip_cost_dtype", "float64"))
        self.ctx = ops._ctx(ctx)
        self.ref_sec, self.mov_sec = Section.from_frame(ref, self.commonCT), Section.from_frame(moving, self.commonCT)
        self.dref = self.dmov = None
        self._worker_ctx = []            # contexts (= streams) of the worker threads beyond the first, kept with their window states
        with stage("sections to the device + binning on the window grid"):
            self.dref = DeviceSection(self.ref_sec, self.cost_dtype, self.ctx)
            self.dmov = DeviceSection(self.mov_sec, self.cost_dtype, self.ctx)
            if cell_grid is not None:
                from ._lib import SAME_EINVAL, SameHipError
                for sec in (self.dref, self.dmov):
                    try:
                        sec.bin(*cell_grid)
                    except SameHipError as e:
                        # the one thing that may be passed over is the REFUSAL of a window grid too fine for the section's extent (> 2^22
                        # cells, SAME_EINVAL): that section keeps the grid of its own and boxes that cut through its cells are tested row
                        # by row -- the same rows either way.  A failed sort or allocation inside the binning is an error of this call.
                        if e.code != SAME_EINVAL:
                            raise

    @staticmethod
    def refusal(ref, moving, commonCT, op, vertex_col=None):
        """Why these inputs go through the frame pipeline instead (None: the sections can hold them).  Everything listed makes the
        reference itself raise inside run_same; the frame pipeline raises the same error at the same window."""
        if np.dtype(op.get("hip_cost_dtype", "float64")) not in (np.dtype(np.float64), np.dtype(np.float32)):
            return "hip_cost_dtype"
        for df in (ref, moving):
            if any(c not in df.columns for c in list(commonCT) + ["X", "Y"]):
                return "a commonCT / coordinate column is missing"
            if any(df[c].dtype.kind not in "fiub" for c in list(commonCT) + ["X", "Y"]):
                return "a commonCT / coordinate column is not numeric"
            if "size" in df.columns and df["size"].dtype.kind not in "fiub":
                return "size is not numeric"
        if vertex_col is not None and vertex_col not in moving.columns:
            return "aligned_delaunay_vertex_col missing"
        if op["ignore_same_type_triangles"] and "cell_type" not in moving.columns:
            return "cell_type missing (same-type triangle rule)"
        if op["ignore_knn_if_matched"] and ("cell_type" not in moving.columns or "cell_type" not in ref.columns):
            return "cell_type missing (priority filter)"
        return None

    def windows(self, plan, triangulate=True, ctx=None, triangulator=None, fetch_triangles=False, collector=None, batch=None):
        from .windows import iter_device_windows

        op = self.op
        return iter_device_windows(self.ref_sec, self.mov_sec, self.dref, self.dmov, plan, radius=op["radius"], knn=op["knn"],
                                   dist_ct_coeff=op["dist_ct_coeff"], min_angle_deg=op.get("min_angle_deg", 15),
                                   ignore_same_type_triangles=op["ignore_same_type_triangles"], no_match_penalty=op["no_match_penalty"],
                                   ctx=self.ctx if ctx is None else ctx, triangulate=triangulate, triangulator=triangulator,
                                   fetch_triangles=fetch_triangles, collector=collector, batch=batch)

    def accumulators(self, contexts, cid):
        """One merge accumulator per worker context (kept with the frames: a pass re-uses the arrays of the last), and the sections' id
        codes for `cid` on the device.  -> ([accumulator per context], whether every id names one row)."""
        from .windows import MergeAccumulator

        accs = self.__dict__.setdefault("_accs", {})
        for c in contexts:
            if id(c) not in accs:
                accs[id(c)] = MergeAccumulator(c)
        (mov_code, ref_code), unique = self.id_codes(cid)
        if self.__dict__.get("_codes_on_device") != cid:
            identity = lambda code: len(code) == 0 or (code[0] == 0 and code[-1] == len(code) - 1 and bool(np.all(code[1:] > code[:-1])))
            for sec, code in ((self.dmov, mov_code), (self.dref, ref_code)):
                sec.set_codes(None if identity(code) else code, int(code.max()) + 1 if len(code) else 0)
            self.__dict__["_codes_on_device"] = cid
        return [accs[id(c)] for c in contexts], unique

    def box_rows(self, box, state):
        """(aligned rows, reference rows) of the frames inside the box -- subset_data of both frames, as row positions."""
        from .windows import _W_ROWS_M, _W_ROWS_R

        state.stage(self.dmov, self.dref, box, 1.0, 1, 1.0)
        return state.fetch(_W_ROWS_M), state.fetch(_W_ROWS_R)

    def id_codes(self, cid):
        """((aligned code per moving row, reference code per ref row), whether every id names one row) -- `frame_id_codes`, made once per
        frames and id column."""
        known = self.__dict__.setdefault("_id_codes", {})
        if cid not in known:
            known[cid] = frame_id_codes(self.moving, self.ref, cid)
        return known[cid][:2]

    def id_uniques(self, cid):
        """the sorted distinct ids of (moving, ref): code -> id"""
        self.id_codes(cid)
        return self._id_codes[cid][2]

    def table_columns(self, cid):
        """The frames' columns that are 8-byte numbers and go into the result table besides the sections' own (sizes, cell ids), resident on
        the device for the device-side gather of the merged table (MergeAccumulator.columns): -> {"mov": [(table column, DeviceBuffer,
        dtype)], "ref": [...]}; columns of other types are gathered by the host.  Uploaded once per frames and id column."""
        known = self.__dict__.setdefault("_table_columns", {})
        if cid not in known:
            def eligible(df, names):
                out = []
                for column, name in names:
                    dt = df[column].dtype if column in df.columns else None
                    if isinstance(dt, np.dtype) and dt.kind in "iuf" and dt.itemsize == 8:
                        host = np.ascontiguousarray(df[column].to_numpy())
                        out.append((name, self.ctx.to_device(host.view(np.uint64)), host.dtype))
                return out
            known[cid] = {"mov": eligible(self.moving, (("size", "size"), (cid, f"Aligned_{cid}"))),
                          "ref": eligible(self.ref, (("size", "ref_size"), (cid, f"Ref_{cid}")))}
        return known[cid]

    def worker_contexts(self, n):
        """n contexts on the sections' device for n worker threads: this object's own first, then extra ones that live (with the window
        states they have grown) until close()."""
        from . import _lib

        while len(self._worker_ctx) < n - 1:
            self._worker_ctx.append(_lib.Context(self.ctx.device))
        return [self.ctx] + self._worker_ctx[:n - 1]

    def close(self):
        for acc in self.__dict__.pop("_accs", {}).values():
            acc.close()
        for held in self.__dict__.pop("_table_columns", {}).values():
            for _name, buf, _dt in held["mov"] + held["ref"]:
                buf.free()
        for c in self._worker_ctx:
            c.close()
        self._worker_ctx = []
        for sec in (self.dref, self.dmov):
            if sec is not None:
                sec.close()
        self.dref = self.dmov = None


def _window_error(dw, optim_params):
    """What the reference raises for a window whose prune finds no pair: run_same's ValueError (src/same.py:1003) -- unless the
    cell-type-priority prune is on, whose summary print divides by the number of rows that kept a pair first (src/knn_utils.py:76)."""
    no_pairs = isinstance(dw.error, ValueError) and str(dw.error).startswith("No valid_pairs after KNN filtering")
    if optim_params["ignore_knn_if_matched"] and no_pairs:
        return ZeroDivisionError("division by zero")
    return dw.error


def _device_pairs(dw):
    """The window's pair list with the reference side compacted the way src/utils.py:734-742 compacts it (np.unique of the used
    reference rows): -> (valid_pairs (P, 2) int64, section rows of the compacted reference cells).  The device keeps reference cells
    under their rows in the box (nothing on the device needs the renumbering)."""
    from .windows import _W_PAIRS, _W_ROWS_R

    pairs, rows_r_box = dw.state.fetch(_W_PAIRS), dw.state.fetch(_W_ROWS_R)
    used = np.zeros(len(rows_r_box), bool)
    used[pairs[:, 1]] = True
    valid_pairs = np.empty((len(pairs), 2), np.int64)
    valid_pairs[:, 0] = pairs[:, 0]
    valid_pairs[:, 1] = (np.cumsum(used) - 1)[pairs[:, 1]]
    return valid_pairs, rows_r_box[used]


def _prepared_from_device(dw, frames, optim_params, gurobi_params, verbose=True, vertex_col=None):
    """PreparedInputs of one window from what the device path left (windows.DeviceWindowResult after filter_finish): pairs, costs, kept
    triangles, weights and signs are fetched as arrays; the two frames are rows of the caller's frames, made when first read."""
    from .windows import _W_COSTS, _W_SIGNS, _W_TRIANGLES, _W_WEIGHTS

    st = dw.state
    valid_pairs, rows_r = _device_pairs(dw)
    rows_m = dw.rows_m
    _say(verbose, f"Number of valid pairs after knn: {len(valid_pairs)}")
    costs = st.fetch(_W_COSTS)
    if optim_params["ignore_knn_if_matched"]:
        # the cell-type-priority filter (src/knn_utils.py:28-65) is a sequential walk over the rows: host, on the fetched lists; the
        # frames are not compacted again (:66-78) and every kept pair keeps the cost the device computed for it
        from .knn import priority_filter

        n_r = max(len(rows_r), 1)
        kept, same_type, keep_all = priority_filter(valid_pairs, dw.axy, frames.ref_sec.xy[rows_r],
                                                    frames.moving["cell_type"].to_numpy()[rows_m],
                                                    frames.ref["cell_type"].to_numpy()[rows_r])
        key = valid_pairs[:, 0] * n_r + valid_pairs[:, 1]
        order = np.argsort(key, kind="stable")
        costs = costs[order[np.searchsorted(key[order], kept[:, 0] * n_r + kept[:, 1])]]
        valid_pairs = list(zip(kept[:, 0].tolist(), kept[:, 1].tolist()))
        _say(verbose, f"Total pairs after filtering: {len(valid_pairs)}")
        _say(verbose, f"Average pairs per matched point: {len(valid_pairs) / (same_type + keep_all):.2f}")
    tris = dw.triangles if dw.triangles is not None else st.fetch(_W_TRIANGLES)
    signs = st.fetch(_W_SIGNS).astype(np.float64)
    weights = st.fetch(_W_WEIGHTS)
    if np.issubdtype(frames.mov_sec.size.dtype, np.integer):
        # integer size columns sum to integers in the reference (triangles.triangle_weights_and_signs)
        weights = weights.astype(np.int64)
    prep = PreparedInputs(lambda: _window_frame(frames.moving, rows_m, vertex_col, aligned=True), lambda: _window_frame(frames.ref, rows_r),
                          valid_pairs, costs, tris, weights, signs, set(), False, optim_params, gurobi_params,
                          n_aligned=len(rows_m), n_ref=len(rows_r))
    prep.device, prep.rows_m, prep.rows_r, prep.sources = dw, rows_m, rows_r, (frames.moving, frames.ref)
    return prep


def _staged_from_device(dw, frames, commonCT, optim_params, gurobi_params, caller_triangles, vertex_col, verbose=True):
    """A window whose triangulation is the caller's: rows, prune and compaction from the device, then the frames are made at once
    (the vertex-id remap, the filter with unconstrained-node removal and the re-indexing of src/same.py:1016-1085 read them) and the
    second half runs as `prepare_same_inputs` runs it."""
    st = _Staged()
    st.verbose, st.commonCT = verbose, commonCT
    st.optim_params, st.gurobi_params = optim_params, gurobi_params
    try:
        valid_pairs, rows_r = _device_pairs(dw)
        _say(verbose, f"Number of valid pairs after knn: {len(valid_pairs)}")
        st.aligned_df = _window_frame(frames.moving, dw.rows_m, vertex_col, aligned=True)
        st.ref_df = _window_frame(frames.ref, rows_r)
        if optim_params["ignore_knn_if_matched"]:
            from .knn import priority_filter

            kept, same_type, keep_all = priority_filter(valid_pairs, dw.axy, frames.ref_sec.xy[rows_r],
                                                        st.aligned_df["cell_type"].to_numpy(),
                                                        st.ref_df["cell_type"].to_numpy())
            valid_pairs = list(zip(kept[:, 0].tolist(), kept[:, 1].tolist()))
            _say(verbose, f"Total pairs after filtering: {len(valid_pairs)}")
            _say(verbose, f"Average pairs per matched point: {len(valid_pairs) / (same_type + keep_all):.2f}")
        st.valid_pairs = valid_pairs
        st.caller_triangles = caller_triangles
    except Exception as e:   # noqa: BLE001 -- raised where the serial flow would have raised it (prepare_same_inputs)
        st.error = e
    return st


class ResidentFrames:
    """`resident_frames(ref, moving, ...)`: the two frames of a window job uploaded and binned ONCE, for callers that run several jobs over
    the same frames (a parameter search, bench.py's steps).  Pass it in place of `ref` (or with the same ref / moving objects) to
    sliding_window_matching / sliding_window_incumbent; `close()` (or the context manager) frees the device memory.  It remembers what
    depends on the frames alone -- their cell-type sets, the window plans per (window_size, overlap, min_cells) -- and holds one pair of
    device sections per (commonCT, cost dtype, window grid)."""

    def __init__(self, ref, moving, ctx=None):
        self.ref_arg, self.moving_arg, self.ctx = ref, moving, ctx
        self.type_sets, self.plans, self._frames = {}, {}, {}

    def frames_for(self, job):
        from .windows import window_cell_grid

        why = _DeviceFrames.refusal(job.ref, job.moving, job.commonCT, job.optim_params, job.vertex_col)
        if why is not None:
            raise ValueError(f"these frames cannot be held as device sections: {why}")
        cell_grid = window_cell_grid(job.grid, job.window_size, job.overlap)
        key = (tuple(job.commonCT), str(np.dtype(job.optim_params.get("hip_cost_dtype", "float64"))), cell_grid)
        if key not in self._frames:
            self._frames[key] = _DeviceFrames(job.ref, job.moving, job.commonCT, job.optim_params, cell_grid, ctx=self.ctx)
        frames = self._frames[key]
        frames.op = job.optim_params          # radius, k, penalties ... are per job; the sections do not depend on them
        return frames

    def close(self):
        for f in self._frames.values():
            f.close()
        self._frames = {}

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def resident_frames(ref, moving, ctx=None):
    """See ResidentFrames."""
    return ResidentFrames(ref, moving, ctx=ctx)


class _WindowJob:
    """What sliding_window_matching settles before its loop (src/same.py:297-505): MetaCell unwrapping, parameters, the cell-type check,
    commonCT, the plan, the resume state and this rank's share of the windows."""

    def __init__(self, ref, moving, commonCT, outprefix, moving_delaunay, moving_delaunay_vertex_col, optim_params, gurobi_params,
                 ignore_precomputed_triangulation, shard, resident=None):
        from .windows import window_grid, window_plan

        # the caller's frames are on the device already (resident_frames): both arguments may name it
        if isinstance(ref, ResidentFrames):
            resident = ref
        if isinstance(moving, ResidentFrames):
            resident = moving
        if resident is not None:
            ref = resident.ref_arg if isinstance(ref, ResidentFrames) else ref
            moving = resident.moving_arg if isinstance(moving, ResidentFrames) else moving
            if ref is not resident.ref_arg or moving is not resident.moving_arg:
                raise ValueError("resident frames were made from other ref / moving objects than the ones passed")
        self.resident = resident

        ref_cell_type_col = moving_cell_type_col = "cell_type"
        optim_params = dict(optim_params or {})
        gurobi_params = dict(gurobi_params or {})
        if hasattr(ref, "metacell_df"):
            mc_ref = ref
            ref = mc_ref.metacell_df
            ref_cell_type_col = getattr(mc_ref, "cell_type_col", ref_cell_type_col)
            if (optim_params.get("cell_id_col") is None) and hasattr(mc_ref, "metacell_idx_col"):
                optim_params["cell_id_col"] = mc_ref.metacell_idx_col
        if hasattr(moving, "metacell_df") and hasattr(moving, "metacell_delaunay"):
            mc = moving
            moving = mc.metacell_df
            if moving_delaunay is None and not ignore_precomputed_triangulation:
                moving_delaunay = mc.metacell_delaunay
            if moving_delaunay_vertex_col is None and hasattr(mc, "metacell_idx_col"):
                moving_delaunay_vertex_col = mc.metacell_idx_col
            moving_cell_type_col = getattr(mc, "cell_type_col", moving_cell_type_col)
            if (optim_params.get("cell_id_col") is None) and hasattr(mc, "metacell_idx_col"):
                optim_params["cell_id_col"] = mc.metacell_idx_col
        self.optim_params = optim_params = init_optim_params(**(optim_params or {}))
        self.gurobi_params = gurobi_params = init_gurobi_params(**(gurobi_params or {}))
        self.window_size, self.overlap = optim_params["window_size"], optim_params["overlap"]
        min_cells = optim_params["min_cells_per_window"]

        ref_types = mov_types = None
        if ref_cell_type_col in ref.columns and moving_cell_type_col in moving.columns:
            if resident is not None and resident.type_sets.get((ref_cell_type_col, moving_cell_type_col)) is not None:
                ref_types, mov_types = resident.type_sets[(ref_cell_type_col, moving_cell_type_col)]
            else:
                ref_types = set(pd.Series(ref[ref_cell_type_col]).dropna().unique().tolist())
                mov_types = set(pd.Series(moving[moving_cell_type_col]).dropna().unique().tolist())
                if resident is not None:
                    resident.type_sets[(ref_cell_type_col, moving_cell_type_col)] = (ref_types, mov_types)
            if ref_types != mov_types:
                raise ValueError(
                    "Cell type categories differ between ref and moving.\n"
                    f"ref ({ref_cell_type_col}) has {len(ref_types)} types, moving ({moving_cell_type_col}) has {len(mov_types)} types.\n"
                    f"Only-in-ref: {sorted(ref_types - mov_types)[:20]}\nOnly-in-moving: {sorted(mov_types - ref_types)[:20]}")
        if commonCT is None:
            if ref_types is None:
                raise ValueError("commonCT is None, but cell_type columns were not found to infer it. Pass commonCT explicitly "
                                 f"or ensure both dataframes have '{ref_cell_type_col}'/'{moving_cell_type_col}'.")
            commonCT = sorted(ref_types)
            missing_ref = [c for c in commonCT if c not in ref.columns]
            missing_mov = [c for c in commonCT if c not in moving.columns]
            if missing_ref or missing_mov:
                raise ValueError("commonCT is None so it was inferred as the unique values of the cell_type column, but those "
                                 f"names are not present as probability/one-hot columns.\nMissing in ref columns (first 20): "
                                 f"{missing_ref[:20]}\nMissing in moving columns (first 20): {missing_mov[:20]}")
        self.ref, self.moving, self.commonCT = ref, moving, commonCT
        self.moving_delaunay, self.vertex_col = moving_delaunay, moving_delaunay_vertex_col
        self.ignore_pre = ignore_precomputed_triangulation
        self.caller_triangulation = moving_delaunay is not None and not ignore_precomputed_triangulation

        self.outprefix, self.output_file, self.all_matches = outprefix, None, []
        if outprefix:
            os.makedirs(outprefix, exist_ok=True)
            self.output_file = os.path.join(outprefix, "matchedDF.csv")
        plan_key = (self.window_size, self.overlap, min_cells)
        if resident is not None and plan_key in resident.plans:
            self.plan, self.grid = resident.plans[plan_key]
            plan = self.plan
        else:
            rxy, mxy = ref[["X", "Y"]].to_numpy(dtype=np.float64), moving[["X", "Y"]].to_numpy(dtype=np.float64)
            self.plan = plan = window_plan(rxy, mxy, self.window_size, self.overlap, min_cells)
            self.grid = window_grid(rxy, mxy, self.window_size, self.overlap)[:2] if plan else None
            if resident is not None:
                resident.plans[plan_key] = (plan, self.grid)
        done_ids = set()
        if self.output_file and os.path.exists(self.output_file):  # resume (src/helpers.py:21-70): skip windows already in the file
            existing = pd.read_csv(self.output_file)
            if "window_id" in existing.columns:
                done_ids = set(int(w) for w in existing["window_id"].unique())
                self.all_matches.append(existing)
        self.mine = self.owner = None
        if shard is not None:                    # (rank, world[, deal]): this process runs the windows the deal gives its rank
            from .windows import deal_windows
            self.owner = deal_windows(plan, int(shard[1]), shard[2] if len(shard) > 2 else "block")
            self.mine = set(np.flatnonzero(self.owner == int(shard[0])).tolist())
        self.todo = [(pos, w) for pos, w in enumerate(plan) if w["grid_id"] not in done_ids and (self.mine is None or pos in self.mine)]

    def window_outprefix(self, w):
        return os.path.join(self.outprefix, f"window_{w['window_id']}") if self.outprefix else None

    def device_frames(self, pipeline=None, ctx=None):
        """The two frames as device-resident sections, or None when this job runs the frame pipeline.  -> (frames, whether this job owns
        them: a caller's ResidentFrames stay up when the job is done)."""
        frames = self._device_frames(pipeline, ctx)
        return frames, (frames is not None and self.resident is None)

    def _device_frames(self, pipeline, ctx):
        if self.resident is not None:
            return self.resident.frames_for(self) if self.todo else None
        if window_pipeline(pipeline) != "device" or not self.todo:
            return None
        if _DeviceFrames.refusal(self.ref, self.moving, self.commonCT, self.optim_params, self.vertex_col) is not None:
            return None
        from .windows import window_cell_grid

        return _DeviceFrames(self.ref, self.moving, self.commonCT, self.optim_params,
                             window_cell_grid(self.grid, self.window_size, self.overlap), ctx=ctx)

    def collect(self, pos, w, window_matches):
        """Central trim + bookkeeping of one window's matches (src/same.py:565-590)."""
        if window_matches.shape[0] > 0:
            tx0, tx1, ty0, ty1 = w["trim"]
            x, y = window_matches["X"].to_numpy(), window_matches["Y"].to_numpy()       # the same comparisons on the columns' arrays
            central = window_matches[(x >= tx0) & (x < tx1) & (y >= ty0) & (y < ty1)].copy()
            central["window_id"] = w["window_id"]
            if self.mine is not None:
                central["__plan_pos"] = pos          # lets the sharded wrapper restore the single-process window order
            if len(central) > 0:
                self.all_matches.append(central)
                if self.outprefix:
                    pd.concat(self.all_matches, ignore_index=True).to_csv(self.output_file, index=False)

    def result(self):
        return pd.concat(self.all_matches, ignore_index=True) if self.all_matches else pd.DataFrame()


def iter_prepared_windows(ref, moving, commonCT, plan, optim_params=None, gurobi_params=None, verbose=False, ctx=None, pipeline=None):
    """Pre-MIP artefacts of every window of `plan` (windows.window_plan), in plan order: yields (window, PreparedInputs).
    Window n+1..n+k are staged ahead and triangulated by the Qhull helpers while the consumer works on window n (the same
    pipelining sliding_window_matching uses); a window whose prune leaves no pairs yields (window, the ValueError).
    pipeline: 'device' (default; both frames resident on the GPU, frames of a PreparedInputs made when read) or 'frames'."""
    op, gp = init_optim_params(**dict(optim_params or {})), init_gurobi_params(**dict(gurobi_params or {}))
    if window_pipeline(pipeline) == "device" and len(plan) and _DeviceFrames.refusal(ref, moving, commonCT, op) is None:
        frames = _DeviceFrames(ref, moving, commonCT, op, None, ctx=ctx)
        try:
            for w, dw in zip(plan, frames.windows(plan, ctx=ctx)):
                if dw.error is not None:
                    err = _window_error(dw, op)
                    # only run_same's own ValueError is a window's answer; anything else ends the walk
                    if not isinstance(err, ValueError):
                        raise err
                    yield w, err
                else:
                    yield w, _prepared_from_device(dw, frames, op, gp, verbose=verbose)
        finally:
            frames.close()
        return
    from . import qhull_pool

    depth = qhull_pool.lookahead()
    qhull_pool.warm(min(depth, len(plan)))
    ref_rows, moving_rows = _WindowSubsetter(ref), _WindowSubsetter(moving)
    ahead = {}
    for q, w in enumerate(plan):
        for nxt in range(q, min(q + 1 + depth, len(plan))):
            if nxt not in ahead:
                box = plan[nxt]["box"]
                ahead[nxt] = _stage_prune(ref_rows.subset(*box), moving_rows.subset(*box), commonCT, None, None,
                                          optim_params, gurobi_params, False, verbose, ctx, prefetch=True, fresh_frames=True)
        st = ahead.pop(q)
        try:
            yield w, prepare_same_inputs(None, None, commonCT, verbose=verbose, ctx=ctx, _staged=st)
        except ValueError as e:
            yield w, e


def sliding_window_matching(ref, moving, commonCT=None, outprefix=None, moving_delaunay=None,
                            moving_delaunay_vertex_col=None, optim_params: Optional[Dict[str, Any]] = None,
                            gurobi_params: Optional[Dict[str, Any]] = None,
                            ignore_precomputed_triangulation: bool = False, _run_window=None, _shard=None, _pipeline=None, _solve=None,
                            _job=None):
    """Same contract as src/same.py:297-595.  `_run_window` (testing hook) replaces run_same; `_shard=(rank, world)` makes this
    call process only its share of the window plan (same_amd.dist.sharded_sliding_window_matching); `_pipeline` = 'device' | 'frames'
    (default: $SAME_WINDOW_PIPELINE, else 'device': both frames resident on the GPU for the whole loop, see the note above);
    `_solve(prep: PreparedInputs, outprefix) -> (matches_df, var_out)` stands in for the solver half of run_same (model assembly, solve,
    post-solve tables) behind the unchanged pre-MIP half -- how bench.py times this signature without a Gurobi licence.  `_job`: the
    job as a caller already settled it (dist.sharded_merged_window_matches reads its plan and deal afterwards)."""
    job = _job if _job is not None else _WindowJob(ref, moving, commonCT, outprefix, moving_delaunay, moving_delaunay_vertex_col,
                                                   optim_params,
                                                   gurobi_params, ignore_precomputed_triangulation, _shard)
    frames, own = job.device_frames(_pipeline)
    try:
        if frames is not None:
            _solver_windows_on_device(job, frames, _run_window, _solve)
        else:
            _solver_windows_on_frames(job, _run_window, _solve)
    finally:
        if own:
            frames.close()
    return job.result()


def _solve_window(solve, commonCT, outprefix, job, staged, prepared):
    """The solver half of one window: run_same's own body, or the caller's stand-in on the finished PreparedInputs."""
    op, gp = job.optim_params, job.gurobi_params
    if solve is None:
        return _run_same(None, None, commonCT, outprefix, job.moving_delaunay, job.vertex_col, op, gp, job.ignore_pre, staged, prepared)
    if isinstance(prepared, Exception):
        raise prepared
    prep = prepared if prepared is not None else prepare_same_inputs(None, None, commonCT, verbose=False, _staged=staged)
    return solve(prep, outprefix)


def _solver_windows_on_device(job, frames, run_window, solve=None):
    """The window loop with both frames resident on the device: per window two library calls + the triangulation; the frames a window's
    run_same body reads are made from the device's row lists."""
    commonCT, op, gp = job.commonCT, job.optim_params, job.gurobi_params
    # a stand-in for run_same takes the window's frames: subset_data of both, from the device's row lists
    if run_window is not None:
        from .windows import DeviceWindow

        state = DeviceWindow(frames.ctx)
        try:
            for pos, w in job.todo:
                rows_m, rows_r = frames.box_rows(w["box"], state)
                window_matches, _ = run_window(aligned_df=job.moving.iloc[rows_m], ref_df=job.ref.iloc[rows_r], commonCT=commonCT,
                                               optim_params=op, gurobi_params=gp, outprefix=job.window_outprefix(w),
                                               aligned_delaunay=job.moving_delaunay, aligned_delaunay_vertex_col=job.vertex_col,
                                               ignore_precomputed_triangulation=job.ignore_pre)
                job.collect(pos, w, window_matches)
        finally:
            state.close()
        return
    plan = [w for _pos, w in job.todo]
    for (pos, w), dw in zip(job.todo, frames.windows(plan, triangulate=not job.caller_triangulation)):
        staged = prepared = None
        if dw.error is not None:
            prepared = _window_error(dw, op)                         # raised by the run_same body, where the reference raises it
        elif job.caller_triangulation:
            staged = _staged_from_device(dw, frames, commonCT, op, gp, job.moving_delaunay, job.vertex_col, verbose=solve is None)
        else:
            prepared = _prepared_from_device(dw, frames, op, gp, verbose=solve is None, vertex_col=job.vertex_col)
        window_matches, _ = _solve_window(solve, commonCT, job.window_outprefix(w), job, staged, prepared)
        job.collect(pos, w, window_matches)


def _solver_windows_on_frames(job, run_window, solve=None):
    """The window loop on host frames (the pipeline of rounds 1-4): window n+1..n+k are cut, pruned and compacted while window n is
    still to run, and their triangulations are computed by helper processes meanwhile (qhull_pool); window n then finds its simplices
    ready.  Outputs are unchanged: the same frames reach the same run_same body in the same order."""
    commonCT, op, gp, todo = job.commonCT, job.optim_params, job.gurobi_params, job.todo
    ref_rows, moving_rows = _WindowSubsetter(job.ref), _WindowSubsetter(job.moving)

    def subsets(w):
        return ref_rows.subset(*w["box"]), moving_rows.subset(*w["box"])

    ahead = {}
    depth = 0
    if run_window is None:
        from . import qhull_pool
        depth = qhull_pool.lookahead()
        qhull_pool.warm(min(depth, len(todo)))      # helpers start (import scipy) while the first window is being pruned

    def stage_window(q):
        _pos_q, w_q = todo[q]
        rs, ms = subsets(w_q)
        ahead[q] = (rs, ms, _stage_prune(rs, ms, commonCT, job.moving_delaunay, job.vertex_col, op, gp, job.ignore_pre, True, None,
                                         prefetch=True, fresh_frames=True))

    for q, (pos, w) in enumerate(todo):
        window_outprefix = job.window_outprefix(w)
        if run_window is not None:
            ref_subset, moving_subset = subsets(w)
            window_matches, _ = run_window(aligned_df=moving_subset, ref_df=ref_subset, commonCT=commonCT, optim_params=op,
                                           gurobi_params=gp, outprefix=window_outprefix, aligned_delaunay=job.moving_delaunay,
                                           aligned_delaunay_vertex_col=job.vertex_col, ignore_precomputed_triangulation=job.ignore_pre)
        else:
            for nxt in range(q, min(q + 1 + depth, len(todo))):
                if nxt not in ahead:
                    stage_window(nxt)
            _ref_subset, _moving_subset, staged = ahead.pop(q)
            window_matches, _ = _solve_window(solve, commonCT, window_outprefix, job, staged, None)
        job.collect(pos, w, window_matches)
