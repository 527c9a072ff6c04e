"""Drop-in host layer: run_same / sliding_window_matching with the reference's signatures
(src/same.py:706-716, :297-307), built on the HIP kernels.

`prepare_same_inputs` is everything run_same does before the Gurobi model exists
(src/same.py:933-1189): KNN prune + frame compaction, triangulation + filter, unconstrained
node removal, simplex map / triangle_info, triangle weights, source signs, pair costs.  It is
exposed separately so parity can be tested without a solver licence.  run_same hands those
artefacts to gurobipy unchanged (same variable order, same constraint names) and replaces the
per-incumbent Python loop of _lazy_orientation_callback with the device sweep.
gurobipy is imported lazily: the pre-MIP path and the sweeps work without it.
"""
import os
from typing import Any, Dict, Optional

import numpy as np
import pandas as pd

from . import ops, sweeps, varout
from ._trace import stage
from ._rows import RowList, ValueList, rows_array, values_array
from .cost import pair_costs
from .init_helpers import apply_mip_start
from .knn import find_knn_with_cell_type_priority, find_knn_within_radius
from .params import init_gurobi_params, init_optim_params
from .triangles import (_remap_triangles_by_vertex_ids, build_simplex_map, filter_triangles_by_radius,
                        precompute_coordinate_maps, precompute_triangle_info, triangle_weights_and_signs)


def _say(verbose, *a):
    if verbose:
        print(*a)


class PreparedInputs:
    """Every pre-MIP artefact of run_same, in the shapes Gurobi consumes.

    The flat artefacts (frames, valid_pairs, costs, triangles, weights, signs) are computed eagerly on the
    device path.  The dict-shaped ones the reference builds with per-element Python loops (coordinate maps,
    valid_pairs_map, simplex map, triangle_info: src/helpers.py:164-210, src/same.py:1096-1099) are built on
    first access: the lazy-constraint solve only needs triangle_info (post-solve report), and a window
    pipeline that only wants pairs, costs and sweeps never pays for them."""

    def __init__(self, aligned_df, ref_df, valid_pairs, costs, aligned_delaunay, triangle_weights, source_signs,
                 unconstrained_nodes, using_precomputed, optim_params, gurobi_params, triangles_as_list=True, n_aligned=None, n_ref=None):
        # a frame, or a function that makes it on first access (the device-resident window path: the frames are rows of the caller's
        # frames that only the solver hand-over and the post-solve tables ever read)
        self._aligned_df, self._ref_df = aligned_df, ref_df
        self.valid_pairs = valid_pairs          # (P,2) ndarray, or list of tuples after unconstrained-node removal
        # the list-shaped artefacts are kept as the arrays the kernels produced; the lists the reference hands to the
        # solver (c, aligned_delaunay, triangle_weights, source_signs) are made on first access
        self.costs_array = np.asarray(costs, dtype=np.float64)        # c[idx], same order as valid_pairs
        self.triangles_array = rows_array(aligned_delaunay)           # (Tr, 3); row index = q_tri id
        self.weights_array, self.signs_array = np.asarray(triangle_weights), np.asarray(source_signs, dtype=np.float64)
        self._triangles_as_list = triangles_as_list
        self.unconstrained_nodes, self.using_precomputed = unconstrained_nodes, using_precomputed
        self.n_aligned = len(aligned_df) if n_aligned is None else int(n_aligned)
        self.n_ref = len(ref_df) if n_ref is None else int(n_ref)
        self.optim_params, self.gurobi_params = optim_params, gurobi_params
        self._cache = {}
        self.device = None          # the DeviceWindowResult this was made from (the device-resident window path), else None
        self.sources = None         # there: (the caller's moving frame, reference frame) -- rows_m / rows_r index them

    @property
    def aligned_df(self):
        if callable(self._aligned_df):
            self._aligned_df = self._aligned_df()
        return self._aligned_df

    @property
    def ref_df(self):
        if callable(self._ref_df):
            self._ref_df = self._ref_df()
        return self._ref_df

    def _listed(self, key, make):
        if key not in self._cache:
            self._cache[key] = make()
        return self._cache[key]

    @property
    def costs(self):
        return self._listed("costs", lambda: ValueList(self.costs_array))

    @property
    def aligned_delaunay(self):          # list of 3-int rows (an array after unconstrained-node removal, as in the reference flow)
        return self._listed("tris", lambda: (RowList(self.triangles_array) if len(self.triangles_array) else [])
                            if self._triangles_as_list else self.triangles_array)

    @property
    def triangle_weights(self):
        return self._listed("weights", lambda: ValueList(self.weights_array))

    @property
    def source_signs(self):
        return self._listed("signs", lambda: ValueList(self.signs_array))

    def _maps(self):
        if "maps" not in self._cache:
            self._cache["maps"] = precompute_coordinate_maps(self.aligned_df, self.ref_df, self.valid_pairs)
        return self._cache["maps"]

    @property
    def aligned_coords(self):
        return self._maps()[0]

    @property
    def ref_coords(self):
        return self._maps()[1]

    @property
    def valid_pairs_map(self):
        return self._maps()[2]

    @property
    def aligned_simplex_map(self):
        if "simplex" not in self._cache:
            self._cache["simplex"] = build_simplex_map(self.n_aligned, self.aligned_delaunay)
        return self._cache["simplex"]

    @property
    def triangle_info(self):
        if "tinfo" not in self._cache:
            self._cache["tinfo"] = precompute_triangle_info(self.aligned_df, self.aligned_delaunay, self.aligned_simplex_map)
        return self._cache["tinfo"]

    @property
    def ref_coords_xy(self):                    # model._ref_coords (src/same.py:1158)
        if "rxy" not in self._cache:
            rxy = self.ref_df[["X", "Y"]].to_numpy(dtype=np.float64)
            self._cache["rxy"] = {j: (rxy[j, 0], rxy[j, 1]) for j in range(len(rxy))}
        return self._cache["rxy"]


class _Staged:
    """A window after the first half of the pre-MIP path (frames compacted by the prune, pairs known) with its
    triangulation either supplied by the caller or under way in a Qhull helper (qhull_pool).  Produced ahead of time by the
    window loop so that window n+1 is triangulated while window n runs; an error met on the way is kept and raised when
    the window is actually run, exactly where the serial flow would have raised it."""

    __slots__ = ("error", "aligned_df", "ref_df", "valid_pairs", "optim_params", "gurobi_params", "commonCT", "caller_triangles",
                 "ticket", "verbose")

    def __init__(self):
        for name in self.__slots__:
            setattr(self, name, None)


def _stage_prune(ref_df, aligned_df, commonCT, aligned_delaunay, aligned_delaunay_vertex_col, optim_params, gurobi_params,
                 ignore_precomputed_triangulation, verbose, ctx, prefetch, fresh_frames=False):
    """fresh_frames: the two frames were just cut out for this call (a window's subsets) and belong to nobody else, so the
    helper columns can be added to a shallow copy instead of a deep one."""
    st = _Staged()
    st.verbose, st.commonCT = verbose, commonCT
    try:
        _stage_prune_body(st, ref_df, aligned_df, aligned_delaunay, aligned_delaunay_vertex_col, optim_params, gurobi_params,
                          ignore_precomputed_triangulation, verbose, ctx, prefetch, fresh_frames)
    except Exception as e:   # noqa: BLE001 -- re-raised unchanged by prepare_same_inputs when the window is run
        st.error = e
    return st


def _stage_prune_body(st, ref_df, aligned_df, aligned_delaunay, aligned_delaunay_vertex_col, optim_params, gurobi_params,
                      ignore_precomputed_triangulation, verbose, ctx, prefetch, fresh_frames=False):
    optim_params = dict(optim_params or {})
    gurobi_params = dict(gurobi_params or {})
    # MetaCell duck-typing (src/same.py:891-899)
    if hasattr(aligned_df, "metacell_df") and hasattr(aligned_df, "metacell_delaunay"):
        mc = aligned_df
        aligned_df = mc.metacell_df
        if aligned_delaunay is None and not ignore_precomputed_triangulation:
            aligned_delaunay = mc.metacell_delaunay
        if aligned_delaunay_vertex_col is None and hasattr(mc, "metacell_idx_col"):
            aligned_delaunay_vertex_col = mc.metacell_idx_col
        if (optim_params.get("cell_id_col") is None) and hasattr(mc, "metacell_idx_col"):
            optim_params["cell_id_col"] = mc.metacell_idx_col
    st.optim_params = optim_params = init_optim_params(**optim_params)
    st.gurobi_params = init_gurobi_params(**gurobi_params)
    radius, knn = optim_params["radius"], optim_params["knn"]

    # size defaults, stable ids (src/same.py:934-970)
    # only columns are ADDED below (size, __orig_idx, __tri_vid), never edited, so frames nobody else holds need no deep copy
    aligned_df = aligned_df.copy(deep=not fresh_frames)
    ref_df = ref_df.copy(deep=not fresh_frames)
    if "size" not in aligned_df.columns:
        aligned_df["size"] = 1
    if "size" not in ref_df.columns:
        ref_df["size"] = 1
    if "__orig_idx" not in aligned_df.columns:
        aligned_df["__orig_idx"] = aligned_df.index.to_numpy()
    if "__orig_idx" not in ref_df.columns:
        ref_df["__orig_idx"] = ref_df.index.to_numpy()
    if aligned_delaunay_vertex_col is None:
        aligned_df["__tri_vid"] = aligned_df.index.to_numpy()
    else:
        if aligned_delaunay_vertex_col not in aligned_df.columns:
            raise ValueError(f"aligned_delaunay_vertex_col='{aligned_delaunay_vertex_col}' not in aligned_df")
        aligned_df["__tri_vid"] = aligned_df[aligned_delaunay_vertex_col].to_numpy()

    # KNN prune (src/same.py:972-979)
    with stage("prune+compact"):
        if optim_params["ignore_knn_if_matched"]:
            aligned_df, ref_df, valid_pairs = find_knn_with_cell_type_priority(aligned_df, ref_df, radius, knn=knn, verbose=verbose,
                                                                      ctx=ctx)
        else:
            aligned_df, ref_df, valid_pairs = find_knn_within_radius(aligned_df, ref_df, radius, knn=knn, verbose=verbose, ctx=ctx)
    st.aligned_df, st.ref_df, st.valid_pairs = aligned_df, ref_df, valid_pairs
    if len(valid_pairs) == 0:
        raise ValueError("No valid_pairs after KNN filtering. Increase radius and/or knn.")
    # triangulation (src/same.py:1016-1031): the caller's, or Qhull on the compacted aligned cells -- started now in a
    # helper process when the window loop is running ahead, picked up in prepare_same_inputs
    if aligned_delaunay is None or ignore_precomputed_triangulation:
        if prefetch:
            from . import qhull_pool

            st.ticket = qhull_pool.pool().submit(aligned_df[["X", "Y"]].values)
    else:
        st.caller_triangles = aligned_delaunay


def prepare_same_inputs(ref_df, aligned_df, commonCT, aligned_delaunay=None, aligned_delaunay_vertex_col=None,
                        optim_params=None, gurobi_params=None, ignore_precomputed_triangulation=False,
                        verbose=True, ctx=None, _staged=None) -> PreparedInputs:
    """Everything run_same computes before it talks to the solver (src/same.py:933-1189).  `_staged`: the first half done
    ahead of time by the window loop (`_stage_prune`); the other arguments are then ignored."""
    from scipy.spatial import Delaunay  # Qhull stays on the host: it is an input to the kernels

    st = _staged if _staged is not None else _stage_prune(ref_df, aligned_df, commonCT, aligned_delaunay, aligned_delaunay_vertex_col,
                                                          optim_params, gurobi_params, ignore_precomputed_triangulation, verbose,
                                                          ctx, prefetch=False)
    if st.error is not None:
        raise st.error
    aligned_df, ref_df, valid_pairs, commonCT = st.aligned_df, st.ref_df, st.valid_pairs, st.commonCT
    optim_params, gurobi_params = st.optim_params, st.gurobi_params
    radius = optim_params["radius"]
    dist_ct_coeff = optim_params["dist_ct_coeff"]
    min_angle_deg = optim_params.get("min_angle_deg", 15)
    n_aligned, n_ref = len(aligned_df), len(ref_df)

    aligned_coords_array = aligned_df[["X", "Y"]].values
    using_precomputed = st.caller_triangles is not None
    with stage("triangulate (qhull / remap / wait for helper)"):
        if using_precomputed:
            aligned_delaunay = _remap_triangles_by_vertex_ids(st.caller_triangles, vertex_ids=aligned_df["__tri_vid"].to_numpy())
        elif st.ticket is not None:
            aligned_delaunay = st.ticket.result()
        else:
            aligned_delaunay = Delaunay(aligned_coords_array).simplices

    # filter (src/same.py:1033-1053)
    unconstrained_nodes = set()
    with stage("triangle filter"):
        if using_precomputed:
            aligned_delaunay, unconstrained_nodes = filter_triangles_by_radius(
                aligned_coords_array, aligned_delaunay, radius, aligned_df=aligned_df,
                ignore_same_type_triangles=optim_params["ignore_same_type_triangles"], remove_unconstrained_nodes=True,
                min_angle_deg=min_angle_deg, verbose=verbose, ctx=ctx, _rows_as_array=True)
        else:
            aligned_delaunay = filter_triangles_by_radius(
                aligned_coords_array, aligned_delaunay, radius, aligned_df=aligned_df,
                ignore_same_type_triangles=optim_params["ignore_same_type_triangles"], min_angle_deg=min_angle_deg,
                verbose=verbose, ctx=ctx, _rows_as_array=True)

    # unconstrained-node removal + re-index (src/same.py:1055-1085)
    triangles_as_list = True
    if unconstrained_nodes:
        triangles_as_list = False
        _say(verbose, f"\nRemoving {len(unconstrained_nodes)} unconstrained nodes from optimization...")
        vp = np.asarray(valid_pairs, dtype=np.int64).reshape(-1, 2)
        keep_node = np.ones(len(aligned_df), bool)
        keep_node[list(unconstrained_nodes)] = False
        constrained_nodes = np.flatnonzero(keep_node)
        old_to_new = np.full(len(aligned_df), -1, np.int64)
        old_to_new[constrained_nodes] = np.arange(len(constrained_nodes))
        vp = vp[keep_node[vp[:, 0]]]
        valid_pairs = [(int(old_to_new[i]), int(j)) for i, j in vp]
        tri = rows_array(aligned_delaunay)
        tri = tri[keep_node[tri].all(axis=1)] if len(tri) else tri
        aligned_delaunay = old_to_new[tri] if len(tri) else np.array([]).reshape(0, 3)
        aligned_df = aligned_df.iloc[constrained_nodes].reset_index(drop=True)

    with stage("triangle weights + source signs"):
        triangle_weights, source_signs = triangle_weights_and_signs(aligned_df, aligned_delaunay, ctx=ctx, _as_arrays=True)
    # build-only key (not among init_optim_params' defaults, which stay the reference's): fp32 pair costs, BASELINE config 5
    cost_dtype = np.dtype(optim_params.get("hip_cost_dtype", "float64"))
    if cost_dtype not in (np.dtype(np.float64), np.dtype(np.float32)):
        raise ValueError(f"hip_cost_dtype must be 'float64' or 'float32', got {optim_params['hip_cost_dtype']!r}")
    with stage("pair costs"):
        costs = pair_costs(aligned_df, ref_df, valid_pairs, list(commonCT), dist_ct_coeff, ctx=ctx, dtype=cost_dtype, _as_array=True)
    return PreparedInputs(aligned_df, ref_df, valid_pairs, costs, aligned_delaunay, triangle_weights, source_signs,
                          unconstrained_nodes, using_precomputed, optim_params, gurobi_params, triangles_as_list=triangles_as_list)


# ------------------------------------------------------------------------------------------ callback
def make_lazy_callback(GRB, sweep):
    """cb(model, where) with the control flow of _lazy_orientation_callback (src/same.py:621-703);
    the per-triangle loop is one device sweep."""

    def _lazy_orientation_callback(model, where):
        if where != GRB.Callback.MIPSOL:
            return
        max_cuts = getattr(model, "_lazy_max_cuts", None)
        if max_cuts is not None and model._cuts_added >= max_cuts:
            return
        # ask the solver for a flat list in pair order (a list query returns a list: one C call, no dict walk)
        x_list = getattr(model, "_x_list", None)
        if x_list is None:
            x_list = model._x_list = [model._x[i] for i in range(len(model._valid_pairs))]
        x_arr = np.asarray(model.cbGetSolution(x_list), dtype=np.float64)
        remaining = None if max_cuts is None else max(0, max_cuts - model._cuts_added)
        cuts = sweep.select_cuts(x_arr, getattr(model, "_lazy_allowed_flip_fraction", None),
                                 getattr(model, "_lazy_max_cuts_per_incumbent", None), remaining)
        for tri_idx, pa, pb, pc in cuts:
            model.cbLazy(model._x[pa] + model._x[pb] + model._x[pc] <= 2 + model._q_tri[tri_idx])
            model._cuts_added += 1

    return _lazy_orientation_callback


def _load_gurobi_config():
    """KEY=VALUE licence file next to the package, or GUROBI_* environment (src/same.py:598-618)."""
    config = {}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".gurobienv")
    try:
        with open(path, "r") as f:
            for line in f:
                line = line.strip()
                if line and not line.startswith("#") and "=" in line:
                    k, v = line.split("=", 1)
                    config[k.strip()] = v.strip()
    except OSError:
        pass
    return config


# ------------------------------------------------------------------------------------------ run_same
def run_same(ref_df, aligned_df, commonCT, outprefix=None, aligned_delaunay=None, aligned_delaunay_vertex_col=None,
             optim_params: Optional[Dict[str, Any]] = None, gurobi_params: Optional[Dict[str, Any]] = None,
             ignore_precomputed_triangulation: bool = False):
    """Same contract as src/same.py:706-1489: returns (matches_df, var_out)."""
    return _run_same(ref_df, aligned_df, commonCT, outprefix, aligned_delaunay, aligned_delaunay_vertex_col, optim_params,
                     gurobi_params, ignore_precomputed_triangulation, None)


def _run_same(ref_df, aligned_df, commonCT, outprefix, aligned_delaunay, aligned_delaunay_vertex_col, optim_params, gurobi_params,
              ignore_precomputed_triangulation, staged, prepared=None):
    """run_same; `staged` = the window's first half, done ahead of time by the window loop (None: do it now); `prepared` = ALL of
    the pre-MIP path done already (the device-resident window path), or the exception it ended in."""
    try:
        import gurobipy as gp
        from gurobipy import GRB, Model, quicksum
    except ImportError as e:  # the solver is proprietary; everything before it is available without it
        raise ImportError("run_same needs gurobipy for the MIP solve; use prepare_same_inputs() for the pre-MIP "
                          "artefacts and same_amd.sweeps for the violation sweeps") from e

    log_dir = os.path.join(os.getcwd(), "gurobi_logs")
    os.makedirs(log_dir, exist_ok=True)
    cfg = _load_gurobi_config()
    options = {
        "WLSACCESSID": os.environ.get("GUROBI_WLSACCESSID", "") or cfg.get("WLSACCESSID", ""),
        "WLSSECRET": os.environ.get("GUROBI_WLSSECRET", "") or cfg.get("WLSSECRET", ""),
        "LICENSEID": int(os.environ.get("GUROBI_LICENSEID", 0)) or int(cfg.get("LICENSEID", 0)),
        "OutputFlag": 1,
        "LogFile": os.path.join(log_dir, f"gurobi_{os.getpid()}.log"),
    }
    try:
        env = gp.Env(params=options)
        if isinstance(prepared, Exception):
            raise prepared
        prep = prepared if prepared is not None else prepare_same_inputs(
            ref_df, aligned_df, commonCT, aligned_delaunay, aligned_delaunay_vertex_col, optim_params, gurobi_params,
            ignore_precomputed_triangulation, _staged=staged)
        op, gpar = prep.optim_params, prep.gurobi_params
        if len(prep.valid_pairs) == 0:
            # every node was unconstrained under the caller's triangulation, so nothing is left to match.  (The reference
            # builds and "solves" the empty model and then dies with IndexError at src/same.py:1253; the no-solution return
            # of :1474-1477 is the usable form of the same outcome, and lets sliding windows carry on.)
            print("No valid_pairs left after removing unconstrained nodes; nothing to optimise")
            out_df = pd.DataFrame()
            if outprefix:
                os.makedirs(outprefix, exist_ok=True)
                out_df.to_csv(os.path.join(outprefix, "matches_df.csv"), index=False)
            return out_df, {}
        aligned_df, ref_df = prep.aligned_df, prep.ref_df
        valid_pairs, c, tris = prep.valid_pairs, prep.costs_array, prep.triangles_array   # same values as the reference's lists
        n_aligned, n_ref = prep.n_aligned, prep.n_ref
        lazy = op["lazy_constraints"]
        cell_id_col = op["cell_id_col"]

        model = Model("optimal_matches", env=env)
        x = model.addVars(len(valid_pairs), vtype=GRB.BINARY, lb=0, ub=1, name="x")
        penalty_vars = model.addVars(n_ref, vtype=GRB.CONTINUOUS, lb=0, ub=1000, name="penalty")
        no_match_vars = model.addVars(n_aligned, vtype=GRB.CONTINUOUS, lb=0, ub=1, name="no_match")
        model.update()
        _add_basic_constraints(model, quicksum, valid_pairs, op["max_matches"], x, penalty_vars, no_match_vars, ref_df,
                               op["ref_metacell_match_multiplier"])

        if lazy:
            q_tri = model.addVars(len(tris), vtype=GRB.CONTINUOUS, lb=0, name="q_tri")
            model.update()
            model._x, model._q_tri, model._valid_pairs = x, q_tri, valid_pairs
            model._aligned_delaunay, model._source_signs, model._ref_coords = tris, prep.signs_array, prep.ref_coords_xy
            model._cuts_added = 0
            model._lazy_max_cuts = gpar["lazy_max_cuts"]
            model._lazy_allowed_flip_fraction = gpar["lazy_allowed_flip_fraction"]
            model._lazy_max_cuts_per_incumbent = gpar["lazy_max_cuts_per_incumbent"]
            model.Params.LazyConstraints = 1
            model.Params.Method = gp.GRB.METHOD_PDHG
            model.Params.PDHGGPU = 1
            area_penalty_vars = [q_tri[i] for i in range(len(tris))]
        else:
            print("Using EAGER constraint generation (O(n*k^3) memory)")
            area_penalty_vars, _ = _add_spatial_constraints_eager(model, GRB, x, prep)
        model.update()

        sizes = aligned_df["size"].to_numpy()
        model.setObjective(
            quicksum(c[idx] * x[idx] for idx in range(len(valid_pairs)))
            + op["penalty_coeff"] * quicksum(penalty_vars[j] for j in range(n_ref))
            + op["no_match_penalty"] * quicksum(sizes[i] * no_match_vars[i] for i in range(n_aligned))
            + op["delaunay_penalty"] * quicksum(prep.weights_array[idx] * v for idx, v in enumerate(area_penalty_vars)),
            GRB.MINIMIZE)
        with stage("MIP start"):
            apply_mip_start(x_vars=x, no_match_vars=no_match_vars, valid_pairs=valid_pairs, costs=c, n_aligned=n_aligned,
                            n_ref=n_ref, aligned_sizes=aligned_df["size"].to_numpy(dtype=float),
                            no_match_penalty=op["no_match_penalty"], max_matches=op["max_matches"],
                            init_method=gpar["init_method"], init_big_m=gpar["init_big_m"],
                            init_hungarian_max_n=gpar["init_hungarian_max_n"], verbose=True)

        if outprefix:
            os.makedirs(outprefix, exist_ok=True)
            model_file = os.path.join(outprefix, "matching_model.lp")
        else:
            model_file = "matching_model.lp"
        model.write(model_file)
        model.Params.timeLimit = float(gpar["time_limit"]) if gpar["time_limit"] is not None else float("inf")
        model.Params.MIPGap = float(gpar["mip_gap"])
        if gpar["mip_focus"] is not None:
            model.Params.MIPFocus = int(gpar["mip_focus"])
        if gpar["cuts"] is not None:
            model.Params.Cuts = int(gpar["cuts"])
        if gpar["heuristics"] is not None:
            model.Params.Heuristics = float(gpar["heuristics"])

        with stage("solve (incl. lazy sweeps)"):
            if lazy:
                sweep = sweeps.LazyOrientationSweep(valid_pairs, tris, prep.signs_array, ref_df[["X", "Y"]].to_numpy(dtype=np.float64),
                                                    n_aligned)
                model.optimize(make_lazy_callback(GRB, sweep))
                print(f"Lazy cuts added: {model._cuts_added}")
            else:
                model.optimize()
        time_limit_reached = model.status == GRB.TIME_LIMIT
        solve_time = model.Runtime

        if model.status == GRB.OPTIMAL or model.status == GRB.TIME_LIMIT:
            with stage("post-solve sweeps + tables"):
                out_df, var_out = _post_solve(prep, commonCT, x, no_match_vars, penalty_vars, area_penalty_vars, model,
                                              cell_id_col, time_limit_reached, solve_time, outprefix)
        else:
            print("No optimal solution found")
            out_df, var_out = pd.DataFrame(), {}
        if outprefix:
            out_df.to_csv(os.path.join(outprefix, "matches_df.csv"), index=False)
        return out_df, var_out
    finally:
        try:
            if os.path.exists(log_dir) and not os.listdir(log_dir):
                os.rmdir(log_dir)
        except OSError:
            pass


def _add_basic_constraints(model, quicksum, valid_pairs, max_matches, x, penalty_vars, no_match_vars, ref_df, multiplier):
    """Assignment constraints in the order and with the names of src/helpers.py:102-161."""
    pairs = np.asarray(valid_pairs, dtype=np.int64).reshape(-1, 2)
    by_ref, by_aligned = {}, {}
    for idx, (ip, jp) in enumerate(pairs.tolist()):
        by_ref.setdefault(jp, []).append(idx)
        by_aligned.setdefault(ip, []).append(idx)
    rsize = ref_df["size"].to_numpy() if "size" in ref_df.columns else None
    has_meta = rsize is not None and bool((rsize > 1).any())
    if has_meta and multiplier is None:
        multiplier = int(rsize.max())
    for j, idxs in by_ref.items():
        limit = multiplier * max_matches if (has_meta and rsize[j] > 1) else max_matches
        model.addConstr(quicksum(x[i] for i in idxs) <= limit, name=f"max_matches_{j}")
    model.update()
    for i, idxs in by_aligned.items():
        model.addConstr(quicksum(x[q] for q in idxs) <= 1, name=f"one_match_{i}")
    model.update()
    for j, idxs in by_ref.items():
        model.addConstr(quicksum(x[i] for i in idxs) - penalty_vars[j] <= 1, name=f"penalty_{j}")
    model.update()
    for i, idxs in by_aligned.items():
        model.addConstr(quicksum(x[q] for q in idxs) + no_match_vars[i] == 1, name=f"no_match_{i}")
    model.update()


def _add_spatial_constraints_eager(model, GRB, x, prep):
    """lazy_constraints=False: the variables and constraints of add_spatial_constraints_triangle_based
    (src/helpers.py:444-573), in its order and with its names.  Per triangle one `area_penalty_tri{t}_{p1}_{p2}_{p3}`
    and, for every combination (idx1, idx2, idx3) of its vertices' candidate pairs, `z_tri{t}_{idx1}_{idx2}_{idx3}` with
    z <= x1, z <= x2, z <= x3, z >= x1 + x2 + x3 - 2 and (aligned sign * ref sign) * z >= -area_penalty.  Both signs are
    sign(round(cross, 3)) (signed_area_terms :398-411, calc_ref_area :425-441); the Tr*k^3 reference signs come from one
    `same_eager_signs` launch instead of the reference's process pool.  Building the Python constraint objects stays a
    host loop -- that is the solver's API."""
    pairs = np.asarray(prep.valid_pairs, dtype=np.int64).reshape(-1, 2)
    tris = np.asarray(prep.triangles_array, dtype=np.int64)
    n_aligned = prep.n_aligned
    order = np.argsort(pairs[:, 0], kind="stable")          # valid_pairs_imap: pair indices per aligned row, in pair order
    counts = np.bincount(pairs[:, 0], minlength=n_aligned)
    k = max(int(counts.max()) if len(counts) else 0, 1)
    starts = np.concatenate(([0], np.cumsum(counts)))[:-1]
    slot = np.arange(len(pairs)) - np.repeat(starts, counts)
    cand_ref = np.full((n_aligned, k), -1, np.int32)
    cand_pair = np.full((n_aligned, k), -1, np.int64)
    cand_ref[pairs[order, 0], slot] = pairs[order, 1]
    cand_pair[pairs[order, 0], slot] = order
    axy = prep.aligned_df[["X", "Y"]].to_numpy(dtype=np.float64)
    rxy = prep.ref_df[["X", "Y"]].to_numpy(dtype=np.float64)
    ref_sign = ops.eager_signs(rxy, tris, cand_ref)                                            # (Tr, k, k, k), 2 = no such combination
    aligned_sign = ops.eager_signs(axy, tris, np.arange(n_aligned, dtype=np.int32)[:, None]).reshape(-1)

    area_penalty_vars, z_penalty_vars, all_constraints = [], [], []
    for t, (p1, p2, p3) in enumerate(tris.tolist()):
        a_sign = int(aligned_sign[t])
        apv = model.addVar(vtype=GRB.CONTINUOUS, lb=0, name=f"area_penalty_tri{t}_{p1}_{p2}_{p3}")
        area_penalty_vars.append(apv)
        for s1 in range(counts[p1]):
            idx1 = int(cand_pair[p1, s1])
            for s2 in range(counts[p2]):
                idx2 = int(cand_pair[p2, s2])
                for s3 in range(counts[p3]):
                    idx3 = int(cand_pair[p3, s3])
                    z = model.addVar(vtype=GRB.CONTINUOUS, lb=0, ub=1, name=f"z_tri{t}_{idx1}_{idx2}_{idx3}")
                    z_penalty_vars.append(z)
                    all_constraints.append(z <= x[idx1])
                    all_constraints.append(z <= x[idx2])
                    all_constraints.append(z <= x[idx3])
                    all_constraints.append(z >= x[idx1] + x[idx2] + x[idx3] - 2)
                    all_constraints.append(a_sign * int(ref_sign[t, s1, s2, s3]) * z >= -apv)
    print(f"Adding {len(all_constraints)} constraints to model...")
    model.addConstrs((constraint for constraint in all_constraints))
    model.update()
    return area_penalty_vars, z_penalty_vars


def _post_solve(prep, commonCT, x, no_match_vars, penalty_vars, area_penalty_vars, model, cell_id_col,
                time_limit_reached, solve_time, outprefix):
    """Match table + violation analysis (src/same.py:1258-1472)."""
    aligned_df, ref_df, valid_pairs, tris = prep.aligned_df, prep.ref_df, prep.valid_pairs, prep.aligned_delaunay
    n_pairs = len(valid_pairs)
    xv = np.fromiter((x[i].x for i in range(n_pairs)), dtype=np.float64, count=n_pairs)
    pairs = np.asarray(valid_pairs, dtype=np.int64).reshape(-1, 2)
    sel = np.flatnonzero(xv > 0.5)
    out_df = pd.DataFrame([(int(i), int(j)) for i, j in pairs[sel]], columns=["aligned_idx", "ref_idx"])
    for ct in list(commonCT) + ["X", "Y"]:
        out_df[ct] = out_df["aligned_idx"].map(aligned_df[ct])
    for ct in ["X", "Y"]:
        out_df[f"ref_{ct}"] = out_df["ref_idx"].map(ref_df[ct])
    out_df["size"] = out_df["aligned_idx"].map(aligned_df["size"])
    out_df["ref_size"] = out_df["ref_idx"].map(ref_df["size"])
    out_df[f"Ref_{cell_id_col}"] = out_df["ref_idx"].map(ref_df[cell_id_col])
    out_df[f"Aligned_{cell_id_col}"] = out_df["aligned_idx"].map(aligned_df[cell_id_col])
    out_df["time_limit_reached"] = time_limit_reached

    violations = sweeps.verify_spatial_preservation(aligned_df=aligned_df, ref_df=ref_df, matches_df=out_df,
                                                    triangle_info=prep.triangle_info)
    sweeps.print_violation_report(violations)

    violation_points = set(violations["points_with_violations"])
    penalty_points = set()
    for idx, var in enumerate(area_penalty_vars):
        if var.x > 1e-6:
            for p in tris[idx]:
                penalty_points.add(p)
    points_both = violation_points & penalty_points

    aligned_to_ref = {int(i): int(j) for i, j in pairs[sel]}  # later pairs win (src/same.py:1370-1373)
    before, after, flipped, matched_vertices = sweeps.triangle_area_flips(aligned_df, ref_df, tris, aligned_to_ref)

    var_out = {
        "x": list(xv),
        "no_match_vars": [no_match_vars[i].x for i in range(prep.n_aligned)],
        "penalty_vars": [penalty_vars[j].x for j in range(prep.n_ref)],
        "area_penalty_vars": [v.x for v in area_penalty_vars],
        "violations": violations,
        "violation_penalty_comparison": {"points_both": list(points_both),
                                         "points_only_violations": list(violation_points - penalty_points),
                                         "points_only_penalties": list(penalty_points - violation_points)},
        "triangle_data": {"triangles": list(tris) if isinstance(tris, list) else tris,   # a plain list of rows, as the reference stores
                          "triangle_info": prep.triangle_info,
                          "aligned_simplex_map": prep.aligned_simplex_map, "areas_before": before, "areas_after": after,
                          "flipped_triangles": flipped, "matched_vertices": matched_vertices},
        "lazy_constraints": bool(prep.optim_params["lazy_constraints"]),
        "lazy_cuts_added": model._cuts_added if prep.optim_params["lazy_constraints"] else 0,
    }
    if outprefix:
        # the reference pickles this dict into var_out.npy (src/same.py:1455-1462) and its own reader and notebooks open that
        # file (src/helpers.py:682), so a drop-in writes it too; beside it goes a JSON + npz pair that loads without executing
        # anything (varout.py), which is what load_matching_results here prefers.  SAME_LEGACY_VAR_OUT=0 leaves the pickle out.
        varout.save(outprefix, var_out)
        if os.environ.get("SAME_LEGACY_VAR_OUT", "1") != "0":
            np.save(os.path.join(outprefix, "var_out.npy"), var_out, allow_pickle=True)
        aligned_df.to_csv(os.path.join(outprefix, "aligned_df.csv"), index=False)
        ref_df.to_csv(os.path.join(outprefix, "ref_df.csv"), index=False)
    flipped_nodes = set()
    for t in flipped:
        for v in tris[t]:
            flipped_nodes.add(v)
    out_df["triangle_violation"] = out_df["aligned_idx"].isin(flipped_nodes)
    out_df["filtered_violation"] = out_df["aligned_idx"].isin(points_both)
    out_df["run_time"] = solve_time
    return out_df, var_out


# the window API (sliding_window_matching, iter_prepared_windows, resident_frames, the two pipelines behind them) lives in
# same_amd/window_api.py; its names stay importable from here
def __getattr__(name):
    from . import window_api

    if hasattr(window_api, name):
        return getattr(window_api, name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
