"""Flatten what run_same / sliding_window_matching return -- and the model the solver double received -- into plain arrays.

Used twice with the SAME solver double (tests/fake_gurobipy.py): by tools/gen_golden.py on the reference's run_same
(build container only) to write tests/golden/run_same_mock.npz, and by the GPU tests on same_amd.run_same to compare.
Everything order-sensitive in the reference's outputs is kept in order (variables, constraints, cuts, match rows,
violation lists, triangle_info insertion order); the two containers the reference builds from Python sets
(triangles_with_violations, points_with_violations and the three comparison lists) are compared sorted."""
import numpy as np
import pandas as pd

import fake_gurobipy as fg

_SENSE = {"<=": -1, "==": 0, ">=": 1}


def _col(a):
    a = np.asarray(a)
    if a.dtype == bool:
        return a.astype(np.uint8)
    if a.dtype == object:
        return a.astype(str)
    return a


def _constraints(prefix, constrs, index):
    canon = fg.canonical_constraints(constrs)
    width = max([len(c[2]) for c in canon] + [1])
    tv, tc = np.full((len(canon), width), -1, np.int32), np.zeros((len(canon), width))
    for q, c in enumerate(canon):
        for t, (n, k) in enumerate(c[2]):
            tv[q, t], tc[q, t] = index[n], k
    return {f"{prefix}_names": np.array([n or "" for n, _ in constrs], dtype=str),
            f"{prefix}_sense": np.array([_SENSE[c[0]] for c in canon], dtype=np.int8),
            f"{prefix}_const": np.array([c[1] for c in canon], dtype=np.float64), f"{prefix}_term_var": tv, f"{prefix}_term_coef": tc}


def record_model(model):
    names = [v.VarName for v in model.vars]
    index = {n: i for i, n in enumerate(names)}
    out = {"var_names": np.array(names, dtype=str), "var_lb": np.array([float(v.lb) for v in model.vars]),
           "var_ub": np.array([np.inf if v.ub is None else float(v.ub) for v in model.vars]),
           "var_start": np.array([np.nan if v.Start is None else float(v.Start) for v in model.vars])}
    out.update(_constraints("con", model.constrs, index))
    out.update(_constraints("cut", [(None, c) for c in model.lazy], index))
    obj = np.zeros(len(names))
    for v, k in model.objective.terms.items():
        obj[index[v.VarName]] = k
    out["objective"] = obj
    out["objective_const"] = np.array([model.objective.const])
    out["params"] = np.array(sorted(f"{k}={v}" for k, v in vars(model.Params).items()), dtype=str)
    return out


def record_violations(v):
    def rows(lst):
        return np.array([(d["triangle_idx"], d["point1"]["aligned_idx"], d["point2"]["aligned_idx"],
                          d["point1"]["ref_idx"], d["point2"]["ref_idx"]) for d in lst], dtype=np.int64).reshape(-1, 5)
    s = v["violation_summary"]
    return {"viol_x": rows(v["x_order_violations"]), "viol_y": rows(v["y_order_violations"]),
            "viol_tris": np.array(sorted(int(t) for t in v["triangles_with_violations"]), dtype=np.int64),
            "viol_points": np.array(sorted(int(p) for p in v["points_with_violations"]), dtype=np.int64),
            "viol_summary": np.array([s["total_triangles"], s["violated_triangles"], s["total_comparisons"], s["total_violations"]], dtype=np.int64),
            "viol_percent": np.array([s["percent_triangles_violated"], s["percent_violations"]], dtype=np.float64)}


def record_frame(prefix, df):
    out = {f"{prefix}_columns": np.array(list(df.columns), dtype=str)}
    for c in df.columns:
        out[f"{prefix}__{c}"] = _col(df[c].to_numpy())
    return out


def record_run(out_df, var_out, model):
    out = record_frame("out", out_df)
    out.update(record_model(model))
    if not var_out:
        out["empty_var_out"] = np.array([1])
        return out
    for k in ("x", "no_match_vars", "penalty_vars", "area_penalty_vars"):
        out[k] = np.array(var_out[k], dtype=np.float64)
    out.update(record_violations(var_out["violations"]))
    for k, lst in var_out["violation_penalty_comparison"].items():
        out[f"cmp_{k}"] = np.array(sorted(int(p) for p in lst), dtype=np.int64)
    td = var_out["triangle_data"]
    tris = np.asarray([list(t) for t in td["triangles"]], dtype=np.int64).reshape(-1, 3)
    out["triangles"] = tris
    info = td["triangle_info"]
    keys = list(info.keys())
    out["info_keys"] = np.array(keys, dtype=np.int64)                                  # insertion order is part of the contract
    out["info_vertices"] = np.array([list(info[k]["vertices"]) for k in keys], dtype=np.int64).reshape(-1, 3)
    out["info_bounds"] = np.array([[info[k]["bounds"][b] for b in ("min_x", "max_x", "min_y", "max_y")] for k in keys], dtype=np.float64).reshape(-1, 4)
    out["info_extreme"] = np.array([[info[k][b] for b in ("max_x_vertex", "min_x_vertex", "max_y_vertex", "min_y_vertex")] for k in keys],
                                   dtype=np.int64).reshape(-1, 4)
    smap = td["aligned_simplex_map"]
    out["smap_keys"] = np.array(list(smap.keys()), dtype=np.int64)
    out["smap_counts"] = np.array([len(smap[k]) for k in smap], dtype=np.int64)
    out["smap_items"] = np.array([t for k in smap for t in sorted(smap[k])], dtype=np.int64)
    out["areas_before"] = np.array([td["areas_before"][t] for t in range(len(tris))], dtype=np.float64)
    out["areas_after"] = np.array([np.nan if td["areas_after"][t] is None else td["areas_after"][t] for t in range(len(tris))], dtype=np.float64)
    out["matched_vertices"] = np.array([td["matched_vertices"][t] for t in range(len(tris))], dtype=np.uint8).reshape(-1, 3)
    out["flipped_triangles"] = np.array(td["flipped_triangles"], dtype=np.int64)
    out["lazy"] = np.array([int(bool(var_out["lazy_constraints"])), int(var_out["lazy_cuts_added"])])
    return out


def assert_same_record(got, want, prefix=""):
    """Every key of `want` (optionally stored under a prefix) equals `got`'s, NaNs included."""
    keys = [k[len(prefix):] for k in want if k.startswith(prefix)] if prefix else list(want)
    assert keys, f"no keys under prefix {prefix!r}"
    missing = [k for k in keys if k not in got]
    assert not missing, f"missing in result: {missing[:8]}"
    extra = [k for k in got if k not in keys]
    assert not extra, f"unexpected in result: {extra[:8]}"
    for k in keys:
        a, b = np.asarray(got[k]), np.asarray(want[prefix + k])
        assert a.shape == b.shape, (k, a.shape, b.shape)
        if a.dtype.kind in "fc":
            assert np.array_equal(a, b.astype(a.dtype), equal_nan=True), k
        else:
            assert np.array_equal(a.astype(str) if a.dtype.kind in "US" else a, b.astype(str) if b.dtype.kind in "US" else b), k


def tiler_inputs(cfg):
    """Seeded frames for one window-tiler configuration (shared by the generator and the GPU test through the fixture's
    parameter table: n_ref, n_mov, side, seed, window_size, overlap, min_cells, hole)."""
    n_ref, n_mov, side, seed, ws, ov, mc, hole = cfg
    rng = np.random.default_rng(int(seed))

    def frame(n, id0):
        xy = rng.uniform(0, side, (int(n), 2))
        if hole:   # thin out a corner block so that some windows fall under min_cells and merge right / down
            keep = ~((xy[:, 0] < side * 0.35) & (xy[:, 1] < side * 0.45) & (rng.random(int(n)) < 0.85))
            xy = xy[keep]
        df = pd.DataFrame({'X': xy[:, 0], 'Y': xy[:, 1]})
        df['cell_type'] = 'a'
        df['a'] = 1.0
        df['Cell_Num_Old'] = np.arange(len(df)) + id0
        return df
    return frame(n_ref, 100000), frame(n_mov, 0)


def random_run_same_config(q):
    """Seeded parameter combination #q for the run_same sweep fixture (shared by the generator and the GPU test):
    -> (n_cells, n_types, optim overrides, gurobi overrides)."""
    rng = np.random.default_rng(9000 + q)
    lazy = bool(rng.random() < 0.75)
    op = dict(radius=float(rng.choice([12.0, 18.0, 25.0])), knn=int(rng.choice([2, 3]) if not lazy else rng.choice([3, 5, 8])),
              min_angle_deg=[None, 0, 10, 15, 30][int(rng.integers(0, 5))],
              ignore_same_type_triangles=bool(rng.random() < 0.5), ignore_knn_if_matched=bool(rng.random() < 0.4),
              lazy_constraints=lazy, max_matches=int(rng.choice([1, 1, 2])), dist_ct_coeff=float(rng.choice([1.0, 0.5, 3.0])),
              no_match_penalty=float(rng.choice([5.0, 100.0, 10000.0])), penalty_coeff=float(rng.choice([1.0, 100.0])),
              delaunay_penalty=float(rng.choice([1.0, 5.0, 10.0])))
    gp = dict(init_method=[None, "greedy", "hungarian"][int(rng.integers(0, 3))], lazy_allowed_flip_fraction=float(rng.choice([0.0, 0.05, 0.5])),
              lazy_max_cuts_per_incumbent=int(rng.choice([3, 1000])), lazy_max_cuts=[None, 7][int(rng.integers(0, 2))])
    if gp["init_method"] == "hungarian" and op["max_matches"] != 1:
        gp["init_method"] = "greedy"            # the reference raises for hungarian with max_matches != 1 (tested separately)
    return int(rng.choice([120, 180])), int(rng.choice([2, 3, 5])), op, gp
