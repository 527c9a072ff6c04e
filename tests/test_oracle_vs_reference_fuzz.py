"""The oracle against the REFERENCE ITSELF on seeded random inputs -- where the reference is mounted (the build container; never the
GPU box).  The committed fixtures pin the oracle on 17 recorded cases; this walks the same functions over a few hundred small random
ones (ragged sizes, empty results, every flag combination, NaN coordinates in the window loop), so that what every GPU parity test is
checked against is itself checked against the code it restates, beyond the recorded cases.  Tie-free inputs where the reference's
own order is unspecified (its unstable argsort over equal distances)."""
import contextlib
import io
import os
import sys

import numpy as np
import pandas as pd
import pytest

SEED = 1000 * int(os.environ.get("SAME_REF_FUZZ_SEED", "0"))      # other seeds for a longer run (0 = the suite's own cases)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="the reference is not mounted here")


@pytest.fixture(scope="module")
def ref():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    saved = sys.modules.get("gurobipy")
    import fake_gurobipy

    fake_gurobipy.install()            # src/same.py imports gurobipy at module scope; the window loop below never reaches a solver
    from ref_loader import load_reference

    ns = load_reference(with_run_same=True)
    yield ns
    if saved is None:
        sys.modules.pop("gurobipy", None)
    else:
        sys.modules["gurobipy"] = saved


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def _cells(rng, n, side, T=3, with_size=True):
    df = pd.DataFrame(rng.gamma(0.3, 30.0, (n, T)), columns=[f"t{q}" for q in range(T)])
    df.insert(0, "Y", rng.uniform(0, side, n))
    df.insert(0, "X", rng.uniform(0, side, n))
    df["cell_type"] = rng.choice(np.array(["a", "b", "c"], dtype=object), n)
    if with_size:
        df["size"] = rng.integers(1, 4, n)
    df["Cell_Num_Old"] = rng.permutation(n) + 5
    return df


def test_prune_and_priority_filter(ref, oracle):
    """utils.find_knn_within_radius (src/utils.py:709-742) and knn_utils.find_knn_with_cell_type_priority (src/knn_utils.py:5-78)."""
    rng = np.random.default_rng(101 + SEED)
    nonempty = 0
    for case in range(60):
        a, r = _cells(rng, int(rng.integers(1, 120)), 60.0), _cells(rng, int(rng.integers(1, 150)), 60.0)
        radius, knn = float(rng.choice([0.0, 2.0, 6.0, 15.0, 200.0])), int(rng.choice([1, 2, 5, 8, 40]))
        want = quiet(ref.utils.find_knn_within_radius, a, r, radius, knn)
        got = oracle.find_knn_within_radius(a, r, radius, knn)
        assert want[0].equals(got[0]) and want[1].equals(got[1]), case
        assert np.array_equal(np.asarray(want[2], dtype=np.int64).reshape(-1, 2), np.asarray(got[2], dtype=np.int64).reshape(-1, 2)), case
        if len(want[2]) == 0:
            continue
        nonempty += 1
        wantp = quiet(ref.knn_utils.find_knn_with_cell_type_priority, a, r, radius, knn)
        gotp = oracle.find_knn_with_cell_type_priority(a, r, radius, knn)
        assert wantp[0].equals(gotp[0]) and wantp[1].equals(gotp[1]) and [tuple(map(int, p)) for p in wantp[2]] == [tuple(map(int, p)) for p in gotp[2]], case
    assert nonempty > 30


def test_triangle_filter_info_and_order_sweep(ref, oracle):
    """helpers.filter_triangles_by_radius (src/helpers.py:233-395), precompute_triangle_info (:184-210),
    violationhelper.verify_spatial_preservation (src/violationhelper.py:1-134)."""
    from scipy.spatial import Delaunay

    rng = np.random.default_rng(202 + SEED)
    kept_any = violated_any = 0
    for case in range(50):
        a = _cells(rng, int(rng.integers(4, 160)), 80.0)
        pts = a[["X", "Y"]].to_numpy()
        tris = Delaunay(pts).simplices
        kw = dict(radius=float(rng.choice([8.0, 20.0, 60.0])), aligned_df=a, ignore_same_type_triangles=bool(case % 2),
                  ensure_min_triangle_per_node=bool(case % 3), min_angle_deg=[15, None, 0, 35][case % 4])
        if case % 5 == 0:
            want, want_un = quiet(ref.helpers.filter_triangles_by_radius, pts, tris, remove_unconstrained_nodes=True, **kw)
            got, got_un = oracle.filter_triangles_by_radius(pts, tris, remove_unconstrained_nodes=True, **kw)
            assert set(map(int, want_un)) == set(map(int, got_un)), case
        else:
            want = quiet(ref.helpers.filter_triangles_by_radius, pts, tris, **kw)
            got = oracle.filter_triangles_by_radius(pts, tris, **kw)
        assert [tuple(map(int, t)) for t in want] == [tuple(map(int, t)) for t in got], case
        if not len(want):
            continue
        kept_any += 1
        smap = {i: set() for i in range(len(a))}
        for q, t in enumerate(want):
            for v in t:
                smap[int(v)].add(q)
        want_info = ref.helpers.precompute_triangle_info(a, want, smap)
        got_info = oracle.precompute_triangle_info(a, got, oracle.simplex_map(len(a), got))
        assert list(want_info) == list(got_info), case                 # insertion order is part of the contract
        for k in want_info:
            assert list(map(int, want_info[k]["vertices"])) == list(map(int, got_info[k]["vertices"])) and want_info[k]["bounds"] == got_info[k]["bounds"]
            assert all(int(want_info[k][f]) == int(got_info[k][f]) for f in ("max_x_vertex", "min_x_vertex", "max_y_vertex", "min_y_vertex"))
        r = _cells(rng, int(rng.integers(3, 120)), 80.0)
        ai = rng.choice(len(a), int(rng.integers(1, len(a) + 1)), replace=False)
        m = pd.DataFrame({"aligned_idx": ai, "ref_idx": rng.integers(0, len(r), len(ai))})
        if case % 4 == 1 and len(m) > 2:
            m = pd.concat([m, m.iloc[:2].assign(ref_idx=0)], ignore_index=True)        # an aligned cell twice: the later row wins
        wv = quiet(ref.violationhelper.verify_spatial_preservation, a, r, m, want_info)
        gv = oracle.verify_spatial_preservation(a, r, m, got_info)
        assert wv["violation_summary"] == gv["violation_summary"], case
        assert sorted(map(int, wv["triangles_with_violations"])) == sorted(map(int, gv["triangles_with_violations"]))
        assert sorted(map(int, wv["points_with_violations"])) == sorted(map(int, gv["points_with_violations"]))
        for name in ("x_order_violations", "y_order_violations"):
            flat = lambda lst: [(int(e["triangle_idx"]), int(e["point1"]["aligned_idx"]), int(e["point1"]["ref_idx"]), int(e["point2"]["aligned_idx"]),
                                 int(e["point2"]["ref_idx"])) + tuple(float(v) for p in ("point1", "point2") for k_, v in sorted(e[p].items()) if "idx" not in k_)
                                for e in lst]
            assert flat(wv[name]) == flat(gv[name]), (case, name)
        violated_any += bool(wv["violation_summary"]["total_violations"])
    assert kept_any > 30 and violated_any > 10


def test_mip_start_heuristics(ref, oracle):
    """init_helpers.compute_mip_start_pairs (src/init_helpers.py:46-177): greedy with exact cost ties (Python's stable sort decides),
    rows that prefer to stay unmatched, the Hungarian start with its size cut-off."""
    rng = np.random.default_rng(303 + SEED)
    for case in range(80):
        n_a, n_r = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        pairs = [(int(i), int(j)) for i in range(n_a) for j in rng.choice(n_r, int(rng.integers(0, min(n_r, 5) + 1)), replace=False)]
        if case % 6 == 0:
            pairs = pairs + pairs[:3]                      # a pair listed twice (the Hungarian start keeps the LAST index)
        costs = list(rng.integers(0, 8, len(pairs)).astype(float) if case % 2 else rng.gamma(2.0, 3.0, len(pairs)))
        kw = dict(valid_pairs=pairs, costs=costs, n_aligned=n_a, n_ref=n_r, aligned_sizes=rng.integers(1, 4, n_a).astype(float),
                  no_match_penalty=float(rng.choice([0.5, 3.0, 1e4])), max_matches=1, init_method=["greedy", "hungarian"][case % 3 == 2],
                  init_big_m=1e9, init_hungarian_max_n=int(rng.choice([20, 2000])), verbose=False)
        want = ref.init_helpers.compute_mip_start_pairs(**kw)
        got = oracle.compute_mip_start_pairs(**kw)
        assert [tuple(map(int, t)) for t in want[0]] == [tuple(map(int, t)) for t in got[0]] and set(map(int, want[1])) == set(map(int, got[1])), case


def test_window_loop_with_rows_that_have_no_coordinates(ref, oracle):
    """sliding_window_matching's tiling / merging / trimming (src/same.py:481-590) RUN AS-IS with run_same replaced by a recorder, on
    layouts in which some rows have NaN coordinates: pandas' min / max skip them (:481-482), no window receives them, and the job runs.
    The oracle's window_plan must name the same windows, hand each the same cells in the same order and keep the same central rows."""
    rng = np.random.default_rng(404 + SEED)
    calls = []

    def recorder(aligned_df, ref_df, commonCT, optim_params, gurobi_params, outprefix, aligned_delaunay, aligned_delaunay_vertex_col,
                 ignore_precomputed_triangulation):
        calls.append((aligned_df["Cell_Num_Old"].to_numpy().copy(), ref_df["Cell_Num_Old"].to_numpy().copy()))
        return pd.DataFrame({"X": aligned_df["X"].to_numpy(), "Y": aligned_df["Y"].to_numpy(), "Aligned_Cell_Num_Old": aligned_df["Cell_Num_Old"].to_numpy()}), {}

    saved = ref.same.run_same
    ref.same.run_same = recorder
    windows = 0
    try:
        for case in range(12):
            side = float(rng.choice([150.0, 333.3, 500.0]))
            frames = []
            for n in (int(rng.integers(300, 1500)), int(rng.integers(300, 1500))):
                df = _cells(rng, n, side, T=1, with_size=False)
                hole = (df["X"] < side * 0.3) & (df["Y"] < side * 0.4) & (rng.random(n) < 0.9)          # under-populated corner: merges
                df = df[~hole].reset_index(drop=True)
                df["Cell_Num_Old"] = np.arange(len(df))                                                    # id = row position
                if case % 2:
                    df.loc[rng.integers(0, len(df), 7), "X"] = np.nan
                    df.loc[rng.integers(0, len(df), 4), "Y"] = np.nan
                frames.append(df)
            r_df, m_df = frames
            ws = int(rng.choice([60, 100, 170]))
            op = dict(window_size=ws, overlap=int(rng.choice([0, ws // 5, ws // 2])), min_cells_per_window=int(rng.choice([10, 40])))
            res = quiet(ref.same.sliding_window_matching, r_df.copy(), m_df.copy(), commonCT=["t0"], optim_params=dict(op))
            rxy, mxy = r_df[["X", "Y"]].to_numpy(dtype=np.float64), m_df[["X", "Y"]].to_numpy(dtype=np.float64)
            plan = oracle.window_plan(rxy, mxy, ws, op["overlap"], op["min_cells_per_window"])
            assert len(plan) == len(calls), (case, len(plan), len(calls))
            want_rows = []
            for w, (a_ids, r_ids) in zip(plan, calls):
                ma, mr = oracle.window_mask(mxy, *w["box"]), oracle.window_mask(rxy, *w["box"])
                assert np.array_equal(np.flatnonzero(ma), a_ids) and np.array_equal(np.flatnonzero(mr), r_ids), (case, w["window_id"])
                tx0, tx1, ty0, ty1 = w["trim"]
                c = np.flatnonzero(ma & (mxy[:, 0] >= tx0) & (mxy[:, 0] < tx1) & (mxy[:, 1] >= ty0) & (mxy[:, 1] < ty1))
                want_rows.append(np.column_stack((c, np.full(len(c), w["window_id"]))))
            got_rows = res[["Aligned_Cell_Num_Old", "window_id"]].to_numpy(dtype=np.int64) if len(res) else np.zeros((0, 2), np.int64)
            assert np.array_equal(got_rows, np.concatenate(want_rows + [np.zeros((0, 2), np.int64)]).astype(np.int64)), case
            windows += len(plan)
            calls.clear()
    finally:
        ref.same.run_same = saved
    assert windows > 60


def test_inline_loops_of_run_same(ref, oracle):
    """run_same's inline pre-MIP and sweep loops (src/same.py:1180-1189 pair costs on object-dtype rows, :1128-1146 weights and source
    signs, :634-669 the lazy-constraint body, :1362-1402 signed-area flips) cannot be imported; tools/gen_golden.py drives them in the
    expression order of the cited lines on the reference's own frames (and calls helpers.calculate_signed_area as-is).  The oracle against
    those drivers on random inputs: type columns with exact zeros, integer and float sizes, collinear triples (sign 0), partial matchings,
    several pairs above 0.5 for one aligned row (the last one wins)."""
    from scipy.spatial import Delaunay

    import gen_golden as gg          # tools/ (on sys.path through the `ref` fixture): loads the reference the same way

    rng = np.random.default_rng(505 + SEED)
    flips = checked_total = 0
    for case in range(25):
        T = int(rng.integers(1, 6))
        a, r = _cells(rng, int(rng.integers(5, 60)), 40.0, T=T), _cells(rng, int(rng.integers(5, 60)), 40.0, T=T)
        cols = [f"t{q}" for q in range(T)]
        for df in (a, r):
            df[cols] = df[cols].where(rng.random((len(df), T)) > 0.2, 0.0)          # proportions with exact zeros
        if case % 3 == 0:
            a["size"] = a["size"] * 1.5
        if case % 4 == 0:
            a.loc[a.index[:3], ["X", "Y"]] = [[1.0, 1.0], [2.0, 2.0], [3.0, 3.0]]    # a collinear triple
        pairs = [(int(i), int(j)) for i in range(len(a)) for j in rng.choice(len(r), int(rng.integers(0, 4)), replace=False)]
        if not pairs:
            continue
        w = float(rng.choice([1.0, 0.37, 2.5]))
        want_c = gg.ref_pair_costs(a, r, pairs, cols, w)
        got_c = np.asarray(oracle.pair_costs(a, r, pairs, cols, w), dtype=np.float64)
        assert np.array_equal(want_c, got_c), case
        tris = Delaunay(a[["X", "Y"]].to_numpy() + rng.normal(0, 1e-9, (len(a), 2))).simplices        # (jitter only for Qhull: the frames keep the collinear triple)
        if case % 4 == 0:
            tris = np.vstack([tris, [[0, 1, 2]]])
        want_w, want_s = gg.ref_weights_signs(a, tris)
        assert np.array_equal(np.asarray(oracle.triangle_weights(a, tris), dtype=np.float64), want_w), case
        got_s = np.asarray(oracle.source_signs(a, tris), dtype=np.float64)
        assert np.array_equal(got_s, want_s), case
        x = (rng.random(len(pairs)) < 0.6).astype(float) * rng.choice([0.51, 1.0, 0.99], len(pairs))
        want_checked, want_viol = gg.ref_lazy_sweep(x, pairs, tris, want_s, r)
        got_checked, got_viol = oracle.lazy_orientation_sweep(x, pairs, tris, got_s, r[["X", "Y"]].to_numpy(), len(a))
        assert int(got_checked) == int(want_checked) and [int(v[0]) if np.ndim(v) else int(v) for v in got_viol] == want_viol.tolist(), case
        a2r = {}
        for idx, (i, j) in enumerate(pairs):
            if x[idx] > 0.5:
                a2r[i] = j
        wb, wa, wf, wm3 = gg.ref_area_flips(a, r, tris, a2r)
        match = np.full(len(a), -1, np.int32)
        for i, j in a2r.items():
            match[i] = j
        gb, ga, gm3, gf = oracle.area_flip(a[["X", "Y"]].to_numpy(), r[["X", "Y"]].to_numpy(), np.asarray(tris, dtype=np.int32), match)
        assert np.array_equal(gb, wb) and np.array_equal(ga, wa, equal_nan=True) and np.array_equal(np.flatnonzero(gf), wf), case
        assert np.array_equal(np.asarray(gm3, dtype=np.uint8).reshape(-1, 3), wm3.reshape(-1, 3)), case
        flips += len(wf)
        checked_total += int(want_checked)
    assert checked_total > 100 and flips > 10


def test_metacell_collapse(ref, oracle):
    """metacell_utils.greedy_triangle_collapse (src/metacell_utils.py:160-561) on random small sections: size limits 1 .. 9, r_max / angle
    rules on and off, an extra text column carried along."""
    rng = np.random.default_rng(606 + SEED)
    collapsed = 0
    for case in range(14):
        df = _cells(rng, int(rng.integers(6, 140)), float(rng.choice([30.0, 60.0])), T=2, with_size=False)
        df["batch"] = np.where(np.arange(len(df)) % 3 == 0, "b0", "b1")
        kw = dict(max_metacell_size=int(rng.choice([1, 3, 4, 9])), r_max=[None, 8.0, 25.0][case % 3], min_angle_deg=[10, None, 25][case % 3])
        want = quiet(ref.metacell_utils.greedy_triangle_collapse, df, use_alpha_shape=False, return_object=True, **kw)
        gdf, gtri, gorig = oracle.greedy_triangle_collapse(df, **kw)
        wdf = want.metacell_df
        assert list(wdf.columns) == list(gdf.columns) and len(wdf) == len(gdf), case
        assert [list(map(int, m)) for m in wdf["members"]] == [list(map(int, m)) for m in gdf["members"]], case
        for c in wdf.columns:
            if c != "members":
                assert np.array_equal(wdf[c].to_numpy(), gdf[c].to_numpy()), (case, c)
        assert np.array_equal(np.asarray(want.metacell_delaunay, dtype=np.int64).reshape(-1, 3), np.asarray(gtri, dtype=np.int64).reshape(-1, 3)), case
        assert np.array_equal(np.asarray(want.original_delaunay, dtype=np.int64).reshape(-1, 3), np.asarray(gorig, dtype=np.int64).reshape(-1, 3)), case
        collapsed += len(df) - len(wdf)
    assert collapsed > (100 if SEED == 0 else 40)        # coverage of the suite's own seed; other seeds (SAME_REF_FUZZ_SEED) collapse 90-130
