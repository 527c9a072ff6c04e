"""GPU parity: the HIP path (through the ctypes C ABI) against the reference's golden fixtures
and against the CPU oracle on seeded inputs.  Integer / index / flag outputs and fp64 costs are
compared bit-exactly (the fp64 contract of BASELINE.json is 1e-6 relative; the kernels keep the
reference's operation order, so equality is what we assert)."""
import numpy as np
import pandas as pd
import pytest

from conftest import FULL_CASES, frames_from_golden, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import same_amd
    from same_amd import _lib

    assert _lib.device_count() >= 1, "GPU tests need a device; the product path has no CPU fallback"
    return same_amd


@pytest.fixture(scope="module")
def ops(hip):
    from same_amd import ops as _ops

    return _ops


def _compacted(g):
    a_df, r_df, cols = frames_from_golden(g)
    return a_df.iloc[g["kept_aligned"]].reset_index(drop=True), r_df.iloc[g["kept_ref"]].reset_index(drop=True), cols


# ------------------------------------------------------------------------------------------ a2 / a3
@pytest.mark.parametrize("case", FULL_CASES)
def test_knn_golden(hip, case):
    g = load_golden(case)
    a_df, r_df, _ = frames_from_golden(g)
    na, nr, pairs = hip.find_knn_within_radius(a_df, r_df, g["params"][0], knn=int(g["params"][1]), verbose=False)
    assert np.array_equal(na["__row"].to_numpy(), g["kept_aligned"])
    assert np.array_equal(nr["__row"].to_numpy(), g["kept_ref"])
    assert np.array_equal(np.asarray(pairs, dtype=np.int64), g["pairs"])
    _, _, prio = hip.find_knn_with_cell_type_priority(a_df, r_df, g["params"][0], knn=int(g["params"][1]), verbose=False)
    assert np.array_equal(np.asarray(prio, dtype=np.int64).reshape(-1, 2), g["pairs_priority"])


def test_knn_adversarial(hip, ops, oracle):
    g = load_golden("adversarial")
    a = pd.DataFrame({"X": g["edge_axy"][:, 0], "Y": g["edge_axy"][:, 1]})
    r = pd.DataFrame({"X": g["edge_rxy"][:, 0], "Y": g["edge_rxy"][:, 1], "__row": np.arange(len(g["edge_rxy"]))})
    _, nr, pairs = hip.find_knn_within_radius(a, r, 5.0, knn=4, verbose=False)
    assert np.array_equal(np.asarray(pairs), g["edge_pairs"]) and np.array_equal(nr["__row"].to_numpy(), g["edge_kept_ref"])
    a = pd.DataFrame({"X": g["sparse_axy"][:, 0], "Y": g["sparse_axy"][:, 1], "__row": np.arange(len(g["sparse_axy"]))})
    r = pd.DataFrame({"X": g["sparse_rxy"][:, 0], "Y": g["sparse_rxy"][:, 1], "__row": np.arange(len(g["sparse_rxy"]))})
    na, nr, pairs = hip.find_knn_within_radius(a, r, 4.0, knn=3, verbose=False)
    assert np.array_equal(na["__row"].to_numpy(), g["sparse_kept_aligned"])
    assert np.array_equal(nr["__row"].to_numpy(), g["sparse_kept_ref"])
    assert np.array_equal(np.asarray(pairs), g["sparse_pairs"])
    # exact ties on a grid: documented rule (d2, ref index) == oracle, bit for bit
    idx, d2, cnt = ops.knn_prune(g["grid_axy"], g["grid_rxy"], 1.5, 6)
    oidx, od2, ocnt = oracle.knn_prune(g["grid_axy"], g["grid_rxy"], 1.5, 6)
    assert np.array_equal(idx, oidx) and np.array_equal(d2, od2) and np.array_equal(cnt, ocnt)
    # empty reference set / no neighbour at all
    idx, d2, cnt = ops.knn_prune(g["grid_axy"], np.zeros((0, 2)), 1.5, 6)
    assert (idx == -1).all() and (cnt == 0).all() and np.isinf(d2).all()
    idx, _, cnt = ops.knn_prune(g["grid_axy"], g["grid_rxy"] + 1000.0, 1.5, 6)
    assert (idx == -1).all() and (cnt == 0).all()
    with pytest.raises(ValueError):
        hip.prepare_same_inputs(r.assign(c1=1.0, cell_type="c1"), a.assign(X=a["X"] + 1e6, c1=1.0, cell_type="c1"), ["c1"],
                                optim_params={"radius": 1.0}, verbose=False)


@pytest.mark.parametrize("n_m,n_r,k,radius", [(1000, 1300, 8, 10.0), (2500, 2500, 32, 25.0), (777, 4100, 64, 60.0), (513, 65, 5, 400.0),
                                               (64, 5000, 1, 30.0), (900, 3000, 65, 70.0), (300, 5000, 200, 150.0), (700, 2500, 448, 1e9),
                                               (50, 600, 448, 40.0)])
def test_knn_vs_oracle_seeded(ops, oracle, n_m, n_r, k, radius):
    from same_amd import synth

    r = synth.make_cells(n_r, 3, seed=10, rho=0.01)
    m = synth.make_cells(n_m, 3, seed=11, side=r["side"])
    idx, d2, cnt = ops.knn_prune(m["xy"], r["xy"], radius, k)
    oidx, od2, ocnt = oracle.knn_prune(m["xy"], r["xy"], radius, k)
    assert np.array_equal(cnt, ocnt)
    assert np.array_equal(idx, oidx)
    assert np.array_equal(d2, od2)
    # row blocks are independent: any block equals the slice of the whole (sharding invariant)
    b0, b1 = n_m // 3, n_m // 3 + 200
    bidx, _, bcnt = ops.knn_prune(m["xy"], r["xy"], radius, k, row_begin=b0, row_end=min(b1, n_m))
    assert np.array_equal(bidx, idx[b0:b1]) and np.array_equal(bcnt, cnt[b0:b1])


@pytest.mark.parametrize("mode", ["grid", "brute"])
def test_knn_modes_agree(ops, oracle, mode, monkeypatch):
    """The uniform-grid path and the brute-force path are the same function (bit for bit)."""
    from same_amd import synth

    monkeypatch.setenv("SAME_KNN_MODE", mode)
    rng = np.random.default_rng(8)
    cases = []
    r = synth.make_cells(30000, 3, seed=20)
    cases.append((synth.make_jittered(r, seed=21)["xy"], r["xy"], 25.0, 32))
    cases.append((rng.uniform(-50, 1050, (3000, 2)), rng.uniform(0, 1000, (5000, 2)), 40.0, 16))      # aligned outside the ref box
    cases.append((rng.uniform(0, 10, (500, 2)), rng.uniform(0, 10, (4000, 2)), 6.0, 64))               # dense: overflow prune path
    cases.append((rng.uniform(0, 1e4, (2000, 2)), rng.uniform(0, 1e4, (3000, 2)), 0.5, 8))            # sparse: radius << spacing
    cases.append((rng.uniform(0, 100, (300, 2)), np.repeat(rng.uniform(0, 100, (50, 2)), 20, 0), 15.0, 10))  # duplicates
    cases.append((rng.uniform(0, 100, (300, 2)), np.column_stack((rng.uniform(0, 100, 700), np.full(700, 3.0))), 9.0, 7))  # collinear refs
    cases.append((rng.uniform(0, 100, (300, 2)), np.tile([[5.0, 5.0]], (600, 1)), 200.0, 12))          # all refs coincide
    g = np.stack(np.meshgrid(np.arange(40.0), np.arange(40.0)), -1).reshape(-1, 2)
    cases.append((g + 0.5, g, 1.5, 9))                                                               # exact ties, cell-edge distances
    cases.append((g, g * 1.0, 1.0, 5))                                                               # distance exactly == radius
    for axy, rxy, radius, k in cases:
        idx, d2, cnt = ops.knn_prune(axy, rxy, radius, k)
        oidx, od2, ocnt = oracle.knn_prune(axy, rxy, radius, k)
        assert np.array_equal(cnt, ocnt) and np.array_equal(idx, oidx) and np.array_equal(d2, od2)


def test_knn_overflow_prune_path(ops, oracle):
    """More than KNN_CAP (128) in-radius candidates per row forces the in-kernel prune."""
    rng = np.random.default_rng(5)
    rxy = rng.uniform(0, 10, size=(3000, 2))
    axy = rng.uniform(2, 8, size=(100, 2))
    for k in (3, 32, 64):
        idx, d2, cnt = ops.knn_prune(axy, rxy, 6.0, k)
        oidx, od2, ocnt = oracle.knn_prune(axy, rxy, 6.0, k)
        assert np.array_equal(idx, oidx) and np.array_equal(d2, od2) and np.array_equal(cnt, ocnt)
    # duplicates (exact d2 ties) inside the overflow path
    rxy2 = np.repeat(rxy[:600], 4, axis=0)
    idx, d2, cnt = ops.knn_prune(axy, rxy2, 6.0, 48)
    oidx, od2, ocnt = oracle.knn_prune(axy, rxy2, 6.0, 48)
    assert np.array_equal(idx, oidx) and np.array_equal(d2, od2)


# ------------------------------------------------------------------------------------------ a4 / dense
@pytest.mark.parametrize("case", FULL_CASES)
def test_pair_cost_golden(hip, ops, oracle, case):
    g = load_golden(case)
    na, nr, cols = _compacted(g)
    sel = g["cost_sel"]
    c = np.array(hip.pair_costs(na, nr, g["pairs"][sel], cols, g["params"][3]))
    assert np.array_equal(c, g["costs"])  # bit-exact vs the reference's pandas loop
    c2 = np.array(hip.pair_costs(na, nr, g["pairs"][sel[:200]], cols, 2.5))
    assert np.array_equal(c2, g["costs_w2p5"])
    call = np.array(hip.pair_costs(na, nr, g["pairs"], cols, g["params"][3]))
    assert np.array_equal(call, g["all_costs"])
    D = hip.dense_cost_matrix(na, nr, cols, g["params"][3])
    p = g["pairs"]
    assert np.array_equal(D[p[:, 0], p[:, 1]], g["all_costs"])
    A, R = na[cols].to_numpy(), nr[cols].to_numpy()
    axy, rxy = na[["X", "Y"]].to_numpy(), nr[["X", "Y"]].to_numpy()
    assert np.array_equal(D, oracle.dense_cost(A, R, axy, rxy, float(g["params"][3])))
    D32 = ops.dense_cost(A, R, axy, rxy, float(g["params"][3]), dtype=np.float32)
    O32 = oracle.dense_cost(A, R, axy, rxy, float(g["params"][3]), dtype=np.float32)
    assert np.array_equal(D32, O32)
    # fp32 variant (BASELINE cfg 5) vs fp64: inputs are rounded to fp32 on a 0-100 scale, so the error is
    # absolute, ~eps32 * 100 per term (|a-r| cancels): tolerance 1e-6 of the value scale (100*T) + 1e-5 relative
    np.testing.assert_allclose(D32, D, rtol=1e-5, atol=1e-6 * 100 * max(len(cols), 1))


# > 48: the row-blocked kernel (pieces of 8 types + remainder)
@pytest.mark.parametrize("T", [0, 1, 2, 7, 20, 24, 25, 33, 48, 49, 55, 56, 64, 70, 129, 300])
@pytest.mark.parametrize("w", [1.0, 0.37])
def test_dense_cost_all_T(ops, oracle, T, w):
    rng = np.random.default_rng(T)
    n_m, n_r = 257, 1031  # ragged: not multiples of the tile
    A = rng.dirichlet(np.full(max(T, 1), 0.3), size=n_m)[:, :T] * 100 if T else np.zeros((n_m, 0))
    R = rng.dirichlet(np.full(max(T, 1), 0.3), size=n_r)[:, :T] * 100 if T else np.zeros((n_r, 0))
    axy, rxy = rng.uniform(0, 300, (n_m, 2)), rng.uniform(0, 300, (n_r, 2))
    D = ops.dense_cost(A, R, axy, rxy, w, 3, 250)
    assert np.array_equal(D, oracle.dense_cost(A, R, axy, rxy, w, 3, 250))
    D32 = ops.dense_cost(A, R, axy, rxy, w, 3, 250, dtype=np.float32)
    assert np.array_equal(D32, oracle.dense_cost(A, R, axy, rxy, w, 3, 250, dtype=np.float32))
    # pair kernel == dense kernel on sampled pairs
    pairs = np.column_stack((rng.integers(3, 250, 500), rng.integers(0, n_r, 500))).astype(np.int32)
    c = ops.pair_cost(A, R, axy, rxy, pairs, w)
    assert np.array_equal(c, D[pairs[:, 0] - 3, pairs[:, 1]])


def test_cost_edge_shapes(ops):
    z = np.zeros((0, 2))
    assert ops.dense_cost(np.zeros((0, 4)), np.zeros((5, 4)), z, np.zeros((5, 2)), 1.0).shape == (0, 5)
    assert ops.dense_cost(np.zeros((3, 4)), np.zeros((0, 4)), np.zeros((3, 2)), z, 1.0).shape == (3, 0)
    assert len(ops.pair_cost(np.zeros((3, 4)), np.zeros((2, 4)), np.zeros((3, 2)), np.zeros((2, 2)), np.zeros((0, 2), np.int32), 1.0)) == 0
    from same_amd._lib import SameHipError
    with pytest.raises(SameHipError):  # out-of-range pair index is reported, never dereferenced
        ops.pair_cost(np.zeros((3, 4)), np.zeros((2, 4)), np.zeros((3, 2)), np.zeros((2, 2)), np.array([[0, 2]], np.int32), 1.0)
    with pytest.raises(SameHipError):
        ops.knn_prune(np.zeros((3, 2)), np.zeros((3, 2)), 1.0, 449)      # above SAME_MAX_KNN


# ------------------------------------------------------------------------------------------ a7 / a8 / a9
@pytest.mark.parametrize("case", FULL_CASES)
def test_triangle_filter_golden(hip, case):
    g = load_golden(case)
    na, _, _ = _compacted(g)
    pts = na[["X", "Y"]].to_numpy()
    radius = g["params"][0]
    mad = None if g["params"][2] < 0 else g["params"][2]
    for tag, kw, rad in (("plain", dict(ignore_same_type_triangles=False, min_angle_deg=mad), radius),
                         ("type", dict(ignore_same_type_triangles=True, min_angle_deg=mad), radius),
                         ("noangle", dict(ignore_same_type_triangles=True, min_angle_deg=None), radius),
                         ("a30", dict(ignore_same_type_triangles=True, min_angle_deg=30, ensure_min_triangle_per_node=False), radius),
                         ("tight", dict(ignore_same_type_triangles=True, min_angle_deg=mad), radius * 0.35)):
        kept, unc = hip.filter_triangles_by_radius(pts, g["delaunay"], rad, aligned_df=na, remove_unconstrained_nodes=True,
                                                   verbose=False, **kw)
        assert np.array_equal(np.array(kept, dtype=np.int64).reshape(-1, 3), g[f"tri_{tag}"]), tag
        assert sorted(unc) == g[f"unc_{tag}"].tolist(), tag
    tris = g["tri_plain"]
    w, s = hip.triangle_weights_and_signs(na, tris)
    assert np.array_equal(np.array(w, dtype=np.float64), g["tri_weights"]) and np.array_equal(np.array(s), g["source_signs"])
    info = hip.precompute_triangle_info(na, tris, hip.build_simplex_map(len(na), tris))
    keys = list(info.keys())
    assert keys == g["tinfo_keys"].tolist()
    b = np.array([[info[k]["bounds"][q] for q in ("min_x", "max_x", "min_y", "max_y")] for k in keys])
    e = np.array([[info[k][q] for q in ("max_x_vertex", "min_x_vertex", "max_y_vertex", "min_y_vertex")] for k in keys])
    assert np.array_equal(b, g["tinfo_bounds"]) and np.array_equal(e, g["tinfo_extreme"])


def test_triangle_filter_adversarial(hip, ops):
    g = load_golden("adversarial")
    tdf = pd.DataFrame({"X": g["adv_pts"][:, 0], "Y": g["adv_pts"][:, 1], "cell_type": g["adv_type"].astype(object)})
    for tag, kw in (("45", dict(min_angle_deg=45, ignore_same_type_triangles=False)),
                    ("45t", dict(min_angle_deg=45, ignore_same_type_triangles=True)),
                    ("none", dict(min_angle_deg=None, ignore_same_type_triangles=True)),
                    ("0", dict(min_angle_deg=0, ignore_same_type_triangles=False)),
                    ("15", dict(min_angle_deg=15, ignore_same_type_triangles=True))):
        for rad in (10.0, 3.0, 1.0):
            kept, unc = hip.filter_triangles_by_radius(g["adv_pts"], g["adv_tris"], rad, aligned_df=tdf,
                                                       remove_unconstrained_nodes=True, verbose=False, **kw)
            assert np.array_equal(np.array(kept, dtype=np.int64).reshape(-1, 3), g[f"adv_kept_{tag}_{rad}"]), (tag, rad)
            assert sorted(unc) == g[f"adv_unc_{tag}_{rad}"].tolist(), (tag, rad)
    assert hip.filter_triangles_by_radius(g["adv_pts"], np.zeros((0, 3), dtype=int), 10.0, aligned_df=tdf,
                                          ignore_same_type_triangles=True, verbose=False) == []
    sign, _ = ops.tri_sign_weight(g["adv_pts"], None, g["adv_tris"])
    assert np.array_equal(sign.astype(np.float64), g["adv_signs"])


def test_triangles_vs_oracle_seeded(ops, oracle):
    from scipy.spatial import Delaunay
    from same_amd import synth

    m = synth.make_cells(20000, 6, seed=3)
    tris = Delaunay(m["xy"]).simplices
    for mad, rad in ((15, 25.0), (5, 12.0), (None, 30.0), (40, 100.0)):
        en, thr = oracle.cos_threshold(mad)
        cls, perim, mc = ops.tri_classify(m["xy"], tris, rad, en, thr, m["cell_type"])
        ocls, operim, omc = oracle.tri_classify(m["xy"], tris, rad, mad, m["cell_type"])
        assert np.array_equal(cls, ocls) and np.array_equal(perim, operim) and np.array_equal(mc, omc)
    s, w = ops.tri_sign_weight(m["xy"], m["size"] * 1.5, tris)
    os_, ow = oracle.tri_sign_weight(m["xy"], m["size"] * 1.5, tris)
    assert np.array_equal(s, os_) and np.array_equal(w, ow)


# ------------------------------------------------------------------------------------------ a5
@pytest.mark.parametrize("case", FULL_CASES)
def test_mip_start_golden(hip, case):
    g = load_golden(case)
    na, nr, _ = _compacted(g)
    vp = [tuple(p) for p in g["pairs"].tolist()]
    kw = dict(valid_pairs=vp, costs=list(g["all_costs"]), n_aligned=len(na), n_ref=len(nr),
              aligned_sizes=na["size"].to_numpy(dtype=float), max_matches=1, verbose=False)
    ch, un = hip.compute_mip_start_pairs(no_match_penalty=g["params"][4], init_method="greedy", **kw)
    assert np.array_equal(np.array(ch, dtype=np.int64).reshape(-1, 3), g["greedy_chosen"]) and sorted(un) == g["greedy_unmatched"].tolist()
    ch, un = hip.compute_mip_start_pairs(no_match_penalty=float(g["greedy_lo_penalty"][0]), init_method="greedy", **kw)
    assert np.array_equal(np.array(ch, dtype=np.int64).reshape(-1, 3),
                          g["greedy_lo_chosen"]) and sorted(un) == g["greedy_lo_unmatched"].tolist()
    ch, un = hip.compute_mip_start_pairs(no_match_penalty=g["params"][4], init_method="hungarian", init_hungarian_max_n=100000, **kw)
    assert np.array_equal(np.array(ch, dtype=np.int64).reshape(-1, 3),
                          g["hungarian_chosen"]) and sorted(un) == g["hungarian_unmatched"].tolist()
    assert hip.compute_mip_start_pairs(no_match_penalty=1.0, init_method="hungarian", init_hungarian_max_n=10, **kw) == ([], set())
    with pytest.raises(ValueError):
        hip.compute_mip_start_pairs(no_match_penalty=1.0, init_method="bogus", **kw)
    with pytest.raises(ValueError):
        hip.compute_mip_start_pairs(no_match_penalty=1.0, init_method="hungarian", **{**kw, "max_matches": 2})


def test_assign_matrix_duplicates(ops, oracle):
    pairs = np.array([[0, 1], [1, 0], [0, 1], [2, 2], [0, 1]], np.int32)
    costs = np.array([5.0, 6.0, 7.0, 8.0, 9.0])
    un = np.array([100.0, 200.0, 300.0])
    m = ops.assign_matrix(pairs, costs, un, 3, 4, 1e9)
    o = np.empty((3, 7))
    oracle.lib().orc_assign_matrix(pairs, costs, 5, un, 3, 4, 1e9, o.reshape(-1))
    assert np.array_equal(m, o) and m[0, 1] == 9.0
    assert np.array_equal(ops.pair_rowmin(pairs, costs, 4), np.array([5.0, 6.0, 8.0, np.inf]))


# ------------------------------------------------------------------------------------------ a10 / a11 / a12 / a14
@pytest.mark.parametrize("case", FULL_CASES)
def test_sweeps_golden(hip, ops, case):
    g = load_golden(case)
    na, nr, _ = _compacted(g)
    tris = g["tri_plain"]
    rxy = nr[["X", "Y"]].to_numpy()
    sweep = hip.LazyOrientationSweep(g["pairs"], tris, g["source_signs"], rxy, len(na))
    for _ in range(2):  # the bound state is reused across incumbents
        checked, viol, pidx = sweep.sweep(g["x_vals"])
        assert checked == int(g["lazy_checked"][0])
        assert [v[0] for v in viol] == g["lazy_violating"].tolist()
        assert all(tuple(tris[v[0]]) == tuple(v[1:]) for v in viol)
    ch = g["greedy_chosen"]
    assert np.array_equal(pidx[ch[:, 0]], ch[:, 2])
    # cut selection logic (src/same.py:671-692)
    n_v = len(g["lazy_violating"])
    assert len(sweep.select_cuts(g["x_vals"], allowed_frac=None, per_inc_limit=None)) == n_v
    assert len(sweep.select_cuts(g["x_vals"], allowed_frac=None, per_inc_limit=7)) == min(7, n_v)
    assert sweep.select_cuts(g["x_vals"], allowed_frac=1.0) == []
    assert len(sweep.select_cuts(g["x_vals"], allowed_frac=0.0, per_inc_limit=1000, remaining_global=3)) == min(3, n_v)
    assert sweep.select_cuts(np.zeros(len(g["x_vals"]))) == []
    # a11
    m_df = pd.DataFrame({"aligned_idx": ch[:, 0], "ref_idx": ch[:, 1]})
    info = hip.precompute_triangle_info(na, tris, hip.build_simplex_map(len(na), tris))
    from test_oracle_golden import _check_violations
    _check_violations(hip.verify_spatial_preservation(na, nr, m_df, info), g, "")
    # a12
    before, after, flipped, m3 = hip.triangle_area_flips(na, nr, tris, {int(i): int(j) for i, j, _ in ch})
    assert np.array_equal(np.array([before[t] for t in range(len(tris))]), g["area_before"])
    aft = np.array([np.nan if after[t] is None else after[t] for t in range(len(tris))])
    assert np.array_equal(aft, g["area_after"], equal_nan=True)
    assert flipped == g["area_flipped"].tolist()
    assert np.array_equal(np.array([m3[t] for t in range(len(tris))], dtype=np.uint8), g["area_matched3"])
    # a14
    combos = g["eager_combos"]
    tr = np.arange(3 * len(combos), dtype=np.int32).reshape(-1, 3)
    s = ops.eager_signs(rxy, tr, combos.reshape(-1, 1).astype(np.int32))
    assert np.array_equal(s.reshape(-1), g["eager_signs"])


@pytest.mark.parametrize("case", ["simulated_st", "simulated_elastic"])
def test_stored_reference_runs(hip, case):
    g = load_golden(case)
    a = pd.DataFrame({"X": g["aligned_xy"][:, 0], "Y": g["aligned_xy"][:, 1]})
    r = pd.DataFrame({"X": g["ref_xy"][:, 0], "Y": g["ref_xy"][:, 1]})
    m = pd.DataFrame({"aligned_idx": g["matches"][:, 0], "ref_idx": g["matches"][:, 1]})
    info = {int(k): {"vertices": vtx} for k, vtx in zip(g["tinfo_keys"], g["tinfo_vertices"])}
    from test_oracle_golden import _check_violations
    v = hip.verify_spatial_preservation(a, r, m, info)
    _check_violations(v, g, "stored_")


def test_sweeps_adversarial(hip, ops):
    g = load_golden("adversarial")
    pts, tris, rxy = g["adv_pts"], g["adv_tris"], g["sw_rxy"]
    a = pd.DataFrame({"X": pts[:, 0], "Y": pts[:, 1]})
    r = pd.DataFrame({"X": rxy[:, 0], "Y": rxy[:, 1]})
    m = pd.DataFrame({"aligned_idx": g["sw_matches"][:, 0], "ref_idx": g["sw_matches"][:, 1]})
    info = hip.precompute_triangle_info(a, tris, hip.build_simplex_map(len(a), tris))
    assert list(info.keys()) == g["sw_tinfo_keys"].tolist()
    from test_oracle_golden import _check_violations
    _check_violations(hip.verify_spatial_preservation(a, r, m, info), g, "sw_")
    a2r = {}
    for ai, ri in g["sw_matches"]:
        a2r[int(ai)] = int(ri)
    before, after, flipped, m3 = hip.triangle_area_flips(a, r, tris, a2r)
    assert np.array_equal(np.array([before[t] for t in range(len(tris))]), g["sw_area_before"])
    assert flipped == g["sw_area_flipped"].tolist()
    sweep = hip.LazyOrientationSweep(g["sw_pairs"], tris, g["adv_signs"], rxy, len(pts))
    checked, viol, _ = sweep.sweep(g["sw_x"])  # a later pair with x>0.5 overrides an earlier one
    assert checked == int(g["sw_lazy_checked"][0]) and [t[0] for t in viol] == g["sw_lazy_violating"].tolist()
    # empty triangle list
    e = hip.LazyOrientationSweep(g["sw_pairs"], np.zeros((0, 3), dtype=int), np.zeros(0), rxy, len(pts))
    assert e.sweep(g["sw_x"])[:2] == (0, [])


def test_sweeps_vs_oracle_seeded(ops, oracle):
    from scipy.spatial import Delaunay
    from same_amd import synth

    ref = synth.make_cells(60000, 4, seed=0)
    mov = synth.make_jittered(ref, seed=1)
    tris = Delaunay(mov["xy"]).simplices.astype(np.int32)  # ~114k triangles: several compaction chunks
    sign, _ = ops.tri_sign_weight(mov["xy"], None, tris)
    rng = np.random.default_rng(2)
    match = rng.integers(-1, len(ref["xy"]), len(mov["xy"])).astype(np.int32)
    near = ops.knn_prune(mov["xy"], ref["xy"], 25.0, 1, want_d2=False)[0][:, 0]
    match = np.where(rng.random(len(match)) < 0.8, near, match).astype(np.int32)
    sweep = ops.BoundSweep(tris, sign, ref["xy"], len(mov["xy"]))
    checked, viol, flag = sweep.sweep_match(match, want_flag=True)
    ochecked, oviol, oflag = oracle.orient_sweep(tris, sign, ref["xy"], match)
    assert checked == ochecked and np.array_equal(viol, oviol) and np.array_equal(flag, oflag)
    assert np.all(np.diff(viol) > 0)  # ascending
    e, tf, pf, counts = ops.xyorder_sweep(mov["xy"], ref["xy"], tris, match)
    oe, otf, opf, oc = oracle.xyorder_sweep(mov["xy"], ref["xy"], tris, match)
    assert np.array_equal(e, oe) and np.array_equal(tf, otf) and np.array_equal(pf, opf) and np.array_equal(counts, oc)
    b, a, m3, fl = ops.area_flip(mov["xy"], ref["xy"], tris, match)
    ob, oa, om3, ofl = oracle.area_flip(mov["xy"], ref["xy"], tris, match)
    assert np.array_equal(b, ob) and np.array_equal(a, oa, equal_nan=True) and np.array_equal(m3, om3) and np.array_equal(fl, ofl)
    # all unmatched / all matched to one point (every ref triangle collinear -> sign 0 -> skipped)
    assert sweep.sweep_match(np.full(len(match), -1, np.int32))[0] == 0
    assert sweep.sweep_match(np.zeros(len(match), np.int32))[0] == 0


# ------------------------------------------------------------------------------------------ a13
def test_window_plan_vs_oracle(hip, oracle):
    from same_amd import synth

    ref = synth.make_cells(6000, 3, seed=0, side=1000.0)
    mov = synth.make_jittered(ref, seed=1)
    mov["xy"][:, 0] = np.clip(mov["xy"][:, 0], 0, None)
    # carve a hole so that merge-right / merge-down trigger
    hole = (ref["xy"][:, 0] > 300) & (ref["xy"][:, 0] < 520) & (ref["xy"][:, 1] > 250) & (ref["xy"][:, 1] < 700)
    rxy = ref["xy"][~hole]
    merged = 0
    for ws, ov, mc in ((300, 100, 450), (250, 0, 330), (400, 150, 10), (180, 60, 170)):
        plan = hip.window_plan(rxy, mov["xy"], ws, ov, mc)
        oplan = oracle.window_plan(rxy, mov["xy"], ws, ov, mc)
        assert len(plan) == len(oplan) and len(plan) > 0
        for p, o in zip(plan, oplan):
            for key in ("i0", "j0", "i", "j", "window_id", "box", "trim", "n_ref", "n_mov"):
                assert p[key] == o[key], (ws, ov, mc, key)
        merged += sum((p["i"], p["j"]) != (p["i0"], p["j0"]) for p in plan)
    assert merged > 0  # the merge-right / merge-down branches were exercised
    # rows without coordinates: the reference's extent skips them (pandas min / max, src/same.py:481-482) and no window holds them
    # (tests/test_oracle_vs_reference_fuzz.py runs the reference's own loop on such layouts against the oracle)
    rng = np.random.default_rng(8)
    rn, mn = rxy.copy(), mov["xy"].copy()
    rn[rng.integers(0, len(rn), 9), 0] = np.nan
    mn[rng.integers(0, len(mn), 6), 1] = np.nan
    plan, oplan = hip.window_plan(rn, mn, 300, 100, 450), oracle.window_plan(rn, mn, 300, 100, 450)
    keys = ("window_id", "box", "trim", "n_ref", "n_mov")
    assert len(plan) == len(oplan) > 0 and all(p[key] == o[key] for p, o in zip(plan, oplan) for key in keys)
    with pytest.raises(OverflowError):
        hip.window_plan(np.vstack([rxy, [[np.inf, 0.0]]]), mov["xy"], 300, 100, 450)


# ------------------------------------------------------------------------------------------ end to end (pre-MIP)
@pytest.mark.parametrize("case", FULL_CASES)
def test_prepare_same_inputs_golden(hip, case):
    g = load_golden(case)
    a_df, r_df, cols = frames_from_golden(g)
    mad = None if g["params"][2] < 0 else g["params"][2]
    prep = hip.prepare_same_inputs(r_df, a_df, cols, optim_params=dict(radius=g["params"][0], knn=int(g["params"][1]),
                                   min_angle_deg=mad, ignore_same_type_triangles=False, dist_ct_coeff=g["params"][3]),
                                   verbose=False)
    assert np.array_equal(np.asarray(prep.valid_pairs, dtype=np.int64), g["pairs"])
    assert np.array_equal(np.array(prep.costs), g["all_costs"])
    assert np.array_equal(np.array(prep.aligned_delaunay, dtype=np.int64).reshape(-1, 3), g["tri_plain"])
    assert np.array_equal(np.array(prep.triangle_weights, dtype=np.float64), g["tri_weights"])
    assert np.array_equal(np.array(prep.source_signs), g["source_signs"])
    assert list(prep.triangle_info.keys()) == g["tinfo_keys"].tolist()
    assert prep.n_aligned == len(g["kept_aligned"]) and prep.n_ref == len(g["kept_ref"])
    assert prep.ref_coords_xy[0] == tuple(prep.ref_df[["X", "Y"]].iloc[0])


# ------------------------------------------------------------------------------------------ multi-GPU plumbing on one GPU
def test_rccl_gather_single_rank_and_device_sharded_path(ops, oracle):
    """RCCL communicator of size 1 on this GPU: init, all-gather (= copy), destroy; and the device form of the
    sharded prune+cost produces exactly the single-GPU pairs/costs (what N ranks then concatenate)."""
    from same_amd import _lib, synth
    from same_amd.dist import RcclGroup as RcclGather, hip_block_compute, sharded_knn_cost_device

    ctx = _lib.Context(0)  # own context: the communicator lives and dies with it
    ref = synth.make_cells(5000, 6, seed=0)
    mov = synth.make_jittered(ref, seed=1)
    g = RcclGather(ctx, 1, 0, lambda b: b)
    try:
        compute = hip_block_compute(ctx, mov["types"], ref["types"], mov["xy"], ref["xy"], 25.0, 16, 1.0)
        idx, cost = sharded_knn_cost_device(ctx, compute, len(mov["xy"]), 16, g)
        # overlapped form: gather on the communication stream while more compute is queued, then wait + reuse
        n = len(mov["xy"])
        didx, dcost, _ = compute(0, n, n)
        gidx, gcost = ctx.alloc(n * 16 * 4), ctx.alloc(n * 16 * 8)
        g.wait()                                            # nothing issued yet: a no-op
        g.allgather_dev_async(didx, gidx, n * 16 * 4)
        g.allgather_dev_async(dcost, gcost, n * 16 * 8)
        g.wait()                                            # compute stream now ordered after both gathers
        ctx.check(ctx.lib.same_dev_memset(ctx.handle, didx.ptr, 0, n * 16 * 4), "memset")   # safe to overwrite the send buffer
        ctx.sync()
        assert np.array_equal(gidx.download((n, 16), np.int32), idx) and np.array_equal(gcost.download((n, 16), np.float64), cost)
    finally:
        g.close()
    oidx, _, _ = oracle.knn_prune(mov["xy"], ref["xy"], 25.0, 16)
    assert np.array_equal(idx, oidx)
    rr, cc = np.nonzero(oidx >= 0)
    want = oracle.pair_cost_arrays(mov["types"], ref["types"], mov["xy"], ref["xy"], np.column_stack((rr, oidx[rr, cc])), 1.0)
    assert np.array_equal(cost[rr, cc], want) and np.isinf(cost[oidx < 0]).all()
    ctx.close()


def test_sharded_sweeps_rccl_single_rank_and_block_forms(ops, oracle):
    """The triangle-block sharded sweeps: (i) ShardedSweeps over a size-1 RCCL communicator -- every exchange a
    real ncclAllGather / ncclAllReduce -- equals the single-GPU sweeps and the oracle; (ii) the block entry points run as
    the blocks of 1, 2, 3 and 8 ranks on this one GPU (flags written block by block into one array, then compacted) give
    the same checked count and ascending flipped list."""
    import ctypes
    from scipy.spatial import Delaunay
    from same_amd import _lib, synth
    from same_amd.dist import RcclGroup, ShardedSweeps, tri_block

    ctx = _lib.Context(0)
    L, H = ctx.lib, ctx.handle
    ref = synth.make_cells(30000, 4, seed=3)
    mov = synth.make_jittered(ref, seed=4, sigma=6.0)
    n_m = len(mov["xy"])
    tris = np.ascontiguousarray(Delaunay(mov["xy"]).simplices, dtype=np.int32)
    Tr = len(tris)
    idx, _, cnt = ops.knn_prune(mov["xy"], ref["xy"], 25.0, 4, ctx=ctx)
    match = np.where(cnt > 0, idx[:, 0], -1).astype(np.int32)
    match[::7] = -1                                                   # some unmatched vertices
    sign, _ = ops.tri_sign_weight(mov["xy"], None, tris, ctx=ctx)
    want_checked, want_viol, want_flag = oracle.orient_sweep(tris, sign, ref["xy"], match)
    we, wtf, wpf, wcounts = oracle.xyorder_sweep(mov["xy"], ref["xy"], tris, match)
    wb, wa, wm3, wfl = oracle.area_flip(mov["xy"], ref["xy"], tris, match)
    assert len(want_viol) > 50 and wcounts[1] > 0

    sweep = ctypes.c_void_p()
    ctx.check(L.same_sweep_bind(H, tris.ctypes.data, Tr, sign.ctypes.data, ref["xy"].ctypes.data, len(ref["xy"]), n_m, None, 0,
                                ctypes.byref(sweep)), "bind")
    dax, drx, dtris, dmatch = ctx.to_device(mov["xy"]), ctx.to_device(ref["xy"]), ctx.to_device(tris), ctx.to_device(match)
    comm = RcclGroup(ctx, 1, 0, lambda b: b)
    try:
        assert comm.rccl_version() > 20000
        sh = ShardedSweeps(ctx, comm, sweep, dax, drx, dtris, Tr, n_m)
        for _ in range(2):
            checked, viol = sh.run(dmatch)
            out = sh.download()
            assert checked == want_checked and np.array_equal(viol, want_viol) and np.array_equal(out["flag"], want_flag)
            assert np.array_equal(out["edge"], we) and np.array_equal(out["tri_flag"], wtf) and np.array_equal(out["point_flag"], wpf)
            assert np.array_equal(out["counts"], wcounts)
            assert np.array_equal(out["before"], wb) and np.array_equal(out["after"], wa, equal_nan=True)
            assert np.array_equal(out["matched3"], wm3) and np.array_equal(out["flipped"], wfl)
        # all-reduce forms on their own
        buf = ctx.to_device(np.array([5, 7, 11], np.uint64))
        comm.allreduce_dev(buf, 3, _lib.DT_U64, _lib.OP_SUM)
        ctx.sync()
        assert buf.download((3,), np.uint64).tolist() == [5, 7, 11]
        assert L.same_allreduce_dev(H, buf.ptr, 3, 99, 0) == -22 and L.same_allreduce_dev(H, buf.ptr, 3, 0, 99) == -22
    finally:
        comm.close()
    # (ii) block forms, as G ranks would issue them
    chk, nv = ctypes.c_int64(0), ctypes.c_int64(0)
    viol = np.empty(Tr, np.int32)
    for world in (1, 2, 3, 8):
        _, _, block = tri_block(Tr, world, 0)
        dflag = ctx.alloc(block * world)
        ctx.check(L.same_dev_memset(H, dflag.ptr, 0, dflag.nbytes), "memset")
        for r in range(world):
            t0, t1, _ = tri_block(Tr, world, r)
            ctx.check(L.same_orient_flags_dev(sweep, dmatch.ptr, t0, t1, dflag.ptr), "flags")
        ctx.check(L.same_orient_from_flags_dev(sweep, dflag.ptr, ctypes.byref(chk), viol.ctypes.data, ctypes.byref(nv)), "from_flags")
        assert chk.value == want_checked and np.array_equal(viol[: nv.value], want_viol)
        assert np.array_equal(dflag.download((Tr,), np.uint8), want_flag)
    # blocks need not start on a wave: ragged blocks give the same flags and list
    ctx.check(L.same_dev_memset(H, dflag.ptr, 0, dflag.nbytes), "memset")
    for t0, t1 in ((0, 10), (10, 301), (301, Tr)):
        ctx.check(L.same_orient_flags_dev(sweep, dmatch.ptr, t0, t1, dflag.ptr), "flags")
    ctx.check(L.same_orient_from_flags_dev(sweep, dflag.ptr, ctypes.byref(chk), viol.ctypes.data, ctypes.byref(nv)), "from_flags")
    assert chk.value == want_checked and np.array_equal(viol[: nv.value], want_viol)
    assert np.array_equal(dflag.download((Tr,), np.uint8), want_flag)
    assert L.same_orient_flags_dev(sweep, dmatch.ptr, 0, Tr + 1, dflag.ptr) == -22
    # first candidate of a padded list
    didx = ctx.to_device(idx)
    dm2 = ctx.alloc(n_m * 4)
    ctx.check(L.same_first_candidate_dev(H, didx.ptr, n_m, 4, dm2.ptr), "first")
    assert np.array_equal(dm2.download((n_m,), np.int32), idx[:, 0])
    L.same_sweep_unbind(sweep)
    ctx.close()


_SHARDED_WINDOW_PARAMS = dict(radius=25, knn=6, window_size=300, overlap=20, min_cells_per_window=20, hip_cost_dtype="float32")


def _sharded_device_worker(rank, world, rdv, out_dir, transport="host"):
    """One rank of `world`: device-side sharded prune + cost and ShardedSweeps; every rank's complete outputs are written for
    the parent to check.  transport "host": all ranks on GPU 0, exchanging through the host transport (RCCL refuses several
    ranks on one device); "rccl": rank r on GPU r, a real RCCL communicator between the devices."""
    import ctypes
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import numpy as np
    from scipy.spatial import Delaunay
    from same_amd import _lib, ops, synth
    from same_amd.dist import HostGroup, HostTransport, RcclGroup, ShardedSweeps, hip_block_compute, sharded_knn_cost_device

    ctx = _lib.Context(rank if transport == "rccl" else 0)
    L, H = ctx.lib, ctx.handle
    with HostGroup(rank, world, rdv_dir=rdv, timeout=300) as group:
        if transport == "rccl":
            comm = RcclGroup(ctx, world, rank, lambda b: group.bcast_bytes(b or b""))
            info = comm.info()     # what the communicator itself says: ncclCommCount / ncclCommUserRank / ncclCommCuDevice
            assert info["nranks"] == world and info["rank"] == rank and info["device"] == rank, info
        else:
            comm = HostTransport(ctx, group)
        ref = synth.make_cells(9000, 5, seed=3)
        mov = synth.make_jittered(ref, seed=4, sigma=6.0)
        n_m = len(mov["xy"])
        compute = hip_block_compute(ctx, mov["types"], ref["types"], mov["xy"], ref["xy"], 25.0, 8, 1.0)
        idx, cost = sharded_knn_cost_device(ctx, compute, n_m, 8, comm)          # row blocks of the cost build + prune
        tris = np.ascontiguousarray(Delaunay(mov["xy"]).simplices, dtype=np.int32)
        Tr = len(tris)
        match = np.where(idx[:, 0] >= 0, idx[:, 0], -1).astype(np.int32)
        match[::7] = -1
        sign, _ = ops.tri_sign_weight(mov["xy"], None, tris, ctx=ctx)
        sweep = ctypes.c_void_p()
        ctx.check(L.same_sweep_bind(H, tris.ctypes.data, Tr, sign.ctypes.data, ref["xy"].ctypes.data, len(ref["xy"]), n_m, None, 0,
                                    ctypes.byref(sweep)), "bind")
        dax, drx, dtris, dmatch = ctx.to_device(mov["xy"]), ctx.to_device(ref["xy"]), ctx.to_device(tris), ctx.to_device(match)
        sh = ShardedSweeps(ctx, comm, sweep, dax, drx, dtris, Tr, n_m)           # triangle blocks of the three sweeps
        checked, viol = sh.run(dmatch)
        out = sh.download()
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), idx=idx, cost=cost, checked=checked, viol=viol, match=match, tris=tris,
                 sign=sign, **out)
        # whole windows in runs of the plan + the window merge per rank, the seam rows over the same communicator (dist.MergeChannel)
        from same_amd.dist import sharded_merged_window_incumbent

        merged = sharded_merged_window_incumbent(synth.to_frame(ref), synth.to_frame(mov), commonCT=synth.type_columns(5), group=group,
                                                 ctx=ctx,
                                                 comm=comm, optim_params=_SHARDED_WINDOW_PARAMS)
        merged.to_pickle(os.path.join(out_dir, f"merged{rank}.pkl"))
        group.barrier()
        L.same_sweep_unbind(sweep)
        comm.close()
    ctx.close()


def _check_sharded_rank_outputs(tmp_path, oracle, world):
    from same_amd import synth

    ref = synth.make_cells(9000, 5, seed=3)
    mov = synth.make_jittered(ref, seed=4, sigma=6.0)
    oidx, _, _ = oracle.knn_prune(mov["xy"], ref["xy"], 25.0, 8)
    rr, cc = np.nonzero(oidx >= 0)
    ocost = oracle.pair_cost_arrays(mov["types"], ref["types"], mov["xy"], ref["xy"], np.column_stack((rr, oidx[rr, cc])), 1.0)
    for r in range(world):
        o = np.load(tmp_path / f"rank{r}.npz")
        assert np.array_equal(o["idx"], oidx) and np.array_equal(o["cost"][rr, cc], ocost) and np.isinf(o["cost"][oidx < 0]).all()
        tris, match, sign = o["tris"], o["match"], o["sign"]
        och, oviol, oflag = oracle.orient_sweep(tris, sign, ref["xy"], match)
        assert int(o["checked"]) == och and np.array_equal(o["viol"], oviol) and np.array_equal(o["flag"], oflag) and len(oviol) > 10
        oe, otf, opf, oc = oracle.xyorder_sweep(mov["xy"], ref["xy"], tris, match)
        assert np.array_equal(o["edge"], oe) and np.array_equal(o["tri_flag"], otf) and np.array_equal(o["point_flag"], opf)
        assert np.array_equal(o["counts"], oc)
        ob, oa, om3, ofl = oracle.area_flip(mov["xy"], ref["xy"], tris, match)
        assert np.array_equal(o["before"], ob) and np.array_equal(o["after"], oa, equal_nan=True)
        assert np.array_equal(o["matched3"], om3) and np.array_equal(o["flipped"], ofl)
    # the window merge dealt over the ranks, its seam rows exchanged over the ranks' transport: the parts are the single process's table
    import pandas as pd

    import same_amd
    from same_amd.merge import join_merged_parts

    want = same_amd.sliding_window_incumbent(synth.to_frame(ref), synth.to_frame(mov), commonCT=synth.type_columns(5),
                                             optim_params=dict(_SHARDED_WINDOW_PARAMS), merge=True)
    parts = [pd.read_pickle(tmp_path / f"merged{r}.pkl") for r in range(world)]
    assert len(want) > 5000 and all(0 < len(p) < len(want) for p in parts) and join_merged_parts(parts).equals(want)


def test_rccl_between_devices_sharded_paths(tmp_path, oracle):
    """RCCL WITH PEERS (needs at least two GPUs; skipped on a one-GPU box): rank r on device r, a real communicator between them
    -- ncclAllGather of the pruned candidate lists (sharded_knn_cost_device), then ShardedSweeps' grouped flag all-gathers and
    counter / point-flag all-reduces -- and every rank's complete outputs equal the oracle's, bit for bit.  Fresh child
    processes (the parent never creates a communicator), at most 4 ranks."""
    import multiprocessing as mp
    import tempfile
    from same_amd import _lib

    ndev = _lib.device_count()
    if ndev < 2:
        pytest.skip(f"{ndev} GPU visible: RCCL between devices needs at least two")
    world = min(ndev, 4)
    ctx = mp.get_context("spawn")
    with tempfile.TemporaryDirectory(prefix="same_rdv_rccl_") as rdv:
        procs = [ctx.Process(target=_sharded_device_worker, args=(r, world, rdv, str(tmp_path), "rccl")) for r in range(world)]
        [p.start() for p in procs]
        [p.join(600) for p in procs]
        for p in procs:
            if p.is_alive():
                p.kill()
        assert [p.exitcode for p in procs] == [0] * world
    _check_sharded_rank_outputs(tmp_path, oracle, world)


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_device_paths_with_several_ranks_on_one_gpu(tmp_path, oracle, world):
    """The DEVICE forms of the sharded cost build and of the sharded sweeps with 2 and 3 real ranks (one process each, sharing
    this GPU, exchanging through the host transport): every rank ends up with the single-GPU outputs, equal to the oracle --
    row blocks, triangle blocks, padded tails, gathers, OR / sum reductions and the rebuilt ascending list included."""
    import multiprocessing as mp
    import tempfile
    from same_amd import synth

    ctx = mp.get_context("spawn")
    with tempfile.TemporaryDirectory(prefix="same_rdv_gpu_") as rdv:
        procs = [ctx.Process(target=_sharded_device_worker, args=(r, world, rdv, str(tmp_path))) for r in range(world)]
        [p.start() for p in procs]
        [p.join(600) for p in procs]
        assert [p.exitcode for p in procs] == [0] * world
    _check_sharded_rank_outputs(tmp_path, oracle, world)


@pytest.mark.parametrize("mode", ["auto", "grid", "brute"])
def test_knn_index_equals_unindexed_prune(ops, oracle, mode, monkeypatch):
    """Caller-held index of the reference set (same_knn_index_build): pruning against it is bit-identical to the
    rebuild-every-call entry point, for whole sets, row blocks, several k, and the degenerate sets that fall back to brute force."""
    import ctypes
    from same_amd import _lib, synth

    if mode != "auto":
        monkeypatch.setenv("SAME_KNN_MODE", mode)
    ctx = _lib.Context(0)
    L, H = ctx.lib, ctx.handle
    rng = np.random.default_rng(12)
    r = synth.make_cells(20000, 3, seed=20)
    cases = [(synth.make_jittered(r, seed=21)["xy"], r["xy"], 25.0),
             (rng.uniform(-50, 1050, (3000, 2)), rng.uniform(0, 1000, (5000, 2)), 40.0),
             (rng.uniform(0, 100, (300, 2)), np.tile([[5.0, 5.0]], (2600, 1)), 200.0),       # all refs coincide: no usable grid
             (rng.uniform(0, 10, (500, 2)), rng.uniform(0, 10, (1000, 2)), 6.0),              # small set: brute-force plan
             (rng.uniform(0, 10, (50, 2)), np.zeros((0, 2)), 3.0)]                            # empty reference set
    for axy, rxy, radius in cases:
        dax, drx = ctx.to_device(axy), ctx.to_device(rxy)
        ix = ctypes.c_void_p()
        ctx.check(L.same_knn_index_build(H, drx.ptr, len(rxy), radius, ctypes.byref(ix)), "index_build")
        n = len(axy)
        for k in (1, 8, 32, 100):
            for b, e in ((0, n), (n // 3, n // 3 + min(200, n - n // 3))):
                rows = e - b
                d1, d2, c1 = ctx.alloc(rows * k * 4), ctx.alloc(rows * k * 8), ctx.alloc(rows * 4)
                ctx.check(L.same_knn_prune_indexed_dev(H, ix, dax.ptr, b, e, k, d1.ptr, d2.ptr, c1.ptr), "indexed")
                oi, od2, ocnt = oracle.knn_prune(axy, rxy, radius, k, b, e)
                assert np.array_equal(d1.download((rows, k), np.int32), oi)
                assert np.array_equal(d2.download((rows, k), np.float64), od2) and np.array_equal(c1.download((rows,), np.int32), ocnt)
        bad = _lib.Context(0)
        assert L.same_knn_prune_indexed_dev(bad.handle, ix, dax.ptr, 0, 1, 4, d1.ptr, None, c1.ptr) == -22   # another context's index
        bad.close()
        L.same_knn_index_destroy(ix)
    ctx.close()


# ------------------------------------------------------------------------------------------ SURVEY 8(f1)
def test_check_triangle_violations_golden(hip):
    from same_amd.eval_utils import check_triangle_violations
    from test_oracle_golden import EVAL_CASES, EVAL_KEYS, eval_inputs

    g = load_golden("eval_tri")
    out_df, mc = eval_inputs(g)
    for tag, kw in EVAL_CASES:
        df, stats = check_triangle_violations(out_df, mc, **kw)
        assert np.array_equal(df["in_violating_triangle"].to_numpy().astype(np.uint8), g[f"viol_{tag}"]), tag
        assert np.array_equal(np.array([stats[k] for k in EVAL_KEYS], dtype=np.float64), g[f"stats_{tag}"]), tag


def test_tri_flip_stats_vs_oracle_seeded(ops, oracle):
    from scipy.spatial import Delaunay
    from same_amd import synth

    ref = synth.make_cells(40000, 5, seed=0)
    mov = synth.make_jittered(ref, seed=1)
    tris = Delaunay(mov["xy"]).simplices.astype(np.int32)
    rng = np.random.default_rng(3)
    near = ops.knn_prune(mov["xy"], ref["xy"], 25.0, 3, want_d2=False)[0]
    pick = near[np.arange(len(near)), rng.integers(0, 3, len(near))]
    matched = (pick >= 0) & (rng.random(len(pick)) > 0.05)
    mapped = np.where(matched[:, None], ref["xy"][np.maximum(pick, 0)], 0.0)
    mapped[rng.integers(0, len(mapped), 20)] = np.nan
    for tid in (None, mov["cell_type"]):
        f, nt, nf = ops.tri_flip_stats(mov["xy"], mapped, matched, tris, tid)
        of, ont, onf = oracle.tri_flip_stats(mov["xy"], mapped, matched, tris, tid)
        assert np.array_equal(f, of) and np.array_equal(nt, ont) and np.array_equal(nf, onf)
    assert (f & 4).sum() > 100


def test_greedy_match_equals_sequential_scan(ops, oracle):
    """Device greedy (parallel local-minimum rule) == the reference's sort + scan, incl. exact cost ties."""
    rng = np.random.default_rng(0)
    for n_m, n_r, k, quant in ((3000, 2500, 8, None), (2000, 2000, 16, 0.5), (500, 40, 6, None), (50, 3000, 32, 1.0)):
        pi = np.repeat(np.arange(n_m), k)
        pj = rng.integers(0, n_r, n_m * k)
        keep = rng.random(len(pi)) > 0.2
        pairs = np.column_stack((pi, pj))[keep]
        costs = rng.gamma(2.0, 20.0, len(pairs))
        if quant:  # heavy ties: the (cost, pair index) order must decide
            costs = np.round(costs / quant) * quant
        sizes = rng.integers(1, 4, n_m).astype(float)
        penalty = float(np.median(costs)) / 1.5
        vp = [tuple(p) for p in pairs.tolist()]
        want, want_un = oracle.compute_mip_start_pairs(valid_pairs=vp, costs=list(costs), n_aligned=n_m, n_ref=n_r,
                                                       aligned_sizes=sizes, no_match_penalty=penalty, max_matches=1,
                                                       init_method="greedy", verbose=False)
        import same_amd
        got, got_un = same_amd.compute_mip_start_pairs(valid_pairs=vp, costs=list(costs), n_aligned=n_m, n_ref=n_r,
                                                       aligned_sizes=sizes, no_match_penalty=penalty, max_matches=1,
                                                       init_method="greedy", verbose=False)
        assert got == want and got_un == want_un
        assert len(got) > 0
    # degenerate: no pair at all / every row prefers to stay unmatched
    mp, rounds = ops.greedy_match(np.zeros((0, 2), np.int32), np.zeros(0), 5, 5, np.ones(5, np.uint8))
    assert (mp == -1).all()
    mp, _ = ops.greedy_match(np.array([[0, 0], [1, 1]], np.int32), np.array([1.0, 2.0]), 2, 2, np.zeros(2, np.uint8))
    assert (mp == -1).all()


def test_greedy_chain_is_resolved_in_batches(ops, oracle):
    """A monotone chain -- pair (i, i) costs 2i, pair (i + 1, i) costs 2i + 1 -- lets the parallel rule select exactly ONE pair a
    round (every (i, i) is pre-empted at its row by (i, i - 1) until that one dies): 5 000 rounds for 9 999 pairs.  The result is
    still the reference's scan (src/init_helpers.py:109-133), and the host reads the device once per BATCH of rounds, not once
    per round: read-backs <= rounds / 4."""
    import same_amd
    from same_amd import _lib

    n = 5000
    pairs = np.empty((2 * n - 1, 2), np.int32)
    pairs[0::2] = np.column_stack((np.arange(n), np.arange(n)))
    pairs[1::2] = np.column_stack((np.arange(1, n), np.arange(n - 1)))
    costs = np.arange(2 * n - 1, dtype=np.float64)
    ctx = _lib.default_context()
    before = ctx.stats()
    mp, rounds = ops.greedy_match(pairs, costs, n, n, np.ones(n, np.uint8))
    after = ctx.stats()
    assert np.array_equal(mp, 2 * np.arange(n)) and rounds == n
    readbacks = after["greedy_readbacks"] - before["greedy_readbacks"]
    assert 1 <= readbacks <= rounds // 4, (readbacks, rounds)
    kw = dict(valid_pairs=[tuple(p) for p in pairs.tolist()], costs=list(costs), n_aligned=n, n_ref=n, aligned_sizes=np.ones(n),
              no_match_penalty=1e9, max_matches=1, init_method="greedy", verbose=False)
    assert same_amd.compute_mip_start_pairs(**kw) == oracle.compute_mip_start_pairs(**kw)
    # the usual case -- a few rounds -- is ONE read-back
    rng = np.random.default_rng(3)
    pi, pj = np.repeat(np.arange(2000), 8), rng.integers(0, 2000, 16000)
    before = ctx.stats()
    mp, rounds = ops.greedy_match(np.column_stack((pi, pj)).astype(np.int32), rng.gamma(2.0, 20.0, 16000), 2000, 2000,
                                  np.ones(2000, np.uint8))
    readbacks = ctx.stats()["greedy_readbacks"] - before["greedy_readbacks"]
    # the batch schedule is 4, 4, 8, ... rounds per read: up to 4 rounds take ONE read-back, and never more than one per 4 rounds + 1
    assert rounds >= 1 and 1 <= readbacks <= 1 + rounds // 4, (rounds, readbacks)
    if rounds < 4:
        assert readbacks == 1, (rounds, readbacks)
    # a case that certainly resolves inside the first batch: disjoint pairs -> one round, ONE read
    n1 = 3000
    before = ctx.stats()
    mp, rounds = ops.greedy_match(np.column_stack((np.arange(n1), np.arange(n1))).astype(np.int32), np.arange(n1, dtype=np.float64), n1,
                                  n1, np.ones(n1, np.uint8))
    assert rounds == 1 and np.array_equal(mp, np.arange(n1)) and ctx.stats()["greedy_readbacks"] - before["greedy_readbacks"] == 1


# ------------------------------------------------------------------------------------------ SURVEY 8(f2)
@pytest.mark.parametrize("which", ["s3", "s6", "s9", "s1", "seeded"])
def test_greedy_triangle_collapse_golden(hip, which):
    from same_amd.metacell_utils import greedy_triangle_collapse
    from test_oracle_golden import check_metacells, metacell_inputs

    g = load_golden("metacell")
    df, kw, num_cols, other_cols = metacell_inputs(which)
    mc = greedy_triangle_collapse(df, return_object=True, verbose=False, **kw)
    check_metacells(g, "seeded" if which == "seeded" else f"q_{which}", mc.metacell_df, mc.metacell_delaunay,
                    None if which == "seeded" else mc.original_delaunay, num_cols, other_cols)
    assert mc.to_summary_dict()["n_metacells"] == len(mc.metacell_df)
    assert mc.metacell_delaunay_to_xy().shape == (len(mc.metacell_delaunay), 3, 2)
    assert mc.original_delaunay_to_row_indices().shape == np.asarray(mc.original_delaunay).shape


def test_greedy_triangle_collapse_vs_oracle_seeded(hip, oracle, ops):
    """A larger collapse (several iterations, metacells of up to 8 cells) against the oracle, and the metacell object
    fed straight into the pre-MIP path the way the example scripts do."""
    import same_amd
    from same_amd import synth
    from same_amd.metacell_utils import greedy_triangle_collapse

    cells = synth.make_cells(6000, 3, seed=9)
    df = synth.to_frame(cells)
    kw = dict(max_metacell_size=8, r_max=40, min_angle_deg=10)
    mc = greedy_triangle_collapse(df, return_object=True, verbose=False, **kw)
    omdf, otri, oorig = oracle.greedy_triangle_collapse(df, **kw)
    m = mc.metacell_df
    assert list(m.columns) == list(omdf.columns) and len(m) == len(omdf) < 0.8 * len(df)
    for col in ("X", "Y", "size", "c1", "c2", "c3", "metacell_id"):
        assert np.array_equal(m[col].to_numpy(), omdf[col].to_numpy()), col
    assert m["members"].tolist() == omdf["members"].tolist() and m["cell_type"].tolist() == omdf["cell_type"].tolist()
    assert np.array_equal(np.asarray(mc.metacell_delaunay), otri) and np.array_equal(np.asarray(mc.original_delaunay), oorig)
    assert int(m["size"].max()) > 3
    ref_mc = greedy_triangle_collapse(synth.to_frame(synth.make_jittered(cells, seed=2)).assign(Cell_Num_Old=lambda d: np.arange(len(d))),
                                      return_object=True, verbose=False, **kw)
    prep = same_amd.prepare_same_inputs(ref_mc.metacell_df, mc, synth.type_columns(3),
                                        optim_params=dict(radius=40, knn=6, cell_id_col=None), verbose=False)
    assert prep.using_precomputed and prep.optim_params["cell_id_col"] == "metacell_id" and len(prep.valid_pairs) > 1000
    # the device selection equals a sequential scan on a random hypergraph with heavy key ties
    rng = np.random.default_rng(0)
    items = rng.integers(0, 3000, size=(20000, 3)).astype(np.int32)
    items = items[(items[:, 0] != items[:, 1]) & (items[:, 1] != items[:, 2]) & (items[:, 0] != items[:, 2])]
    keys = np.round(rng.gamma(2.0, 3.0, len(items)), 1)
    sel, rounds = ops.greedy_disjoint(items, keys, 3000)
    used, want = set(), np.zeros(len(items), bool)
    for q in np.argsort(keys, kind="stable"):
        a, b, c = items[q]
        if a not in used and b not in used and c not in used:
            want[q] = True
            used.update((a, b, c))
    assert np.array_equal(sel, want) and rounds > 1


# ------------------------------------------------------------------------------------------ C-ABI error convention
def test_c_abi_error_codes(hip):
    """Negative codes, outputs never written out of bounds, message available; no exception crosses the ABI."""
    import ctypes
    from same_amd import _lib

    ctx = _lib.default_context()
    L, H = ctx.lib, ctx.handle
    z = np.zeros((4, 2))
    idx = np.full((4, 3), 7, np.int32); cnt = np.full(4, 7, np.int32)
    EINVAL, ERANGE = -22, -34
    # k = 0
    assert L.same_knn_prune(H, z.ctypes.data, 4, z.ctypes.data, 4, 0, 4, 1.0, 0, idx.ctypes.data, None, cnt.ctypes.data) == EINVAL
    # row_end > n_m
    assert L.same_knn_prune(H, z.ctypes.data, 4, z.ctypes.data, 4, 0, 9, 1.0, 3, idx.ctypes.data, None, cnt.ctypes.data) == EINVAL
    assert L.same_knn_prune(H, z.ctypes.data, 4, z.ctypes.data, 4, 0, 4, float("nan"), 3, idx.ctypes.data, None, cnt.ctypes.data) == EINVAL
    # NULL input
    assert L.same_knn_prune(H, None, 4, z.ctypes.data, 4, 0, 4, 1.0, 3, idx.ctypes.data, None, cnt.ctypes.data) == EINVAL
    # outputs untouched
    assert (idx == 7).all() and (cnt == 7).all()
    # NULL ctx
    assert L.same_knn_prune(None, z.ctypes.data, 4, z.ctypes.data, 4, 0, 4, 1.0, 3, idx.ctypes.data, None, cnt.ctypes.data) == EINVAL
    tri = np.array([[0, 1, 9]], np.int32)
    out = np.zeros(1, np.int8)
    assert L.same_tri_sign_weight(H, z.ctypes.data, None, 4, tri.ctypes.data, 1, out.ctypes.data, None) == ERANGE
    assert b"triangles[2] = 9" in L.same_last_error(H)
    m = np.array([0, 1, 2, 99], np.int32)
    o8 = np.zeros(16, np.uint8); o64 = np.zeros(3, np.int64)
    assert L.same_xyorder_sweep(H, z.ctypes.data, 4, z.ctypes.data, 4, np.array([[0, 1, 2]], np.int32).ctypes.data, 1, m.ctypes.data,
                                o8.ctypes.data, o8.ctypes.data, o8.ctypes.data, o64.ctypes.data) == ERANGE
    chk, nv = ctypes.c_int64(0), ctypes.c_int64(0)
    # no sweep handle
    assert L.same_orient_sweep(None, m.ctypes.data, 4, ctypes.byref(chk), None, ctypes.byref(nv), None) == EINVAL
    sw = ctypes.c_void_p()
    t1, s1 = np.array([[0, 1, 2]], np.int32), np.ones(1, np.int8)
    assert L.same_sweep_bind(H, t1.ctypes.data, 1, s1.ctypes.data, z.ctypes.data, 4, 4, None, 0, ctypes.byref(sw)) == 0
    v1 = np.zeros(1, np.int32)
    # not the bound length
    assert L.same_orient_sweep(sw, m.ctypes.data, 3, ctypes.byref(chk), v1.ctypes.data, ctypes.byref(nv), None) == EINVAL
    # match[3] = 99
    assert L.same_orient_sweep(sw, m.ctypes.data, 4, ctypes.byref(chk), v1.ctypes.data, ctypes.byref(nv), None) == ERANGE
    # bound without pairs (P = 0)
    assert L.same_orient_sweep_x(sw, z.ctypes.data, 2, ctypes.byref(chk), v1.ctypes.data, ctypes.byref(nv), None, None, None) == EINVAL
    L.same_sweep_unbind(sw)
    big = ctypes.c_void_p()
    # 64 TiB: ENOMEM (or EIO)
    assert L.same_dev_alloc(H, 1 << 46, ctypes.byref(big)) in (-12, -5)
    assert L.same_dense_cost_f64_dev(H, None, None, 20, None, None, 10, 0, 5, 1.0, None, 10) == EINVAL
    assert L.same_dense_cost_f64_dev(H, z.ctypes.data, z.ctypes.data, 5000, z.ctypes.data, z.ctypes.data, 10, 0, 5, 1.0, z.ctypes.data,
                                     10) == EINVAL  # T > SAME_MAX_TYPES
    assert L.same_ctx_create(99, ctypes.byref(big)) == EINVAL


def test_check_alignment_golden(hip):
    from same_amd import synth
    from same_amd.eval_utils import check_alignment

    g = load_golden("eval_tri")
    q = synth.to_frame(synth.make_cells(700, 4, seed=31, side=100.0))
    t = synth.to_frame(synth.make_cells(900, 4, seed=32, side=100.0))
    for k in (1, 5):
        df, score = check_alignment(q, t, "X", "Y", kNN=k)
        assert np.array_equal(df[f"_{k}NN_match"].to_numpy().astype(np.uint8), g[f"align_match_{k}"])
        assert score == float(g[f"align_score_{k}"][0])
    assert df.shape[0] == 700 and np.array_equal(check_alignment(q, t, "X", "Y", kNN=1)[0]["_1NN_match_ctype"].to_numpy().astype(str),
                                                 g["align_ctype_1"])
    with pytest.raises(ValueError):
        check_alignment(q.drop(columns=["cell_type"]), t, "X", "Y")


def test_release_scratch_and_rebind(ops, oracle):
    from same_amd import _lib
    ctx = _lib.default_context()
    rng = np.random.default_rng(1)
    xy, rxy = rng.uniform(0, 50, (300, 2)), rng.uniform(0, 50, (200, 2))
    tris = rng.integers(0, 300, (500, 3)).astype(np.int32)
    sign, _ = ops.tri_sign_weight(xy, None, tris)
    match = rng.integers(-1, 200, 300).astype(np.int32)
    sw = ops.BoundSweep(tris, sign, rxy, 300)
    before = sw.sweep_match(match)
    ctx.release_scratch()                      # drops every staging block, including the bound sweep state
    after = sw.sweep_match(match)              # re-binds transparently
    assert before[0] == after[0] and np.array_equal(before[1], after[1])
    assert np.array_equal(ops.knn_prune(xy, rxy, 5.0, 4)[0], oracle.knn_prune(xy, rxy, 5.0, 4)[0])


def test_plain_c_abi_demo_runs(tmp_path):
    """The C ABI from a plain C11 program: prune, pair costs, sign, bind, sweep, dense tile, window-merge de-duplication, error code."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "abi_demo"
    lib_dir = os.path.join(root, "same_amd")
    subprocess.check_call(["gcc", "-std=c11", f"-I{os.path.join(root, 'include')}", os.path.join(root, "examples", "abi_demo.c"), "-o",
                           str(exe),
                           f"-L{lib_dir}", "-lsame_hip", f"-Wl,-rpath,{lib_dir}", "-lm"])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "checked 4 triangles" in r.stdout and "bad pair -> -34" in r.stdout
    assert "merge de-duplication keeps 4 of 6 rows: 3 2 1 4" in r.stdout
    assert "window: 6 aligned, 7 ref, 6 kept" in r.stdout and "(costs equal the pair list's); 4 of 4 triangles kept" in r.stdout
    assert sum(line.startswith("pair (") for line in r.stdout.splitlines()) >= 6
    # the demo's aligned cells sit on two horizontal lines: every edge along a line is an order tie of the XY-order sweep
    import re

    own = re.search(r"own triangulator: the cells above are left to Qhull; bent off the line answered, 5 triangles; "
                    r"order ties of the window: (\d+)", r.stdout)
    assert own and int(own.group(1)) > 0
    # the two windows of the batch merged on the device: every pair proposed twice, window 0's rows stay, nothing is left to the host
    said = "window merge on the device: 12 rows from two windows -> 6 after the de-duplication, 0 left to the host, 6 merged rows"
    assert said in r.stdout
    assert "(window 0's, aligned rows ascending)" in r.stdout


def test_negative_radius_acts_as_its_magnitude(hip):
    """cKDTree compares squared distances (the reference's query_ball_point with r=-10 returns the r=10 ball)."""
    from conftest import frames_from_golden, load_golden

    g = load_golden("cfg1_500")
    a_df, r_df, _ = frames_from_golden(g)
    pos = hip.find_knn_within_radius(a_df, r_df, 10, 8, verbose=False)
    neg = hip.find_knn_within_radius(a_df, r_df, -10, 8, verbose=False)
    assert np.array_equal(pos[2], neg[2]) and len(pos[0]) == len(neg[0]) and np.array_equal(pos[2], g["pairs"])


@pytest.mark.parametrize("T,k,n_m,n_r", [(20, 32, 3000, 2500), (2, 8, 700, 900), (3, 5, 1000, 50), (37, 16, 800, 1200),
                                         (127, 4, 1100, 300), (130, 4, 1100, 300), (1, 64, 90, 400), (0, 8, 600, 600), (20, 32, 100, 100)])
def test_padded_cost_kernels_vs_oracle(hip, oracle, T, k, n_m, n_r, monkeypatch):
    """same_padded_cost_f64_dev (the all-gather payload): LDS-staged kernel and the plain gather kernel against the oracle's
    pair costs, for type counts on both sides of every dispatch boundary, ragged last waves and sub-blocks of rows."""
    from same_amd import _lib

    rng = np.random.default_rng(T * 1000 + k)
    A, R = rng.gamma(0.4, 20.0, (n_m, T)), rng.gamma(0.4, 20.0, (n_r, T))
    axy, rxy = rng.uniform(0, 200, (n_m, 2)), rng.uniform(0, 200, (n_r, 2))
    idx = rng.integers(-1, n_r, (n_m, k)).astype(np.int32)
    idx[rng.random((n_m, k)) < 0.3] = -1
    ctx = _lib.default_context()
    dA, dR, dax, drx, didx = (ctx.to_device(np.ascontiguousarray(x)) for x in (A, R, axy, rxy, idx))
    for b, e in ((0, n_m), (17, n_m - 5), (n_m // 2, n_m // 2 + 1)):
        dout = ctx.alloc(max(e - b, 1) * k * 8)
        ctx.check(ctx.lib.same_padded_cost_f64_dev(ctx.handle, dA.ptr, dR.ptr, T, dax.ptr, drx.ptr, b, e, k, didx.ptr + b * k * 4, 0.75,
                                                   dout.ptr), "same_padded_cost_f64_dev")
        got = dout.download((e - b, k), np.float64)
        rr, cc = np.nonzero(idx[b:e] >= 0)
        want = np.full((e - b, k), np.inf)
        want[rr, cc] = oracle.pair_cost_arrays(A, R, axy, rxy, np.column_stack((rr + b, idx[b:e][rr, cc])), 0.75)
        assert np.array_equal(got, want), (T, k, b, e)


@pytest.mark.parametrize("T,k,n_m,n_r", [(1, 3, 200, 300), (2, 8, 513, 700), (5, 8, 1000, 900), (8, 32, 777, 1500), (20, 32, 600, 2000),
                                        (33, 5, 300, 400), (64, 16, 260, 500), (150, 4, 200, 300)])
def test_fp32_pair_and_padded_costs_vs_oracle(hip, oracle, T, k, n_m, n_r):
    """BASELINE config 5's fp32 costs: same_pair_cost_f32 and same_padded_cost_f32_dev (both kernel forms) are bit-equal to
    the oracle's float evaluation of src/same.py:1180-1189 and to the matching elements of the fp32 dense build, and stay
    within the forward bound (T + 4) * 2^-24 * (sum of operand magnitudes) of the fp64 costs."""
    from same_amd import _lib, ops

    rng = np.random.default_rng(T * 77 + k)
    A, R = rng.gamma(0.4, 20.0, (n_m, T)), rng.gamma(0.4, 20.0, (n_r, T))
    axy, rxy = rng.uniform(0, 200, (n_m, 2)), rng.uniform(0, 200, (n_r, 2))
    idx = rng.integers(0, n_r, (n_m, k)).astype(np.int32)
    idx[rng.random((n_m, k)) < 0.3] = -1
    rr, cc = np.nonzero(idx >= 0)
    pairs = np.column_stack((rr, idx[rr, cc]))
    want = oracle.pair_cost_arrays(A, R, axy, rxy, pairs, 0.75, dtype=np.float32)
    got = ops.pair_cost(A, R, axy, rxy, pairs, 0.75, dtype=np.float32)
    assert got.dtype == np.float32 and np.array_equal(got, want)
    D = ops.dense_cost(A, R, axy, rxy, 0.75, dtype=np.float32)
    assert np.array_equal(D[pairs[:, 0], pairs[:, 1]], got)
    c64 = oracle.pair_cost_arrays(A, R, axy, rxy, pairs, 0.75)
    pa, pr = pairs[:, 0], pairs[:, 1]
    mag = 0.75 * ((np.abs(A[pa]) + np.abs(R[pr])).sum(axis=1) + 0.001 * (np.abs(axy[pa]) + np.abs(rxy[pr])).sum(axis=1))
    assert (np.abs(got.astype(np.float64) - c64) <= (T + 4) * 2.0 ** -24 * mag).all()
    ctx = _lib.default_context()
    f32 = [np.ascontiguousarray(x, dtype=np.float32) for x in (A, R, axy, rxy)]
    dA, dR, dax, drx = (ctx.to_device(x) for x in f32)
    didx = ctx.to_device(idx)
    for b, e in ((0, n_m), (17, n_m - 5), (n_m // 2, n_m // 2 + 1)):
        dout = ctx.alloc(max(e - b, 1) * k * 4)
        ctx.check(ctx.lib.same_padded_cost_f32_dev(ctx.handle, dA.ptr, dR.ptr, T, dax.ptr, drx.ptr, b, e, k, didx.ptr + b * k * 4, 0.75,
                                                   dout.ptr), "same_padded_cost_f32_dev")
        gotp = dout.download((e - b, k), np.float32)
        r2, c2 = np.nonzero(idx[b:e] >= 0)
        wantp = np.full((e - b, k), np.inf, np.float32)
        wantp[r2, c2] = oracle.pair_cost_arrays(A, R, axy, rxy, np.column_stack((r2 + b, idx[b:e][r2, c2])), 0.75, dtype=np.float32)
        assert np.array_equal(gotp, wantp), (T, k, b, e)
    bad = np.array([[0, n_r]], np.int32)
    with pytest.raises(_lib.SameHipError):
        ops.pair_cost(A, R, axy, rxy, bad, 1.0, dtype=np.float32)


@pytest.mark.parametrize("T", [0, 1, 3, 8, 19, 20, 32])
@pytest.mark.parametrize("w", [1.0, 0.37])
def test_dense_cost_q32_opt_in_build(ops, oracle, T, w):
    """The opt-in fixed-point dense build (v_sad_u32 on a common 32-bit grid; NOT reference arithmetic, never a default):
    bit-equal to its oracle twin; within w * T * 2^-s of the reference-exact fp64 build; and -- the contract BASELINE.json
    states for fp64 costs -- within 1e-6 RELATIVE on every output, including near-identical cells whose type sum is far
    below the grid's resolution (those are recomputed in fp64 inside the kernel)."""
    from same_amd import synth

    n_m, n_r = 300, 1031   # ragged: not multiples of the tile
    r = synth.make_cells(n_r, max(T, 1), seed=40 + T)
    m = synth.make_cells(n_m, max(T, 1), seed=41 + T, side=r["side"])
    A, R = m["types"][:, :T].copy(), r["types"][:, :T].copy()
    if T:   # near-identical and identical cells: type sums of 0, 1e-9, 1e-4, 0.5 ... next to ordinary ones
        A[10] = R[5]
        A[11] = R[6] + 1e-9
        A[12] = R[7] * (1 + 1e-6)
        A[13] = R[8] + 0.5 / T
    grid = ops.quantize_types(A, R)
    got, bound = ops.dense_cost_q32(A, R, m["xy"], r["xy"], w, 7, 290, grid=grid)
    want = oracle.dense_cost_q32(A, R, m["xy"], r["xy"], w, grid[0], grid[1], 7, 290)
    assert np.array_equal(got, want)
    exact = oracle.dense_cost(A, R, m["xy"], r["xy"], w, 7, 290)
    assert bound == T * 2.0 ** -grid[1] and (T == 0 or grid[1] >= 22)            # rows on the 0-100 scale: 2^-24 steps
    assert np.max(np.abs(got - exact)) <= w * bound + 1e-12 * np.max(exact)
    assert np.max(np.abs(got - exact) / exact) <= 1e-6 * (1 + 1e-9)
    if T:
        assert got[10 - 7, 5] == exact[10 - 7, 5] and got[11 - 7, 6] == exact[11 - 7, 6]          # recomputed exactly
        # the pure grid result (rel_tol = 0) keeps the absolute bound only
        raw, _ = ops.dense_cost_q32(A, R, m["xy"], r["xy"], w, 7, 290, grid=grid, rel_tol=0.0)
        assert np.array_equal(raw, oracle.dense_cost_q32(A, R, m["xy"], r["xy"], w, grid[0], grid[1], 7, 290, rel_tol=0.0))
        assert np.max(np.abs(raw - exact)) <= w * bound + 1e-12 * np.max(exact)
    # values far off the probability scale still get a grid that cannot overflow 32 bits
    big = ops.quantize_types(A * 1e6 - 5e5, R * 1e6 - 5e5)
    got2, bound2 = ops.dense_cost_q32(A * 1e6 - 5e5, R * 1e6 - 5e5, m["xy"], r["xy"], w, 0, 50, grid=big)
    exact2 = oracle.dense_cost(A * 1e6 - 5e5, R * 1e6 - 5e5, m["xy"], r["xy"], w, 0, 50)
    assert np.max(np.abs(got2 - exact2) / np.abs(exact2)) <= 1e-6 * (1 + 1e-9)


def test_dense_cost_q32_limits(ops):
    from same_amd import _lib

    ctx = _lib.default_context()
    z = ctx.alloc(4096)
    L, H = ctx.lib, ctx.handle
    # T > SAME_Q32_MAX_TYPES
    assert L.same_dense_cost_q32_dev(H, z.ptr, z.ptr, z.ptr, z.ptr, 33, z.ptr, z.ptr, 4, 0, 1, 1.0, 1.0, 1e-6, z.ptr, 4) == -22
    # odd pitch
    assert L.same_dense_cost_q32_dev(H, z.ptr, z.ptr, z.ptr, z.ptr, 4, z.ptr, z.ptr, 5, 0, 1, 1.0, 1.0, 1e-6, z.ptr, 5) == -22
    # inv_scale must be > 0
    assert L.same_dense_cost_q32_dev(H, z.ptr, z.ptr, z.ptr, z.ptr, 4, z.ptr, z.ptr, 4, 0, 1, 1.0, 0.0, 1e-6, z.ptr, 4) == -22
    # a tolerance needs the fp64 matrices
    assert L.same_dense_cost_q32_dev(H, z.ptr, z.ptr, None, None, 4, z.ptr, z.ptr, 4, 0, 1, 1.0, 1.0, 1e-6, z.ptr, 4) == -22
    with pytest.raises(ValueError):
        ops.quantize_types(np.array([[np.nan, 1.0]]), np.ones((2, 2)))


def test_dense_cost_f32_dev_entry_point(hip, oracle):
    """same_dense_cost_f32_dev (resident operands, cfg-5 fp32 variant): equals the host-buffer form and the oracle's fp32 build."""
    from same_amd import _lib, ops

    rng = np.random.default_rng(8)
    n_m, n_r, T = 700, 1000, 12
    A, R = rng.gamma(0.4, 20.0, (n_m, T)).astype(np.float32), rng.gamma(0.4, 20.0, (n_r, T)).astype(np.float32)
    axy, rxy = rng.uniform(0, 200, (n_m, 2)).astype(np.float32), rng.uniform(0, 200, (n_r, 2)).astype(np.float32)
    ctx = _lib.default_context()
    dA, dR, dax, drx = (ctx.to_device(x) for x in (A, R, axy, rxy))
    b, e = 33, 650
    out = ctx.alloc((e - b) * n_r * 4)
    ctx.check(ctx.lib.same_dense_cost_f32_dev(ctx.handle, dA.ptr, dR.ptr, T, dax.ptr, drx.ptr, n_r, b, e, 0.5, out.ptr, n_r),
              "same_dense_cost_f32_dev")
    got = out.download((e - b, n_r), np.float32)
    assert np.array_equal(got, ops.dense_cost(A, R, axy, rxy, 0.5, b, e, dtype=np.float32))
    assert np.array_equal(got, oracle.dense_cost(A, R, axy, rxy, 0.5, b, e, dtype=np.float32))


def test_integration_md_stub_runs_as_written(oracle):
    """The ctypes stub INTEGRATION.md gives a reference maintainer is executed verbatim (only the library path is made absolute)
    and its two wrappers are checked against the oracle."""
    import os
    import re
    from same_amd import _lib

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md"), encoding="utf-8").read()
    block = re.search(r"```python\n(# src/_samehip\.py.*?)```", text, re.S).group(1)
    assert 'ctypes.CDLL("libsame_hip.so")' in block
    ns = {}
    exec(compile(block.replace('ctypes.CDLL("libsame_hip.so")', f'ctypes.CDLL({_lib.LIB_PATH!r})'), "INTEGRATION.md", "exec"), ns)
    rng = np.random.default_rng(4)
    axy, rxy = rng.uniform(0, 100, (300, 2)), rng.uniform(0, 100, (400, 2))
    A, R = rng.gamma(0.5, 20.0, (300, 6)), rng.gamma(0.5, 20.0, (400, 6))
    idx, cnt = ns["knn_prune"](axy, rxy, 9.0, 5)
    oidx, _, ocnt = oracle.knn_prune(axy, rxy, 9.0, 5)
    assert np.array_equal(idx, oidx) and np.array_equal(cnt, ocnt)
    rr, cc = np.nonzero(idx >= 0)
    pairs = np.column_stack((rr, idx[rr, cc]))
    assert np.array_equal(ns["pair_cost"](A, R, axy, rxy, pairs, 1.5), oracle.pair_cost_arrays(A, R, axy, rxy, pairs, 1.5))


def test_sweep_called_from_other_threads(oracle):
    """The lazy callback is re-entered from solver threads (src/same.py:1241): sweeps issued concurrently from several Python
    threads on the shared default context return what the calling thread asked for.  Two sweep objects of DIFFERENT shapes
    (two models alive at once, as with overlapping windows) are driven together: each owns its device state
    (same_sweep handle) and every call gets its own output buffer, so neither can see the other's arrays."""
    import threading
    from conftest import load_golden
    from same_amd import _lib, sweeps

    models = []
    for case in ("cfg2_small", "cfg1_500"):
        g = load_golden(case)
        pairs, tris, sign = g["pairs"], g["tri_plain"], g["source_signs"].astype(np.int8)
        n_a = int(pairs[:, 0].max()) + 1
        rxy = g["in_ref_xy"][g["kept_ref"]]
        models.append((sweeps.LazyOrientationSweep(pairs, tris, sign, rxy, n_a), pairs, tris, sign, rxy, n_a))
    assert models[0][0].bound.Tr != models[1][0].bound.Tr and models[0][0].bound.P != models[1][0].bound.P
    rng = np.random.default_rng(0)
    jobs = []   # (model index, x)
    for q, p in enumerate((0.05, 0.2, 0.5, 0.9, 0.0, 1.0, 0.3, 0.7)):
        m = q % 2
        jobs.append((m, (rng.random(len(models[m][1])) < p).astype(float)))
    want = [oracle.lazy_orientation_sweep(x, *models[m][1:]) for m, x in jobs]
    got = [None] * len(jobs)
    errors = []

    def work(q):
        try:
            m, x = jobs[q]
            for _ in range(20):
                checked, viol, _ = models[m][0].sweep(x)
                got[q] = (checked, [(t, int(a), int(b), int(c)) for t, a, b, c in viol])
        except Exception as e:   # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=(q,)) for q in range(len(jobs))]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors, errors
    for q in range(len(jobs)):
        assert got[q][0] == want[q][0] and got[q][1] == [tuple(int(v) for v in row) for row in want[q][1]], q
    # a vector of the other model's length is refused, not read
    with pytest.raises(_lib.SameHipError):
        models[0][0].bound.sweep_x(jobs[1][1])


@pytest.mark.gpu
def test_orientation_sweep_repeated_and_with_a_long_flipped_list(oracle):
    """The per-incumbent sweep called again and again on one handle (as the solver does), and a handle whose flipped list is
    longer than the 4096-entry head that comes back with the counters in one copy: same answers every time, equal to the oracle."""
    from conftest import load_golden
    from scipy.spatial import Delaunay
    from same_amd import ops, sweeps, synth

    g = load_golden("cfg2_small")
    pairs, tris, sign = g["pairs"], g["tri_plain"], g["source_signs"].astype(np.int8)
    n_a = int(pairs[:, 0].max()) + 1
    rxy = g["in_ref_xy"][g["kept_ref"]]
    rng = np.random.default_rng(5)
    xs = [(rng.random(len(pairs)) < p).astype(float) for p in (0.1, 0.5, 1.0, 0.0, 0.8)]
    sw = sweeps.LazyOrientationSweep(pairs, tris, sign, rxy, n_a)
    for rep in range(3):
        for q, x in enumerate(xs):
            checked, viol, _ = sw.sweep(x)
            want = oracle.lazy_orientation_sweep(x, pairs, tris, sign, rxy, n_a)
            as_tuples = lambda rows: [tuple(int(v) for v in row) for row in rows]
            assert int(checked) == want[0] and as_tuples(viol) == as_tuples(want[1]), (rep, q)
    ref = synth.make_cells(30000, 2, seed=11)
    mov = synth.make_jittered(ref, seed=12)
    big_tris = np.ascontiguousarray(Delaunay(mov["xy"]).simplices, dtype=np.int32)
    big_match = rng.integers(0, len(ref["xy"]), len(mov["xy"])).astype(np.int32)
    bsign, _ = ops.tri_sign_weight(mov["xy"], None, big_tris)
    big = ops.BoundSweep(big_tris, bsign, ref["xy"], len(mov["xy"]))
    pc, pv = big.sweep_match(big_match)
    pc2, pv2 = big.sweep_match(big_match)
    assert len(pv) > 4096 and pc == pc2 and np.array_equal(pv, pv2)
    wc, wv, _ = oracle.orient_sweep(big_tris, np.asarray(bsign).astype(np.int8), ref["xy"], big_match)
    assert pc == wc and np.array_equal(pv, np.asarray(wv, dtype=np.int32))


@pytest.mark.gpu
@pytest.mark.parametrize("map_mode", [0, 2, 4])
def test_dense_cost_is_bit_exact_under_every_block_map(oracle, map_mode, tmp_path):
    """The dense kernel ships three block -> (column tile, row chunk) maps and picks one by shape (2 or 4 for 16-byte stores, 0
    for the scalar-store fallback); SAME_DENSE_MAP forces one (a test hook).  Whatever the map, every output is computed by the
    same instruction sequence, so the matrix must not change: a child process per map builds ragged fp64 and fp32 tiles at
    several T, the parent compares with the oracle."""
    import os
    import subprocess
    import sys
    from conftest import ROOT

    rng = np.random.default_rng(100 + map_mode)
    n_m, n_r = 1500, 2311
    cases = {}
    for T in (0, 5, 16, 20):
        A = rng.dirichlet(np.full(max(T, 1), 0.3), size=n_m)[:, :T] * 100 if T else np.zeros((n_m, 0))
        R = rng.dirichlet(np.full(max(T, 1), 0.3), size=n_r)[:, :T] * 100 if T else np.zeros((n_r, 0))
        cases[T] = (A, R, rng.uniform(0, 300, (n_m, 2)), rng.uniform(0, 300, (n_r, 2)))
    np.savez(tmp_path / "in.npz", **{f"{k}_{T}": v for T, c in cases.items() for k, v in zip("ARxy", c)})
    code = f"""
import sys, numpy as np
sys.path.insert(0, {str(ROOT)!r})
from same_amd import ops
d = np.load({str(tmp_path / 'in.npz')!r})
out = {{}}
for T in (0, 5, 16, 20):
    a = [d[f"{{k}}_{{T}}"] for k in "ARxy"]
    out[f"f64_{{T}}"] = ops.dense_cost(*a, 1.0, 7, 1400)
    out[f"f32_{{T}}"] = ops.dense_cost(*a, 2.5, 7, 1400, dtype=np.float32)
np.savez({str(tmp_path / 'out.npz')!r}, **out)
"""
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SAME_DENSE_MAP=str(map_mode)), capture_output=True, text=True,
                       timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    got = np.load(tmp_path / "out.npz")
    for T, a in cases.items():
        assert np.array_equal(got[f"f64_{T}"], oracle.dense_cost(*a, 1.0, 7, 1400)), T
        assert np.array_equal(got[f"f32_{T}"], oracle.dense_cost(*a, 2.5, 7, 1400, dtype=np.float32)), T
