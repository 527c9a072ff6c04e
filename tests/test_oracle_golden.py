"""Pin the CPU oracle against outputs of the reference itself (tests/golden/, made by
tools/gen_golden.py from the imported reference).  CPU only."""
import numpy as np
import pandas as pd
import pytest

from conftest import FULL_CASES, frames_from_golden, load_golden


def _tie_free_rows(axy, rxy, pairs):
    """Rows whose candidate distances are all distinct (reference order is unspecified on ties)."""
    ok = {}
    for i in np.unique(pairs[:, 0]):
        js = pairs[pairs[:, 0] == i, 1]
        d = np.sqrt(((rxy[js] - axy[i]) ** 2).sum(1))
        ok[i] = len(np.unique(d)) == len(d)
    return ok


@pytest.mark.parametrize("case", FULL_CASES)
def test_knn_pairs_and_compaction(oracle, case):
    g = load_golden(case)
    a_df, r_df, _ = frames_from_golden(g)
    radius, knn = g["params"][0], int(g["params"][1])
    na, nr, pairs = oracle.find_knn_within_radius(a_df, r_df, radius, knn=knn)
    assert np.array_equal(na["__row"].to_numpy(), g["kept_aligned"])
    assert np.array_equal(nr["__row"].to_numpy(), g["kept_ref"])
    assert np.array_equal(np.asarray(pairs, dtype=np.int64), g["pairs"])  # bit-exact, order included


def test_knn_edges_sparse_and_ties(oracle):
    g = load_golden("adversarial")
    # radius is inclusive (<=); farther points excluded; unused refs compacted away
    a = pd.DataFrame({"X": g["edge_axy"][:, 0], "Y": g["edge_axy"][:, 1]})
    r = pd.DataFrame({"X": g["edge_rxy"][:, 0], "Y": g["edge_rxy"][:, 1], "__row": np.arange(len(g["edge_rxy"]))})
    _, nr, pairs = oracle.find_knn_within_radius(a, r, 5.0, knn=4)
    assert np.array_equal(np.asarray(pairs), g["edge_pairs"])
    assert np.array_equal(nr["__row"].to_numpy(), g["edge_kept_ref"])
    # sparse: rows without neighbours are dropped and pairs re-indexed
    a = pd.DataFrame({"X": g["sparse_axy"][:, 0], "Y": g["sparse_axy"][:, 1], "__row": np.arange(len(g["sparse_axy"]))})
    r = pd.DataFrame({"X": g["sparse_rxy"][:, 0], "Y": g["sparse_rxy"][:, 1], "__row": np.arange(len(g["sparse_rxy"]))})
    na, nr, pairs = oracle.find_knn_within_radius(a, r, 4.0, knn=3)
    assert len(na) < len(a) and len(nr) < len(r)
    assert np.array_equal(na["__row"].to_numpy(), g["sparse_kept_aligned"])
    assert np.array_equal(nr["__row"].to_numpy(), g["sparse_kept_ref"])
    assert np.array_equal(np.asarray(pairs), g["sparse_pairs"])
    # grid with exact ties: same candidate multiset of distances per row; identical where tie-free
    a = pd.DataFrame({"X": g["grid_axy"][:, 0], "Y": g["grid_axy"][:, 1]})
    r = pd.DataFrame({"X": g["grid_rxy"][:, 0], "Y": g["grid_rxy"][:, 1]})
    _, _, pairs = oracle.find_knn_within_radius(a, r, 1.5, knn=6)
    pairs = np.asarray(pairs)
    gp = g["grid_pairs"]
    assert len(pairs) == len(gp) and np.array_equal(pairs[:, 0], gp[:, 0])
    d_mine = np.sqrt(((g["grid_rxy"][pairs[:, 1]] - g["grid_axy"][pairs[:, 0]]) ** 2).sum(1))
    d_ref = np.sqrt(((g["grid_rxy"][gp[:, 1]] - g["grid_axy"][gp[:, 0]]) ** 2).sum(1))
    assert np.array_equal(d_mine, d_ref)  # same distance sequence; tie members may permute
    # documented tie rule: within equal distance, ascending ref index
    for i in np.unique(pairs[:, 0]):
        sel = pairs[:, 0] == i
        key = list(zip(d_mine[sel], pairs[sel, 1]))
        assert key == sorted(key)


@pytest.mark.parametrize("case", FULL_CASES)
def test_knn_priority(oracle, case):
    g = load_golden(case)
    a_df, r_df, _ = frames_from_golden(g)
    _, _, prio = oracle.find_knn_with_cell_type_priority(a_df, r_df, g["params"][0], knn=int(g["params"][1]))
    assert np.array_equal(np.asarray(prio, dtype=np.int64).reshape(-1, 2), g["pairs_priority"])


@pytest.mark.parametrize("case", FULL_CASES)
def test_pair_costs_bit_exact(oracle, case):
    g = load_golden(case)
    a_df, r_df, cols = frames_from_golden(g)
    na, nr = a_df.iloc[g["kept_aligned"]].reset_index(drop=True), r_df.iloc[g["kept_ref"]].reset_index(drop=True)
    sel = g["cost_sel"]
    c = np.array(oracle.pair_costs(na, nr, g["pairs"][sel], cols, g["params"][3]))
    assert np.array_equal(c, g["costs"])
    c2 = np.array(oracle.pair_costs(na, nr, g["pairs"][sel[:200]], cols, 2.5))
    assert np.array_equal(c2, g["costs_w2p5"])
    # the dense builder is the same expression on every (i, j)
    A, R = na[cols].to_numpy(), nr[cols].to_numpy()
    D = oracle.dense_cost(A, R, na[["X", "Y"]].to_numpy(), nr[["X", "Y"]].to_numpy(), float(g["params"][3]))
    p = g["pairs"][sel]
    assert np.array_equal(D[p[:, 0], p[:, 1]], g["costs"])


def _compacted(g):
    a_df, r_df, cols = frames_from_golden(g)
    return (a_df.iloc[g["kept_aligned"]].reset_index(drop=True), r_df.iloc[g["kept_ref"]].reset_index(drop=True), cols)


@pytest.mark.parametrize("case", FULL_CASES)
def test_triangle_filter(oracle, case):
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
        kept, unc = oracle.filter_triangles_by_radius(pts, g["delaunay"], rad, aligned_df=na,
                                                      remove_unconstrained_nodes=True, **kw)
        assert np.array_equal(np.array(kept, dtype=np.int64).reshape(-1, 3), g[f"tri_{tag}"]), tag
        assert sorted(unc) == g[f"unc_{tag}"].tolist(), tag


def test_triangle_filter_adversarial(oracle):
    g = load_golden("adversarial")
    tdf = pd.DataFrame({"X": g["adv_pts"][:, 0], "Y": g["adv_pts"][:, 1], "cell_type": g["adv_type"].astype(object)})
    for tag, kw in (("45", dict(min_angle_deg=45, ignore_same_type_triangles=False)),
                    ("45t", dict(min_angle_deg=45, ignore_same_type_triangles=True)),
                    ("none", dict(min_angle_deg=None, ignore_same_type_triangles=True)),
                    ("0", dict(min_angle_deg=0, ignore_same_type_triangles=False)),
                    ("15", dict(min_angle_deg=15, ignore_same_type_triangles=True))):
        for rad in (10.0, 3.0, 1.0):
            kept, unc = oracle.filter_triangles_by_radius(g["adv_pts"], g["adv_tris"], rad, aligned_df=tdf,
                                                          remove_unconstrained_nodes=True, **kw)
            assert np.array_equal(np.array(kept, dtype=np.int64).reshape(-1, 3), g[f"adv_kept_{tag}_{rad}"]), (tag, rad)
            assert sorted(unc) == g[f"adv_unc_{tag}_{rad}"].tolist(), (tag, rad)
    kept = oracle.filter_triangles_by_radius(g["adv_pts"], np.zeros((0, 3), dtype=int), 10.0, aligned_df=tdf,
                                             ignore_same_type_triangles=True)
    assert len(kept) == 0 and g["adv_empty_kept"].shape == (0, 3)
    sign, _ = oracle.tri_sign_weight(g["adv_pts"], np.ones(len(g["adv_pts"])), g["adv_tris"])
    assert np.array_equal(sign.astype(np.float64), g["adv_signs"])


@pytest.mark.parametrize("case", FULL_CASES)
def test_weights_signs_info(oracle, case):
    g = load_golden(case)
    na, _, _ = _compacted(g)
    tris = g["tri_plain"]
    assert np.array_equal(np.array(oracle.triangle_weights(na, tris)), g["tri_weights"])
    assert np.array_equal(np.array(oracle.source_signs(na, tris)), g["source_signs"])
    info = oracle.precompute_triangle_info(na, tris, oracle.simplex_map(len(na), tris))
    keys = list(info.keys())
    assert keys == g["tinfo_keys"].tolist()  # dict insertion order is part of the contract
    b = np.array([[info[k]["bounds"][q] for q in ("min_x", "max_x", "min_y", "max_y")] for k in keys])
    e = np.array([[info[k][q] for q in ("max_x_vertex", "min_x_vertex", "max_y_vertex", "min_y_vertex")] for k in keys])
    assert np.array_equal(b, g["tinfo_bounds"]) and np.array_equal(e, g["tinfo_extreme"])


@pytest.mark.parametrize("case", FULL_CASES)
def test_mip_start(oracle, case):
    g = load_golden(case)
    na, nr, _ = _compacted(g)
    vp = [tuple(p) for p in g["pairs"].tolist()]
    kw = dict(valid_pairs=vp, costs=list(g["all_costs"]), n_aligned=len(na), n_ref=len(nr),
              aligned_sizes=na["size"].to_numpy(dtype=float), max_matches=1, verbose=False)
    ch, un = oracle.compute_mip_start_pairs(no_match_penalty=g["params"][4], init_method="greedy", **kw)
    assert np.array_equal(np.array(ch, dtype=np.int64).reshape(-1, 3), g["greedy_chosen"])
    assert sorted(un) == g["greedy_unmatched"].tolist()
    ch, un = oracle.compute_mip_start_pairs(no_match_penalty=float(g["greedy_lo_penalty"][0]), init_method="greedy", **kw)
    assert np.array_equal(np.array(ch, dtype=np.int64).reshape(-1, 3), g["greedy_lo_chosen"])
    assert sorted(un) == g["greedy_lo_unmatched"].tolist()
    ch, un = oracle.compute_mip_start_pairs(no_match_penalty=g["params"][4], init_method="hungarian",
                                            init_hungarian_max_n=100000, **kw)
    assert np.array_equal(np.array(ch, dtype=np.int64).reshape(-1, 3), g["hungarian_chosen"])
    assert sorted(un) == g["hungarian_unmatched"].tolist()
    with pytest.raises(ValueError):
        oracle.compute_mip_start_pairs(no_match_penalty=1.0, init_method="bogus", **kw)


@pytest.mark.parametrize("case", FULL_CASES)
def test_sweeps(oracle, case):
    g = load_golden(case)
    na, nr, _ = _compacted(g)
    tris = g["tri_plain"]
    vp = [tuple(p) for p in g["pairs"].tolist()]
    rxy = nr[["X", "Y"]].to_numpy()
    # a10
    checked, viol = oracle.lazy_orientation_sweep(g["x_vals"], vp, tris, g["source_signs"], rxy, len(na))
    assert checked == int(g["lazy_checked"][0])
    assert [v[0] for v in viol] == g["lazy_violating"].tolist()
    assert all(tuple(tris[v[0]]) == v[1:] for v in viol)
    # a11
    m_df = pd.DataFrame({"aligned_idx": g["greedy_chosen"][:, 0], "ref_idx": g["greedy_chosen"][:, 1]})
    info = oracle.precompute_triangle_info(na, tris, oracle.simplex_map(len(na), tris))
    v = oracle.verify_spatial_preservation(na, nr, m_df, info)
    _check_violations(v, g, "")
    # a12
    match = np.full(len(na), -1, np.int32)
    match[g["greedy_chosen"][:, 0]] = g["greedy_chosen"][:, 1]
    before, after, m3, fl = oracle.area_flip(na[["X", "Y"]].to_numpy(), rxy, tris, match)
    assert np.array_equal(before, g["area_before"])
    assert np.array_equal(after, g["area_after"], equal_nan=True)
    assert np.array_equal(np.nonzero(fl)[0], g["area_flipped"])
    assert np.array_equal(m3, g["area_matched3"])
    # a14
    combos = g["eager_combos"]
    tr = np.arange(3 * len(combos), dtype=np.int32).reshape(-1, 3)  # triangle t -> candidate rows 3t..3t+2
    cand = combos.reshape(-1, 1).astype(np.int32)
    s = oracle.eager_signs(rxy, tr, cand)
    assert np.array_equal(s.reshape(-1), g["eager_signs"])


def _check_violations(v, g, prefix):
    def rows(lst):
        return np.array([(d["triangle_idx"], d["point1"]["aligned_idx"], d["point2"]["aligned_idx"],
                          d["point1"]["ref_idx"], d["point2"]["ref_idx"]) for d in lst], dtype=np.int64).reshape(-1, 5)
    s = v["violation_summary"]
    assert [s["total_triangles"], s["violated_triangles"], s["total_comparisons"], s["total_violations"]] == \
        g[f"{prefix}viol_summary"].tolist()
    assert np.array_equal(rows(v["x_order_violations"]), g[f"{prefix}viol_x"])  # traversal order included
    assert np.array_equal(rows(v["y_order_violations"]), g[f"{prefix}viol_y"])
    assert sorted(int(t) for t in v["triangles_with_violations"]) == g[f"{prefix}viol_tris"].tolist()
    assert sorted(int(p) for p in v["points_with_violations"]) == g[f"{prefix}viol_points"].tolist()
    assert np.array_equal(np.array([s["percent_triangles_violated"], s["percent_violations"]]), g[f"{prefix}viol_percent"])


@pytest.mark.parametrize("case", ["simulated_st", "simulated_elastic"])
def test_stored_reference_runs(oracle, case):
    """Known answers stored by the reference authors' own run (examples/simulated_*/var_out.npy)."""
    g = load_golden(case)
    a = pd.DataFrame({"X": g["aligned_xy"][:, 0], "Y": g["aligned_xy"][:, 1]})
    r = pd.DataFrame({"X": g["ref_xy"][:, 0], "Y": g["ref_xy"][:, 1]})
    m = pd.DataFrame({"aligned_idx": g["matches"][:, 0], "ref_idx": g["matches"][:, 1]})
    info = {int(k): {"vertices": vtx} for k, vtx in zip(g["tinfo_keys"], g["tinfo_vertices"])}
    v = oracle.verify_spatial_preservation(a, r, m, info)
    _check_violations(v, g, "stored_")
    _check_violations(v, g, "")


def test_sweeps_adversarial(oracle):
    g = load_golden("adversarial")
    pts, tris, rxy = g["adv_pts"], g["adv_tris"], g["sw_rxy"]
    a = pd.DataFrame({"X": pts[:, 0], "Y": pts[:, 1]})
    r = pd.DataFrame({"X": rxy[:, 0], "Y": rxy[:, 1]})
    m = pd.DataFrame({"aligned_idx": g["sw_matches"][:, 0], "ref_idx": g["sw_matches"][:, 1]})
    info = oracle.precompute_triangle_info(a, tris, oracle.simplex_map(len(a), tris))
    assert list(info.keys()) == g["sw_tinfo_keys"].tolist()
    v = oracle.verify_spatial_preservation(a, r, m, info)
    _check_violations(v, g, "sw_")
    match = np.full(len(pts), -1, np.int32)
    for ai, ri in g["sw_matches"]:
        match[ai] = ri
    before, after, m3, fl = oracle.area_flip(pts, rxy, tris, match)
    assert np.array_equal(before, g["sw_area_before"]) and np.array_equal(after, g["sw_area_after"], equal_nan=True)
    assert np.array_equal(np.nonzero(fl)[0], g["sw_area_flipped"]) and np.array_equal(m3, g["sw_area_matched3"])
    checked, viol = oracle.lazy_orientation_sweep(g["sw_x"], [tuple(p) for p in g["sw_pairs"].tolist()], tris,
                                                  g["adv_signs"], rxy, len(pts))
    assert checked == int(g["sw_lazy_checked"][0]) and [t[0] for t in viol] == g["sw_lazy_violating"].tolist()


EVAL_KEYS = ('total_triangles', 'triangles_with_all_matched', 'triangles_processed', 'triangles_same_type_skipped',
             'triangles_flipped', 'percent_flipped', 'nodes_in_violating_triangles', 'percent_nodes_violating')
EVAL_CASES = (('default', {}), ('alltypes', dict(ignore_same_type_triangles=False)), ('local', dict(node_local=True)),
              ('local_strict', dict(node_local=True, majority_threshold=0.3, min_flips=2)),
              ('local_alltypes', dict(node_local=True, ignore_same_type_triangles=False, majority_threshold=0.75)))


def eval_inputs(g):
    mdf = pd.DataFrame({'X': g['mxy'][:, 0], 'Y': g['mxy'][:, 1]}, index=g['ids'])
    out_df = pd.DataFrame({'aligned_metacell_index': g['o_id'], 'matched_ref_index': g['o_ref'], 'mapped_x': g['o_mx'],
                           'mapped_y': g['o_my'], 'cell_type': g['o_type'].astype(object)})

    class MC:
        metacell_df = mdf
        metacell_delaunay = g['tri_ids']

    return out_df, MC()


def test_check_triangle_violations(oracle):
    """eval_utils.check_triangle_violations, incl. duplicate ids, a NaN coordinate, an unknown id."""
    g = load_golden('eval_tri')
    out_df, mc = eval_inputs(g)
    for tag, kw in EVAL_CASES:
        df, stats = oracle.check_triangle_violations(out_df, mc, **kw)
        assert np.array_equal(df['in_violating_triangle'].to_numpy().astype(np.uint8), g[f'viol_{tag}']), tag
        assert np.array_equal(np.array([stats[k] for k in EVAL_KEYS], dtype=np.float64), g[f'stats_{tag}']), tag


def metacell_inputs(which):
    """Inputs of the metacell fixture: the shipped synthetic query frame is rebuilt from the synthetic_example fixture."""
    from same_amd import synth
    if which == 'seeded':
        df = synth.to_frame(synth.make_cells(1500, 4, seed=5))
        df['batch'] = np.where(np.arange(len(df)) % 3 == 0, 'b0', 'b1')
        df['flag'] = (np.arange(len(df)) % 2).astype(np.int64)
        return df, dict(max_metacell_size=5, r_max=30, min_angle_deg=12), ['c1', 'c2', 'c3', 'c4', 'size', 'flag'], ['batch']
    g = load_golden('metacell_inputs')
    q = pd.DataFrame({'X': g['q_xy'][:, 0], 'Y': g['q_xy'][:, 1], 'cell_type': g['q_type'].astype(object), 'c1': g['q_c'][:, 0],
                      'c2': g['q_c'][:, 1], 'c3': g['q_c'][:, 2], 'quadrant': g['q_quadrant'].astype(object), 'cell_idx': g['q_idx']})
    kw = {'s3': dict(max_metacell_size=3, r_max=5, min_angle_deg=5), 's6': dict(max_metacell_size=6, r_max=None, min_angle_deg=10),
          's9': dict(max_metacell_size=9, r_max=4, min_angle_deg=None), 's1': dict(max_metacell_size=1, r_max=5, min_angle_deg=5)}[which]
    return q, dict(original_idx_col='cell_idx', **kw), ['c1', 'c2', 'c3'], ['quadrant']


def check_metacells(g, prefix, mdf, tri, orig_tri, num_cols, other_cols):
    assert list(mdf.columns) == g[f'{prefix}_cols'].tolist()
    assert np.array_equal(mdf[['X', 'Y']].to_numpy(dtype=np.float64), g[f'{prefix}_xy'])      # centroids bit-exact
    assert np.array_equal(mdf['size'].to_numpy(dtype=np.int64), g[f'{prefix}_size'])
    assert np.array_equal(mdf['cell_type'].to_numpy().astype(str), g[f'{prefix}_type'])
    members = mdf['members'].tolist()
    assert np.array_equal(np.array([m for ms in members for m in ms], dtype=np.int64), g[f'{prefix}_members'])
    assert np.array_equal(np.cumsum([0] + [len(ms) for ms in members]), g[f'{prefix}_member_off'])
    assert np.array_equal(mdf[num_cols].to_numpy(dtype=np.float64), g[f'{prefix}_num'])
    assert np.array_equal(mdf[other_cols].to_numpy().astype(str), g[f'{prefix}_other'])
    assert np.array_equal(mdf['metacell_id'].to_numpy(dtype=np.int64), g[f'{prefix}_mcid'])
    assert np.array_equal(np.asarray(tri, dtype=np.int64).reshape(-1, 3), g[f'{prefix}_tri'])
    if orig_tri is not None:
        assert np.array_equal(np.asarray(orig_tri, dtype=np.int64).reshape(-1, 3), g[f'{prefix}_orig_tri'])


@pytest.mark.parametrize("which", ['s3', 's6', 's9', 's1', 'seeded'])
def test_greedy_triangle_collapse(oracle, which):
    """metacell_utils.greedy_triangle_collapse."""
    g = load_golden('metacell')
    df, kw, num_cols, other_cols = metacell_inputs(which)
    mdf, tri, orig = oracle.greedy_triangle_collapse(df, **kw)
    check_metacells(g, 'seeded' if which == 'seeded' else f'q_{which}', mdf, tri, None if which == 'seeded' else orig, num_cols, other_cols)


def test_oracle_reproduces_reference_run_same_model(oracle):
    """CPU cross-check of the oracle against the reference's own run_same (tests/golden/run_same_mock.npz, 'lazy_greedy'):
    pairs, triangles, objective coefficients (pair costs, triangle weights), greedy MIP start and the callback's cuts that the
    reference handed to the recording solver double are what the oracle computes from the same frames."""
    from scipy.spatial import Delaunay
    from same_amd import synth

    g = load_golden("run_same_mock")
    P = "lazy_greedy/"
    cells = synth.make_cells(300, 4, seed=41)
    r_df = synth.to_frame(cells)
    a_df = synth.to_frame(synth.make_jittered(cells, seed=42))
    cols = synth.type_columns(4)
    na, nr, pairs = oracle.find_knn_within_radius(a_df, r_df, 20, 4)
    names = [str(n) for n in g[P + "var_names"]]
    n_x = sum(n.startswith("x[") for n in names)
    assert n_x == len(pairs) and sum(n.startswith("penalty[") for n in names) == len(nr) and sum(n.startswith("no_match[") for n in names) == len(na)
    costs = np.array(oracle.pair_costs(na, nr, pairs, cols, 1))
    assert np.array_equal(g[P + "objective"][:n_x], costs)                                  # c[idx] * x[idx]
    xy = na[["X", "Y"]].values
    kept = oracle.filter_triangles_by_radius(xy, Delaunay(xy).simplices, 20, aligned_df=na, ignore_same_type_triangles=True, min_angle_deg=15)
    assert np.array_equal(np.asarray(kept, dtype=np.int64).reshape(-1, 3), g[P + "triangles"])
    sign, weight = oracle.tri_sign_weight(xy, na["size"].to_numpy(dtype=float), kept)
    q0 = names.index("q_tri[0]")
    from same_amd.params import init_optim_params
    dflt = init_optim_params()
    assert np.array_equal(g[P + "objective"][q0:q0 + len(kept)], dflt["delaunay_penalty"] * weight)
    chosen, unmatched = oracle.compute_mip_start_pairs(valid_pairs=pairs, costs=list(costs), n_aligned=len(na), n_ref=len(nr),
                                                       aligned_sizes=na["size"].to_numpy(dtype=float),
                                                       no_match_penalty=dflt["no_match_penalty"], max_matches=1, init_method="greedy",
                                                       verbose=False)
    start = np.zeros(n_x)
    start[[q for _, _, q in chosen]] = 1.0
    assert np.array_equal(g[P + "var_start"][:n_x], start)
    checked, viol = oracle.lazy_orientation_sweep(start, pairs, kept, sign, nr[["X", "Y"]].to_numpy(dtype=float), len(na))
    cut_q = [names[v] for row, coef in zip(g[P + "cut_term_var"], g[P + "cut_term_coef"]) for v, c in zip(row, coef) if v >= 0 and c < 0]
    assert cut_q == [f"q_tri[{t}]" for t, _, _, _ in viol[:25]] and int(g[P + "lazy"][1]) == len(cut_q)
