"""SURVEY 8(f3, f4): host-side rows (window merge, metacell unpacking) against reference-generated fixtures.
CPU part: the list bookkeeping touches no kernel (the metacell objects it consumes are rebuilt with the oracle); the
batched assignments of strategy='nearest' are checked here for the ORACLE (against scipy, which is what the reference
calls, and against the reference-generated fixture) and on the GPU for the product (test_unpack_nearest_on_device)."""
import numpy as np
import pandas as pd
import pytest

from conftest import load_golden


def _metacells(oracle):
    from same_amd import synth
    cells = synth.make_cells(900, 3, seed=21)
    a_df = synth.to_frame(cells); a_df["Cell_Num_Old"] = np.arange(len(a_df)) * 2 + 5
    r_df = synth.to_frame(synth.make_jittered(cells, seed=22)); r_df["Cell_Num_Old"] = np.arange(len(r_df)) * 3 + 1
    ma, _, _ = oracle.greedy_triangle_collapse(a_df, max_metacell_size=5, r_max=40, min_angle_deg=10)
    mr, _, _ = oracle.greedy_triangle_collapse(r_df, max_metacell_size=4, r_max=40, min_angle_deg=10)
    return a_df, r_df, ma, mr


def test_unpack_metacell_matches(oracle):
    from same_amd.metacell_utils import unpack_metacell_matches

    g = load_golden("unpack_merge")
    a_df, r_df, ma, mr = _metacells(oracle)
    mm = pd.DataFrame(g["mm"], columns=["Aligned_metacell_id", "Ref_metacell_id"])
    res = unpack_metacell_matches(mm, ma, mr, strategy="distribute")
    assert np.array_equal(res[["Aligned_cell_id", "Ref_cell_id"]].to_numpy(dtype=np.int64), g["unpack_dist_both"])
    res = unpack_metacell_matches(mm.assign(Ref_metacell_id=mm["Ref_metacell_id"] % len(r_df)), ma, r_df)
    assert np.array_equal(res[["Aligned_cell_id", "Ref_cell_id"]].to_numpy(dtype=np.int64), g["unpack_simple"])
    with pytest.raises(ValueError, match="requires aligned_df"):
        unpack_metacell_matches(mm, ma, r_df, strategy="nearest")
    with pytest.raises(ValueError, match="must provide both"):
        unpack_metacell_matches(mm, ma, mr, strategy="nearest", aligned_df=a_df)
    with pytest.raises(ValueError, match="Unknown strategy"):
        unpack_metacell_matches(mm, ma, mr, strategy="bogus")
    assert len(unpack_metacell_matches(mm.iloc[:0], ma, mr)) == 0


def _unpack_csr(mm, ma, mr, a_df, r_df):
    a_lists = [ma["members"].iloc[i] for i in mm["Aligned_metacell_id"]]
    r_lists = [mr["members"].iloc[j] for j in mm["Ref_metacell_id"]]
    a_off = np.concatenate(([0], np.cumsum([len(m) for m in a_lists])))
    r_off = np.concatenate(([0], np.cumsum([len(m) for m in r_lists])))
    a_flat, r_flat = np.concatenate(a_lists), np.concatenate(r_lists)
    axy = a_df.set_index("Cell_Num_Old").loc[a_flat, ["X", "Y"]].to_numpy()
    rxy = r_df.set_index("Cell_Num_Old").loc[r_flat, ["X", "Y"]].to_numpy()
    return a_off, r_off, a_flat, r_flat, axy, rxy


def test_oracle_batched_assign_matches_reference_fixture(oracle):
    g = load_golden("unpack_merge")
    a_df, r_df, ma, mr = _metacells(oracle)
    mm = pd.DataFrame(g["mm"], columns=["Aligned_metacell_id", "Ref_metacell_id"])
    a_off, r_off, a_flat, r_flat, axy, rxy = _unpack_csr(mm, ma, mr, a_df, r_df)
    assert (np.diff(a_off) > np.diff(r_off)).any() and (np.diff(a_off) < np.diff(r_off)).any()   # tiled and plain cases
    local = oracle.batched_assign(a_off, r_off, axy, rxy)
    got = np.column_stack((a_flat, r_flat[np.repeat(r_off[:-1], np.diff(a_off)) + local]))
    assert np.array_equal(got, g["unpack_near_both"])


@pytest.mark.parametrize("mode", ["uniform", "lattice", "near_ties"])
def test_oracle_batched_assign_equals_scipy(oracle, mode):
    """The solver is scipy's in the reference: pin the restatement on random and heavily tied problems."""
    rng = np.random.default_rng({"uniform": 1, "lattice": 2, "near_ties": 3}[mode])
    for trial in range(8):
        n = 1500
        na, nr = rng.integers(0 if trial == 0 else 1, 14 + 20 * (trial % 2), n), rng.integers(1, 14 + 20 * (trial // 4), n)
        a_off, r_off = np.concatenate(([0], np.cumsum(na))), np.concatenate(([0], np.cumsum(nr)))
        if mode == "uniform":
            axy, rxy = rng.uniform(0, 100, (a_off[-1], 2)), rng.uniform(0, 100, (r_off[-1], 2))
        elif mode == "lattice":
            axy, rxy = rng.integers(0, 4, (a_off[-1], 2)).astype(float), rng.integers(0, 4, (r_off[-1], 2)).astype(float)
        else:
            axy = rng.integers(0, 3, (a_off[-1], 2)) * 0.1 + rng.choice([0, 1e-9], (a_off[-1], 2))
            rxy = rng.integers(0, 3, (r_off[-1], 2)) * 0.1
        assert np.array_equal(oracle.batched_assign(a_off, r_off, axy, rxy), oracle.batched_assign_scipy(a_off, r_off, axy, rxy)), (mode, trial)


@pytest.mark.gpu
def test_unpack_nearest_on_device(oracle):
    from same_amd.metacell_utils import unpack_metacell_matches

    g = load_golden("unpack_merge")
    a_df, r_df, ma, mr = _metacells(oracle)
    mm = pd.DataFrame(g["mm"], columns=["Aligned_metacell_id", "Ref_metacell_id"])
    res = unpack_metacell_matches(mm, ma, mr, strategy="nearest", aligned_df=a_df, ref_df=r_df,
                                  aligned_original_idx_col="Cell_Num_Old", ref_original_idx_col="Cell_Num_Old")
    assert np.array_equal(res[["Aligned_cell_id", "Ref_cell_id"]].to_numpy(dtype=np.int64), g["unpack_near_both"])
    # frames indexed by the cell id directly (no *_original_idx_col), and a missing member label
    res2 = unpack_metacell_matches(mm, ma, mr, strategy="nearest", aligned_df=a_df.set_index("Cell_Num_Old", drop=False),
                                   ref_df=r_df.set_index("Cell_Num_Old", drop=False))
    assert res2.equals(res)
    with pytest.raises(KeyError):
        unpack_metacell_matches(mm, ma, mr, strategy="nearest", aligned_df=a_df.iloc[:10], ref_df=r_df,
                                aligned_original_idx_col="Cell_Num_Old", ref_original_idx_col="Cell_Num_Old")
    bad = r_df.copy()
    bad["X"] = np.nan
    with pytest.raises(ValueError, match="infeasible"):
        unpack_metacell_matches(mm, ma, mr, strategy="nearest", aligned_df=a_df, ref_df=bad,
                                aligned_original_idx_col="Cell_Num_Old", ref_original_idx_col="Cell_Num_Old")


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["uniform", "lattice", "near_ties"])
def test_batched_assign_kernel_vs_oracle(oracle, mode):
    from same_amd import ops

    rng = np.random.default_rng({"uniform": 11, "lattice": 12, "near_ties": 13}[mode])
    for trial in range(6):
        n = [1, 7, 300, 5000, 20000, 257][trial]
        na, nr = rng.integers(0 if trial == 2 else 1, 12 + 30 * (trial % 2), n), rng.integers(1, 12 + 30 * (trial // 3), n)
        a_off, r_off = np.concatenate(([0], np.cumsum(na))), np.concatenate(([0], np.cumsum(nr)))
        if mode == "uniform":
            axy, rxy = rng.uniform(0, 100, (a_off[-1], 2)), rng.uniform(0, 100, (r_off[-1], 2))
        elif mode == "lattice":
            axy, rxy = rng.integers(0, 4, (a_off[-1], 2)).astype(float), rng.integers(0, 4, (r_off[-1], 2)).astype(float)
        else:
            axy = rng.integers(0, 3, (a_off[-1], 2)) * 0.1 + rng.choice([0, 1e-9], (a_off[-1], 2))
            rxy = rng.integers(0, 3, (r_off[-1], 2)) * 0.1
        assert np.array_equal(ops.batched_assign(a_off, r_off, axy, rxy), oracle.batched_assign(a_off, r_off, axy, rxy)), (mode, trial)
    # degenerate shapes and the error codes of the ABI
    assert len(ops.batched_assign([0], [0], np.empty((0, 2)), np.empty((0, 2)))) == 0
    assert len(ops.batched_assign([0, 0], [0, 3], np.empty((0, 2)), np.zeros((3, 2)))) == 0
    from same_amd._lib import SameHipError
    with pytest.raises(SameHipError):            # aligned members but no ref member
        ops.batched_assign([0, 2], [0, 0], np.zeros((2, 2)), np.empty((0, 2)))
    with pytest.raises(SameHipError):
        ops.batched_assign([0, 1], [0, 1], np.full((1, 2), np.inf), np.zeros((1, 2)))
    with pytest.raises(SameHipError):            # more members than SAME_ASSIGN_MAX_MEMBERS
        ops.batched_assign([0, 513], [0, 1], np.zeros((513, 2)), np.zeros((1, 2)))
    big = np.random.default_rng(3).uniform(0, 50, (300, 2))     # a large but allowed problem, tiled columns
    assert np.array_equal(ops.batched_assign([0, 300], [0, 130], big, big[:130] + 0.25),
                          oracle.batched_assign([0, 300], [0, 130], big, big[:130] + 0.25))


def _merge_input():
    rows = []
    for w in range(3):
        for q in range(40):
            a = 10 * w + q
            rows.append({"window_id": w, "Aligned_Cell_Num_Old": a, "Ref_Cell_Num_Old": 1000 + a, "X": float(a), "Y": float(w),
                         "filtered_violation": bool((a + w) % 5 == 0)})
    dfm = pd.DataFrame(rows)
    dfm.loc[dfm.index[7], "filtered_violation"] = np.nan
    return [dfm[dfm.window_id == w].copy() for w in range(3)]


def _dedup_frames(g, tag):
    rows, viol = g[f"{tag}_in"], g[f"{tag}_in_viol"]
    dfm = pd.DataFrame({"window_id": rows[:, 0], "Aligned_Cell_Num_Old": rows[:, 1], "Ref_Cell_Num_Old": rows[:, 2],
                        "X": np.arange(len(rows), dtype=float), "Y": 0.0,
                        "filtered_violation": [None if np.isnan(v) else bool(v) for v in viol]})
    return [part.copy() for part in np.split(dfm, g[f"{tag}_cuts"])]


def check_merge_dedup_golden(dedup):
    """The reference's own output on tables built to show which duplicate of every (aligned, ref) pair survives its
    de-duplication (tests/golden/merge_dedup.npz, tools/gen_golden.py mergededup): X names the surviving input row."""
    from same_amd.merge import merge_window_matches_unique_ref

    g = load_golden("merge_dedup")
    for tag in ("a", "b"):
        res = merge_window_matches_unique_ref(_dedup_frames(g, tag), _dedup=dedup)
        assert np.array_equal(res["X"].to_numpy().astype(np.int64), g[f"{tag}_out_rows"]), tag
        assert np.array_equal(res[["window_id", "Aligned_Cell_Num_Old", "Ref_Cell_Num_Old"]].to_numpy(dtype=np.int64), g[f"{tag}_out"])
        assert np.array_equal(res["filtered_violation"].to_numpy().astype(np.uint8), g[f"{tag}_out_viol"])


def random_dedup_tables(rng, n, n_pairs, n_win):
    a = rng.integers(0, max(1, int(np.sqrt(n_pairs)) * 2), n).astype(np.int32)
    r = rng.integers(0, max(1, int(np.sqrt(n_pairs)) * 2), n).astype(np.int32)
    return (rng.random(n) < 0.35), rng.integers(0, n_win, n).astype(np.int32), a, r


def test_merge_dedup_oracle_is_the_reference_step(oracle):
    """The oracle's de-duplication (oracle/same_oracle.c: orc_merge_dedup) against the reference's own outputs and against the
    literal pandas calls of src/helpers.py:748-753 on random tables with heavy duplication and ties."""
    check_merge_dedup_golden(oracle.merge_dedup)
    rng = np.random.default_rng(8)
    for n, n_pairs, n_win in ((0, 1, 1), (1, 1, 1), (63, 10, 2), (64, 900, 5), (65, 4, 1), (5000, 300, 12), (20000, 20000, 40)):
        v, w, a, r = random_dedup_tables(rng, n, n_pairs, n_win)
        assert np.array_equal(oracle.merge_dedup(v, w, a, r), oracle.merge_dedup_pandas(v, w, a, r)), n


def test_merge_window_matches_unique_ref(oracle):
    from same_amd.merge import merge_window_matches_unique_ref as merge_on_gpu

    def merge_window_matches_unique_ref(lst, **kw):    # host logic of the merge, de-duplication by the oracle (no GPU here)
        return merge_on_gpu(lst, _dedup=oracle.merge_dedup, **kw)

    g = load_golden("unpack_merge")
    res = merge_window_matches_unique_ref(_merge_input())
    assert np.array_equal(res[["window_id", "Aligned_Cell_Num_Old", "Ref_Cell_Num_Old"]].to_numpy(dtype=np.int64), g["merge_rows"])
    assert np.array_equal(res["filtered_violation"].to_numpy().astype(np.uint8), g["merge_viol"])
    assert merge_window_matches_unique_ref([]).empty
    with pytest.raises(ValueError, match="Missing required columns"):
        merge_window_matches_unique_ref([pd.DataFrame({"window_id": [0]})])
    # conflicting proposals: the result is one-to-one and of maximum cardinality
    rng = np.random.default_rng(0)
    df = pd.DataFrame({"window_id": rng.integers(0, 4, 300), "Aligned_Cell_Num_Old": rng.integers(0, 80, 300),
                       "Ref_Cell_Num_Old": rng.integers(0, 80, 300), "X": 0.0, "Y": 0.0, "filtered_violation": rng.random(300) < 0.3})
    out = merge_window_matches_unique_ref([df])
    assert out["Aligned_Cell_Num_Old"].is_unique and out["Ref_Cell_Num_Old"].is_unique
    import networkx as nx
    G = nx.Graph()
    G.add_edges_from((f"a{a}", f"r{r}") for a, r in zip(df["Aligned_Cell_Num_Old"], df["Ref_Cell_Num_Old"]))
    want = len(nx.bipartite.hopcroft_karp_matching(G, top_nodes={n for n in G if n[0] == "a"})) // 2
    assert len(out) == want
    # the matching chosen among equally large ones does not depend on the order the window tables arrive in (ranks deal windows)
    shuffled = df.sample(frac=1.0, random_state=3).reset_index(drop=True)
    out2 = merge_window_matches_unique_ref([shuffled.iloc[150:], shuffled.iloc[:150]])
    key = ["Aligned_Cell_Num_Old", "Ref_Cell_Num_Old"]
    assert out[key].to_numpy().tolist() == out2[key].to_numpy().tolist()
    # a duplicated (aligned, ref) pair keeps the non-violating row, then the smaller window id
    d = pd.DataFrame({"window_id": [3, 1, 2], "Aligned_Cell_Num_Old": [7, 7, 7], "Ref_Cell_Num_Old": [9, 9, 9], "X": 0.0, "Y": 0.0,
                      "filtered_violation": [False, True, False]})
    assert merge_window_matches_unique_ref([d])["window_id"].tolist() == [2]
    # ids of any hashable kind (strings) and window ids that are not small non-negative ints
    d2 = pd.DataFrame({"window_id": [-5, 7.5, -5, 2 ** 40], "Aligned_Cell_Num_Old": ["c1", "c1", "c2", "c3"],
                       "Ref_Cell_Num_Old": ["r9", "r9", "r9", "r1"], "X": 0.0, "Y": 0.0, "filtered_violation": [True, True, False, False]})
    got = merge_window_matches_unique_ref([d2])
    assert len(got) == 2 and set(got["Ref_Cell_Num_Old"]) == {"r9", "r1"}


def test_merge_takes_a_missing_violation_flag_as_a_violation_for_any_column_type(oracle):
    """src/helpers.py:746-751: `.fillna(True).astype(bool)`.  A float column with NaN, an object column with None and pandas' nullable
    'boolean' column with pd.NA (whose truth value is undefined: it has to be masked before the cast) all read missing as True."""
    from same_amd.merge import merge_window_matches_unique_ref

    base = pd.DataFrame({"window_id": [0, 1, 0, 1], "Aligned_Cell_Num_Old": [7, 7, 8, 8], "Ref_Cell_Num_Old": [3, 3, 4, 4], "X": 0.0, "Y": 0.0})
    for flags in (pd.array([None, False, True, None], dtype="boolean"), np.array([np.nan, 0.0, 1.0, np.nan]),
                  np.array([None, False, True, None], dtype=object)):
        got = merge_window_matches_unique_ref([base.assign(filtered_violation=flags)], _dedup=oracle.merge_dedup)
        want = base.assign(filtered_violation=pd.Series(flags).fillna(True).astype(bool)).sort_values(
            ["filtered_violation", "window_id"], kind="mergesort").drop_duplicates(["Aligned_Cell_Num_Old", "Ref_Cell_Num_Old"])
        assert got["filtered_violation"].dtype == bool and got["window_id"].tolist() == want.sort_values("Aligned_Cell_Num_Old")["window_id"].tolist() == [1, 0]
        assert got["filtered_violation"].tolist() == [False, True]


def test_merge_matching_equals_the_matching_on_the_whole_graph(oracle):
    """merge_window_matches_unique_ref runs Hopcroft-Karp only on the cells some window disagrees about (an edge whose two cells have no
    other edge is in every maximum matching) and numbers those cells densely in the order of their ids.  The result must be the table
    the straightforward statement gives -- ONE graph over all surviving pairs, nodes numbered through the sorted ids
    (src/helpers.py:755-815) --, row for row: random tables from conflict-free to all-conflict, integer and string ids."""
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import maximum_bipartite_matching

    from same_amd.merge import _node_numbers, merge_window_matches_unique_ref

    def whole_graph(df, kept):
        (ac, na), (rc, nr) = _node_numbers(df["Aligned_Cell_Num_Old"].values[kept]), _node_numbers(df["Ref_Cell_Num_Old"].values[kept])
        g = csr_matrix((np.arange(1, len(ac) + 1, dtype=np.int64), (ac, rc)), shape=(na, nr))
        g.sort_indices()
        m = maximum_bipartite_matching(g, perm_type="column")
        node = np.repeat(np.arange(na, dtype=np.int64), np.diff(g.indptr))
        return df.iloc[kept[g.data[m[node] == g.indices] - 1]].reset_index(drop=True)

    rng = np.random.default_rng(1)
    rows = lone_only = contested = 0
    for case in range(240):
        n = int(rng.choice([1, 2, 5, 50, 300, 3000]))
        ids = int(rng.choice([1, 3, max(2, n // 3), n, 4 * n, 50 * n]))
        a, r = rng.integers(0, ids, n), rng.integers(0, ids, n)
        if case % 3 == 1:                           # a tiled run: one-to-one but for a few cells of the overlaps
            a, r = rng.permutation(max(n, ids))[:n], rng.permutation(max(n, ids))[:n]
            if case % 2:
                q = rng.integers(0, n, max(1, n // 10))
                a[q] = a[rng.integers(0, n, len(q))]
        if case % 3 == 2:
            a, r = np.array([f"c{v}" for v in a], dtype=object), np.array([f"r{v}" for v in r], dtype=object)
        df = pd.DataFrame({"window_id": rng.integers(0, 5, n), "Aligned_Cell_Num_Old": a, "Ref_Cell_Num_Old": r, "X": rng.random(n), "Y": 0.0,
                           "filtered_violation": rng.random(n) < 0.3})
        seen = {}

        def dedup(v, w, ac, rc):
            seen["kept"] = np.asarray(oracle.merge_dedup(v, w, ac, rc), dtype=np.int64)
            return seen["kept"]

        got = merge_window_matches_unique_ref([df], _dedup=dedup)
        want = whole_graph(df, seen["kept"])
        assert list(got.columns) == list(want.columns) and got.equals(want), (case, n, ids)
        rows += len(got)
        lone_only += len(got) == len(seen["kept"])
        contested += len(got) < len(seen["kept"])
    assert rows > 50_000 and lone_only > 20 and contested > 100       # both the shortcut alone and the graph path carried cases


def test_window_codes_follow_sort_values_for_any_column():
    """The device de-duplication sees window ids as int32 ranks in THEIR order: small non-negative integers as they are, anything
    else ranked -- and a missing id (None / NaN) last, where the reference's sort_values puts it (src/helpers.py:748)."""
    import pandas as pd

    from same_amd.merge import _window_codes

    assert np.array_equal(_window_codes(np.array([3, 0, 7])), [3, 0, 7])
    assert np.array_equal(_window_codes(np.array([3, -2, 2 ** 40])), [1, 0, 2])
    for col in (np.array(["w10", None, "w2", "w10"], dtype=object), np.array([2.5, np.nan, -1.0, 2.5])):
        codes = _window_codes(col)
        order = pd.DataFrame({"w": col}).sort_values("w", kind="mergesort").index.to_numpy()
        assert np.array_equal(np.argsort(codes, kind="stable"), order) and codes[0] == codes[3]


def test_priority_filter_closed_form_equals_the_reference_walk():
    """knn_utils.find_knn_with_cell_type_priority's pair filter (src/knn_utils.py:28-65) is written in the reference as a walk over
    the aligned rows carrying a set of claimed references; same_amd.knn.priority_filter is its closed form.  Pinned against the
    reference's own output on the three fixtures (`pairs_priority`) and against a literal restatement of the walk on 300 small
    random cases with heavy distance ties."""
    from conftest import frames_from_golden, load_golden
    from same_amd.knn import priority_filter

    for case in ("synthetic_example", "cfg1_500", "cfg2_small"):
        g = load_golden(case)
        a_df, r_df, _ = frames_from_golden(g)
        na = a_df.iloc[g["kept_aligned"]].reset_index(drop=True)
        nr = r_df.iloc[g["kept_ref"]].reset_index(drop=True)
        got, one, all_ = priority_filter(g["pairs"], na[["X", "Y"]].to_numpy(), nr[["X", "Y"]].to_numpy(), na["cell_type"].to_numpy(),
                                         nr["cell_type"].to_numpy())
        assert np.array_equal(got, g["pairs_priority"]) and one + all_ == len(np.unique(g["pairs"][:, 0]))
    rng = np.random.default_rng(0)
    for trial in range(300):
        n_m, n_r = (int(v) for v in rng.integers(1, 60, 2))
        P = int(rng.integers(0, 400))
        pairs = np.unique(np.column_stack((rng.integers(0, n_m, P), rng.integers(0, n_r, P))), axis=0).reshape(-1, 2)
        rng.shuffle(pairs)
        pairs = pairs[np.argsort(pairs[:, 0], kind="stable")]
        axy, rxy = rng.integers(0, 6, (n_m, 2)).astype(float), rng.integers(0, 6, (n_r, 2)).astype(float)
        at, rt = rng.integers(0, 3, n_m), rng.integers(0, 3, n_r)
        got, one, all_ = priority_filter(pairs, axy, rxy, at, rt)
        want, taken, n_one, n_all = [], set(), 0, 0
        for i in (np.unique(pairs[:, 0]) if len(pairs) else []):
            rows = [(q, int(j)) for q, (ii, j) in enumerate(pairs) if ii == i]
            d = [np.sqrt((axy[i, 0] - rxy[j, 0]) ** 2 + (axy[i, 1] - rxy[j, 1]) ** 2) for _, j in rows]
            js = [rows[z][1] for z in sorted(range(len(rows)), key=lambda z: (d[z], rows[z][0]))]   # stable sort by distance (:40-49)
            if rt[js[0]] == at[i] and js[0] not in taken:                                          # :56-59
                want.append((int(i), js[0]))
                taken.add(js[0])
                n_one += 1
            else:                                                                                   # :64
                want.extend((int(i), j) for j in js)
                n_all += 1
        assert [tuple(r) for r in got.tolist()] == want and (one, all_) == (n_one, n_all), trial
    empty, a, b = priority_filter(np.array([]), np.zeros((0, 2)), np.zeros((0, 2)), np.zeros(0), np.zeros(0))
    assert len(empty) == 0 and (a, b) == (0, 0)


def test_gpu_telemetry_reads_a_hwmon_tree(tmp_path):
    """same_amd/telemetry.py against a hand-made sysfs tree: power, clock, junction / HBM temperatures found by label, the cap,
    and the steady-state window; a node that is missing is reported as missing, never guessed."""
    import time
    from same_amd.telemetry import GpuTelemetry

    hw = tmp_path / "0000:05:00.0" / "hwmon" / "hwmon3"
    hw.mkdir(parents=True)
    files = {"power1_input": "1372000000", "power1_cap": "1400000000", "freq1_input": "1800000000",
             "temp1_label": "edge", "temp1_input": "61000", "temp2_label": "junction", "temp2_input": "83000", "temp2_crit": "100000",
             "temp3_label": "mem", "temp3_input": "72000", "temp3_crit": "115000", "temp4_label": "other", "temp4_input": "1000"}
    for name, text in files.items():
        (hw / name).write_text(text + "\n")
    t = GpuTelemetry("0000:05:00.0", period_s=0.005, sysfs_root=str(tmp_path))
    assert t.available() and t.cap_w == 1400.0 and set(t.temp_paths) == {"edge", "junction", "mem"}
    t.start()
    time.sleep(0.1)
    s = t.stop()
    assert s["samples"] >= 8 and s["power_steady"]["mean"] == 1372.0 and s["sclk_steady"]["mean"] == 1800.0
    assert s["temperature_steady"]["junction"]["mean"] == 83.0 and s["temperature_steady"]["mem"]["max"] == 72.0
    assert s["temperature_crit_c"] == {"edge": None, "junction": 100.0, "mem": 115.0}
    none = GpuTelemetry("0000:06:00.0", sysfs_root=str(tmp_path))
    assert not none.available() and none.sample()[1:4] == (None, None, None)


@pytest.mark.gpu
def test_merge_dedup_on_device(oracle):
    """f3 on the GPU (csrc/merge.hip through ops.merge_dedup): (i) the whole merge, default path, against the reference's own
    outputs -- both the disjoint-pair tables that expose which duplicate survives and the small unpack_merge table; (ii) the
    de-duplication against the oracle, bit for bit, from the empty table through sizes around the wave (64), the LDS sort
    block (2048) and the first global bitonic stages, to 300k rows of window tables with every pair proposed several times."""
    from same_amd import ops
    from same_amd.merge import merge_window_matches_unique_ref

    check_merge_dedup_golden(None)                      # None = the product's default: ops.merge_dedup on the GPU
    g = load_golden("unpack_merge")
    res = merge_window_matches_unique_ref(_merge_input())
    assert np.array_equal(res[["window_id", "Aligned_Cell_Num_Old", "Ref_Cell_Num_Old"]].to_numpy(dtype=np.int64), g["merge_rows"])
    assert np.array_equal(res["filtered_violation"].to_numpy().astype(np.uint8), g["merge_viol"])
    rng = np.random.default_rng(9)
    for n, n_pairs, n_win in ((0, 1, 1), (1, 1, 1), (2, 1, 1), (63, 10, 2), (64, 900, 5), (65, 4, 1), (2047, 50, 3), (2048, 2048, 7),
                              (2049, 10, 1), (4096, 100, 9), (5000, 300, 12), (70000, 70000, 100), (300000, 40000, 400)):
        v, w, a, r = random_dedup_tables(rng, n, n_pairs, n_win)
        got = ops.merge_dedup(v, w, a, r)
        assert got.dtype == np.int32 and np.array_equal(got, oracle.merge_dedup(v, w, a, r)), n
    # all rows one pair; all rows distinct pairs; every row violating; window ids at the top of the int32 range
    n = 3000
    one = ops.merge_dedup(rng.random(n) < 0.5, rng.integers(0, 4, n), np.zeros(n, np.int32), np.zeros(n, np.int32))
    assert len(one) == 1
    distinct = ops.merge_dedup(np.ones(n, bool), np.full(n, 2 ** 31 - 1), np.arange(n), np.arange(n)[::-1].copy())
    assert np.array_equal(distinct, np.arange(n))
    from same_amd._lib import SameHipError
    with pytest.raises(SameHipError):
        ops.merge_dedup([False], [-1], [0], [0])
    with pytest.raises(ValueError):
        ops.merge_dedup([False, True], [0], [0], [0])
