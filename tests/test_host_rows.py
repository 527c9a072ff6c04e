"""SURVEY 8(f3, f4): host-side rows (window merge, metacell unpacking) against reference-generated fixtures.
CPU only: these functions touch no kernel (the metacell objects they consume are rebuilt with the oracle)."""
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
    res = unpack_metacell_matches(mm, ma, mr, strategy="nearest", aligned_df=a_df, ref_df=r_df,
                                  aligned_original_idx_col="Cell_Num_Old", ref_original_idx_col="Cell_Num_Old")
    assert np.array_equal(res[["Aligned_cell_id", "Ref_cell_id"]].to_numpy(dtype=np.int64), g["unpack_near_both"])
    res = unpack_metacell_matches(mm.assign(Ref_metacell_id=mm["Ref_metacell_id"] % len(r_df)), ma, r_df)
    assert np.array_equal(res[["Aligned_cell_id", "Ref_cell_id"]].to_numpy(dtype=np.int64), g["unpack_simple"])
    with pytest.raises(ValueError, match="requires aligned_df"):
        unpack_metacell_matches(mm, ma, r_df, strategy="nearest")
    with pytest.raises(ValueError, match="must provide both"):
        unpack_metacell_matches(mm, ma, mr, strategy="nearest", aligned_df=a_df)
    with pytest.raises(ValueError, match="Unknown strategy"):
        unpack_metacell_matches(mm, ma, mr, strategy="bogus")
    assert len(unpack_metacell_matches(mm.iloc[:0], ma, mr)) == 0


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


def test_merge_window_matches_unique_ref():
    from same_amd.merge import merge_window_matches_unique_ref

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
    # a duplicated (aligned, ref) pair keeps the non-violating row, then the smaller window id
    d = pd.DataFrame({"window_id": [3, 1, 2], "Aligned_Cell_Num_Old": [7, 7, 7], "Ref_Cell_Num_Old": [9, 9, 9], "X": 0.0, "Y": 0.0,
                      "filtered_violation": [False, True, False]})
    assert merge_window_matches_unique_ref([d])["window_id"].tolist() == [2]
