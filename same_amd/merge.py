"""Window-merge helpers, SURVEY 8(f3): helpers.merge_window_matches_unique_ref and
helpers.load_matching_results (src/helpers.py:667-815), same signatures.

Host-side graph matching on the small merged match table (not data-parallel).  The
de-duplication rule and the maximum-cardinality matching are the reference's; the matching itself is
computed with scipy's Hopcroft-Karp on integer node ids, which makes the choice among equally large
matchings deterministic (the reference's networkx call iterates a set of string labels, so its choice
varies with PYTHONHASHSEED -- the reference output is one of the maximum matchings, as this is)."""
import os

import numpy as np
import pandas as pd


def load_matching_results(outprefix):
    """src/helpers.py:667-689 -> (var_out, aligned_df, ref_df, matches_df).
    Reads the pickle-free `var_out.json` + `var_out.npz` pair run_same writes (same_amd/varout.py); a directory written by
    the reference itself (only `var_out.npy`, a pickle) is read through the numpy-only unpickler, never `allow_pickle=True`."""
    from . import varout

    if os.path.exists(os.path.join(outprefix, "var_out.json")):
        var_out = varout.load(outprefix)
    else:
        var_out = varout.load_legacy_npy(os.path.join(outprefix, "var_out.npy"))
    return (var_out, pd.read_csv(os.path.join(outprefix, "aligned_df.csv")), pd.read_csv(os.path.join(outprefix, "ref_df.csv")),
            pd.read_csv(os.path.join(outprefix, "matches_df.csv")))


def merge_window_matches_unique_ref(matches_list, cell_id_col="Cell_Num_Old"):
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import maximum_bipartite_matching

    if not matches_list:
        return pd.DataFrame()
    merged_df = pd.concat(matches_list, ignore_index=True)
    aligned_col, ref_col = f"Aligned_{cell_id_col}", f"Ref_{cell_id_col}"
    required = ["window_id", aligned_col, ref_col, "X", "Y", "filtered_violation"]
    missing = [c for c in required if c not in merged_df.columns]
    if missing:
        raise ValueError(f"Missing required columns in matches: {missing}")
    merged_df["filtered_violation"] = merged_df["filtered_violation"].fillna(True).astype(bool)
    # one row per (aligned, ref) pair: non-violating first, then the smaller window id (:748-753)
    merged_df = merged_df.sort_values(by=["filtered_violation", "window_id"], ascending=[True, True], kind="mergesort")
    merged_df = merged_df.drop_duplicates(subset=[aligned_col, ref_col], keep="first")
    a_codes, a_uniques = pd.factorize(merged_df[aligned_col].values, sort=True)
    r_codes, _ = pd.factorize(merged_df[ref_col].values, sort=True)
    n_a, n_r = len(a_uniques), int(r_codes.max()) + 1 if len(r_codes) else 0
    graph = csr_matrix((np.ones(len(a_codes), np.int8), (a_codes, r_codes)), shape=(n_a, n_r))
    match_r = maximum_bipartite_matching(graph, perm_type="column")      # ref index matched to each aligned node, -1 = none
    row_of_edge = {(a, r): i for i, (a, r) in enumerate(zip(a_codes.tolist(), r_codes.tolist()))}
    selected = [row_of_edge[(a, int(r))] for a, r in enumerate(match_r.tolist()) if r >= 0]   # aligned ids ascending (:799-808)
    return merged_df.iloc[selected].copy().reset_index(drop=True)
