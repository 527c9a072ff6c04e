"""Window-merge helpers, SURVEY 8(f3): helpers.merge_window_matches_unique_ref and
helpers.load_matching_results (src/helpers.py:667-815), same signatures.

Two halves.  The de-duplication of the concatenated per-window tables (src/helpers.py:745-753: stable sort by
violation and window id, first row of every (aligned, ref) pair) is data-parallel and runs on the GPU
(csrc/merge.hip, `ops.merge_dedup`: key build, bitonic sort, hash-table first-occurrence, ordered compaction).  The
maximum-cardinality matching on the rows that survive (:755-815) is a sequential graph algorithm
and stays on the host: scipy's Hopcroft-Karp on integer node ids, which makes the choice among equally large
matchings deterministic (the reference's networkx call iterates a set of string labels, so its choice varies with
PYTHONHASHSEED -- the reference output is one of the maximum matchings, as this is)."""
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


GATHER_THREADS = 8      # column-parallel copies of a 10^6-row table (22 columns x 950k rows on a 16-CPU host: 16.9 ms on one thread, 4.5 on four, 1.9 on eight)


def _window_codes(window_id):
    """window ids as non-negative int32 in their own order (they are small non-negative ints in every reference flow,
    src/same.py:582; anything else -- negative, huge, float -- is ranked first, which preserves the order)."""
    w = np.asarray(window_id)
    if w.dtype.kind in "iu" and (len(w) == 0 or (w.min() >= 0 and w.max() < 2 ** 31)):
        return w.astype(np.int32)
    # ranks in sorted order; missing ids (None / NaN in an object or float column) go last, where the reference's sort_values
    # (na_position='last', src/helpers.py:748) puts them
    codes, uniques = pd.factorize(w, sort=True)
    return np.where(codes < 0, len(uniques), codes).astype(np.int32)


def _equality_codes(values):
    """int32 codes with equal id <=> equal code, for the device de-duplication: the ids themselves when they are small
    non-negative integers (cell numbers usually are), a hash factorisation otherwise (strings, huge or negative numbers, NaN)."""
    v = np.asarray(values)
    if v.dtype.kind in "iu" and (len(v) == 0 or (v.min() >= 0 and v.max() < 2 ** 31)):
        return v.astype(np.int32)
    return pd.factorize(v, use_na_sentinel=False)[0].astype(np.int32)


def _node_numbers(values):
    """Node numbers for the matching graph in the ORDER of the ids (the reference numbers its nodes through sorted(unique ids),
    src/helpers.py:760-763) -> (numbers, node count).  Dense-ish non-negative integer ids are their own numbers (ids that do
    not occur are isolated nodes); anything else is factorised with sorted uniques."""
    v = np.asarray(values)
    if v.dtype.kind in "iu" and len(v) and v.min() >= 0 and v.max() < 4 * len(v) + 1024:
        return v.astype(np.int64), int(v.max()) + 1
    codes, uniques = pd.factorize(v, sort=True)
    return codes.astype(np.int64), len(uniques)


def _present(codes, n):
    """bool[n]: which node numbers occur in `codes`"""
    out = np.zeros(n, bool)
    out[codes] = True
    return out


def _take_rows(df, rows):
    """df.iloc[rows].reset_index(drop=True), column by column when every column is a plain numpy dtype: no block bookkeeping, and the
    columns of a table of 10^6 rows are gathered side by side (numpy copies without the interpreter lock; the merge runs when the
    window loop is over, so the host is otherwise idle).  Frames with extension dtypes take pandas' own path."""
    if len(rows) < 50_000 or not all(isinstance(dt, np.dtype) for dt in df.dtypes) or not df.columns.is_unique:
        return df.iloc[rows].reset_index(drop=True)
    from concurrent.futures import ThreadPoolExecutor

    cols = [df[c].to_numpy() for c in df.columns]
    with ThreadPoolExecutor(max_workers=GATHER_THREADS) as pool:
        taken = list(pool.map(lambda col: col.take(rows), cols))
    return pd.DataFrame(dict(zip(df.columns, taken)), copy=False)


def merge_window_matches_unique_ref(matches_list, cell_id_col="Cell_Num_Old", _dedup=None):
    """src/helpers.py:692-815.  `_dedup(viol, window_id, aligned_code, ref_code) -> surviving row indices` defaults to the HIP
    kernel chain (`ops.merge_dedup`); the CPU tests pass the oracle's restatement of the same step."""
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import maximum_bipartite_matching

    from ._trace import stage as marked

    if not matches_list:
        return pd.DataFrame()
    with marked("merge: concatenate the window tables"):
        merged_df = pd.concat(matches_list, ignore_index=True)
        aligned_col, ref_col = f"Aligned_{cell_id_col}", f"Ref_{cell_id_col}"
        required = ["window_id", aligned_col, ref_col, "X", "Y", "filtered_violation"]
        missing = [c for c in required if c not in merged_df.columns]
        if missing:
            raise ValueError(f"Missing required columns in matches: {missing}")
        fv = merged_df["filtered_violation"]
        if fv.dtype != bool:          # .fillna(True).astype(bool) of src/helpers.py:746-751 on the column's array (the pandas call warns about its own downcast)
            v = fv.to_numpy()
            merged_df["filtered_violation"] = np.where(pd.isna(v), True, v.astype(bool))
    # one row per (aligned, ref) pair: non-violating first, then the smaller window id, then the earlier row (:748-753);
    # the ids may be anything hashable, the device sees integer codes of them (equal id <=> equal code)
    if _dedup is None:
        from . import ops

        _dedup = ops.merge_dedup      # the device step; there is no host substitute in the product (a missing GPU raises SameHipError)
    with marked("merge: de-duplication (codes + device)"):
        kept = _dedup(merged_df["filtered_violation"].to_numpy(), _window_codes(merged_df["window_id"].to_numpy()),
                      _equality_codes(merged_df[aligned_col].values), _equality_codes(merged_df[ref_col].values))
        kept = np.asarray(kept, dtype=np.int64)     # rows of merged_df that survive, in the order the reference's frame has after :748-753
    with marked("merge: graph of the surviving pairs"):
        (a_codes, n_a), (r_codes, n_r) = _node_numbers(merged_df[aligned_col].values[kept]), _node_numbers(merged_df[ref_col].values[kept])
        # An edge whose two cells have no other edge is in every maximum matching: only the cells that some window disagrees about
        # (an aligned cell proposed for two references, a reference proposed to two aligned cells) need the graph algorithm -- in a
        # tiled run these are the cells of the window overlaps at most.  row_of[a] = the position in `kept` of a's matched edge.
        deg_a, deg_r = np.bincount(a_codes, minlength=n_a), np.bincount(r_codes, minlength=n_r)
        row_of = np.full(n_a, -1, np.int64)
        if len(a_codes) == 0 or (deg_a.max() <= 1 and deg_r.max() <= 1):      # nobody disagrees (a tiled run whose overlaps agree): all edges stand
            rest = np.zeros(0, np.int64)
            row_of[a_codes] = np.arange(len(a_codes), dtype=np.int64)
        else:
            lone = (deg_a[a_codes] == 1) & (deg_r[r_codes] == 1)
            rest = np.flatnonzero(~lone)
            lone_at = np.flatnonzero(lone)
            row_of[a_codes[lone_at]] = lone_at
        graph = None
        if len(rest):
            # the contested cells renumbered densely IN THE ORDER of their ids (node numbers come from the sorted ids, and every
            # adjacency list is sorted below): the matching chosen among equally large ones depends on the ids alone, not on the
            # order the window tables arrived in (1 rank or 8)
            a_rest, r_rest = a_codes[rest], r_codes[rest]
            a_new, r_new = np.cumsum(_present(a_rest, n_a)) - 1, np.cumsum(_present(r_rest, n_r)) - 1
            # every edge carries its position in `kept` (+1: an explicit zero would be dropped), so the rows of the matched edges
            # can be read off the matrix afterwards without a second sort; edges are unique after the de-duplication: nothing is summed
            graph = csr_matrix((rest + 1, (a_new[a_rest], r_new[r_rest])), shape=(int(a_new[-1]) + 1, int(r_new[-1]) + 1))
            graph.sort_indices()
    with marked("merge: maximum matching"):
        if graph is not None:
            match_r = maximum_bipartite_matching(graph, perm_type="column")  # ref node matched to each aligned node, -1 = none (structure only)
    with marked("merge: rows of the matched pairs"):
        if graph is not None:
            node_of_edge = np.repeat(np.arange(graph.shape[0], dtype=np.int64), np.diff(graph.indptr))
            won = graph.data[match_r[node_of_edge] == graph.indices] - 1       # positions in `kept` of the contested edges that are matched
            row_of[a_codes[won]] = won
        selected = row_of[row_of >= 0]                                         # aligned ids ascending (:799-808)
        return _take_rows(merged_df, kept[selected])                           # ONE gather of the frame: the rows of the matched edges
