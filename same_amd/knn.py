"""KNN prune within a radius: utils.find_knn_within_radius (src/utils.py:709-742) and
knn_utils.find_knn_with_cell_type_priority (src/knn_utils.py:5-78), same signatures.

The per-row radius query + top-k runs in the HIP kernel (csrc/knn.hip); the frame compaction
(np.unique of used rows, iloc + reset_index, pair re-indexing) is host work on small index
arrays, vectorised here instead of the reference's dict lookups."""
import numpy as np

from . import ops


def _xy(df):
    return np.ascontiguousarray(df[["X", "Y"]].to_numpy(dtype=np.float64))


def pairs_from_padded(idx, row_offset=0):
    """Row-major walk of the -1 padded lists: aligned i ascending, rank ascending within i."""
    # the lists are filled from the left, but nothing here relies on it: a boolean take walks the array in the same row-major
    # order np.nonzero would, at a fraction of its cost
    kept = idx >= 0
    out = np.empty((int(np.count_nonzero(kept)), 2), np.int64)
    out[:, 0] = np.repeat(np.arange(int(row_offset), int(row_offset) + len(idx), dtype=np.int64), kept.sum(axis=1))
    out[:, 1] = idx[kept]
    return out


def compact_pairs(aligned_df, ref_df, knn_pairs):
    """src/utils.py:734-742."""
    # np.unique(..., return_inverse=True) of row / ref indices, without sorting: indices are positions in the frames
    def _unique_inverse(col, n):
        used = np.zeros(n, dtype=bool)
        used[col] = True
        return np.flatnonzero(used), (np.cumsum(used) - 1)[col]

    ua, inv_a = _unique_inverse(knn_pairs[:, 0], len(aligned_df)) if len(knn_pairs) else (np.zeros(0, np.int64), np.zeros(0, np.int64))
    ur, inv_r = _unique_inverse(knn_pairs[:, 1], len(ref_df)) if len(knn_pairs) else (np.zeros(0, np.int64), np.zeros(0, np.int64))
    new_aligned_df = aligned_df.iloc[ua].reset_index(drop=True)
    new_ref_df = ref_df.iloc[ur].reset_index(drop=True)
    new_valid_pairs = np.column_stack((inv_a.reshape(-1), inv_r.reshape(-1))).astype(np.int64)
    if len(knn_pairs) == 0:
        new_valid_pairs = np.array([])  # what np.array([]) of an empty comprehension gives (src/utils.py:741)
    return new_aligned_df, new_ref_df, new_valid_pairs


def find_knn_within_radius(aligned_df, ref_df, radius=25, knn=5, verbose=True, ctx=None):
    axy, rxy = _xy(aligned_df), _xy(ref_df)
    # the argument checks cKDTree makes for the reference (src/utils.py:714,722), with its messages
    if not np.isfinite(rxy).all():
        raise ValueError("data must be finite, check for nan or inf values")
    if len(rxy) and not np.isfinite(axy).all():
        raise ValueError("'x' must be finite, check for nan or inf values")
    radius = float(radius)
    if radius != radius or int(knn) <= 0 or len(rxy) == 0:   # NaN radius / knn=0 select nothing there (ValueError upstream in run_same)
        idx = np.full((len(axy), 1), -1, np.int32)
    else:
        # cKDTree compares squared distances, so a negative radius acts as its magnitude
        idx, _, _ = ops.knn_prune(axy, rxy, abs(radius), knn, want_d2=False, ctx=ctx)
    knn_pairs = pairs_from_padded(idx)
    if verbose:
        print(f"Number of valid pairs after knn: {len(knn_pairs)}")
    return compact_pairs(aligned_df, ref_df, knn_pairs)


def priority_filter(all_pairs, axy, rxy, atype, rtype):
    """The pair filter of knn_utils.find_knn_with_cell_type_priority (src/knn_utils.py:28-65) on arrays.
    -> (filtered pairs (n, 2) int64 in the reference's order, rows that kept one pair, rows that kept all).

    The reference walks the aligned rows in ascending order carrying a set of already claimed references: a row whose NEAREST
    reference has its cell type and is not yet claimed keeps only that pair and claims it; every other row keeps all its pairs.
    Only nearest references are ever claimed, and only by such rows, so the walk has a closed form: among the rows whose
    nearest reference j has their type, the first one (smallest row) gets j -- one `np.unique(..., return_index=True)`."""
    all_pairs = np.asarray(all_pairs, dtype=np.int64).reshape(-1, 2)
    if len(all_pairs) == 0:
        return all_pairs, 0, 0
    # the reference re-sorts each row by sqrt(dx^2+dy^2) with a stable sort (src/knn_utils.py:40-49)
    d = np.sqrt((axy[all_pairs[:, 0], 0] - rxy[all_pairs[:, 1], 0]) ** 2 + (axy[all_pairs[:, 0], 1] - rxy[all_pairs[:, 1], 1]) ** 2)
    order = np.lexsort((np.arange(len(d)), d, all_pairs[:, 0]))
    pi, pj = all_pairs[order, 0], all_pairs[order, 1]
    starts = np.flatnonzero(np.r_[True, pi[1:] != pi[:-1]])
    row_of_pair = np.cumsum(np.r_[True, pi[1:] != pi[:-1]]) - 1          # index into `starts` for every pair
    nearest = pj[starts]
    same = np.asarray(rtype)[nearest] == np.asarray(atype)[pi[starts]]
    cand = np.flatnonzero(same)                                            # rows (in walk order) that may claim their nearest
    _, first = np.unique(nearest[cand], return_index=True)                 # first row per claimed reference
    winner = np.zeros(len(starts), bool)
    winner[cand[first]] = True
    keep = ~winner[row_of_pair]
    keep[starts[winner]] = True                                            # a winning row keeps its nearest pair only
    return np.column_stack((pi[keep], pj[keep])), int(winner.sum()), int(len(starts) - winner.sum())


def find_knn_with_cell_type_priority(aligned_df, ref_df, radius, knn=5, verbose=True, ctx=None):
    """src/knn_utils.py:5-78: the radius / top-k prune, then the cell-type-priority filter over the pruned lists."""
    aligned_df, ref_df, all_pairs = find_knn_within_radius(aligned_df, ref_df, radius, knn=knn, verbose=verbose, ctx=ctx)
    pairs, same_type, keep_all = priority_filter(all_pairs, _xy(aligned_df), _xy(ref_df), aligned_df["cell_type"].to_numpy(),
                                                 ref_df["cell_type"].to_numpy())
    filtered = list(zip(pairs[:, 0].tolist(), pairs[:, 1].tolist()))       # a list of (i, j) tuples, as the reference returns
    if verbose:
        print(f"Total pairs after filtering: {len(filtered)}")
        # the reference divides unguarded here (src/knn_utils.py:76)
        print(f"Average pairs per matched point: {len(filtered) / (same_type + keep_all):.2f}")
    elif same_type + keep_all == 0:
        raise ZeroDivisionError("division by zero")  # same failure as the reference's summary print
    return aligned_df, ref_df, filtered
