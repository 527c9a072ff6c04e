"""MIP-start heuristics, same signatures as src/init_helpers.py.

The per-row minimum pair cost (:118-122) and the dense assignment matrix fill (:151-155) run in
csrc/sweep.hip; the greedy sort + scan (:109-133) is resolved on the device by an equivalent
parallel rule (csrc/match.hip); scipy's linear_sum_assignment stays on the host."""
from typing import List, Optional, Set, Tuple

import numpy as np

from . import ops
from ._rows import values_array


_METHODS = ("greedy", "hungarian")


def _last_pair_index(pairs, n_ref, rows, cols):
    """Index into `pairs` of the LAST occurrence of each (row, col) (the reference's dict keeps the last one,
    src/init_helpers.py:160), or -1 where the pair does not exist.  One sort + one binary search for all queries."""
    if len(pairs) == 0 or len(rows) == 0:
        return np.full(len(rows), -1, np.int64)
    keys = pairs[:, 0] * np.int64(n_ref) + pairs[:, 1]
    order = np.argsort(keys, kind="stable")
    sorted_keys = keys[order]
    want = np.asarray(rows, np.int64) * np.int64(n_ref) + np.asarray(cols, np.int64)
    pos = np.searchsorted(sorted_keys, want, side="right") - 1
    hit = (pos >= 0) & (sorted_keys[np.maximum(pos, 0)] == want)
    return np.where(hit, order[np.maximum(pos, 0)], -1)


def compute_mip_start_pairs(*, valid_pairs, costs, n_aligned, n_ref, aligned_sizes, no_match_penalty, max_matches,
                            init_method, init_big_m: float = 1e9, init_hungarian_max_n: int = 2000, verbose: bool = True,
                            ctx=None) -> Tuple[List[Tuple[int, int, int]], Set[int]]:
    """src/init_helpers.py:46-177 -> ([(aligned i, ref j, variable index)], {aligned rows started as unmatched})."""
    method = str(init_method).lower()
    # the reference's argument errors (src/init_helpers.py:94-107), same texts
    for bad, text in ((method not in _METHODS, f"Unknown init_method={init_method!r}. Use 'greedy' or 'hungarian'."),
                      (method == "hungarian" and max_matches != 1, "init_method='hungarian' requires max_matches == 1."),
                      (len(valid_pairs) != len(costs), "valid_pairs and costs must have the same length.")):
        if bad:
            raise ValueError(text)

    cost = values_array(costs, dtype=float)
    stay_cost = float(no_match_penalty) * np.asarray(aligned_sizes, dtype=float)     # price of leaving row i unmatched
    pairs = np.asarray(valid_pairs, dtype=np.int64).reshape(-1, 2)

    if method == "greedy":
        # per-row minimum on the device; then the reference's stable sort + sequential scan (:109-133) resolved by an
        # equivalent parallel rule (csrc/match.hip).  The chosen pairs come back one per aligned row and are listed in
        # the scan's own order: by cost, ties by pair index.
        wants_match = ops.pair_rowmin(pairs, cost, n_aligned, ctx=ctx) < stay_cost
        pair_of_row, _rounds = ops.greedy_match(pairs, cost, n_aligned, n_ref, wants_match, ctx=ctx)
        taken = pair_of_row[pair_of_row >= 0].astype(np.int64)
        taken = taken[np.lexsort((taken, cost[taken]))]
        started = list(zip(pairs[taken, 0].tolist(), pairs[taken, 1].tolist(), taken.tolist()))
        return started, set(np.flatnonzero(pair_of_row < 0).tolist())

    if n_aligned + n_ref > int(init_hungarian_max_n):
        if verbose:
            print(f"Skipping Hungarian init: n_aligned+n_ref={n_aligned+n_ref} > init_hungarian_max_n={init_hungarian_max_n}")
        return [], set()
    from scipy.optimize import linear_sum_assignment

    # dense [n_aligned][n_ref + n_aligned] matrix (:151-155) filled on the device, solved by scipy on the host
    dense = ops.assign_matrix(pairs, cost, stay_cost, n_aligned, n_ref, float(init_big_m), ctx=ctx)
    rows, cols = linear_sum_assignment(dense)
    # a row is started as matched when it was given a real reference column whose entry is a pair cost (not the big-M
    # filler), otherwise as unmatched (:163-175); the assignment is one-to-one, so no reference can be taken twice
    real = (cols < n_ref) & (dense[rows, cols] < 0.5 * float(init_big_m))
    var = _last_pair_index(pairs, n_ref, rows[real], cols[real])
    started = [(int(i), int(j), int(v)) for i, j, v in zip(rows[real], cols[real], var) if v >= 0]
    return started, set(rows[~real].tolist())


def apply_mip_start(*, x_vars, no_match_vars, valid_pairs, costs, n_aligned, n_ref, aligned_sizes, no_match_penalty,
                    max_matches, init_method: Optional[str], init_big_m: float = 1e9, init_hungarian_max_n: int = 2000,
                    verbose: bool = True, ctx=None) -> None:
    """src/init_helpers.py:180-244: put the heuristic solution into the solver variables' `.Start`.
    End state as in the reference: every pair variable 0 except the chosen ones; a row's no-match variable is 1 only
    if the heuristic left it unmatched (a chosen pair always clears it)."""
    if init_method is None:
        return
    started, left_out = compute_mip_start_pairs(
        valid_pairs=valid_pairs, costs=costs, n_aligned=n_aligned, n_ref=n_ref, aligned_sizes=aligned_sizes,
        no_match_penalty=no_match_penalty, max_matches=max_matches, init_method=init_method, init_big_m=init_big_m,
        init_hungarian_max_n=init_hungarian_max_n, verbose=verbose, ctx=ctx)
    if not (started or left_out):      # skipped Hungarian start: leave the solver's defaults alone
        return
    x0 = np.zeros(len(valid_pairs))
    idle0 = np.zeros(int(n_aligned))
    idle0[np.fromiter(left_out, np.int64, len(left_out))] = 1.0
    for i, _j, v in started:
        x0[v] = 1.0
        idle0[i] = 0.0
    for v, value in enumerate(x0.tolist()):
        x_vars[v].Start = value
    for i, value in enumerate(idle0.tolist()):
        no_match_vars[i].Start = value
    if verbose:
        print(f"Initialized MIP start ({str(init_method).lower()}): {len(started)} matches, {len(left_out)} unmatched")
