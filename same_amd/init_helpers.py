"""MIP-start heuristics, same signatures as src/init_helpers.py.

The per-row minimum pair cost (:118-122) and the dense assignment matrix fill (:151-155) run in
csrc/sweep.hip; the greedy sort + scan (:109-133) is resolved on the device by an equivalent
parallel rule (csrc/match.hip); scipy's linear_sum_assignment stays on the host."""
from typing import List, Optional, Set, Tuple

import numpy as np

from . import ops


def compute_mip_start_pairs(*, valid_pairs, costs, n_aligned, n_ref, aligned_sizes, no_match_penalty, max_matches,
                            init_method, init_big_m: float = 1e9, init_hungarian_max_n: int = 2000, verbose: bool = True,
                            ctx=None) -> Tuple[List[Tuple[int, int, int]], Set[int]]:
    method = str(init_method).lower()
    if method not in {"greedy", "hungarian"}:
        raise ValueError(f"Unknown init_method={init_method!r}. Use 'greedy' or 'hungarian'.")
    if method == "hungarian" and max_matches != 1:
        raise ValueError("init_method='hungarian' requires max_matches == 1.")
    if len(valid_pairs) != len(costs):
        raise ValueError("valid_pairs and costs must have the same length.")

    costs_arr = np.asarray(costs, dtype=float)
    unmatched_cost = float(no_match_penalty) * np.asarray(aligned_sizes, dtype=float)
    pairs = np.asarray(valid_pairs, dtype=np.int64).reshape(-1, 2)
    chosen_pairs: List[Tuple[int, int, int]] = []
    chosen_unmatched: Set[int] = set()

    if method == "greedy":
        best_cost_per_i = ops.pair_rowmin(pairs, costs_arr, n_aligned, ctx=ctx)
        prefer_match = best_cost_per_i < unmatched_cost
        # the sort + sequential scan of the reference (:109-133), resolved on the device by an equivalent
        # parallel rule (csrc/match.hip); the chosen pairs come back per aligned row and are put in the
        # reference's order (cost, then pair index = its stable sort) here
        match_pair, _rounds = ops.greedy_match(pairs, costs_arr, n_aligned, n_ref, prefer_match, ctx=ctx)
        sel = match_pair[match_pair >= 0].astype(np.int64)
        sel = sel[np.lexsort((sel, costs_arr[sel]))]
        chosen_pairs = [(int(pairs[idx, 0]), int(pairs[idx, 1]), int(idx)) for idx in sel]
        used_aligned = match_pair >= 0
        chosen_unmatched = set(np.flatnonzero(~used_aligned).tolist())
    else:
        if (n_aligned + n_ref) > int(init_hungarian_max_n):
            if verbose:
                print(f"Skipping Hungarian init: n_aligned+n_ref={n_aligned+n_ref} > init_hungarian_max_n={init_hungarian_max_n}")
            return [], set()
        from scipy.optimize import linear_sum_assignment

        cost_mat = ops.assign_matrix(pairs, costs_arr, unmatched_cost, n_aligned, n_ref, float(init_big_m), ctx=ctx)
        row_ind, col_ind = linear_sum_assignment(cost_mat)
        pair_to_var_idx = {(int(i), int(j)): idx for idx, (i, j) in enumerate(pairs.tolist())}
        used_ref: Set[int] = set()
        for i, col in zip(row_ind, col_ind):
            i, col = int(i), int(col)
            if col < n_ref and cost_mat[i, col] < float(init_big_m) * 0.5:
                if col in used_ref:
                    continue
                used_ref.add(col)
                var_idx = pair_to_var_idx.get((i, col))
                if var_idx is not None:
                    chosen_pairs.append((i, col, int(var_idx)))
            else:
                chosen_unmatched.add(i)
    return chosen_pairs, chosen_unmatched


def apply_mip_start(*, x_vars, no_match_vars, valid_pairs, costs, n_aligned, n_ref, aligned_sizes, no_match_penalty,
                    max_matches, init_method: Optional[str], init_big_m: float = 1e9, init_hungarian_max_n: int = 2000,
                    verbose: bool = True, ctx=None) -> None:
    """src/init_helpers.py:180-244: set .Start on the solver's variables."""
    if init_method is None:
        return
    chosen_pairs, chosen_unmatched = compute_mip_start_pairs(
        valid_pairs=valid_pairs, costs=costs, n_aligned=n_aligned, n_ref=n_ref, aligned_sizes=aligned_sizes,
        no_match_penalty=no_match_penalty, max_matches=max_matches, init_method=init_method, init_big_m=init_big_m,
        init_hungarian_max_n=init_hungarian_max_n, verbose=verbose, ctx=ctx)
    if not chosen_pairs and not chosen_unmatched:
        return
    for var_idx in range(len(valid_pairs)):
        x_vars[var_idx].Start = 0.0
    for i in range(n_aligned):
        no_match_vars[i].Start = 1.0 if i in chosen_unmatched else 0.0
    for i, _j, var_idx in chosen_pairs:
        x_vars[var_idx].Start = 1.0
        no_match_vars[int(i)].Start = 0.0
    if verbose:
        print(f"Initialized MIP start ({str(init_method).lower()}): {len(chosen_pairs)} matches, {len(chosen_unmatched)} unmatched")
