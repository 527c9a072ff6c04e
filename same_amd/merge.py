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


# column-parallel copies of a 10^6-row table (22 columns x 950k rows on a 16-CPU host: 16.9 ms on one thread, 4.5 on four, 1.9 on eight)
GATHER_THREADS = 8


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


def _resolve_rows(a_ids, r_ids, viol, window_id, dedup, seam=None, mark=True):
    """src/helpers.py:745-815 on columns: the de-duplication of the (aligned, ref) pairs, then ONE maximum matching of the pairs that
    survive -> the input rows of the matched pairs, aligned ids ascending (:799-808).

    `seam` (bool per row; the per-rank form, `merged_part_rows`) marks rows whose aligned or reference cell another rank's rows may name
    too.  Such a row, and every row connected to it through shared cells, cannot be decided from this rank's rows alone:
    -> (rows decided here, rows for the step all ranks share).  Rows decided here are final: the graph's connected components are matched
    independently of each other (Hopcroft-Karp's phases only couple them through the length of the shortest augmenting path, and a
    component without a path of that length is left as it is -- tests/test_host_rows.py::test_matching_of_components_equals_the_whole),
    so a component that lies inside one rank gets the matching the single pass over the whole table would give it."""
    from scipy.sparse import coo_matrix, csr_matrix
    from scipy.sparse.csgraph import connected_components, maximum_bipartite_matching

    from contextlib import nullcontext

    from ._trace import stage

    marked = stage if mark else (lambda _name: nullcontext())     # the common seam step is ONE stage of its caller's

    with marked("merge: de-duplication (codes + device)"):
        kept = dedup(np.asarray(viol), _window_codes(window_id), _equality_codes(a_ids), _equality_codes(r_ids)) if len(a_ids) else ()
        kept = np.asarray(kept, dtype=np.int64)     # rows that survive, in the order the reference's frame has after :748-753
    with marked("merge: graph of the surviving pairs"):
        (a_codes, n_a), (r_codes, n_r) = _node_numbers(np.asarray(a_ids)[kept]), _node_numbers(np.asarray(r_ids)[kept])
        # An edge whose two cells have no other edge is in every maximum matching: only the cells that some window disagrees about
        # (an aligned cell proposed for two references, a reference proposed to two aligned cells) need the graph algorithm -- in a
        # tiled run these are the cells of the window overlaps at most.  row_of[a] = the position in `kept` of a's matched edge.
        deg_a, deg_r = np.bincount(a_codes, minlength=n_a), np.bincount(r_codes, minlength=n_r)
        row_of = np.full(n_a, -1, np.int64)
        shared = np.zeros(len(kept), bool) if seam is None else np.asarray(seam, dtype=bool)[kept]
        # nobody disagrees (a tiled run whose overlaps agree): all edges stand
        if len(a_codes) == 0 or (deg_a.max() <= 1 and deg_r.max() <= 1):
            rest = np.zeros(0, np.int64)
            lone_at = np.flatnonzero(~shared) if seam is not None else np.arange(len(a_codes), dtype=np.int64)
        else:
            lone = (deg_a[a_codes] == 1) & (deg_r[r_codes] == 1)
            rest = np.flatnonzero(~lone)
            lone_at = np.flatnonzero(lone & ~shared)
        row_of[a_codes[lone_at]] = lone_at
        graph = None
        if len(rest):
            # the contested cells renumbered densely IN THE ORDER of their ids (node numbers come from the sorted ids, and every
            # adjacency list is sorted below): the matching chosen among equally large ones depends on the ids alone, not on the
            # order the window tables arrived in (1 rank or 8)
            a_rest, r_rest = a_codes[rest], r_codes[rest]
            a_new, r_new = np.cumsum(_present(a_rest, n_a)) - 1, np.cumsum(_present(r_rest, n_r)) - 1
            ai, ri = a_new[a_rest], r_new[r_rest]
            n1, n2 = int(a_new[-1]) + 1, int(r_new[-1]) + 1
            if seam is not None and shared.any():
                # contested rows whose component holds a shared row go to the common step with it; the rest is matched here
                _n, label = connected_components(coo_matrix((np.ones(len(ai), np.int8), (ai, n1 + ri)), shape=(n1 + n2, n1 + n2)),
                                                 directed=False)
                tainted = np.zeros(_n, bool)
                tainted[label[ai[shared[rest]]]] = True
                away = tainted[label[ai]]
                shared[rest[away]] = True
                rest, ai, ri = rest[~away], ai[~away], ri[~away]
            # every edge carries its position in `kept` (+1: an explicit zero would be dropped), so the rows of the matched edges
            # can be read off the matrix afterwards without a second sort; edges are unique after the de-duplication: nothing is summed
            if len(rest):
                graph = csr_matrix((rest + 1, (ai, ri)), shape=(n1, n2))
                graph.sort_indices()
    with marked("merge: maximum matching"):
        if graph is not None:
            # ref node matched to each aligned node, -1 = none (structure only)
            match_r = maximum_bipartite_matching(graph, perm_type="column")
    with marked("merge: rows of the matched pairs"):
        if graph is not None:
            node_of_edge = np.repeat(np.arange(graph.shape[0], dtype=np.int64), np.diff(graph.indptr))
            won = graph.data[match_r[node_of_edge] == graph.indices] - 1       # positions in `kept` of the contested edges that are matched
            row_of[a_codes[won]] = won
        selected = kept[row_of[row_of >= 0]]                                   # aligned ids ascending (:799-808)
    return selected if seam is None else (selected, kept[shared])


def _device_dedup(ctx=None):
    """the device step (ops.merge_dedup on `ctx`); there is no host substitute in the product: a missing GPU raises SameHipError"""
    from . import ops

    if ctx is None:
        return ops.merge_dedup
    return lambda viol, window_id, a_code, r_code: ops.merge_dedup(viol, window_id, a_code, r_code, ctx=ctx)


def merge_window_matches_unique_ref(matches_list, cell_id_col="Cell_Num_Old", _dedup=None):
    """src/helpers.py:692-815.  `_dedup(viol, window_id, aligned_code, ref_code) -> surviving row indices` defaults to the HIP
    kernel chain (`ops.merge_dedup`); the CPU tests pass the oracle's restatement of the same step."""
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
        merged_df["filtered_violation"] = _violation_flags(merged_df["filtered_violation"])
    # one row per (aligned, ref) pair: non-violating first, then the smaller window id, then the earlier row (:748-753);
    # the ids may be anything hashable, the device sees integer codes of them (equal id <=> equal code)
    rows = _resolve_rows(merged_df[aligned_col].values, merged_df[ref_col].values, merged_df["filtered_violation"].to_numpy(),
                         merged_df["window_id"].to_numpy(), _dedup or _device_dedup())
    with marked("merge: rows of the matched pairs"):
        return _take_rows(merged_df, rows)                                     # ONE gather of the frame: the rows of the matched edges


def _violation_flags(fv):
    """`.fillna(True).astype(bool)` of src/helpers.py:746-751 on the column's values (the pandas call warns about its own downcast):
    missing -> True (worst).  Extension dtypes (a nullable 'boolean' column holds pd.NA, whose truth value is undefined) are masked
    before they are cast."""
    if fv.dtype == bool:
        return fv.to_numpy()
    v = fv.to_numpy(dtype=object) if not isinstance(fv.dtype, np.dtype) else fv.to_numpy()
    na = np.asarray(pd.isna(v), dtype=bool)
    out = np.ones(len(v), bool)
    out[~na] = v[~na].astype(bool)
    return out


# ---- the merge over ranks: every rank decides what only it can see, one small common step decides the seams ----------------------
def seam_rows(pos, coords, plan, owner, me, reach):
    """bool per row of this rank's table: may another rank's table name the row's aligned or reference cell?

    A row of window p has its aligned cell inside p's central trim (src/same.py:565-582) and its reference cell within `reach` of it
    (the prune's radius; any bound on max(|X - ref_X|, |Y - ref_Y|) over ALL ranks' rows).  Another rank's window q can name the same
    aligned cell only if the cell lies in q's trim, the same reference cell only if that lies within `reach` of q's trim: two box
    tests per foreign window close enough to p to matter (none for a window in the middle of this rank's block).  Valid where a
    cell id names ONE row of its frame (callers check; otherwise every row is a seam row).  pos: plan position per row (< 0: unknown
    -> seam), rows of one window adjacent; coords(b, e) -> (X, Y, ref_X, ref_Y) of rows b..e (only asked for windows near a border)."""
    n = len(pos)
    seam = np.zeros(n, bool)
    owner = np.asarray(owner)
    foreign = np.flatnonzero(owner != me)
    if n == 0 or len(foreign) == 0:
        return seam
    pos = np.asarray(pos, dtype=np.int64)
    seam[pos < 0] = True
    trims = np.array([w["trim"] for w in plan], dtype=np.float64).reshape(-1, 4)
    ft = trims[foreign]
    cut = np.flatnonzero(np.diff(pos)) + 1
    begins, ends = np.concatenate(([0], cut)), np.concatenate((cut, [n]))
    for b, e in zip(begins.tolist(), ends.tolist()):
        p = int(pos[b])
        if p < 0:
            continue
        t = trims[p]
        near = ft[(ft[:, 0] - reach <= t[1] + reach) & (ft[:, 1] + reach >= t[0] - reach)
                  & (ft[:, 2] - reach <= t[3] + reach) & (ft[:, 3] + reach >= t[2] - reach)]
        if not len(near):
            continue
        x, y, u, v = coords(b, e)
        hit = seam[b:e]
        for x0, x1, y0, y1 in near.tolist():
            hit |= (x >= x0) & (x < x1) & (y >= y0) & (y < y1)
            hit |= (u >= x0 - reach) & (u <= x1 + reach) & (v >= y0 - reach) & (v <= y1 + reach)
    return seam


def seam_tables(plan, owner, me, reach):
    """`seam_rows` as tables for the device (same_merge_acc_begin): per plan position of rank `me` the central regions of OTHER ranks'
    windows that lie within 2 * reach of its own -> (near_start int32[len(plan) + 1], near_boxes float64[.., 4])."""
    owner = np.asarray(owner)
    trims = np.array([w["trim"] for w in plan], dtype=np.float64).reshape(-1, 4)
    foreign = np.flatnonzero(owner != me)
    ft = trims[foreign]
    start, boxes = np.zeros(len(plan) + 1, np.int32), []
    for p in range(len(plan)):
        if owner[p] == me and len(ft):
            t = trims[p]
            near = ft[(ft[:, 0] - reach <= t[1] + reach) & (ft[:, 1] + reach >= t[0] - reach)
                      & (ft[:, 2] - reach <= t[3] + reach) & (ft[:, 3] + reach >= t[2] - reach)]
            boxes.append(near)
            start[p + 1] = start[p] + len(near)
        else:
            start[p + 1] = start[p]
    return start, (np.concatenate(boxes) if boxes else np.zeros((0, 4)))


def already_deduplicated(viol, window_id, a_codes, r_codes):
    """the `dedup` of rows that are one per (aligned, ref) pair already (the device accumulator's REST rows): all of them, in order"""
    return np.arange(len(a_codes), dtype=np.int64)


def merged_part_rows(a_ids, r_ids, viol, window_id, pos, seam, rank=0, exchange=None, _dedup=None):
    """The window merge (src/helpers.py:692-815) of a table that is dealt over ranks, as seen by ONE rank: `a_ids` ... `pos` are the
    columns of this rank's rows (its windows in plan order, rows in window order; `pos` = plan position of the row's window), `seam`
    what `seam_rows` says of them.  -> the rows of this rank that are in the merged table, aligned ids ascending.

    Every rank de-duplicates and matches what only it can see (the de-duplication's choice -- not violating, then the smaller window id,
    then the earlier row -- is a minimum, hence associative); only seam rows and the rows connected to them travel: `exchange(table) ->
    [table of rank 0, ...]` (dist.allgather_table: one small all-gather) and every rank runs the same step on the same gathered rows.
    The ranks' parts laid together and ordered by aligned id are the single process's merged table, row for row."""
    from ._trace import stage as marked

    dedup = _dedup or _device_dedup()
    a_ids, r_ids = np.asarray(a_ids), np.asarray(r_ids)
    viol, window_id = np.asarray(viol, dtype=bool), np.asarray(window_id)
    if exchange is None:
        return _resolve_rows(a_ids, r_ids, viol, window_id, dedup)
    mine, sent = part_decided_here(a_ids, r_ids, viol, window_id, pos, seam, rank, dedup)
    with marked("merge: seam rows exchanged"):
        parts = exchange(sent)
    with marked("merge: seam step (the same on every rank)"):
        return part_after_seam_step(a_ids, mine, parts, rank, dedup)


def part_decided_here(a_ids, r_ids, viol, window_id, pos, seam, rank, dedup, seq=None):
    """First half of `merged_part_rows`: -> (rows of this rank decided from its own rows, the table of rows it sends to the common step).
    seq: a row's place in its window's table (default: its place in this rank's table, which is in window order)."""
    mine, common = _resolve_rows(a_ids, r_ids, viol, window_id, dedup, seam=seam)
    # `order`: the place a row has in the single process's concatenation -- plan position, then the row's place in its window's table
    place = common if seq is None else np.asarray(seq, dtype=np.int64)[common]
    sent = {"a": a_ids[common], "r": r_ids[common], "viol": viol[common].view(np.uint8), "window": window_id[common],
            "order": (np.asarray(pos, dtype=np.int64)[common] << 32) | place, "row": common, "rank": np.full(len(common), rank, np.int32)}
    return mine, sent


def part_after_seam_step(a_ids, mine, parts, rank, dedup):
    """Second half: the common step on every rank's sent rows (identical on every rank) -> this rank's rows, aligned ids ascending."""
    parts = [p for p in parts if len(p["row"])]
    if not parts:
        return mine
    col = lambda c: np.concatenate([p[c] for p in parts])
    order = np.argsort(col("order"), kind="stable")
    won = order[_resolve_rows(col("a")[order], col("r")[order], col("viol")[order].view(bool), col("window")[order], dedup, mark=False)]
    won = won[col("rank")[won] == rank]
    mine = np.concatenate((mine, col("row")[won].astype(np.int64)))
    return mine if a_ids is None else mine[np.argsort(a_ids[mine], kind="stable")]


def join_merged_parts(parts, cell_id_col="Cell_Num_Old"):
    """The ranks' parts of a merged table (each ordered by aligned id) -> the single process's table."""
    parts = [p for p in parts if p is not None and len(p)]
    if not parts:
        return pd.DataFrame()
    whole = pd.concat(parts, ignore_index=True)
    return _take_rows(whole, np.argsort(whole[f"Aligned_{cell_id_col}"].to_numpy(), kind="stable"))


def merge_table_part(table, plan, owner, channel, cell_id_col="Cell_Num_Old", reach=None, ids_unique=True, id_codes=None, _dedup=None):
    """`merged_part_rows` for a rank's pre-merge table as the window loop leaves it (sliding_window_matching / sliding_window_incumbent with
    `_shard`: the columns of src/same.py:1264-1278 + window_id + `__plan_pos`) -> this rank's part of the merged table, aligned ids
    ascending, `__plan_pos` dropped.  `channel` (dist.MergeChannel; None = one process: the whole merge) carries the seam rows; `reach`
    None measures the bound seam_rows needs (the largest coordinate distance between a row's two cells, over all ranks).
    id_codes(aligned ids, ref ids) -> (codes, codes): integer ranks of the ids in the ids' order, the SAME on every rank (window_api.
    frame_id_codes): the merge then compares and exchanges the codes -- one numeric all-gather whatever the ids are."""
    from ._trace import stage as marked

    aligned_col, ref_col = f"Aligned_{cell_id_col}", f"Ref_{cell_id_col}"
    if table is None or len(table) == 0:
        table = pd.DataFrame({c: np.zeros(0, dt) for c, dt in ((aligned_col, np.int64), (ref_col, np.int64), ("X", float), ("Y", float),
                                                               ("ref_X", float), ("ref_Y", float), ("filtered_violation", bool),
                                                               ("window_id", np.int64), ("__plan_pos", np.int64))})
    missing = [c for c in ["window_id", aligned_col, ref_col, "X", "Y", "filtered_violation"] if c not in table.columns]
    if missing:
        raise ValueError(f"Missing required columns in matches: {missing}")
    viol = _violation_flags(table["filtered_violation"])
    a_ids, r_ids, wid = table[aligned_col].to_numpy(), table[ref_col].to_numpy(), table["window_id"].to_numpy()
    if id_codes is not None:
        a_ids, r_ids = id_codes(a_ids, r_ids)
    keep = [c for c in table.columns if c != "__plan_pos"]
    if channel is None or channel.world == 1:
        rows = merged_part_rows(a_ids, r_ids, viol, wid, None, None, _dedup=_dedup)
    else:
        with marked("merge: seam rows marked"):
            pos = table["__plan_pos"].to_numpy() if "__plan_pos" in table.columns else np.full(len(table), -1, np.int64)
            pos = np.where(pd.isna(pos), -1, pos).astype(np.int64)
            if ids_unique and all(c in table.columns for c in ("ref_X", "ref_Y")):
                x, y, u, v = (table[c].to_numpy(dtype=np.float64) for c in ("X", "Y", "ref_X", "ref_Y"))
                if reach is None:
                    far = max(float(np.abs(x - u).max()), float(np.abs(y - v).max())) if len(x) else 0.0
                    reach = channel.max(far if far == far else np.inf)
                seam = (seam_rows(pos, lambda b, e: (x[b:e], y[b:e], u[b:e], v[b:e]), plan, owner, channel.rank, reach)
                        if np.isfinite(reach) else np.ones(len(table), bool))
            else:
                seam = np.ones(len(table), bool)        # nothing to reason from: every row goes to the common step (correct, not scalable)
        rows = merged_part_rows(a_ids, r_ids, viol, wid, pos, seam, channel.rank, channel.tables, _dedup=_dedup)
    with marked("merge: rows of the matched pairs"):
        out = _take_rows(table[keep] if len(keep) != len(table.columns) else table, rows)
        out["filtered_violation"] = viol[rows]
    return out
