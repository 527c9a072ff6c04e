"""eval_utils.check_triangle_violations (src/eval_utils.py:66-223), same signature: triangle
orientation flips after alignment, and which nodes sit in flipped triangles (optionally by a
node-local majority rule).  The triangle loop runs in csrc/match.hip (tri_flip_stats_kernel);
id bookkeeping and the per-node rule are host work on small arrays."""
import numpy as np
import pandas as pd

from . import ops


def check_triangle_violations(outputDF, mc_align, aligned_id_col="aligned_metacell_index",
                              ref_id_col="matched_ref_index", mapped_x_col="mapped_x", mapped_y_col="mapped_y",
                              cell_type_col="cell_type", ignore_same_type_triangles=True, node_local=False,
                              majority_threshold=0.5, min_flips=1, verbose=False, ctx=None):
    outputDF = outputDF.copy()
    tri_ids = np.asarray(mc_align.metacell_delaunay)
    tri_ids = tri_ids.reshape(-1, 3) if tri_ids.size else np.zeros((0, 3), dtype=np.int64)
    mdf = mc_align.metacell_df
    n = len(mdf)
    node_index = pd.Index(mdf.index)
    # node = position in metacell_df; ids absent from it cannot be processed (the reference's `except: continue`)
    tri_pos = node_index.get_indexer(tri_ids.reshape(-1)).reshape(-1, 3)
    has_node = (tri_pos >= 0).all(axis=1)
    # rows of outputDF per aligned id; the dict comprehensions of the reference keep the LAST row per id
    ids = outputDF[aligned_id_col].to_numpy()
    out_pos = node_index.get_indexer(ids)
    matched = np.zeros(n, np.uint8)
    last_row = np.full(n, -1, np.int64)
    ok = out_pos >= 0
    last_row[out_pos[ok]] = np.flatnonzero(ok)
    matched[out_pos[ok]] = 1
    mapped = np.zeros((n, 2))
    rows = last_row[last_row >= 0]
    mapped[last_row >= 0, 0] = outputDF[mapped_x_col].to_numpy(dtype=np.float64)[rows]
    mapped[last_row >= 0, 1] = outputDF[mapped_y_col].to_numpy(dtype=np.float64)[rows]
    type_id = None
    if ignore_same_type_triangles:
        codes = pd.factorize(outputDF[cell_type_col].to_numpy(), use_na_sentinel=False)[0].astype(np.int32)
        type_id = np.full(n, -1, np.int32)
        type_id[last_row >= 0] = codes[rows]
    axy = mdf[["X", "Y"]].to_numpy(dtype=np.float64)

    tris = tri_pos[has_node].astype(np.int32)
    flag, node_tri, node_flip = ops.tri_flip_stats(axy, mapped, matched, tris, type_id, ctx=ctx)
    all_matched = (flag & 1).astype(bool)
    same = (flag & 2).astype(bool)
    flipped = (flag & 4).astype(bool)
    # triangles whose three ids are in outputDF but not all in metacell_df: counted, then skipped
    lost = ~has_node
    lost_matched = 0
    lost_same = 0
    if lost.any():
        full = np.isin(tri_ids[lost], ids).all(axis=1)
        lost_matched = int(full.sum())
        if ignore_same_type_triangles and lost_matched:
            ct = outputDF[cell_type_col].to_numpy()
            lastrow_of = {v: r for r, v in enumerate(ids)}
            for tri in tri_ids[lost][full]:
                t0, t1, t2 = (ct[lastrow_of[v]] for v in tri)
                lost_same += int(t0 == t1 == t2)

    considered = all_matched & ~same
    sign_flips = flipped[considered]
    unique_ids = outputDF[aligned_id_col].unique()
    upos = node_index.get_indexer(unique_ids)
    n_tri = np.where(upos >= 0, node_tri[np.maximum(upos, 0)], 0).astype(np.int64)
    n_flip = np.where(upos >= 0, node_flip[np.maximum(upos, 0)], 0).astype(np.int64)
    if node_local:
        with np.errstate(divide="ignore", invalid="ignore"):
            frac = n_flip / n_tri
        viol = (n_tri > 0) & (n_flip >= min_flips) & (frac >= majority_threshold)
    else:
        viol = n_flip > 0
    node_in_violating_triangle = dict(zip(unique_ids.tolist(), viol.tolist()))
    outputDF["in_violating_triangle"] = outputDF[aligned_id_col].map(node_in_violating_triangle).fillna(False)
    stats = {
        "total_triangles": len(tri_ids),
        "triangles_with_all_matched": int(all_matched.sum()) + lost_matched,
        "triangles_processed": int(all_matched.sum()) + lost_matched,
        "triangles_same_type_skipped": int(same.sum()) + lost_same,
        "triangles_flipped": int(np.sum(sign_flips)) if len(sign_flips) else 0,
        "percent_flipped": (100.0 * np.sum(sign_flips) / len(sign_flips) if len(sign_flips) else 0.0),
        "nodes_in_violating_triangles": int(outputDF["in_violating_triangle"].sum()),
        "percent_nodes_violating": 100.0 * outputDF["in_violating_triangle"].mean(),
    }
    if verbose:
        print(stats)
    return outputDF, stats


def check_alignment(queryDF, templateDF, xcol, ycol, ctype_col="cell_type", kNN=1, ctx=None):
    """eval_utils.check_alignment (src/eval_utils.py:6-55), same signature: is the query cell's type among the
    types of its kNN nearest template cells?  cKDTree.query(k) = the k nearest by Euclidean distance; here the
    radius-free form of the prune kernel (radius = +inf, ranking by (d2, template index))."""
    queryDF = queryDF.copy()
    required = {xcol, ycol, ctype_col}
    if not required.issubset(queryDF.columns) or not required.issubset(templateDF.columns):
        raise ValueError(f"Both DataFrames must contain the columns: {required}")
    q = queryDF[[xcol, ycol]].to_numpy(dtype=np.float64)
    t = templateDF[[xcol, ycol]].to_numpy(dtype=np.float64)
    idx, _, _ = ops.knn_prune(q, t, float("inf"), int(kNN), want_d2=False, ctx=ctx)
    qt = queryDF[ctype_col].to_numpy()
    tt = templateDF[ctype_col].to_numpy()
    near_types = tt[np.maximum(idx, 0)]
    col = "_" + str(kNN) + "NN_match"
    queryDF.loc[:, col] = ((near_types == qt[:, None]) & (idx >= 0)).any(axis=1)
    if kNN == 1:
        queryDF.loc[:, "_" + str(kNN) + "NN_match_ctype"] = near_types[:, 0]
    return queryDF, queryDF[col].mean()
