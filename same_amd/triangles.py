"""Triangle pipeline of run_same: normalise / remap (src/same.py:245-290), filter
(helpers.filter_triangles_by_radius, src/helpers.py:233-395), weights + source signs
(src/same.py:1128-1146), simplex map and triangle_info (src/same.py:1096-1099,
src/helpers.py:184-210).  Per-triangle arithmetic runs in csrc/tri.hip; the host keeps the
order-dependent bookkeeping (the same-type re-add pass, dict-shaped outputs)."""
import numpy as np
import pandas as pd

from . import ops
from ._rows import RowList, ValueList, rows_array


# ----------------------------------------------------------------------------- a6 (host)
def _as_triangle_array(delaunay_like):
    """Whatever a caller hands in as a triangulation -> integer rows (n, 3), or None for None.  Same contract as src/same.py:245-259:
    a frame gives its first three columns, an empty input an empty (0, 3) array, any other shape the reference's ValueError."""
    if delaunay_like is None:
        return None
    frame = isinstance(delaunay_like, pd.DataFrame)
    rows = np.asarray(delaunay_like.iloc[:, :3] if frame else delaunay_like)
    if rows.size == 0:
        return np.empty((0, 3), dtype=int)
    if rows.ndim == 2 and rows.shape[1] == 3:
        return rows.astype(int, copy=False)
    raise ValueError(f"aligned_delaunay must have shape (n, 3); got {rows.shape}")


def _remap_triangles_by_vertex_ids(triangles, vertex_ids):
    """src/same.py:262-290: vertex-id space -> row indices, dropping triangles with a missing vertex."""
    tri = _as_triangle_array(triangles)
    if tri is None:
        return None
    if tri.size == 0:
        return tri
    vertex_ids = np.asarray(vertex_ids)
    # dict semantics of the reference: for duplicate ids the LAST row wins
    uniq, first_of_reversed = np.unique(vertex_ids[::-1], return_index=True)
    last_row = len(vertex_ids) - 1 - first_of_reversed
    flat = tri.reshape(-1)
    pos = np.searchsorted(uniq, flat)
    pos_c = np.clip(pos, 0, max(len(uniq) - 1, 0))
    hit = (uniq[pos_c] == flat) if len(uniq) else np.zeros(len(flat), bool)
    remapped = np.where(hit, last_row[pos_c] if len(uniq) else -1, -1).astype(int).reshape(tri.shape)
    return remapped[(remapped >= 0).all(axis=1)]


# ----------------------------------------------------------------------------- a7
def _ordered_key(c):
    u = int(np.float64(c).view(np.int64))
    return u if u >= 0 else -(u & 0x7FFFFFFFFFFFFFFF)


def _from_key(k):
    return float(np.int64(k).view(np.float64)) if k >= 0 else -float(np.int64(-k).view(np.float64))


def _angle_fails(c, min_angle_deg):
    # the reference's own expression on a clipped cosine (src/helpers.py:287-288, :319)
    return bool(np.degrees(np.arccos(c)) < min_angle_deg)


_thr_cache = {}


def cos_threshold(min_angle_deg):
    """(enabled, thr): `degrees(arccos(c)) < min_angle_deg`  <=>  `c >= thr` for clipped cosines.

    arccos/degrees are non-increasing in c, so the failing cosines form an upper interval of the
    double lattice; its lower end is found by bisection with numpy's own arccos/degrees, which
    lets the kernel decide the angle rule with a compare and no device-side arccos."""
    if min_angle_deg is None:
        return 0, float("inf")
    key = float(min_angle_deg)
    if key in _thr_cache:
        return _thr_cache[key]
    if not _angle_fails(1.0, min_angle_deg):
        res = (1, float("inf"))
    elif _angle_fails(-1.0, min_angle_deg):
        res = (1, float("-inf"))
    else:
        lo, hi = _ordered_key(-1.0), _ordered_key(1.0)
        while hi - lo > 1:
            mid = (lo + hi) // 2
            if _angle_fails(_from_key(mid), min_angle_deg):
                hi = mid
            else:
                lo = mid
        res = (1, _from_key(hi))
    _thr_cache[key] = res
    return res


def classify_triangles(points, triangles, radius, min_angle_deg, type_id=None, ctx=None):
    """Per-triangle class (0 keep, 1 radius, 2 angle, 3 same type) + perimeter, from the kernel.

    Cosines within a few ulp of the threshold are re-decided on the host with the reference's
    literal arccos/degrees expression, so a libm that is not perfectly monotone cannot flip a
    knife-edge triangle."""
    en, thr = cos_threshold(min_angle_deg)
    tris = np.asarray(triangles).reshape(-1, 3)
    cls, perim, maxcos = ops.tri_classify(points, tris, radius, en, thr, type_id, ctx=ctx)
    if en and np.isfinite(thr) and len(cls):
        near = np.flatnonzero((np.abs(maxcos - thr) <= 8 * np.spacing(abs(thr))) & (cls != 1))
        for t in near:
            fails = _angle_fails(maxcos[t], min_angle_deg)
            a, b, c = tris[t]
            if fails:
                cls[t] = 2
            else:
                cls[t] = 3 if (type_id is not None and type_id[a] == type_id[b] == type_id[c]) else 0
    return cls, perim


def filter_triangles_by_radius(points, triangles, radius, aligned_df=None, ignore_same_type_triangles=False,
                               ensure_min_triangle_per_node=True, remove_unconstrained_nodes=False,
                               min_angle_deg=15, verbose=True, ctx=None, _rows_as_array=False, _type_id=None):
    """Same signature and return shapes as src/helpers.py:233-395 (a list of kept triangle rows [+ set]).
    `_rows_as_array` (package-internal): hand back the (n, 3) array the list would be made of, so a caller that only
    feeds it to kernels does not pay for 10^5 little row objects.  `_type_id` (package-internal): integer codes of the
    points' cell types (equal type <=> equal code) in place of `aligned_df["cell_type"]`, for callers without a frame."""
    points = np.asarray(points)
    tris = rows_array(triangles)
    use_type = bool(ignore_same_type_triangles and (aligned_df is not None or _type_id is not None))
    type_id = None
    if use_type:
        type_id = (np.ascontiguousarray(_type_id, dtype=np.int32) if _type_id is not None
                   else pd.factorize(aligned_df["cell_type"].to_numpy(), use_na_sentinel=False)[0].astype(np.int32))
    cls, perim = classify_triangles(points, tris, radius, min_angle_deg, type_id, ctx=ctx)

    keep_idx = np.flatnonzero(cls == 0)
    n_points = len(points)
    has_kept = np.zeros(n_points, bool)
    has_kept[tris[keep_idx].reshape(-1)] = True
    any_valid = np.zeros(n_points, bool)
    any_valid[tris[(cls == 0) | (cls == 3)].reshape(-1)] = True
    order = keep_idx                          # triangle indices kept, in the order the reference's list has them

    if verbose:
        print("\nTriangle filtering summary:")
        print(f"Total triangles: {len(tris)}")
        print(f"Triangles skipped (radius): {int((cls == 1).sum())}")
        if min_angle_deg is not None:
            print(f"Triangles skipped (min_angle < {min_angle_deg}°): {int((cls == 2).sum())}")
        if ignore_same_type_triangles:
            print(f"Triangles skipped (same type): {int((cls == 3).sum())}")
        print(f"Triangles kept: {len(order)}")

    if use_type and ensure_min_triangle_per_node:
        # src/helpers.py:331-340 + :365-389.  best same-type triangle per vertex = smallest perimeter,
        # first one in input order on ties (strict <): a stable argsort by perimeter gives exactly that.
        same = np.flatnonzero(cls == 3)
        if len(same):
            cand_v = tris[same].reshape(-1)
            cand_t = np.repeat(same, 3)
            o = np.lexsort((cand_t, perim[cand_t]))       # by perimeter, then input order
            v_sorted, t_sorted = cand_v[o], cand_t[o]
            first = np.unique(v_sorted, return_index=True)[1]
            best_t = np.full(n_points, -1, np.int64)                 # best same-type triangle of every vertex that has one
            best_t[v_sorted[first]] = t_sorted[first]
            missing = np.flatnonzero(~has_kept & any_valid)
            cand = best_t[missing]
            cand = cand[cand >= 0]                                     # in ascending node order, as the reference walks them
            # the reference de-duplicates by the triangle's vertex tuple against every triangle kept so far (src/helpers.py:375-381);
            # a candidate contains a node that no kept triangle contains, so it can only collide with a triangle added in this
            # pass: keep the first occurrence of every distinct vertex row, in walk order
            if len(cand):
                _, first_of_row = np.unique(tris[cand], axis=0, return_index=True)
                cand = cand[np.sort(first_of_row)]
            order = np.concatenate((keep_idx, cand))
            added = len(cand)
            if verbose and added:
                print(f"Added back {added} same-type triangles to ensure >=1 triangle per node")
                print(f"Final triangles kept: {len(order)}")

    if _rows_as_array:
        filtered = tris[order] if len(order) else np.zeros((0, 3), dtype=tris.dtype)
    else:
        filtered = RowList(tris[order]) if len(order) else []   # a list of triangle rows, as the reference returns
    if remove_unconstrained_nodes:
        return filtered, set(np.flatnonzero(~any_valid).tolist())
    return filtered


# ----------------------------------------------------------------------------- a8
def triangle_weights_and_signs(aligned_df, triangles, ctx=None, _as_arrays=False):
    """-> (list of weights, list of np.float64 signs) as run_same builds them (src/same.py:1128-1146);
    `_as_arrays` (package-internal): the two arrays instead."""
    xy = aligned_df[["X", "Y"]].to_numpy(dtype=np.float64)
    size = aligned_df["size"].to_numpy(dtype=np.float64)
    sign, weight = ops.tri_sign_weight(xy, size, rows_array(triangles), ctx=ctx)
    size_dtype = aligned_df["size"].dtype
    if np.issubdtype(size_dtype, np.integer):
        weight = weight.astype(np.int64)  # integer size columns sum to integers in the reference
    if _as_arrays:
        return weight, sign.astype(np.float64)
    return ValueList(weight), ValueList(sign.astype(np.float64))


# ----------------------------------------------------------------------------- a9 (host, flat arrays -> dicts)
def build_simplex_map(n_aligned, triangles):
    """src/same.py:1096-1099."""
    m = {i: set() for i in range(n_aligned)}
    for idx, simplex in enumerate(triangles):
        for i in simplex:
            m[i].add(idx)
    return m


def precompute_triangle_info(aligned_df, aligned_delaunay, aligned_simplex_map):
    """src/helpers.py:184-210.  Insertion order (vertices ascending, then each vertex's set order)
    is part of the contract: verify_spatial_preservation and var_out iterate it."""
    xy = aligned_df[["X", "Y"]].to_numpy(dtype=np.float64)
    X, Y = xy[:, 0], xy[:, 1]
    info = {}
    for ip in range(len(aligned_df)):
        for s in aligned_simplex_map[ip]:
            if s not in info:
                simplex = aligned_delaunay[s]
                xs = [X[i] for i in simplex]
                ys = [Y[i] for i in simplex]
                min_x, max_x, min_y, max_y = min(xs), max(xs), min(ys), max(ys)
                info[s] = {
                    "vertices": simplex,
                    "bounds": {"min_x": min_x, "max_x": max_x, "min_y": min_y, "max_y": max_y},
                    "max_x_vertex": simplex[xs.index(max_x)],
                    "min_x_vertex": simplex[xs.index(min_x)],
                    "max_y_vertex": simplex[ys.index(max_y)],
                    "min_y_vertex": simplex[ys.index(min_y)],
                }
    return info


def precompute_coordinate_maps(aligned_df, ref_df, valid_pairs):
    """src/helpers.py:164-181."""
    from collections import defaultdict

    axy = aligned_df[["X", "Y"]].to_numpy(dtype=np.float64)
    rxy = ref_df[["X", "Y"]].to_numpy(dtype=np.float64)
    aligned_coords = {i: {"X": axy[i, 0], "Y": axy[i, 1]} for i in range(len(axy))}
    ref_coords = {i: {"X": rxy[i, 0], "Y": rxy[i, 1]} for i in range(len(rxy))}
    valid_pairs_map = defaultdict(list)
    for idx, (ip, jp) in enumerate(valid_pairs):
        valid_pairs_map[ip].append((idx, jp))
    return aligned_coords, ref_coords, valid_pairs_map
