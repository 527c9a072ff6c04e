"""Metacell creation, SURVEY 8(f2): `MetaCell` and `greedy_triangle_collapse` with the signatures of
src/metacell_utils.py:25-157 and :160-561.

Each collapse iteration is: Delaunay of the current metacells (scipy/Qhull on the host, as in the
reference) -> per-triangle validity / same-type / size / perimeter (csrc/match.hip
`collapse_candidates_kernel`) -> vertex-disjoint greedy selection in perimeter order
(`same_greedy_disjoint`, the device form of the reference's sort + scan) -> merge the selected triples.
The merge keeps the reference's arithmetic: centroids and numeric columns are true means over the
ORIGINAL member cells (pandas `mean` = numpy pairwise sum / count), computed here for all merged
metacells of equal member count at once with the reduced axis contiguous, which is bit-identical to
the per-metacell calls.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional

import numpy as np
import pandas as pd

from . import _lib, ops
from .triangles import cos_threshold


_EMPTY_TRI = (0, 3)
_EMPTY_TRI_XY = (0, 3, 2)


class MetaCell:
    """What a collapse returns when `return_object=True`: the input frame, the metacell frame, both triangulations and
    the column names that tie them together.  The path itself only reads `metacell_df`, `metacell_delaunay`,
    `metacell_idx_col` and `cell_type_col` (duck-typed in run_same / sliding_window_matching); the remaining members
    are the reference container's conveniences (src/metacell_utils.py:25-157), kept under the same names and pinned
    against the reference object by tests/golden/run_same_mock.npz (`mc_helpers/*`)."""

    FIELDS = ("original_df", "params", "x_col", "y_col", "cell_type_col", "original_idx_col", "metacell_idx_col",
              "original_delaunay", "metacell_df", "metacell_delaunay")

    def __init__(self, *args, **fields):
        fields.update(zip(self.FIELDS, args))
        lacking = [f for f in self.FIELDS if f not in fields]
        extra = [f for f in fields if f not in self.FIELDS]
        if lacking or extra or len(args) > len(self.FIELDS):
            raise TypeError(f"MetaCell takes exactly the fields {self.FIELDS}" + (f"; missing {lacking}" if lacking else "")
                            + (f"; unknown {extra}" if extra else ""))
        for name in self.FIELDS:
            setattr(self, name, fields[name])

    def __repr__(self):
        return f"MetaCell({len(self.original_df)} cells -> {len(self.metacell_df)} metacells, idx col {self.metacell_idx_col!r})"

    # -- id space -> row space -------------------------------------------------------------------------------------
    def _vertex_rows(self, triangles, on_missing):
        """Rows of original_df for triangles given in original_idx_col ids; triangles with an unknown id are dropped
        (or reported, on_missing='error').  One vectorised lookup for the whole list."""
        tri = np.asarray(self.original_delaunay if triangles is None else triangles)
        if tri.size == 0:
            return np.empty(_EMPTY_TRI, dtype=int)
        if tri.ndim != 2 or tri.shape[1] != 3:
            raise ValueError(f"triangles must have shape (n, 3); got {tri.shape}")
        lookup = pd.Index(self.original_df[self.original_idx_col].to_numpy())
        rows = lookup.get_indexer(tri.ravel()).astype(int).reshape(-1, 3)
        known = rows >= 0
        if known.all():
            return rows
        if on_missing == "error":
            unknown = sorted(set(tri[~known].tolist()), key=repr)
            raise KeyError(f"Found triangle vertices not in original_df[{self.original_idx_col}]: {unknown[:10]}")
        return rows[known.all(axis=1)]

    def original_delaunay_to_row_indices(self, triangles: Optional[np.ndarray] = None, *, on_missing: str = "drop") -> np.ndarray:
        return self._vertex_rows(triangles, on_missing)

    original_delaunay_to_pos = original_delaunay_to_row_indices   # the reference offers both names for the same map

    # -- coordinates of triangle corners -----------------------------------------------------------------------------
    def _corner_xy(self, frame, rows):
        if rows.size == 0:
            return np.empty(_EMPTY_TRI_XY, dtype=float)
        return frame[[self.x_col, self.y_col]].to_numpy(dtype=float)[rows]

    def original_delaunay_to_xy(self, triangles: Optional[np.ndarray] = None, *, on_missing: str = "drop") -> np.ndarray:
        return self._corner_xy(self.original_df, self._vertex_rows(triangles, on_missing))

    def metacell_delaunay_to_xy(self) -> np.ndarray:
        return self._corner_xy(self.metacell_df, np.asarray(self.metacell_delaunay).astype(int, copy=False))

    # -- bookkeeping ---------------------------------------------------------------------------------------------------
    def metacell_members(self, metacell_idx: int) -> List[Any]:
        return list(self.metacell_df["members"].iloc[int(metacell_idx)])

    def to_summary_dict(self) -> Dict[str, Any]:
        def n_tri(t):
            return int(getattr(t, "shape", [0])[0])

        out = {"n_original": len(self.original_df), "n_metacells": len(self.metacell_df), "params": dict(self.params)}
        out.update({name: getattr(self, name) for name in self.FIELDS if name.endswith("_col")})
        out.update(n_original_triangles=n_tri(self.original_delaunay), n_metacell_triangles=n_tri(self.metacell_delaunay))
        return out


_WELL_KNOWN_ID_COLUMNS = ("Cell_Num", "Cell_Num_Old", "cell_id", "Cell_ID", "ID", "id")   # never averaged (src/metacell_utils.py:334)


def _filter_valid(coords, triangles, r_max, min_angle_deg, ctx):
    """filter_triangles (src/metacell_utils.py:262-293) without the alpha shape: rows of `triangles` that are valid."""
    if len(triangles) == 0:
        return np.array([]).reshape(0, 3)
    en, thr = cos_threshold(min_angle_deg)
    n = len(coords)
    flag, _, _ = ops.collapse_candidates(coords, triangles, r_max, en, thr, np.zeros(n, np.int32), np.ones(n), 0.0, ctx=ctx)
    kept = triangles[(flag & 1).astype(bool)]
    return kept if len(kept) else np.array([]).reshape(0, 3)


def _mean_over_members(values, groups):
    """pandas `Series.mean()` of `values[pos]` for every group of positions: sum of the non-NaN entries in
    member order (numpy pairwise, reduced axis contiguous) / their count."""
    out = np.empty(len(groups))
    by_len = {}
    for g, pos in enumerate(groups):
        by_len.setdefault(len(pos), []).append(g)
    for m, gs in by_len.items():
        v = values[np.array([groups[g] for g in gs], dtype=np.int64).reshape(len(gs), m)]
        v = np.ascontiguousarray(v, dtype=np.float64)
        nan = np.isnan(v)
        if nan.any():
            cnt = (~nan).sum(axis=1)
            with np.errstate(invalid="ignore", divide="ignore"):
                out[gs] = np.where(nan, 0.0, v).sum(axis=1) / cnt
        else:
            out[gs] = v.sum(axis=1) / m
    return out


def greedy_triangle_collapse(aligned_df, max_metacell_size=3, max_iterations=1000, r_max=None, min_angle_deg=10,
                             use_alpha_shape=False, alpha=0.05, *, original_idx_col: str = "Cell_Num_Old",
                             metacell_idx_col: str = "metacell_id", x_col: str = "X", y_col: str = "Y",
                             cell_type_col: str = "cell_type", return_object: bool = False, verbose: bool = True, ctx=None):
    from scipy.spatial import Delaunay

    if use_alpha_shape:
        try:
            from alphashape import alphashape  # noqa: F401
            from shapely.geometry import Polygon  # noqa: F401
            raise NotImplementedError("alpha-shape containment (shapely) is outside this package; pass use_alpha_shape=False")
        except ImportError:  # the reference prints this and carries on without the alpha shape (:271-273)
            print("Warning: alphashape not available, skipping alpha shape filtering")
            use_alpha_shape = False

    # input contract of src/metacell_utils.py:296-309 (messages are part of it: callers match on them)
    have = set(aligned_df.columns)
    missing = [c for c in (x_col, y_col, cell_type_col, original_idx_col) if c not in have]
    if missing:
        raise ValueError(f"Input dataframe missing required columns: {missing}")
    aligned_df = aligned_df.copy()
    original_ids_by_pos = aligned_df[original_idx_col].to_numpy()
    repeated = pd.Index(original_ids_by_pos).duplicated()          # every occurrence after the first
    if repeated.any():
        dups = original_ids_by_pos[repeated][:5].tolist()
        raise ValueError(f"'{original_idx_col}' must be unique per original cell. Found duplicates (examples): {dups}")

    # Qhull calls are the bulk of a collapse (1.14 of 1.35 s at 100k cells, one per iteration plus two).  Two of them repeat a
    # triangulation that was just made -- iteration 0 triangulates the original cells again, and the final triangulation
    # repeats the last iteration's when the loop stops for lack of candidates -- so those are reused; and each iteration's
    # triangulation is started in a Qhull helper process as soon as the merged coordinates are known, while the rest of the
    # frame merge is still being done here.  Same points -> same simplices: results are unchanged (tests/golden/metacell.npz).
    from . import qhull_pool

    original_coords = aligned_df[[x_col, y_col]].to_numpy()
    ready = None                      # (points, simplices or ticket) of a triangulation made ahead of its use
    if len(original_coords) >= 4:
        original_raw = Delaunay(original_coords).simplices
        ready = (np.asarray(original_coords, dtype=np.float64), original_raw)
        original_delaunay_pos = _filter_valid(original_coords, original_raw, r_max, min_angle_deg, ctx)
    else:
        original_delaunay_pos = np.array([], dtype=int).reshape(0, 3)

    def triangulate(points):
        """Delaunay(points).simplices, from `ready` when it holds exactly these points."""
        nonlocal ready
        have, ready = ready, None
        if have is not None and have[0].shape == points.shape and np.array_equal(have[0], points):
            return have[1].result() if hasattr(have[1], "result") else have[1]
        return Delaunay(points).simplices
    # pre-collapse triangles in ORIGINAL-id space (:326-331); an empty list keeps the id dtype
    original_delaunay = (original_ids_by_pos[original_delaunay_pos.astype(int)] if original_delaunay_pos.size
                         else np.empty((0, 3), dtype=original_ids_by_pos.dtype))

    # columns that are averaged over a metacell's members = everything that is neither a coordinate, the cell type, nor an
    # id: the well-known id names plus the two the caller named (:333-340), as a set difference in frame order
    not_averaged = {x_col, y_col, cell_type_col, original_idx_col, metacell_idx_col, *_WELL_KNOWN_ID_COLUMNS}
    other_cols = [c for c in aligned_df.columns if c not in not_averaged]

    # every cell starts as a metacell of size 1 (:331-348); members are ORIGINAL ids
    metacell_df = pd.DataFrame({x_col: aligned_df[x_col].to_numpy(), y_col: aligned_df[y_col].to_numpy(),
                                cell_type_col: aligned_df[cell_type_col].to_numpy()})
    metacell_df["size"] = 1
    metacell_df["members"] = [[v] for v in original_ids_by_pos.tolist()]
    for c in other_cols:
        metacell_df[c] = aligned_df[c].to_numpy()
    metacell_df[metacell_idx_col] = range(len(metacell_df))
    member_pos = [[i] for i in range(len(aligned_df))]            # positions in aligned_df, parallel to `members`
    orig_x = aligned_df[x_col].to_numpy(dtype=np.float64)
    orig_y = aligned_df[y_col].to_numpy(dtype=np.float64)
    en, thr = cos_threshold(min_angle_deg)

    last_made = None
    if verbose:
        print(f"Starting greedy triangle collapse:\n  Initial cells: {len(aligned_df)}\n  Max metacell size: {max_metacell_size}")
    for iteration in range(max_iterations):
        coords = metacell_df[[x_col, y_col]].values
        if len(coords) < 4:
            break
        triangles_raw = triangulate(coords)
        last_made = (np.asarray(coords, dtype=np.float64), triangles_raw)
        type_id = pd.factorize(metacell_df[cell_type_col].to_numpy(), use_na_sentinel=False)[0].astype(np.int32)
        size = metacell_df["size"].to_numpy(dtype=np.float64)
        flag, perim, total = ops.collapse_candidates(coords, triangles_raw, r_max, en, thr, type_id, size, max_metacell_size, ctx=ctx)
        if not (flag & 1).any():
            break                                                   # no valid triangles (:386-389)
        cand = np.flatnonzero(flag & 2)                             # in triangle order = the reference's candidate order
        if len(cand) == 0:
            break                                                   # nothing collapsible (:433-436)
        tri_c = triangles_raw[cand]
        selected, _ = ops.greedy_disjoint(tri_c, perim[cand], len(coords), ctx=ctx)
        sel = np.flatnonzero(selected)
        sel = sel[np.lexsort((sel, perim[cand][sel]))]              # batch order = stable sort by priority (:439)
        batch = tri_c[sel]
        groups = [member_pos[a] + member_pos[b] + member_pos[c] for a, b, c in batch.tolist()]
        members = metacell_df["members"].tolist()
        merged = {
            x_col: _mean_over_members(orig_x, groups), y_col: _mean_over_members(orig_y, groups),
            cell_type_col: metacell_df[cell_type_col].to_numpy()[batch[:, 0]],
            "size": metacell_df["size"].to_numpy()[batch].sum(axis=1),
            "members": [members[a] + members[b] + members[c] for a, b, c in batch.tolist()],
        }
        remove = batch.reshape(-1)
        keep_mask = np.ones(len(metacell_df), bool)
        keep_mask[remove] = False
        # the next iteration's points are known now (kept metacells, then the merged ones): triangulate them in a helper
        # while the remaining columns are merged below
        next_coords = np.vstack((np.asarray(coords, dtype=np.float64)[keep_mask], np.column_stack((merged[x_col], merged[y_col]))))
        last_made = None
        if len(next_coords) >= 4:
            ready = (next_coords, qhull_pool.pool().submit(next_coords))
        for col in other_cols:                 # the averaged / carried columns, in frame order
            if col in ("size", "members"):     # an input column of that name is the metacell's own bookkeeping from here on
                continue
            if pd.api.types.is_numeric_dtype(metacell_df[col]):
                merged[col] = _mean_over_members(aligned_df[col].to_numpy(dtype=np.float64), groups)   # true mean over original cells
            else:
                merged[col] = metacell_df[col].to_numpy()[batch[:, 0]]                                  # first vertex's value
        member_pos = [member_pos[i] for i in np.flatnonzero(keep_mask)] + groups
        metacell_df = metacell_df.drop(remove).reset_index(drop=True)
        metacell_df = pd.concat([metacell_df, pd.DataFrame(merged)], ignore_index=True)
        metacell_df[metacell_idx_col] = range(len(metacell_df))

    final_coords = metacell_df[[x_col, y_col]].values
    if len(final_coords) >= 4:
        if last_made is not None and ready is None:   # the loop stopped on these very points: their triangulation exists
            ready = last_made
        final_delaunay = _filter_valid(final_coords, triangulate(np.asarray(final_coords)), r_max, min_angle_deg, ctx)
    else:
        final_delaunay = np.array([]).reshape(0, 3)
    if verbose:
        print(f"\nCollapse complete:\n  Original cells: {len(aligned_df)}\n  Final metacells: {len(metacell_df)}\n"
              f"  Avg metacell size: {metacell_df['size'].mean():.1f}\n  Final triangles: {len(final_delaunay)}")
    if return_object:
        params = {"max_metacell_size": max_metacell_size, "max_iterations": max_iterations, "r_max": r_max,
                  "min_angle_deg": min_angle_deg, "use_alpha_shape": use_alpha_shape, "alpha": alpha}
        return MetaCell(original_df=aligned_df, params=params, x_col=x_col, y_col=y_col, cell_type_col=cell_type_col,
                        original_idx_col=original_idx_col, metacell_idx_col=metacell_idx_col,
                        original_delaunay=original_delaunay, metacell_df=metacell_df, metacell_delaunay=final_delaunay)
    return metacell_df, final_delaunay


def _member_xy(frame, labels, x_col, y_col):
    """frame.loc[labels, [x, y]].values for one flat label list (KeyError on a missing label, as .loc raises)."""
    if not frame.index.is_unique:
        raise ValueError("unpack_metacell_matches: the cell id index has duplicate labels")
    pos = frame.index.get_indexer(labels)
    if (pos < 0).any():
        missing = [lab for lab, q in zip(labels, pos) if q < 0][:5]
        raise KeyError(f"{missing} not in index")
    return np.ascontiguousarray(frame[[x_col, y_col]].to_numpy(dtype=np.float64)[pos])


def unpack_metacell_matches(metacell_matches, metacell_aligned_df, metacell_ref_df, aligned_df=None, ref_df=None,
                            strategy="distribute", aligned_original_idx_col: Optional[str] = None,
                            ref_original_idx_col: Optional[str] = None, x_col: str = "X", y_col: str = "Y"):
    """Metacell-level matches -> individual cell matches, signature and results of src/metacell_utils.py:564-766
    (SURVEY 8(f4)).  The list bookkeeping stays on the host; the per-match optimal assignments of
    strategy='nearest' (cdist + np.tile + linear_sum_assignment per match, :711-761) run as ONE batched launch
    (`same_batched_assign`, csrc/match.hip), which follows scipy's solver step for step so ties resolve as in the
    reference.  Non-finite member coordinates raise ValueError (scipy: "cost matrix is infeasible")."""
    aligned_lookup = ref_lookup = None
    if aligned_df is not None and aligned_original_idx_col is not None:
        if aligned_original_idx_col not in aligned_df.columns:
            raise ValueError(f"aligned_df missing aligned_original_idx_col='{aligned_original_idx_col}'")
        aligned_lookup = aligned_df.set_index(aligned_original_idx_col, drop=False)
    if ref_df is not None and ref_original_idx_col is not None:
        if ref_original_idx_col not in ref_df.columns:
            raise ValueError(f"ref_df missing ref_original_idx_col='{ref_original_idx_col}'")
        ref_lookup = ref_df.set_index(ref_original_idx_col, drop=False)
    ref_has_metacells = ("members" in metacell_ref_df.columns
                         and metacell_ref_df["members"].apply(lambda x: isinstance(x, list)).any())
    if ref_has_metacells and strategy == "nearest" and (aligned_df is None or ref_df is None):
        raise ValueError("When ref has metacells and strategy='nearest', must provide both aligned_df and ref_df "
                         "for nearest neighbor unpacking.")
    if strategy == "nearest" and aligned_df is None:
        raise ValueError("strategy='nearest' requires aligned_df parameter")

    a_members = metacell_aligned_df["members"].tolist()
    r_members = metacell_ref_df["members"].tolist() if ref_has_metacells else None
    a_ids = metacell_matches["Aligned_metacell_id"].tolist()
    r_ids = metacell_matches["Ref_metacell_id"].tolist()
    out_a, out_r = [], []
    if ref_has_metacells and strategy == "nearest" and a_ids:      # optimal assignment on pairwise distances (:687-742)
        a_lists, r_lists = [a_members[i] for i in a_ids], [r_members[j] for j in r_ids]
        a_off = np.concatenate(([0], np.cumsum([len(m) for m in a_lists]))).astype(np.int64)
        r_off = np.concatenate(([0], np.cumsum([len(m) for m in r_lists]))).astype(np.int64)
        out_a = [m for ms in a_lists for m in ms]
        r_flat = [m for ms in r_lists for m in ms]
        if out_a:
            axy = _member_xy(aligned_lookup if aligned_lookup is not None else aligned_df, out_a, x_col, y_col)
            rxy = _member_xy(ref_lookup if ref_lookup is not None else ref_df, r_flat, x_col, y_col)
            try:
                local = ops.batched_assign(a_off, r_off, axy, rxy)
            except _lib.SameHipError as e:
                if e.code == _lib.SAME_ERANGE:
                    raise ValueError("cost matrix is infeasible") from e
                raise
            pick = np.repeat(r_off[:-1], np.diff(a_off)) + local
            out_r = [r_flat[q] for q in pick]
    else:
        for a_idx, r_idx in zip(a_ids, r_ids):
            am = a_members[a_idx]
            if not ref_has_metacells:
                if strategy in ("distribute", "nearest"):       # every member -> the same reference cell (:652-669)
                    out_a.extend(am)
                    out_r.extend([r_idx] * len(am))
                continue
            rm = r_members[r_idx]
            if strategy == "distribute":                         # deal the ref members round-robin (:675-685)
                out_a.extend(am)
                out_r.extend(rm[i % len(rm)] for i in range(len(am)))
            else:
                raise ValueError(f"Unknown strategy: {strategy}")
    if not out_a:
        return pd.DataFrame([])
    return pd.DataFrame({"Aligned_cell_id": out_a, "Ref_cell_id": out_r})
