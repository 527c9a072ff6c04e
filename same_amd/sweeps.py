"""Violation sweeps under a candidate matching.

* LazyOrientationSweep -- the arithmetic body of _lazy_orientation_callback
  (src/same.py:621-703): matching from the solver's x vector, per-triangle orientation sign of
  the matched reference vertices against the source sign, checked count and the ascending list
  of flipped triangles.  State stays resident on the GPU between incumbents.
* verify_spatial_preservation / print_violation_report -- src/violationhelper.py, same
  signature and the same dict.
* triangle_area_flips -- src/same.py:1355-1402 (signed areas before / after, flipped list).
"""
import numpy as np

from . import ops
from ._rows import rows_array, values_array
from ._trace import stage

_EDGES = ((0, 1), (0, 2), (1, 2))


class LazyOrientationSweep:
    """Device-resident replacement for the model._* state the callback reads (src/same.py:1153-1158)."""

    def __init__(self, valid_pairs, aligned_delaunay, source_signs, ref_xy, n_aligned, ctx=None):
        self.tris = rows_array(aligned_delaunay)
        pairs = np.asarray(valid_pairs, dtype=np.int64).reshape(-1, 2)
        self.bound = ops.BoundSweep(self.tris, values_array(source_signs).astype(np.int8), ref_xy, n_aligned, pairs, ctx=ctx)

    def sweep(self, x_vals):
        """-> (checked, violating_tris [(tri_idx, a, b, c)] ascending, match_pair_idx array)."""
        with stage("lazy orientation sweep"):
            checked, viol, _match, pidx = self.bound.sweep_x(np.asarray(x_vals, dtype=np.float64))
        t = self.tris
        violating = [(int(i), t[i][0], t[i][1], t[i][2]) for i in viol]
        return checked, violating, pidx

    def select_cuts(self, x_vals, allowed_frac=None, per_inc_limit=None, remaining_global=None):
        """The decision logic of src/same.py:671-692 -> list of (tri_idx, pair_idx_a, pair_idx_b, pair_idx_c)."""
        checked, violating, pidx = self.sweep(x_vals)
        if checked == 0 or not violating:
            return []
        if allowed_frac is not None and len(violating) / float(checked) <= allowed_frac:
            return []
        cuts = []
        for tri_idx, a, b, c in violating:
            if per_inc_limit is not None and len(cuts) >= per_inc_limit:
                break
            if remaining_global is not None and remaining_global - len(cuts) <= 0:
                break
            cuts.append((tri_idx, int(pidx[a]), int(pidx[b]), int(pidx[c])))
        return cuts


def match_vector(n_aligned, aligned_idx, ref_idx):
    """match[i] = ref matched to aligned i (-1 none); later rows win like the reference's dicts."""
    m = np.full(int(n_aligned), -1, np.int32)
    m[np.asarray(aligned_idx, dtype=np.int64)] = np.asarray(ref_idx, dtype=np.int64)
    return m


def verify_spatial_preservation(aligned_df, ref_df, matches_df, triangle_info, tolerance=1e-6, ctx=None, _sweep=None):
    """Same report as src/violationhelper.py:1-134 (`tolerance` is unused there too).
    `_sweep(axy, rxy, tris, match) -> (edge, tri_flag, point_flag, counts)` replaces the single-GPU launch (the
    triangle-block sharded form in dist.py passes its own)."""
    axy = aligned_df[["X", "Y"]].to_numpy(dtype=np.float64)
    rxy = ref_df[["X", "Y"]].to_numpy(dtype=np.float64)
    match = match_vector(len(aligned_df), matches_df["aligned_idx"].to_numpy(), matches_df["ref_idx"].to_numpy())
    keys = list(triangle_info.keys())
    verts = [triangle_info[k]["vertices"] for k in keys]
    tris = np.array([list(v) for v in verts], dtype=np.int64).reshape(-1, 3)
    edge, tflag, pflag, counts = _sweep(axy, rxy, tris, match) if _sweep else ops.xyorder_sweep(axy, rxy, tris, match, ctx=ctx)

    violations = {"x_order_violations": [], "y_order_violations": [], "triangles_with_violations": set(),
                  "points_with_violations": set(),
                  "violation_summary": {"total_triangles": len(triangle_info), "violated_triangles": 0,
                                        "total_comparisons": 0, "total_violations": 0}}
    for n in np.flatnonzero(tflag):
        k = keys[n]
        v = verts[n]
        for e, (p, q) in enumerate(_EDGES):
            f = edge[n, e]
            if not f & 6:
                continue
            v1, v2 = v[p], v[q]
            r1, r2 = match[v1], match[v2]
            if f & 2:
                violations["x_order_violations"].append({
                    "triangle_idx": k,
                    "point1": {"aligned_idx": v1, "ref_idx": r1, "orig_x": axy[v1, 0], "matched_x": rxy[r1, 0]},
                    "point2": {"aligned_idx": v2, "ref_idx": r2, "orig_x": axy[v2, 0], "matched_x": rxy[r2, 0]}})
            if f & 4:
                violations["y_order_violations"].append({
                    "triangle_idx": k,
                    "point1": {"aligned_idx": v1, "ref_idx": r1, "orig_y": axy[v1, 1], "matched_y": rxy[r1, 1]},
                    "point2": {"aligned_idx": v2, "ref_idx": r2, "orig_y": axy[v2, 1], "matched_y": rxy[r2, 1]}})
        violations["triangles_with_violations"].add(k)
    violations["triangles_with_violations"] = list(violations["triangles_with_violations"])
    violations["points_with_violations"] = list(np.flatnonzero(pflag))
    s = violations["violation_summary"]
    s["total_comparisons"], s["total_violations"], s["violated_triangles"] = (int(c) for c in counts)
    s["percent_triangles_violated"] = (s["violated_triangles"] / s["total_triangles"] * 100 if s["total_triangles"] > 0 else 0)
    s["percent_violations"] = (s["total_violations"] / s["total_comparisons"] * 100 if s["total_comparisons"] > 0 else 0)
    return violations


def print_violation_report(violations):
    """The report of src/violationhelper.py:136-147, line for line (the text is the reference's output format)."""
    s = violations["violation_summary"]
    title = "Spatial Preservation Violation Report"
    report = ["", title, "=" * len(title),
              "Total triangles analyzed: %d" % s["total_triangles"],
              "Triangles with violations: %d (%.2f%%)" % (s["violated_triangles"], s["percent_triangles_violated"]),
              "Total position comparisons: %d" % s["total_comparisons"],
              "Total violations found: %d (%.2f%%)" % (s["total_violations"], s["percent_violations"]),
              "Number of points involved in violations: %d" % len(violations["points_with_violations"])]
    print("\n".join(report))


def triangle_area_flips(aligned_df, ref_df, aligned_delaunay, aligned_to_ref, ctx=None, _sweep=None):
    """src/same.py:1355-1402 -> (areas_before dict, areas_after dict (None if unmatched),
    flipped list ascending, matched_vertices dict).  `_sweep(axy, rxy, tris, match)` as in verify_spatial_preservation."""
    axy = aligned_df[["X", "Y"]].to_numpy(dtype=np.float64)
    rxy = ref_df[["X", "Y"]].to_numpy(dtype=np.float64)
    match = np.full(len(aligned_df), -1, np.int32)
    for i, j in aligned_to_ref.items():
        match[int(i)] = int(j)
    tris = rows_array(aligned_delaunay)
    before, after, m3, fl = _sweep(axy, rxy, tris, match) if _sweep else ops.area_flip(axy, rxy, tris, match, ctx=ctx)
    n = len(tris)
    areas_before = {t: before[t] for t in range(n)}
    areas_after = {t: (after[t] if m3[t].all() else None) for t in range(n)}
    matched_vertices = {t: [bool(b) for b in m3[t]] for t in range(n)}
    flipped = [int(t) for t in np.flatnonzero(fl)]
    return areas_before, areas_after, flipped, matched_vertices
