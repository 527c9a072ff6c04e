"""CPU oracle for the SAME pre-MIP hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; ``same_amd/`` never does (tests/test_boundary.py enforces it).

Arithmetic lives in ``same_oracle.c`` (scalar C, reference operation order); this module
binds it with ctypes and restates the reference's *host* logic around it (frame compaction,
the same-type re-add pass, dict-shaped outputs) as plain, literal Python.  Every function
cites the reference lines it follows (file:line into /root/reference).  The oracle is pinned
against outputs of the imported reference in ``tests/golden/`` (tests/test_oracle_golden.py).
"""
import ctypes
import os
import subprocess
from collections import defaultdict

import numpy as np
import pandas as pd

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_f64 = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_f32 = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_i32 = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_i8 = np.ctypeslib.ndpointer(np.int8, flags="C_CONTIGUOUS")
_u8 = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_i64 = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_I64 = ctypes.c_int64
_INT = ctypes.c_int
_DBL = ctypes.c_double
_VP = ctypes.c_void_p


def build(force=False):
    """Compile libsame_oracle.so with gcc (no-op when up to date).  SAME_ORACLE_SANITIZE=1 selects the
    -fsanitize=address,undefined build (libsame_oracle_asan.so; the process must run with libasan preloaded --
    tests/test_oracle_sanitized.py does that in a child process, on the CPU only)."""
    asan = os.environ.get("SAME_ORACLE_SANITIZE") == "1"
    so = os.path.join(_HERE, "libsame_oracle_asan.so" if asan else "libsame_oracle.so")
    src = os.path.join(_HERE, "same_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, os.path.basename(so)] + (["-B"] if force else []))
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        L.orc_pair_cost_f64.argtypes = [_f64, _f64, _INT, _f64, _f64, _i32, _I64, _DBL, _f64]
        L.orc_dense_cost_f64.argtypes = [_f64, _f64, _INT, _f64, _f64, _I64, _I64, _I64, _DBL, _f64, _I64]
        L.orc_dense_cost_f32.argtypes = [_f32, _f32, _INT, _f32, _f32, _I64, _I64, _I64, ctypes.c_float, _f32, _I64]
        L.orc_pair_cost_f32.argtypes = [_f32, _f32, _INT, _f32, _f32, _i32, _I64, ctypes.c_float, _f32]
        _u32q = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
        L.orc_quantize_u32.argtypes = [_f64, _I64, _DBL, _DBL, _u32q]
        L.orc_dense_cost_q32.argtypes = [_u32q, _u32q, _f64, _f64, _INT, _f64, _f64, _I64, _I64, _I64, _DBL, _DBL, _DBL, _f64, _I64]
        L.orc_knn_prune.argtypes = [_f64, _f64, _I64, _I64, _I64, _DBL, _INT, _i32, _f64, _i32]
        L.orc_tri_classify.argtypes = [_f64, _i32, _I64, _DBL, _INT, _DBL, _VP, _u8, _f64, _f64]
        L.orc_tri_sign_weight.argtypes = [_f64, _VP, _i32, _I64, _i8, _VP]
        L.orc_orient_sweep.argtypes = [_i32, _I64, _i8, _f64, _i32, _u8, _i64, _i32, _i64]
        L.orc_xyorder_sweep.argtypes = [_f64, _f64, _i32, _I64, _i32, _I64, _u8, _u8, _u8, _i64]
        L.orc_area_flip.argtypes = [_f64, _f64, _i32, _I64, _i32, _f64, _f64, _u8, _u8]
        L.orc_pair_rowmin.argtypes = [_i32, _f64, _I64, _I64, _f64]
        L.orc_assign_matrix.argtypes = [_i32, _f64, _I64, _f64, _I64, _I64, _DBL, _f64]
        L.orc_eager_signs.argtypes = [_f64, _i32, _I64, _i32, _INT, _i8]
        L.orc_window_mask.argtypes = [_f64, _I64, _DBL, _DBL, _DBL, _DBL, _u8]
        _u32 = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
        L.orc_tri_flip_stats.argtypes = [_f64, _f64, _u8, _VP, _i32, _I64, _I64, _u8, _u32, _u32]
        L.orc_batched_assign.argtypes = [_I64, _i64, _i64, _f64, _f64, _i32]
        L.orc_merge_dedup.argtypes = [_u8, _i32, _i32, _i32, _I64, _i32]
        L.orc_merge_dedup.restype = ctypes.c_int64
        _LIB = L
    return _LIB


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _xy(df):
    return _c(df[["X", "Y"]].to_numpy(dtype=np.float64), np.float64)


# --------------------------------------------------------------------------- a2
def knn_prune(axy, rxy, radius, knn, row_begin=0, row_end=None):
    """Padded per-row candidate lists; semantics of src/utils.py:720-728 (see same_oracle.c)."""
    axy, rxy = _c(axy, np.float64), _c(rxy, np.float64)
    row_end = len(axy) if row_end is None else row_end
    n = row_end - row_begin
    idx = np.empty((n, knn), np.int32)
    d2 = np.empty((n, knn), np.float64)
    cnt = np.empty(n, np.int32)
    rc = lib().orc_knn_prune(axy, rxy, row_begin, row_end, len(rxy), float(radius), int(knn), idx, d2, cnt)
    assert rc == 0
    return idx, d2, cnt


def pairs_from_padded(idx, row_offset=0):
    """Row-major walk of the padded lists -> (P,2) int64 pairs in src/utils.py:731 order."""
    rows, cols = np.nonzero(idx >= 0)
    return np.column_stack((rows.astype(np.int64) + row_offset, idx[rows, cols].astype(np.int64)))


def compact_pairs(aligned_df, ref_df, knn_pairs):
    """src/utils.py:734-742: np.unique of used rows, iloc + reset_index, re-index the pairs."""
    ua = np.unique(knn_pairs[:, 0])
    ur = np.unique(knn_pairs[:, 1])
    new_aligned = aligned_df.iloc[ua].reset_index(drop=True)
    new_ref = ref_df.iloc[ur].reset_index(drop=True)
    map_a = {old: new for new, old in enumerate(ua)}
    map_r = {old: new for new, old in enumerate(ur)}
    new_pairs = np.array([(map_a[i], map_r[j]) for i, j in knn_pairs])
    return new_aligned, new_ref, new_pairs


def find_knn_within_radius(aligned_df, ref_df, radius=25, knn=5):
    """Same signature and outputs as src/utils.py:709-742."""
    idx, _, _ = knn_prune(_xy(aligned_df), _xy(ref_df), radius, knn)
    knn_pairs = pairs_from_padded(idx)
    if len(knn_pairs) == 0:
        knn_pairs = np.empty((0, 2), int)
    return compact_pairs(aligned_df, ref_df, knn_pairs)


def find_knn_with_cell_type_priority(aligned_df, ref_df, radius, knn=5):
    """src/knn_utils.py:5-78 (sequential: ref_points_matched carries across rows)."""
    aligned_df, ref_df, all_pairs = find_knn_within_radius(aligned_df, ref_df, radius, knn=knn)
    by_aligned = defaultdict(list)
    for i, j in all_pairs:
        by_aligned[int(i)].append(int(j))
    axy, rxy = _xy(aligned_df), _xy(ref_df)
    atype = aligned_df["cell_type"].to_numpy()
    rtype = ref_df["cell_type"].to_numpy()
    out, taken = [], set()
    for i in range(len(aligned_df)):
        if i not in by_aligned:
            continue
        pts = [(j, rtype[j], np.sqrt((axy[i, 0] - rxy[j, 0]) ** 2 + (axy[i, 1] - rxy[j, 1]) ** 2))
               for j in by_aligned[i]]
        pts.sort(key=lambda x: x[2])  # stable, src/knn_utils.py:49
        j0, t0 = pts[0][0], pts[0][1]
        if t0 == atype[i] and j0 not in taken:
            out.append((i, j0))
            taken.add(j0)
        else:
            out.extend((i, j) for j, _, _ in pts)
    return aligned_df, ref_df, out


# --------------------------------------------------------------------------- a4
def pair_cost_arrays(A, R, axy, rxy, pairs, w, dtype=np.float64):
    dt = np.dtype(dtype)
    A, R = _c(A, dt), _c(R, dt)
    pairs = _c(pairs, np.int32).reshape(-1, 2)
    out = np.empty(len(pairs), dt)
    T = A.shape[1] if A.ndim == 2 else 0
    fn = lib().orc_pair_cost_f64 if dt == np.float64 else lib().orc_pair_cost_f32
    fn(A.reshape(-1), R.reshape(-1), T, _c(axy, dt), _c(rxy, dt), pairs, len(pairs), float(w), out)
    return out


def pair_costs(aligned_df, ref_df, valid_pairs, commonCT, dist_ct_coeff):
    """src/same.py:1180-1189 -> list of float, one per pair, in pair order."""
    A = aligned_df[list(commonCT)].to_numpy(dtype=np.float64)
    R = ref_df[list(commonCT)].to_numpy(dtype=np.float64)
    return list(pair_cost_arrays(A, R, _xy(aligned_df), _xy(ref_df), np.asarray(valid_pairs), dist_ct_coeff))


def dense_cost(A, R, axy, rxy, w, row_begin=0, row_end=None, dtype=np.float64):
    dt = np.dtype(dtype)
    A, R, axy, rxy = _c(A, dt), _c(R, dt), _c(axy, dt), _c(rxy, dt)
    row_end = len(A) if row_end is None else row_end
    out = np.empty((row_end - row_begin, len(R)), dt)
    fn = lib().orc_dense_cost_f64 if dt == np.float64 else lib().orc_dense_cost_f32
    fn(A.reshape(-1), R.reshape(-1), A.shape[1], axy, rxy, len(R), row_begin, row_end, w, out.reshape(-1), len(R))
    return out


def dense_cost_q32(A, R, axy, rxy, w, offset, log2_scale, row_begin=0, row_end=None, rel_tol=1e-6):
    """Twin of the opt-in fixed-point dense build (ops.dense_cost_q32): same grid, exact integer sums, fp64 remainder."""
    A, R = _c(A, np.float64), _c(R, np.float64)
    row_end = len(A) if row_end is None else row_end
    T = A.shape[1]
    Aq, Rq = np.empty(A.shape, np.uint32), np.empty(R.shape, np.uint32)
    scale = float(2.0 ** log2_scale)
    lib().orc_quantize_u32(A.reshape(-1), A.size, float(offset), scale, Aq.reshape(-1))
    lib().orc_quantize_u32(R.reshape(-1), R.size, float(offset), scale, Rq.reshape(-1))
    out = np.empty((row_end - row_begin, len(R)), np.float64)
    lib().orc_dense_cost_q32(Aq.reshape(-1), Rq.reshape(-1), A.reshape(-1), R.reshape(-1), T, _c(axy, np.float64), _c(rxy, np.float64),
                             len(R), row_begin, row_end, float(w), 1.0 / scale, float(rel_tol), out.reshape(-1), len(R))
    return out


# --------------------------------------------------------------------------- a7
def _ordered_key(c):
    """Monotone int64 key of a double (for bisection on the double lattice)."""
    u = np.float64(c).view(np.int64)
    return int(u) if u >= 0 else int(-(u & np.int64(0x7FFFFFFFFFFFFFFF)))


def _from_key(k):
    if k >= 0:
        return float(np.int64(k).view(np.float64))
    return float(-np.int64(-k).view(np.float64))


def angle_fails(c, min_angle_deg):
    """The reference's own test on one clipped cosine: src/helpers.py:287-288,319."""
    return bool(np.degrees(np.arccos(c)) < min_angle_deg)


def cos_threshold(min_angle_deg):
    """Smallest double c in [-1,1] with degrees(arccos(c)) < min_angle_deg (monotone in c).

    Returns (enabled, thr): thr = +inf when no cosine fails, -inf when every one does."""
    if min_angle_deg is None:
        return 0, float("inf")
    if not angle_fails(1.0, min_angle_deg):
        return 1, float("inf")
    if angle_fails(-1.0, min_angle_deg):
        return 1, float("-inf")
    lo, hi = _ordered_key(-1.0), _ordered_key(1.0)  # fails(lo) False, fails(hi) True
    while hi - lo > 1:
        mid = (lo + hi) // 2
        if angle_fails(_from_key(mid), min_angle_deg):
            hi = mid
        else:
            lo = mid
    return 1, _from_key(hi)


def tri_classify(points, triangles, radius, min_angle_deg, type_id=None):
    xy = _c(points, np.float64)
    tris = _c(triangles, np.int32).reshape(-1, 3)
    Tr = len(tris)
    cls = np.empty(Tr, np.uint8)
    perim = np.empty(Tr, np.float64)
    maxcos = np.empty(Tr, np.float64)
    en, thr = cos_threshold(min_angle_deg)
    tptr = None if type_id is None else _c(type_id, np.int32)
    lib().orc_tri_classify(xy, tris, Tr, float(radius), en, thr,
                           None if tptr is None else tptr.ctypes.data, cls, perim, maxcos)
    # maxcos == 2.0 encodes the reference's "angle 0" for a zero-length side (helpers.py:284):
    # 2.0 >= thr holds exactly when 0 < min_angle_deg (thr is +inf otherwise).
    return cls, perim, maxcos


def filter_triangles_by_radius(points, triangles, radius, aligned_df=None, ignore_same_type_triangles=False,
                               ensure_min_triangle_per_node=True, remove_unconstrained_nodes=False,
                               min_angle_deg=15):
    """src/helpers.py:233-395, same return shapes (list of triangle rows [+ set])."""
    triangles = np.asarray(triangles)
    tris = triangles.reshape(-1, 3) if triangles.size else triangles.reshape(0, 3)
    use_type = bool(ignore_same_type_triangles and aligned_df is not None)
    type_id = None
    if use_type:
        import pandas as pd
        type_id = pd.factorize(aligned_df["cell_type"].to_numpy(), use_na_sentinel=False)[0].astype(np.int32)
    cls, perim, _ = tri_classify(points, tris, radius, min_angle_deg, type_id)
    filtered = []
    nodes_with_triangle, nodes_any_valid, best = set(), set(), {}
    for t, tri in enumerate(tris):
        if cls[t] in (1, 2):
            continue
        for v in tri:
            nodes_any_valid.add(int(v))
        if cls[t] == 3:
            if ensure_min_triangle_per_node:
                score = float(perim[t])
                for v in tri:
                    prev = best.get(int(v))
                    if prev is None or score < prev[0]:
                        best[int(v)] = (score, tri)
            continue
        filtered.append(tri)
        for v in tri:
            nodes_with_triangle.add(int(v))
    n_points = len(points)
    unconstrained = set(range(n_points)) - nodes_any_valid
    if use_type and ensure_min_triangle_per_node:
        missing = [i for i in range(n_points) if i not in nodes_with_triangle and i not in unconstrained]
        if missing:
            added_set = set(tuple(map(int, tri)) for tri in filtered)
            for i in missing:
                cand = best.get(int(i))
                if cand is None:
                    continue
                key = tuple(map(int, cand[1]))
                if key not in added_set:
                    filtered.append(cand[1])
                    added_set.add(key)
    if remove_unconstrained_nodes:
        return filtered, unconstrained
    return filtered


# --------------------------------------------------------------------------- a8
def tri_sign_weight(xy, size, triangles):
    tris = _c(triangles, np.int32).reshape(-1, 3)
    sign = np.empty(len(tris), np.int8)
    weight = np.empty(len(tris), np.float64)
    size = _c(size, np.float64)
    lib().orc_tri_sign_weight(_c(xy, np.float64), size.ctypes.data, tris, len(tris), sign, weight.ctypes.data)
    return sign, weight


def triangle_weights(aligned_df, triangles):
    """src/same.py:1128-1135."""
    return list(tri_sign_weight(_xy(aligned_df), aligned_df["size"].to_numpy(dtype=np.float64), triangles)[1])


def source_signs(aligned_df, triangles):
    """src/same.py:1139-1146 -> list of np.float64 in {-1, 0, 1}."""
    return list(tri_sign_weight(_xy(aligned_df), np.ones(len(aligned_df)), triangles)[0].astype(np.float64))


# --------------------------------------------------------------------------- a9
def precompute_triangle_info(aligned_df, aligned_delaunay, aligned_simplex_map):
    """src/helpers.py:184-210 (dict insertion order = scan vertices ascending, then set order)."""
    xy = _xy(aligned_df)
    info = {}
    for ip in range(len(aligned_df)):
        for s in aligned_simplex_map[ip]:
            if s not in info:
                simplex = aligned_delaunay[s]
                coords = [(i, xy[i, 0], xy[i, 1]) for i in simplex]
                min_x = min(c[1] for c in coords); max_x = max(c[1] for c in coords)
                min_y = min(c[2] for c in coords); max_y = max(c[2] for c in coords)
                info[s] = {
                    "vertices": simplex,
                    "bounds": {"min_x": min_x, "max_x": max_x, "min_y": min_y, "max_y": max_y},
                    "max_x_vertex": next(c[0] for c in coords if c[1] == max_x),
                    "min_x_vertex": next(c[0] for c in coords if c[1] == min_x),
                    "max_y_vertex": next(c[0] for c in coords if c[2] == max_y),
                    "min_y_vertex": next(c[0] for c in coords if c[2] == min_y),
                }
    return info


def simplex_map(n_aligned, triangles):
    """src/same.py:1096-1099."""
    m = {i: set() for i in range(n_aligned)}
    for idx, simplex in enumerate(triangles):
        for i in simplex:
            m[i].add(idx)
    return m


# --------------------------------------------------------------------------- a10
def matching_from_x(x_vals, valid_pairs, n_aligned):
    """src/same.py:634-639: last pair with x>0.5 wins per aligned index."""
    match = np.full(n_aligned, -1, np.int32)
    pair_idx = np.full(n_aligned, -1, np.int64)
    for idx, (ip, jp) in enumerate(valid_pairs):
        if x_vals[idx] > 0.5:
            match[ip] = jp
            pair_idx[ip] = idx
    return match, pair_idx


def orient_sweep(triangles, src_sign, rxy, match):
    tris = _c(triangles, np.int32).reshape(-1, 3)
    flag = np.empty(len(tris), np.uint8)
    viol = np.empty(max(len(tris), 1), np.int32)
    checked = np.zeros(1, np.int64)
    nviol = np.zeros(1, np.int64)
    lib().orc_orient_sweep(tris, len(tris), _c(src_sign, np.int8), _c(rxy, np.float64), _c(match, np.int32),
                           flag, checked, viol, nviol)
    return int(checked[0]), viol[: int(nviol[0])].copy(), flag


def lazy_orientation_sweep(x_vals, valid_pairs, triangles, src_signs, ref_xy, n_aligned):
    """src/same.py:631-669 -> (checked, [(tri_idx, a, b, c), ...] ascending)."""
    match, _ = matching_from_x(x_vals, valid_pairs, n_aligned)
    tris = np.asarray(triangles).reshape(-1, 3)
    checked, viol, _ = orient_sweep(tris, np.asarray(src_signs).astype(np.int8), ref_xy, match)
    return checked, [(int(t), int(tris[t][0]), int(tris[t][1]), int(tris[t][2])) for t in viol]


# --------------------------------------------------------------------------- a11
def xyorder_sweep(axy, rxy, triangles, match):
    tris = _c(triangles, np.int32).reshape(-1, 3)
    n_m = len(axy)
    edge = np.empty((len(tris), 3), np.uint8)
    tflag = np.empty(len(tris), np.uint8)
    pflag = np.empty(n_m, np.uint8)
    counts = np.zeros(3, np.int64)
    lib().orc_xyorder_sweep(_c(axy, np.float64), _c(rxy, np.float64), tris, len(tris), _c(match, np.int32),
                            n_m, edge.reshape(-1), tflag, pflag, counts)
    return edge, tflag, pflag, counts


_EDGES = ((0, 1), (0, 2), (1, 2))


def verify_spatial_preservation(aligned_df, ref_df, matches_df, triangle_info, tolerance=1e-6):
    """src/violationhelper.py:1-134: same dict, built from the flat sweep outputs."""
    axy, rxy = _xy(aligned_df), _xy(ref_df)
    match = np.full(len(aligned_df), -1, np.int32)
    for a, r in zip(matches_df["aligned_idx"].to_numpy(), matches_df["ref_idx"].to_numpy()):
        match[int(a)] = int(r)  # last row wins, violationhelper.py:38-39
    keys = list(triangle_info.keys())
    tris = np.array([list(triangle_info[k]["vertices"]) for k in keys], dtype=np.int32).reshape(-1, 3)
    edge, tflag, pflag, counts = xyorder_sweep(axy, rxy, tris, match)
    out = {"x_order_violations": [], "y_order_violations": [], "triangles_with_violations": set(),
           "points_with_violations": set(),
           "violation_summary": {"total_triangles": len(triangle_info), "violated_triangles": 0,
                                 "total_comparisons": 0, "total_violations": 0}}
    for n, k in enumerate(keys):
        for e, (p, q) in enumerate(_EDGES):
            f = edge[n, e]
            if not f & 1:
                continue
            v1, v2 = tris[n][p], tris[n][q]
            r1, r2 = match[v1], match[v2]
            for bit, ax, name in ((2, 0, "x"), (4, 1, "y")):
                if f & bit:
                    out[f"{name}_order_violations"].append({
                        "triangle_idx": k,
                        "point1": {"aligned_idx": v1, "ref_idx": r1, f"orig_{name}": axy[v1, ax], f"matched_{name}": rxy[r1, ax]},
                        "point2": {"aligned_idx": v2, "ref_idx": r2, f"orig_{name}": axy[v2, ax], f"matched_{name}": rxy[r2, ax]}})
        if tflag[n]:
            out["triangles_with_violations"].add(k)
    out["points_with_violations"] = list(np.nonzero(pflag)[0])
    out["triangles_with_violations"] = list(out["triangles_with_violations"])
    s = out["violation_summary"]
    s["total_comparisons"], s["total_violations"], s["violated_triangles"] = (int(c) for c in counts)
    s["percent_triangles_violated"] = s["violated_triangles"] / s["total_triangles"] * 100 if s["total_triangles"] > 0 else 0
    s["percent_violations"] = s["total_violations"] / s["total_comparisons"] * 100 if s["total_comparisons"] > 0 else 0
    return out


# --------------------------------------------------------------------------- a12
def area_flip(axy, rxy, triangles, match):
    tris = _c(triangles, np.int32).reshape(-1, 3)
    before = np.empty(len(tris)); after = np.empty(len(tris))
    m3 = np.empty((len(tris), 3), np.uint8); fl = np.empty(len(tris), np.uint8)
    lib().orc_area_flip(_c(axy, np.float64), _c(rxy, np.float64), tris, len(tris), _c(match, np.int32),
                        before, after, m3.reshape(-1), fl)
    return before, after, m3, fl


# --------------------------------------------------------------------------- a5
def compute_mip_start_pairs(*, valid_pairs, costs, n_aligned, n_ref, aligned_sizes, no_match_penalty,
                            max_matches, init_method, init_big_m=1e9, init_hungarian_max_n=2000, verbose=True):
    """src/init_helpers.py:46-177."""
    method = str(init_method).lower()
    if method not in {"greedy", "hungarian"}:
        raise ValueError(f"Unknown init_method={init_method!r}. Use 'greedy' or 'hungarian'.")
    if method == "hungarian" and max_matches != 1:
        raise ValueError("init_method='hungarian' requires max_matches == 1.")
    if len(valid_pairs) != len(costs):
        raise ValueError("valid_pairs and costs must have the same length.")
    costs_arr = np.asarray(costs, dtype=float)
    unmatched_cost = float(no_match_penalty) * np.asarray(aligned_sizes, dtype=float)
    pairs32 = _c(np.asarray(valid_pairs).reshape(-1, 2), np.int32)
    chosen, unmatched = [], set()
    if method == "greedy":
        order = np.argsort(costs_arr, kind="stable")  # list.sort is stable, init_helpers.py:111-112
        best = np.empty(n_aligned)
        lib().orc_pair_rowmin(pairs32, _c(costs_arr, np.float64), len(pairs32), n_aligned, best)
        prefer = best < unmatched_cost
        used_a, used_r = set(), set()
        for idx in order:
            i, j = int(pairs32[idx, 0]), int(pairs32[idx, 1])
            if i in used_a or j in used_r or not prefer[i]:
                continue
            chosen.append((i, j, int(idx)))
            used_a.add(i); used_r.add(j)
        unmatched = set(range(n_aligned)) - used_a
    else:
        if (n_aligned + n_ref) > int(init_hungarian_max_n):
            return [], set()
        from scipy.optimize import linear_sum_assignment
        mat = np.empty((n_aligned, n_ref + n_aligned))
        lib().orc_assign_matrix(pairs32, _c(costs_arr, np.float64), len(pairs32), _c(unmatched_cost, np.float64),
                                n_aligned, n_ref, float(init_big_m), mat.reshape(-1))
        row_ind, col_ind = linear_sum_assignment(mat)
        pair_to_var = {(int(i), int(j)): idx for idx, (i, j) in enumerate(pairs32)}
        used_r = set()
        for i, col in zip(row_ind, col_ind):
            i, col = int(i), int(col)
            if col < n_ref and mat[i, col] < float(init_big_m) * 0.5:
                if col in used_r:
                    continue
                used_r.add(col)
                v = pair_to_var.get((i, col))
                if v is not None:
                    chosen.append((i, col, int(v)))
            else:
                unmatched.add(i)
    return chosen, unmatched


# --------------------------------------------------------------------------- a14
def eager_signs(rxy, triangles, cand):
    tris = _c(triangles, np.int32).reshape(-1, 3)
    cand = _c(cand, np.int32)
    k = cand.shape[1]
    out = np.empty((len(tris), k, k, k), np.int8)
    lib().orc_eager_signs(_c(rxy, np.float64), tris, len(tris), cand.reshape(-1), k, out.reshape(-1))
    return out


# --------------------------------------------------------------------------- a13
def window_grid(ref_xy, mov_xy, window_size, overlap):
    """src/same.py:481-488.  The reference takes min / max of pandas columns, which skip NaN: nanmin / nanmax on the arrays here."""
    with np.errstate(all="ignore"):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            x_min = min(np.nanmin(ref_xy[:, 0]), np.nanmin(mov_xy[:, 0])); x_max = max(np.nanmax(ref_xy[:, 0]), np.nanmax(mov_xy[:, 0]))
            y_min = min(np.nanmin(ref_xy[:, 1]), np.nanmin(mov_xy[:, 1])); y_max = max(np.nanmax(ref_xy[:, 1]), np.nanmax(mov_xy[:, 1]))
    step = window_size - overlap
    return (list(range(int(x_min), int(x_max), step)), list(range(int(y_min), int(y_max), step)),
            (x_min, x_max, y_min, y_max))


def window_mask(xy, x0, x1, y0, y1):
    """src/same.py:293-295."""
    xy = _c(xy, np.float64)
    out = np.empty(len(xy), np.uint8)
    lib().orc_window_mask(xy, len(xy), float(x0), float(x1), float(y0), float(y1), out)
    return out.astype(bool)


def window_plan(ref_xy, mov_xy, window_size, overlap, min_cells):
    """src/same.py:509-582: the sequence of (i, j, box, central-trim box) the loop visits,
    including merge-right / merge-down and the skipped grid cells."""
    xs, ys, (x_min, x_max, y_min, y_max) = window_grid(ref_xy, mov_xy, window_size, overlap)
    plan = []
    i = 0
    while i < len(xs):
        j = 0
        while j < len(ys):
            x, y = xs[i], ys[j]
            i0, j0 = i, j
            x0, x1, y0, y1 = x, x + window_size, y, y + window_size
            nr = int(window_mask(ref_xy, x0, x1, y0, y1).sum()); nm = int(window_mask(mov_xy, x0, x1, y0, y1).sum())
            if nr < min_cells or nm < min_cells:
                if i + 1 < len(xs):
                    x1 = xs[i + 1] + window_size
                    nr = int(window_mask(ref_xy, x0, x1, y0, y1).sum()); nm = int(window_mask(mov_xy, x0, x1, y0, y1).sum())
                    if nr >= min_cells and nm >= min_cells:
                        i += 1
                if (nr < min_cells or nm < min_cells) and j + 1 < len(ys):
                    y1 = ys[j + 1] + window_size
                    nr = int(window_mask(ref_xy, x0, x1, y0, y1).sum()); nm = int(window_mask(mov_xy, x0, x1, y0, y1).sum())
                    if nr >= min_cells and nm >= min_cells:
                        j += 1
            if nr >= min_cells and nm >= min_cells:
                left, right = x == int(x_min), x1 >= int(x_max)
                top, bottom = y == int(y_min), y1 >= int(y_max)
                trim = (x0 if left else x0 + overlap / 2, x1 if right else x1 - overlap / 2,
                        y0 if top else y0 + overlap / 2, y1 if bottom else y1 - overlap / 2)
                plan.append({"i0": i0, "j0": j0, "i": i, "j": j, "window_id": len(xs) * j + i,
                             "box": (x0, x1, y0, y1), "trim": trim, "n_ref": nr, "n_mov": nm})
            j += 1
        i += 1
    return plan


# --------------------------------------------------------------------------- f1
def tri_flip_stats(axy, mapped_xy, matched, triangles, type_id=None):
    tris = _c(triangles, np.int32).reshape(-1, 3)
    n = len(axy)
    flag = np.empty(len(tris), np.uint8)
    nt, nf = np.empty(n, np.uint32), np.empty(n, np.uint32)
    tid = None if type_id is None else _c(type_id, np.int32)
    lib().orc_tri_flip_stats(_c(axy, np.float64), _c(mapped_xy, np.float64), _c(matched, np.uint8),
                             None if tid is None else tid.ctypes.data, tris, len(tris), n, flag, nt, nf)
    return flag, nt, nf


def check_triangle_violations(outputDF, mc_align, aligned_id_col='aligned_metacell_index', ref_id_col='matched_ref_index',
                              mapped_x_col='mapped_x', mapped_y_col='mapped_y', cell_type_col='cell_type',
                              ignore_same_type_triangles=True, node_local=False, majority_threshold=0.5, min_flips=1,
                              verbose=False):
    """src/eval_utils.py:66-223, literal id bookkeeping around the C triangle loop."""
    outputDF = outputDF.copy()
    triangles = np.asarray(mc_align.metacell_delaunay).reshape(-1, 3)
    mdf = mc_align.metacell_df
    pos = {v: i for i, v in enumerate(mdf.index)}
    row_of = {row[aligned_id_col]: idx for idx, row in outputDF.iterrows()}          # last row wins
    n = len(mdf)
    matched = np.zeros(n, np.uint8); mapped = np.zeros((n, 2)); type_id = np.full(n, -1, np.int32)
    types = {}
    for v, idx in row_of.items():
        if v in pos:
            p = pos[v]
            matched[p] = 1
            mapped[p] = (outputDF.loc[idx, mapped_x_col], outputDF.loc[idx, mapped_y_col])
            type_id[p] = types.setdefault(outputDF.loc[idx, cell_type_col], len(types))
    usable = [t for t in triangles if all(v in pos for v in t)]
    lost = [t for t in triangles if not all(v in pos for v in t)]
    tris = np.array([[pos[v] for v in t] for t in usable], dtype=np.int32).reshape(-1, 3)
    flag, nt, nf = tri_flip_stats(mdf[['X', 'Y']].to_numpy(dtype=np.float64), mapped, matched, tris,
                                  type_id if ignore_same_type_triangles else None)
    lost_matched = [t for t in lost if all(v in row_of for v in t)]
    lost_same = sum(1 for t in lost_matched if ignore_same_type_triangles and
                    len({outputDF.loc[row_of[v], cell_type_col] for v in t}) == 1)
    m, same, fl = (flag & 1).astype(bool), (flag & 2).astype(bool), (flag & 4).astype(bool)
    sign_flips = fl[m & ~same]
    node_viol = {}
    for x in outputDF[aligned_id_col].unique():
        n_tri = int(nt[pos[x]]) if x in pos else 0
        n_flip = int(nf[pos[x]]) if x in pos else 0
        if node_local:
            node_viol[x] = bool(n_tri > 0 and n_flip >= min_flips and (n_flip / n_tri) >= majority_threshold)
        else:
            node_viol[x] = n_flip > 0
    outputDF['in_violating_triangle'] = outputDF[aligned_id_col].map(node_viol).fillna(False)
    stats = {'total_triangles': len(triangles), 'triangles_with_all_matched': int(m.sum()) + len(lost_matched),
             'triangles_processed': int(m.sum()) + len(lost_matched), 'triangles_same_type_skipped': int(same.sum()) + lost_same,
             'triangles_flipped': int(np.sum(sign_flips)) if len(sign_flips) else 0,
             'percent_flipped': (100.0 * np.sum(sign_flips) / len(sign_flips) if len(sign_flips) else 0.0),
             'nodes_in_violating_triangles': int(outputDF['in_violating_triangle'].sum()),
             'percent_nodes_violating': 100.0 * outputDF['in_violating_triangle'].mean()}
    return outputDF, stats


# --------------------------------------------------------------------------- f2
def _mc_angle(p1, p2, p3):
    """compute_angle of src/metacell_utils.py:233-239 (no zero-length guard)."""
    v1, v2 = p1 - p2, p3 - p2
    with np.errstate(all="ignore"):
        c = np.dot(v1, v2) / (np.linalg.norm(v1) * np.linalg.norm(v2))
        return np.degrees(np.arccos(np.clip(c, -1, 1)))


def _mc_valid(tc, r_max, min_angle_deg):
    """is_triangle_valid, src/metacell_utils.py:241-260."""
    p1, p2, p3 = tc
    if r_max is not None and max(np.linalg.norm(p2 - p1), np.linalg.norm(p3 - p2), np.linalg.norm(p1 - p3)) > r_max:
        return False
    if min_angle_deg is not None and min(_mc_angle(p2, p1, p3), _mc_angle(p1, p2, p3), _mc_angle(p1, p3, p2)) < min_angle_deg:
        return False
    return True


def _mc_filter(coords, triangles, r_max, min_angle_deg):
    kept = [tri for tri in triangles if _mc_valid(coords[tri], r_max, min_angle_deg)]
    return np.array(kept) if kept else np.array([]).reshape(0, 3)


def greedy_triangle_collapse(aligned_df, max_metacell_size=3, max_iterations=1000, r_max=None, min_angle_deg=10, *,
                             original_idx_col="Cell_Num_Old", metacell_idx_col="metacell_id", x_col="X", y_col="Y",
                             cell_type_col="cell_type"):
    """src/metacell_utils.py:296-514 without the alpha shape -> (metacell_df, metacell_delaunay, original_delaunay)."""
    import pandas as pd
    from scipy.spatial import Delaunay

    aligned_df = aligned_df.copy()
    by_id = aligned_df.set_index(original_idx_col, drop=False)
    oc = aligned_df[[x_col, y_col]].to_numpy()
    ids = aligned_df[original_idx_col].to_numpy()
    opos = _mc_filter(oc, Delaunay(oc).simplices, r_max, min_angle_deg) if len(oc) >= 4 else np.array([], dtype=int).reshape(0, 3)
    original_delaunay = ids[opos.astype(int)] if opos.size else np.array([], dtype=ids.dtype).reshape(0, 3)
    id_cols = [c for c in aligned_df.columns if c in ["Cell_Num", "Cell_Num_Old", "cell_id", "Cell_ID", "ID", "id"]]
    if original_idx_col not in id_cols:
        id_cols.append(original_idx_col)
    if metacell_idx_col in aligned_df.columns and metacell_idx_col not in id_cols:
        id_cols.append(metacell_idx_col)
    rows = []
    for _, row in aligned_df.iterrows():
        rows.append({x_col: row[x_col], y_col: row[y_col], cell_type_col: row[cell_type_col], "size": 1,
                     "members": [row[original_idx_col]],
                     **{c: row[c] for c in aligned_df.columns if c not in [x_col, y_col, cell_type_col] + id_cols}})
    mdf = pd.DataFrame(rows)
    mdf[metacell_idx_col] = range(len(mdf))
    for _ in range(max_iterations):
        coords = mdf[[x_col, y_col]].values
        if len(coords) < 4:
            break
        tris = _mc_filter(coords, Delaunay(coords).simplices, r_max, min_angle_deg)
        if len(tris) == 0:
            break
        cands = []
        ctype, csize = mdf[cell_type_col].to_numpy(), mdf["size"].to_numpy()
        for tri in tris:
            a, b, c = tri
            if not (ctype[a] == ctype[b] == ctype[c]):
                continue
            total = csize[a] + csize[b] + csize[c]
            if total > max_metacell_size:
                continue
            per = np.linalg.norm(coords[a] - coords[b]) + np.linalg.norm(coords[b] - coords[c]) + np.linalg.norm(coords[c] - coords[a])
            cands.append((per, [a, b, c], total))
        if not cands:
            break
        cands.sort(key=lambda q: q[0])
        used, batch = set(), []
        for cand in cands:
            a, b, c = cand[1]
            if a not in used and b not in used and c not in used:
                batch.append(cand)
                used.update([a, b, c])
        merged, drop = [], []
        for per, (a, b, c), total in batch:
            drop.extend([a, b, c])
            members = mdf.iloc[a]["members"] + mdf.iloc[b]["members"] + mdf.iloc[c]["members"]
            mc = by_id.loc[members, [x_col, y_col]]
            new = {x_col: mc[x_col].mean(), y_col: mc[y_col].mean(), cell_type_col: mdf.iloc[a][cell_type_col], "size": total,
                   "members": members}
            for col in mdf.columns:
                if col in [x_col, y_col, cell_type_col, "size", "members", metacell_idx_col] + id_cols:
                    continue
                new[col] = by_id.loc[members, col].mean() if pd.api.types.is_numeric_dtype(mdf[col]) else mdf.iloc[a][col]
            merged.append(new)
        mdf = mdf.drop(drop).reset_index(drop=True)
        if merged:
            mdf = pd.concat([mdf, pd.DataFrame(merged)], ignore_index=True)
        mdf[metacell_idx_col] = range(len(mdf))
    fc = mdf[[x_col, y_col]].values
    final = _mc_filter(fc, Delaunay(fc).simplices, r_max, min_angle_deg) if len(fc) >= 4 else np.array([]).reshape(0, 3)
    return mdf, final, original_delaunay


def batched_assign(a_off, r_off, axy, rxy):
    """f4 (src/metacell_utils.py:711-761): one small assignment per metacell match; CSR member lists.
    -> int32 local ref-member index per aligned member."""
    a_off, r_off = _c(a_off, np.int64), _c(r_off, np.int64)
    axy, rxy = _c(np.asarray(axy, np.float64).reshape(-1, 2), np.float64), _c(np.asarray(rxy, np.float64).reshape(-1, 2), np.float64)
    out = np.full(int(a_off[-1]), -1, np.int32)
    rc = lib().orc_batched_assign(len(a_off) - 1, a_off, r_off, axy.reshape(-1), rxy.reshape(-1), out)
    if rc:
        raise RuntimeError(f"orc_batched_assign failed ({rc})")
    return out


def batched_assign_scipy(a_off, r_off, axy, rxy):
    """The literal reference form of the same thing (cdist + np.tile + linear_sum_assignment per match)."""
    from scipy.optimize import linear_sum_assignment
    from scipy.spatial.distance import cdist

    axy, rxy = np.asarray(axy, np.float64).reshape(-1, 2), np.asarray(rxy, np.float64).reshape(-1, 2)
    out = np.full(int(a_off[-1]), -1, np.int32)
    for p in range(len(a_off) - 1):
        a, r = axy[a_off[p]:a_off[p + 1]], rxy[r_off[p]:r_off[p + 1]]
        if len(a) == 0:
            continue
        d = cdist(a, r)
        if len(a) > len(r):
            d = np.tile(d, (1, int(np.ceil(len(a) / len(r)))))
        rows, cols = linear_sum_assignment(d)
        out[a_off[p] + rows] = cols % len(r)
    return out


def merge_dedup(viol, window_id, aligned_code, ref_code):
    """f3, the de-duplication step of merge_window_matches_unique_ref (src/helpers.py:745-753): stable sort by (violation,
    window id), first row of every (aligned, ref) pair.  -> surviving row indices in the sorted order (int32)."""
    viol = _c(np.asarray(viol).astype(bool), np.uint8)
    w, a, r = _c(window_id, np.int32), _c(aligned_code, np.int32), _c(ref_code, np.int32)
    out = np.empty(max(len(viol), 1), np.int32)
    m = lib().orc_merge_dedup(viol, w, a, r, len(viol), out)
    if m < 0:
        raise RuntimeError(f"orc_merge_dedup failed ({m})")
    return out[:m].copy()


def merge_dedup_pandas(viol, window_id, aligned_code, ref_code):
    """The literal reference form of the same step (pandas mergesort + drop_duplicates, src/helpers.py:748-753)."""
    df = pd.DataFrame({"v": np.asarray(viol).astype(bool), "w": np.asarray(window_id), "a": np.asarray(aligned_code), "r": np.asarray(ref_code)})
    df = df.sort_values(by=["v", "w"], ascending=[True, True], kind="mergesort").drop_duplicates(subset=["a", "r"], keep="first")
    return df.index.to_numpy().astype(np.int32)
