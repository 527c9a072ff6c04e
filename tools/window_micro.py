"""Where one cfg-5 window's host milliseconds go on the column pipeline: every sub-operation of windows.iter_window_arrays timed
alone, 100 repetitions on one mid-section window of a 1M-cell section.  Usage: python tools/window_micro.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scipy.spatial import Delaunay
from same_amd import ops, synth
from same_amd.knn import pairs_from_padded
from same_amd.triangles import classify_triangles, filter_triangles_by_radius
from same_amd.windows import Section, window_plan

T = 8
ref = synth.make_cells(1_000_000, T, seed=0); mov = synth.make_jittered(ref, seed=1)
cols = synth.type_columns(T)
rs = Section(ref["xy"], ref["types"], np.unique(ref["cell_type"], return_inverse=True)[1], None)
ms = Section(mov["xy"], mov["types"], np.unique(mov["cell_type"], return_inverse=True)[1], None)
plan = window_plan(ref["xy"], mov["xy"], 1200, 300, 10)
w = plan[len(plan) // 2]
REP = 100


def timeit(name, fn):
    fn()
    t0 = time.perf_counter()
    for _ in range(REP):
        out = fn()
    dt = (time.perf_counter() - t0) / REP * 1e3
    print(f"{name:58s} {dt:7.3f} ms", flush=True)
    return out


rows_r = timeit("GridRows.rows(ref)", lambda: rs.grid.rows(*w["box"]))
rows_m = timeit("GridRows.rows(moving)", lambda: ms.grid.rows(*w["box"]))
axy = timeit("moving.xy[rows_m]", lambda: ms.xy[rows_m]); rxy = rs.xy[rows_r]
idx = timeit("ops.knn_prune (host buffers, H2D + kernel + D2H)", lambda: ops.knn_prune(axy, rxy, 25.0, 8, want_d2=False))[0]
kp = timeit("pairs_from_padded", lambda: pairs_from_padded(idx))


def compact():
    ua_ = np.zeros(len(axy), bool); ua_[kp[:, 0]] = True
    ur_ = np.zeros(len(rxy), bool); ur_[kp[:, 1]] = True
    ua, ur = np.flatnonzero(ua_), np.flatnonzero(ur_)
    pairs = np.column_stack(((np.cumsum(ua_) - 1)[kp[:, 0]], (np.cumsum(ur_) - 1)[kp[:, 1]])).astype(np.int64)
    return ua, ur, pairs, np.ascontiguousarray(axy[ua]), np.ascontiguousarray(rxy[ur])


ua, ur, pairs, caxy, crxy = timeit("compaction (masks, cumsum, column_stack, take xy)", compact)
rm, rr = rows_m[ua], rows_r[ur]
tris = timeit("scipy Delaunay in-process (what the helpers hide)", lambda: Delaunay(caxy).simplices)
tid = ms.type_id[rm].astype(np.int32)
timeit("classify_triangles (kernel + 8-ulp recheck)", lambda: classify_triangles(caxy, tris, 25.0, 15, tid))
ftris = timeit("filter_triangles_by_radius (classify + host re-add pass)", lambda: filter_triangles_by_radius(
    caxy, tris, 25.0, ignore_same_type_triangles=True, min_angle_deg=15, verbose=False, _rows_as_array=True, _type_id=tid))
size = np.ones(len(rm))
timeit("ops.tri_sign_weight", lambda: ops.tri_sign_weight(caxy, size, ftris))
A, R = timeit("types[rows] x2", lambda: (ms.types[rm], rs.types[rr]))
timeit("ops.pair_cost f32 (conversions + H2D + kernel + D2H)", lambda: ops.pair_cost(A, R, caxy, crxy, pairs, 1.0, dtype=np.float32))
A32, R32, a32, r32, p32 = A.astype(np.float32), R.astype(np.float32), caxy.astype(np.float32), crxy.astype(np.float32), pairs.astype(np.int32)
timeit("ops.pair_cost f32, operands already float32 / int32", lambda: ops.pair_cost(A32, R32, a32, r32, p32, 1.0, dtype=np.float32))
costs = ops.pair_cost(A, R, caxy, crxy, pairs, 1.0, dtype=np.float32).astype(np.float64)
timeit("ops.pair_rowmin", lambda: ops.pair_rowmin(p32, costs, len(rm)))
wants = ops.pair_rowmin(p32, costs, len(rm)) < 100.0
por = timeit("ops.greedy_match", lambda: ops.greedy_match(p32, costs, len(rm), len(rr), wants))[0]
ai = np.flatnonzero(por >= 0); match = np.full(len(rm), -1, np.int32); match[ai] = p32[por[ai], 1]
sign = ops.tri_sign_weight(caxy, size, ftris)[0]


def sweep():
    sw = ops.BoundSweep(ftris, sign, crxy, len(rm)); out = sw.sweep_match(match); sw.close(); return out


timeit("BoundSweep bind + sweep_match + close", sweep)
timeit("ops.xyorder_sweep", lambda: ops.xyorder_sweep(caxy, crxy, ftris, match))
timeit("ops.area_flip", lambda: ops.area_flip(caxy, crxy, ftris, match))
print(f"window: {len(rows_m)} aligned x {len(rows_r)} ref cells, {len(pairs)} pairs, {len(tris)} -> {len(ftris)} triangles")
