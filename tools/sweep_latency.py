"""Host-visible latency of the per-incumbent orientation sweep (what the Gurobi callback pays) at BASELINE cfg 3 scale."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scipy.spatial import Delaunay
from same_amd import ops, synth, knn, sweeps

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
ref = synth.make_cells(n, 20, seed=0); mov = synth.make_jittered(ref, seed=1)
idx, _, cnt = ops.knn_prune(mov["xy"], ref["xy"], 25.0, 32)
pairs = knn.pairs_from_padded(idx)
tris = Delaunay(mov["xy"]).simplices
sign, _ = ops.tri_sign_weight(mov["xy"], None, tris)
sw = sweeps.LazyOrientationSweep(pairs, tris, sign, ref["xy"], len(mov["xy"]))
x = np.zeros(len(pairs)); first = np.flatnonzero(np.r_[True, pairs[1:, 0] != pairs[:-1, 0]]); x[first] = 1.0
for _ in range(3): sw.bound.sweep_x(x)
t = time.perf_counter(); N = 50
for _ in range(N): checked, viol, match, pidx = sw.bound.sweep_x(x)
dt = (time.perf_counter() - t) / N
m = np.where(cnt > 0, idx[:, 0], -1).astype(np.int32)
for _ in range(3): sw.bound.sweep_match(m)
t = time.perf_counter()
for _ in range(N): sw.bound.sweep_match(m)
dt2 = (time.perf_counter() - t) / N
t = time.perf_counter()
for _ in range(10): ops.xyorder_sweep(mov["xy"], ref["xy"], tris, m)
dt3 = (time.perf_counter() - t) / 10
print(f"cells {len(mov['xy'])} pairs {len(pairs)} triangles {len(tris)} checked {checked} flipped {len(viol)}")
print(f"orientation sweep from x (H2D {len(pairs)*8/1e6:.1f} MB + 5 kernels + read-back): {dt*1e3:.3f} ms per incumbent")
print(f"orientation sweep from match vector: {dt2*1e3:.3f} ms;  xy-order sweep host-buffer call: {dt3*1e3:.3f} ms")
print(f"reference CPython loop at 3.7 us/triangle would be {len(tris)*3.7e-3:.0f} ms per incumbent")
