"""BASELINE cfg 5 shape: a 1M-cell section tiled into overlapping windows; per window the whole pre-MIP path
(prune + compaction, Delaunay, triangle filter, weights/signs, pair costs, greedy start, orientation / XY-order sweeps
under that start) through the Python boundary.  Reports windows/s and cells/s on one GPU (windows are independent:
N GPUs take the plan round-robin, same_amd.windows.assign_windows)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import same_amd
from same_amd import synth
from same_amd.windows import window_plan

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
max_windows = int(sys.argv[2]) if len(sys.argv) > 2 else 12
T = 8
ref = synth.make_cells(n, T, seed=0); mov = synth.make_jittered(ref, seed=1)
r_df, m_df = synth.to_frame(ref), synth.to_frame(mov)
cols = synth.type_columns(T)
t = time.perf_counter()
plan = window_plan(ref["xy"], mov["xy"], 1200, 300, 10)
t_plan = time.perf_counter() - t
print(f"{n} cells: {len(plan)} windows planned in {t_plan*1e3:.1f} ms (one batched device count pass per section)")
op = dict(radius=25, knn=8, no_match_penalty=100)
done = cells = pairs = tris = 0
t = time.perf_counter()
for w in plan[:: max(1, len(plan) // max_windows)][:max_windows]:
    x0, x1, y0, y1 = w["box"]
    rs, ms = same_amd.subset_data(r_df, x0, x1, y0, y1), same_amd.subset_data(m_df, x0, x1, y0, y1)
    prep = same_amd.prepare_same_inputs(rs, ms, cols, optim_params=op, verbose=False)
    ch, un = same_amd.compute_mip_start_pairs(valid_pairs=prep.valid_pairs, costs=prep.costs, n_aligned=prep.n_aligned, n_ref=prep.n_ref,
                                              aligned_sizes=prep.aligned_df["size"].to_numpy(dtype=float), no_match_penalty=100,
                                              max_matches=1, init_method="greedy", verbose=False)
    x = np.zeros(len(prep.valid_pairs)); x[[c[2] for c in ch]] = 1.0
    sw = same_amd.LazyOrientationSweep(prep.valid_pairs, prep.aligned_delaunay, prep.source_signs, prep.ref_df[["X", "Y"]].to_numpy(), prep.n_aligned)
    checked, viol, _ = sw.sweep(x)
    done += 1; cells += prep.n_aligned; pairs += len(prep.valid_pairs); tris += len(prep.aligned_delaunay)
dt = time.perf_counter() - t
print(f"{done} windows: {dt/done*1e3:.1f} ms/window, {cells/dt:.3e} aligned cells/s, {pairs/dt:.3e} pairs/s, {tris/dt:.3e} triangles/s "
      f"(avg {cells//done} cells, {pairs//done} pairs, {tris//done} triangles per window)")
