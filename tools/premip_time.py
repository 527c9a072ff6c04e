"""Wall time of the whole pre-MIP path (prepare_same_inputs) at BASELINE cfg2 / cfg3 shapes, by stage."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import same_amd
from same_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
T, k = 20, 32
ref = synth.make_cells(n, T, seed=0); mov = synth.make_jittered(ref, seed=1)
r_df, m_df = synth.to_frame(ref), synth.to_frame(mov)
cols = synth.type_columns(T)
op = dict(radius=25, knn=k)
same_amd.prepare_same_inputs(r_df.iloc[:2000], m_df.iloc[:2000], cols, optim_params=op, verbose=False)  # warm up library/context
t = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
prep = same_amd.prepare_same_inputs(r_df, m_df, cols, optim_params=op, verbose=False)
pr.disable()
dt = time.perf_counter() - t
print(f"n={n}: prepare_same_inputs {dt:.3f} s  pairs={len(prep.valid_pairs)} triangles={len(prep.aligned_delaunay)}")
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
