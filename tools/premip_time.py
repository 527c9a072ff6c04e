"""Wall time of the whole pre-MIP path (prepare_same_inputs) at BASELINE cfg2 / cfg3 / cfg4 shapes, by stage
(same_amd._trace stage markers; under `rocprofv3 --marker-trace --kernel-trace` the same stages appear as rocTX ranges).
Usage: python tools/premip_time.py [n_cells]"""
import os, sys, time
os.environ.setdefault("SAME_TRACE", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import same_amd
from same_amd import _trace, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
T, k = 20, 32
ref = synth.make_cells(n, T, seed=0); mov = synth.make_jittered(ref, seed=1)
r_df, m_df = synth.to_frame(ref), synth.to_frame(mov)
cols = synth.type_columns(T)
op = dict(radius=25, knn=k)
same_amd.prepare_same_inputs(r_df.iloc[:2000], m_df.iloc[:2000], cols, optim_params=op, verbose=False)  # warm up library/context
_trace.reset()
t = time.perf_counter()
prep = same_amd.prepare_same_inputs(r_df, m_df, cols, optim_params=op, verbose=False)
dt = time.perf_counter() - t
print(f"n={n}: prepare_same_inputs {dt:.3f} s  pairs={len(prep.valid_pairs)} triangles={len(prep.triangles_array)}")
for name, (calls, sec) in sorted(_trace.report().items(), key=lambda kv: -kv[1][1]):
    print(f"  {name:48s} {sec * 1e3:9.2f} ms  ({100 * sec / dt:4.1f} %)")
