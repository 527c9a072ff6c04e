"""cProfile of merge_window_matches_unique_ref on the table sliding_window_incumbent makes for a synthetic section: where the serial tail
of a `bench.py --workload cfg5` step goes.  Usage: python3 tools/merge_profile.py [cells=1000000]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import same_amd                                    # noqa: E402
from same_amd import _trace, synth                 # noqa: E402
from same_amd.merge import merge_window_matches_unique_ref   # noqa: E402

n, T = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 8
ref = synth.make_cells(n, T, seed=0)
mov = synth.make_jittered(ref, seed=1)
r_df, m_df = synth.to_frame(ref), synth.to_frame(mov)
r_df["Cell_Num_Old"], m_df["Cell_Num_Old"] = np.arange(len(r_df)), np.arange(len(m_df))
cols = synth.type_columns(T)
op = dict(radius=25, knn=8, no_match_penalty=100, hip_cost_dtype="float32", window_size=1200, overlap=300, min_cells_per_window=10)
with same_amd.resident_frames(r_df, m_df) as res:
    table = same_amd.sliding_window_incumbent(res, res, commonCT=cols, optim_params=dict(op))
print(f"{len(table)} rows x {len(table.columns)} columns, {table['window_id'].nunique()} windows", flush=True)
merge_window_matches_unique_ref([table])
_trace.enable()
for _ in range(3):
    _trace.reset()
    t0 = time.perf_counter()
    out = merge_window_matches_unique_ref([table])
    dt = time.perf_counter() - t0
    print(f"merge: {dt * 1e3:.1f} ms -> {len(out)} rows; " + ", ".join(f"{k.split(': ')[-1]} {v[1] * 1e3:.1f}" for k, v in _trace.report().items()), flush=True)
pr = cProfile.Profile()
pr.enable()
merge_window_matches_unique_ref([table])
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
