"""Where a window of the reference's signature goes on the device pipeline (bench.py's `api_path`, part (a)): cProfile of
sliding_window_matching on a 300k-cell section with `incumbent_of_prepared` standing in for the solver half.
Usage: python3 tools/api_path_profile.py [cells=300000]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import same_amd                                    # noqa: E402
from same_amd import synth                         # noqa: E402
from same_amd.incumbent import incumbent_of_prepared   # noqa: E402

n, T = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000, 8
ref = synth.make_cells(n, T, seed=0)
mov = synth.make_jittered(ref, seed=1)
r_df, m_df = synth.to_frame(ref), synth.to_frame(mov)
r_df["Cell_Num_Old"], m_df["Cell_Num_Old"] = np.arange(len(r_df)), np.arange(len(m_df))
cols = synth.type_columns(T)
op = dict(radius=25, knn=8, no_match_penalty=100, hip_cost_dtype="float32", window_size=1200, overlap=300, min_cells_per_window=10)
solve = lambda prep, _o: (incumbent_of_prepared(prep, cols, True)[0], {})
with same_amd.resident_frames(r_df, m_df) as res:
    same_amd.sliding_window_matching(res, res, commonCT=cols, optim_params=dict(op), _solve=solve)      # warm: buffers, helpers
    t0 = time.perf_counter()
    out = same_amd.sliding_window_matching(res, res, commonCT=cols, optim_params=dict(op), _solve=solve)
    dt = time.perf_counter() - t0
    nw = out["window_id"].nunique()
    print(f"{nw} windows in {dt:.3f} s = {nw / dt:.0f} windows/s, {dt / nw * 1e3:.2f} ms per window, {len(out)} matches")
    pr = cProfile.Profile()
    pr.enable()
    same_amd.sliding_window_matching(res, res, commonCT=cols, optim_params=dict(op), _solve=solve)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumtime").print_stats(45)
