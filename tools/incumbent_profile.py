"""cProfile of same_amd.sliding_window_incumbent (one worker thread, resident frames) on a synthetic section: where the host's time
per window goes in the product function bench.py --workload cfg5 times.  Usage: python3 tools/incumbent_profile.py [cells=400000]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import same_amd                                    # noqa: E402
from same_amd import synth                         # noqa: E402

n, T = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000, 8
ref = synth.make_cells(n, T, seed=0)
mov = synth.make_jittered(ref, seed=1)
r_df, m_df = synth.to_frame(ref), synth.to_frame(mov)
r_df["Cell_Num_Old"], m_df["Cell_Num_Old"] = np.arange(len(r_df)), np.arange(len(m_df))
cols = synth.type_columns(T)
op = dict(radius=25, knn=8, no_match_penalty=100, hip_cost_dtype="float32", window_size=1200, overlap=300, min_cells_per_window=10)
with same_amd.resident_frames(r_df, m_df) as res:
    same_amd.sliding_window_incumbent(res, res, commonCT=cols, optim_params=dict(op), workers=1)
    for workers in (1, 2):
        t0 = time.perf_counter()
        out = same_amd.sliding_window_incumbent(res, res, commonCT=cols, optim_params=dict(op), workers=workers)
        dt = time.perf_counter() - t0
        nw = out["window_id"].nunique()
        print(f"workers {workers}: {nw} windows in {dt:.3f} s = {nw / dt:.0f} windows/s, {dt / nw * 1e3:.2f} ms per window, {len(out)} rows")
    pr = cProfile.Profile()
    pr.enable()
    same_amd.sliding_window_incumbent(res, res, commonCT=cols, optim_params=dict(op), workers=1)
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
