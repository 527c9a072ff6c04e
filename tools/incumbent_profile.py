"""cProfile of same_amd.sliding_window_incumbent (resident frames) on a synthetic section, with Qhull in the picture and with the
triangulations remembered: where the host's time per window goes in the product function bench.py --workload cfg5 times.
Usage: python3 tools/incumbent_profile.py [cells=1000000] [merge=1]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import same_amd                                    # noqa: E402
from same_amd import _trace, synth                 # noqa: E402
from same_amd import windows as W                  # noqa: E402

n, T = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 8
merge = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
ref = synth.make_cells(n, T, seed=0)
mov = synth.make_jittered(ref, seed=1)
r_df, m_df = synth.to_frame(ref), synth.to_frame(mov)
r_df["Cell_Num_Old"], m_df["Cell_Num_Old"] = np.arange(len(r_df)), np.arange(len(m_df))
cols = synth.type_columns(T)
op = dict(radius=25, knn=8, no_match_penalty=100, hip_cost_dtype="float32", window_size=1200, overlap=300, min_cells_per_window=10)
_trace.enable(True)
with same_amd.resident_frames(r_df, m_df) as res:
    call = lambda **kw: same_amd.sliding_window_incumbent(res, res, commonCT=cols, optim_params=dict(op), merge=merge, return_stats=True, **kw)
    call(workers=2)
    for workers in (1, 2):
        t0 = time.perf_counter()
        out, stats = call(workers=workers)
        dt = time.perf_counter() - t0
        print(f"workers {workers}: {len(stats)} windows in {dt:.3f} s = {len(stats) / dt:.0f} windows/s, {dt / len(stats) * 1e3:.2f} ms per window, {len(out)} rows")
    cache = W.TriangulationCache()
    call(workers=2, triangulator=cache)
    for workers in (1, 2):
        _trace.reset()
        t0 = time.perf_counter()
        for _ in range(3):
            out, stats = call(workers=workers, triangulator=cache)
        dt = (time.perf_counter() - t0) / 3
        print(f"triangulations given, workers {workers}: {len(stats)} windows in {dt * 1e3:.1f} ms = {len(stats) / dt:.0f} windows/s")
        for name, (c, sec) in sorted(_trace.report().items(), key=lambda e: -e[1][1])[:14]:
            print(f"    {sec / 3 * 1e3:8.2f} ms  {c // 3:5d} x  {name}")
    pr = cProfile.Profile()
    pr.enable()
    call(workers=1, triangulator=cache)
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(30)
