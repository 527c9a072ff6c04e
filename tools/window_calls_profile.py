"""cProfile of windows.iter_device_windows with the triangulations remembered (what bench.py reports as `window_calls_only`): where the
host's share of a window goes once Qhull is out of the picture.  Usage: python3 tools/window_calls_profile.py [cells=300000]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from same_amd import synth                         # noqa: E402
from same_amd import windows as W                  # noqa: E402

n, T = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000, 8
ref = synth.make_cells(n, T, seed=0)
mov = synth.make_jittered(ref, seed=1)
r_df, m_df = synth.to_frame(ref), synth.to_frame(mov)
cols = synth.type_columns(T)
plan = W.window_plan(ref["xy"], mov["xy"], 1200, 300, 10)
xs, ys, _ = W.window_grid(ref["xy"], mov["xy"], 1200, 300)
grid = W.window_cell_grid((xs, ys), 1200, 300)
ref_sec, mov_sec = W.Section.from_frame(r_df, cols), W.Section.from_frame(m_df, cols)
dref, dmov = W.DeviceSection(ref_sec, "float32").bin(*grid), W.DeviceSection(mov_sec, "float32").bin(*grid)
cache = W.TriangulationCache()
kw = dict(radius=25, knn=8, dist_ct_coeff=1.0, min_angle_deg=15, ignore_same_type_triangles=True, no_match_penalty=100.0, triangulator=cache)
for _ in range(2):
    list(W.iter_device_windows(ref_sec, mov_sec, dref, dmov, plan, **kw))
for batch in (1, 4, 8, 16, 32):
    t0 = time.perf_counter()
    for _ in range(3):
        for _dw in W.iter_device_windows(ref_sec, mov_sec, dref, dmov, plan, batch=batch, **kw):
            pass
    dt = (time.perf_counter() - t0) / 3
    print(f"batch {batch:2d}: {len(plan)} windows in {dt * 1e3:.1f} ms = {len(plan) / dt:.0f} windows/s, {dt / len(plan) * 1e3:.3f} ms per window")
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    for _dw in W.iter_device_windows(ref_sec, mov_sec, dref, dmov, plan, **kw):
        pass
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
