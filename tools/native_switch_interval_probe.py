"""Does the interpreter's thread switch interval matter to the opt-in triangulator route?  A thread that returns from a ctypes call
must take the GIL back; while another thread runs Python it may wait for up to the switch interval (5 ms by default), and the route
has 16-24 triangulator threads and 2-4 worker threads doing exactly that.  Same job as tools/native_delaunay_profile.py, 24
triangulator threads, 2 and 3 workers, for several switch intervals, interleaved.  Usage: python3 tools/native_switch_interval_probe.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import same_amd                                    # noqa: E402
from same_amd import delaunay, synth               # noqa: E402

n, T = 1_000_000, 8
ref = synth.make_cells(n, T, seed=0)
mov = synth.make_jittered(ref, seed=1)
r_df, m_df = synth.to_frame(ref), synth.to_frame(mov)
r_df["Cell_Num_Old"], m_df["Cell_Num_Old"] = np.arange(len(r_df)), np.arange(len(m_df))
cols = synth.type_columns(T)
op = dict(radius=25, knn=8, no_match_penalty=100, hip_cost_dtype="float32", window_size=1200, overlap=300, min_cells_per_window=10)
default = sys.getswitchinterval()
with same_amd.resident_frames(r_df, m_df) as res:
    tr = delaunay.NativeTriangulator(threads=24)
    call = lambda workers: same_amd.sliding_window_incumbent(res, res, commonCT=cols, optim_params=dict(op), merge=True, return_stats=True,
                                                             workers=workers, triangulator=tr)
    call(2)
    for rep in range(3):
        for interval in (default, 1e-3, 2e-4, 5e-5):
            sys.setswitchinterval(interval)
            for workers in (2, 3):
                call(workers)
                t0 = time.perf_counter()
                for _ in range(3):
                    out, stats = call(workers)
                dt = (time.perf_counter() - t0) / 3
                print(f"switch interval {interval * 1e3:5.2f} ms, workers {workers}: {len(stats) / dt:6.0f} windows/s", flush=True)
    sys.setswitchinterval(default)
    tr.close()
