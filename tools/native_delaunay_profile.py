"""The window path with libsame_hip's own triangulator (optim_params["hip_delaunay"] = "native"): windows/s of
same_amd.sliding_window_incumbent on resident frames for several numbers of worker threads and triangulator threads, the stage
times of the best, and the triangulator alone against scipy on one window's points.
Usage: python3 tools/native_delaunay_profile.py [cells=1000000]"""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import same_amd                                    # noqa: E402
from same_amd import _trace, delaunay, synth       # noqa: E402

n, T = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 8
ref = synth.make_cells(n, T, seed=0)
mov = synth.make_jittered(ref, seed=1)
r_df, m_df = synth.to_frame(ref), synth.to_frame(mov)
r_df["Cell_Num_Old"], m_df["Cell_Num_Old"] = np.arange(len(r_df)), np.arange(len(m_df))
cols = synth.type_columns(T)
op = dict(radius=25, knn=8, no_match_penalty=100, hip_cost_dtype="float32", window_size=1200, overlap=300, min_cells_per_window=10)

# the triangulator alone: one window's worth of points
xy = m_df[["X", "Y"]].to_numpy()
box = xy[(xy[:, 0] > 3000) & (xy[:, 0] < 4200) & (xy[:, 1] > 3000) & (xy[:, 1] < 4200)]
from scipy.spatial import Delaunay                 # noqa: E402

for name, fn in (("same_delaunay2d", lambda: delaunay.native_simplices(box)), ("scipy.spatial.Delaunay", lambda: Delaunay(box).simplices)):
    fn()
    best = float("inf")
    for _ in range(5):                                   # the best of five batches: the first ones may run on a core that is still clocking up
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        best = min(best, (time.perf_counter() - t0) / 10)
    print(f"{name}: {len(box)} points in {best * 1e3:.2f} ms (one thread, best of 5 batches of 10)")
for threads in (4, 8, 16):
    with ThreadPoolExecutor(threads) as ex:
        t0 = time.perf_counter()
        list(ex.map(lambda _q: delaunay.native_simplices(box), range(threads * 8)))
        dt = time.perf_counter() - t0
    print(f"same_delaunay2d on {threads} threads: {dt / (threads * 8) * 1e3:.2f} ms per set of the batch ({threads * 8 / dt:.0f} sets/s)")

_trace.enable(True)
with same_amd.resident_frames(r_df, m_df) as res:
    want = same_amd.sliding_window_incumbent(res, res, commonCT=cols, optim_params=dict(op), merge=True, workers=2)
    best = (0.0, None)
    for tri_threads in (8, 12, 16, 24):
        tr = delaunay.NativeTriangulator(threads=tri_threads)
        for workers in (1, 2, 3, 4):
            call = lambda: same_amd.sliding_window_incumbent(res, res, commonCT=cols, optim_params=dict(op), merge=True, return_stats=True,
                                                             workers=workers, triangulator=tr)
            out, stats = call()
            assert out.equals(want)
            _trace.reset()
            t0 = time.perf_counter()
            for _ in range(3):
                call()
            dt = (time.perf_counter() - t0) / 3
            rate = len(stats) / dt
            print(f"triangulator threads {tri_threads:2d}, workers {workers}: {len(stats)} windows in {dt * 1e3:.1f} ms = {rate:.0f} "
                  f"windows/s; "
                  f"sent back to Qhull {tr.asked_qhull} of {tr.submitted}", flush=True)
            if rate > best[0]:
                best = (rate, (tri_threads, workers, {k: v for k, v in _trace.report().items()}))
        tr.close()
    tri_threads, workers, rep = best[1]
    print(f"stages of the best ({tri_threads} triangulator threads, {workers} workers), per pass, summed over the workers:")
    for name, (c, sec) in sorted(rep.items(), key=lambda e: -e[1][1])[:16]:
        print(f"    {sec / 3 * 1e3:8.2f} ms  {c // 3:5d} x  {name}")
