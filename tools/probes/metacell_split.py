import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scipy.spatial as sp
from same_amd import synth
from same_amd.metacell_utils import greedy_triangle_collapse
T = {"qhull": 0.0, "n": 0}
_D = sp.Delaunay
class Timed(_D):
    def __init__(self, *a, **k):
        t = time.perf_counter(); super().__init__(*a, **k); T["qhull"] += time.perf_counter() - t; T["n"] += 1
sp.Delaunay = Timed
n = 100000
cells = synth.make_cells(n, 6, seed=3); df = synth.to_frame(cells)
df["cell_type"] = np.where(np.arange(n) % 7 < 5, "A", df["cell_type"])
greedy_triangle_collapse(df.iloc[:3000], max_metacell_size=5, r_max=40, min_angle_deg=10, verbose=False)
T["qhull"] = 0; T["n"] = 0
t = time.perf_counter()
mdf, tri = greedy_triangle_collapse(df, max_metacell_size=5, r_max=40, min_angle_deg=10, verbose=False)
dt = time.perf_counter() - t
print(f"total {dt:.3f} s, Qhull {T['qhull']:.3f} s in {T['n']} calls, rest {dt - T['qhull']:.3f} s")
