"""Wall time of greedy_triangle_collapse by stage, on a seeded frame of n cells."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from same_amd import synth
from same_amd.metacell_utils import greedy_triangle_collapse

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
size = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cells = synth.make_cells(n, 6, seed=3)
df = synth.to_frame(cells)
df["cell_type"] = np.where(np.arange(n) % 7 < 5, "A", df["cell_type"])   # mostly one type so that many triangles collapse
greedy_triangle_collapse(df.iloc[:3000], max_metacell_size=size, r_max=40, min_angle_deg=10, verbose=False)  # warm up
t = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
mdf, tri = greedy_triangle_collapse(df, max_metacell_size=size, r_max=40, min_angle_deg=10, verbose=False)
pr.disable()
dt = time.perf_counter() - t
print(f"n={n}: greedy_triangle_collapse {dt:.3f} s -> {len(mdf)} metacells, {len(tri)} triangles")
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
