"""Wall time of greedy_triangle_collapse on a seeded frame of n cells: with the Qhull helpers (triangulations
of the coming iteration started while the frame merge finishes) and without (SAME_QHULL_WORKERS=0 behaviour), plus the time
of the Qhull calls a strictly serial run would make."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    from scipy.spatial import Delaunay
    from same_amd import qhull_pool, synth
    from same_amd.metacell_utils import greedy_triangle_collapse

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    cells = synth.make_cells(n, 6, seed=3)
    df = synth.to_frame(cells)
    df["cell_type"] = np.where(np.arange(n) % 7 < 5, "A", df["cell_type"])   # mostly one type so that many triangles collapse
    greedy_triangle_collapse(df.iloc[:3000], max_metacell_size=size, r_max=40, min_angle_deg=10, verbose=False)  # warm up
    t = time.perf_counter()
    mdf, tri = greedy_triangle_collapse(df, max_metacell_size=size, r_max=40, min_angle_deg=10, verbose=False)
    dt = time.perf_counter() - t
    t = time.perf_counter(); Delaunay(df[["X", "Y"]].values); t_q0 = time.perf_counter() - t
    t = time.perf_counter(); Delaunay(mdf[["X", "Y"]].values); t_q1 = time.perf_counter() - t
    print(f"n={n}: greedy_triangle_collapse {dt:.3f} s -> {len(mdf)} metacells, {len(tri)} triangles, {qhull_pool.pool().n} Qhull helpers")
    print(f"  one Qhull call on the original cells {t_q0:.3f} s, on the final metacells {t_q1:.3f} s "
          f"(round 1 made both of these twice: 7 calls, 1.14 of 1.35 s)")
    saved = qhull_pool._pool
    qhull_pool._pool = qhull_pool.QhullPool(0)
    t = time.perf_counter()
    mdf0, tri0 = greedy_triangle_collapse(df, max_metacell_size=size, r_max=40, min_angle_deg=10, verbose=False)
    dt0 = time.perf_counter() - t
    qhull_pool._pool = saved
    assert mdf0.drop(columns=["members"]).equals(mdf.drop(columns=["members"])) and np.array_equal(tri0, tri)
    print(f"  without helpers (triangulations reused, none prefetched): {dt0:.3f} s, same result")


if __name__ == "__main__":
    main()
