"""BASELINE cfg 5 shape: a 1M-cell section tiled into overlapping windows; per window the whole pre-MIP path
(prune + compaction, Delaunay, triangle filter, weights/signs, pair costs, greedy start, orientation sweep under that
start) through the Python boundary.  Two passes over the same windows: strictly serial (every window pays its own Qhull
call in-process: SAME_QHULL_WORKERS=0 behaviour) and pipelined (same_amd.iter_prepared_windows: windows n+1..n+k pruned
ahead, triangulated by helper processes while window n runs).  Reports ms/window for both, the Qhull time alone and the
remaining main-process time, i.e. max(host, device) that the pipelined figure is to be compared with.
Usage: python tools/window_bench.py [n_cells] [max_windows] [f32]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import same_amd
    from scipy.spatial import Delaunay
    from same_amd import qhull_pool, synth
    from same_amd.windows import window_plan

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    max_windows = int(sys.argv[2]) if len(sys.argv) > 2 else 10**9      # default: the whole plan
    f32 = len(sys.argv) > 3 and sys.argv[3] == "f32"
    T = 8
    ref = synth.make_cells(n, T, seed=0); mov = synth.make_jittered(ref, seed=1)
    r_df, m_df = synth.to_frame(ref), synth.to_frame(mov)
    cols = synth.type_columns(T)
    t = time.perf_counter()
    plan = window_plan(ref["xy"], mov["xy"], 1200, 300, 10)
    print(f"{n} cells: {len(plan)} windows planned in {(time.perf_counter() - t) * 1e3:.1f} ms (one batched device count pass per section)")
    op = dict(radius=25, knn=8, no_match_penalty=100)
    if f32:
        op["hip_cost_dtype"] = "float32"
    picked = plan[:: max(1, len(plan) // max_windows)][:max_windows]

    def consume(prep):
        ch, un = same_amd.compute_mip_start_pairs(valid_pairs=prep.valid_pairs, costs=prep.costs_array, n_aligned=prep.n_aligned, n_ref=prep.n_ref,
                                                  aligned_sizes=prep.aligned_df["size"].to_numpy(dtype=float), no_match_penalty=100,
                                                  max_matches=1, init_method="greedy", verbose=False)
        x = np.zeros(len(prep.valid_pairs)); x[[c[2] for c in ch]] = 1.0
        sw = same_amd.LazyOrientationSweep(prep.valid_pairs, prep.triangles_array, prep.signs_array, prep.ref_df[["X", "Y"]].to_numpy(), prep.n_aligned)
        checked, viol, _ = sw.sweep(x)
        sw.bound.close()
        return prep.n_aligned, len(prep.valid_pairs), len(prep.triangles_array), checked, len(viol)

    def serial():
        out = []
        for w in picked:
            x0, x1, y0, y1 = w["box"]
            rs, ms = same_amd.subset_data(r_df, x0, x1, y0, y1), same_amd.subset_data(m_df, x0, x1, y0, y1)
            out.append(consume(same_amd.prepare_same_inputs(rs, ms, cols, optim_params=op, verbose=False)))
        return out

    def pipelined():
        return [consume(prep) for _w, prep in same_amd.iter_prepared_windows(r_df, m_df, cols, picked, optim_params=op)]

    serial()                                   # warm-up: scratch slots, helper start-up
    pipelined()
    t = time.perf_counter(); a = serial(); t_serial = time.perf_counter() - t
    t = time.perf_counter(); b = pipelined(); t_pipe = time.perf_counter() - t
    assert a == b, "pipelined windows differ from serial windows"
    # Qhull alone on the same point sets (what the helpers hide)
    t_q = 0.0
    for w in picked:
        x0, x1, y0, y1 = w["box"]
        ms = same_amd.subset_data(m_df, x0, x1, y0, y1)
        t = time.perf_counter(); Delaunay(ms[["X", "Y"]].values); t_q += time.perf_counter() - t
    k = len(picked)
    cells, pairs, tris = (sum(r[q] for r in a) for q in range(3))
    rest = (t_serial - t_q) / k
    print(f"{k} windows ({'fp32' if f32 else 'fp64'} costs; avg {cells // k} cells, {pairs // k} pairs, {tris // k} triangles per window), "
          f"{qhull_pool.pool().n} Qhull helpers")
    print(f"serial     {t_serial / k * 1e3:7.1f} ms/window  ({cells / t_serial:.3e} aligned cells/s)")
    print(f"  of which Qhull {t_q / k * 1e3:6.1f} ms/window, everything else (main process: device calls + host glue) {rest * 1e3:6.1f} ms/window")
    print(f"pipelined  {t_pipe / k * 1e3:7.1f} ms/window  ({cells / t_pipe:.3e} aligned cells/s) = "
          f"{t_pipe / k / max(rest, t_q / k / max(qhull_pool.pool().n, 1)):.2f} x max(main-process time, Qhull time / helpers)")


if __name__ == "__main__":
    main()
