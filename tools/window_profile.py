"""cProfile of the main process over the pipelined window loop of tools/window_bench.py (where does the host time go once
Qhull is hidden?).  Usage: python tools/window_profile.py [n_cells] [windows]"""
import cProfile, os, pstats, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import same_amd
    from same_amd import synth
    from same_amd.windows import window_plan

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    T = 8
    ref = synth.make_cells(n, T, seed=0); mov = synth.make_jittered(ref, seed=1)
    r_df, m_df = synth.to_frame(ref), synth.to_frame(mov)
    cols = synth.type_columns(T)
    plan = window_plan(ref["xy"], mov["xy"], 1200, 300, 10)
    picked = plan[:: max(1, len(plan) // k)][:k]
    op = dict(radius=25, knn=8, no_match_penalty=100)

    def run():
        for _w, prep in same_amd.iter_prepared_windows(r_df, m_df, cols, picked, optim_params=op):
            ch, un = same_amd.compute_mip_start_pairs(valid_pairs=prep.valid_pairs, costs=prep.costs_array, n_aligned=prep.n_aligned, n_ref=prep.n_ref,
                                                      aligned_sizes=prep.aligned_df["size"].to_numpy(dtype=float), no_match_penalty=100,
                                                      max_matches=1, init_method="greedy", verbose=False)
            x = np.zeros(len(prep.valid_pairs)); x[[c[2] for c in ch]] = 1.0
            sw = same_amd.LazyOrientationSweep(prep.valid_pairs, prep.triangles_array, prep.signs_array, prep.ref_df[["X", "Y"]].to_numpy(), prep.n_aligned)
            sw.sweep(x)
            sw.bound.close()

    run()
    pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)


if __name__ == "__main__":
    main()
