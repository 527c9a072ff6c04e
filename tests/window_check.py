"""Shared by the GPU tests of the device-resident window path: one window of windows.iter_device_windows against the same window of
the column pipeline (windows.iter_window_arrays) and the host-buffer entry points run on its arrays."""
import numpy as np


def check_window(W, ops, wa, dw, penalty):
    """Everything the device path computes for a window equals the column pipeline's; -> number of matched aligned cells."""
    st = dw.state
    assert np.array_equal(dw.rows_m, wa.rows_m) and np.array_equal(dw.axy, wa.axy)
    pairs, rows_r = st.fetch(W._W_PAIRS), st.fetch(W._W_ROWS_R)
    assert dw.counts == (len(st.fetch(W._W_ROWS_M)), len(rows_r), len(wa.rows_m), len(wa.pairs))
    assert np.array_equal(pairs[:, 0], wa.pairs[:, 0]) and np.array_equal(rows_r[pairs[:, 1]], wa.rows_r[wa.pairs[:, 1]])
    assert np.array_equal(st.fetch(W._W_ROWS_M)[st.fetch(W._W_KEPT)], wa.rows_m)
    assert np.array_equal(st.fetch(W._W_COSTS), wa.costs)
    assert np.array_equal(dw.triangles, wa.triangles) and dw.n_triangles == len(wa.triangles)
    assert np.array_equal(st.fetch(W._W_SIGNS), wa.signs.astype(np.int8))
    assert np.array_equal(st.fetch(W._W_WEIGHTS), np.asarray(wa.weights, dtype=np.float64))
    # the incumbent and the sweeps through the host-buffer entry points on the column pipeline's arrays
    p32 = wa.pairs.astype(np.int32)
    wants = ops.pair_rowmin(p32, wa.costs, wa.n_aligned) < penalty * wa.size.astype(float)
    pair_of_row, rounds = ops.greedy_match(p32, wa.costs, wa.n_aligned, wa.n_ref, wants)
    match = np.where(pair_of_row >= 0, p32[np.maximum(pair_of_row, 0), 1], -1).astype(np.int32)
    assert np.array_equal(dw.match_row, np.where(match >= 0, wa.rows_r[np.maximum(match, 0)], -1))
    sw = ops.BoundSweep(wa.triangles, wa.signs, wa.rxy, wa.n_aligned)
    checked, viol = sw.sweep_match(match)
    sw.close()
    _e, _t, pflag, counts = ops.xyorder_sweep(wa.axy, wa.rxy, wa.triangles, match)
    _b, _a, _m3, flipped = ops.area_flip(wa.axy, wa.rxy, wa.triangles, match)
    assert np.array_equal(dw.point_flag, pflag)
    assert dw.stats == dict(checked=int(checked), flipped=len(viol), xy_comparisons=int(counts[0]), xy_violations=int(counts[1]),
                            xy_triangles=int(counts[2]), area_flips=int(np.count_nonzero(flipped)), greedy_rounds=int(rounds),
                            matched=int(np.count_nonzero(match >= 0)))
    return int(np.count_nonzero(match >= 0))
