"""BASELINE.json configurations at their full sizes.

cfg2 and cfg3 are small enough for the CPU oracle to check every output.  The 100k x 100k dense
build (the metric's configuration), cfg4 and cfg5 are checked through size-independent
properties: row-block independence (the sharding invariant), transpose symmetry of the L1 cost
(|a-r| == |r-a| bit for bit), agreement of the dense and pair kernels, agreement of the two
independent prune paths (grid vs brute force) on every row, sampled rows against the oracle."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    from same_amd import _lib, ops, synth

    assert _lib.device_count() >= 1
    return _lib, ops, synth


def test_cfg2_10k_full_vs_oracle(env, oracle):
    """synthetic 10k x 10k, T=20, fp64 cost + k=32 KNN prune: every output against the oracle."""
    _lib, ops, synth = env
    ref = synth.make_cells(10_000, 20, seed=0)
    mov = synth.make_cells(10_000, 20, seed=1, side=ref["side"])
    idx, d2, cnt = ops.knn_prune(mov["xy"], ref["xy"], 25.0, 32)
    oidx, od2, ocnt = oracle.knn_prune(mov["xy"], ref["xy"], 25.0, 32)
    assert np.array_equal(idx, oidx) and np.array_equal(d2, od2) and np.array_equal(cnt, ocnt)
    assert 150_000 < int(cnt.sum()) < 320_000  # SURVEY measured 192 600 at this shape
    rr, cc = np.nonzero(idx >= 0)
    pairs = np.column_stack((rr, idx[rr, cc]))
    c = ops.pair_cost(mov["types"], ref["types"], mov["xy"], ref["xy"], pairs, 1.0)
    assert np.array_equal(c, oracle.pair_cost_arrays(mov["types"], ref["types"], mov["xy"], ref["xy"], pairs, 1.0))
    D = ops.dense_cost(mov["types"], ref["types"], mov["xy"], ref["xy"], 1.0)      # 10k x 10k x 8 B = 0.8 GB
    assert np.array_equal(D[pairs[:, 0], pairs[:, 1]], c)
    for b in (0, 3333, 9000):  # oracle rows (the full 1e8-pair oracle pass would take a minute; 3 x 1000 rows suffice with the line above)
        assert np.array_equal(D[b:b + 1000], oracle.dense_cost(mov["types"], ref["types"], mov["xy"], ref["xy"], 1.0, b, b + 1000))


def test_cfg3_50k_sweeps_full_vs_oracle(env, oracle):
    """ISS-Heart scale: 50k x 50k, sweeps over the Delaunay triangulation (~100k triangles, ~300k edge tests)."""
    from scipy.spatial import Delaunay
    _lib, ops, synth = env
    ref = synth.make_cells(50_000, 8, seed=0)
    mov = synth.make_jittered(ref, seed=1)
    tris = Delaunay(mov["xy"]).simplices.astype(np.int32)
    assert len(tris) > 90_000
    en, thr = oracle.cos_threshold(15)
    cls, perim, mc = ops.tri_classify(mov["xy"], tris, 25.0, en, thr, mov["cell_type"])
    ocls, operim, omc = oracle.tri_classify(mov["xy"], tris, 25.0, 15, mov["cell_type"])
    assert np.array_equal(cls, ocls) and np.array_equal(perim, operim) and np.array_equal(mc, omc)
    kept = tris[cls == 0]
    sign, w = ops.tri_sign_weight(mov["xy"], mov["size"], kept)
    osign, ow = oracle.tri_sign_weight(mov["xy"], mov["size"], kept)
    assert np.array_equal(sign, osign) and np.array_equal(w, ow)
    idx, _, cnt = ops.knn_prune(mov["xy"], ref["xy"], 25.0, 32, want_d2=False)
    rng = np.random.default_rng(0)
    pick = np.minimum((rng.random(len(cnt)) * np.maximum(cnt, 1)).astype(int), 31)  # a random candidate per row
    match = np.where(cnt > 0, idx[np.arange(len(cnt)), pick], -1).astype(np.int32)
    sweep = ops.BoundSweep(kept, sign, ref["xy"], len(mov["xy"]))
    checked, viol, flag = sweep.sweep_match(match, want_flag=True)
    och, oviol, oflag = oracle.orient_sweep(kept, sign, ref["xy"], match)
    assert checked == och and np.array_equal(viol, oviol) and np.array_equal(flag, oflag) and len(viol) > 1000
    e, tf, pf, counts = ops.xyorder_sweep(mov["xy"], ref["xy"], kept, match)
    oe, otf, opf, oc = oracle.xyorder_sweep(mov["xy"], ref["xy"], kept, match)
    assert np.array_equal(e, oe) and np.array_equal(tf, otf) and np.array_equal(pf, opf) and np.array_equal(counts, oc)
    assert int(counts[0]) > 200_000  # ~3 edge tests per fully matched triangle
    b, a, m3, fl = ops.area_flip(mov["xy"], ref["xy"], kept, match)
    ob, oa, om3, ofl = oracle.area_flip(mov["xy"], ref["xy"], kept, match)
    assert np.array_equal(b, ob) and np.array_equal(a, oa, equal_nan=True) and np.array_equal(fl, ofl) and np.array_equal(m3, om3)
    # permutation invariance: shuffling triangles permutes the flags and leaves the counters alone
    perm = rng.permutation(len(kept))
    s2 = ops.BoundSweep(kept[perm], sign[perm], ref["xy"], len(mov["xy"]))
    c2, v2, f2 = s2.sweep_match(match, want_flag=True)
    assert c2 == checked and np.array_equal(f2, flag[perm]) and len(v2) == len(viol)


def test_dense_100k_properties(env, oracle):
    """The metric's configuration: 100k x 100k fp64, T=20 (80 GB per matrix, two matrices resident)."""
    _lib, ops, synth = env
    n, T = 100_000, 20
    ctx = _lib.default_context()
    L, H = ctx.lib, ctx.handle
    ref = synth.make_cells(n, T, seed=0)
    mov = synth.make_cells(n, T, seed=1, side=ref["side"])
    dA, dR = ctx.to_device(mov["types"]), ctx.to_device(ref["types"])
    dax, drx = ctx.to_device(mov["xy"]), ctx.to_device(ref["xy"])
    C = ctx.alloc_spread(n * n * 8)   # cost(mov rows, ref cols), in a block laid over the HBM regions (csrc/spread.hip) ...
    assert C.spread_info["spread"] is True and C.spread_info["chunks_gib"] == 75
    Ct = ctx.alloc(n * n * 8)         # ... and cost(ref rows, mov cols) in a plain one: must be the exact transpose
    ctx.check(L.same_dense_cost_f64_dev(H, dA.ptr, dR.ptr, T, dax.ptr, drx.ptr, n, 0, n, 1.0, C.ptr, n), "dense")
    ctx.check(L.same_dense_cost_f64_dev(H, dR.ptr, dA.ptr, T, drx.ptr, dax.ptr, n, 0, n, 1.0, Ct.ptr, n), "dense")
    ctx.sync()
    rng = np.random.default_rng(0)
    rows = np.sort(rng.choice(n, 24, replace=False))
    got = {int(i): C.download((n,), np.float64, offset_bytes=int(i) * n * 8) for i in rows}
    # (1) sampled rows == oracle rows
    for i in rows[:6]:
        assert np.array_equal(got[int(i)], oracle.dense_cost(mov["types"], ref["types"], mov["xy"], ref["xy"], 1.0, int(i), int(i) + 1)[0])
    # (2) transpose symmetry, bit for bit: C[i, :] == Ct[:, i] -- gather the column from 24 x 4096 row pieces
    cols = np.sort(rng.choice(n, 4096, replace=False))
    for i in rows[:8]:
        col_piece = np.array([Ct.download((1,), np.float64, offset_bytes=(int(j) * n + int(i)) * 8)[0] for j in cols[:256]])
        assert np.array_equal(col_piece, got[int(i)][cols[:256]])
    # (3) dense == pair kernel on random pairs across the whole matrix
    pj = rng.integers(0, n, size=(len(rows), 5000))
    pairs = np.column_stack((np.repeat(rows, 5000), pj.reshape(-1))).astype(np.int32)
    c = ops.pair_cost(mov["types"], ref["types"], mov["xy"], ref["xy"], pairs, 1.0)
    assert np.array_equal(c, np.concatenate([got[int(i)][pj[q]] for q, i in enumerate(rows)]))
    # (4) row-block independence: any block computed alone equals the same rows of the full build
    blk = ctx.alloc(1000 * n * 8)
    for b in (0, 37_123, n - 1000):
        ctx.check(L.same_dense_cost_f64_dev(H, dA.ptr, dR.ptr, T, dax.ptr, drx.ptr, n, b, b + 1000, 1.0, blk.ptr, n), "dense")
        ctx.sync()
        for q in (0, 499, 999):
            full_row = C.download((n,), np.float64, offset_bytes=(b + q) * n * 8)
            assert np.array_equal(blk.download((n,), np.float64, offset_bytes=q * n * 8), full_row)
    # (5) a ragged sub-problem (odd column count -> scalar-store variant) agrees with the vector variant
    odd = ctx.alloc(64 * 99_999 * 8)
    ctx.check(L.same_dense_cost_f64_dev(H, dA.ptr, dR.ptr, T, dax.ptr, drx.ptr, 99_999, 500, 564, 1.0, odd.ptr, 99_999), "dense")
    ctx.sync()
    assert np.array_equal(odd.download((99_999,), np.float64, offset_bytes=63 * 99_999 * 8),
                          C.download((n,), np.float64, offset_bytes=563 * n * 8)[:99_999])
    for buf in (C, Ct, blk, odd):
        buf.free()


def test_knn_100k_grid_vs_brute_and_oracle(env, oracle, monkeypatch):
    _lib, ops, synth = env
    n = 100_000
    ref = synth.make_cells(n, 2, seed=0)
    mov = synth.make_cells(n, 2, seed=1, side=ref["side"])
    monkeypatch.setenv("SAME_KNN_MODE", "grid")
    gi, gd, gc = ops.knn_prune(mov["xy"], ref["xy"], 25.0, 32)
    monkeypatch.setenv("SAME_KNN_MODE", "brute")
    bi, bd, bc = ops.knn_prune(mov["xy"], ref["xy"], 25.0, 32)
    assert np.array_equal(gi, bi) and np.array_equal(gd, bd) and np.array_equal(gc, bc)   # all 100k rows, two independent paths
    oi, od, oc = oracle.knn_prune(mov["xy"], ref["xy"], 25.0, 32, 40_000, 41_000)
    assert np.array_equal(gi[40_000:41_000], oi) and np.array_equal(gd[40_000:41_000], od)
    with np.errstate(invalid="ignore"):
        dd = np.diff(gd, axis=1)
    assert (dd[np.isfinite(gd[:, 1:])] >= 0).all()  # rows sorted by distance
    # idempotence: pruning against only the refs that were ever selected gives the same lists
    used = np.unique(gi[gi >= 0])
    monkeypatch.delenv("SAME_KNN_MODE")
    ri, rd, rc = ops.knn_prune(mov["xy"], ref["xy"][used], 25.0, 32)
    assert np.array_equal(np.where(ri >= 0, used[np.maximum(ri, 0)], -1), gi) and np.array_equal(rd, gd)


def test_cfg4_200k_row_block_sharding_invariant(env, oracle):
    """200k x 200k in 8 row blocks of 25k (what 8 ranks compute) == the one-shot result == the oracle on sampled rows."""
    _lib, ops, synth = env
    n, k, T = 200_000, 32, 20
    ref = synth.make_cells(n, T, seed=0)
    mov = synth.make_cells(n, T, seed=1, side=ref["side"])
    ctx = _lib.default_context()
    L, H = ctx.lib, ctx.handle
    dA, dR = ctx.to_device(mov["types"]), ctx.to_device(ref["types"])
    dax, drx = ctx.to_device(mov["xy"]), ctx.to_device(ref["xy"])
    didx, dcost, dcnt = ctx.alloc(n * k * 4), ctx.alloc(n * k * 8), ctx.alloc(n * 4)

    def run(b, e, off):
        ctx.check(L.same_knn_prune_dev(H, dax.ptr, drx.ptr, n, b, e, 25.0, k, didx.ptr + off * k * 4, None, dcnt.ptr + off * 4), "knn")
        ctx.check(L.same_padded_cost_f64_dev(H, dA.ptr, dR.ptr, T, dax.ptr, drx.ptr, b, e, k, didx.ptr + off * k * 4, 1.0,
                                             dcost.ptr + off * k * 8), "cost")

    run(0, n, 0)
    whole_idx, whole_cost = didx.download((n, k), np.int32), dcost.download((n, k), np.float64)
    ctx.check(L.same_dev_memset(H, didx.ptr, 0, n * k * 4), "memset")
    for r in range(8):
        run(r * 25_000, (r + 1) * 25_000, r * 25_000)
    assert np.array_equal(didx.download((n, k), np.int32), whole_idx)
    assert np.array_equal(dcost.download((n, k), np.float64), whole_cost)
    assert np.isinf(whole_cost[whole_idx < 0]).all() and np.isfinite(whole_cost[whole_idx >= 0]).all()
    # not only self-consistent: three 1000-row samples (first block, a block boundary, the last rows) against the oracle
    for b in (0, 24_500, 199_000):
        oi, _, _ = oracle.knn_prune(mov["xy"], ref["xy"], 25.0, k, b, b + 1000)
        assert np.array_equal(whole_idx[b:b + 1000], oi)
        rr, cc = np.nonzero(oi >= 0)
        want = oracle.pair_cost_arrays(mov["types"], ref["types"], mov["xy"], ref["xy"], np.column_stack((rr + b, oi[rr, cc])), 1.0)
        assert np.array_equal(whole_cost[b:b + 1000][rr, cc], want)
    # and the caller-held index (what the ranks of bench.py use) gives the same lists
    ix = ctypes.c_void_p()
    ctx.check(L.same_knn_index_build(H, drx.ptr, n, 25.0, ctypes.byref(ix)), "index")
    ctx.check(L.same_dev_memset(H, didx.ptr, 0, n * k * 4), "memset")
    for r in range(8):
        ctx.check(L.same_knn_prune_indexed_dev(H, ix, dax.ptr, r * 25_000, (r + 1) * 25_000, k, didx.ptr + r * 25_000 * k * 4, None,
                                               dcnt.ptr + r * 25_000 * 4), "indexed")
    assert np.array_equal(didx.download((n, k), np.int32), whole_idx)
    L.same_knn_index_destroy(ix)


def test_cfg5_1m_cells_windows_fp32(env, oracle):
    """1M-cell section tiled into windows (BASELINE config 5): the plan's counts against plain masks, the round-robin deal,
    and THREE windows (a corner, the middle, the far edge) through the whole fp32 window pipeline at the Python boundary --
    prune + compaction, Delaunay + filter, weights / signs, fp32 pair costs, greedy start on those costs, and all three
    sweeps under that start -- every output against the oracle."""
    import pandas as pd
    from scipy.spatial import Delaunay

    import same_amd
    from same_amd.windows import assign_windows, window_plan

    _lib, ops, synth = env
    T = 8
    ref = synth.make_cells(1_000_000, T, seed=0)            # side 10 000
    mov = synth.make_jittered(ref, seed=1)
    plan = window_plan(ref["xy"], mov["xy"], 1200, 300, 10)
    assert len(plan) > 100
    for w in plan[::17]:
        x0, x1, y0, y1 = w["box"]
        m = (mov["xy"][:, 0] >= x0) & (mov["xy"][:, 0] < x1) & (mov["xy"][:, 1] >= y0) & (mov["xy"][:, 1] < y1)
        r = (ref["xy"][:, 0] >= x0) & (ref["xy"][:, 0] < x1) & (ref["xy"][:, 1] >= y0) & (ref["xy"][:, 1] < y1)
        assert w["n_mov"] == int(m.sum()) and w["n_ref"] == int(r.sum())
    shards = assign_windows(plan, 8)
    assert sorted(q for s in shards for q in s) == list(range(len(plan)))
    r_df, m_df = synth.to_frame(ref), synth.to_frame(mov)
    cols = synth.type_columns(T)
    op = dict(radius=25, knn=8, no_match_penalty=100, hip_cost_dtype="float32")
    big_last = [w for w in plan if w["n_mov"] > 5_000][-1]
    assert plan[0]["n_mov"] > 5_000 and plan[len(plan) // 2]["n_mov"] > 5_000 and plan[-1]["n_mov"] < 1_000   # + the thin edge strip
    expect = []
    for w in (plan[0], plan[len(plan) // 2], big_last, plan[-1]):
        x0, x1, y0, y1 = w["box"]
        rs, ms = same_amd.subset_data(r_df, x0, x1, y0, y1), same_amd.subset_data(m_df, x0, x1, y0, y1)
        assert len(ms) == w["n_mov"] >= 10
        prep = same_amd.prepare_same_inputs(rs, ms, cols, optim_params=op, verbose=False)
        # a2: pairs + compaction
        na, nr, pairs = oracle.find_knn_within_radius(ms, rs, 25, 8)
        pairs = np.asarray(pairs, dtype=np.int64)
        assert np.array_equal(np.asarray(prep.valid_pairs, dtype=np.int64), pairs)
        assert np.array_equal(prep.aligned_df["Cell_Num_Old"].to_numpy(), na["Cell_Num_Old"].to_numpy())
        assert np.array_equal(prep.ref_df["Cell_Num_Old"].to_numpy(), nr["Cell_Num_Old"].to_numpy())
        # a4 in float: bit-equal to the oracle's float evaluation, 1e-5 relative to the fp64 costs
        A, R = na[cols].to_numpy(), nr[cols].to_numpy()
        axy, rxy = na[["X", "Y"]].to_numpy(), nr[["X", "Y"]].to_numpy()
        c32 = oracle.pair_cost_arrays(A, R, axy, rxy, pairs, 1.0, dtype=np.float32)
        got_c = np.array(prep.costs)
        assert got_c.dtype == np.float64 and np.array_equal(got_c.astype(np.float32), c32) and np.array_equal(got_c, c32.astype(np.float64))
        # against the fp64 costs: a forward bound, not a relative one -- |ax - rx| is formed in float from coordinates of
        # magnitude ~1e4, so a pair of near-identical cells (cost ~1e-3 here: jittered copies) keeps only the absolute
        # accuracy of its operands: (T + 4) roundings of 2^-24 on the operand magnitudes
        c64 = oracle.pair_cost_arrays(A, R, axy, rxy, pairs, 1.0)
        mag = (np.abs(A[pairs[:, 0]]) + np.abs(R[pairs[:, 1]])).sum(axis=1)
        mag = mag + 0.001 * (np.abs(axy[pairs[:, 0]]) + np.abs(rxy[pairs[:, 1]])).sum(axis=1)
        assert (np.abs(got_c - c64) <= (T + 4) * 2.0 ** -24 * mag).all()
        # a6-a8: triangulation (Qhull on the host, as in the reference), filter, weights, signs
        tri = oracle.filter_triangles_by_radius(axy, Delaunay(axy).simplices, 25, aligned_df=na, ignore_same_type_triangles=True,
                                                min_angle_deg=15)
        tri = np.asarray(tri, dtype=np.int64).reshape(-1, 3)
        assert np.array_equal(np.asarray(prep.aligned_delaunay, dtype=np.int64).reshape(-1, 3), tri)
        assert prep.triangle_weights == oracle.triangle_weights(na, tri) and prep.source_signs == oracle.source_signs(na, tri)
        # a5 / f1 on the fp32 costs
        kw = dict(valid_pairs=prep.valid_pairs, costs=prep.costs, n_aligned=prep.n_aligned, n_ref=prep.n_ref,
                  aligned_sizes=na["size"].to_numpy(dtype=float), no_match_penalty=100, max_matches=1, init_method="greedy", verbose=False)
        ch, un = same_amd.compute_mip_start_pairs(**kw)
        och, oun = oracle.compute_mip_start_pairs(**kw)
        assert ch == och and un == oun and len(ch) > 0.5 * prep.n_aligned
        # a10 under that start
        x = np.zeros(len(pairs))
        x[[c[2] for c in ch]] = 1.0
        sw = same_amd.LazyOrientationSweep(prep.valid_pairs, tri, prep.source_signs, rxy, prep.n_aligned)
        checked, viol, _ = sw.sweep(x)
        ochecked, oviol = oracle.lazy_orientation_sweep(x, pairs, tri, prep.source_signs, rxy, prep.n_aligned)
        assert checked == ochecked and [tuple(int(q) for q in v) for v in viol] == [tuple(int(q) for q in v) for v in oviol]
        # a11 / a12
        m_pairs = pd.DataFrame({"aligned_idx": [c[0] for c in ch], "ref_idx": [c[1] for c in ch]})
        info = prep.triangle_info
        rep, orep = same_amd.verify_spatial_preservation(na, nr, m_pairs, info), oracle.verify_spatial_preservation(na, nr, m_pairs, info)
        assert rep["violation_summary"] == orep["violation_summary"]
        assert rep["x_order_violations"] == orep["x_order_violations"] and rep["y_order_violations"] == orep["y_order_violations"]
        assert sorted(rep["points_with_violations"]) == sorted(orep["points_with_violations"])
        match = np.full(prep.n_aligned, -1, np.int32)
        match[m_pairs["aligned_idx"].to_numpy()] = m_pairs["ref_idx"].to_numpy()
        before, after, flipped, m3 = same_amd.triangle_area_flips(na, nr, tri, {int(i): int(j) for i, j, _ in ch})
        ob, oa, om3, ofl = oracle.area_flip(axy, rxy, tri, match)
        assert np.array_equal(np.array([before[t] for t in range(len(tri))]), ob)
        assert np.array_equal(np.array([np.nan if after[t] is None else after[t] for t in range(len(tri))]), oa, equal_nan=True)
        assert flipped == np.flatnonzero(ofl).tolist()
        expect.append(dict(window=w, rows_m=na["Cell_Num_Old"].to_numpy(), rows_r=nr["Cell_Num_Old"].to_numpy(), pairs=pairs, c32=c32,
                           tri=tri, signs=np.asarray(prep.source_signs, dtype=np.int8), chosen=och, checked=ochecked, flipped=len(oviol),
                           area_flips=int(np.count_nonzero(ofl)), xy=orep["violation_summary"]))
    # the same four windows with both 1M-cell sections resident on the device (csrc/window.hip: what bench.py --workload cfg5 times)
    from same_amd import windows as W

    assert np.array_equal(r_df["Cell_Num_Old"].to_numpy(), np.arange(len(r_df)))
    assert np.array_equal(m_df["Cell_Num_Old"].to_numpy(), np.arange(len(m_df)))
    ref_sec, mov_sec = W.Section.from_frame(r_df, cols), W.Section.from_frame(m_df, cols)
    dref, dmov = W.DeviceSection(ref_sec, "float32"), W.DeviceSection(mov_sec, "float32")
    got = W.iter_device_windows(ref_sec, mov_sec, dref, dmov, [e["window"] for e in expect], radius=25, knn=8, dist_ct_coeff=1.0,
                                min_angle_deg=15, ignore_same_type_triangles=True, no_match_penalty=100.0, fetch_triangles=True)
    for e, dw in zip(expect, got):
        assert dw.error is None and np.array_equal(dw.rows_m, e["rows_m"])
        dp, rows_r = dw.state.fetch(W._W_PAIRS), dw.state.fetch(W._W_ROWS_R)
        assert np.array_equal(dp[:, 0], e["pairs"][:, 0]) and np.array_equal(rows_r[dp[:, 1]], e["rows_r"][e["pairs"][:, 1]])
        assert np.array_equal(dw.state.fetch(W._W_COSTS).astype(np.float32), e["c32"])
        assert np.array_equal(dw.triangles, e["tri"]) and np.array_equal(dw.state.fetch(W._W_SIGNS), e["signs"])
        match_o = np.full(len(e["rows_m"]), -1, np.int64)
        for i, j, _p in e["chosen"]:
            match_o[i] = e["rows_r"][j]
        assert np.array_equal(dw.match_row, match_o)
        st = dw.stats
        want_counts = (e["checked"], e["flipped"], e["area_flips"], len(e["chosen"]))
        assert (st["checked"], st["flipped"], st["area_flips"], st["matched"]) == want_counts
        assert st["xy_comparisons"] == e["xy"]["total_comparisons"] and st["xy_violations"] == e["xy"]["total_violations"]
    dref.close()
    dmov.close()
    # an unknown cost dtype is refused at the boundary
    with pytest.raises(ValueError):
        same_amd.prepare_same_inputs(rs, ms, cols, optim_params=dict(op, hip_cost_dtype="float16"), verbose=False)


def test_window_stage_with_thousands_of_scan_blocks(env):
    """The single-launch scans of the window path (csrc/scan.h) well past one look-back window of 64 blocks: ONE box over a whole
    1.2M-row section, (a) on a grid of 7 x 7 cells -- the cell-run path with 1.2M candidates: 4 700 blocks in the scatter's scan --, (b) on
    25-unit cells -- the box covers far more than 64 cells: the mask over the section, 74 blocks in its compaction -- and (c) a box
    that cuts through the cells (flagged candidates compacted by a 2 300-block scan).  Rows = np.flatnonzero of the box test; kept rows,
    pairs and their order = the host-buffer prune's padded lists (src/utils.py:709-742); nothing depends on the grid."""
    import same_amd  # noqa: F401
    from same_amd import ops, synth
    from same_amd import windows as W
    from same_amd.knn import pairs_from_padded

    n, T, k, r = 1_200_000, 2, 4, 10.0
    ref = synth.make_cells(n, T, seed=70)
    mov = synth.make_cells(n, T, seed=71)
    side = float(max(ref["xy"].max(), mov["xy"].max())) + 1.0
    ref_sec, mov_sec = W.Section(ref["xy"], ref["types"], None, None), W.Section(mov["xy"], mov["types"], None, None)
    idx, _, cnt = ops.knn_prune(mov["xy"], ref["xy"], r, k, want_d2=False)
    want_pairs = pairs_from_padded(idx)
    kept = np.flatnonzero(cnt > 0)
    assert len(want_pairs) > 2_000_000 and 0.5 * n < len(kept) < n
    half = (0.0, side / 2 + 3.3, -5.0, side + 5.0)
    in_m = np.flatnonzero((mov["xy"][:, 0] >= half[0]) & (mov["xy"][:, 0] < half[1]))
    in_r = np.flatnonzero((ref["xy"][:, 0] >= half[0]) & (ref["xy"][:, 0] < half[1]))
    st = W.DeviceWindow()
    for name, cell in (("7 x 7 cells", side / 7.0 + 1e-6), ("25-unit cells", 25.0)):
        dref, dmov = W.DeviceSection(ref_sec, "float64"), W.DeviceSection(mov_sec, "float64")
        dref.bin(0.0, 0.0, cell)
        dmov.bin(0.0, 0.0, cell)
        n_m, n_r, n_kept, n_pairs = st.stage(dmov, dref, (-1.0, side + 1.0, -1.0, side + 1.0), r, k, 1.0)
        assert (n_m, n_r, n_kept, n_pairs) == (n, n, len(kept), len(want_pairs)), name
        assert np.array_equal(st.fetch(W._W_ROWS_M), np.arange(n)) and np.array_equal(st.fetch(W._W_ALIGNED_ROWS), kept), name
        got = st.fetch(W._W_PAIRS)
        # all refs in the box: window number = row
        assert np.array_equal(kept[got[:, 0]], want_pairs[:, 0]) and np.array_equal(got[:, 1], want_pairs[:, 1]), name
        # half the section: the box cuts through a column of cells (7 x 7) / covers too many cells (25-unit)
        n_m, n_r, _nk, _np = st.stage(dmov, dref, half, r, k, 1.0)
        assert (n_m, n_r) == (len(in_m), len(in_r)), name
        assert np.array_equal(st.fetch(W._W_ROWS_M), in_m) and np.array_equal(st.fetch(W._W_ROWS_R), in_r), name
        gp, rows_r, rows_ua = st.fetch(W._W_PAIRS), st.fetch(W._W_ROWS_R), st.fetch(W._W_ALIGNED_ROWS)
        sel = np.isin(want_pairs[:, 0], in_m)          # pairs of aligned rows in the box whose reference cell is in the box as well ...
        # ... as long as no in-box row loses a nearer out-of-box candidate to the k limit: compare rows whose whole list is inside
        full = np.ones(n, bool)
        outside = np.ones(n, bool)
        outside[in_r] = False
        bad_rows = np.unique(want_pairs[outside[want_pairs[:, 1]], 0])
        full[bad_rows] = False
        keep_rows = full[rows_ua[gp[:, 0]]]
        assert np.array_equal(np.column_stack((rows_ua[gp[:, 0]], rows_r[gp[:, 1]]))[keep_rows],
                              want_pairs[sel & full[want_pairs[:, 0]]]), name
        dref.close()
        dmov.close()
    st.close()


def test_merge_dedup_properties_at_cfg5_scale(env):
    """The window-merge de-duplication at the size a 1M-cell section produces and beyond (2M rows, past the oracle-in-seconds
    range used elsewhere), through properties that do not need the oracle: the survivors come out in stable (violation, window,
    row) order; every (aligned, ref) pair survives exactly once; the survivor of a pair is its minimum in that order; running
    the survivors through again changes nothing."""
    _lib, ops, synth = env
    rng = np.random.default_rng(11)
    n = 2_000_000
    a = rng.integers(0, 900_000, n).astype(np.int32)
    r = (a + rng.integers(0, 2, n)).astype(np.int32)               # mostly the same ref per aligned cell: long duplicate runs
    viol = rng.random(n) < 0.3
    win = rng.integers(0, 144, n).astype(np.int32)
    kept = ops.merge_dedup(viol, win, a, r)
    order_key = (viol[kept].astype(np.int64) << 52) | (win[kept].astype(np.int64) << 32) | kept.astype(np.int64)
    assert (np.diff(order_key) > 0).all()                                         # stable (violation, window, row) order
    pair = a.astype(np.int64) << 32 | r.astype(np.int64)
    assert len(np.unique(pair[kept])) == len(kept) == len(np.unique(pair))       # every pair exactly once
    full_key = (viol.astype(np.int64) << 52) | (win.astype(np.int64) << 32) | np.arange(n, dtype=np.int64)
    o = np.lexsort((full_key, pair))
    first = o[np.r_[True, pair[o][1:] != pair[o][:-1]]]                           # minimum key of every pair
    assert np.array_equal(np.sort(first), np.sort(kept.astype(np.int64)))
    again = ops.merge_dedup(viol[kept], win[kept], a[kept], r[kept])
    assert np.array_equal(again, np.arange(len(kept)))                            # idempotent


def test_cfg5_1m_cells_own_triangulator_gives_the_same_merged_table(env):
    """BASELINE config 5 at full size through the product function, window merge included, with the windows triangulated by
    libsame_hip's own triangulator (optim_params["hip_delaunay"] = "native") and by scipy: the 950k-row merged tables and every window's
    counters are identical, and nearly every window (13 600 aligned cells each) is answered by the library itself."""
    import same_amd
    from same_amd import delaunay

    _lib, ops, synth = env
    T = 8
    ref = synth.make_cells(1_000_000, T, seed=0)
    mov = synth.make_jittered(ref, seed=1)
    r_df, m_df = synth.to_frame(ref), synth.to_frame(mov)
    r_df["Cell_Num_Old"], m_df["Cell_Num_Old"] = np.arange(len(r_df)), np.arange(len(m_df))
    cols = synth.type_columns(T)
    op = dict(radius=25, knn=8, no_match_penalty=100, hip_cost_dtype="float32", window_size=1200, overlap=300, min_cells_per_window=10)
    tr = delaunay.shared()
    with same_amd.resident_frames(r_df, m_df) as res:
        want, want_stats = same_amd.sliding_window_incumbent(res, res, commonCT=cols, optim_params=dict(op), merge=True, return_stats=True)
        before = (tr.submitted, tr.asked_qhull)
        got, got_stats = same_amd.sliding_window_incumbent(res, res, commonCT=cols, optim_params=dict(op, hip_delaunay="native"),
                                                           merge=True, return_stats=True)
    windows, sent_back = tr.submitted - before[0], tr.asked_qhull - before[1]
    assert len(want) > 900_000 and got.equals(want) and got_stats == want_stats
    assert windows == len(want_stats) > 100 and sent_back <= windows // 10, (windows, sent_back)
