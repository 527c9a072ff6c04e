"""Seeded randomised sweep of small shapes: every kernel against the oracle on ragged sizes, duplicate points,
exact ties, radii from tiny to whole-set, k up to 448, T up to 90 (both dense kernels), fp32 and fixed-point cost variants,
indexed prunes, triangle-block sweeps for random rank counts, sparse and full matchings.  Cheap per case;
the point is to visit shape/edge combinations the fixed fixtures do not."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROUNDS = int(os.environ.get("SAME_FUZZ_ROUNDS", "1"))   # soak runs: more rounds = more seeds (round 0 is the default suite)
FIRST = int(os.environ.get("SAME_FUZZ_FIRST_ROUND", "0"))   # ... and a later soak continues where an earlier one stopped (other seeds)


@pytest.fixture(scope="module")
def ops():
    from same_amd import _lib, ops as _ops

    assert _lib.device_count() >= 1
    return _ops


def _points(rng, n, side, mode):
    if mode == 0:
        return rng.uniform(0, side, (n, 2))
    if mode == 1:  # lattice: exact distance ties
        return rng.integers(0, max(2, int(side // 3)), (n, 2)).astype(float) * 3.0
    if mode == 2:  # clustered + duplicates
        c = rng.uniform(0, side, (max(1, n // 20), 2))
        return c[rng.integers(0, len(c), n)] + rng.choice([0.0, 0.0, 0.5, 1.0], (n, 2))
    return np.column_stack((rng.uniform(0, side, n), np.full(n, side / 2)))  # collinear


def test_fuzz_knn_and_costs(ops, oracle):
    for rnd in range(FIRST, FIRST + ROUNDS):
        if ROUNDS > 1 and rnd % 20 == 0:
            print(f"knn/cost soak round {rnd}", flush=True)
        rng = np.random.default_rng(2024 + 7919 * rnd)
        for case in range(120):
            n_m, n_r = int(rng.integers(1, 700)), int(rng.integers(1, 900))
            if case % 10 == 0:
                n_r = int(rng.integers(2100, 5000))  # above the grid threshold
            T, k = int(rng.integers(0, 41)), int(rng.choice([1, 2, 5, 8, 31, 32, 33, 64, 65, 100, 200, 448]))
            if case % 7 == 3:
                T = int(rng.integers(41, 91))          # one column per lane, then the row-blocked kernel from 48
            side = float(rng.choice([10.0, 100.0, 1000.0]))
            radius = float(rng.choice([0.0, 0.5, 3.0, side / 10, side / 3, side * 2]))
            axy, rxy = _points(rng, n_m, side, int(rng.integers(0, 4))), _points(rng, n_r, side, int(rng.integers(0, 4)))
            idx, d2, cnt = ops.knn_prune(axy, rxy, radius, k)
            oidx, od2, ocnt = oracle.knn_prune(axy, rxy, radius, k)
            assert np.array_equal(idx, oidx) and np.array_equal(d2, od2) and np.array_equal(cnt, ocnt), (case, n_m, n_r, k, radius)
            A = rng.gamma(0.3, 30.0, (n_m, T))
            R = rng.gamma(0.3, 30.0, (n_r, T))
            w = float(rng.choice([1.0, 0.25, 3.7]))
            rr, cc = np.nonzero(idx >= 0)
            pairs = np.column_stack((rr, idx[rr, cc])).astype(np.int32)
            if len(pairs):
                c = ops.pair_cost(A, R, axy, rxy, pairs, w)
                assert np.array_equal(c, oracle.pair_cost_arrays(A, R, axy, rxy, pairs, w)), (case, T)
            b, e = sorted(rng.integers(0, n_m + 1, 2))
            D = ops.dense_cost(A, R, axy, rxy, w, int(b), int(e))
            assert np.array_equal(D, oracle.dense_cost(A, R, axy, rxy, w, int(b), int(e))), (case, T, n_r, b, e)
            if case % 4 == 0:
                D32 = ops.dense_cost(A, R, axy, rxy, w, int(b), int(e), dtype=np.float32)
                assert np.array_equal(D32, oracle.dense_cost(A, R, axy, rxy, w, int(b), int(e), dtype=np.float32)), (case, T)
                if len(pairs):   # fp32 pair costs (config 5) == the oracle's float twin == elements of the fp32 dense build
                    c32 = ops.pair_cost(A, R, axy, rxy, pairs, w, dtype=np.float32)
                    assert np.array_equal(c32, oracle.pair_cost_arrays(A, R, axy, rxy, pairs, w, dtype=np.float32)), (case, T)
                    inb = (pairs[:, 0] >= b) & (pairs[:, 0] < e)
                    assert np.array_equal(c32[inb], D32[pairs[inb, 0] - int(b), pairs[inb, 1]]), (case, T)
            if case % 3 == 1 and T <= 32 and e > b:   # opt-in fixed-point build: twin-equal, and within 1e-6 relative of the exact build
                grid = ops.quantize_types(A, R)
                Q, bound = ops.dense_cost_q32(A, R, axy, rxy, w, int(b), int(e), grid=grid)
                assert np.array_equal(Q, oracle.dense_cost_q32(A, R, axy, rxy, w, grid[0], grid[1], int(b), int(e))), (case, T)
                nz = D != 0
                assert (np.abs(Q - D)[nz] <= 1e-6 * (1 + 1e-9) * np.abs(D)[nz]).all() and (np.abs(Q - D)[~nz] <= w * bound + 1e-300).all(), (case, T)


def test_fuzz_triangles_and_sweeps(ops, oracle):
    from scipy.spatial import Delaunay
    for rnd in range(FIRST, FIRST + ROUNDS):
        if ROUNDS > 1 and rnd % 20 == 0:
            print(f"triangle/sweep soak round {rnd}", flush=True)
        rng = np.random.default_rng(77 + 7919 * rnd)
        for case in range(80):
            n, n_r = int(rng.integers(4, 900)), int(rng.integers(1, 900))
            side = float(rng.choice([10.0, 200.0]))
            xy = _points(rng, n, side, int(rng.integers(0, 3)))
            try:
                tris = Delaunay(xy).simplices.astype(np.int32)
            except Exception:  # degenerate input for Qhull: use random triples (incl. repeated vertices)
                tris = rng.integers(0, n, (int(rng.integers(1, 300)), 3)).astype(np.int32)
            if case % 5 == 0:
                tris = np.vstack([tris, rng.integers(0, n, (20, 3)).astype(np.int32)])  # arbitrary, possibly degenerate triples
            rxy = _points(rng, n_r, side, int(rng.integers(0, 4)))
            types = rng.integers(0, 3, n).astype(np.int32)
            size = rng.integers(1, 5, n).astype(float)
            mad = rng.choice([None, 0, 5, 15, 45, 60, 90])
            mad = None if mad is None else float(mad)
            radius = float(rng.choice([side / 20, side / 5, side * 3]))
            en, thr = oracle.cos_threshold(mad)
            tid = types if case % 2 else None
            cls, perim, mc = ops.tri_classify(xy, tris, radius, en, thr, tid)
            ocls, operim, omc = oracle.tri_classify(xy, tris, radius, mad, tid)
            assert np.array_equal(cls, ocls) and np.array_equal(perim, operim) and np.array_equal(mc, omc, equal_nan=True), case
            sign, wgt = ops.tri_sign_weight(xy, size, tris)
            osign, owgt = oracle.tri_sign_weight(xy, size, tris)
            assert np.array_equal(sign, osign) and np.array_equal(wgt, owgt), case
            match = rng.integers(-1, n_r, n).astype(np.int32)
            match[rng.random(n) < rng.choice([0.0, 0.3, 0.9])] = -1
            sweep = ops.BoundSweep(tris, sign, rxy, n)
            checked, viol, flag = sweep.sweep_match(match, want_flag=True)
            och, oviol, oflag = oracle.orient_sweep(tris, sign, rxy, match)
            assert checked == och and np.array_equal(viol, oviol) and np.array_equal(flag, oflag), case
            e, tf, pf, counts = ops.xyorder_sweep(xy, rxy, tris, match)
            oe, otf, opf, oc = oracle.xyorder_sweep(xy, rxy, tris, match)
            assert np.array_equal(e, oe) and np.array_equal(tf, otf) and np.array_equal(pf, opf) and np.array_equal(counts, oc), case
            b, a, m3, fl = ops.area_flip(xy, rxy, tris, match)
            ob, oa, om3, ofl = oracle.area_flip(xy, rxy, tris, match)
            assert np.array_equal(b, ob) and np.array_equal(a, oa, equal_nan=True) and np.array_equal(m3, om3) and np.array_equal(fl,
                                                                      ofl), case
            mapped = np.where((match >= 0)[:, None], rxy[np.maximum(match, 0)], 0.0)
            f, nt, nf = ops.tri_flip_stats(xy, mapped, (match >= 0), tris, tid)
            of, ont, onf = oracle.tri_flip_stats(xy, mapped, (match >= 0), tris, tid)
            assert np.array_equal(f, of) and np.array_equal(nt, ont) and np.array_equal(nf, onf), case


def test_fuzz_matching_from_x_and_greedy(ops, oracle):
    for rnd in range(FIRST, FIRST + ROUNDS):
        if ROUNDS > 1 and rnd % 20 == 0:
            print(f"matching soak round {rnd}", flush=True)
        rng = np.random.default_rng(5 + 7919 * rnd)
        for case in range(60):
            n_m, n_r = int(rng.integers(1, 400)), int(rng.integers(1, 400))
            P = int(rng.integers(0, 3000))
            pairs = np.column_stack((np.sort(rng.integers(0, n_m, P)), rng.integers(0, n_r, P))).astype(np.int32)
            x = rng.random(P) * rng.choice([0.6, 1.0, 1.4])
            tris = rng.integers(0, n_m, (int(rng.integers(1, 200)), 3)).astype(np.int32)
            sign = rng.integers(-1, 2, len(tris)).astype(np.int8)
            rxy = rng.uniform(0, 50, (n_r, 2))
            sw = ops.BoundSweep(tris, sign, rxy, n_m, pairs)
            checked, viol, match, pidx = sw.sweep_x(x)
            omatch, opidx = oracle.matching_from_x(x, pairs.tolist(), n_m)
            assert np.array_equal(match, omatch) and np.array_equal(pidx.astype(np.int64), opidx), case
            och, oviol, _ = oracle.orient_sweep(tris, sign, rxy, omatch)
            assert checked == och and np.array_equal(viol, oviol), case
            costs = np.round(rng.gamma(2.0, 5.0, P), int(rng.choice([1, 6])))
            prefer = rng.random(n_m) < 0.8
            mp, _ = ops.greedy_match(pairs, costs, n_m, n_r, prefer.astype(np.uint8))
            used_a, used_r, want = set(), set(), np.full(n_m, -1, np.int32)
            for q in np.argsort(costs, kind="stable"):
                i, j = int(pairs[q, 0]), int(pairs[q, 1])
                if i in used_a or j in used_r or not prefer[i]:
                    continue
                want[i] = q; used_a.add(i); used_r.add(j)
            assert np.array_equal(mp, want), case
            un = rng.uniform(1, 100, n_m)
            assert np.array_equal(ops.pair_rowmin(pairs, costs, n_m),
                                  np.array([costs[pairs[:, 0] == i].min() if (pairs[:,
                                                                      0] == i).any() else np.inf for i in range(n_m)])), case
            o = np.empty((n_m, n_r + n_m))
            oracle.lib().orc_assign_matrix(pairs, costs, P, un, n_m, n_r, 1e9, o.reshape(-1))
            assert np.array_equal(ops.assign_matrix(pairs, costs, un, n_m, n_r, 1e9), o), case


def test_fuzz_knn_index_and_block_sweeps(ops, oracle):
    """Caller-held KNN index == un-indexed prune on random shapes / row blocks; the orientation sweep issued as the triangle
    blocks of a random number of ranks == the whole sweep."""
    import ctypes
    from same_amd import _lib
    from same_amd.dist import tri_block

    ctx = _lib.default_context()
    L, H = ctx.lib, ctx.handle
    for rnd in range(FIRST, FIRST + ROUNDS):
        rng = np.random.default_rng(911 + 7919 * rnd)
        for case in range(40):
            n_m, n_r = int(rng.integers(1, 600)), int(rng.choice([0, 1, 50, 900, 2100, 4000]))
            side = float(rng.choice([10.0, 100.0, 1000.0]))
            radius = float(rng.choice([0.0, 0.5, 3.0, side / 10, side / 3, side * 2]))
            k = int(rng.choice([1, 5, 32, 64, 65, 200]))
            axy, rxy = _points(rng, n_m, side, int(rng.integers(0, 4))), _points(rng, n_r, side, int(rng.integers(0, 4)))
            with ctx.lock:
                dax, drx = ctx.to_device(axy), ctx.to_device(rxy)
                ix = ctypes.c_void_p()
                ctx.check(L.same_knn_index_build(H, drx.ptr, n_r, radius, ctypes.byref(ix)), "index")
                b, e = sorted(int(v) for v in rng.integers(0, n_m + 1, 2))
                rows = e - b
                di, dd, dc = ctx.alloc(max(rows, 1) * k * 4), ctx.alloc(max(rows, 1) * k * 8), ctx.alloc(max(rows, 1) * 4)
                ctx.check(L.same_knn_prune_indexed_dev(H, ix, dax.ptr, b, e, k, di.ptr, dd.ptr, dc.ptr), "indexed")
                got = di.download((rows, k), np.int32), dd.download((rows, k), np.float64), dc.download((rows,), np.int32)
                L.same_knn_index_destroy(ix)
            want = oracle.knn_prune(axy, rxy, radius, k, b, e)
            assert all(np.array_equal(g, w_) for g, w_ in zip(got, want)), (case, n_m, n_r, k, radius, b, e)
            # triangle blocks
            n = max(n_m, 3)
            Tr = int(rng.integers(1, 2500))
            tris = rng.integers(0, n, (Tr, 3)).astype(np.int32)
            sign = rng.integers(-1, 2, Tr).astype(np.int8)
            rr = _points(rng, max(n_r, 1), side, 0)
            match = rng.integers(-1, len(rr), n).astype(np.int32)
            och, oviol, oflag = oracle.orient_sweep(tris, sign, rr, match)
            world = int(rng.choice([1, 2, 3, 5, 8]))
            with ctx.lock:
                sw = ctypes.c_void_p()
                ctx.check(L.same_sweep_bind(H, tris.ctypes.data, Tr, sign.ctypes.data, rr.ctypes.data, len(rr), n, None, 0,
                                            ctypes.byref(sw)), "bind")
                dmatch = ctx.to_device(match)
                _, _, block = tri_block(Tr, world, 0)
                dflag = ctx.alloc(block * world)
                ctx.check(L.same_dev_memset(H, dflag.ptr, 0, dflag.nbytes), "memset")
                for r in rng.permutation(world):                     # ranks finish in any order
                    t0, t1, _ = tri_block(Tr, world, int(r))
                    ctx.check(L.same_orient_flags_dev(sw, dmatch.ptr, t0, t1, dflag.ptr), "flags")
                chk, nv = ctypes.c_int64(0), ctypes.c_int64(0)
                viol = np.empty(Tr, np.int32)
                ctx.check(L.same_orient_from_flags_dev(sw, dflag.ptr, ctypes.byref(chk), viol.ctypes.data, ctypes.byref(nv)), "from_flags")
                flags = dflag.download((Tr,), np.uint8)
                L.same_sweep_unbind(sw)
            assert chk.value == och and np.array_equal(viol[: nv.value], oviol) and np.array_equal(flags, oflag), (case, Tr, world)


def test_fuzz_merge_dedup(ops, oracle):
    """The window-merge de-duplication (csrc/merge.hip) against the oracle on random tables: sizes that fall on either side of
    the wave, of the 2 048-key LDS sort block and of the first global bitonic stages; few or many windows (long runs of equal
    sort keys apart from the row index); pair codes drawn from tiny to sparse id spaces; all / none / some rows violating."""
    for rnd in range(FIRST, FIRST + ROUNDS):
        if ROUNDS > 1 and rnd % 20 == 0:
            print(f"merge soak round {rnd}", flush=True)
        rng = np.random.default_rng(77 + 104729 * rnd)
        for case in range(60):
            n = int(rng.choice([0, 1, 2, 63, 64, 65, 127, 2047, 2048, 2049, 4095, 4097, int(rng.integers(3, 9000)),
                                int(rng.integers(9000, 40000))]))
            n_win = int(rng.choice([1, 2, 7, 300, 2 ** 31 - 1]))
            ids = int(rng.choice([1, 3, 50, 5000, 2 ** 31 - 1]))
            viol = rng.random(n) < float(rng.choice([0.0, 0.3, 1.0]))
            win = rng.integers(0, n_win, n, endpoint=(n_win < 2 ** 31 - 1)).astype(np.int32) if n_win > 1 else np.zeros(n, np.int32)
            a = rng.integers(0, ids, n).astype(np.int32)
            r = rng.integers(0, ids, n).astype(np.int32)
            got = ops.merge_dedup(viol, win, a, r)
            assert np.array_equal(got, oracle.merge_dedup(viol, win, a, r)), (rnd, case, n, n_win, ids)


def test_fuzz_device_windows(ops):
    """The device-resident window path (csrc/window.hip) against the column pipeline + host-buffer entry points on random small
    sections: ragged sizes, T from 0, k across both prune kernels' capacities, lattice / clustered / duplicate points, radii from
    tiny to whole-section, fp32 and fp64 costs, integer and float sizes, boxes that are empty on one side, every filter setting."""
    from scipy.spatial import QhullError
    from window_check import check_window

    from same_amd import windows as W

    for rnd in range(FIRST, FIRST + ROUNDS):
        if ROUNDS > 1 and rnd % 20 == 0:
            print(f"device window soak round {rnd}", flush=True)
        rng = np.random.default_rng(77 + 104729 * rnd)
        done = errors = max_rounds = 0
        for case in range(36):
            n_r, n_m = int(rng.integers(5, 2500)), int(rng.integers(5, 2500))
            T, k = int(rng.choice([0, 1, 3, 8, 20, 33])), int(rng.choice([1, 2, 5, 8, 31, 64, 65, 130]))
            side = float(rng.choice([50.0, 400.0]))
            radius = float(rng.choice([0.5, 4.0, side / 8, side / 2, side * 2]))
            mode_r, mode_m = int(rng.integers(0, 3)), int(rng.integers(0, 3))
            rxy, mxy = _points(rng, n_r, side, mode_r), _points(rng, n_m, side, mode_m)
            size = rng.integers(1, 4, n_m) if case % 2 else rng.uniform(0.5, 3.0, n_m)
            ref_sec = W.Section(rxy, rng.gamma(0.3, 30.0, (n_r, T)), rng.integers(0, 3, n_r).astype(np.int32), None)
            mov_sec = W.Section(mxy, rng.gamma(0.3, 30.0, (n_m, T)), rng.integers(0, 3, n_m).astype(np.int32) if case % 5 else None, size)
            dt = "float32" if case % 3 else "float64"
            dref, dmov = W.DeviceSection(ref_sec, dt), W.DeviceSection(mov_sec, dt)
            cut = float(rng.uniform(0.2, 0.8)) * side
            plan = [dict(box=(-1.0, side + 5.0, -1.0, side + 5.0)), dict(box=(-1.0, cut, -1.0, side + 5.0)),
                    dict(box=(cut, side + 5.0, cut, side + 5.0)), dict(box=(side + 10.0, side + 20.0, 0.0, 1.0))]
            kw = dict(radius=radius, knn=k, dist_ct_coeff=float(rng.choice([1.0, 0.3])), min_angle_deg=[15, None, 35][case % 3],
                      ignore_same_type_triangles=bool(case % 4))
            penalty = float(rng.choice([100.0, 5.0, 0.05]))
            try:
                arrays = list(W.iter_window_arrays(ref_sec, mov_sec, plan, cost_dtype=dt, **kw))
            # a window whose kept cells Qhull cannot triangulate (collinear / too few): both forms raise
            except (QhullError, ValueError):
                with pytest.raises((QhullError, ValueError)):
                    list(W.iter_device_windows(ref_sec, mov_sec, dref, dmov, plan, no_match_penalty=penalty, fetch_triangles=True, **kw))
                errors += 1
                continue
            for wa, dw in zip(arrays, W.iter_device_windows(ref_sec, mov_sec, dref, dmov, plan, no_match_penalty=penalty,
                                                            fetch_triangles=True, **kw)):
                assert (wa.error is None) == (dw.error is None), (case, wa.window)
                if wa.error is None:
                    check_window(W, ops, wa, dw, penalty)
                    done += 1
                    max_rounds = max(max_rounds, dw.stats["greedy_rounds"])
            dref.close()
            dmov.close()
        assert done > 30, (done, errors)
        # some window needs more greedy rounds than the finish call enqueues up front: the "keep going" path ran
        assert max_rounds > 3, max_rounds


def test_device_window_argument_checks(ops):
    """The window entry points refuse what they cannot run instead of running it: sections that do not belong together, a fetch of
    the wrong size or before its producer, triangles out of range, finish without stage."""
    from same_amd import windows as W
    from same_amd._lib import SameHipError

    rng = np.random.default_rng(5)
    a = W.Section(rng.uniform(0, 100, (400, 2)), rng.random((400, 3)), rng.integers(0, 2, 400).astype(np.int32), None)
    b = W.Section(rng.uniform(0, 100, (300, 2)), rng.random((300, 4)), None, None)          # another number of type columns
    da, da32, db = W.DeviceSection(a, "float64"), W.DeviceSection(a, "float32"), W.DeviceSection(b, "float64")
    st = W.DeviceWindow()
    box = (0.0, 100.0, 0.0, 100.0)
    with pytest.raises(SameHipError):
        st.finish(np.zeros((1, 3), np.int32), 1.0)                  # nothing staged
    for bad in ((da, db), (da, da32)):                             # T differs; cost types differ
        with pytest.raises(SameHipError):
            st.stage(bad[0], bad[1], box, 10.0, 4, 1.0)
    with pytest.raises(SameHipError):
        st.stage(da, da, box, -1.0, 4, 1.0)
    with pytest.raises(SameHipError):
        st.stage(da, da, box, 10.0, 0, 1.0)
    assert st.stage(da, da, (200.0, 300.0, 0.0, 1.0), 10.0, 4, 1.0) == (0, 0, 0, 0)          # nobody in the box
    assert len(st.fetch(W._W_ROWS_M)) == 0 and len(st.fetch(W._W_ALIGNED_XY)) == 0
    n_m, n_r, kept, n_pairs = st.stage(da, da, box, 10.0, 4, 1.0)
    assert n_m == n_r == kept == 400 and 400 <= n_pairs <= 1600
    with pytest.raises(SameHipError):
        st.fetch(W._W_SIGNS)                                         # before finish
    with pytest.raises(SameHipError):
        st.ctx.check(st.ctx.lib.same_window_fetch(st.handle, W._W_PAIRS, np.zeros(4, np.int32).ctypes.data, 16), "fetch")   # wrong size
    with pytest.raises(SameHipError):
        # unknown selector
        st.ctx.check(st.ctx.lib.same_window_fetch(st.handle, 99, np.zeros(4, np.int32).ctypes.data, 16), "fetch")
    with pytest.raises(SameHipError):
        st.finish(np.array([[0, 1, 400]], np.int32), 1.0)            # vertex out of range: reported, not dereferenced
    with pytest.raises(SameHipError):
        st.filter_finish(np.array([[0, 1, -1]], np.int32), 10.0, 1, 0.5, 0.0, True, 1.0)
    # batches: a window twice, windows of two contexts, more windows than SAME_WINDOW_BATCH_MAX
    from same_amd import _lib
    st2, other = W.DeviceWindow(), _lib.Context(st.ctx.device)
    st3 = W.DeviceWindow(other)
    for bad in ([st, st], [st, st3], [st] * (W.WINDOW_BATCH_MAX + 1)):
        with pytest.raises(SameHipError):
            W.stage_windows(bad, da, da, [box] * len(bad), 10.0, 4, 1.0)
    # ... and a batch of two gives what two single calls give (the first window of the batch is the one staged above)
    half = (0.0, 50.0, 0.0, 100.0)
    assert W.stage_windows([st, st2], da, da, [box, half], 10.0, 4, 1.0) == [(n_m, n_r, kept, n_pairs),
                                                                      st2.stage(da, da, half, 10.0, 4, 1.0)]
    from scipy.spatial import Delaunay
    tri = [Delaunay(s_.fetch(W._W_ALIGNED_XY)).simplices for s_ in (st, st2)]
    both = W.filter_finish_windows([st, st2], tri, 10.0, 1, 0.9, 0.0, True, 1.0)
    for s_, t_, got in zip((st, st2), tri, both):
        one = s_.filter_finish(t_, 10.0, 1, 0.9, 0.0, True, 1.0)
        assert got[:3] == one[:3] and np.array_equal(got[3], one[3]) and np.array_equal(got[4], one[4]) and got[5] == one[5]
        kept_tris = s_.fetch(W._W_TRIANGLES)
        again = s_.finish(kept_tris, 1.0)               # the kept triangles handed back as the caller's own: the same match and sweeps
        assert np.array_equal(again[0], one[3]) and np.array_equal(again[1], one[4]) and again[2] == one[5]
    # a batch in which ONE window is refused is refused as a whole, before anything is enqueued; the windows stay usable
    bad_tri = [tri[0], np.array([[0, 1, 10 ** 6]], np.int32)]
    with pytest.raises(SameHipError):
        W.filter_finish_windows([st, st2], bad_tri, 10.0, 1, 0.9, 0.0, True, 1.0)
    with pytest.raises(SameHipError):
        W.stage_windows([st, st2], da, db, [box, half], 10.0, 4, 1.0)          # sections that do not belong together
    assert W.stage_windows([st, st2], da, da, [box, half], 10.0, 4, 1.0)[0] == (n_m, n_r, kept, n_pairs)
    redo = W.filter_finish_windows([st, st2], tri, 10.0, 1, 0.9, 0.0, True, 1.0)
    assert all(np.array_equal(a_[3], b_[3]) and a_[5] == b_[5] for a_, b_ in zip(redo, both))
    st2.close()
    st3.close()
    other.close()
    st.stage(da, da, box, 10.0, 4, 1.0)
    row, flag, stats = st.finish(np.zeros((0, 3), np.int32), 1.0)    # no triangles at all: a match and empty sweeps
    assert stats["checked"] == stats["xy_comparisons"] == stats["area_flips"] == 0 and stats["matched"] == np.count_nonzero(row >= 0) > 0
    assert not flag.any()
    for h in (st, da, da32, db):
        h.close()


def test_window_calls_with_more_windows_than_a_launch_takes():
    """Every kernel of the two window calls takes up to eight windows per launch (blockIdx.y = window).  A call of 23 windows of very
    different sizes -- whole section, slivers, empty boxes, boxes that cut through cells -- runs as groups of 8 + 8 + 7; one filter_finish
    call over windows staged on TWO pairs of sections (fp64 costs with cell types; fp32 costs without) is cut into groups that agree on
    the cost type and on whether same-type triangles come back.  Every window must get what a call of its own gives."""
    from scipy.spatial import Delaunay

    from same_amd import windows as W

    rng = np.random.default_rng(21)
    xy = rng.uniform(0, 300, (9000, 2))
    a = W.Section(xy, rng.gamma(0.3, 30.0, (9000, 4)), rng.integers(0, 3, 9000).astype(np.int32), rng.integers(1, 4, 9000))
    b = W.Section(xy + rng.normal(0, 0.5, xy.shape), rng.gamma(0.3, 30.0, (9000, 4)), None, None)
    da, db = W.DeviceSection(a, "float64"), W.DeviceSection(b, "float32")
    boxes = [(0.0, 300.0, 0.0, 300.0), (310.0, 320.0, 0.0, 10.0), (0.0, 2.0, 0.0, 300.0)]
    for _ in range(20):
        x0, y0 = rng.uniform(0, 250, 2)
        boxes.append((float(x0), float(x0 + rng.uniform(3, 120)), float(y0), float(y0 + rng.uniform(3, 120))))
    n = len(boxes)
    states = [W.DeviceWindow() for _ in range(n)]
    solo = W.DeviceWindow()
    kw = (12.0, 6, 1.0)
    counts = W.stage_windows(states, da, da, boxes, *kw)
    assert len({c[2] for c in counts}) > 15 and (0, 0, 0, 0) in counts        # sizes from nothing to the whole section
    tris, want = [], []
    for st, box, c in zip(states, boxes, counts):
        assert solo.stage(da, da, box, *kw) == c
        for what in (W._W_PAIRS, W._W_COSTS, W._W_ROWS_M, W._W_ALIGNED_ROWS):
            assert np.array_equal(st.fetch(what), solo.fetch(what))
        pts = st.fetch(W._W_ALIGNED_XY)
        tris.append(Delaunay(pts).simplices if len(pts) >= 3 else np.zeros((0, 3), np.int32))
        want.append(solo.filter_finish(tris[-1], 12.0, 1, 0.9, 0.0, True, 50.0) if c[2] else None)
    # the same boxes' first seven on the other pair of sections, then ONE filter_finish call over all windows, the pairs interleaved
    states32 = [W.DeviceWindow() for _ in range(7)]
    counts32 = W.stage_windows(states32, db, db, boxes[:7], *kw)
    tris32, want32 = [], []
    for st, box, c in zip(states32, boxes[:7], counts32):
        assert solo.stage(db, db, box, *kw) == c
        pts = st.fetch(W._W_ALIGNED_XY)
        tris32.append(Delaunay(pts).simplices if len(pts) >= 3 else np.zeros((0, 3), np.int32))
        want32.append(solo.filter_finish(tris32[-1], 12.0, 1, 0.9, 0.0, True, 50.0) if c[2] else None)
    # a box that is empty on one side is the caller's error case: not finished
    order = [("a", q) for q in range(n) if counts[q][0] and counts[q][1]]
    for q in range(7):
        if counts32[q][0] and counts32[q][1]:
            order.insert(3 * q + 1, ("b", q))
    assert len(order) > 24
    mixed = [states[q] if k == "a" else states32[q] for k, q in order]
    got = W.filter_finish_windows(mixed, [tris[q] if k == "a" else tris32[q] for k, q in order], 12.0, 1, 0.9, 0.0, True, 50.0)
    checked = 0
    for (k, q), g in zip(order, got):
        w_ = (want if k == "a" else want32)[q]
        if w_ is None:
            continue
        assert g[:3] == w_[:3] and np.array_equal(g[3], w_[3]) and np.array_equal(g[4], w_[4]) and g[5] == w_[5], (k, q)
        st = (states if k == "a" else states32)[q]
        assert len(st.fetch(W._W_TRIANGLES)) == g[0] + g[1]
        checked += 1
    assert checked >= 25
    for h in states + states32 + [solo, da, db]:
        h.close()


def test_prune_indices_dropped_while_other_threads_use_them():
    """A section keeps at most 16 prune indices (one per radius, least recently used dropped).  Three threads with a context each cycle
    24 radii over the SAME sections, each in its own order, so that indices are dropped from the table while another thread's stage call
    prunes with them: every call must still give what a single thread gives for its radius (the call holds its index until it has waited)."""
    import threading

    from same_amd import _lib
    from same_amd import windows as W

    rng = np.random.default_rng(11)
    # >= 2048 rows: grid indices (device arrays of their own)
    sec = W.Section(rng.uniform(0, 200, (6000, 2)), rng.random((6000, 3)), None, None)
    dsec = W.DeviceSection(sec, "float64")
    radii = [3.0 + 0.37 * q for q in range(24)]
    box = (0.0, 200.0, 0.0, 200.0)
    solo = W.DeviceWindow()
    want = {}
    for r in radii:
        counts = solo.stage(dsec, dsec, box, r, 6, 1.0)
        want[r] = (counts, solo.fetch(W._W_PAIRS).copy(), solo.fetch(W._W_COSTS).copy())
    solo.close()
    failures = []

    def walk(seed):
        ctx = _lib.Context(dsec.ctx.device)
        st = W.DeviceWindow(ctx)
        try:
            order = np.random.default_rng(seed).permutation(len(radii))
            for rep in range(3):
                for q in order:
                    r = radii[int(q)]
                    counts = st.stage(dsec, dsec, box, r, 6, 1.0)
                    if counts != want[r][0] or not np.array_equal(st.fetch(W._W_PAIRS),
                                                                  want[r][1]) or not np.array_equal(st.fetch(W._W_COSTS), want[r][2]):
                        failures.append((seed, rep, r))
        except BaseException as e:   # noqa: BLE001 -- reported by the assertion below
            failures.append((seed, repr(e)))
        finally:
            st.close()
            ctx.close()

    threads = [threading.Thread(target=walk, args=(s_,)) for s_ in (1, 2, 3)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not failures, failures[:5]
    # Radii nobody has asked for yet, asked for AT THE SAME MOMENT: two threads the same one (the table's lock is not held across a build,
    # so both build it; the second to finish drops its build and takes the first's), two more a radius each -- while a fifth keeps staging
    # against a radius that is there.  Everybody gets the single thread's answer; four rounds of fresh radii.
    fresh = [[40.0 + rnd, 40.0 + rnd, 50.0 + rnd, 60.0 + rnd, radii[0]] for rnd in range(4)]
    solo = W.DeviceWindow()
    for r in sorted({r for rnd in fresh for r in rnd}):
        want[r] = (solo.stage(dsec, dsec, box, r, 6, 1.0), solo.fetch(W._W_PAIRS).copy(), solo.fetch(W._W_COSTS).copy())
    solo.close()
    dsec.close()
    dsec = W.DeviceSection(sec, "float64")            # a section whose table is empty again
    gate = threading.Barrier(5)

    def race(lane):
        ctx = _lib.Context(dsec.ctx.device)
        st = W.DeviceWindow(ctx)
        try:
            for rnd in range(4):
                gate.wait(60)
                r = fresh[rnd][lane]
                for _rep in range(3 if lane == 4 else 1):
                    counts = st.stage(dsec, dsec, box, r, 6, 1.0)
                    if counts != want[r][0] or not np.array_equal(st.fetch(W._W_PAIRS),
                                                                  want[r][1]) or not np.array_equal(st.fetch(W._W_COSTS), want[r][2]):
                        failures.append(("race", lane, rnd, r))
        except BaseException as e:   # noqa: BLE001 -- reported by the assertion below
            failures.append(("race", lane, repr(e)))
            gate.abort()
        finally:
            st.close()
            ctx.close()

    threads = [threading.Thread(target=race, args=(lane,)) for lane in range(5)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    dsec.close()
    assert not failures, failures[:5]


def test_scans_hold_when_no_block_ever_sees_a_predecessor():
    """csrc/scan.h never waits: a block that finds a predecessor's word unpublished recomputes that block's counters from the input.
    In a normal run that path is rare and timing-dependent, so here it carries everything: SAME_SCAN_FORCE_RECOMPUTE=1 makes every
    look-back treat every predecessor as unpublished.  The window path, the merge, the prune index and the sweeps must not notice."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SAME_SCAN_FORCE_RECOMPUTE="1", SAME_FUZZ_ROUNDS="1")
    sel = ["tests/test_gpu_fuzz.py::test_fuzz_device_windows", "tests/test_gpu_fuzz.py::test_fuzz_knn_and_costs",
           "tests/test_gpu_run_same.py::test_device_windows_equal_the_column_pipeline",
           "tests/test_gpu_run_same.py::test_window_rows_do_not_depend_on_the_section_grid",
           "tests/test_gpu_parity.py::test_sharded_sweeps_rccl_single_rank_and_block_forms",
           "tests/test_host_rows.py::test_merge_dedup_on_device",
           "tests/test_gpu_merge.py", "tests/test_gpu_fuzz.py::test_fuzz_window_merge_routes_agree"]
    res = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-k", "not scans_hold"] + sel, cwd=root, env=env,
                         capture_output=True, text=True, timeout=1500)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-1000:]
    assert " passed" in res.stdout


def _assert_prepared_equal(a, b, what):
    assert type(a) is type(b) or (isinstance(a, Exception) and isinstance(b, Exception)), what
    if isinstance(a, Exception):
        assert str(a) == str(b), what
        return 0
    assert np.array_equal(np.asarray(a.valid_pairs, dtype=np.int64).reshape(-1, 2),
                          np.asarray(b.valid_pairs, dtype=np.int64).reshape(-1, 2)), what
    assert np.array_equal(a.costs_array, b.costs_array) and a.costs_array.dtype == b.costs_array.dtype, what
    assert np.array_equal(a.triangles_array, b.triangles_array), what
    assert np.array_equal(a.signs_array,
                          b.signs_array) and a.weights_array.dtype == b.weights_array.dtype and np.array_equal(a.weights_array,
                                                                      b.weights_array), what
    assert (a.n_aligned, a.n_ref) == (b.n_aligned, b.n_ref) == (len(a.aligned_df), len(a.ref_df)), what
    for fa, fb in ((a.aligned_df, b.aligned_df), (a.ref_df, b.ref_df)):
        assert list(fa.columns) == list(fb.columns) and fa.equals(fb), what
    return 1


def test_fuzz_window_pipelines_agree():
    """The two pipelines behind sliding_window_matching -- frames resident on the device vs frames cut on the host -- on random small
    jobs the fixed tests do not visit: rows with NaN / infinite coordinates, duplicate points and lattices, string or shuffled index
    labels, integer-typed type columns and sizes, missing `size`, k larger than a window's references, radii from too small for any pair
    to window-sized, no angle rule, same-type triangles kept, the priority filter, fp32 costs, window sizes whose grid the boxes cut
    through.  Every window's PreparedInputs (pairs, costs, triangles, weights, signs, BOTH frames) must be equal, errors included, and so
    must the solver-free tables of all three routes."""
    import pandas as pd

    import same_amd
    from same_amd.windows import window_plan

    done = errors = 0
    for rnd in range(FIRST, FIRST + ROUNDS):
        if ROUNDS > 1 and rnd % 20 == 0:
            print(f"window pipelines soak round {rnd}", flush=True)
        rng = np.random.default_rng(31337 + 104729 * rnd)
        for case in range(14):
            n_r, n_m, T = int(rng.integers(150, 1400)), int(rng.integers(150, 1400)), int(rng.integers(1, 7))
            side = float(rng.choice([120.0, 400.0]))
            frames = []
            for n in (n_r, n_m):
                xy = _points(rng, n, side, int(rng.choice([0, 0, 0, 1, 2])))
                df = pd.DataFrame(rng.gamma(0.3, 30.0, (n, T)), columns=[f"t{q}" for q in range(T)])
                if case % 5 == 1:
                    df["t0"] = (df["t0"] * 3).astype(np.int64)            # an integer-typed type column
                df.insert(0, "Y", xy[:, 1])
                df.insert(0, "X", xy[:, 0])
                df["cell_type"] = rng.choice(np.array(["a", "b", "c"], dtype=object), n)
                if case % 3:
                    df["size"] = rng.integers(1, 4, n) if case % 2 else rng.integers(1, 4, n) * 1.5
                df["Cell_Num_Old"] = rng.permutation(n) + 10
                if case % 4 == 2:
                    # rows no window holds (pandas' min / max skip them: src/same.py:481-482)
                    df.loc[df.index[rng.integers(0, n, 5)], "X"] = np.nan
                if case == 13:
                    df.loc[df.index[rng.integers(0, n, 2)], "Y"] = np.inf       # int(inf): the reference's OverflowError
                if case % 6 == 3:
                    df.index = [f"cell{q}" for q in rng.permutation(n)]
                elif case % 6 == 4:
                    df.index = rng.permutation(n) * 2
                frames.append(df)
            ref, mov = frames
            cols = [f"t{q}" for q in range(T)]
            ws = int(rng.choice([60, 90, 150, 400]))
            op = dict(radius=float(rng.choice([0.5, 6.0, 15.0, 40.0])), knn=int(rng.choice([1, 3, 8, 40])), window_size=ws,
                      overlap=int(rng.choice([0, ws // 4, ws // 3])),
                      min_cells_per_window=int(rng.choice([5, 30])), dist_ct_coeff=float(rng.choice([1.0, 0.4])),
                      min_angle_deg=[15, None, 30][case % 3],
                      ignore_same_type_triangles=bool(case % 4), hip_cost_dtype="float32" if case % 3 == 0 else "float64",
                      no_match_penalty=float(rng.choice([100.0, 2.0])), ignore_knn_if_matched=(case % 7 == 5))
            try:
                plan = window_plan(ref[["X", "Y"]].to_numpy(dtype=np.float64), mov[["X", "Y"]].to_numpy(dtype=np.float64), ws,
                                   op["overlap"], op["min_cells_per_window"])
            # an infinite extent: int(inf) in the reference's grid (src/same.py:481-488) -- both pipelines raise before any window
            except OverflowError:
                for pipe in ("device", "frames"):
                    with pytest.raises(OverflowError):
                        same_amd.sliding_window_incumbent(ref, mov, commonCT=cols, optim_params=dict(op), _pipeline=pipe)
                errors += 1
                continue
            if not plan:
                continue
            from scipy.spatial import QhullError

            try:
                b = list(same_amd.iter_prepared_windows(ref, mov, cols, plan, optim_params=dict(op), pipeline="frames"))
            except (QhullError, ZeroDivisionError) as stop:
                # a window whose kept cells Qhull cannot triangulate (too few / collinear), or -- with the cell-type-priority prune -- one
                # without any pair (its summary print divides by zero, src/knn_utils.py:76): run_same dies there, on either pipeline
                with pytest.raises((QhullError, ZeroDivisionError)):
                    list(same_amd.iter_prepared_windows(ref, mov, cols, plan, optim_params=dict(op), pipeline="device"))
                for pipe in ("device", "frames"):
                    with pytest.raises((QhullError, ValueError, ZeroDivisionError)):
                        same_amd.sliding_window_incumbent(ref, mov, commonCT=cols, optim_params=dict(op), _pipeline=pipe)
                errors += 1
                del stop
                continue
            a = list(same_amd.iter_prepared_windows(ref, mov, cols, plan, optim_params=dict(op), pipeline="device"))
            assert len(a) == len(b) == len(plan)
            ok = [_assert_prepared_equal(pa, pb, (rnd, case, w["window_id"])) for (w, pa), (_w, pb) in zip(a, b)]
            tables = []
            for kw in (dict(_pipeline="device"), dict(_route="general", _pipeline="device"), dict(_route="general", _pipeline="frames")):
                try:
                    tables.append(same_amd.sliding_window_incumbent(ref, mov, commonCT=cols, optim_params=dict(op),
                                                                    window_local_indices=True, **kw))
                except ValueError as e:
                    tables.append(str(e))
            if sum(ok) < len(ok):                           # a window without pairs: every route raises run_same's error
                assert all(isinstance(t, str) and t == tables[0] and "No valid_pairs" in t for t in tables), (rnd, case)
                errors += 1
            else:
                assert all(isinstance(t, pd.DataFrame) for t in tables), (rnd, case, tables)
                for t in tables[1:]:
                    assert list(t.columns) == list(tables[0].columns) and len(t) == len(tables[0]), (rnd, case)
                    for c in t.columns:
                        assert np.array_equal(t[c].to_numpy(), tables[0][c].to_numpy()), (rnd, case, c)
                done += 1
    assert done >= 6 * ROUNDS and errors >= 1, (done, errors)


def _random_window_job(rng, case):
    """two random frames + window parameters (a lighter form of test_fuzz_window_pipelines_agree's generator: jobs whose every window
    has pairs, crowded so that neighbouring windows disagree about cells)"""
    import pandas as pd

    n_r, T = int(rng.integers(300, 1500)), int(rng.integers(1, 6))
    side = float(rng.choice([150.0, 400.0]))
    rxy = _points(rng, n_r, side, int(rng.choice([0, 0, 0, 1, 2])))
    copies = int(rng.choice([1, 2, 3]))                         # several suitors per reference cell
    mxy = np.concatenate([rxy[rng.random(n_r) < 0.9] + rng.normal(0, float(rng.choice([0.5, 3.0])), (1, 2)) for _ in range(copies)])
    mxy = mxy + rng.normal(0, 2.0, mxy.shape)
    frames = []
    for xy in (rxy, mxy):
        n = len(xy)
        df = pd.DataFrame(rng.gamma(0.3, 30.0, (n, T)), columns=[f"t{q}" for q in range(T)])
        if case % 5 == 1:
            df["t0"] = (df["t0"] * 3).astype(np.int64)            # an integer-typed type column: the table's columns come from the host
        df.insert(0, "Y", xy[:, 1])
        df.insert(0, "X", xy[:, 0])
        df["cell_type"] = rng.choice(np.array(["a", "b", "c"], dtype=object), n)
        if case % 3:
            df["size"] = rng.integers(1, 4, n) if case % 2 else rng.integers(1, 4, n) * 1.5
        ids = rng.permutation(n) * 3 + 10
        # string ids: codes, host-gathered id columns
        df["Cell_Num_Old"] = ids if case % 4 else np.array([f"c{v:06d}" for v in ids], dtype=object)
        if case % 6 == 2:
            df.loc[df.index[rng.integers(0, n, 4)], "X"] = np.nan
        if case % 6 == 3:
            df.index = [f"cell{q}" for q in rng.permutation(n)]
        frames.append(df)
    ws = int(rng.choice([70, 110, 200]))
    op = dict(radius=float(rng.choice([8.0, 15.0, 30.0])), knn=int(rng.choice([2, 5, 12])), window_size=ws,
              overlap=int(rng.choice([0, 6, ws // 4])),
              min_cells_per_window=int(rng.choice([5, 30])), min_angle_deg=[15, None, 30][case % 3],
              ignore_same_type_triangles=bool(case % 4),
              hip_cost_dtype="float32" if case % 2 else "float64", no_match_penalty=float(rng.choice([100.0, 5.0])))
    return frames[0], frames[1], [f"t{q}" for q in range(T)], op


class _ThreadHub:
    def __init__(self, world):
        import threading

        self.world, self.slots, self.barrier = world, [None] * world, threading.Barrier(world)


class _ThreadMergeChannel:
    """dist.MergeChannel's interface between threads of this process (one rank per thread, a context each)"""

    def __init__(self, hub, rank):
        self.hub, self.rank, self.world, self.sent_rows, self.gather_ms = hub, rank, hub.world, 0, 0.0

    def _all(self, v):
        self.hub.slots[self.rank] = v
        self.hub.barrier.wait(120)
        out = list(self.hub.slots)
        self.hub.barrier.wait(120)
        return out

    def tables(self, table):
        return self._all(table)

    def max(self, v):
        return max(self._all(float(v)))


def test_fuzz_window_merge_routes_agree(oracle):
    """The window table and the window merge on random crowded jobs, every way the product can make them: the table through the device
    accumulator (columns from the device, or from the host where a column is not an 8-byte number) against the per-window host route;
    merge=True on the device against merge_window_matches_unique_ref of that table, against the general route's merge on row keys, and
    dealt over 2-3 ranks (threads, a context each) under either deal -- all equal, row for row, including when every scan is forced to
    recompute (the soak runs this file under SAME_SCAN_FORCE_RECOMPUTE=1 too)."""
    import threading

    import pandas as pd

    import same_amd
    from same_amd import _lib
    from same_amd.merge import join_merged_parts, merge_window_matches_unique_ref

    done = contested = 0
    for rnd in range(FIRST, FIRST + ROUNDS):
        if ROUNDS > 1 and rnd % 20 == 0:
            print(f"window merge soak round {rnd}", flush=True)
        rng = np.random.default_rng(777 + 15485863 * rnd)
        for case in range(10):
            ref, mov, cols, op = _random_window_job(rng, case)
            try:
                host_table = same_amd.sliding_window_incumbent(ref, mov, commonCT=cols, optim_params=dict(op), _route="general",
                                                               _pipeline="device")
            except Exception as e:  # noqa: BLE001 -- a window without pairs / a set Qhull refuses: every route raises alike
                with pytest.raises(type(e)):
                    same_amd.sliding_window_incumbent(ref, mov, commonCT=cols, optim_params=dict(op), merge=True)
                continue
            if len(host_table) == 0:
                continue
            table = same_amd.sliding_window_incumbent(ref, mov, commonCT=cols, optim_params=dict(op))       # the accumulator, end to end
            assert list(table.columns) == list(host_table.columns) and len(table) == len(host_table), (rnd, case)
            for c in table.columns:
                assert table[c].dtype == host_table[c].dtype and np.array_equal(table[c].to_numpy(), host_table[c].to_numpy()), (rnd,
                                                                      case, c)
            want = merge_window_matches_unique_ref([host_table], _dedup=oracle.merge_dedup)
            contested += len(want) < len(host_table)
            got = same_amd.sliding_window_incumbent(ref, mov, commonCT=cols, optim_params=dict(op), merge=True)
            assert list(got.columns) == list(want.columns) and got.equals(want), (rnd, case, len(got), len(want))
            general = same_amd.sliding_window_incumbent(ref, mov, commonCT=cols, optim_params=dict(op), merge=True, _route="general")
            assert general.equals(want), (rnd, case)
            if case % 3 == 0:        # with the windows' own reference indices: the keys go through the host-side builders
                with_idx = same_amd.sliding_window_incumbent(ref, mov, commonCT=cols, optim_params=dict(op), window_local_indices=True)
                merged_idx = same_amd.sliding_window_incumbent(ref, mov, commonCT=cols, optim_params=dict(op), window_local_indices=True,
                                                               merge=True)
                assert merged_idx.equals(merge_window_matches_unique_ref([with_idx], _dedup=oracle.merge_dedup)), (rnd, case)
                assert merged_idx.drop(columns=["ref_idx"]).equals(want), (rnd, case)
            world, deal = int(rng.choice([2, 3])), str(rng.choice(["block", "round_robin"]))
            hub, parts, errors = _ThreadHub(world), [None] * world, []

            def rank_body(rank):
                ctx = _lib.Context(_lib.default_context().device)
                try:
                    route = {} if rank % 2 == 0 else dict(_route="general")        # ranks on different routes send the same seam rows
                    parts[rank] = same_amd.sliding_window_incumbent(ref, mov, commonCT=cols, optim_params=dict(op), merge=True, ctx=ctx,
                                                                    workers=1,
                                                                    _shard=(rank, world, deal),
                                                                    _merge_channel=_ThreadMergeChannel(hub, rank), **route)
                except BaseException as e:  # noqa: BLE001
                    errors.append(e)
                    hub.barrier.abort()
                finally:
                    ctx.close()

            threads = [threading.Thread(target=rank_body, args=(r,)) for r in range(world)]
            [t.start() for t in threads]
            [t.join(300) for t in threads]
            assert not errors, (rnd, case, world, deal, errors[:1])
            joined = join_merged_parts(parts, "Cell_Num_Old")
            assert joined.equals(want), (rnd, case, world, deal)
            done += 1
    assert done >= 5 * ROUNDS and contested >= 2 * ROUNDS, (done, contested)


def test_fuzz_native_triangulator_gives_the_same_tables():
    """Random crowded jobs (the window-merge family's generator, with coordinate styles that provoke order ties: reference cells on whole
    coordinates, aligned cells sharing x values) through `sliding_window_incumbent` with optim_params["hip_delaunay"] = "native" and with
    the default (scipy): the plain table with every window's counters, and the merged table, are identical -- whether a window was
    answered by libsame_hip's triangulator, left to Qhull by it (lattices, duplicates), or finished again with scipy's simplices after
    an order tie.  Errors (a window without pairs, a set Qhull refuses) are the same errors."""
    import same_amd
    from same_amd import delaunay

    tr = delaunay.shared()
    done = native_windows = sent_back = 0
    for rnd in range(FIRST, FIRST + ROUNDS):
        if ROUNDS > 1 and rnd % 20 == 0:
            print(f"native triangulator soak round {rnd}", flush=True)
        rng = np.random.default_rng(4242 + 104729 * rnd)
        for case in range(10):
            ref, mov, cols, op = _random_window_job(rng, case)
            style = int(rng.integers(0, 4))
            if style == 1:
                ref[["X", "Y"]] = np.round(ref[["X", "Y"]].to_numpy())
            elif style == 2:
                mov["X"] = np.round(mov["X"].to_numpy() * 2) / 2
            elif style == 3:
                ref[["X", "Y"]] = ref[["X", "Y"]].to_numpy() + 2.0e4          # far from the origin: Qhull's allowance grows, margins shrink
                mov[["X", "Y"]] = mov[["X", "Y"]].to_numpy() + 2.0e4
            for merge in (False, True):
                kw = dict(commonCT=cols, merge=merge, return_stats=True)
                try:
                    want = same_amd.sliding_window_incumbent(ref, mov, optim_params=dict(op), **kw)
                except Exception as e:  # noqa: BLE001
                    with pytest.raises(type(e)):
                        same_amd.sliding_window_incumbent(ref, mov, optim_params=dict(op, hip_delaunay="native"), **kw)
                    continue
                before = (tr.submitted, tr.asked_qhull)
                got = same_amd.sliding_window_incumbent(ref, mov, optim_params=dict(op, hip_delaunay="native"), **kw)
                assert list(got[0].columns) == list(want[0].columns) and got[0].equals(want[0]), (rnd, case, style, merge)
                assert got[1] == want[1], (rnd, case, style, merge)
                native_windows += tr.submitted - before[0]
                sent_back += tr.asked_qhull - before[1]
                done += 1
    assert done >= 10 * ROUNDS and native_windows > 20 * ROUNDS and 0 < sent_back < native_windows, (done, native_windows, sent_back)
