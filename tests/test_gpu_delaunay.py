"""The window path with libsame_hip's own triangulator (optim_params["hip_delaunay"] = "native", same_amd/delaunay.py) against the
window path with scipy's -- the reference's call (src/same.py:1023).  The triangulator gives Qhull's SET of triangles in another
order; the device counts the places where a window's numbers hang on that order (`order ties`, include/same_hip.h) and such a window
is finished again with scipy's simplices.  Held here: the counter's soundness (no tie counted => any order of the same triangles
gives the same match rows, flags and sweep counters), and the product function's tables being identical both ways."""
import numpy as np
import pandas as pd
import pytest

pytestmark = pytest.mark.gpu


def _reordered(rng, simplices):
    """the same triangles: rows shuffled, every row's corners in a random one of the six orders"""
    s = np.asarray(simplices, np.int32)[rng.permutation(len(simplices))]
    perms = np.array([[0, 1, 2], [1, 2, 0], [2, 0, 1], [0, 2, 1], [2, 1, 0], [1, 0, 2]])
    return np.ascontiguousarray(np.take_along_axis(s, perms[rng.integers(0, 6, len(s))], axis=1))


def _sections(rng, style, n=6000, side=260.0):
    from same_amd import windows as W

    rxy = rng.uniform(0, side, (n, 2))
    mxy = rxy + rng.normal(0, 1.0, rxy.shape)
    if style == "integer_ref":                     # reference cells on whole coordinates: XY-order edges whose ends share an x or a y
        rxy = np.round(rxy / 2) * 2
    elif style == "half_step_x":                   # aligned cells share x values; no four of them on a circle
        mxy[:, 0] = np.round(mxy[:, 0] * 2) / 2
    elif style == "lattice_ref":                   # matched reference triples on a line: signed areas of exactly (or nearly) zero
        g = int(np.sqrt(n)) + 1
        gx, gy = np.meshgrid(np.arange(g) * (side / g) + 1000.5, np.arange(g) * (side / g) + 1000.25)
        rxy = np.c_[gx.ravel(), gy.ravel()][:n]
        mxy = rxy + rng.normal(0, 0.8, rxy.shape)
    elif style == "lattice_moving":                # equal perimeters among a node's same-type triangles
        g = int(np.sqrt(n)) + 1
        gx, gy = np.meshgrid(np.arange(g) * 3.0, np.arange(g) * 3.0)
        mxy = np.c_[gx.ravel(), gy.ravel()][:n]
        rxy = mxy + rng.normal(0, 0.8, mxy.shape)
    T = 3
    ref = W.Section(rxy, rng.gamma(0.3, 30.0, (n, T)), rng.integers(0, 2, n).astype(np.int32), rng.integers(1, 4, n))
    mov = W.Section(mxy, rng.gamma(0.3, 30.0, (n, T)), rng.integers(0, 2, n).astype(np.int32), rng.integers(1, 4, n))
    return ref, mov


@pytest.mark.parametrize("style", ["generic", "integer_ref", "half_step_x", "lattice_ref", "lattice_moving"])
def test_no_order_tie_means_any_order_gives_the_same_window(style):
    """Qhull's simplices of a staged window against the SAME triangles reordered (rows shuffled, corners permuted, orientation
    included): whenever the library counts no order tie for a reordering, the window's matched rows, per-cell flags and sweep counters
    are those of Qhull's order, bit for bit.  Generic coordinates never count a tie; the degenerate styles do, and some of their
    windows really do come out differently -- those are the ones the count must catch."""
    from scipy.spatial import Delaunay

    from same_amd import windows as W
    from same_amd.triangles import cos_threshold

    rng = np.random.default_rng({"generic": 1, "integer_ref": 2, "half_step_x": 3, "lattice_ref": 4, "lattice_moving": 5}[style])
    ref, mov = _sections(rng, style)
    dref, dmov = W.DeviceSection(ref, "float64"), W.DeviceSection(mov, "float64")
    st = W.DeviceWindow()
    en, thr = cos_threshold(15)
    tied = differed = clean = 0
    x0 = 1000.0 if style == "lattice_ref" else 0.0
    for _ in range(10):
        bx, by = x0 + rng.uniform(0, 150), x0 + rng.uniform(0, 150)
        box = (float(bx), float(bx + rng.uniform(40, 110)), float(by), float(by + rng.uniform(40, 110)))
        n_m, n_r, kept, n_pairs = st.stage(dmov, dref, box, 6.0, 4, 1.0)
        if kept < 10:
            continue
        simplices = Delaunay(st.fetch(W._W_ALIGNED_XY)).simplices
        args = (8.0, en, thr, 0.0, True, 100.0)
        k0, k1, near, m0, f0, s0 = st.filter_finish(simplices, *args)
        assert near == 0
        base_ties = st.order_ties
        kept_set = {tuple(sorted(t)) for t in st.fetch(W._W_TRIANGLES).tolist()}
        for _again in range(3):
            k0b, k1b, near, m1, f1, s1 = st.filter_finish(_reordered(rng, simplices), *args)
            same = (k0b, k1b) == (k0, k1) and np.array_equal(m0, m1) and np.array_equal(f0, f1) and s0 == s1
            same = same and {tuple(sorted(t)) for t in st.fetch(W._W_TRIANGLES).tolist()} == kept_set
            if st.order_ties == 0:
                assert same and base_ties == 0, (style, box)
                clean += 1
            else:
                tied += 1
                differed += not same
    if style == "generic":
        assert clean >= 24 and tied == 0
    else:
        assert tied > 0, style
    if style in ("integer_ref", "lattice_ref"):
        assert differed > 0          # the order does change these windows: the count is what keeps the native route from using them
    st.close()
    dref.close()
    dmov.close()


def _frames(rng, n, side, integer_ref=False):
    T = 4
    rxy = rng.uniform(0, side, (n, 2))
    mxy = rxy[rng.random(n) < 0.93] + rng.normal(0, 1.5, (1, 2))
    mxy = mxy + rng.normal(0, 1.0, mxy.shape)
    if integer_ref:
        rxy = np.round(rxy)
    out = []
    for xy in (rxy, mxy):
        df = pd.DataFrame(rng.gamma(0.3, 30.0, (len(xy), T)), columns=[f"t{q}" for q in range(T)])
        df.insert(0, "Y", xy[:, 1])
        df.insert(0, "X", xy[:, 0])
        df["cell_type"] = rng.choice(np.array(["a", "b", "c"], dtype=object), len(xy))
        df["Cell_Num_Old"] = rng.permutation(len(xy)) * 2 + 5
        out.append(df)
    return out[0], out[1], [f"t{q}" for q in range(T)]


@pytest.mark.parametrize("integer_ref", [False, True])
def test_tables_with_the_native_triangulator_are_the_tables_with_scipy(integer_ref):
    """`sliding_window_incumbent` on resident frames, merged and plain (with the per-window statistics): hip_delaunay = 'native' gives
    the very tables of the default.  Generic coordinates: every window is answered by the library and none goes back to scipy; with
    reference cells on whole coordinates windows count order ties and are finished again with scipy's simplices -- same tables still."""
    import same_amd
    from same_amd import delaunay

    rng = np.random.default_rng(11 + integer_ref)
    r_df, m_df, cols = _frames(rng, 30000, 900.0, integer_ref)
    op = dict(radius=12, knn=6, window_size=200, overlap=40, min_cells_per_window=10, hip_cost_dtype="float32")
    resident = same_amd.resident_frames(r_df, m_df)
    tr = delaunay.shared()
    try:
        for merge in (True, False):
            want = same_amd.sliding_window_incumbent(resident, resident, commonCT=cols, optim_params=dict(op), merge=merge,
                                                     return_stats=True)
            before = (tr.submitted, tr.asked_qhull)
            got = same_amd.sliding_window_incumbent(resident, resident, commonCT=cols, optim_params=dict(op, hip_delaunay="native"),
                                                    merge=merge,
                                                    return_stats=True)
            asked, back = tr.submitted - before[0], tr.asked_qhull - before[1]
            assert len(want[0]) > 15000 and got[0].equals(want[0]) and list(got[0].columns) == list(want[0].columns)
            assert got[1] == want[1] and len(want[1]) >= 25
            assert asked == len(want[1])
            assert (back > 0) if integer_ref else (back == 0), (asked, back)
    finally:
        resident.close()
