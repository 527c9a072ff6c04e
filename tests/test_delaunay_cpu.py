"""libsame_hip's own triangulator (csrc/delaunay.cpp, same_amd/delaunay.py) against scipy.spatial.Delaunay -- the call the reference
makes (src/same.py:1023).  Host code: runs without a GPU.  An answer must be scipy's SET of triangles; where the points are too
close to degenerate the answer must be "ask Qhull" (None), never a guess."""
import numpy as np
import pytest
from scipy.spatial import Delaunay

from same_amd import delaunay


def _canonical(tris):
    t = np.sort(np.asarray(tris, np.int64), axis=1)
    return t[np.lexsort((t[:, 2], t[:, 1], t[:, 0]))]


def _point_sets(seed, n_sets):
    rng = np.random.default_rng(seed)
    for _ in range(n_sets):
        n = int(rng.integers(3, 4000))
        off = rng.choice([0.0, 1e3, 1e4, -2e4]) * rng.choice([0, 1, 1], 2)
        kind = int(rng.integers(0, 4))
        if kind == 0:
            pts = rng.uniform(0, 1200, (n, 2))
        elif kind == 1:
            pts = rng.normal(0, 1, (n, 2)) * rng.uniform(1, 300, 2)
        elif kind == 2:
            c = rng.uniform(0, 1200, (max(n // 40, 1), 2))
            pts = c[rng.integers(0, len(c), n)] + rng.normal(0, rng.uniform(0.5, 20), (n, 2))
        else:
            pts = np.c_[rng.uniform(0, 3000, n), rng.uniform(0, 30, n)]
        yield kind, pts + off


def test_answers_are_scipys_triangles():
    answered = 0
    for kind, pts in _point_sets(5, 60):
        got = delaunay.native_simplices(pts)
        if got is None:
            continue
        answered += 1
        want = Delaunay(pts).simplices
        assert got.dtype == np.int32 and got.shape == want.shape, (kind, len(pts))
        assert np.array_equal(_canonical(got), _canonical(want)), (kind, len(pts))
        a, b, c = (pts[got[:, q]] for q in range(3))                # counter-clockwise, every one
        assert ((b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (b[:, 1] - a[:, 1]) * (c[:, 0] - a[:, 0]) > 0).all()
    assert answered >= 50


def test_the_smallest_sets():
    tri = np.array([[0.0, 0.0], [1.0, 0.1], [0.2, 1.0]])
    assert _canonical(delaunay.native_simplices(tri)).tolist() == [[0, 1, 2]]
    quad = np.array([[0.0, 0.0], [2.0, 0.1], [2.1, 1.3], [0.1, 1.0]])
    assert np.array_equal(_canonical(delaunay.native_simplices(quad)), _canonical(Delaunay(quad).simplices))
    for n in (0, 1, 2):
        assert delaunay.native_simplices(np.zeros((n, 2))) is None


@pytest.mark.parametrize("name", ["lattice", "duplicates", "collinear", "circle", "hull_straight", "nan", "far_away"])
def test_degenerate_sets_are_left_to_qhull(name):
    rng = np.random.default_rng(3)
    base = rng.uniform(0, 100, (300, 2))
    if name == "lattice":
        gx, gy = np.meshgrid(np.arange(20.0), np.arange(20.0))
        pts = np.c_[gx.ravel(), gy.ravel()]
    elif name == "duplicates":
        pts = np.vstack([base, base[:5]])
    elif name == "collinear":
        pts = np.c_[np.arange(50.0), 2 * np.arange(50.0) + 1]
    elif name == "circle":
        a = np.arange(64) * (2 * np.pi / 64)
        pts = np.c_[np.cos(a), np.sin(a)] * 50 + 50          # cocircular: every triangulation of the polygon is a Delaunay one
    elif name == "hull_straight":                      # three hull points exactly on a line, the rest inside
        pts = np.vstack([base * 0.5 + 25, [[0.0, 0.0], [50.0, 0.0], [100.0, 0.0], [100.0, 100.0], [0.0, 100.0]]])
    elif name == "nan":
        pts = base.copy()
        pts[7, 1] = np.nan
    else:                                              # so far from the origin that Qhull's lifted coordinate has no digits left
        pts = base + 3e9
    assert delaunay.native_simplices(pts) is None


def test_margin_is_reported_and_guard_is_honoured():
    pts = np.random.default_rng(1).uniform(0, 1000, (2000, 2))
    tris, margin = delaunay.native_simplices(pts, with_margin=True)
    assert tris is not None and margin > delaunay.GUARD
    none, same_margin = delaunay.native_simplices(pts, guard=margin * 2, with_margin=True)
    assert none is None and same_margin == margin


def test_triangulator_tickets_and_fallback(monkeypatch):
    monkeypatch.setenv("SAME_QHULL_WORKERS", "0")      # the fallback asks scipy in this process
    tr = delaunay.NativeTriangulator(threads=3)
    try:
        rng = np.random.default_rng(2)
        sets = [rng.uniform(0, 500, (int(rng.integers(50, 900)), 2)) for _ in range(12)]
        gx, gy = np.meshgrid(np.arange(12.0), np.arange(12.0))
        sets.append(np.c_[gx.ravel(), gy.ravel()])
        tickets = [tr.submit(p, key=q) for q, p in enumerate(sets)]
        for q, (t, p) in enumerate(zip(tickets, sets)):
            got, want = t.result(), Delaunay(p).simplices
            if q < 12:
                assert t.native and np.array_equal(_canonical(got), _canonical(want))
                assert np.array_equal(t.qhull(), want) and not t.native
            else:
                assert not t.native and np.array_equal(got, want)            # the lattice: scipy's own, in scipy's order
        assert tr.submitted == 13 and tr.asked_qhull == 13
    finally:
        tr.close()


def test_mode_switch(monkeypatch):
    monkeypatch.delenv("SAME_DELAUNAY", raising=False)
    assert delaunay.mode({}) == "qhull" and delaunay.mode({"hip_delaunay": "native"}) == "native"
    monkeypatch.setenv("SAME_DELAUNAY", "native")
    assert delaunay.mode(None) == "native" and delaunay.mode({"hip_delaunay": "qhull"}) == "qhull"
    with pytest.raises(ValueError):
        delaunay.mode({"hip_delaunay": "cgal"})


def test_triangulator_steps_aside_where_most_windows_go_back(monkeypatch):
    """Lattices only: after WINDOW tickets that all went to scipy the next ones go to the helper pool directly (`bypassed`), and
    `reset()` ends that."""
    monkeypatch.setenv("SAME_QHULL_WORKERS", "0")
    tr = delaunay.NativeTriangulator(threads=2)
    try:
        gx, gy = np.meshgrid(np.arange(9.0), np.arange(9.0))
        lattice = np.c_[gx.ravel(), gy.ravel()]
        want = Delaunay(lattice).simplices
        for q in range(tr.WINDOW + 5):
            t = tr.submit(lattice)
            assert np.array_equal(t.result(), want) and not t.native
        assert tr.asked_qhull == tr.WINDOW and tr.bypassed == 5
        tr.reset()
        t = tr.submit(np.random.default_rng(0).uniform(0, 100, (200, 2)))
        assert t.result() is not None and t.native and tr.bypassed == 5
    finally:
        tr.close()


def test_triangulator_under_the_sanitizers(tmp_path):
    """csrc/delaunay.cpp built with -fsanitize=address,undefined and driven from C++ (tests/native/delaunay_test.cpp): random and
    degenerate sets, bad arguments, eight threads at once; every answer checked for the local Delaunay property."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "delaunay_test"
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-Wall", "-Werror", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                    "-pthread",
                    "-ffp-contract=off", "-I", os.path.join(root, "include"), os.path.join(root, "same_amd", "csrc", "delaunay.cpp"),
                    os.path.join(root, "tests", "native", "delaunay_test.cpp"), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr


def test_qhull_is_exact_wherever_the_margin_is_clear_of_its_allowance():
    """The model behind the guard, pinned (tools/delaunay_margin.py runs it at scale): sets pushed so far from the origin that Qhull's
    uncentred lifted coordinate runs out of digits.  With guard = 0 the library gives the EXACT triangulation (its signs are computed on
    centred differences) and the set's margin; scipy's triangles must be the exact ones wherever the margin is 1 or more -- the library
    only ever answers above 16 -- and, far enough out, some sets must show scipy erring (or the test is not testing anything)."""
    rng = np.random.default_rng(17)
    clear = erred = 0
    for _ in range(60):
        n = int(rng.integers(50, 1500))
        off = 10.0 ** rng.uniform(4, 8.5) * rng.choice([-1, 1], 2)
        kind = int(rng.integers(0, 3))
        if kind == 0:
            pts = rng.uniform(0, 1200, (n, 2))
        elif kind == 1:
            c = rng.uniform(0, 1200, (max(n // 40, 1), 2))
            pts = c[rng.integers(0, len(c), n)] + rng.normal(0, rng.uniform(0.5, 20), (n, 2))
        else:
            pts = np.c_[rng.uniform(0, 3000, n), rng.uniform(0, 30, n)]
        pts = pts + off
        exact, margin = delaunay.native_simplices(pts, guard=0.0, with_margin=True)
        if exact is None:
            continue
        qhull = Delaunay(pts).simplices
        same = len(exact) == len(qhull) and np.array_equal(_canonical(exact), _canonical(qhull))
        if margin >= 1.0:
            assert same, (margin, n, kind, off)
            clear += 1
        erred += not same
    assert clear >= 15 and erred >= 5, (clear, erred)
