"""The Qhull helper processes (same_amd/qhull_pool.py): simplices identical to the in-process scipy call, errors raised as
scipy raises them, a dead helper replaced, and the no-helper mode."""
import numpy as np
import pytest
from scipy.spatial import Delaunay


@pytest.fixture()
def pool():
    from same_amd.qhull_pool import QhullPool

    p = QhullPool(2)
    yield p
    p.close()


def test_helpers_return_the_in_process_simplices(pool):
    rng = np.random.default_rng(0)
    sets = [rng.uniform(0, 1000, (n, 2)) for n in (4, 50, 3000, 12_000, 7, 900)]
    sets.append(np.stack(np.meshgrid(np.arange(30.0), np.arange(30.0)), -1).reshape(-1, 2))     # co-circular lattice
    tickets = [pool.submit(p) for p in sets]          # more requests than helpers: earlier answers are drained on the way
    for p, t in zip(sets, tickets):
        want = Delaunay(p).simplices
        got = t.result()
        assert got.dtype == want.dtype == np.int32 and np.array_equal(got, want)
        assert t.result() is got                      # a ticket can be asked twice
    assert len(pool.procs) == 2


def test_errors_surface_as_scipy_raises_them(pool):
    from scipy.spatial import QhullError

    for bad in (np.zeros((3, 2)), np.array([[0.0, 0.0], [1.0, 1.0], [2.0, 2.0], [3.0, 3.0]])):   # too few / collinear points
        with pytest.raises((QhullError, ValueError)) as here:
            Delaunay(bad)
        with pytest.raises(type(here.value)):
            pool.submit(bad).result()
    ok = np.random.default_rng(1).uniform(0, 10, (100, 2))
    assert np.array_equal(pool.submit(ok).result(), Delaunay(ok).simplices)      # the helper survived


def test_a_dead_helper_is_replaced(pool):
    pts = np.random.default_rng(2).uniform(0, 10, (500, 2))
    assert np.array_equal(pool.submit(pts).result(), Delaunay(pts).simplices)
    victim = pool.procs[0]
    victim.kill()                                     # exactly the PID this pool started
    victim.wait()
    for _ in range(4):                                # both helpers get used; the dead one answers in-process, then is replaced
        assert np.array_equal(pool.submit(pts).result(), Delaunay(pts).simplices)
    assert all(p.poll() is None for p in pool.procs)


def test_no_helper_mode_and_default_size(monkeypatch):
    from same_amd import qhull_pool

    p = qhull_pool.QhullPool(0)
    pts = np.random.default_rng(3).uniform(0, 10, (200, 2))
    assert np.array_equal(p.submit(pts).result(), Delaunay(pts).simplices) and p.procs == []
    monkeypatch.setenv("SAME_QHULL_WORKERS", "3")
    assert qhull_pool.default_workers() == 3
    monkeypatch.setenv("SAME_QHULL_WORKERS", "0")
    assert qhull_pool.default_workers() == 0
    monkeypatch.delenv("SAME_QHULL_WORKERS")
    assert 0 <= qhull_pool.default_workers() <= 4
