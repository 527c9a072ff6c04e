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


def test_a_reply_stream_out_of_step_retires_the_helper(pool):
    """Stray bytes where a reply should be (wrong magic, another request's number, an implausible count): the answer is
    computed in-process, the helper is REPLACED before it is used again, and later requests are right."""
    import io
    import struct
    from same_amd import qhull_pool

    pts = np.random.default_rng(4).uniform(0, 10, (300, 2))
    want = Delaunay(pts).simplices
    assert np.array_equal(pool.submit(pts).result(), want)
    for junk in (b"Qhull banner: hello\n" * 3,                                            # text on the pipe
                 struct.pack("<iqq", qhull_pool._MAGIC, 999_999, 5) + b"\0" * 60,           # an answer to some other request
                 struct.pack("<iqq", qhull_pool._MAGIC, pool.seq + 1, 10 ** 9)):            # a count no triangulation of 300 points has
        victim = pool.procs[0]
        real = victim.stdout
        victim.stdout = io.BytesIO(junk)
        t = pool.submit(pts)
        assert t.worker == 0
        assert np.array_equal(t.result(), want)
        assert pool.procs[0] is not victim and victim.poll() is not None      # retired, not reused
        real.close()
        assert np.array_equal(pool.submit(pts).result(), want)               # the replacement answers correctly


def test_helper_output_on_stdout_does_not_reach_the_protocol():
    """The worker keeps the protocol on a private copy of its stdout and points fd 1 at stderr: a library that prints there
    cannot shift the framing."""
    import struct
    import subprocess
    import sys
    from same_amd import qhull_pool

    noisy = qhull_pool._WORKER.replace("inp = sys.stdin.buffer", "inp = sys.stdin.buffer\nprint('banner on stdout')\nos.write(1, b'raw bytes on fd 1')")
    p = subprocess.Popen([sys.executable, "-c", noisy], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    pts = np.random.default_rng(5).uniform(0, 10, (50, 2))
    p.stdin.write(struct.pack("<qq", 7, len(pts)) + pts.tobytes() + struct.pack("<qq", 0, -1))
    p.stdin.flush()
    out, err = p.communicate(timeout=60)
    magic, seq, n = struct.unpack("<iqq", out[:20])
    assert magic == qhull_pool._MAGIC and seq == 7 and len(out) == 20 + 12 * n
    assert np.array_equal(np.frombuffer(out[20:], np.int32).reshape(n, 3), Delaunay(pts).simplices)
    assert b"banner on stdout" in err and b"raw bytes on fd 1" in err


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
    assert 1 <= qhull_pool.default_workers() <= 24 and qhull_pool.default_workers() <= 2 * qhull_pool.cpu_budget()


def test_the_cpu_budget_is_divided_by_the_ranks_on_the_host(monkeypatch):
    """Every rank of a job sees the same affinity mask and cgroup quota, so the helper count is the rank's share of the budget:
    world 8 on a 16-CPU budget starts at most 24 helpers in total (it was 8 x 24); the L3 domains are dealt the same way."""
    from same_amd import qhull_pool

    for var in ("SAME_QHULL_WORKERS", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "WORLD_SIZE", "RANK", "SAME_LOCAL_WORLD", "SAME_RDV_DIR", "MASTER_ADDR"):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setattr(qhull_pool, "cpu_budget", lambda: 16)
    assert qhull_pool.local_world() == (1, 0) and qhull_pool.default_workers() == 24
    for world, per_rank in ((2, 12), (4, 6), (8, 3), (32, 1)):
        monkeypatch.setenv("LOCAL_WORLD_SIZE", str(world))
        assert qhull_pool.default_workers() == per_rank and world * per_rank <= max(24, world)
    monkeypatch.delenv("LOCAL_WORLD_SIZE")
    # without LOCAL_WORLD_SIZE: WORLD_SIZE counts only when the rendezvous is visibly on this host
    monkeypatch.setenv("WORLD_SIZE", "8")
    monkeypatch.setenv("RANK", "5")
    assert qhull_pool.local_world() == (1, 0)                                   # could be eight hosts
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    assert qhull_pool.local_world() == (8, 5) and qhull_pool.default_workers() == 3
    monkeypatch.setenv("MASTER_ADDR", "10.0.0.7")
    monkeypatch.setenv("SAME_RDV_DIR", "/tmp/x")                                # bench.py's own launcher: one host by construction
    assert qhull_pool.local_world() == (8, 5)
    monkeypatch.setenv("SAME_LOCAL_WORLD", "2")                                 # explicit override
    assert qhull_pool.local_world()[0] == 2 and qhull_pool.default_workers() == 12
    monkeypatch.setenv("SAME_QHULL_WORKERS", "7")                               # the explicit count is per process, not divided
    assert qhull_pool.default_workers() == 7
    monkeypatch.delenv("SAME_QHULL_WORKERS")
    # A bound launch (slurm --cpu-bind, numactl, per-rank cpusets) gives every rank a mask of its own, which is not to be divided again --
    # but a mask's SIZE does not say so (one container cpuset for all ranks is small too): without evidence the budget is divided, ...
    import os
    import socket
    monkeypatch.setenv("SAME_LOCAL_WORLD", "8")
    monkeypatch.setenv("LOCAL_RANK", "5")
    monkeypatch.setattr(os, "cpu_count", lambda: 128)
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(16)))
    monkeypatch.setattr(qhull_pool, "_LEARNED_SHARERS", None)
    assert qhull_pool.cpu_sharers() == (8, 5) and qhull_pool.default_workers() == 3

    class Group:                       # ... and with the ranks' masks compared (learn_cpu_sharing over the job's host group) it is decided by them
        rank, world = 5, 8

        def __init__(self, masks, hosts=None):
            self.masks, self.hosts = masks, hosts or [socket.gethostname()] * 8

        def allgather_object(self, mine):
            return [mine if r == self.rank else {"host": self.hosts[r], "mask": self.masks[r], "rank": r} for r in range(8)]

    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(80, 96)))
    assert qhull_pool.learn_cpu_sharing(Group([list(range(16 * r, 16 * r + 16)) for r in range(8)])) == (1, 0)      # one slice per rank: bound
    assert qhull_pool.cpu_sharers() == (1, 0) and qhull_pool.default_workers() == 24
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(16)))
    assert qhull_pool.learn_cpu_sharing(Group([list(range(16))] * 8)) == (8, 5)       # ONE 16-CPU cpuset for all eight ranks of a 128-CPU host: shared
    assert qhull_pool.cpu_sharers() == (8, 5) and qhull_pool.default_workers() == 3
    other = ["elsewhere"] * 4 + [socket.gethostname()] * 4                            # the same mask on another host is another host's CPUs
    assert qhull_pool.learn_cpu_sharing(Group([list(range(16))] * 8, other)) == (4, 1)
    pairs = [list(range(32 * (r // 2), 32 * (r // 2) + 32)) for r in range(8)]        # two ranks per NUMA node: each pair shares its node
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(pairs[5]))
    assert qhull_pool.learn_cpu_sharing(Group(pairs)) == (2, 1)
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(128)))         # every rank sees the whole host: shared
    assert qhull_pool.learn_cpu_sharing(Group([list(range(128))] * 8)) == (8, 5) and qhull_pool.default_workers() == 3
    monkeypatch.setenv("SAME_CPU_SHARERS", "2")
    assert qhull_pool.cpu_sharers() == (2, 1) and qhull_pool.default_workers() == 12
    monkeypatch.setattr(qhull_pool, "_LEARNED_SHARERS", None)
    doms = [[c] for c in range(8)]
    assert [qhull_pool._domain_share(doms, 4, r) for r in range(4)] == [[[0], [1]], [[2], [3]], [[4], [5]], [[6], [7]]]
    assert [qhull_pool._domain_share(doms[:2], 4, r) for r in range(4)] == [[[0]], [[0]], [[1]], [[1]]]
    assert qhull_pool._domain_share(doms, 1, 0) == doms and qhull_pool._domain_share([], 8, 3) == []


def test_helpers_are_confined_to_one_cache_domain_each(monkeypatch):
    """Placement (SAME_QHULL_PIN, on by default): helper i may only run on the CPUs of ONE last-level-cache domain, consecutive
    helpers on different ones, the helpers of another local rank shifted; with a single domain, or switched off, nothing is
    pinned.  The simplices do not depend on it."""
    import os

    from same_amd import qhull_pool

    allowed = sorted(os.sched_getaffinity(0))
    real = qhull_pool._l3_domains()
    assert real == [] or sorted(c for d in real for c in d) == allowed          # a partition of what we may run on, or unknown
    assert 1 <= qhull_pool.cpu_budget() <= len(allowed)
    if len(allowed) < 2:
        pytest.skip("one CPU: nothing to place")
    halves = [allowed[: len(allowed) // 2], allowed[len(allowed) // 2:]]
    monkeypatch.setattr(qhull_pool, "_l3_domains", lambda: halves)
    pts = np.random.default_rng(3).uniform(0, 100, (500, 2))
    want = Delaunay(pts).simplices
    for var in ("LOCAL_RANK", "LOCAL_WORLD_SIZE", "WORLD_SIZE", "RANK", "SAME_LOCAL_WORLD", "SAME_RDV_DIR"):
        monkeypatch.delenv(var, raising=False)
    p = qhull_pool.QhullPool(3)                      # one rank on the host: its helpers alternate over both domains
    try:
        tickets = [p.submit(pts) for _ in range(3)]
        assert all(np.array_equal(t.result(), want) for t in tickets)
        assert [sorted(os.sched_getaffinity(proc.pid)) for proc in p.procs] == [halves[i % 2] for i in range(3)]
    finally:
        p.close()
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "2")     # two ranks on the host: each keeps its helpers in its own half
    for local_rank in (0, 1):
        monkeypatch.setenv("LOCAL_RANK", str(local_rank))
        p = qhull_pool.QhullPool(3)
        try:
            tickets = [p.submit(pts) for _ in range(3)]
            assert all(np.array_equal(t.result(), want) for t in tickets)
            assert [sorted(os.sched_getaffinity(proc.pid)) for proc in p.procs] == [halves[local_rank]] * 3
        finally:
            p.close()
    monkeypatch.delenv("LOCAL_WORLD_SIZE")
    monkeypatch.delenv("LOCAL_RANK")
    monkeypatch.setenv("SAME_QHULL_PIN", "0")
    p = qhull_pool.QhullPool(2)
    try:
        assert p.domains == [] and np.array_equal(p.submit(pts).result(), want)
        assert sorted(os.sched_getaffinity(p.procs[0].pid)) == allowed
    finally:
        p.close()
    monkeypatch.delenv("SAME_QHULL_PIN")
    monkeypatch.setattr(qhull_pool, "_l3_domains", lambda: [allowed])
    assert qhull_pool.QhullPool(2).domains == []                                   # one domain: nothing to choose


def test_threads_share_the_pool_without_waiting_on_each_other():
    """Several threads handing over and picking up through one pool (the window loop's worker threads): every ticket gets its own
    simplices, whatever the interleaving; a thread blocked on Qhull does not hold the pool's lock (a thread whose answer is ready
    gets it while another still waits)."""
    import threading
    import time

    from same_amd.qhull_pool import QhullPool

    pool = QhullPool(3)
    rng = np.random.default_rng(9)
    sets = [rng.uniform(0, 500, (int(n), 2)) for n in rng.integers(20, 4000, 40)]
    want = [Delaunay(p).simplices for p in sets]
    bad = []

    def worker(q):
        mine = list(range(q, len(sets), 4))
        tickets = []
        for j in mine:
            tickets.append((j, pool.submit(sets[j])))
            if len(tickets) > 2:
                i, t = tickets.pop(0)
                if not np.array_equal(t.result(), want[i]):
                    bad.append(i)
        for i, t in tickets:
            if not np.array_equal(t.result(), want[i]):
                bad.append(i)

    threads = [threading.Thread(target=worker, args=(q,)) for q in range(4)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not bad
    # the lock is free while a result is awaited: a big set in flight, its thread waiting, and a small one goes through meanwhile
    big = rng.uniform(0, 1000, (150_000, 2))
    slow = pool.submit(big)
    box = {}
    t0 = time.perf_counter()

    def wait_for_it():
        box["n"] = len(slow.result())
        box["dt"] = time.perf_counter() - t0

    waiter = threading.Thread(target=wait_for_it)
    waiter.start()
    time.sleep(0.02)                                       # the waiter is inside result() by now
    small = pool.submit(sets[0])
    assert np.array_equal(small.result(), want[0])
    quick = time.perf_counter() - t0
    waiter.join()
    assert box["n"] > 100_000 and quick < 0.6 * box["dt"], (quick, box["dt"])
    pool.close()


def test_triangulation_cache_remembers_windows_by_id(monkeypatch):
    """windows.TriangulationCache (bench.py's diagnostic pass): the first request of a window goes to the helper pool and its answer is
    kept under (id, number of points, checksum of their coordinates); the second is answered from memory; other points under the same id
    ask the pool again; a window without an id is never remembered.  The simplices are scipy's either way."""
    monkeypatch.setenv("SAME_QHULL_WORKERS", "1")
    from same_amd import qhull_pool
    from same_amd.windows import TriangulationCache

    monkeypatch.setattr(qhull_pool, "_pool", None)
    try:
        rng = np.random.default_rng(11)
        a, b = rng.uniform(0, 50, (300, 2)), rng.uniform(0, 50, (200, 2))
        cache = TriangulationCache()
        first = cache.submit(a, key=7)
        assert not cache.known and np.array_equal(first.result(), Delaunay(a).simplices) and len(cache.known) == 1
        again = cache.submit(a.copy(), key=7)           # the same window again: answered from memory, no helper asked
        assert isinstance(again, TriangulationCache._Ready) and np.array_equal(again.result(), Delaunay(a).simplices)
        # the id alone does not decide: a cache reused with another plan / section whose windows reuse ids must not answer with stale simplices
        other = cache.submit(b, key=7)
        assert not isinstance(other, TriangulationCache._Ready) and np.array_equal(other.result(), Delaunay(b).simplices) and len(cache.known) == 2
        assert np.array_equal(cache.submit(b, key=None).result(), Delaunay(b).simplices) and len(cache.known) == 2
    finally:
        if qhull_pool._pool is not None:
            qhull_pool._pool.close()
        monkeypatch.setattr(qhull_pool, "_pool", None)
