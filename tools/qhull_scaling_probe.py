"""How the Qhull helper pool scales with the number of helpers on this host, and whether placing them on distinct physical
cores changes it.  No GPU call.  python3 tools/qhull_scaling_probe.py [points_per_set]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from same_amd import qhull_pool


def cores():
    """One logical CPU per physical core among those this process may run on."""
    allowed = sorted(os.sched_getaffinity(0))
    seen, out = set(), []
    for c in allowed:
        try:
            sib = open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read().strip()
        except OSError:
            sib = str(c)
        if sib not in seen:
            seen.add(sib)
            out.append(c)
    return allowed, out


def run(helpers, sets, pin, pool_places=False):
    pool = qhull_pool.QhullPool(helpers, pin=pool_places)
    warm = [pool.submit(sets[0][:200]) for _ in range(helpers)]
    for t in warm:
        t.result()
    if pin:
        for i, p in enumerate(pool.procs):
            os.sched_setaffinity(p.pid, {pin[i % len(pin)]})
    t0 = time.perf_counter()
    tickets = []
    for i, s in enumerate(sets):
        tickets.append(pool.submit(s))
        if i >= helpers:
            tickets[i - helpers].result()
    for t in tickets:
        t.result()
    dt = time.perf_counter() - t0
    pool.close()
    return dt / len(sets) * 1e3


n = int(sys.argv[1]) if len(sys.argv) > 1 else 13000
rng = np.random.default_rng(0)
sets = [rng.uniform(0, 1200, (n, 2)) for _ in range(96)]
allowed, phys = cores()
try:
    quota = open("/sys/fs/cgroup/cpu.max").read().strip()
except OSError:
    quota = "?"
print(f"# logical CPUs allowed {len(allowed)}, physical cores among them {len(phys)}, cgroup cpu.max '{quota}', {len(sets)} sets of {n} points", flush=True)
from scipy.spatial import Delaunay
t0 = time.perf_counter()
for s in sets[:8]:
    Delaunay(s)
print(f"one Delaunay in this process, alone: {(time.perf_counter() - t0) / 8 * 1e3:.1f} ms", flush=True)
step = max(1, len(phys) // 16)
spread = phys[::step]                      # cores far apart (different CCDs where the numbering allows)
print(f"# L3 domains seen by the pool: {len(qhull_pool._l3_domains())}", flush=True)
for helpers in (4, 8, 12, 16):
    a = run(helpers, sets, None)
    b = run(helpers, sets, None, pool_places=True)
    c = run(helpers, sets, spread)
    print(f"helpers {helpers:2d}: {a:6.2f} ms/set unpinned ({1e3 / a:5.0f}/s) | {b:6.2f} the pool's placement, one L3 domain each ({1e3 / b:5.0f}/s) | "
          f"{c:6.2f} one CPU each, spread", flush=True)
