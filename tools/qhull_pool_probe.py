"""Where the Qhull helper pool's main-process time goes: write of the points, wait for the helper, read of the simplices."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from same_amd import qhull_pool

rng = np.random.default_rng(0)
sets = [rng.uniform(0, 1200, (13000, 2)) for _ in range(64)]
pool = qhull_pool.QhullPool(int(sys.argv[1]) if len(sys.argv) > 1 else 8)
orig_read = pool._read
t_read = []


def timed_read(w, ticket):
    t0 = time.perf_counter()
    out = orig_read(w, ticket)
    t_read.append(time.perf_counter() - t0)
    return out


pool._read = timed_read
for gap_ms in (0.0, 7.0):
    t_read.clear()
    t_submit = []
    tickets = []
    t_all = time.perf_counter()
    for i, p in enumerate(sets):
        t0 = time.perf_counter()
        tickets.append(pool.submit(p))
        t_submit.append(time.perf_counter() - t0)
        if gap_ms:
            time.sleep(gap_ms * 1e-3)          # the main process's own work per window
        if i >= pool.n:
            tickets[i - pool.n].result()
    for t in tickets:
        t.result()
    wall = time.perf_counter() - t_all
    print(f"helpers {pool.n}, {gap_ms} ms of other work per window: {wall / len(sets) * 1e3:.2f} ms/window wall; submit mean {np.mean(t_submit) * 1e3:.3f} ms "
          f"(max {np.max(t_submit) * 1e3:.2f}); _read mean {np.mean(t_read) * 1e3:.3f} ms (max {np.max(t_read) * 1e3:.2f}, n={len(t_read)})", flush=True)
pool.close()
